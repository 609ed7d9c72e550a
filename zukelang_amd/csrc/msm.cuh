// Pippenger multi-scalar multiplication for BLS12-381 G1 / G2 on gfx950 -- internal interface.
#pragma once
#include "zk_common.h"

namespace zk {

enum Curve { CURVE_G1 = 0, CURVE_G2 = 1 };

static inline size_t aff_bytes(Curve c) { return c == CURVE_G1 ? 96 : 192; }
static inline size_t xyzz_bytes(Curve c) { return c == CURVE_G1 ? 192 : 384; }

// Signed-digit window layout: nw = floor(255 / c) + 1 windows of c bits (the top window absorbs
// the last carry: scalars are < r < 2^255), digits in [-2^(c-1), 2^(c-1)], buckets 1..2^(c-1).
// fold: the digits are taken from min(s, r - s) < r / 2 < 0.4529 * 2^255 with the sign carried to every digit (resident keys: their points are in the
// prime-order subgroup, (r - s) P = -s P).  The recoding constant K is < 2^(c nw - 1) (1 + 2^-(c-1)), so min(s, r - s) + K < 2^255 and a width that
// DIVIDES 255 (c = 3, 5, 15, 17) needs no window for the last carry: 15 windows at c = 17 instead of 16.
// Two preconditions, both enforced where the bases are built / the scalars enter: (1) the points have order r -- MsmBases::in_subgroup, set only when
// the [r] P = O test ran on them or they were derived on the device from points it ran on; a base set without it never folds, whatever its width;
// (2) the scalars are canonical (< r) -- every entry point that takes scalars checks it (the Fr stage for a witness, zk_groth16_msm_partial_async
// for caller-owned device vectors).
static inline bool msm_fold(uint32_t c, bool precomp, bool in_subgroup) { return precomp && in_subgroup && 255 % c == 0; }
static inline uint32_t msm_windows(uint32_t c, bool fold = false) { return fold && 255 % c == 0 ? 255 / c : 255 / c + 1; }

struct MsmBases {
    Curve curve = CURVE_G1;
    uint64_t n = 0;          // points
    uint32_t c = 0;          // window bits
    uint32_t nw = 0;         // windows
    bool precomp = false;    // table holds 2^(c*j) * P_i for j < nw at [j*n + i]; one bucket set
    bool fold = false;       // digits of min(s, r - s): see msm_windows
    bool in_subgroup = false; // every point is known to have order r (checked at upload, or derived on the device from checked points)
    DevBuf table;            // affine Montgomery points in the 128-byte record layout of ec.cuh (TableLayout): canonical limbs, one cache line per lane and gather
    DevBuf ident;            // one byte per point, 1 = the base is the identity (the sort never files it into a bucket: table entries the accumulate loop meets are genuine points)
};

struct MsmWorkspace {
    DevBuf counts, offsets, cursor, sorted, buckets, head, tail, red, wsum;
    DevBuf worklist;                      // [0] = count, then bucket ids whose fix-up needs a whole workgroup
    DevBuf wgcount;                       // LDS-privatised sort: [workgroup][bucket] counts, then ranks
    uint32_t sort_wgs = 0;               // 0 = global-atomic path
    // more than 2^15 buckets (windows above 16 bits): two-level counting sort -- level 1 files (bucket, reference) records into
    // 512 coarse bins with the LDS-privatised sort above, level 2 sorts every bin by its fine bucket bits in ONE workgroup's LDS
    uint32_t sort_fine_bits = 0;         // 0 = single level
    DevBuf sorted2;                       // level-1 output: 8-byte records (bucket id, reference)
    DevBuf coarse;                        // level-1 counts | offsets | cursor of the coarse bins (3 x (bins + 1) words)
    uint64_t cap_points = 0;
    uint32_t c = 0, nw = 0;
    bool precomp = false;
    Curve curve = CURVE_G1;
    uint32_t nbuckets = 0;      // total bucket slots (windows * 2^(c-1) in classic mode)
    uint32_t chunk = 0;         // sorted entries per accumulate thread
    uint64_t nthreads = 0;
    // batch-affine rounds (resident keys with enough points per bucket): R = ba_rounds halvings before the XYZZ accumulate
    uint32_t ba_rounds = 0;
    uint64_t ba_max_entries = 0;          // upper bound of sorted entries this workspace will see (non-zero scalars x windows)
    DevBuf ba_offs;                       // R arrays of nbuckets + 1 offsets: round r (1-based) at (r - 1) * (nbuckets + 1)
    DevBuf ba_e[2];                       // ping-pong entry arrays of raw affine points: E_1 -> [0], E_2 -> [1], E_3 -> [0], ...
    uint64_t ba_cap[2] = {0, 0};          // their capacities in points
    // what the accumulate and the reduction of the CURRENT product read (set by msm_sort_accumulate*)
    const uint32_t* red_offsets = nullptr;
    uint32_t red_chunk = 0;
};

static inline dim3 grid_for(uint64_t n, unsigned threads) { return dim3((unsigned)((n + threads - 1) / threads)); }

// ---- the counting sort's geometry (msm_sort.hip), shared with the workspace sizing of msm.hip
static constexpr uint32_t MAX_SORT_JOBS = 4;          // MSMs over the same bases per chain of sort launches (blockIdx.y = job)
static constexpr uint32_t SORT_MAX_BUCKETS = 32768;   // buckets whose whole histogram fits one workgroup's LDS (128 KiB): the single-level LDS sort
static constexpr uint32_t SORT_FEW_BINS = 512;        // coarse bins of the two-level sort's first level (2 KiB of LDS)
static constexpr uint32_t SORT_MAX_FINE = 4096;       // fine buckets one workgroup of the second level ranks in LDS

uint32_t msm_auto_window(uint64_t n, bool precomp);
// check_subgroup: also require [r] P = O of every point (proving keys: the reference's of_bytes_exn raises otherwise, curve.ml:199-212)
int msm_bases_from_bytes(MsmBases& b, Curve curve, const uint8_t* host_bytes, uint64_t n, uint32_t c, bool precomp, hipStream_t s, bool check_subgroup = false);
// d_affine: n DENSE affine points (96 / 192 B each); it must stay valid until the stream has run the table build
// in_subgroup: the caller vouches that the points have order r (they were derived on the device from a base set whose in_subgroup is set)
int msm_bases_from_device_affine(MsmBases& b, Curve curve, const void* d_affine, uint64_t n, uint32_t c, bool precomp, hipStream_t s, bool in_subgroup = false);
// base points [lo, lo + count) of the set (window 0 of the table) back in the dense affine format, exact
int msm_bases_dense(const MsmBases& b, uint64_t lo, uint64_t count, void* d_dense, hipStream_t s);
// max_nonzero: upper bound of the non-zero scalars this workspace's products ever carry (0 = all of b.n): sizes the batch-affine buffers
// launch_entries: sorted entries of one whole accumulate launch these workspaces take part in (0 = unknown): picks the chunk length that fills a whole
// number of rounds of the chip's resident lanes
int msm_workspace_alloc(MsmWorkspace& w, const MsmBases& b, uint64_t max_nonzero = 0, uint64_t launch_entries = 0);
// d_scalars: n canonical (non-Montgomery) Fr, 32 B each, on device.  d_result: one XYZZ point.
int msm_run(const MsmBases& b, MsmWorkspace& w, const void* d_scalars, void* d_result_xyzz, hipStream_t s);
// the same in two halves: sort + bucket accumulation per MSM, then ONE chain of reduction launches for up to 8 MSMs
// over the same bases (their workspaces share bucket count and chunk length); outs[i]: one dense XYZZ point each
int msm_sort_accumulate(const MsmBases& b, MsmWorkspace& w, const void* d_scalars, hipStream_t s);
// ... and for up to 4 MSMs over the same bases (different scalar vectors) as one chain of launches
int msm_sort_accumulate_many(const MsmBases& b, MsmWorkspace* const* ws, const void* const* d_scalars, uint32_t count, hipStream_t s);
// its two halves: the sort alone (msm_sort.hip), and the accumulation over sorted references -- ws[i]'s own or those of from[i], a workspace of ANOTHER
// base set with the same geometry (points, window plan, identity flags) that sorted the same scalar vector
int msm_sort_launch(const MsmBases& b, MsmWorkspace* const* ws, const void* const* d_scalars, uint32_t count, hipStream_t s);
int msm_accumulate_sorted(const MsmBases& b, MsmWorkspace* const* ws, MsmWorkspace* const* from, uint32_t count, hipStream_t s);
// may product `b` read the sort of a product over `a` with the same scalars?  (same curve-independent geometry and the same identity flags, compared on the device)
int msm_bases_same_geometry(const MsmBases& a, const MsmBases& b, bool* same, hipStream_t s);
int msm_reduce(const MsmBases& b, MsmWorkspace* const* ws, void* const* outs, uint32_t count, hipStream_t s);
// the same for n1 products in G1 and n2 in G2 together (one window plan): one chain of launches for both curves
int msm_reduce_mixed(const MsmBases* b1, MsmWorkspace* const* ws1, void* const* outs1, uint32_t n1,
                     const MsmBases* b2, MsmWorkspace* const* ws2, void* const* outs2, uint32_t n2, hipStream_t s);
// step 4 of msm_run (msm_acc_g1.hip / msm_acc_g2.hip) for up to 4 MSMs over the same table in one launch (blockIdx.y = job)
static constexpr uint32_t MAX_ACC_JOBS = 4;
// waves per SIMD of the accumulate kernels (msm_acc.cuh compiles to them, msm_workspace_alloc sizes the chunks for whole rounds of them)
// Measured alternatives (round 3, same box A/B at 2^16 and 2^20): G1 at THREE waves fits 168 registers (15 of them spilled once the table entry no
// longer sits in scratch) and is no faster, alone or pipelined -- the chip runs at its power cap while proving (1.3 kW, 2.23 GHz) and the mixed
// addition alone, operands in registers, runs at this kernel's rate (scripts/proto/madd_rate.hip); G2 at ONE wave (293 registers, no spill) is 4 %
// slower alone at 2^20, 13 % at 2^16, and 1 % slower pipelined than two waves with 40 spilled registers.
static constexpr uint32_t ACC_WAVES_G1 = 2, ACC_WAVES_G2 = 2;
struct AccJobs {
    const uint32_t* offsets[MAX_ACC_JOBS];
    const uint32_t* sorted[MAX_ACC_JOBS];
    uint8_t* buckets[MAX_ACC_JOBS];
    uint8_t* head[MAX_ACC_JOBS];
    uint8_t* tail[MAX_ACC_JOBS];
    const uint8_t* pts[MAX_ACC_JOBS];       // raw = true: the entries are affine points (raw limb layout) instead of table references
};
int msm_accumulate_launch(Curve curve, uint64_t nthreads, const void* table, const AccJobs& jobs, uint32_t count, uint32_t nb, uint32_t chunk, hipStream_t s, bool raw = false);
// Batch-affine halving rounds (msm_ba.cuh / msm_ba.hip) between the sort and the accumulate: `rounds` launches over the jobs'
// sorted references, leaving the entry array of the last round in ws[i]->ba_e[(rounds - 1) & 1] and its offsets in ba_offs.
struct MsmWorkspace;
int msm_batch_affine_rounds(const MsmBases& b, MsmWorkspace* const* ws, uint32_t count, uint32_t rounds, hipStream_t s);
// XYZZ (device) -> uncompressed bytes (host); count points
int points_xyzz_to_bytes(Curve curve, const void* d_xyzz, uint64_t count, uint8_t* host_out, hipStream_t s);
// same, asynchronous: device XYZZ -> device bytes (no allocation, no synchronization)
int points_xyzz_to_bytes_dev(Curve curve, const void* d_xyzz, uint64_t count, void* d_bytes, hipStream_t s);
// the points of a proof (n1 <= 8 dense XYZZ in G1, n2 <= 4 in G2) -> uncompressed bytes at d_out + off[i], in one launch
int proof_points_to_bytes_dev(const void* d_g1, uint32_t n1, const uint32_t* off1, const void* d_g2, uint32_t n2, const uint32_t* off2, void* d_out, hipStream_t s);
// bytes (device copy of host encoding) -> affine Montgomery; *d_flag |= 1 not on curve, |= 2 bad encoding
int points_bytes_to_affine(Curve curve, void* d_affine, const void* d_bytes, uint64_t n, int* d_flag, hipStream_t s);
int points_affine_to_bytes(Curve curve, void* d_bytes, const void* d_affine, uint64_t n, hipStream_t s);
// n ZCash-COMPRESSED points (host, 48 / 96 B) -> uncompressed (host, 96 / 192 B): square roots, curve and subgroup checks on the device (msm_points.hip)
int points_decompress(Curve curve, const uint8_t* in, uint64_t n, uint8_t* out, hipStream_t s);
// out[i] = scalars[i] * G (affine Montgomery on device); scalars canonical on device
int fixed_base_mul(Curve curve, void* d_affine_out, const void* d_scalars, uint64_t n, hipStream_t s);
// acc[i] = sum_j parts[j*stride + i] over j < count   (XYZZ, tiny: cross-GPU partial reduction)
int xyzz_sum_columns(Curve curve, void* d_out, const void* d_parts, uint32_t count, uint32_t npoints, hipStream_t s);

}  // namespace zk
