// Pippenger multi-scalar multiplication over BLS12-381 G1 / G2 for gfx950.
//
// Computes the same group element as the reference's left folds of single scalar
// multiplications: G.apply_powers (src/lib/zk/curve.ml:112-118), G.dot / sum_map (:91-103).
//
// MI355X-first pipeline (all integer work; no MFMA):
//   1 count    scalars (32 B, coalesced) -> signed c-bit digits -> histogram of bucket ids
//   2 scan     exclusive prefix sum of the histogram (bucket -> start of its run)
//   3 scatter  point references (index | sign) written bucket-contiguously: a counting sort
//   4 accumulate  the sorted run is cut into equal chunks, one per lane: every lane does the same
//              number of mixed additions XYZZ += affine regardless of the digit distribution
//              (no per-bucket load imbalance); runs inside a chunk go straight to their bucket,
//              runs cut by a chunk border go to per-lane head / tail slots
//   5 fixup    one lane per bucket adds the few border partials of its run
//   6 reduce   sum_b (b+1) * B_b by per-lane running sums over 2^k-bucket slices + one small
//              scalar multiple per slice, then a wave-shuffle-free LDS tree per window
//   7 final    Horner over the windows (classic mode) -> one XYZZ point
// With `precomp` the base table holds 2^(c*j) * P_i for every window j (HBM is 288 GB: a 2^22
// Groth16 key costs 32 GB), all windows share ONE bucket set and step 7's 255 serial doublings
// disappear; this is the mode the resident proving key uses.
#include "ec.cuh"
#include "msm.cuh"
#include "msm_tail.cuh"

#include <stdlib.h>
#include <string.h>
#include <string>
#include <type_traits>

namespace zk {

// ------------------------------------------------------------------ byte <-> Montgomery conversions
// 48 B big-endian <-> 12 little-endian dense words (the plain integer, not Montgomery)
FF_INLINE FpWords fpw_from_be(const uint8_t* p) {
    FpWords r;
    const uint32_t* w = reinterpret_cast<const uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 12; i++) r.w[i] = __builtin_bswap32(w[11 - i]);
    return r;
}
FF_INLINE void fpw_to_be(uint8_t* p, const FpWords& a) {
    uint32_t* w = reinterpret_cast<uint32_t*>(p);
#pragma unroll
    for (int i = 0; i < 12; i++) w[11 - i] = __builtin_bswap32(a.w[i]);
}
FF_INLINE bool fpw_canonical(const FpWords& a) { return words_are_canonical<FpParams>(a.w); }

// G1: x | y ; G2: x1 | x0 | y1 | y0  (ZCash uncompressed)
FF_INLINE int aff_decode(Aff<Fp>& out, const uint8_t* p) {
    uint8_t flags = p[0];
    if (flags & 0x80) return 2;                       // compressed encodings are not accepted here
    if (flags & 0x40) { out = aff_inf<Fp>(); return 0; }
    const FpWords x = fpw_from_be(p), y = fpw_from_be(p + 48);
    if (!fpw_canonical(x) || !fpw_canonical(y)) return 2;
    out = {fp_to_mont(x), fp_to_mont(y)};
    return 0;
}
FF_INLINE int aff_decode(Aff<Fp2>& out, const uint8_t* p) {
    uint8_t flags = p[0];
    if (flags & 0x80) return 2;
    if (flags & 0x40) { out = aff_inf<Fp2>(); return 0; }
    const FpWords x1 = fpw_from_be(p), x0 = fpw_from_be(p + 48), y1 = fpw_from_be(p + 96), y0 = fpw_from_be(p + 144);
    if (!fpw_canonical(x0) || !fpw_canonical(x1) || !fpw_canonical(y0) || !fpw_canonical(y1)) return 2;
    out = {{fp_to_mont(x0), fp_to_mont(x1)}, {fp_to_mont(y0), fp_to_mont(y1)}};
    return 0;
}
FF_INLINE void aff_encode(uint8_t* p, const Aff<Fp>& a) {
    if (aff_is_inf(a)) {
        uint32_t* w = reinterpret_cast<uint32_t*>(p);
        for (int i = 0; i < 24; i++) w[i] = 0;
        p[0] = 0x40;
        return;
    }
    fpw_to_be(p, fp_from_mont(a.x));
    fpw_to_be(p + 48, fp_from_mont(a.y));
}
FF_INLINE void aff_encode(uint8_t* p, const Aff<Fp2>& a) {
    if (aff_is_inf(a)) {
        uint32_t* w = reinterpret_cast<uint32_t*>(p);
        for (int i = 0; i < 48; i++) w[i] = 0;
        p[0] = 0x40;
        return;
    }
    fpw_to_be(p, fp_from_mont(a.x.c1));
    fpw_to_be(p + 48, fp_from_mont(a.x.c0));
    fpw_to_be(p + 96, fp_from_mont(a.y.c1));
    fpw_to_be(p + 144, fp_from_mont(a.y.c0));
}

template <class F> __global__ void k_bytes_to_affine(uint8_t* dst, const uint8_t* src, uint64_t n, int* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8;
    Aff<F> a;
    int rc = aff_decode(a, src + B * i);
    if (rc) { atomicOr(flag, 2); a = aff_inf<F>(); }
    else if (!aff_on_curve(a)) { atomicOr(flag, 1); a = aff_inf<F>(); }
    aff_store<F>(dst + B * i, a);
}
template <class F> __global__ void k_affine_to_bytes(uint8_t* dst, const uint8_t* src, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8;
    aff_encode(dst + B * i, aff_load<F>(src + B * i));
}
template <class F> __global__ void k_xyzz_to_bytes(uint8_t* dst, const uint8_t* src, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8;
    aff_encode(dst + B * i, xyzz_to_aff(xyzz_load<F>(src + 2 * B * i)));
}

// All points of a proof in ONE launch (single-lane conversions, each with its own inversion, side by side instead of
// one after the other): blocks [0, n1) take the G1 points g1[i] -> out + off.g1[i], blocks [n1, n1 + n2) the G2 points.
struct ProofOffsets {
    uint32_t g1[8], g2[4];
};
__global__ __launch_bounds__(64) void k_proof_to_bytes(const uint8_t* g1, uint32_t n1, const uint8_t* g2, ProofOffsets off, uint8_t* out) {
    if (threadIdx.x != 0) return;
    const uint32_t b = blockIdx.x;
    if (b < n1) aff_encode(out + off.g1[b], xyzz_to_aff(xyzz_load<Fp>(g1 + 192 * (size_t)b)));
    else aff_encode(out + off.g2[b - n1], xyzz_to_aff(xyzz_load<Fp2>(g2 + 384 * (size_t)(b - n1))));
}

// ------------------------------------------------------------------ prime-order subgroup check of uploaded KEY points: [r] P = O
// The reference's points come from Bls12_381.G1/G2.of_bytes_exn / of_compressed_bytes_exn (curve.ml:199-212), which raise on a point of the curve
// that lies outside the r-torsion; a key uploaded to the library as raw bytes gets the same treatment here.  Plain double-and-add over the bits of r
// (a compile-time constant: the branch is wave-uniform), out-of-line group operations: ~255 doublings + 127 additions per point, 0.3 s of a 2^20 key.
__device__ static const uint32_t FR_ORDER_BITS[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u, 0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
template <class F> __global__ __launch_bounds__(128) void k_subgroup_check(const uint8_t* __restrict__ dense, uint64_t n, int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8;
    const Aff<F> p = aff_load<F>(dense + B * i);
    if (aff_is_inf(p)) return;
    Xyzz<F> acc = xyzz_from_aff(p);                      // the top bit (254) of r
    for (int b = 253; b >= 0; b--) {
        acc = xyzz_dbl(acc);
        if ((FR_ORDER_BITS[b >> 5] >> (b & 31)) & 1u) xyzz_madd(acc, p);
    }
    if (!xyzz_is_inf(acc)) atomicOr(flag, 4);
}

// ------------------------------------------------------------------ base tables: table[j*n + i] = 2^(c*j) * P_i, j < nw
// `dense` holds the n base points in the dense affine format (what the key arrived as); the table takes them -- and with nw > 1 their
// multiples by 2^(c j) -- in the 128-byte record layout of ec.cuh (TableLayout).
template <class F> __global__ void k_precompute(uint8_t* table, const uint8_t* __restrict__ dense, uint64_t n, uint32_t c, uint32_t nw) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8, TB = TableLayout<F>::ENTRY;
    Aff<F> p = aff_load<F>(dense + B * i);
    tab_store(table + TB * i, p);
    for (uint32_t j = 1; j < nw; j++) {
        Xyzz<F> q = xyzz_dbl_aff(p);
        for (uint32_t k = 1; k < c; k++) q = xyzz_dbl(q);
        p = xyzz_to_aff(q);
        tab_store(table + TB * ((uint64_t)j * n + i), p);
    }
}
// window 0 of a table back in the dense affine format (key derivation, re-sharding, zk_*_pool_points): exact -- the records hold canonical limbs
template <class F> __global__ void k_table_to_dense(uint8_t* __restrict__ dense, const uint8_t* __restrict__ table, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8, TB = TableLayout<F>::ENTRY;
    aff_store<F>(dense + B * i, tab_load((const F*)nullptr, table + TB * i));
}

// flags[i] = 1 iff base i is the identity (its table entries 2^(cj) P are the identity for every window, and only those:
// neither curve has points of even order)
template <class F> __global__ void k_ident_flags(uint8_t* flags, const uint8_t* dense, uint64_t n) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    constexpr int B = FieldOps<F>::WORDS * 8;
    const uint4* q = reinterpret_cast<const uint4*>(dense + B * i);
    uint32_t o = 0;
    for (int k = 0; k < B / 16; k++) { const uint4 x = q[k]; o |= x.x | x.y | x.z | x.w; }
    flags[i] = o == 0 ? 1 : 0;
}

// ------------------------------------------------------------------ digits
// Signed c-bit digits d_j in [-(2^(c-1) - 1), 2^(c-1)] with sum_j d_j 2^(cj) = s.  Adding the constant
// K = sum_j (2^(c-1) - 1) 2^(cj) turns the recoding into plain base-2^c digit extraction:
// d_j = ((s + K) >> cj & mask) - (2^(c-1) - 1), so every (scalar, window) pair is independent and
// gets its own lane: one atomic per lane in flight instead of nw dependent ones.
struct DigitArgs {
    uint64_t n;
    uint32_t c, nw, precomp, nb_per_window;
    uint32_t K[9];         // the recoding constant, 288 bits
    const uint8_t* ident;  // precomp: 1 = base i is the identity: it never enters a bucket (nullptr: no filter)
    uint32_t coarse_shift;  // two-level sort, level 1: histogram / rank by bucket >> coarse_shift and emit (bucket, reference) records
    uint32_t scalar_major;  // LDS sorts: a workgroup owns a range of SCALARS and files all their digits (each scalar is read once per pass,
                            // not once per window: 13-16x less scalar traffic in the two passes); 0: a range of (scalar, window) pairs, window-major
    uint32_t alias_windows; // EXPERIMENT, compiled in only with -DZK_EXPERIMENTS (scripts/table_alias_ab.sh; results WRONG by design): every window
                            // reads window 0's table entries -- same additions and gathers, 1/16 of the table footprint.  Always 0 in the shipped library.
    uint32_t fold;          // digits of min(s, r - s), sign carried to every digit (msm.cuh: msm_windows): bit 31 of the ninth word of a prepared scalar is the sign
};
// scalar i plus the recoding constant (9 words); false: the scalar is zero or its base is the identity -- no digit of it enters a bucket
FF_INLINE bool digits_prepare(const uint32_t* __restrict__ scalars, uint64_t i, const DigitArgs& a, uint32_t s[9]) {
    const uint32_t* sp = scalars + 8 * i;
    uint4 lo = reinterpret_cast<const uint4*>(sp)[0], hi = reinterpret_cast<const uint4*>(sp)[1];
    s[0] = lo.x; s[1] = lo.y; s[2] = lo.z; s[3] = lo.w; s[4] = hi.x; s[5] = hi.y; s[6] = hi.z; s[7] = hi.w; s[8] = 0;
    if ((s[0] | s[1] | s[2] | s[3] | s[4] | s[5] | s[6] | s[7]) == 0) return false;
    if (a.ident && a.ident[i]) return false;
    uint32_t flip = 0;
    if (a.fold) {                                        // wave-uniform
        uint32_t t[8];
        int64_t bw = 0;
#pragma unroll
        for (int k = 0; k < 8; k++) {
            bw += (int64_t)FR_MOD[k] - (int64_t)s[k];
            t[k] = (uint32_t)bw;
            bw >>= 32;
        }
        bool less = false, decided = false;              // t < s, from the top word down (r is odd: t != s)
#pragma unroll
        for (int k = 7; k >= 0; k--) {
            if (!decided && t[k] != s[k]) { less = t[k] < s[k]; decided = true; }
        }
        if (less) {
            flip = 0x80000000u;
#pragma unroll
            for (int k = 0; k < 8; k++) s[k] = t[k];
        }
    }
    uint64_t cy = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {
        cy += (uint64_t)s[k] + a.K[k];
        s[k] = (uint32_t)cy;
        cy >>= 32;
    }
    s[8] |= flip;                                        // c nw <= 276 bits: the ninth word uses 20 bits at most
    return true;
}
FF_INLINE bool digit_at(const uint32_t s[9], uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val);
FF_INLINE bool digit_of(const uint32_t* __restrict__ scalars, uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val) {
    uint32_t s[9];
    return digits_prepare(scalars, i, a, s) && digit_at(s, i, j, a, key, val);
}
FF_INLINE bool digit_at(const uint32_t s[9], uint64_t i, uint32_t j, const DigitArgs& a, uint32_t& key, uint32_t& val) {
    const uint32_t off = j * a.c, w = off >> 5, b = off & 31;
    uint32_t x0 = 0, x1 = 0;
#pragma unroll
    for (int k = 0; k < 9; k++) {        // static indexing keeps the scalar in registers
        if ((int)w == k) x0 = s[k];
        if ((int)w + 1 == k) x1 = k == 8 ? s[k] & 0x7fffffffu : s[k];
    }
    const uint64_t x = ((uint64_t)x1 << 32) | x0;
    const uint32_t e = (uint32_t)(x >> b) & ((1u << a.c) - 1);
    const uint32_t bias = (1u << (a.c - 1)) - 1;
    if (e == bias) return false;                        // digit 0
    const uint32_t below = e < bias ? 1u : 0u;
    const uint32_t d = below ? bias - e : e - bias;      // the digit's magnitude
    const uint32_t neg = below ^ (s[8] >> 31);           // ... its sign, turned round for a folded scalar
    key = (a.precomp ? 0u : j * a.nb_per_window) + (d - 1);
#ifdef ZK_EXPERIMENTS
    val = (uint32_t)(a.precomp && !a.alias_windows ? (uint64_t)j * a.n + i : i) | (neg << 31);
#else
    val = (uint32_t)(a.precomp ? (uint64_t)j * a.n + i : i) | (neg << 31);
#endif
    return true;
}
// Wave-aggregated atomic increment.  Boolean-heavy witnesses put a large share of the digits into ONE bucket
// (scalar 1 = digit 1 of window 0): same-address atomics serialise and the sort of a 2^16 proof went from 1.1 to
// 5.0 ms.  Up to three rounds peel off the key of the wave's first pending lane when at least 8 lanes share it
// (one atomic for all of them, ranks from the ballot); everything else -- all of a uniform input -- does its own atomic.
// The (scalar, window) pairs are laid out WINDOW-major (pair g = window g / n of scalar g % n), so the lanes of a wave
// hold the same window of 64 consecutive scalars -- that is where equal digits sit side by side.
// Returns the slot of this lane's entry (meaningful for the scatter; the count ignores it).
template <class Counter> FF_INLINE uint32_t wave_aggregated_add(Counter* __restrict__ ctr, bool ok, uint32_t key) {
    const uint32_t lane = __lane_id();
    uint32_t pos = 0;
    uint64_t pending = __ballot(ok);
    for (int round = 0; round < 3 && pending; round++) {
        const int leader = __ffsll((unsigned long long)pending) - 1;
        const uint32_t k0 = (uint32_t)__shfl((int)key, leader);
        const uint64_t same = __ballot(ok && key == k0);
        if (__popcll(same) < 8) break;                                   // wave-uniform
        uint32_t base = 0;
        if (lane == (uint32_t)leader) base = atomicAdd(&ctr[k0], (uint32_t)__popcll(same));
        base = (uint32_t)__shfl((int)base, leader);
        if (ok && key == k0) {
            pos = base + (uint32_t)__popcll(same & (((uint64_t)1 << lane) - 1));
            ok = false;
        }
        pending &= ~same;
    }
    if (ok) pos = atomicAdd(&ctr[key], 1u);
    return pos;
}
// The sort kernels serve up to 4 MSMs over the same bases in one launch (blockIdx.y = job): same digit geometry,
// different scalar vectors and buffers.
static constexpr uint32_t MAX_SORT_JOBS = 4;
struct SortJobs {
    const uint32_t* scalars[MAX_SORT_JOBS];
    uint32_t* counts[MAX_SORT_JOBS];
    uint32_t* offsets[MAX_SORT_JOBS];
    uint32_t* cursor[MAX_SORT_JOBS];
    uint32_t* sorted[MAX_SORT_JOBS];
    uint32_t* wgcount[MAX_SORT_JOBS];
    uint2* sorted2[MAX_SORT_JOBS];         // two-level sort: level-1 records
};
__global__ void k_msm_count(SortJobs jobs, DigitArgs a) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ counts = jobs.counts[blockIdx.y];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;      // no early return: the ballots need whole waves
    uint32_t key = 0, val = 0;
    const bool ok = g < a.n * a.nw && digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
    (void)wave_aggregated_add(counts, ok, key);
}
__global__ void k_msm_scatter(SortJobs jobs, DigitArgs a) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ cursor = jobs.cursor[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    uint32_t key = 0, val = 0;
    const bool ok = g < a.n * a.nw && digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
    const uint32_t pos = wave_aggregated_add(cursor, ok, key);
    if (ok) sorted[pos] = val;
}
// ---- LDS-privatised counting sort (one bucket set of <= 2^15 buckets: the resident-key mode)
// Global atomics saturate at a few G/s chip-wide, which made the sort as expensive as the accumulate
// at 2^20.  Each workgroup instead owns a contiguous range of (scalar, window) pairs and histograms it
// in LDS (2^15 counters = 128 KiB of the 160 KiB), writes its column of the [bucket][workgroup] count
// matrix, one exclusive scan over that matrix gives every workgroup its private cursor per bucket, and
// the scatter pass ranks with LDS atomics again.  No global atomic at all, and the bucket offsets
// fall out of the same scan.
static constexpr uint32_t SORT_THREADS = 1024;
static constexpr uint32_t SORT_MAX_BUCKETS = 32768;
// LDS_BINS: counters the kernel reserves -- SORT_MAX_BUCKETS (128 KiB: one workgroup per CU and hardly any LDS left for the accumulate kernels of the
// other proofs in flight) or SORT_FEW_BINS for the 512 coarse bins of the two-level sort's first level (2 KiB: the sort of a 2^20 proof no longer
// evicts the accumulate workgroups from the compute units it runs on)
static constexpr uint32_t SORT_FEW_BINS = 512;
// k_sort_scatter_staged (level 1 of the two-level sort with its records staged through LDS) -- measured at 2^20, same box: its writes 1.14 -> 0.74 GB per proof
// (sort kernels 3.2 -> 2.8 GB as 2 FETCH + WRITE), but the launch is 2 % LONGER (the extra LDS pass and five barriers per tile cost more than the stores
// save) and the pipelined prover loses 0.6-0.9 %: off unless ZK_SORT_COARSE_STAGED=1 (profiles/r04_sort_fine_staged.txt)
static constexpr bool SORT_COARSE_STAGED_DEFAULT = false;
template <uint32_t LDS_BINS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_count_lds(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    uint32_t* __restrict__ wgcount = jobs.wgcount[blockIdx.y];
    __shared__ uint32_t hist[LDS_BINS];
    const uint32_t wg = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) hist[b] = 0;
    __syncthreads();
    if (a.scalar_major) {
        const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);          // per_wg counts scalars here
        for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {                           // whole waves: the aggregated add ballots
            const uint64_t i = i0 + threadIdx.x;
            uint32_t sk[9];
            const bool live = i < hi && digits_prepare(scalars, i, a, sk);
            for (uint32_t j = 0; j < a.nw; j++) {
                uint32_t key = 0, val = 0;
                const bool ok = live && digit_at(sk, i, j, a, key, val);
                (void)wave_aggregated_add(hist, ok, key >> a.coarse_shift);
            }
        }
    } else {
        const uint64_t total = a.n * a.nw, lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, total);
        for (uint64_t g = lo + threadIdx.x; g < hi; g += SORT_THREADS) {
            uint32_t key = 0, val = 0;
            const bool ok = digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
            (void)wave_aggregated_add(hist, ok, key >> a.coarse_shift);
        }
    }
    __syncthreads();
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) wgcount[(uint64_t)wg * nb + b] = hist[b];   // [workgroup][bucket]: coalesced
}
template <uint32_t LDS_BINS>
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter_lds(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    const uint32_t* __restrict__ base = jobs.wgcount[blockIdx.y];
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    __shared__ uint32_t cur[LDS_BINS];
    const uint32_t wg = blockIdx.x;
    for (uint32_t b = threadIdx.x; b < nb; b += SORT_THREADS) cur[b] = offsets[b] + base[(uint64_t)wg * nb + b];
    __syncthreads();
    if (a.scalar_major) {
        const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);
        for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {
            const uint64_t i = i0 + threadIdx.x;
            uint32_t sk[9];
            const bool live = i < hi && digits_prepare(scalars, i, a, sk);
            for (uint32_t j = 0; j < a.nw; j++) {
                uint32_t key = 0, val = 0;
                const bool ok = live && digit_at(sk, i, j, a, key, val);
                const uint32_t pos = wave_aggregated_add(cur, ok, key >> a.coarse_shift);
                if (ok) {
                    if (a.coarse_shift) jobs.sorted2[blockIdx.y][pos] = make_uint2(key, val);
                    else sorted[pos] = val;
                }
            }
        }
        return;
    }
    const uint64_t total = a.n * a.nw, lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, total);
    for (uint64_t g = lo + threadIdx.x; g < hi; g += SORT_THREADS) {
        uint32_t key = 0, val = 0;
        const bool ok = digit_of(scalars, g % a.n, (uint32_t)(g / a.n), a, key, val);
        const uint32_t pos = wave_aggregated_add(cur, ok, key >> a.coarse_shift);
        if (ok) {
            if (a.coarse_shift) jobs.sorted2[blockIdx.y][pos] = make_uint2(key, val);
            else sorted[pos] = val;
        }
    }
}
// Level 1 of the two-level sort with its 8-byte records STAGED through LDS (round 4; ZK_SORT_COARSE_STAGED): the plain scatter stores every record on its own
// and the counters see 1.14 GB written per 2^20 proof for 545 MB of records.  Here the workgroup files the digits of COARSE_STAGE_WINDOWS windows of its 1024
// scalars (<= 8 k records) per tile: rank per coarse bin with LDS atomics, scan the 512 tile counts, lay the records out in bin order in LDS and store them
// from there -- consecutive lanes write the consecutive records of a bin (runs of ~16 = 128 bytes) and the next tile continues every run.  Scalar-major only.
static constexpr uint32_t COARSE_STAGE_WINDOWS = 8, COARSE_STAGE_TILE = SORT_THREADS * COARSE_STAGE_WINDOWS;
__global__ __launch_bounds__(SORT_THREADS) void k_sort_scatter_staged(SortJobs jobs, DigitArgs a, uint64_t per_wg, uint32_t nb) {
    const uint32_t* __restrict__ scalars = jobs.scalars[blockIdx.y];
    const uint32_t* __restrict__ base = jobs.wgcount[blockIdx.y];
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint2* __restrict__ out = jobs.sorted2[blockIdx.y];
    __shared__ uint32_t cur[SORT_FEW_BINS];          // the workgroup's cursor in every coarse bin
    __shared__ uint32_t tcnt[SORT_FEW_BINS];         // per tile: counts, then (count << 16 | exclusive offset inside the tile)
    __shared__ uint32_t wtot[SORT_THREADS / 64];
    __shared__ uint32_t tile_n;
    __shared__ uint2 stage[COARSE_STAGE_TILE];
    const uint32_t wg = blockIdx.x, t = threadIdx.x, lane = t & 63u, wv = t >> 6;
    for (uint32_t b = t; b < nb; b += SORT_THREADS) { cur[b] = offsets[b] + base[(uint64_t)wg * nb + b]; tcnt[b] = 0; }
    __syncthreads();
    const uint64_t lo = (uint64_t)wg * per_wg, hi = min(lo + per_wg, a.n);
    for (uint64_t i0 = lo; i0 < hi; i0 += SORT_THREADS) {                           // whole waves: the aggregated add ballots
        const uint64_t i = i0 + t;
        uint32_t sk[9];
        const bool live = i < hi && digits_prepare(scalars, i, a, sk);
        for (uint32_t j0 = 0; j0 < a.nw; j0 += COARSE_STAGE_WINDOWS) {
            uint32_t key[COARSE_STAGE_WINDOWS], val[COARSE_STAGE_WINDOWS], rk[COARSE_STAGE_WINDOWS];
            uint32_t okm = 0;
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++) {
                key[k] = 0; val[k] = 0;
                const bool ok = live && j0 + k < a.nw && digit_at(sk, i, j0 + k, a, key[k], val[k]);
                rk[k] = wave_aggregated_add(tcnt, ok, key[k] >> a.coarse_shift);
                okm |= (ok ? 1u : 0u) << k;
            }
            __syncthreads();
            {   // exclusive scan of the tile's counts: one thread per bin (nb <= 512 <= SORT_THREADS)
                const uint32_t c = t < nb ? tcnt[t] : 0;
                uint32_t incl = c;
#pragma unroll
                for (int d = 1; d < 64; d <<= 1) {
                    const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                    if (lane >= (uint32_t)d) incl += y;
                }
                if (lane == 63) wtot[wv] = incl;
                __syncthreads();
                uint32_t before = 0;
#pragma unroll
                for (uint32_t w = 0; w < SORT_THREADS / 64; w++) before += w < wv ? wtot[w] : 0u;
                if (t < nb) tcnt[t] = (before + incl - c) | (c << 16);          // offset < 8192 (13 bits) | count <= 8192 (14 bits)
                if (t == SORT_THREADS - 1) tile_n = before + incl;
            }
            __syncthreads();
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++)
                if (okm >> k & 1u) stage[(tcnt[key[k] >> a.coarse_shift] & 0xffffu) + rk[k]] = make_uint2(key[k], val[k]);
            __syncthreads();
            const uint32_t tn = tile_n;
#pragma unroll
            for (uint32_t k = 0; k < COARSE_STAGE_WINDOWS; k++) {          // slot q holds record number (q - offset) of its bin in this tile
                const uint32_t q = k * SORT_THREADS + t;
                if (q < tn) {
                    const uint2 r = stage[q];
                    const uint32_t bin = r.x >> a.coarse_shift;
                    out[cur[bin] + q - (tcnt[bin] & 0xffffu)] = r;
                }
            }
            __syncthreads();
            if (t < nb) { cur[t] += tcnt[t] >> 16; tcnt[t] = 0; }
            __syncthreads();
        }
    }
}
// Column scan of the [workgroup][bucket] count matrix: one lane per bucket walks down the workgroups
// (row-coalesced), turning counts into each workgroup's exclusive rank inside the bucket and leaving
// the bucket totals, which the single-workgroup k_scan below turns into bucket offsets.
__global__ void k_sort_colscan(SortJobs jobs, uint32_t nb, uint32_t nwg) {
    uint32_t* __restrict__ cnt = jobs.wgcount[blockIdx.y];
    uint32_t* __restrict__ totals = jobs.counts[blockIdx.y];
    const uint32_t b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= nb) return;
    uint32_t run = 0;
    for (uint32_t wg0 = 0; wg0 < nwg; wg0 += 8) {          // eight independent loads in flight per lane: the walk is latency, not bandwidth
        uint32_t x[8];
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) x[j] = wg0 + j < nwg ? cnt[(uint64_t)(wg0 + j) * nb + b] : 0u;
#pragma unroll
        for (uint32_t j = 0; j < 8; j++) {
            if (wg0 + j < nwg) cnt[(uint64_t)(wg0 + j) * nb + b] = run;
            run += x[j];
        }
    }
    totals[b] = run;
}

// single workgroup: offsets[k] = sum_{q<k} counts[q], offsets[nb] = total; cursor = offsets.
// The usual geometries (2^12 .. 2^15 counters, a multiple of 4096) run in tiles of 4096: every thread holds one 16-byte vector of each tile
// (up to eight COALESCED loads issued back to back), a tile is scanned with wave shuffles and sixteen wave totals in LDS, and both outputs
// leave as coalesced 16-byte stores.  The general path below walks a contiguous chunk per thread with dependent-latency scalar loads (one
// HBM round trip per counter: 120-150 us per launch on a 2^16 proof's critical path).
__global__ __launch_bounds__(1024) void k_scan(SortJobs jobs, uint32_t nb) {
    const uint32_t* __restrict__ counts = jobs.counts[blockIdx.x];
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.x];
    uint32_t* __restrict__ cursor = jobs.cursor[blockIdx.x];
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    if (nb <= 32768 && (nb & 4095) == 0) {
        const uint32_t tiles = nb >> 12, lane = t & 63u, wv = t >> 6;
        const uint4* __restrict__ src = reinterpret_cast<const uint4*>(counts);
        uint4* __restrict__ o1 = reinterpret_cast<uint4*>(offsets);
        uint4* __restrict__ o2 = reinterpret_cast<uint4*>(cursor);
        uint4 v[8];
#pragma unroll
        for (int j = 0; j < 8; j++) v[j] = (uint32_t)j < tiles ? src[t + 1024u * j] : make_uint4(0, 0, 0, 0);
        uint32_t carry = 0;
#pragma unroll
        for (int j = 0; j < 8; j++) {
            if ((uint32_t)j >= tiles) break;                       // block-uniform
            const uint32_t s4 = v[j].x + v[j].y + v[j].z + v[j].w;
            uint32_t incl = s4;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += y;
            }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            uint32_t before = 0, tot = 0;
#pragma unroll
            for (uint32_t w = 0; w < 16; w++) {
                const uint32_t y = part[w];
                before += w < wv ? y : 0u;
                tot += y;
            }
            __syncthreads();
            uint32_t run = carry + before + incl - s4;
            uint4 o;
            o.x = run; run += v[j].x;
            o.y = run; run += v[j].y;
            o.z = run; run += v[j].z;
            o.w = run;
            o1[t + 1024u * j] = o;
            o2[t + 1024u * j] = o;
            carry += tot;
        }
        if (t == 0) offsets[nb] = carry;
        return;
    }
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(lo + per, nb);
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += counts[k];
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < 1024; d <<= 1) {
        uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = part[t] - s;
    for (uint32_t k = lo; k < hi; k++) {
        offsets[k] = run;
        cursor[k] = run;
        run += counts[k];
    }
    if (t == 1023) offsets[nb] = part[1023];
}

// ---- two-level sort, level 2: ONE workgroup per coarse bin.  The bin's records are consecutive (level 1); the workgroup counts
// their fine bucket bits in LDS, scans the counts (these ARE the final bucket offsets: bin start + exclusive prefix -- no global
// atomic, no separate scan launch), and scatters the references with LDS cursors.  A bin that swallowed a skewed share of the
// digits (boolean-heavy witnesses) is simply a longer loop for its workgroup: the per-record work is a few instructions.
static constexpr uint32_t SORT_MAX_FINE = 4096;
// (scalar, window) pairs per coarse bin from which the second level stages its scatter through LDS (k_sort_fine_staged): measured on the pipelined
// prover -- 2^16 constraints (2 k pairs per bin, a quarter of a tile) -1.3 %, 2^18 (8 k) -0.6 %, 2^20 (c = 20: 27 k) +0.1 % with the lone proof 0.4 ms shorter and
// the sort's un-overlapped time 1.95 -> 1.50 ms, 2^22 (106 k) +1.1 % and 8.0 -> 6.0 ms
static constexpr uint64_t SORT_FINE_STAGED_MIN = 16384;
__global__ __launch_bounds__(SORT_THREADS) void k_sort_fine(SortJobs jobs, uint32_t fine_bits, uint32_t nbins, uint32_t nb) {
    const uint2* __restrict__ rec = jobs.sorted2[blockIdx.y];
    const uint32_t* __restrict__ coff = jobs.cursor[blockIdx.y];          // coarse offsets (level 1 left them in its cursor array)
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    __shared__ uint32_t cnt[SORT_MAX_FINE];
    __shared__ uint32_t part[SORT_THREADS];
    const uint32_t bin = blockIdx.x, t = threadIdx.x, nf = 1u << fine_bits, fm = nf - 1;
    const uint32_t lo = coff[bin], hi = coff[bin + 1];
    for (uint32_t f = t; f < nf; f += SORT_THREADS) cnt[f] = 0;
    __syncthreads();
    // (both passes fetch the record of the NEXT round before ranking the current one: a round is otherwise one HBM round trip long)
    uint32_t kn = lo + t < hi ? rec[lo + t].x : 0;
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {          // whole waves keep the ballots of wave_aggregated_add valid
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint32_t key = kn & fm;
        const uint32_t in = i + SORT_THREADS;
        kn = in < hi ? rec[in].x : 0;
        (void)wave_aggregated_add(cnt, ok, ok ? key : 0);
    }
    __syncthreads();
    // exclusive scan of cnt[0..nf): every thread owns nf / SORT_THREADS consecutive counters (1..4)
    const uint32_t per = (nf + SORT_THREADS - 1) / SORT_THREADS;
    uint32_t s = 0;
    for (uint32_t k = 0; k < per; k++) { const uint32_t f = t * per + k; if (f < nf) s += cnt[f]; }
    part[t] = s;
    __syncthreads();
    for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
        const uint32_t v = t >= d ? part[t - d] : 0;
        __syncthreads();
        part[t] += v;
        __syncthreads();
    }
    uint32_t run = lo + part[t] - s;
    for (uint32_t k = 0; k < per; k++) {
        const uint32_t f = t * per + k;
        if (f < nf) {
            const uint32_t c = cnt[f];
            offsets[(uint64_t)bin * nf + f] = run;
            cnt[f] = run;                                  // becomes the bucket's cursor
            run += c;
        }
    }
    if (bin == nbins - 1 && t == 0) offsets[nb] = hi;
    __syncthreads();
    uint2 rn = lo + t < hi ? rec[lo + t] : make_uint2(0, 0);
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint2 r = rn;
        const uint32_t in = i + SORT_THREADS;
        rn = in < hi ? rec[in] : make_uint2(0, 0);
        const uint32_t pos = wave_aggregated_add(cnt, ok, r.x & fm);
        if (ok) sorted[pos] = r.y;
    }
}

// The same level-2 pass with its scatter STAGED through LDS (round 4; ZK_SORT_FINE_STAGED): the plain form stores every 4-byte reference on its own,
// scattered over the bin's ~130 KB output range -- the counters see 1.53 GB written per 2^20 proof for 272 MB of references (lines leave the L2 half
// written).  Here a workgroup takes its bin in tiles of 8 k records, ranks the tile's records per fine bucket (LDS atomics), scans the tile's counts,
// lays the references out in bucket order in LDS and writes them from there: consecutive threads store the consecutive references of a bucket (runs of
// ~8 = 32 bytes at 2^20 and 2^22, where c = 20 leaves 1024 fine buckets per bin, and the next tile continues every run).  Counters at 2^20: this kernel's writes 1.53 GB -> 0.51 GB per proof, all
// kernels' 5.50 -> 4.47 GB (profiles/r04_sort_fine_staged.txt).  Counting pass, scan and offsets as in k_sort_fine.
static constexpr uint32_t SORT_TILE_PER_THREAD = 8, SORT_TILE = SORT_THREADS * SORT_TILE_PER_THREAD;
static inline size_t sort_fine_staged_lds(uint32_t fine_bits) { return 4 * ((size_t)2 << fine_bits); }          // cnt + tcnt
__global__ __launch_bounds__(SORT_THREADS) void k_sort_fine_staged(SortJobs jobs, uint32_t fine_bits, uint32_t nbins, uint32_t nb) {
    const uint2* __restrict__ rec = jobs.sorted2[blockIdx.y];
    const uint32_t* __restrict__ coff = jobs.cursor[blockIdx.y];
    uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    extern __shared__ uint32_t sort_dyn[];
    const uint32_t bin = blockIdx.x, t = threadIdx.x, nf = 1u << fine_bits, fm = nf - 1, lane = t & 63u, wv = t >> 6;
    uint32_t* cnt = sort_dyn;                        // pass 1: counts; then every bucket's cursor in the output
    uint32_t* tcnt = sort_dyn + nf;                  // per tile: counts, then (count << 16 | exclusive offset inside the tile)
    __shared__ uint32_t part[SORT_THREADS];
    __shared__ uint32_t st_val[SORT_TILE];           // the tile's references in bucket order ...
    __shared__ uint16_t st_key[SORT_TILE];           // ... and the bucket of each
    const uint32_t lo = coff[bin], hi = coff[bin + 1];
    for (uint32_t f = t; f < nf; f += SORT_THREADS) { cnt[f] = 0; tcnt[f] = 0; }
    __syncthreads();
    uint32_t kn = lo + t < hi ? rec[lo + t].x : 0;
    for (uint32_t base = lo; base < hi; base += SORT_THREADS) {
        const uint32_t i = base + t;
        const bool ok = i < hi;
        const uint32_t key = kn & fm;
        const uint32_t in = i + SORT_THREADS;
        kn = in < hi ? rec[in].x : 0;
        (void)wave_aggregated_add(cnt, ok, ok ? key : 0);
    }
    __syncthreads();
    const uint32_t per = (nf + SORT_THREADS - 1) / SORT_THREADS;          // <= 4 (SORT_MAX_FINE)
    {
        uint32_t s = 0;
        for (uint32_t k = 0; k < per; k++) { const uint32_t f = t * per + k; if (f < nf) s += cnt[f]; }
        part[t] = s;
        __syncthreads();
        for (uint32_t d = 1; d < SORT_THREADS; d <<= 1) {
            const uint32_t v = t >= d ? part[t - d] : 0;
            __syncthreads();
            part[t] += v;
            __syncthreads();
        }
        uint32_t run = lo + part[t] - s;
        for (uint32_t k = 0; k < per; k++) {
            const uint32_t f = t * per + k;
            if (f < nf) {
                const uint32_t c = cnt[f];
                offsets[(uint64_t)bin * nf + f] = run;
                cnt[f] = run;
                run += c;
            }
        }
        if (bin == nbins - 1 && t == 0) offsets[nb] = hi;
    }
    __syncthreads();
    for (uint32_t base = lo; base < hi; base += SORT_TILE) {
        const uint32_t tile = hi - base < SORT_TILE ? hi - base : SORT_TILE;
        uint32_t f[SORT_TILE_PER_THREAD], v[SORT_TILE_PER_THREAD], rk[SORT_TILE_PER_THREAD];
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {          // whole waves: the aggregated add ballots
            const uint32_t j = k * SORT_THREADS + t;
            const bool ok = j < tile;
            const uint2 r = ok ? rec[base + j] : make_uint2(0, 0);
            f[k] = r.x & fm;
            v[k] = r.y;
            rk[k] = wave_aggregated_add(tcnt, ok, ok ? f[k] : 0);
        }
        __syncthreads();
        {   // exclusive scan of the tile's counts (each thread owns `per` consecutive buckets): wave shuffles + sixteen wave totals
            uint32_t c[4], s = 0;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t ff = t * per + k;
                c[k] = k < per && ff < nf ? tcnt[ff] : 0;
                s += c[k];
            }
            uint32_t incl = s;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t y = (uint32_t)__shfl_up((int)incl, d);
                if (lane >= (uint32_t)d) incl += y;
            }
            if (lane == 63) part[wv] = incl;
            __syncthreads();
            uint32_t before = 0;
#pragma unroll
            for (uint32_t w = 0; w < SORT_THREADS / 64; w++) before += w < wv ? part[w] : 0u;
            uint32_t run = before + incl - s;
#pragma unroll
            for (uint32_t k = 0; k < 4; k++) {
                const uint32_t ff = t * per + k;
                if (k < per && ff < nf) {
                    tcnt[ff] = run | (c[k] << 16);          // offset in the tile (< 8192: 13 bits) | the tile's count of this bucket (<= 8192: 14 bits)
                    run += c[k];
                }
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {
            const uint32_t j = k * SORT_THREADS + t;
            if (j < tile) {
                const uint32_t slot = (tcnt[f[k]] & 0xffffu) + rk[k];
                st_val[slot] = v[k];
                st_key[slot] = (uint16_t)f[k];
            }
        }
        __syncthreads();
#pragma unroll
        for (uint32_t k = 0; k < SORT_TILE_PER_THREAD; k++) {          // slot j holds reference number (j - offset) of its bucket in this tile
            const uint32_t j = k * SORT_THREADS + t;
            if (j < tile) {
                const uint32_t ff = st_key[j];
                sorted[cnt[ff] + j - (tcnt[ff] & 0xffffu)] = st_val[j];
            }
        }
        __syncthreads();
        for (uint32_t ff = t; ff < nf; ff += SORT_THREADS) {
            cnt[ff] += tcnt[ff] >> 16;
            tcnt[ff] = 0;
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------ accumulate: msm_acc_g1.hip / msm_acc_g2.hip; fix-up of the chunk partials and
// digit sums of the bucket reduction (one lane, or lane pair, per point): msm_red.hip -- a translation unit of its own, with the field products
// expanded in place
static constexpr uint32_t BA_FINISH_CHUNK = 8;       // sorted entries per lane of the XYZZ accumulate that follows the batch-affine rounds
// acc[i] = sum_j parts[j * npoints + i] on dense XYZZ points: the sum of the ranks' / devices' partial sums of a proof.  One lane per point in G1, a lane PAIR
// in G2 (F = Fp2H), additions expanded in place on the lane's registers: round 3's form (a whole Fp2 point per lane through the out-of-line addition)
// carried 3 KiB of private memory per lane -- 1.6 GiB of scratch reserved on every queue the kernel was dispatched on.
template <class F> __global__ __launch_bounds__(64) void k_xyzz_sum_columns(uint8_t* out, const uint8_t* parts, uint32_t count, uint32_t npoints) {
    constexpr int XB = FieldOps<F>::WORDS * 16;          // dense XYZZ bytes of one point: 192 (G1) / 384 (G2: FieldOps<Fp2H> keeps Fp2's memory layout)
    const uint32_t i = (blockIdx.x * blockDim.x + threadIdx.x) / RawLayout<F>::LANES;
    if (i >= npoints) return;          // both lanes of a pair together
    Xyzz<F> acc = xyzz_inf<F>();
    for (uint32_t j = 0; j < count; j++) {
        const Xyzz<F> q = xyzz_load<F>(parts + (uint64_t)XB * ((uint64_t)j * npoints + i));
        xyzz_add_impl(acc, q);
    }
    xyzz_store<F>(out + (uint64_t)XB * i, acc);
}

// ------------------------------------------------------------------ fixed-base: out[i] = s_i * G
// pow2[k] = 2^k * G (affine), 256 entries per curve, built once by 256 lanes.
template <class F> FF_INLINE Aff<F> generator();
template <> FF_INLINE Aff<Fp> generator<Fp>() {
    // canonical generator coordinates as little-endian words, converted to Montgomery
    const FpWords x = {{0xdb22c6bbu, 0xfb3af00au, 0xf97a1aefu, 0x6c55e83fu, 0x171bac58u, 0xa14e3a3fu,
                        0x9774b905u, 0xc3688c4fu, 0x4fa9ac0fu, 0x2695638cu, 0x3197d794u, 0x17f1d3a7u}};
    const FpWords y = {{0x46c5e7e1u, 0x0caa2329u, 0xa2888ae4u, 0xd03cc744u, 0x2c04b3edu, 0x00db18cbu,
                        0xd5d00af6u, 0xfcf5e095u, 0x741d8ae4u, 0xa09e30edu, 0xe3aaa0f1u, 0x08b3f481u}};
    return {fp_to_mont(x), fp_to_mont(y)};
}
template <> FF_INLINE Aff<Fp2> generator<Fp2>() {
    const FpWords x0 = {{0xc121bdb8u, 0xd48056c8u, 0xa805bbefu, 0x0bac0326u, 0x7ae3d177u, 0xb4510b64u,
                         0xfa403b02u, 0xc6e47ad4u, 0x2dc51051u, 0x26080527u, 0xf08f0a91u, 0x024aa2b2u}};
    const FpWords x1 = {{0x5d042b7eu, 0xe5ac7d05u, 0x13945d57u, 0x334cf112u, 0xdc7f5049u, 0xb5da61bbu,
                         0x9920b61au, 0x596bd0d0u, 0x88274f65u, 0x7dacd3a0u, 0x52719f60u, 0x13e02b60u}};
    const FpWords y0 = {{0x08b82801u, 0xe1935486u, 0x3baca289u, 0x923ac9ccu, 0x5160d12cu, 0x6d429a69u,
                         0x8cbdd3a7u, 0xadfd9baau, 0xda2e351au, 0x8cc9cdc6u, 0x727d6e11u, 0x0ce5d527u}};
    const FpWords y1 = {{0xf05f79beu, 0xaaa9075fu, 0x5cec1da1u, 0x3f370d27u, 0x572e99abu, 0x267492abu,
                         0x85a763afu, 0xcb3e287eu, 0x2bc28b99u, 0x32acd2b0u, 0x2ea734ccu, 0x0606c4a0u}};
    return {{fp_to_mont(x0), fp_to_mont(x1)}, {fp_to_mont(y0), fp_to_mont(y1)}};
}
template <class F> __global__ void k_gen_pow2_table(uint8_t* table) {
    constexpr int AB = FieldOps<F>::WORDS * 8;
    uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= 256) return;
    Aff<F> g = generator<F>();
    Xyzz<F> q = xyzz_from_aff(g);
    for (uint32_t i = 0; i < k; i++) q = xyzz_dbl(q);
    aff_store<F>(table + AB * k, xyzz_to_aff(q));
}
template <class F>
__global__ __launch_bounds__(128) void k_fixed_base_mul(uint8_t* __restrict__ out, const uint32_t* __restrict__ scalars,
                                                        uint64_t n, const uint8_t* __restrict__ pow2) {
    constexpr int AB = FieldOps<F>::WORDS * 8;
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Xyzz<F> acc = xyzz_inf<F>();
    for (uint32_t w = 0; w < 8; w++) {
        uint32_t bits = scalars[8 * i + w];
        while (bits) {
            uint32_t b = __builtin_ctz(bits);
            bits &= bits - 1;
            Aff<F> p = aff_load<F>(pow2 + AB * (32 * w + b));
            xyzz_madd(acc, p);
        }
    }
    aff_store<F>(out + AB * i, xyzz_to_aff(acc));
}

// Work counters of one product's accumulate launch (profiling level 2 only: bench.py's ALU model counts the additions that really run).
// out[0] = sorted entries, out[1] = non-empty buckets, out[2] = runs that start on a chunk border, out[3] = runs that start on step 1 of a chunk.
__global__ void k_acc_stats(const uint32_t* __restrict__ offsets, uint32_t nb, uint32_t chunk, unsigned long long* __restrict__ out) {
    const uint32_t k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k == 0) atomicAdd(&out[0], (unsigned long long)offsets[nb]);
    if (k >= nb) return;
    const uint32_t s = offsets[k], e = offsets[k + 1];
    if (e == s) return;
    atomicAdd(&out[1], 1ull);
    if (s % chunk == 0) atomicAdd(&out[2], 1ull);
    if (s % chunk == 1) atomicAdd(&out[3], 1ull);
}

// ================================================================== host side
static inline dim3 grid_for(uint64_t n, unsigned threads) { return dim3((unsigned)((n + threads - 1) / threads)); }

uint32_t msm_auto_window(uint64_t n, bool precomp) {
    // work ~ nw * n additions + buckets * (2 reduce additions); pick the c minimizing it, capped so the
    // accumulate pass still has >= ~128k lanes of work
    double best = 1e300;
    uint32_t bc = 8;
    // resident keys: at most 2^15 buckets, so the whole histogram of the counting sort fits in LDS
    const uint32_t cmax = precomp ? 16 : 20;
    for (uint32_t c = 6; c <= cmax; c++) {
        double nw = msm_windows(c);          // the plain count: the choice per size stays the measured one (a folded width then runs with one window fewer)
        double buckets = (precomp ? 1.0 : nw) * (double)(1u << (c - 1));
        double cost = nw * (double)n + 3.0 * buckets;
        if (cost < best) { best = cost; bc = c; }
    }
    return bc;
}

template <class F> static int bases_finish(MsmBases& b, const void* d_dense, hipStream_t s) {
    // every base set carries its identity flags: the sort never files an identity base into a bucket, so the accumulate loop can take table
    // entries for genuine points (no identity test per addition) in BOTH table modes
    ZKCHK(b.ident.alloc(b.n));
    hipLaunchKernelGGL(k_ident_flags<F>, grid_for(b.n, 256), dim3(256), 0, s, b.ident.as<uint8_t>(), (const uint8_t*)d_dense, b.n);
    ScopedTimer t("msm_precompute", s);
    hipLaunchKernelGGL(k_precompute<F>, grid_for(b.n, 64), dim3(64), 0, s, b.table.as<uint8_t>(), (const uint8_t*)d_dense, b.n, b.c, b.precomp ? b.nw : 1u);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
size_t table_entry_bytes(Curve c) { return c == CURVE_G1 ? TableLayout<Fp>::ENTRY : TableLayout<Fp2>::ENTRY; }
int msm_bases_dense(const MsmBases& b, uint64_t lo, uint64_t count, void* d_dense, hipStream_t s) {
    if (lo + count > b.n) ZK_FAIL(ZK_ERR_ARG, "msm_bases_dense: range outside the base set");
    if (!count) return ZK_OK;
    const uint8_t* src = b.table.as<uint8_t>() + table_entry_bytes(b.curve) * lo;
    if (b.curve == CURVE_G1) hipLaunchKernelGGL(k_table_to_dense<Fp>, grid_for(count, 128), dim3(128), 0, s, (uint8_t*)d_dense, src, count);
    else hipLaunchKernelGGL(k_table_to_dense<Fp2>, grid_for(count, 128), dim3(128), 0, s, (uint8_t*)d_dense, src, count);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
static int bases_setup(MsmBases& b, Curve curve, uint64_t n, uint32_t c, bool precomp, bool in_subgroup) {
    if (n == 0) ZK_FAIL(ZK_ERR_ARG, "msm: empty base set");
    if (c == 0) {
        const char* e = ::zk::opt("ZK_MSM_WINDOW");         // window-size sweeps (BASELINE config 3); key set-up, not a per-proof path
        c = e ? (uint32_t)atoi(e) : msm_auto_window(n, precomp);
    }
    if (c < 2 || c > 22) ZK_FAIL(ZK_ERR_ARG, "msm: window_bits must be in [2, 22]");
    b.curve = curve; b.n = n; b.c = c; b.precomp = precomp; b.in_subgroup = in_subgroup; b.fold = msm_fold(c, precomp, in_subgroup); b.nw = msm_windows(c, b.fold);
    if ((precomp ? (uint64_t)b.nw : 1) * n >= ((uint64_t)1 << 31)) ZK_FAIL(ZK_ERR_ARG, "msm: too many points for 31-bit references");
    return b.table.alloc(table_entry_bytes(curve) * n * (precomp ? b.nw : 1));
}
int points_bytes_to_affine(Curve curve, void* d_aff, const void* d_bytes, uint64_t n, int* d_flag, hipStream_t s) {
    if (!n) return ZK_OK;
    if (curve == CURVE_G1) hipLaunchKernelGGL(k_bytes_to_affine<Fp>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_aff, (const uint8_t*)d_bytes, n, d_flag);
    else hipLaunchKernelGGL(k_bytes_to_affine<Fp2>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_aff, (const uint8_t*)d_bytes, n, d_flag);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int points_affine_to_bytes(Curve curve, void* d_bytes, const void* d_aff, uint64_t n, hipStream_t s) {
    if (!n) return ZK_OK;
    if (curve == CURVE_G1) hipLaunchKernelGGL(k_affine_to_bytes<Fp>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_bytes, (const uint8_t*)d_aff, n);
    else hipLaunchKernelGGL(k_affine_to_bytes<Fp2>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_bytes, (const uint8_t*)d_aff, n);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int points_xyzz_to_bytes_dev(Curve curve, const void* d_xyzz, uint64_t count, void* d_bytes, hipStream_t s) {
    if (curve == CURVE_G1) hipLaunchKernelGGL(k_xyzz_to_bytes<Fp>, grid_for(count, 64), dim3(64), 0, s, (uint8_t*)d_bytes, (const uint8_t*)d_xyzz, count);
    else hipLaunchKernelGGL(k_xyzz_to_bytes<Fp2>, grid_for(count, 64), dim3(64), 0, s, (uint8_t*)d_bytes, (const uint8_t*)d_xyzz, count);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int proof_points_to_bytes_dev(const void* d_g1, uint32_t n1, const uint32_t* off1, const void* d_g2, uint32_t n2, const uint32_t* off2, void* d_out, hipStream_t s) {
    if (n1 > 8 || n2 > 4) ZK_FAIL(ZK_ERR_ARG, "proof_points_to_bytes_dev: at most 8 G1 and 4 G2 points");
    ProofOffsets off{};
    for (uint32_t i = 0; i < n1; i++) off.g1[i] = off1[i];
    for (uint32_t i = 0; i < n2; i++) off.g2[i] = off2[i];
    ScopedTimer t("proof_to_bytes", s);
    hipLaunchKernelGGL(k_proof_to_bytes, dim3(n1 + n2), dim3(64), 0, s, (const uint8_t*)d_g1, n1, (const uint8_t*)d_g2, off, (uint8_t*)d_out);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int points_xyzz_to_bytes(Curve curve, const void* d_xyzz, uint64_t count, uint8_t* host_out, hipStream_t s) {
    DevBuf tmp;
    ZKCHK(tmp.alloc(aff_bytes(curve) * count));
    ZKCHK(points_xyzz_to_bytes_dev(curve, d_xyzz, count, tmp.p, s));
    HIPCHK(hipMemcpyAsync(host_out, tmp.p, aff_bytes(curve) * count, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    return ZK_OK;
}
int msm_bases_from_device_affine(MsmBases& b, Curve curve, const void* d_affine, uint64_t n, uint32_t c, bool precomp, hipStream_t s, bool in_subgroup) {
    ZKCHK(bases_setup(b, curve, n, c, precomp, in_subgroup));
    return curve == CURVE_G1 ? bases_finish<Fp>(b, d_affine, s) : bases_finish<Fp2>(b, d_affine, s);
}
int msm_bases_from_bytes(MsmBases& b, Curve curve, const uint8_t* host_bytes, uint64_t n, uint32_t c, bool precomp, hipStream_t s, bool check_subgroup) {
    ZKCHK(bases_setup(b, curve, n, c, precomp, check_subgroup));      // folded digits only for points the [r] P = O test below has passed
    DevBuf raw, dense, flag;
    ZKCHK(raw.alloc(aff_bytes(curve) * n));
    ZKCHK(dense.alloc(aff_bytes(curve) * n));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(raw.p, host_bytes, aff_bytes(curve) * n, hipMemcpyHostToDevice, s));
    ZKCHK(points_bytes_to_affine(curve, dense.p, raw.p, n, flag.as<int>(), s));
    if (check_subgroup) {
        ScopedTimer t("subgroup_check", s);
        if (curve == CURVE_G1) hipLaunchKernelGGL(k_subgroup_check<Fp>, grid_for(n, 128), dim3(128), 0, s, (const uint8_t*)dense.as<uint8_t>(), n, flag.as<int>());
        else hipLaunchKernelGGL(k_subgroup_check<Fp2>, grid_for(n, 128), dim3(128), 0, s, (const uint8_t*)dense.as<uint8_t>(), n, flag.as<int>());
    }
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h & 2) ZK_FAIL(ZK_ERR_ARG, "point encoding: compressed flag set or coordinate >= p");
    if (h & 1) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "a base point is not on the curve");
    if (h & 4) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "a key point is on the curve but outside the prime-order subgroup (of_bytes_exn, curve.ml:199-212)");
    ZKCHK((curve == CURVE_G1 ? bases_finish<Fp>(b, dense.p, s) : bases_finish<Fp2>(b, dense.p, s)));
    HIPCHK(hipStreamSynchronize(s));          // `dense` is released on return: the table build has read it
    return ZK_OK;
}

// Batch-affine rounds before the XYZZ accumulate: halve the runs until ~2-3 entries per bucket are left (uniform digits; the
// XYZZ finisher takes whatever a skewed bucket still holds).  Resident keys only (one bucket set).
static uint32_t ba_rounds_for(const MsmBases& b, uint32_t nbuckets) {
    if (!b.precomp) return 0;
    // ZK_MSM_BA_ROUNDS: 0 = off, k > 0 = exactly k rounds (tests, experiments), unset = auto for the curves ZK_MSM_BA_CURVES names
    // (bit 0: G1, bit 1: G2).  Default: none -- measured on MI355X: the rounds re-read what they add from HBM, ~1 KB
    // per G1 addition in round 0 (two random 128-byte-line gathers of each table entry), and lose to the register-resident XYZZ
    // accumulate for G1; for G2 they break even.  Read per workspace, not cached: tests switch it.
    const int forced = ::zk::opt("ZK_MSM_BA_ROUNDS") ? atoi(::zk::opt("ZK_MSM_BA_ROUNDS")) : -1;
    if (forced == 0) return 0;
    if (forced > 0) return forced > 24 ? 24 : (uint32_t)forced;
    const int curves = ::zk::opt("ZK_MSM_BA_CURVES") ? atoi(::zk::opt("ZK_MSM_BA_CURVES")) : 0;
    if (!(curves & (b.curve == CURVE_G1 ? 1 : 2))) return 0;
    const uint64_t mean = b.n * b.nw / nbuckets;
    uint32_t r = 0;
    while (((uint64_t)8 << r) <= mean) r++;                 // mean in [8, 16) -> 1 round ... [2^(k+2), 2^(k+3)) -> k rounds
    return r > 24 ? 24 : r;
}
int msm_workspace_alloc(MsmWorkspace& w, const MsmBases& b, uint64_t max_nonzero, uint64_t launch_entries) {
    w.curve = b.curve; w.c = b.c; w.nw = b.nw; w.precomp = b.precomp; w.cap_points = b.n;
    const uint32_t nbw = 1u << (b.c - 1);
    w.nbuckets = (b.precomp ? 1 : b.nw) * nbw;
    const uint64_t maxN = b.n * b.nw;
    // >= 2 waves per SIMD when there is enough work; chunks of at least 16 entries (with a dozen proofs in flight the
    // chip is full anyway, and every chunk boundary costs a partial sum to store and to fix up: 8 -> 16 is +2.5 % proofs/s
    // at 2^16, 16..24 measure the same, 32 and up lose to the tail of the longest chunk)
    static const uint64_t target_threads = ::zk::opt("ZK_MSM_TARGET_THREADS") ? (uint64_t)atoll(::zk::opt("ZK_MSM_TARGET_THREADS")) : 256 * 1024;   // tuning knob
    uint32_t chunk = (uint32_t)((maxN + target_threads - 1) / target_threads);
    static const uint32_t chunk_min = ::zk::opt("ZK_MSM_CHUNK_MIN") ? (uint32_t)atoi(::zk::opt("ZK_MSM_CHUNK_MIN")) : 16;   // tuning knob
    if (launch_entries && chunk >= 64) {            // short chunks (small keys) already fill whole rounds: at 2^16, 2 rounds of 16-entry chunks beat 1 round of 32; 48 instead of 64 (the G2 product of a 2^20 key: one round of 208 instead of four of 52) measures the same
        // The caller knows how many sorted entries ONE accumulate launch carries (several products over these bases, e.g. Groth16's A and C):
        // cut it into a WHOLE number of rounds of the chip's resident lanes (256 CUs x 4 SIMDs x ACC_WAVES_G1 = ACC_WAVES_G2 = 2 waves x 64 lanes, half as many
        // chunks for the lane pairs of G2).  Every lane does the same number of additions, so a launch whose chunks fill 2.65 rounds costs three (2^20
        // constraints: 157 entries per chunk by the rule above; 208 makes it two rounds).
        const uint64_t round = b.curve == CURVE_G1 ? (uint64_t)256 * 4 * ACC_WAVES_G1 * 64 : (uint64_t)256 * 4 * ACC_WAVES_G2 * 64 / 2;
        const uint64_t k = (launch_entries + round * 208 - 1) / (round * 208);          // rounds, chunks of at most ~208 entries
        const uint64_t c2 = (launch_entries + k * round - 1) / (k * round);
        if (c2 > chunk_min) chunk = (uint32_t)c2;
        else chunk = chunk_min;
    }
    if (chunk < chunk_min) chunk = chunk_min;
    w.chunk = chunk;
    w.nthreads = (maxN + chunk - 1) / chunk;
    const size_t XB = b.curve == CURVE_G1 ? RawLayout<Fp>::XYZZ : RawLayout<Fp2>::XYZZ;      // intermediate points: raw layout
    ZKCHK(w.counts.alloc(4 * (size_t)(w.nbuckets + 1)));
    ZKCHK(w.offsets.alloc(4 * (size_t)(w.nbuckets + 1)));
    ZKCHK(w.cursor.alloc(4 * (size_t)(w.nbuckets + 1)));
    ZKCHK(w.sorted.alloc(4 * (size_t)maxN));
    w.sort_wgs = 0;
    w.sort_fine_bits = 0;
    static constexpr uint32_t COARSE_BINS = 512;
    // Two levels from 2^15 buckets and 2^20 (scalar, window) pairs up.  Above 2^15 buckets the histogram does not fit LDS; AT 2^15 it does (the
    // single-level sort below), but its 128 KiB per workgroup evict the accumulate workgroups of the other proofs in flight from every compute unit
    // the sort runs on: two levels (2 KiB + 20 KiB of LDS) measure +1.6 % proofs/s at 2^16, +1.9 % at 2^18 with the same single-proof latency.
    // ZK_SORT_TWO_LEVEL=0: never (A/B; above 2^15 buckets that is the global-atomic sort); ZK_SORT_TWO_LEVEL_MIN: log2 of the pair threshold.
    const int two_level = ::zk::opt("ZK_SORT_TWO_LEVEL") ? atoi(::zk::opt("ZK_SORT_TWO_LEVEL")) : 1;
    const int two_level_min = ::zk::opt("ZK_SORT_TWO_LEVEL_MIN") ? atoi(::zk::opt("ZK_SORT_TWO_LEVEL_MIN")) : (w.nbuckets > 2 * SORT_MAX_BUCKETS ? 22 : 20);          // 2^16 buckets (c = 17) sort in two levels from 2^20 pairs like 2^15
    if (b.precomp && w.nbuckets >= SORT_MAX_BUCKETS && w.nbuckets / COARSE_BINS <= SORT_MAX_FINE && maxN >= ((uint64_t)1 << two_level_min) && two_level != 0) {
        w.sort_fine_bits = ceil_log2(w.nbuckets / COARSE_BINS);
        uint64_t wgs = maxN / (4 * (uint64_t)COARSE_BINS);
        w.sort_wgs = (uint32_t)(wgs > 256 ? 256 : wgs < 1 ? 1 : wgs);          // < 1: only when ZK_SORT_TWO_LEVEL_MIN forces the two levels on a handful of pairs
        ZKCHK(w.wgcount.alloc(4 * (size_t)COARSE_BINS * w.sort_wgs));
        ZKCHK(w.sorted2.alloc(8 * (size_t)maxN));
        ZKCHK(w.coarse.alloc(4 * (size_t)3 * (COARSE_BINS + 1)));
    } else if (b.precomp && w.nbuckets <= SORT_MAX_BUCKETS) {
        // every workgroup zeroes and flushes nbuckets counters: give it >= 4 pairs per counter to amortise that
        uint64_t wgs = maxN / (4 * (uint64_t)w.nbuckets);
        if (wgs > 256) wgs = 256;
        // tuning knob.  With scalar-major passes 8 workgroups already beat the global atomics (2^16: 37.3 -> 38.7-39.4 M constraints/s, the
        // sort 0.61 -> 0.45 ms per proof; below 8 -- pools of 2^14 constraints -- the global path wins: 23.6 M against 22.3-22.6); it was 64
        // with window-major passes
        static const uint64_t min_wgs = ::zk::opt("ZK_SORT_MIN_WGS") ? (uint64_t)atoll(::zk::opt("ZK_SORT_MIN_WGS")) : 8;
        if (wgs >= min_wgs) {                               // below that too few workgroups: the global-atomic path is cheaper
            w.sort_wgs = (uint32_t)wgs;
            ZKCHK(w.wgcount.alloc(4 * (size_t)w.nbuckets * wgs));
        }
    }
    w.ba_rounds = ba_rounds_for(b, w.nbuckets);
    w.red_offsets = w.offsets.as<uint32_t>();
    w.red_chunk = w.chunk;
    uint64_t part_slots = w.nthreads;
    if (w.ba_rounds) {
        const uint64_t nz = max_nonzero && max_nonzero < b.n ? max_nonzero : b.n;
        w.ba_max_entries = nz * b.nw;
        const size_t PB = 2 * (b.curve == CURVE_G1 ? RawLayout<Fp>::ELEM : RawLayout<Fp2>::ELEM);
        w.ba_cap[0] = (w.ba_max_entries >> 1) + w.nbuckets;
        w.ba_cap[1] = (w.ba_max_entries >> 2) + w.nbuckets;
        ZKCHK(w.ba_offs.alloc(4 * (size_t)w.ba_rounds * (w.nbuckets + 1)));
        ZKCHK(w.ba_e[0].alloc(PB * w.ba_cap[0]));
        if (w.ba_rounds > 1) ZKCHK(w.ba_e[1].alloc(PB * w.ba_cap[1]));
        // the finisher works on what R rounds leave: short chunks (a few entries per bucket), one partial slot per chunk
        const uint64_t left = (w.ba_max_entries >> w.ba_rounds) + w.nbuckets;
        const uint64_t fin_threads = (left + BA_FINISH_CHUNK - 1) / BA_FINISH_CHUNK;
        if (fin_threads > part_slots) part_slots = fin_threads;
    }
    ZKCHK(w.buckets.alloc(XB * w.nbuckets));
    ZKCHK(w.head.alloc(XB * part_slots));
    ZKCHK(w.tail.alloc(XB * part_slots));
    ZKCHK(w.worklist.alloc(4 * (size_t)(w.nbuckets + 1)));
    const DigitPlan dp = digit_plan(b.c);
    ZKCHK(w.red.alloc(XB * (size_t)(dp.nd0 + dp.nd1) * (b.precomp ? 1 : b.nw)));
    ZKCHK(w.wsum.alloc(XB * tail_wsum_points(dp, b.precomp ? 1 : b.nw)));        // block sums of the weighting step (msm_tail.cuh)
    return ZK_OK;
}

// Steps 1-4 for `count` MSMs over the same bases: digits, counting sort, bucket accumulation (leaves raw bucket sums and
// chunk partials in the workspaces).  One chain of launches for all of them.
int msm_sort_accumulate_many(const MsmBases& b, MsmWorkspace* const* ws, const void* const* d_scalars, uint32_t count, hipStream_t s) {
    if (count == 0 || count > MAX_SORT_JOBS) ZK_FAIL(ZK_ERR_ARG, "msm_sort_accumulate_many: 1..4 MSMs per batch");
    SortJobs sj{};
    AccJobs aj{};
    for (uint32_t i = 0; i < count; i++) {
        MsmWorkspace& w = *ws[i];
        if (w.c != b.c || w.precomp != b.precomp || w.curve != b.curve || w.cap_points < b.n || w.nbuckets != ws[0]->nbuckets || w.chunk != ws[0]->chunk ||
            w.sort_wgs != ws[0]->sort_wgs || w.sort_fine_bits != ws[0]->sort_fine_bits)
            ZK_FAIL(ZK_ERR_ARG, "msm: workspace does not match bases");
        sj.scalars[i] = (const uint32_t*)d_scalars[i];
        sj.counts[i] = w.counts.as<uint32_t>(); sj.offsets[i] = w.offsets.as<uint32_t>(); sj.cursor[i] = w.cursor.as<uint32_t>();
        sj.sorted[i] = w.sorted.as<uint32_t>(); sj.wgcount[i] = w.wgcount.as<uint32_t>(); sj.sorted2[i] = w.sorted2.as<uint2>();
        aj.offsets[i] = w.offsets.as<uint32_t>(); aj.sorted[i] = w.sorted.as<uint32_t>();
        aj.buckets[i] = w.buckets.as<uint8_t>(); aj.head[i] = w.head.as<uint8_t>(); aj.tail[i] = w.tail.as<uint8_t>();
    }
    MsmWorkspace& w = *ws[0];
    const uint32_t nbw = 1u << (b.c - 1);
    // scalar-major LDS sorts need ONE bucket set (resident keys: every window files into the same 2^(c-1) buckets); ZK_SORT_SCALAR_MAJOR=0 restores
    // the window-major ranges
    static const bool want_sm = !(::zk::opt("ZK_SORT_SCALAR_MAJOR") && atoi(::zk::opt("ZK_SORT_SCALAR_MAJOR")) == 0);
    const bool sm = want_sm && b.precomp && w.sort_wgs != 0;
#ifdef ZK_EXPERIMENTS
    static const uint32_t alias = (::zk::opt("ZK_EXPERIMENT_TABLE_ALIAS") && atoi(::zk::opt("ZK_EXPERIMENT_TABLE_ALIAS"))) ? 1u : 0u;
#else
    const uint32_t alias = 0u;
#endif
    DigitArgs da{b.n, b.c, b.nw, b.precomp ? 1u : 0u, nbw, {0, 0, 0, 0, 0, 0, 0, 0, 0}, b.ident.as<uint8_t>(),
                 0u, sm ? 1u : 0u, alias, b.fold ? 1u : 0u};
    for (uint32_t j = 0; j < b.nw; j++) {               // K += (2^(c-1) - 1) << (c*j)
        uint64_t v = ((uint64_t)1 << (b.c - 1)) - 1;
        uint32_t off = j * b.c, wd = off >> 5, sh = off & 31;
        unsigned __int128 add = (unsigned __int128)v << sh;
        uint64_t cy = 0;
        for (uint32_t k = wd; k < 9; k++) {
            cy += (uint64_t)da.K[k] + (uint32_t)(add & 0xffffffffu);
            da.K[k] = (uint32_t)cy;
            cy >>= 32;
            add >>= 32;
            if (!add && !cy) break;
        }
    }
    {
        ScopedTimer t("msm_sort", s);
        if (w.sort_fine_bits) {
            // level 1 over the coarse bins (its counts / offsets / cursor live in `coarse`), level 2 writes the real offsets and references
            const uint32_t bins = w.nbuckets >> w.sort_fine_bits;
            if (bins > SORT_FEW_BINS) ZK_FAIL(ZK_ERR_ARG, "two-level sort: more coarse bins than its first level reserves counters for");
            SortJobs l1 = sj;
            for (uint32_t i = 0; i < count; i++) {
                uint32_t* c3 = ws[i]->coarse.as<uint32_t>();
                l1.counts[i] = c3; l1.offsets[i] = c3 + (bins + 1); l1.cursor[i] = c3 + 2 * (bins + 1);
            }
            DigitArgs d1 = da;
            d1.coarse_shift = w.sort_fine_bits;
            const uint64_t total = sm ? b.n : b.n * b.nw, per_wg = (total + w.sort_wgs - 1) / w.sort_wgs;
            hipLaunchKernelGGL(k_sort_count_lds<SORT_FEW_BINS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            dim3 gc = grid_for(bins, 256);
            gc.y = count;
            hipLaunchKernelGGL(k_sort_colscan, gc, dim3(256), 0, s, l1, bins, w.sort_wgs);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, l1, bins);
            const char* e_cs = ZK_FORM_ENV("ZK_SORT_COARSE_STAGED");          // a kernel-form switch (zk_common.h)
            const bool coarse_staged = sm && (e_cs ? atoi(e_cs) != 0 : SORT_COARSE_STAGED_DEFAULT);
            if (coarse_staged) hipLaunchKernelGGL(k_sort_scatter_staged, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            else hipLaunchKernelGGL(k_sort_scatter_lds<SORT_FEW_BINS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, l1, d1, per_wg, bins);
            SortJobs l2 = sj;
            for (uint32_t i = 0; i < count; i++) l2.cursor[i] = l1.offsets[i];          // the coarse offsets (k_scan wrote offsets = cursor; the scatter advanced neither: it ranks in LDS)
            const char* e_st = ZK_FORM_ENV("ZK_SORT_FINE_STAGED");          // a kernel-form switch (zk_common.h)
            bool staged = e_st ? atoi(e_st) != 0 : b.n * b.nw / bins >= SORT_FINE_STAGED_MIN;
            // its static LDS (part + the two staging tiles) plus 8 bytes per fine bucket must fit the device's per-workgroup limit (84 KB at c = 22; the
            // MI355X allows 160 KB): asked once per device, the plain form serves wherever it does not fit
            if (staged) {
                static int lds_limit[64] = {0};
                const int dev = ctx().device >= 0 && ctx().device < 64 ? ctx().device : 0;
                if (!lds_limit[dev]) {
                    int v = 0;
                    lds_limit[dev] = hipDeviceGetAttribute(&v, hipDeviceAttributeMaxSharedMemoryPerBlock, ctx().device) == hipSuccess && v > 0 ? v : 65536;
                }
                const size_t need = sort_fine_staged_lds(w.sort_fine_bits) + 4 * (size_t)SORT_THREADS + 8 * (size_t)SORT_TILE;
                if (need > (size_t)lds_limit[dev]) staged = false;
            }
            if (staged) hipLaunchKernelGGL(k_sort_fine_staged, dim3(bins, count), dim3(SORT_THREADS), sort_fine_staged_lds(w.sort_fine_bits), s, l2, w.sort_fine_bits, bins, w.nbuckets);
            else hipLaunchKernelGGL(k_sort_fine, dim3(bins, count), dim3(SORT_THREADS), 0, s, l2, w.sort_fine_bits, bins, w.nbuckets);
        } else if (w.sort_wgs) {
            const uint64_t total = sm ? b.n : b.n * b.nw, per_wg = (total + w.sort_wgs - 1) / w.sort_wgs;
            hipLaunchKernelGGL(k_sort_count_lds<SORT_MAX_BUCKETS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, sj, da, per_wg, w.nbuckets);
            dim3 gc = grid_for(w.nbuckets, 256);
            gc.y = count;
            hipLaunchKernelGGL(k_sort_colscan, gc, dim3(256), 0, s, sj, w.nbuckets, w.sort_wgs);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, sj, w.nbuckets);
            hipLaunchKernelGGL(k_sort_scatter_lds<SORT_MAX_BUCKETS>, dim3(w.sort_wgs, count), dim3(SORT_THREADS), 0, s, sj, da, per_wg, w.nbuckets);
        } else {
            for (uint32_t i = 0; i < count; i++) HIPCHK(hipMemsetAsync(ws[i]->counts.p, 0, 4 * (size_t)(w.nbuckets + 1), s));
            dim3 g = grid_for(b.n * b.nw, 256);
            g.y = count;
            hipLaunchKernelGGL(k_msm_count, g, dim3(256), 0, s, sj, da);
            hipLaunchKernelGGL(k_scan, dim3(count), dim3(1024), 0, s, sj, w.nbuckets);
            hipLaunchKernelGGL(k_msm_scatter, g, dim3(256), 0, s, sj, da);
        }
    }
    uint32_t rounds = w.ba_rounds;
    for (uint32_t i = 0; i < count; i++)
        if (ws[i]->ba_rounds < rounds) rounds = ws[i]->ba_rounds;
    {
        // the whole bucket-accumulation stage of the batch (batch-affine rounds + the XYZZ accumulate) is ONE timed family span
        ScopedTimer t(b.curve == CURVE_G1 ? "msm_accumulate_g1" : "msm_accumulate_g2", s, 1);
        if (rounds) {
            ZKCHK(msm_batch_affine_rounds(b, ws, count, rounds, s));
            uint64_t max_entries = 0;
            for (uint32_t i = 0; i < count; i++) {
                MsmWorkspace& wi = *ws[i];
                wi.red_offsets = wi.ba_offs.as<uint32_t>() + (uint64_t)(rounds - 1) * (w.nbuckets + 1);
                wi.red_chunk = BA_FINISH_CHUNK;
                aj.offsets[i] = wi.red_offsets;
                aj.pts[i] = wi.ba_e[(rounds - 1) & 1].as<uint8_t>();
                if (wi.ba_max_entries > max_entries) max_entries = wi.ba_max_entries;
            }
            const uint64_t left = (max_entries >> rounds) + w.nbuckets;
            ZKCHK(msm_accumulate_launch(b.curve, (left + BA_FINISH_CHUNK - 1) / BA_FINISH_CHUNK, b.table.p, aj, count, w.nbuckets, BA_FINISH_CHUNK, s, true));
        } else {
            for (uint32_t i = 0; i < count; i++) { ws[i]->red_offsets = ws[i]->offsets.as<uint32_t>(); ws[i]->red_chunk = ws[i]->chunk; }
            ZKCHK(msm_accumulate_launch(b.curve, w.nthreads, b.table.p, aj, count, w.nbuckets, w.chunk, s));
        }
    }
    if (ctx().profiling >= 2 && !rounds) {
        // what the launch really did (un-overlapped measurement pass only: this synchronises): chunks = lanes that ran, copies = first entries of a
        // chunk or of a run (no product), second steps = the 6-product addition of two affine points, everything else a full mixed addition
        DevBuf st;
        ZKCHK(st.alloc(32));
        unsigned long long h[4], tot[4] = {0, 0, 0, 0}, chunks = 0;
        for (uint32_t i = 0; i < count; i++) {
            HIPCHK(hipMemsetAsync(st.p, 0, 32, s));
            hipLaunchKernelGGL(k_acc_stats, grid_for(w.nbuckets, 256), dim3(256), 0, s, (const uint32_t*)ws[i]->offsets.as<uint32_t>(), w.nbuckets, w.chunk, st.as<unsigned long long>());
            HIPCHK(hipMemcpyAsync(h, st.p, 32, hipMemcpyDeviceToHost, s));
            HIPCHK(hipStreamSynchronize(s));
            for (int q = 0; q < 4; q++) tot[q] += h[q];
            chunks += (h[0] + w.chunk - 1) / w.chunk;
        }
        const char* f = b.curve == CURVE_G1 ? "msm_accumulate_g1" : "msm_accumulate_g2";
        const unsigned long long copies = chunks + tot[1] - tot[2], second = chunks > tot[3] ? chunks - tot[3] : 0;
        profile_count((std::string(f) + ":entries").c_str(), tot[0]);
        profile_count((std::string(f) + ":copies").c_str(), copies);
        profile_count((std::string(f) + ":second_steps").c_str(), second);
        profile_count((std::string(f) + ":full_additions").c_str(), tot[0] > copies + second ? tot[0] - copies - second : 0);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int msm_sort_accumulate(const MsmBases& b, MsmWorkspace& w, const void* d_scalars, hipStream_t s) {
    MsmWorkspace* ws[1] = {&w};
    const void* sc[1] = {d_scalars};
    return msm_sort_accumulate_many(b, ws, sc, 1, s);
}
// Steps 5-7 for a batch of MSMs in one chain of launches: n1 products in G1 (bases b1) and n2 in G2 (bases b2), all with
// the same window plan.  outs[i]: one dense XYZZ point each.
int msm_reduce_mixed(const MsmBases* b1, MsmWorkspace* const* ws1, void* const* outs1, uint32_t n1,
                     const MsmBases* b2, MsmWorkspace* const* ws2, void* const* outs2, uint32_t n2, hipStream_t s) {
    if (n1 + n2 == 0 || n1 + n2 > MAX_TAIL_JOBS) ZK_FAIL(ZK_ERR_ARG, "msm_reduce: 1..8 MSMs per batch");
    const MsmBases& b = n1 ? *b1 : *b2;
    if (n1 && n2 && (b1->c != b2->c || b1->nw != b2->nw || b1->precomp != b2->precomp)) ZK_FAIL(ZK_ERR_ARG, "msm_reduce: one window plan per batch");
    const uint32_t nwin = b.precomp ? 1 : b.nw;
    TailJobs jobs{};
    jobs.n1 = n1;
    uint64_t max_lanes = 0, max_chunks = 0;
    uint32_t max_nb = 0;
    for (uint32_t i = 0; i < n1 + n2; i++) {
        MsmWorkspace& w = i < n1 ? *ws1[i] : *ws2[i - n1];
        const MsmBases& bi = i < n1 ? *b1 : *b2;
        if (w.c != bi.c || w.precomp != bi.precomp || w.curve != bi.curve || bi.curve != (i < n1 ? CURVE_G1 : CURVE_G2))
            ZK_FAIL(ZK_ERR_ARG, "msm_reduce: workspace does not match its bases");
        jobs.j[i] = TailJob{w.red_offsets, w.buckets.as<uint8_t>(), w.head.as<uint8_t>(), w.tail.as<uint8_t>(), w.worklist.as<uint32_t>(),
                            w.red.as<uint8_t>(), w.wsum.as<uint8_t>(), (uint8_t*)(i < n1 ? outs1[i] : outs2[i - n1]), w.nbuckets, w.red_chunk};
        HIPCHK(hipMemsetAsync(w.worklist.p, 0, 4, s));
        const uint64_t lanes = (uint64_t)w.nbuckets * (i < n1 ? 1 : 2);
        if (lanes > max_lanes) max_lanes = lanes;
        if (w.nbuckets > max_nb) max_nb = w.nbuckets;
        const uint64_t chunks = w.red_chunk == w.chunk ? w.nthreads : ~(uint64_t)0;      // after batch-affine rounds the chunk count is not the workspace's: per bucket
        if (chunks > max_chunks) max_chunks = chunks;
    }
    const uint32_t count = n1 + n2;
    const char* fam = n2 == 0 ? "msm_reduce_g1" : (n1 == 0 ? "msm_reduce_g2" : "msm_reduce");
    ScopedTimer t(fam, s);
    const DigitPlan dp = digit_plan(b.c);
    const bool wide = dp.nd0 > DW_POINTS;
    // bucket sums -> digit sums: on slots (msm_tail.hip) where the chain is latency-bound, one lane per point where the launch is throughput-bound
    // (msm_tail.cuh).  ZK_TAIL_SLOTS = 0 / 1 forces one form of the digit sums, ZK_TAIL_FIXUP_SLOTS = 1 puts the fix-up on slots (A/B runs, latency-first
    // deployments).  Digit sums -> product: always msm_tail.hip.
    // (cached; read per call only under ZK_TEST_FORMS=1, where the GPU suite flips them inside one process to hold every form to the oracle)
    const char* e_sums = ZK_FORM_ENV("ZK_TAIL_SLOTS");
    const char* e_fix = ZK_FORM_ENV("ZK_TAIL_FIXUP_SLOTS");
    const char* e_chunk = ZK_FORM_ENV("ZK_FIXUP_BY_CHUNK");
    const int force = e_sums ? atoi(e_sums) : -1, force_fixup = e_fix ? atoi(e_fix) : -1;
    const bool sums_on_slots = force < 0 ? !wide : force != 0;
    if (force_fixup > 0) {                                 // default: lanes at every width (fewer instructions: +1.5 % proofs/s at 2^16, +0.8 % at 2^18; slots: -0.3 ms of a lone 2^18 proof)
        ZKCHK(msm_tail_fixup_slots(jobs, count, n2, max_nb, s));
    } else {
        ScopedTimer t1("msm_reduce:fixup", s);
        // one worker per bucket, or per chunk border where those are fewer (msm_red.hip: k_msm_fixup)
        const bool by_chunk_ok = !(e_chunk && atoi(e_chunk) == 0);      // A/B switch
        ZKCHK(msm_red_fixup_launch(jobs, count, n2, by_chunk_ok && max_chunks < max_nb, by_chunk_ok && max_chunks < max_nb ? max_chunks : max_nb, max_nb, s));
    }
    if (sums_on_slots) {
        ZKCHK(msm_tail_digit_sums_slots(jobs, count, n2, nwin, b.c, s));
    } else {
        const char* e_wg = ZK_FORM_ENV("ZK_DS_WIDE_GROUP");          // a kernel-form switch like the ones above
        const uint32_t wg_env = e_wg ? (uint32_t)atoi(e_wg) : 0;
        ScopedTimer t2("msm_reduce:digit_sums", s);
        ZKCHK(msm_red_digit_sums_launch(jobs, count, n2, nwin, dp, wide, wg_env, s));
    }
    return msm_tail_weight_slots(jobs, count, n2, nwin, b.c, s);
}
int msm_reduce(const MsmBases& b, MsmWorkspace* const* ws, void* const* outs, uint32_t count, hipStream_t s) {
    return b.curve == CURVE_G1 ? msm_reduce_mixed(&b, ws, outs, count, nullptr, nullptr, nullptr, 0, s)
                               : msm_reduce_mixed(nullptr, nullptr, nullptr, 0, &b, ws, outs, count, s);
}
int msm_run(const MsmBases& b, MsmWorkspace& w, const void* d_scalars, void* d_out, hipStream_t s) {
    ZKCHK(msm_sort_accumulate(b, w, d_scalars, s));
    MsmWorkspace* ws[1] = {&w};
    void* outs[1] = {d_out};
    return msm_reduce(b, ws, outs, 1, s);
}
int xyzz_sum_columns(Curve curve, void* d_out, const void* d_parts, uint32_t count, uint32_t npoints, hipStream_t s) {
    if (curve == CURVE_G1) hipLaunchKernelGGL(k_xyzz_sum_columns<Fp>, grid_for(npoints, 64), dim3(64), 0, s, (uint8_t*)d_out, (const uint8_t*)d_parts, count, npoints);
    else hipLaunchKernelGGL(k_xyzz_sum_columns<Fp2H>, grid_for(2 * (uint64_t)npoints, 64), dim3(64), 0, s, (uint8_t*)d_out, (const uint8_t*)d_parts, count, npoints);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

// the generator tables 2^k * G live in the context of the device they were built on (zk_common.h: CtxBufs) and die with it

int fixed_base_mul(Curve curve, void* d_out, const void* d_scalars, uint64_t n, hipStream_t s) {
    if (!ctx().bufs) ZK_FAIL(ZK_ERR_HIP, "fixed_base_mul: no device context (zk_init)");
    DevBuf& tab = ctx().bufs->pow2[curve];
    if (!tab.p) {
        ZKCHK(tab.alloc(aff_bytes(curve) * 256));
        if (curve == CURVE_G1) hipLaunchKernelGGL(k_gen_pow2_table<Fp>, dim3(4), dim3(64), 0, s, tab.as<uint8_t>());
        else hipLaunchKernelGGL(k_gen_pow2_table<Fp2>, dim3(4), dim3(64), 0, s, tab.as<uint8_t>());
        HIPCHK(hipGetLastError());
    }
    if (!n) return ZK_OK;
    ScopedTimer t("fixed_base_mul", s);
    if (curve == CURVE_G1) hipLaunchKernelGGL(k_fixed_base_mul<Fp>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_out, (const uint32_t*)d_scalars, n, tab.as<uint8_t>());
    else hipLaunchKernelGGL(k_fixed_base_mul<Fp2>, grid_for(n, 128), dim3(128), 0, s, (uint8_t*)d_out, (const uint32_t*)d_scalars, n, tab.as<uint8_t>());
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

// Fr canonical check on device scalars
__global__ void k_check_scalars(const uint32_t* s, uint64_t n, int* flag) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr a = fe_load<FrParams>(s + 8 * i);
    if (!fe_is_canonical(a)) *flag = 1;
}
// powers: out[i] = s^i (canonical), sequential products split over lanes by fast exponentiation
__global__ void k_fr_powers(uint32_t* out, const uint32_t* s_canon, uint64_t n) {
    uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    Fr base = fe_to_mont(fe_load<FrParams>(s_canon)), acc = fe_one<FrParams>();
    for (uint64_t e = i; e; e >>= 1) {
        if (e & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    fe_store<FrParams>(out + 8 * i, fe_from_mont(acc));
}

static int msm_api(Curve curve, const uint8_t* bases, size_t nbases, const uint8_t* scalars, size_t nscalars, uint32_t window_bits, uint8_t* out) {
    if (!out) ZK_FAIL(ZK_ERR_ARG, "msm: null output");
    if (nscalars > nbases) ZK_FAIL(ZK_ERR_APPLY_POWERS, "apply_powers");        // curve.ml:116
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    if (nscalars == 0) {                                                        // curve.ml:115: zero
        memset(out, 0, aff_bytes(curve));
        out[0] = 0x40;
        return ZK_OK;
    }
    if (!bases || !scalars) ZK_FAIL(ZK_ERR_ARG, "msm: null input");
    MsmBases b;
    MsmWorkspace w;
    // ZK_MSM_API_PRECOMP=1 runs this entry point through the resident-key machinery (window tables, one bucket set, batch-affine
    // rounds): how the tests reach those kernels with adversarial base sets (duplicates, negations, the identity)
    const bool precomp = ::zk::opt("ZK_MSM_API_PRECOMP") && atoi(::zk::opt("ZK_MSM_API_PRECOMP")) != 0;
    ZKCHK(msm_bases_from_bytes(b, curve, bases, nscalars, window_bits, precomp, c.stream));
    ZKCHK(msm_workspace_alloc(w, b));
    DevBuf sc, res, flag;
    ZKCHK(sc.alloc(32 * nscalars));
    ZKCHK(res.alloc(xyzz_bytes(curve)));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(sc.p, scalars, 32 * nscalars, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_check_scalars, grid_for(nscalars, 256), dim3(256), 0, c.stream, sc.as<uint32_t>(), (uint64_t)nscalars, flag.as<int>());
    ZKCHK(msm_run(b, w, sc.p, res.p, c.stream));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (h) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "msm: scalar >= r");
    return points_xyzz_to_bytes(curve, res.p, 1, out, c.stream);
}
static int of_fr_api(Curve curve, const uint8_t* scalars, size_t n, uint8_t* out) {
    if (n && (!scalars || !out)) ZK_FAIL(ZK_ERR_ARG, "of_Fr: null");
    ZKCHK(ensure_init());
    if (!n) return ZK_OK;
    Ctx& c = ctx();
    DevBuf sc, aff, bytes, flag;
    ZKCHK(sc.alloc(32 * n));
    ZKCHK(aff.alloc(aff_bytes(curve) * n));
    ZKCHK(bytes.alloc(aff_bytes(curve) * n));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(sc.p, scalars, 32 * n, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_check_scalars, grid_for(n, 256), dim3(256), 0, c.stream, sc.as<uint32_t>(), (uint64_t)n, flag.as<int>());
    ZKCHK(fixed_base_mul(curve, aff.p, sc.p, n, c.stream));
    ZKCHK(points_affine_to_bytes(curve, bytes.p, aff.p, n, c.stream));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(out, bytes.p, aff_bytes(curve) * n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (h) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "of_Fr: scalar >= r");
    return ZK_OK;
}
static int powers_api(Curve curve, uint32_t d, const uint8_t* s32, uint8_t* out) {
    if (!s32 || !out) ZK_FAIL(ZK_ERR_ARG, "powers: null");
    ZKCHK(ensure_init());
    Ctx& c = ctx();
    uint64_t n = (uint64_t)d + 1;
    DevBuf s, sc, aff, bytes, flag;
    ZKCHK(s.alloc(32));
    ZKCHK(sc.alloc(32 * n));
    ZKCHK(aff.alloc(aff_bytes(curve) * n));
    ZKCHK(bytes.alloc(aff_bytes(curve) * n));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, c.stream));
    HIPCHK(hipMemcpyAsync(s.p, s32, 32, hipMemcpyHostToDevice, c.stream));
    hipLaunchKernelGGL(k_check_scalars, dim3(1), dim3(64), 0, c.stream, s.as<uint32_t>(), (uint64_t)1, flag.as<int>());
    hipLaunchKernelGGL(k_fr_powers, grid_for(n, 256), dim3(256), 0, c.stream, sc.as<uint32_t>(), s.as<uint32_t>(), n);
    ZKCHK(fixed_base_mul(curve, aff.p, sc.p, n, c.stream));
    ZKCHK(points_affine_to_bytes(curve, bytes.p, aff.p, n, c.stream));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipMemcpyAsync(out, bytes.p, aff_bytes(curve) * n, hipMemcpyDeviceToHost, c.stream));
    HIPCHK(hipStreamSynchronize(c.stream));
    if (h) ZK_FAIL(ZK_ERR_SCALAR_RANGE, "powers: scalar >= r");
    return ZK_OK;
}

}  // namespace zk

using namespace zk;
extern "C" {
int zk_msm_g1(const uint8_t* bases, size_t nbases, const uint8_t* scalars, size_t nscalars, uint32_t window_bits, uint8_t out[96]) {
    return msm_api(CURVE_G1, bases, nbases, scalars, nscalars, window_bits, out);
}
int zk_msm_g2(const uint8_t* bases, size_t nbases, const uint8_t* scalars, size_t nscalars, uint32_t window_bits, uint8_t out[192]) {
    return msm_api(CURVE_G2, bases, nbases, scalars, nscalars, window_bits, out);
}
int zk_g1_of_fr(const uint8_t* scalars, size_t n, uint8_t* out) { return of_fr_api(CURVE_G1, scalars, n, out); }
int zk_g2_of_fr(const uint8_t* scalars, size_t n, uint8_t* out) { return of_fr_api(CURVE_G2, scalars, n, out); }
int zk_g1_powers(uint32_t d, const uint8_t s[32], uint8_t* out) { return powers_api(CURVE_G1, d, s, out); }
int zk_g2_powers(uint32_t d, const uint8_t s[32], uint8_t* out) { return powers_api(CURVE_G2, d, s, out); }
}

