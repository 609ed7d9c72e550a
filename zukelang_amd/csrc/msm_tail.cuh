// Bucket reduction of the Pippenger products (steps 5-7 of msm_run): what the launch chain of msm.hip and the kernels of msm_tail.hip share.
#pragma once
#include "msm.cuh"

namespace zk {

// The reduction kernels serve several MSMs of one proof in ONE launch (blockIdx.z = job): the MSMs share the base
// set (hence bucket count, chunk length, window plan) and differ in the scalars, so their reductions are the same
// launch geometry on different buffers -- one latency-bound chain per proof and curve instead of one per MSM.
static constexpr uint32_t MAX_TAIL_JOBS = 8;
struct TailJob {
    const uint32_t* offsets;
    uint8_t* buckets;
    const uint8_t* head;
    const uint8_t* tail;
    uint32_t* worklist;
    uint8_t* red;
    uint8_t* wsum;
    uint8_t* out;
    uint32_t nb, chunk;          // buckets of this MSM, sorted entries per accumulate chunk
};
// Jobs [0, n1) are G1 products (T = Fp), jobs [n1, n1 + n2) G2 products (T = Fp2H): ONE launch per step serves both
// curves, with the launch geometry of the larger one (blocks and lanes a job has no use for leave at once).
struct TailJobs {
    TailJob j[MAX_TAIL_JOBS];
    uint32_t n1;
};
static constexpr uint32_t FIXUP_SERIAL_MAX = 16;
// R = sum_w w * B_w with w = hi * 2^lb + lo: nd0 low-digit values, nd1 high-digit values, nbw buckets per window
struct DigitPlan {
    uint32_t nbw, lb, nd0, nd1;
};
static inline DigitPlan digit_plan(uint32_t c) {
    DigitPlan p;
    p.nbw = 1u << (c - 1);
    p.lb = (c + 1) / 2;
    p.nd0 = 1u << p.lb;
    p.nd1 = (p.nbw >> p.lb) + 1;
    return p;
}
static constexpr uint32_t DW_POINTS = 256;       // more digit values per half than this (windows above 16 bits): the WIDE forms of the sums kernels
// msm_tail.hip weights the digit sums in blocks of 2^bw values, at most 32 blocks per half; its buffer: T_b | L_b per block and window, then one W per window
static inline uint32_t tail_bw_log(const DigitPlan& p) { return p.lb > 5 ? p.lb - 5 : 0; }
static inline size_t tail_wsum_points(const DigitPlan& p, uint32_t nwin) {
    const uint32_t bw = tail_bw_log(p);
    return (size_t)nwin * (2 * ((p.nd0 >> bw) + ((p.nd1 + (1u << bw) - 1) >> bw)) + 1);
}

// msm_tail.hip: the reduction with every point spread over four slots of a wave (ec_slots.cuh); nwin = bucket sets (1 for resident tables).
//   fixup:      chunk partial sums -> bucket sums.  OPTIONAL (ZK_TAIL_FIXUP_SLOTS=1): this step is bound by the chip's throughput at every size (one
//               addition per chunk border), where a slot addition costs 16 slot-products and four copies of the non-multiply work for 14 useful
//               products -- the one-lane kernels of msm.hip are the default (+1.5 % proofs/s at 2^16); on slots a lone 2^18 proof is 0.3 ms shorter.
//   digit sums: bucket sums -> S in `red`.  Latency-bound up to 2^15 buckets, where the four slots pay (same instruction count, 0.42 -> 0.24 ms of a
//               lone 2^16 proof); above (windows of more than 16 bits) throughput-bound: the one-lane-per-point kernel of msm.hip
//               (2^20 constraints, c = 20, a dozen proofs in flight: 54.5 against 52.5 M constraints/s with both sums steps on slots).
//   weight:     S -> the product (block weights, combine, Horner over the windows): a few hundred points, always on slots.
// msm_red.hip: the same two steps with one lane (or lane pair) per point.  workers = buckets, or chunk borders with by_chunk; wide_group = points per digit
// value in the wide form (16 / 32 / 64; 0 = the default)
int msm_red_fixup_launch(const TailJobs& jobs, uint32_t count, uint32_t n2, bool by_chunk, uint64_t workers, uint32_t max_nb, hipStream_t s);
int msm_red_digit_sums_launch(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, const DigitPlan& dp, bool wide, uint32_t wide_group, hipStream_t s);
int msm_tail_fixup_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t max_nb, hipStream_t s);
int msm_tail_digit_sums_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, uint32_t c, hipStream_t s);
int msm_tail_weight_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, uint32_t c, hipStream_t s);

}  // namespace zk
