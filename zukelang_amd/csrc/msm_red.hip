// Fix-up of the chunk partial sums and digit sums of the bucket reduction with ONE lane (G1) or lane pair (G2) per point: the forms for launches that are
// bound by the chip's throughput (windows above 16 bits: 2^17+ buckets, two additions each).  A translation unit of its own since round 4 so that its
// variants can be built and measured apart from msm.hip (profiles/r04_reduction_ab.txt, one box):
//   * the second operand of every addition is read from memory coordinate by coordinate (xyzz_add_raw_mem) and the equal-x case is out of line;
//   * the field products stay OUT OF LINE: expanded in place (-DZK_RED_INLINE_MUL) the lane-pair branch of a general addition wants ~290 registers --
//     69 spilled at two waves per SIMD, none at one wave but half the latency hiding -- and neither form is faster than the calls;
//   * what pays is the SHAPE of a digit value's sum: 32 instead of 64 points per value (ZK_DS_WIDE_GROUP) keeps more lanes adding for longer:
//     +1.5 % proofs/s at 2^20 with the single-proof latency unchanged (16 points: +1.8 % for +1 ms of latency).
#ifdef ZK_RED_INLINE_MUL          // A/B build (make EXTRA=-DZK_RED_INLINE_MUL msm_red.o): the field products expanded in place, see the header comment
#define ZK_FP_INLINE_MUL 1
#endif
#include "ec.cuh"
#include "msm_tail.cuh"

#include <stdlib.h>

namespace zk {


// A run cut by chunk borders left one partial sum per chunk.  Usually that is a handful per bucket
// (one lane adds them); a bucket that swallowed a large share of the digits (boolean-heavy witnesses,
// or a top window whose digit is only 0 / 1) leaves thousands: those go to a worklist and get a whole
// workgroup each (strided lane sums + LDS tree), so the longest chain is count/256 + 8 instead of count.
//
// Everything from here to k_msm_final works on the RAW point layout (ec.cuh) with the group operations
// expanded in place: a point travels registers -> memory -> registers without reduction or repacking and
// the kernels use no scratch memory.  T is Fp (G1, one lane per point) or Fp2H (G2, a lane PAIR per
// point: half the registers and two instead of three dependent base-field products per Fp2 product --
// these kernels are chains of dependent additions run by a few waves, so latency is what they cost).
template <class T> struct Lanes { static constexpr uint32_t N = RawLayout<T>::LANES; };
// BY_CHUNK = false: one worker (lane, or lane pair) per BUCKET.  BY_CHUNK = true: one worker per CHUNK BORDER -- worker t looks up the bucket of the last
// entry of chunk t (binary search in the offsets) and owns the bucket's fix-up if the run starts in chunk t and goes on beyond it.  With more buckets
// than chunks (windows above 16 bits: 2^19 buckets, runs of ~26 entries inside chunks of ~200) one bucket in eight crosses a border: per bucket, a
// wave ran the additions with a few of its lanes (PMC at 2^20: 317 M wave instructions per proof; 153 M per border).
template <class T, bool BY_CHUNK> FF_INLINE void fixup_body(const TailJob& job) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint32_t* __restrict__ offsets = job.offsets;
    uint8_t* __restrict__ buckets = job.buckets;
    const uint8_t* __restrict__ head = job.head;
    const uint8_t* __restrict__ tail = job.tail;
    uint32_t* __restrict__ worklist = job.worklist;
    const uint32_t chunk = job.chunk;
    const uint32_t worker = (blockIdx.x * blockDim.x + threadIdx.x) / Lanes<T>::N;
    uint32_t kb = worker;
    if constexpr (BY_CHUNK) {
        const uint64_t last = ((uint64_t)worker + 1) * chunk - 1;      // last entry of chunk `worker`
        if (last + 1 >= offsets[job.nb]) return;                        // nothing sorted beyond it: no border
        uint32_t lo = 0, hi = job.nb;                                   // first index whose offset exceeds `last`, minus one: the (non-empty) bucket of that entry
        while (lo < hi) {
            const uint32_t mid = (lo + hi) >> 1;
            if (offsets[mid] > (uint32_t)last) hi = mid;
            else lo = mid + 1;
        }
        kb = lo - 1;
    } else if (kb >= job.nb) return;
    const uint32_t s = offsets[kb], e = offsets[kb + 1];
    if (e == s) return;                                 // empty bucket: nobody reads its slot
    const uint32_t t0 = s / chunk, t1 = (e - 1) / chunk;
    if (t0 == t1) return;                               // whole run inside one chunk: written directly
    if (BY_CHUNK && t0 != worker) return;               // the run began in an earlier chunk: that chunk's worker has it
    if (t1 - t0 > FIXUP_SERIAL_MAX) {
        if ((threadIdx.x & (Lanes<T>::N - 1)) == 0) worklist[1 + atomicAdd(&worklist[0], 1u)] = kb;
        return;
    }
    Xyzz<T> acc = xyzz_load_raw<T>((s == t0 * chunk ? head : tail) + (uint64_t)XB * t0);
    for (uint32_t t = t0 + 1; t <= t1; t++) xyzz_add_raw_mem<T>(acc, head + (uint64_t)XB * t);          // the operand comes from memory coordinate by coordinate: no spill
    xyzz_store_raw<T>(buckets + (uint64_t)XB * kb, acc);
}
// WAVES = waves per SIMD the kernel is compiled for: with the products expanded in place the lane-pair (G2) branch of a general addition wants ~290
// registers -- two waves (256) spill 37-70 of them, one wave (512) none.  ZK_RED_WAVES selects (A/B; the default is the measured best).
template <bool BY_CHUNK, int WAVES> __global__ __launch_bounds__(128, WAVES) void k_msm_fixup(TailJobs jobs) {
    if (blockIdx.z < jobs.n1) fixup_body<Fp, BY_CHUNK>(jobs.j[blockIdx.z]);
    else fixup_body<Fp2H, BY_CHUNK>(jobs.j[blockIdx.z]);
}
// sum of the accumulators of the NT / lanes points of a workgroup, result in point 0
// (GROUP = points per independent sum, a power of two; 0 = the whole workgroup: result in point 0 of every group)
template <class T, int NT, int GROUP = 0> FF_INLINE void block_tree_sum(Xyzz<T>& acc, uint32_t (*lds)[NT]) {
    constexpr uint32_t LP = Lanes<T>::N;
    constexpr uint32_t GP = GROUP ? GROUP : NT / LP;
    const uint32_t t = threadIdx.x, pi = (t / LP) & (GP - 1);
    uint32_t tmp[LANE_POINT_WORDS];
    for (uint32_t d = GP / 2; d >= 1; d >>= 1) {
        __syncthreads();
        if (pi >= d && pi < 2 * d) {
            xyzz_to_words(tmp, acc);
#pragma unroll
            for (int l = 0; l < LANE_POINT_WORDS; l++) lds[l][t] = tmp[l];
        }
        __syncthreads();
        if (pi < d) {
#pragma unroll
            for (int l = 0; l < LANE_POINT_WORDS; l++) tmp[l] = lds[l][t + d * LP];
            Xyzz<T> q;
            xyzz_from_words(q, tmp);
            xyzz_add_impl(acc, q);
        }
    }
}
template <class T> FF_INLINE void fixup_big_body(const TailJob& job, uint32_t (*lds)[256]) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint32_t* __restrict__ offsets = job.offsets;
    uint8_t* __restrict__ buckets = job.buckets;
    const uint8_t* __restrict__ head = job.head;
    const uint8_t* __restrict__ tail = job.tail;
    const uint32_t* __restrict__ worklist = job.worklist;
    const uint32_t chunk = job.chunk;
    constexpr uint32_t LP = Lanes<T>::N, NP = 256 / LP;
    const uint32_t count = worklist[0];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {        // block-uniform loop
        const uint32_t kb = worklist[1 + i];
        const uint32_t s = offsets[kb], e = offsets[kb + 1];
        const uint32_t t0 = s / chunk, t1 = (e - 1) / chunk;
        Xyzz<T> acc = xyzz_inf<T>();
        for (uint32_t t = t0 + threadIdx.x / LP; t <= t1; t += NP) {
            const uint8_t* src = (t == t0 && s != t0 * chunk) ? tail : head;
            const Xyzz<T> q = xyzz_load_raw<T>(src + (uint64_t)XB * t);
            xyzz_add_impl(acc, q);
        }
        block_tree_sum<T, 256>(acc, lds);
        if (threadIdx.x < LP) xyzz_store_raw<T>(buckets + (uint64_t)XB * kb, acc);
        __syncthreads();
    }
}
__global__ __launch_bounds__(256, 2) void k_msm_fixup_big(TailJobs jobs) {
    __shared__ uint32_t lds[LANE_POINT_WORDS][256];
    if (blockIdx.z < jobs.n1) fixup_big_body<Fp>(jobs.j[blockIdx.z], lds);
    else fixup_big_body<Fp2H>(jobs.j[blockIdx.z], lds);
}

// ------------------------------------------------------------------ bucket reduction: R = sum_w w * B_w, w = b + 1
// A lone wave issues one instruction every ~4 cycles, so a chain of dependent EC additions costs
// ~15 us per link whatever the chip is doing: the reduction must be SHALLOW, not merely parallel.
// Write w = hi * 2^lb + lo.  Then R = sum_lo lo * S0[lo] + 2^lb * sum_hi hi * S1[hi] with the digit
// sums S0[d] = sum of buckets whose low digit is d, S1[d] = those whose high digit is d.  Two adds per bucket, ~2x the bucket reads (cheap).
// Empty buckets are recognised from the sort's offsets, so the bucket array is never cleared.
// The kernels below (one lane, or lane pair, per point) are the form for windows ABOVE 16 bits, where 2^17+ buckets make fixup and
// digit sums throughput-bound; up to 2^15 buckets, and for the weighting of the digit sums at every width, msm_tail.hip runs the chain
// on four slots per point.
// 16 points per digit value: every lane sums cnt/16 buckets serially, then a 4-level tree.  (One value per 64-point
// workgroup -- 2-4 buckets per lane, 6 levels -- finishes sooner but keeps four times as many waves busy for two thirds
// of that time; with a dozen proofs in flight SIMD time is what counts.)  NT threads hold NT / lanes points.
// ... G2 (lane pairs: an addition is ~27 us on a lone wave against ~13 us in G1) takes 32 points per digit value: 4-8 serial
// additions + 5 tree levels instead of 8-16 + 4, so the mixed-curve launch does not wait for the G2 chain twice as long
// WIDE (windows above 16 bits: a digit value sums 513-2048 buckets): 64 points per value for both curves, i.e. 8-32 serial
// additions + 6 tree levels instead of 32-128 + 4
// WG = points per digit value in the WIDE form (0: the narrow form).  Round 4: with a dozen proofs in flight the launch is throughput-bound, and the
// tree at the end of a value's sum keeps ever fewer lanes busy (64 points: 8-16 serial additions at full width, then 6 levels at 1/2, 1/4, ...: 72 % of
// the lane time does additions; 16 points: 32-64 serial + 4 levels: 94 %) -- ZK_DS_WIDE_GROUP selects 64 / 32 / 16 (A/B; the default is the measured best).
template <class T, int WG> struct DsGroup { static constexpr uint32_t N = WG ? WG : 16; };
template <int WG> struct DsGroup<Fp2H, WG> { static constexpr uint32_t N = WG ? WG : 32; };
static constexpr int DS_THREADS = 128;
static constexpr uint32_t DS_WIDE_GROUP_DEFAULT = 32;
static constexpr uint32_t RED_WAVES_DEFAULT = 2;
template <class T, int NT, int WG> FF_INLINE void digit_sums_body(const TailJob& job, DigitPlan p, uint32_t (*lds)[NT]) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint8_t* __restrict__ buckets = job.buckets;
    const uint32_t* __restrict__ offsets = job.offsets;
    uint8_t* __restrict__ S = job.red;
    constexpr uint32_t DS_GROUP = DsGroup<T, WG>::N;
    constexpr uint32_t LP = Lanes<T>::N, PER_WG = NT / LP / DS_GROUP;
    const uint32_t win = blockIdx.y, pt = threadIdx.x / LP, sub = pt / DS_GROUP, lane = pt % DS_GROUP;
    if (blockIdx.x * PER_WG >= p.nd0 + p.nd1) return;             // whole workgroup (the launch is sized for the smaller PER_WG)
    const uint32_t b = blockIdx.x * PER_WG + sub;
    const bool valid = b < p.nd0 + p.nd1;
    const uint64_t base = (uint64_t)win * p.nbw;
    const bool low = b < p.nd0;
    const uint32_t d = low ? b : b - p.nd0;
    const uint32_t cnt = low ? p.nd1 : p.nd0;
    Xyzz<T> acc = xyzz_inf<T>();
    if (valid && d != 0) {                          // weight 0 never contributes
        for (uint32_t e = lane; e < cnt; e += DS_GROUP) {
            const uint32_t w = low ? (e << p.lb) + d : (d << p.lb) + e;
            if (w >= 1 && w <= p.nbw && offsets[base + w] != offsets[base + w - 1])
                xyzz_add_raw_mem<T>(acc, buckets + (uint64_t)XB * (base + w - 1));
        }
    }
    block_tree_sum<T, NT, DS_GROUP>(acc, lds);
    if (valid && lane == 0) xyzz_store_raw<T>(S + (uint64_t)XB * ((uint64_t)win * (p.nd0 + p.nd1) + b), acc);
}
template <int WG, int WAVES> __global__ __launch_bounds__(DS_THREADS, WAVES) void k_msm_digit_sums(TailJobs jobs, DigitPlan p) {
    __shared__ uint32_t lds[LANE_POINT_WORDS][DS_THREADS];
    if (blockIdx.z < jobs.n1) digit_sums_body<Fp, DS_THREADS, WG>(jobs.j[blockIdx.z], p, lds);
    else digit_sums_body<Fp2H, DS_THREADS, WG>(jobs.j[blockIdx.z], p, lds);
}

static uint32_t red_waves() {          // a kernel-FORM switch: cached, read per call only under ZK_TEST_FORMS=1 (zk_common.h)
    const char* e = ZK_FORM_ENV("ZK_RED_WAVES");
    const uint32_t w = e ? (uint32_t)atoi(e) : RED_WAVES_DEFAULT;
    return w == 1 || w == 2 ? w : RED_WAVES_DEFAULT;
}
int msm_red_fixup_launch(const TailJobs& jobs, uint32_t count, uint32_t n2, bool by_chunk, uint64_t workers, uint32_t max_nb, hipStream_t s) {
    const uint32_t lanes_per = n2 ? 2 : 1;
    dim3 gf = grid_for(workers * lanes_per, 128);
    gf.z = count;
    const bool one = red_waves() == 1;
    if (by_chunk) {
        if (one) hipLaunchKernelGGL((k_msm_fixup<true, 1>), gf, dim3(128), 0, s, jobs);
        else hipLaunchKernelGGL((k_msm_fixup<true, 2>), gf, dim3(128), 0, s, jobs);
    } else {
        if (one) hipLaunchKernelGGL((k_msm_fixup<false, 1>), gf, dim3(128), 0, s, jobs);
        else hipLaunchKernelGGL((k_msm_fixup<false, 2>), gf, dim3(128), 0, s, jobs);
    }
    hipLaunchKernelGGL(k_msm_fixup_big, dim3(max_nb < 256 ? max_nb : 256, 1, count), dim3(256), 0, s, jobs);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int msm_red_digit_sums_launch(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, const DigitPlan& dp, bool wide, uint32_t wide_group, hipStream_t s) {
    const uint32_t wg = !wide ? 0 : (wide_group == 16 || wide_group == 32 || wide_group == 64 ? wide_group : DS_WIDE_GROUP_DEFAULT);
    const uint32_t grp = wg ? wg : (n2 ? 32u : 16u);          // DsGroup<..>::N of the curve with fewer digit values per workgroup (G2 when present)
    const uint32_t per_wg = n2 ? DS_THREADS / 2 / grp : DS_THREADS / grp;
    const dim3 gd((dp.nd0 + dp.nd1 + per_wg - 1) / per_wg, nwin, count);
    const bool one = red_waves() == 1;
#define ZK_DS_LAUNCH(WG)                                                                                              \
    do {                                                                                                              \
        if (one) hipLaunchKernelGGL((k_msm_digit_sums<WG, 1>), gd, dim3(DS_THREADS), 0, s, jobs, dp);                 \
        else hipLaunchKernelGGL((k_msm_digit_sums<WG, 2>), gd, dim3(DS_THREADS), 0, s, jobs, dp);                     \
    } while (0)
    if (wg == 64) ZK_DS_LAUNCH(64);
    else if (wg == 32) ZK_DS_LAUNCH(32);
    else if (wg == 16) ZK_DS_LAUNCH(16);
    else hipLaunchKernelGGL((k_msm_digit_sums<0, 1>), gd, dim3(DS_THREADS), 0, s, jobs, dp);          // the narrow form (small keys: latency-bound) stays at one wave
#undef ZK_DS_LAUNCH
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
