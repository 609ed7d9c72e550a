// Bucket reduction (steps 5-7 of msm_run) with every point spread over four slots of a wave: ec_slots.cuh explains the group law,
// msm_tail.cuh the buffers.  A dependent addition costs 4 product times instead of the 14 of one lane per point:
//
//   fixup         a bucket whose run of sorted entries was cut by chunk borders: the partial sums of its chunks, one GROUP per bucket (OPTIONAL,
//                 ZK_TAIL_FIXUP_SLOTS=1: this step is throughput-bound and runs on lanes by default, msm_tail.cuh)
//   digit_sums    S0[d] = sum of the buckets whose low digit is d, S1[d] = those whose high digit is d   (16-64 groups per digit value + LDS tree)
//   block_weight  sum_j j * S[d] and sum_j S[d] over blocks of 8-64 digit values (suffix scan + tree through LDS, one small workgroup per block)
//   combine       the blocks of both halves -> the product: 2^lb * V1 + V0, one workgroup per product
//   windows       Horner over the windows when there is one bucket set per window (bases without resident window tables)
//
// Computes the same group element as the reference's sum of G.mul over the query (src/lib/zk/curve.ml:159-191; groth16.ml:116-161 calls it
// through the QAP products): only the association order of the additions differs.
#include "msm_tail.cuh"
#include "ec_slots.cuh"

namespace zk {

template <class T> FF_INLINE T load_slot(const uint8_t* point, uint32_t coord) {
    return load_raw_f((const T*)nullptr, point + (uint32_t)RawLayout<T>::ELEM * coord);
}
template <class T> FF_INLINE void store_slot(uint8_t* point, uint32_t coord, const T& own) {
    store_raw_f(point + (uint32_t)RawLayout<T>::ELEM * coord, own);
}
// own += the point at `point` (raw layout in memory)
template <class T> FF_INLINE void add_from_memory(T& own, const uint8_t* point) {
    const uint32_t sl = slot_id<T>();
    const T qs = load_slot<T>(point, sl), qx = load_slot<T>(point, sl ^ 2);
    xyzz_add_slots(own, qs, qx);
}
// the 14 limbs of a lane <-> one LDS column
template <class T, int NC> FF_INLINE void lds_put(uint32_t (*lds)[NC], uint32_t col, const T& v) {
#pragma unroll
    for (int l = 0; l < FPL; l++) lds[l][col] = slot_limbs(v)[l];
}
template <class T, int NC> FF_INLINE T lds_get(uint32_t (*lds)[NC], uint32_t col) {
    T r;
#pragma unroll
    for (int l = 0; l < FPL; l++) slot_limbs(r)[l] = lds[l][col];
    return r;
}
// own += the point whose lanes wrote their limbs to the columns [col0, col0 + G) (col = column of THIS lane's counterpart there)
template <class T, int NC> FF_INLINE void add_from_lds(T& own, uint32_t (*lds)[NC], uint32_t col) {
    const T qs = lds_get<T, NC>(lds, col), qx = lds_get<T, NC>(lds, col ^ (2 * SlotGeom<T>::LP));
    xyzz_add_slots(own, qs, qx);
}
// sum of the points of a workgroup (GROUP = points per independent sum, a power of two; 0 = all of them): result in point 0 of every group
template <class T, int NT, int GROUP = 0> FF_INLINE void tree_sum_slots(T& own, uint32_t (*lds)[NT]) {
    constexpr uint32_t G = SlotGeom<T>::G;
    constexpr uint32_t GP = GROUP ? GROUP : NT / G;
    const uint32_t t = threadIdx.x, pi = (t / G) & (GP - 1);
    for (uint32_t d = GP / 2; d >= 1; d >>= 1) {
        __syncthreads();
        if (pi >= d && pi < 2 * d) lds_put<T, NT>(lds, t, own);
        __syncthreads();
        if (pi < d) add_from_lds<T, NT>(own, lds, t + d * G);
    }
}

// ------------------------------------------------------------------ fixup
template <class T> FF_INLINE void fixup_body(const TailJob& job) {
    constexpr uint32_t G = SlotGeom<T>::G;
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint32_t kb = (blockIdx.x * blockDim.x + threadIdx.x) / G, sl = slot_id<T>();
    if (kb >= job.nb) return;
    const uint32_t s = job.offsets[kb], e = job.offsets[kb + 1];
    if (e == s) return;                                 // empty bucket: nobody reads its slot
    const uint32_t chunk = job.chunk, t0 = s / chunk, t1 = (e - 1) / chunk;
    if (t0 == t1) return;                               // whole run inside one chunk: written directly
    if (t1 - t0 > FIXUP_SERIAL_MAX) {
        if ((threadIdx.x & (G - 1)) == 0) job.worklist[1 + atomicAdd(&job.worklist[0], 1u)] = kb;
        return;
    }
    T own = load_slot<T>((s == t0 * chunk ? job.head : job.tail) + (uint64_t)XB * t0, sl);
    const uint8_t* q = job.head + (uint64_t)XB * (t0 + 1);
    T qs = load_slot<T>(q, sl), qx = load_slot<T>(q, sl ^ 2);
    for (uint32_t t = t0 + 1; t <= t1; t++) {
        T ns = qs, nx = qx;
        if (t < t1) {                                   // the next partial sum travels while this one is added
            q += XB;
            ns = load_slot<T>(q, sl);
            nx = load_slot<T>(q, sl ^ 2);
        }
        xyzz_add_slots(own, qs, qx);
        qs = ns;
        qx = nx;
    }
    store_slot<T>(job.buckets + (uint64_t)XB * kb, sl, own);
}
static constexpr int FX_THREADS = 128, FXB_THREADS = 256;
__global__ __launch_bounds__(FX_THREADS) void k_tail_fixup(TailJobs jobs) {
    if (blockIdx.z < jobs.n1) fixup_body<Fp>(jobs.j[blockIdx.z]);
    else fixup_body<Fp2H>(jobs.j[blockIdx.z]);
}
// buckets with more partial sums than one group should chain: a workgroup each
template <class T> FF_INLINE void fixup_big_body(const TailJob& job, uint32_t (*lds)[FXB_THREADS]) {
    constexpr uint32_t G = SlotGeom<T>::G, NP = FXB_THREADS / G;
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint32_t chunk = job.chunk, sl = slot_id<T>();
    const uint32_t count = job.worklist[0];
    for (uint32_t i = blockIdx.x; i < count; i += gridDim.x) {        // block-uniform loop
        const uint32_t kb = job.worklist[1 + i];
        const uint32_t s = job.offsets[kb], e = job.offsets[kb + 1];
        const uint32_t t0 = s / chunk, t1 = (e - 1) / chunk;
        T own = slot_zero<T>();
        for (uint32_t t = t0 + threadIdx.x / G; t <= t1; t += NP)
            add_from_memory<T>(own, ((t == t0 && s != t0 * chunk) ? job.tail : job.head) + (uint64_t)XB * t);
        tree_sum_slots<T, FXB_THREADS>(own, lds);
        if (threadIdx.x < G) store_slot<T>(job.buckets + (uint64_t)XB * kb, sl, own);
        __syncthreads();
    }
}
__global__ __launch_bounds__(FXB_THREADS) void k_tail_fixup_big(TailJobs jobs) {
    __shared__ uint32_t lds[FPL][FXB_THREADS];
    if (blockIdx.z < jobs.n1) fixup_big_body<Fp>(jobs.j[blockIdx.z], lds);
    else fixup_big_body<Fp2H>(jobs.j[blockIdx.z], lds);
}

// ------------------------------------------------------------------ digit sums
// DS points per digit value: every group sums cnt / DS buckets serially, then log2(DS) tree levels.  A G2 link costs about twice a G1 link, so G2
// takes twice the points (half the serial part); WIDE (windows above 16 bits: a digit value sums 513-2048 buckets): 64 points for both curves.
template <class T, bool WIDE> struct DsPoints { static constexpr uint32_t N = WIDE ? 64 : 16; };
template <bool WIDE> struct DsPoints<Fp2H, WIDE> { static constexpr uint32_t N = WIDE ? 64 : 32; };
template <bool WIDE> struct DsThreads { static constexpr int N = WIDE ? 512 : 256; };     // = one digit value of G2 per workgroup
template <class T, int NT, bool WIDE> FF_INLINE void digit_sums_body(const TailJob& job, DigitPlan p, uint32_t (*lds)[NT]) {
    constexpr int XB = RawLayout<T>::XYZZ;
    constexpr uint32_t DS = DsPoints<T, WIDE>::N, G = SlotGeom<T>::G, PER_WG = NT / G / DS;
    const uint32_t win = blockIdx.y, pt = threadIdx.x / G, sub = pt / DS, lane = pt % DS, sl = slot_id<T>();
    if (blockIdx.x * PER_WG >= p.nd0 + p.nd1) return;             // whole workgroup (the launch is sized for the smaller PER_WG)
    const uint32_t b = blockIdx.x * PER_WG + sub;
    const bool valid = b < p.nd0 + p.nd1;
    const uint64_t base = (uint64_t)win * p.nbw;
    const bool low = b < p.nd0;
    const uint32_t d = low ? b : b - p.nd0;
    const uint32_t cnt = low ? p.nd1 : p.nd0;
    const uint32_t* __restrict__ offsets = job.offsets;
    T own = slot_zero<T>();
    if (valid && d != 0) {                          // weight 0 never contributes
        for (uint32_t e = lane; e < cnt; e += DS) {
            const uint32_t w = low ? (e << p.lb) + d : (d << p.lb) + e;
            if (w >= 1 && w <= p.nbw && offsets[base + w] != offsets[base + w - 1]) add_from_memory<T>(own, job.buckets + (uint64_t)XB * (base + w - 1));
        }
    }
    tree_sum_slots<T, NT, DS>(own, lds);
    if (valid && lane == 0) store_slot<T>(job.red + (uint64_t)XB * ((uint64_t)win * (p.nd0 + p.nd1) + b), sl, own);
}
template <bool WIDE> __global__ __launch_bounds__(DsThreads<WIDE>::N) void k_tail_digit_sums(TailJobs jobs, DigitPlan p) {
    __shared__ uint32_t lds[FPL][DsThreads<WIDE>::N];
    if (blockIdx.z < jobs.n1) digit_sums_body<Fp, DsThreads<WIDE>::N, WIDE>(jobs.j[blockIdx.z], p, lds);
    else digit_sums_body<Fp2H, DsThreads<WIDE>::N, WIDE>(jobs.j[blockIdx.z], p, lds);
}

// ------------------------------------------------------------------ digit weights
// W = sum_{d < nd0} d * S0[d]  +  2^lb * sum_{d < nd1} d * S1[d]  in two launches of SHORT chains on few lanes (a workgroup that keeps 256 points in
// flight for 16 rounds is bound by the issue rate of its one compute unit, not by the chain):
//
//   block_weight  the digit values of a half go in blocks of BW = 2^bw (d = BW b + j): one small workgroup per block computes
//                 L_b = sum_j j * S[BW b + j]   and   T_b = sum_j S[BW b + j]
//                 as the sum of all suffix sums -- bw scan rounds ("point j += point j + 2^r"; point 0 then holds T_b and is dropped: weight 0)
//                 and bw tree rounds, every one an exchange through LDS.
//   combine       one workgroup per product and window: X_k = sum_b b * T_k[b] the same way on <= 32 points per half (5 + 5 rounds), with the plain
//                 sums Y_k = sum_b L_k[b] riding in the groups the tree rounds leave idle, then one group runs
//                 W = ((X_1 * 2^bw + Y_1) * 2^(lb - bw) + X_0) * 2^bw + Y_0:   lb + bw doublings and three additions.
// bw = lb - 5 (at most 32 blocks per half; bw = 0 below 11-bit windows: no first launch, combine reads the digit sums themselves).
// c = 16: 6 + 10 additions, then 11 doublings + 3 additions -- every link at the latency of one group (4 product times, 3 for a doubling) instead of
// 16 rounds of a full compute unit + 8 doublings + 2 additions.
static constexpr uint32_t CB_POINTS = 32;                        // blocks per half the combine step takes
static constexpr int BWT_MAX_THREADS = 512;                       // 64 points of G2
template <class T> FF_INLINE void block_weight_body(const TailJob& job, DigitPlan p, uint32_t bw, uint32_t (*lds)[BWT_MAX_THREADS]) {
    constexpr int XB = RawLayout<T>::XYZZ;
    constexpr uint32_t G = SlotGeom<T>::G;
    const uint32_t BW = 1u << bw, B0 = p.nd0 >> bw, B1 = (p.nd1 + BW - 1) >> bw;
    const uint32_t k = blockIdx.x < B0 ? 0u : 1u, blk = k ? blockIdx.x - B0 : blockIdx.x;
    const uint32_t win = blockIdx.y, t = threadIdx.x, pt = t / G, sl = slot_id<T>();
    const bool mine = pt < BW;                                   // a G1 job uses half of the lanes of the launch; everyone keeps the barriers
    const uint32_t cnt = k == 0 ? p.nd0 : p.nd1, d = (blk << bw) + pt;
    const uint8_t* base = job.red + (uint64_t)XB * ((uint64_t)win * (p.nd0 + p.nd1) + (k == 0 ? 0 : p.nd0));
    uint8_t* out = job.wsum + (uint64_t)XB * 2 * ((uint64_t)win * (B0 + B1) + blockIdx.x);        // T_b | L_b
    T own = slot_zero<T>();
    if (mine && d >= 1 && d < cnt) own = load_slot<T>(base + (uint64_t)XB * d, sl);
    for (uint32_t r = 0; r < 2 * bw; r++) {
        const bool scan = r < bw;
        const uint32_t step = scan ? (1u << r) : (BW / 2) >> (r - bw);
        if (r == bw && pt == 0) {
            store_slot<T>(out, sl, own);
            own = slot_zero<T>();
        }
        __syncthreads();
        if (mine) lds_put<T, BWT_MAX_THREADS>(lds, t, own);
        __syncthreads();
        if (mine && (scan ? pt + step < BW : pt < step)) add_from_lds<T, BWT_MAX_THREADS>(own, lds, t + step * G);
    }
    if (pt == 0) store_slot<T>(out + XB, sl, own);
}
__global__ __launch_bounds__(BWT_MAX_THREADS) void k_tail_block_weight(TailJobs jobs, DigitPlan p, uint32_t bw) {
    __shared__ uint32_t lds[FPL][BWT_MAX_THREADS];
    if (blockIdx.z < jobs.n1) block_weight_body<Fp>(jobs.j[blockIdx.z], p, bw, lds);
    else block_weight_body<Fp2H>(jobs.j[blockIdx.z], p, bw, lds);
}

static constexpr int CB_THREADS = 2 * CB_POINTS * 8;              // two halves x 32 points x the 8 lanes of a G2 point
static constexpr int CB_COLS = 2 * CB_THREADS;                    // the X points, then the Y points
template <class T> FF_INLINE void combine_body(const TailJob& job, DigitPlan p, uint32_t bw, uint32_t nwin, uint32_t (*lds)[CB_COLS]) {
    constexpr int XB = RawLayout<T>::XYZZ;
    constexpr uint32_t G = SlotGeom<T>::G, YCOL = 2 * CB_POINTS * G;
    const uint32_t BW = 1u << bw, B0 = p.nd0 >> bw, B1 = (p.nd1 + BW - 1) >> bw;
    const uint32_t win = blockIdx.y, t = threadIdx.x, grp = t / G, k = grp / CB_POINTS, pt = grp % CB_POINTS, sl = slot_id<T>();
    const bool mine = grp < 2 * CB_POINTS;                       // G1 jobs use half of the lanes
    const uint32_t nblk = k == 0 ? B0 : B1, lane = t & (G - 1);
    const uint32_t xcol = (k * CB_POINTS + pt) * G + lane;      // this lane's column of X_k[pt] (= t when mine); Y_k[pt] is YCOL further
    if (mine) {
        T x = slot_zero<T>(), y = slot_zero<T>();
        if (pt >= 1 && pt < nblk) {                             // block 0 has weight 0 in X
            if (bw) x = load_slot<T>(job.wsum + (uint64_t)XB * 2 * ((uint64_t)win * (B0 + B1) + (k ? B0 : 0) + pt), sl);
            else x = load_slot<T>(job.red + (uint64_t)XB * ((uint64_t)win * (p.nd0 + p.nd1) + (k ? p.nd0 : 0) + pt), sl);
        }
        if (bw && pt < nblk) y = load_slot<T>(job.wsum + (uint64_t)XB * (2 * ((uint64_t)win * (B0 + B1) + (k ? B0 : 0) + pt) + 1), sl);
        lds_put<T, CB_COLS>(lds, xcol, x);
        lds_put<T, CB_COLS>(lds, YCOL + xcol, y);
    }
    // rounds 0..4: suffix scan of X;  rounds 5..9: tree over X (points 1..31: point 0 is the plain sum, weight 0) and over Y in the idle groups
    for (uint32_t r = 0; r < 10; r++) {
        const bool scan = r < 5;
        const uint32_t step = scan ? (1u << r) : (CB_POINTS / 2) >> (r - 5);
        uint32_t dst = 0, src = 0;
        bool act = false, fresh = false;
        if (mine) {
            if (scan) { act = pt + step < CB_POINTS; dst = xcol; src = xcol + step * G; }
            else if (pt < step) { act = true; dst = xcol; src = xcol + step * G; fresh = r == 5 && pt == 0; }
            else if (pt < 2 * step) { act = true; dst = YCOL + xcol - step * G; src = dst + step * G; }
        }
        __syncthreads();
        T v = slot_zero<T>();
        if (act) {
            if (!fresh) v = lds_get<T, CB_COLS>(lds, dst);
            add_from_lds<T, CB_COLS>(v, lds, src);
        }
        __syncthreads();
        if (act) lds_put<T, CB_COLS>(lds, dst, v);
    }
    __syncthreads();
    if (t >= G) return;
    // one group: ((X_1 * 2^bw + Y_1) * 2^(lb - bw) + X_0) * 2^bw + Y_0      (columns: X_0 at 0, X_1 at CB_POINTS * G)
    T acc = lds_get<T, CB_COLS>(lds, CB_POINTS * G + lane);
    for (uint32_t i = 0; i < bw; i++) acc = xyzz_dbl_slots(acc);
    if (bw) add_from_lds<T, CB_COLS>(acc, lds, YCOL + CB_POINTS * G + lane);
    for (uint32_t i = bw; i < p.lb; i++) acc = xyzz_dbl_slots(acc);
    add_from_lds<T, CB_COLS>(acc, lds, lane);
    for (uint32_t i = 0; i < bw; i++) acc = xyzz_dbl_slots(acc);
    if (bw) add_from_lds<T, CB_COLS>(acc, lds, YCOL + lane);
    if (nwin == 1) store_f(job.out + (uint32_t)(FieldOps<T>::WORDS * 4) * sl, acc);       // the product itself: DENSE, fully reduced (an output of the library)
    else store_slot<T>(job.wsum + (uint64_t)XB * (2 * (uint64_t)nwin * (B0 + B1) + win), sl, acc);      // W_win, behind the block sums: for the Horner step over the windows
}
__global__ __launch_bounds__(CB_THREADS) void k_tail_combine(TailJobs jobs, DigitPlan p, uint32_t bw, uint32_t nwin) {
    __shared__ uint32_t lds[FPL][CB_COLS];
    if (blockIdx.z < jobs.n1) combine_body<Fp>(jobs.j[blockIdx.z], p, bw, nwin, lds);
    else combine_body<Fp2H>(jobs.j[blockIdx.z], p, bw, nwin, lds);
}

// ------------------------------------------------------------------ one bucket set per window (bases without resident window tables)
// result = sum_j 2^(c*j) * W_j by Horner from the top window (one group).
template <class T> FF_INLINE void windows_body(const TailJob& job, uint32_t nw, uint32_t c, uint32_t nblocks) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint8_t* W = job.wsum + (uint64_t)XB * 2 * nw * nblocks;
    if (threadIdx.x >= SlotGeom<T>::G) return;
    const uint32_t sl = slot_id<T>();
    T own = slot_zero<T>();
    for (uint32_t j = nw; j-- > 0;) {
        if (j != nw - 1)
            for (uint32_t k = 0; k < c; k++) own = xyzz_dbl_slots(own);
        add_from_memory<T>(own, W + (uint64_t)XB * j);
    }
    store_f(job.out + (uint32_t)(FieldOps<T>::WORDS * 4) * sl, own);
}
__global__ __launch_bounds__(64) void k_tail_windows(TailJobs jobs, uint32_t nw, uint32_t c, uint32_t nblocks) {
    if (blockIdx.z < jobs.n1) windows_body<Fp>(jobs.j[blockIdx.z], nw, c, nblocks);
    else windows_body<Fp2H>(jobs.j[blockIdx.z], nw, c, nblocks);
}

// ================================================================== host side
int msm_tail_fixup_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t max_nb, hipStream_t s) {
    const uint32_t lanes_per_point = n2 ? SlotGeom<Fp2H>::G : SlotGeom<Fp>::G;
    ScopedTimer t1("msm_reduce:fixup", s);
    const uint64_t lanes = (uint64_t)max_nb * lanes_per_point;
    hipLaunchKernelGGL(k_tail_fixup, dim3((unsigned)((lanes + FX_THREADS - 1) / FX_THREADS), 1, count), dim3(FX_THREADS), 0, s, jobs);
    hipLaunchKernelGGL(k_tail_fixup_big, dim3(max_nb < 256 ? max_nb : 256, 1, count), dim3(FXB_THREADS), 0, s, jobs);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int msm_tail_digit_sums_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, uint32_t c, hipStream_t s) {
    const uint32_t lanes_per_point = n2 ? SlotGeom<Fp2H>::G : SlotGeom<Fp>::G;
    const DigitPlan dp = digit_plan(c);
    const bool wide = dp.nd0 > DW_POINTS;
    ScopedTimer t2("msm_reduce:digit_sums", s);
    // workgroups sized for the curve with fewer digit values per workgroup
    const uint32_t nt = wide ? DsThreads<true>::N : DsThreads<false>::N;
    const uint32_t per_wg = wide ? nt / lanes_per_point / (n2 ? DsPoints<Fp2H, true>::N : DsPoints<Fp, true>::N)
                                 : nt / lanes_per_point / (n2 ? DsPoints<Fp2H, false>::N : DsPoints<Fp, false>::N);
    const dim3 gd((dp.nd0 + dp.nd1 + per_wg - 1) / per_wg, nwin, count);
    if (wide) hipLaunchKernelGGL(k_tail_digit_sums<true>, gd, dim3(nt), 0, s, jobs, dp);
    else hipLaunchKernelGGL(k_tail_digit_sums<false>, gd, dim3(nt), 0, s, jobs, dp);
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
int msm_tail_weight_slots(const TailJobs& jobs, uint32_t count, uint32_t n2, uint32_t nwin, uint32_t c, hipStream_t s) {
    const uint32_t lanes_per_point = n2 ? SlotGeom<Fp2H>::G : SlotGeom<Fp>::G;
    const DigitPlan dp = digit_plan(c);
    const uint32_t bw = tail_bw_log(dp), B0 = dp.nd0 >> bw, B1 = (dp.nd1 + (1u << bw) - 1) >> bw;
    {
        ScopedTimer t3("msm_reduce:digit_weight", s);
        if (bw) hipLaunchKernelGGL(k_tail_block_weight, dim3(B0 + B1, nwin, count), dim3((1u << bw) * lanes_per_point), 0, s, jobs, dp, bw);
        hipLaunchKernelGGL(k_tail_combine, dim3(1, nwin, count), dim3(CB_POINTS * 2 * lanes_per_point), 0, s, jobs, dp, bw, nwin);
    }
    if (nwin > 1) {
        ScopedTimer t4("msm_reduce:final", s);
        hipLaunchKernelGGL(k_tail_windows, dim3(1, 1, count), dim3(64), 0, s, jobs, nwin, c, B0 + B1);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}

}  // namespace zk
