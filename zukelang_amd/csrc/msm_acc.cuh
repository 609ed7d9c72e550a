// Bucket accumulation of the Pippenger MSM (step 4 of msm.hip): shared by msm_acc_g1.hip (field products
// expanded in place) and msm_acc_g2.hip (lane-pair Fp2, products out of line).
#pragma once
#include "ec.cuh"
#include "msm.cuh"

#include <type_traits>

namespace zk {

// one table record (ec.cuh: TabRec, the lane's 128-byte line of the entry) -> the affine operand of the mixed addition; no repacking, the
// conditional negation of y is the only arithmetic (table entries are canonical: -y = 2p - y)
template <class F> FF_INLINE TabRec table_rec_load(const uint8_t* table, uint32_t ref) {
    if constexpr (std::is_same<F, Fp2H>::value) return tab_rec_load(table + (uint64_t)TableLayout<F>::ENTRY * (ref & 0x7fffffffu) + TAB_REC * pair_comp());
    else return tab_rec_load(table + (uint64_t)TableLayout<F>::ENTRY * (ref & 0x7fffffffu));
}
template <class F> FF_INLINE Aff<F> table_rec_point(const TabRec& a, bool negate) {
    const FpB<1> x = tab_rec_x(a), y = tab_rec_y(a);
    Aff<F> r;
    if constexpr (std::is_same<F, Fp2H>::value) {
        r.x = Fp2H(x);
        r.y = negate ? Fp2H(fe_neg(y)) : Fp2H(y);
    } else {
        r.x = x;
        if (negate) r.y = fe_neg(y);
        else r.y = y;
    }
    return r;
}

// ---- zz and zzz of a lane's accumulator PARKED in LDS between their two uses (G2 on lane pairs, products expanded in place): they are read at the top
// of a mixed addition (U2 = x2 ZZ, S2 = y2 ZZZ) and again at its end (ZZ PP, ZZZ PPP); in between they only occupied 28 of the 256 registers a lane has
// at two waves per SIMD, and the kernel spilled 46.  Layout [zz | zzz][16-byte piece][lane]: conflict-free 128-bit LDS accesses.
FF_INLINE uint32_t* park_limbs(Fp& a) { return a.v; }
FF_INLINE uint32_t* park_limbs(Fp2H& a) { return a.v.v; }
FF_INLINE const uint32_t* park_limbs(const Fp& a) { return a.v; }
FF_INLINE const uint32_t* park_limbs(const Fp2H& a) { return a.v.v; }
template <class F> struct ZPark {
    uint4* buf;                                              // [2][4][128] in LDS
    uint32_t lane;
    FF_INLINE F get(int which) const {
        asm volatile("" ::: "memory");                      // a fresh read every time: the point of parking is NOT to keep the value in registers
        uint32_t w[16];
#pragma unroll
        for (int p = 0; p < 4; p++) {
            const uint4 x = buf[(which * 4 + p) * 128 + lane];
            w[4 * p] = x.x; w[4 * p + 1] = x.y; w[4 * p + 2] = x.z; w[4 * p + 3] = x.w;
        }
        F r;
#pragma unroll
        for (int i = 0; i < FPL; i++) park_limbs(r)[i] = w[i];
        return r;
    }
    FF_INLINE void put(int which, const F& a) {
        const uint32_t* l = park_limbs(a);
        buf[(which * 4 + 0) * 128 + lane] = make_uint4(l[0], l[1], l[2], l[3]);
        buf[(which * 4 + 1) * 128 + lane] = make_uint4(l[4], l[5], l[6], l[7]);
        buf[(which * 4 + 2) * 128 + lane] = make_uint4(l[8], l[9], l[10], l[11]);
        buf[(which * 4 + 3) * 128 + lane] = make_uint4(l[12], l[13], 0u, 0u);
        asm volatile("" ::: "memory");
    }
};
// xyzz_madd_impl<F, false> (ec.cuh) with acc = (ax, ay, park[0], park[1]): the same operations in the same order on the same values
template <class F> FF_INLINE void xyzz_madd_parked(F& ax, F& ay, ZPark<F>& pk, const Aff<F>& q) {
    {
        const F zz = pk.get(0);
        if (fe_is_zero(zz)) {                            // the accumulator is the identity
            ax = q.x;
            ay = q.y;
            pk.put(0, FieldOps<F>::one());
            pk.put(1, FieldOps<F>::one());
            return;
        }
    }
    const auto U2 = fe_mul(q.x, pk.get(0));
    const auto S2 = fe_mul(q.y, pk.get(1));
    const auto P = fe_sub(U2, ax);
    const auto R = fe_sub(S2, ay);
    if (fe_is_zero(P)) {
        Xyzz<F> t{ax, ay, pk.get(0), pk.get(1)};
        xyzz_madd_equal_x(t, fe_is_zero(R));
        ax = t.x;
        ay = t.y;
        pk.put(0, t.zz);
        pk.put(1, t.zzz);
        return;
    }
    const auto PP = fe_sqr(P);
    const auto PPP = fe_mul(P, PP);
    const auto Q = fe_mul(ax, PP);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), PPP, Q);
    const auto Y3 = fe_mul_sub(R, fe_sub(Q, X3), ay, PPP);
    ax = X3;
    ay = Y3;
    pk.put(0, F(fe_mul(pk.get(0), PP)));
    pk.put(1, F(fe_mul(pk.get(1), PPP)));
}

// RAW = false: the sorted entries are table references (index | sign).  RAW = true (the finisher of the batch-affine rounds,
// msm_ba.cuh): entry `pos` is the affine point at pts + pos * 2 * RawLayout<F>::ELEM in the raw limb layout, identity = (0, 0).
// GLDS = true (G1, table references): the record of step i+1 travels HBM -> LDS by LDS-DMA (global_load_lds_dwordx4, per-lane source address, no
// VGPR destination) while step i computes, and is read from LDS when its turn comes: the look-ahead costs 7 KiB of LDS per wave instead of 28
// registers per lane in a kernel that sits at the register limit.
// MMADD = true: the second step of a chunk uses the 6-product addition of two affine points (a second, 25 KB copy of the group law beside the
// hot loop); false: every step is the general mixed addition -- less code in the instruction cache for 4 more products once per chunk.
// Waves per SIMD the kernel is compiled for (msm.cuh: ACC_WAVES_G1 = ACC_WAVES_G2 = 2; the host side cuts the chunks for rounds of that many resident
// waves).  The G1 table-reference kernel with the LDS-DMA look-ahead would fit 168 registers = THREE waves without a spill; measured, it is no faster
// alone or pipelined (msm.cuh), so every instantiation runs at two.
template <class F, bool RAW, bool GLDS> struct AccWaves { static constexpr int N = (std::is_same<F, Fp>::value && !RAW && GLDS) ? (int)ACC_WAVES_G1 : (int)ACC_WAVES_G2; };
template <class F, bool RAW, bool GLDS = false, bool MMADD = true>
__global__ __launch_bounds__(128, (AccWaves<F, RAW, GLDS>::N)) void k_msm_accumulate(const uint8_t* __restrict__ table, AccJobs jobs, uint32_t nb, uint32_t chunk) {
    static_assert(!GLDS || !RAW, "the LDS-DMA look-ahead serves table references");
    constexpr bool PARK = GLDS && std::is_same<F, Fp2H>::value && !MMADD;          // G2 with the products expanded in place: ZZ / ZZZ parked in LDS
    __shared__ uint4 park_buf[PARK ? 2 : 1][PARK ? 4 : 1][PARK ? 128 : 1];
    ZPark<F> zpark{&park_buf[0][0][0], threadIdx.x};
    __shared__ uint4 la_buf[GLDS ? 2 : 1][GLDS ? TAB_REC_WORDS / 4 : 1][GLDS ? 64 : 1];      // [wave][16-byte piece][lane]
    const uint32_t* __restrict__ offsets = jobs.offsets[blockIdx.y];
    const uint32_t* __restrict__ sorted = jobs.sorted[blockIdx.y];
    uint8_t* __restrict__ buckets = jobs.buckets[blockIdx.y];
    uint8_t* __restrict__ head = jobs.head[blockIdx.y];
    uint8_t* __restrict__ tail = jobs.tail[blockIdx.y];
    constexpr int XB = RawLayout<F>::XYZZ;       // sums: raw layout
    constexpr bool PAIR = std::is_same<F, Fp2H>::value;      // G2: two lanes per chunk, one Fp2 component each
    const uint64_t t = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> (PAIR ? 1 : 0);
    const uint32_t N = offsets[nb];
    uint64_t start64 = t * chunk;
    if (start64 >= N) return;
    uint32_t pos = (uint32_t)start64;
    const uint32_t pos0 = pos;
    const uint32_t end = min(pos + chunk, N);
    // largest kb with offsets[kb] <= pos (then offsets[kb+1] > pos, so the bucket is non-empty)
    uint32_t lo = 0, hi = nb;
    while (hi - lo > 1) {
        uint32_t mid = (lo + hi) >> 1;
        if (offsets[mid] <= pos) lo = mid; else hi = mid;
    }
    uint32_t kb = lo;
    // Flat loop: one mixed addition per lane per iteration whatever the run boundaries are (a loop
    // nest over runs would make the wave pay the longest run of every lane in turn).  A lane
    // crossing into the next bucket stores its running sum first -- a short divergent epilogue.
    // The reference and the table entry of step i+1 are requested before the mixed addition of step i: the gather latency hides behind ~5k ALU instructions.
    // Order inside an iteration: take the record of this step (its transfer was issued one whole group addition ago) -> run border, if this
    // lane crossed one (stores of the finished sum; the end of the NEXT run is already in a register, prefetched at the previous border:
    // no dependent load on the path) -> request the record of the next step -> the group addition.  With the border after the request every
    // wave stalled for a full HBM round trip whenever ANY of its lanes crossed a run border (the dependent offsets load waits for
    // everything older, the gather just issued included): at 2^20 that is 47 % of the iterations (PMC: 13 % of the wave cycles in s_waitcnt).
    uint32_t bstart = offsets[kb], bend = offsets[kb + 1];
    uint32_t bend2 = kb + 2 <= nb ? offsets[kb + 2] : bend;          // end of the run after this one
    uint32_t seg_start = pos;
    bool first = true;
    Xyzz<F> acc = xyzz_inf<F>();
    if constexpr (PARK) { zpark.put(0, acc.zz); zpark.put(1, acc.zzz); }          // PARK: acc.zz / acc.zzz of the struct are scratch, the truth is in LDS
    // (G2: the products are calls, which drain outstanding loads anyway -- only the reference is fetched ahead there)
    constexpr int RB = RawLayout<F>::ELEM;
    const uint8_t* __restrict__ pts = jobs.pts[blockIdx.y];
    uint32_t v_next = 0, v_next2 = 0;
    TabRec p_next;
    const uint32_t la_wave = (threadIdx.x >> 6) & 1u, la_lane = threadIdx.x & 63u;
    auto la_issue = [&](uint32_t ref) {             // the 112 bytes of one record: seven 16-byte pieces, piece j of lane l at la_buf[wave][j][l]
        const uint8_t* src = table + (uint64_t)TableLayout<F>::ENTRY * (ref & 0x7fffffffu) + (PAIR ? TAB_REC * pair_comp() : 0u);
#pragma unroll
        for (int j = 0; j < TAB_REC_WORDS / 4; j++)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + 16 * j),
                                             (__attribute__((address_space(3))) void*)&la_buf[la_wave][j][0], 16, 0, 0);
    };
    auto la_take = [&]() {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");          // the DMA of this record was issued one whole group addition ago
        TabRec r;
#pragma unroll
        for (int j = 0; j < TAB_REC_WORDS / 4; j++) {
            const uint4 x = la_buf[la_wave][j][la_lane];
            r.w[4 * j] = x.x; r.w[4 * j + 1] = x.y; r.w[4 * j + 2] = x.z; r.w[4 * j + 3] = x.w;
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");        // the reads are done before the next DMA may overwrite the slot
        return r;
    };
    if constexpr (!RAW) {
        v_next = sorted[pos];
        if constexpr (GLDS) {
            v_next2 = pos + 1 < end ? sorted[pos + 1] : 0u;
            la_issue(v_next);
        } else if constexpr (!PAIR) p_next = table_rec_load<F>(table, v_next);
    }
    for (; pos < end; pos++) {
        const uint32_t v = v_next;
        TabRec pk;
        if constexpr (GLDS) pk = la_take();
        else if constexpr (!RAW && !PAIR) pk = p_next;
        if (pos == bend) {                              // run finished inside the chunk
            const bool complete = seg_start == bstart;
            uint8_t* dst = complete ? buckets + (uint64_t)XB * kb : (first ? head + (uint64_t)XB * t : tail + (uint64_t)XB * t);
            if constexpr (PARK) { acc.zz = zpark.get(0); acc.zzz = zpark.get(1); }
            xyzz_store_raw<F>(dst, acc);
            first = false;
            acc = xyzz_inf<F>();
            if constexpr (PARK) { zpark.put(0, acc.zz); zpark.put(1, acc.zzz); }
            kb++; bstart = bend; bend = bend2;
            while (bend == bstart) { kb++; bend = offsets[kb + 1]; }        // empty buckets: rare, the only dependent load left
            bend2 = kb + 2 <= nb ? offsets[kb + 2] : bend;                    // consumed at the next border
            seg_start = pos;
        }
        if constexpr (GLDS) {
            if (pos + 1 < end) la_issue(v_next2);                  // its reference was loaded one step ago: no dependent-load stall here
            v_next = v_next2;
            v_next2 = pos + 2 < end ? sorted[pos + 2] : 0u;
        } else if constexpr (!RAW) {
            if constexpr (PAIR) pk = table_rec_load<F>(table, v);
            if (pos + 1 < end) {
                v_next = sorted[pos + 1];
                if constexpr (!PAIR) p_next = table_rec_load<F>(table, v_next);
            }
        }
        Aff<F> p;
        if constexpr (RAW) p = {load_raw_f((const F*)nullptr, pts + (uint64_t)2 * RB * pos), load_raw_f((const F*)nullptr, pts + (uint64_t)2 * RB * pos + RB)};
        else p = table_rec_point<F>(pk, (v >> 31) != 0);
        // The mixed addition is inlined so the accumulator lives in VGPRs for the whole chunk.  G2 runs
        // as F = Fp2H, one Fp2 component per lane of a pair, which gives it the register footprint of G1.
        // Second step of the chunk (a wave-uniform test): every lane holds the identity (run border just crossed) or
        // the single affine point of step one, so the 6-product addition of two affine points does.
        // Table entries are never the identity (the sort filters identity bases): only the raw-point path tests for it.
        if constexpr (PARK) xyzz_madd_parked<F>(acc.x, acc.y, zpark, p);
        else if constexpr (MMADD) {
            if (pos == pos0 + 1) xyzz_mmadd_impl<F, RAW>(acc, p);
            else xyzz_madd_impl<F, RAW>(acc, p);
        } else xyzz_madd_impl<F, RAW>(acc, p);
    }
    {
        const bool complete = (seg_start == bstart) && (end == bend);
        uint8_t* dst = complete ? buckets + (uint64_t)XB * kb : (first ? head + (uint64_t)XB * t : tail + (uint64_t)XB * t);
        if constexpr (PARK) { acc.zz = zpark.get(0); acc.zzz = zpark.get(1); }
        xyzz_store_raw<F>(dst, acc);
    }
}

}  // namespace zk
