// Batch-affine bucket accumulation for the Pippenger MSM (step 4 of msm.hip), gfx950.
//
// After the counting sort a bucket's entries are consecutive.  Instead of one lane walking a run with XYZZ mixed additions
// (10 field products each, msm_acc.cuh), the runs are HALVED per round by adding neighbours in AFFINE coordinates,
//   lambda = (y2 - y1) / (x2 - x1),   x3 = lambda^2 - x1 - x2,   y3 = lambda (x1 - x3) - y1,
// with the inversions of a lane's L additions shared by Montgomery's trick:
//   pass A (forward)   running products of the denominators d_i; the product BEFORE item i is parked in item i's output slot
//   one inversion per lane (fp_inv.cuh: lockstep, ~45 products' worth: 0.35 per addition at L = 128)
//   pass B (backward)  1/d_i = inv_run * prefix_{i-1}, inv_run *= d_i, then the addition: 5 products + 1 square
// ~6.2 products per addition instead of 10, and no chunk borders to fix up.  Round r turns the entry array E_r (round 0: the
// sorted table references) and the bucket offsets off_r into E_{r+1}, off_{r+1}[b] = sum_{q<b} ceil(cnt_r[q] / 2); output
// item i = (bucket b, j) adds the inputs 2j and 2j+1 of the bucket, an odd last entry is copied.  After R rounds (chosen
// from the mean bucket load so that ~2-3 entries per bucket remain) the EXISTING chunked XYZZ accumulate + fix-up finishes
// whatever is left -- including arbitrarily skewed buckets -- reading affine points instead of table references.
//
// Exact group law, every special case branched: equal points (doubling: the denominator becomes 2 y1 and the numerator
// 3 x1^2, still inside the shared inversion), opposite points (result = identity), identity operands (rounds >= 1; the
// sort already filters identity BASES out).  Intermediate points live in a raw layout: x | y as 16-word slots of the 14
// register limbs (value < 4p); the identity is all-zero x limbs (a genuine x = 0 mod p is stored as p).
//
// Same group elements as the reference's left folds: G.apply_powers / G.dot (src/lib/zk/curve.ml:91-118).
#pragma once
#include "ec.cuh"
#include "fp_inv.cuh"
#include "msm.cuh"

#include <type_traits>

namespace zk {

// ---- value < K p (K <= 1024) -> the same residue below 4p: one quotient estimate from the top limbs, q p subtracted limb-wise
template <int A> FF_INLINE FpB<4> fp_reduce_small(const FpB<A>& a) {
    static_assert(A <= 4096, "fp_reduce_small: bound too large for the float estimate");
    if constexpr (A <= 4) return FpB<4>(a);
    else {
        uint32_t t[FPL];
#pragma unroll
        for (int i = 0; i < FPL; i++) t[i] = a.v[i];
        const float xf = (float)t[13] * 536870912.0f + (float)t[12];
        uint32_t q = (uint32_t)(xf * FP29_PTOP_INV);
        q = q ? q - 1 : 0;                                        // as fp_canon_call: q in {Q-2, Q-1, Q} -> remainder < 3p
        int64_t cy = 0;
#pragma unroll
        for (int i = 0; i < FPL; i++) {
            const int64_t cur = (int64_t)t[i] - (int64_t)((uint64_t)q * FP29_MOD[i]) + cy;
            t[i] = i < FPL - 1 ? ((uint32_t)cur & FP29_MASK) : (uint32_t)cur;
            cy = cur >> FP29_W;
        }
        FpB<4> r;
#pragma unroll
        for (int i = 0; i < FPL; i++) r.v[i] = t[i];
        return r;
    }
}
template <int A> FF_INLINE Fp2HB<4> fp_reduce_small(const Fp2HB<A>& a) { return {fp_reduce_small(a.v)}; }

// all limbs zero (the identity marker of the raw affine layout); pair-uniform for Fp2H
template <int A> FF_INLINE bool ba_exact_zero(const FpB<A>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) o |= a.v[i];
    return o == 0;
}
template <int A> FF_INLINE bool ba_exact_zero(const Fp2HB<A>& a) {
    const int z = ba_exact_zero(a.v) ? 1 : 0;
    return z && __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true);
}
// never store a genuine coordinate as all-zero limbs: 0 mod p becomes p
FF_INLINE FpB<4> ba_nonzero_rep(const FpB<4>& a) {
    const bool z = ba_exact_zero(a);
    FpB<4> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = z ? FP29_MOD[i] : a.v[i];
    return r;
}
FF_INLINE Fp2HB<4> ba_nonzero_rep(const Fp2HB<4>& a) {
    const bool z = ba_exact_zero(a);                       // both components zero
    Fp2HB<4> r;
    const bool c0 = pair_comp() == 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v.v[i] = (z && c0) ? FP29_MOD[i] : a.v.v[i];      // p + 0 u
    return r;
}

// ---- the tight coordinate type of the affine intermediates and its memory form
template <class F> struct BaTypes;
template <> struct BaTypes<Fp> {
    using T = FpB<4>;          // a stored coordinate
    using D = FpB<12>;         // a denominator: x2 - x1 of two stored coordinates
    using P = FpB<16>;         // the running product at rest (a denominator or a product)
    static FF_INLINE T one() { return FpB<4>(fp_one()); }
    static FF_INLINE T zero() { return FpB<4>(fp_zero()); }
};
template <> struct BaTypes<Fp2H> {
    using T = Fp2HB<4>;
    using D = Fp2HB<12>;
    using P = Fp2HB<16>;
    static FF_INLINE T one() { return {FpB<4>(fp_select(pair_comp() != 0, fp_one(), fp_zero()))}; }
    static FF_INLINE T zero() { return {FpB<4>(fp_zero())}; }
};
template <int B> FF_INLINE FpB<B> ba_select(bool take_b, const FpB<B>& a, const FpB<B>& b) { return fp_select(take_b, a, b); }
template <int B> FF_INLINE Fp2HB<B> ba_select(bool take_b, const Fp2HB<B>& a, const Fp2HB<B>& b) { return {fp_select(take_b, a.v, b.v)}; }
FF_INLINE FpB<4> ba_load_coord(const Fp*, const void* p) { return fp_assume<4>(fp_load_raw(p)); }
FF_INLINE Fp2HB<4> ba_load_coord(const Fp2H*, const void* p) { return {fp_assume<4>(fp_load_raw((const char*)p + 64 * pair_comp()))}; }
FF_INLINE FpB<16> ba_load_prod(const Fp*, const void* p) { return fp_assume<16>(fp_load_raw(p)); }
FF_INLINE Fp2HB<16> ba_load_prod(const Fp2H*, const void* p) { return {fp_assume<16>(fp_load_raw((const char*)p + 64 * pair_comp()))}; }
template <int A> FF_INLINE void ba_store_coord(void* p, const FpB<A>& a) { fp_store_raw(p, Fp(a)); }
template <int A> FF_INLINE void ba_store_coord(void* p, const Fp2HB<A>& a) { fp_store_raw((char*)p + 64 * pair_comp(), Fp(a.v)); }
template <class F> struct BaLayout { static constexpr int COORD = RawLayout<F>::ELEM, POINT = 2 * RawLayout<F>::ELEM; };

// one coordinate of a table entry (ec.cuh: 128-byte records of canonical limbs; 56 B per coordinate) -> tight register form; G2 lanes
// take their own record
FF_INLINE FpB<1> ba_table_limbs(const uint8_t* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);          // 56 B = 3 x 16 + 8, 8-byte aligned
    const uint2* q2 = reinterpret_cast<const uint2*>(p);
    FpB<1> r;
    if (((uintptr_t)p & 15) == 0) {
        const uint4 a = q[0], b = q[1], c = q[2];
        const uint2 d = q2[6];
        r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w; r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
        r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w; r.v[12] = d.x; r.v[13] = d.y;
    } else {
        const uint2 d = q2[0];
        const uint4* q1 = reinterpret_cast<const uint4*>(p + 8);
        const uint4 a = q1[0], b = q1[1], c = q1[2];
        r.v[0] = d.x; r.v[1] = d.y; r.v[2] = a.x; r.v[3] = a.y; r.v[4] = a.z; r.v[5] = a.w; r.v[6] = b.x; r.v[7] = b.y;
        r.v[8] = b.z; r.v[9] = b.w; r.v[10] = c.x; r.v[11] = c.y; r.v[12] = c.z; r.v[13] = c.w;
    }
    return r;
}
FF_INLINE FpB<4> ba_table_coord(const Fp*, const uint8_t* entry, int which, bool negate) {
    const FpB<1> c = ba_table_limbs(entry + 4 * FPL * which);
    if (negate) return FpB<4>(fe_neg(c));
    return FpB<4>(c);
}
FF_INLINE Fp2HB<4> ba_table_coord(const Fp2H*, const uint8_t* entry, int which, bool negate) {
    const FpB<1> c = ba_table_limbs(entry + TAB_REC * pair_comp() + 4 * FPL * which);
    if (negate) return {FpB<4>(fe_neg(c))};
    return {FpB<4>(c)};
}

// ---- per-round offsets: off_r[b] = sum_{q<b} ceil(cnt_0[q] / 2^r), r = 1..R, all rounds and jobs in ONE launch
struct BaPlanJobs {
    const uint32_t* off0[MAX_ACC_JOBS];
    uint32_t* offs[MAX_ACC_JOBS];          // R arrays of nb + 1 entries, round r at (r - 1) * (nb + 1)
};
__global__ __launch_bounds__(1024) void k_ba_plan(BaPlanJobs jobs, uint32_t nb) {
    const uint32_t* __restrict__ off0 = jobs.off0[blockIdx.y];
    const uint32_t r = blockIdx.x + 1;
    uint32_t* __restrict__ out = jobs.offs[blockIdx.y] + (uint64_t)blockIdx.x * (nb + 1);
    __shared__ uint32_t part[1024];
    const uint32_t t = threadIdx.x;
    const uint32_t per = (nb + 1023) / 1024;
    const uint32_t lo = t * per, hi = min(lo + per, nb);
    const uint32_t add = (1u << r) - 1;
    uint32_t s = 0;
    for (uint32_t k = lo; k < hi; k++) s += (off0[k + 1] - off0[k] + add) >> r;
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
        out[k] = run;
        run += (off0[k + 1] - off0[k] + add) >> r;
    }
    if (t == 1023) out[nb] = part[1023];
}

// ---- one halving round
struct BaJobs {
    const uint32_t* refs[MAX_ACC_JOBS];      // round 0: the sorted table references
    const uint8_t* src[MAX_ACC_JOBS];        // rounds >= 1: E_r, raw affine points
    uint8_t* dst[MAX_ACC_JOBS];              // E_{r+1}
    const uint32_t* off_in[MAX_ACC_JOBS];
    const uint32_t* off_out[MAX_ACC_JOBS];
    uint64_t cap[MAX_ACC_JOBS];              // capacity of dst in points (a lane never writes beyond it)
};

#ifndef BA_OCC
#define BA_OCC 2
#endif
enum BaKind : int { BA_ADD = 0, BA_DBL = 1, BA_COPY1 = 2, BA_COPY2 = 3, BA_INF = 4 };

// operands of one output item
template <class F, bool FIRST> struct BaItem {
    const uint8_t* p1;
    const uint8_t* p2;
    bool neg1, neg2, pair;
};
template <class F, bool FIRST>
FF_INLINE BaItem<F, FIRST> ba_item(const uint8_t* table, const uint32_t* refs, const uint8_t* src, uint32_t s0, bool pair) {
    BaItem<F, FIRST> it;
    it.pair = pair;
    if constexpr (FIRST) {
        constexpr int AB = TableLayout<F>::ENTRY;
        const uint32_t v1 = refs[s0], v2 = pair ? refs[s0 + 1] : 0u;
        it.p1 = table + (uint64_t)AB * (v1 & 0x7fffffffu);
        it.p2 = table + (uint64_t)AB * (v2 & 0x7fffffffu);
        it.neg1 = (v1 >> 31) != 0;
        it.neg2 = (v2 >> 31) != 0;
    } else {
        it.p1 = src + (uint64_t)BaLayout<F>::POINT * s0;
        it.p2 = it.p1 + BaLayout<F>::POINT;
        it.neg1 = it.neg2 = false;
    }
    return it;
}
template <class F, bool FIRST> FF_INLINE typename BaTypes<F>::T ba_coord(const uint8_t* p, int which, bool negate) {
    if constexpr (FIRST) return ba_table_coord((const F*)nullptr, p, which, which == 1 && negate);
    else return ba_load_coord((const F*)nullptr, p + BaLayout<F>::COORD * which);
}

// Classification of an item and its denominator d (x2 - x1 | 2 y1 | 1).  x1, x2 are loaded by the caller; the y coordinates
// only on the rare path (equal x).
template <class F, bool FIRST>
FF_INLINE int ba_classify(const BaItem<F, FIRST>& it, const typename BaTypes<F>::T& x1, const typename BaTypes<F>::T& x2, typename BaTypes<F>::D& d) {
    using T = typename BaTypes<F>::T;
    using D = typename BaTypes<F>::D;
    d = D(BaTypes<F>::one());
    if (!it.pair) return BA_COPY1;
    if constexpr (!FIRST) {                      // identity operands exist only after a cancellation in an earlier round
        if (ba_exact_zero(x1)) return BA_COPY2;
        if (ba_exact_zero(x2)) return BA_COPY1;
    }
    const D dx = fe_sub(x2, x1);
    if (!fe_is_zero(dx)) { d = dx; return BA_ADD; }
    const T y1 = ba_coord<F, FIRST>(it.p1, 1, it.neg1), y2 = ba_coord<F, FIRST>(it.p2, 1, it.neg2);
    if (!fe_is_zero(fe_sub(y2, y1))) return BA_INF;               // P + (-P)
    if (fe_is_zero(y1)) return BA_INF;                            // 2-torsion (does not exist on these curves)
    d = D(fe_dbl(y1));                                            // doubling: lambda = 3 x1^2 / (2 y1)
    return BA_DBL;
}

template <class F, bool FIRST>
__global__ __launch_bounds__(128, BA_OCC) void k_ba_round(const uint8_t* __restrict__ table, BaJobs jobs, uint32_t nb, uint32_t L) {
    using T = typename BaTypes<F>::T;
    constexpr bool PAIR = std::is_same<F, Fp2H>::value;
    constexpr int PB = BaLayout<F>::POINT, CB = BaLayout<F>::COORD;
    const uint32_t* __restrict__ refs = jobs.refs[blockIdx.y];
    const uint8_t* __restrict__ src = jobs.src[blockIdx.y];
    uint8_t* __restrict__ dst = jobs.dst[blockIdx.y];
    const uint32_t* __restrict__ off_in = jobs.off_in[blockIdx.y];
    const uint32_t* __restrict__ off_out = jobs.off_out[blockIdx.y];
    const uint64_t lane = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) >> (PAIR ? 1 : 0);
    uint64_t items = off_out[nb];
    if (items > jobs.cap[blockIdx.y]) items = jobs.cap[blockIdx.y];        // never beyond the buffer (sizes are upper bounds by construction)
    const uint64_t i0 = lane * L;
    if (i0 >= items) return;
    const uint32_t cnt = (uint32_t)min((uint64_t)L, items - i0);
    // largest b with off_out[b] <= i0: then off_out[b+1] > i0 and the bucket is non-empty
    uint32_t b;
    {
        uint32_t lo = 0, hi = nb;
        while (hi - lo > 1) {
            const uint32_t mid = (lo + hi) >> 1;
            if (off_out[mid] <= i0) lo = mid; else hi = mid;
        }
        b = lo;
    }
    uint32_t ob = off_out[b], oe = off_out[b + 1], ib = off_in[b], ic = off_in[b + 1] - ib;
    // ---- pass A: running products of the denominators; the product before item k waits in item k's output slot
    using D = typename BaTypes<F>::D;
    using PR = typename BaTypes<F>::P;
    PR acc;
    for (uint32_t k = 0; k < cnt; k++) {
        const uint32_t i = (uint32_t)i0 + k;
        if (i >= oe) {
            do { b++; ob = oe; oe = off_out[b + 1]; } while (i >= oe);
            ib = off_in[b];
            ic = off_in[b + 1] - ib;
        }
        const uint32_t j = i - ob;
        const BaItem<F, FIRST> it = ba_item<F, FIRST>(table, refs, src, ib + 2 * j, 2 * j + 1 < ic);
        const T x1 = ba_coord<F, FIRST>(it.p1, 0, false);
        T x2 = x1;
        if (it.pair) x2 = ba_coord<F, FIRST>(it.p2, 0, false);
        D d;
        (void)ba_classify<F, FIRST>(it, x1, x2, d);
        if (k == 0) acc = PR(d);
        else {
            ba_store_coord(dst + (uint64_t)PB * i, acc);
            acc = PR(fe_mul(acc, d));
        }
    }
    // ---- one inversion for the whole batch
    T inv_run = T(fe_inv_fast(acc));
    // ---- pass B, backward
    for (uint32_t k = cnt; k-- > 0;) {
        const uint32_t i = (uint32_t)i0 + k;
        if (i < ob) {
            do { b--; oe = ob; ob = off_out[b]; } while (i < ob);
            ib = off_in[b];
            ic = off_in[b + 1] - ib;
        }
        const uint32_t j = i - ob;
        const BaItem<F, FIRST> it = ba_item<F, FIRST>(table, refs, src, ib + 2 * j, 2 * j + 1 < ic);
        const T x1 = ba_coord<F, FIRST>(it.p1, 0, false);
        const T y1 = ba_coord<F, FIRST>(it.p1, 1, it.neg1);
        T x2 = x1, y2 = y1;
        if (it.pair) {
            x2 = ba_coord<F, FIRST>(it.p2, 0, false);
            y2 = ba_coord<F, FIRST>(it.p2, 1, it.neg2);
        }
        D d;
        const int kind = ba_classify<F, FIRST>(it, x1, x2, d);
        uint8_t* out = dst + (uint64_t)PB * i;
        if (kind == BA_ADD || kind == BA_DBL) {
            T inv = inv_run;
            if (k > 0) {
                const PR pm = ba_load_prod((const F*)nullptr, out);        // product of the denominators before this item
                inv = T(fe_mul(inv_run, pm));
                inv_run = T(fe_mul(inv_run, d));
            }
            D num = fe_sub(y2, y1);
            if (kind == BA_DBL) {                                   // rare: lambda = 3 x1^2 / (2 y1)
                const auto xx = fe_sqr(x1);
                num = D(fe_add(fe_dbl(xx), xx));
            }
            const T lam = T(fe_mul(num, inv));
            const T x3 = ba_nonzero_rep(fp_reduce_small(fe_sub(fe_sqr(lam), fe_add(x1, x2))));
            const T y3 = fp_reduce_small(fe_sub(fe_mul(lam, fe_sub(x1, x3)), y1));
            ba_store_coord(out, x3);
            ba_store_coord(out + CB, y3);
        } else if (kind == BA_INF) {
            ba_store_coord(out, BaTypes<F>::zero());
            ba_store_coord(out + CB, BaTypes<F>::zero());
        } else {
            const bool second = kind == BA_COPY2;
            T cx = second ? x2 : x1;
            if constexpr (FIRST) cx = ba_nonzero_rep(cx);        // a table point with x = 0 (outside the r-torsion) is still a point
            ba_store_coord(out, cx);
            ba_store_coord(out + CB, second ? y2 : y1);
        }
    }
}

}  // namespace zk
