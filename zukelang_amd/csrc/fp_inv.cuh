// Lockstep modular inversion in Fp for gfx950: every lane of a wave inverts its own value and all of them execute the
// SAME instruction sequence (no data-dependent loop, no divergent branch).
//
// The binary extended Euclid of ff.cuh (words_inv) is fine on a single-lane tail, but with 64 lanes inverting at once its
// inner `while (even)` loops cost the maximum over the lanes in every round (~650 product-equivalents per wave).  The
// batch-affine bucket accumulation (msm_ba.cuh) inverts once per lane per batch, so the inversion has to be cheap IN
// LOCKSTEP.  This is Pornin's "Optimized Binary GCD for Modular Inversion" (2020) re-cut for the 14 x 29-bit limbs of
// ff.cuh: 27 outer iterations, each
//   1 takes 64-bit approximations of a and b (their low 29 bits + the 35 bits below the top bit of max(a, b); exact once
//     both fit 64 bits),
//   2 runs 29 branch-free binary-GCD steps on the approximations, collecting update factors |f|, |g| <= 2^29,
//   3 applies the factors to the 14-limb values (a, b) <- ((f0 a + g0 b) / 2^29, (f1 a + g1 b) / 2^29)   [exact division]
//     and to (u, v) <- ((f0 u + g0 v + q p) / 2^29, ...)                                              [division modulo p]
// with the invariant a = u x, b = v x (mod p).  27 * 29 = 783 >= 2 * 381 - 1 steps end with a = 0, b = 1, v = x^-1.
// u, v are SIGNED 14-limb integers (top limb signed) and grow by at most p per iteration (|f| + |g| <= 2^29): |v| < 28 p.
// ~25 k instructions per wave (~45 field products' worth) whatever the operands are; scripts/proto/fp_inv_model.py is the
// integer model that checks the invariants, the approximation rule and the bounds.
//
// Serves the same purpose as Bls12_381.Fq inversion inside the reference's external library (opam bls12-381; every
// affine addition / to_affine of src/lib/zk/curve.ml:159-191 ends in one).
#pragma once
#include "ff.cuh"

namespace zk {

static constexpr int FP_INV_ITERS = 27;

// x: exact 29-bit limbs of a canonical value X < p.  Returns limbs (weakly normalised, value in (4p, 60p)) of X^-1 mod p;
// X = 0 gives 0.  No Montgomery correction: the caller multiplies by R^3 (fe_inv_fast).
__device__ __noinline__ static FpRaw fp_inv29_call(FP_ARGS(x)) {
    FP_UNPACK_ARGS(a, x)
    uint32_t b[FPL];
    int32_t u[FPL], v[FPL];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) { b[i] = FP29_MOD[i]; u[i] = 0; v[i] = 0; any |= a[i]; }
    u[0] = 1;
#pragma unroll 1
    for (int it = 0; it < FP_INV_ITERS; it++) {
        // ---- 1: approximations.  (hi, mid, lo) = the limbs j, j-1, j-2 at the highest j >= 2 where a_j | b_j != 0
        uint32_t ha = a[2], ma = a[1], la = a[0], hb = b[2], mb = b[1], lb = b[0];
        bool small = true;
#pragma unroll
        for (int j = 3; j < FPL; j++) {
            const bool nz = (a[j] | b[j]) != 0;
            ha = nz ? a[j] : ha; ma = nz ? a[j - 1] : ma; la = nz ? a[j - 2] : la;
            hb = nz ? b[j] : hb; mb = nz ? b[j - 1] : mb; lb = nz ? b[j - 2] : lb;
            small = small && !nz;
        }
        const uint32_t top = ha | hb;
        const uint32_t ell = 32u - (uint32_t)__builtin_clz(top | 1u) - (top ? 0u : 1u);        // bit length of the leading limb (0 if it is zero)
        const bool exact = small && ell <= 6;                                                  // both values below 2^64
        uint64_t xa, xb;
        {
            const uint64_t ta = ((uint64_t)ha << 35) | ((uint64_t)ma << 6) | (uint64_t)(la >> 23);
            const uint64_t tb = ((uint64_t)hb << 35) | ((uint64_t)mb << 6) | (uint64_t)(lb >> 23);
            const uint64_t ea = (uint64_t)a[0] | ((uint64_t)a[1] << 29) | ((uint64_t)a[2] << 58);
            const uint64_t eb = (uint64_t)b[0] | ((uint64_t)b[1] << 29) | ((uint64_t)b[2] << 58);
            const uint64_t aa = ((ta >> ell) << 29) | (uint64_t)a[0];
            const uint64_t ab = ((tb >> ell) << 29) | (uint64_t)b[0];
            xa = exact ? ea : aa;
            xb = exact ? eb : ab;
        }
        // ---- 2: 29 binary-GCD steps on the approximations
        int32_t f0 = 1, g0 = 0, f1 = 0, g1 = 1;
#pragma unroll
        for (int s = 0; s < FP29_W; s++) {
            const bool odd = (xa & 1u) != 0;
            const bool sw = odd && xa < xb;
            const uint64_t na = sw ? xb : xa, nb = sw ? xa : xb;
            const int32_t nf0 = sw ? f1 : f0, ng0 = sw ? g1 : g0, nf1 = sw ? f0 : f1, ng1 = sw ? g0 : g1;
            xa = (na - (odd ? nb : 0)) >> 1;
            xb = nb;
            f0 = nf0 - (odd ? nf1 : 0);
            g0 = ng0 - (odd ? ng1 : 0);
            f1 = nf1 << 1;
            g1 = ng1 << 1;
        }
        // ---- 3a: (a, b) <- (f0 a + g0 b, f1 a + g1 b) / 2^29, made non-negative (the sign moves into the factors)
        {
            int64_t ca = 0, cb = 0;
            uint32_t ra[FPL], rb[FPL];
#pragma unroll
            for (int i = 0; i < FPL; i++) {
                ca += (int64_t)f0 * (int64_t)(int32_t)a[i] + (int64_t)g0 * (int64_t)(int32_t)b[i];
                cb += (int64_t)f1 * (int64_t)(int32_t)a[i] + (int64_t)g1 * (int64_t)(int32_t)b[i];
                if (i > 0) { ra[i - 1] = (uint32_t)ca & FP29_MASK; rb[i - 1] = (uint32_t)cb & FP29_MASK; }
                ca >>= FP29_W;
                cb >>= FP29_W;
            }
            ra[FPL - 1] = (uint32_t)ca;               // top limb: 0 or the sign extension
            rb[FPL - 1] = (uint32_t)cb;
            const uint32_t sa = (uint32_t)(ca >> 63), sb = (uint32_t)(cb >> 63);       // all-ones if negative
            uint32_t cya = sa & 1u, cyb = sb & 1u;
#pragma unroll
            for (int i = 0; i < FPL - 1; i++) {
                const uint32_t ta = (ra[i] ^ (sa & FP29_MASK)) + cya, tb = (rb[i] ^ (sb & FP29_MASK)) + cyb;
                a[i] = ta & FP29_MASK; cya = ta >> FP29_W;
                b[i] = tb & FP29_MASK; cyb = tb >> FP29_W;
            }
            a[FPL - 1] = (ra[FPL - 1] ^ sa) + cya;
            b[FPL - 1] = (rb[FPL - 1] ^ sb) + cyb;
            f0 = (int32_t)(((uint32_t)f0 ^ sa) - sa); g0 = (int32_t)(((uint32_t)g0 ^ sa) - sa);
            f1 = (int32_t)(((uint32_t)f1 ^ sb) - sb); g1 = (int32_t)(((uint32_t)g1 ^ sb) - sb);
        }
        // ---- 3b: (u, v) <- (f0 u + g0 v + q p, f1 u + g1 v + q' p) / 2^29, signed limbs
        {
            const uint32_t lu = (uint32_t)f0 * (uint32_t)u[0] + (uint32_t)g0 * (uint32_t)v[0];
            const uint32_t lv = (uint32_t)f1 * (uint32_t)u[0] + (uint32_t)g1 * (uint32_t)v[0];
            const uint32_t qu = (lu * FP29_NINV) & FP29_MASK, qv = (lv * FP29_NINV) & FP29_MASK;
            int64_t cu = 0, cv = 0;
            int32_t nu[FPL], nv[FPL];
#pragma unroll
            for (int i = 0; i < FPL; i++) {
                cu += (int64_t)f0 * (int64_t)u[i] + (int64_t)g0 * (int64_t)v[i] + (int64_t)((uint64_t)qu * FP29_MOD[i]);
                cv += (int64_t)f1 * (int64_t)u[i] + (int64_t)g1 * (int64_t)v[i] + (int64_t)((uint64_t)qv * FP29_MOD[i]);
                if (i > 0) { nu[i - 1] = (int32_t)((uint32_t)cu & FP29_MASK); nv[i - 1] = (int32_t)((uint32_t)cv & FP29_MASK); }
                cu >>= FP29_W;
                cv >>= FP29_W;
            }
            nu[FPL - 1] = (int32_t)cu;
            nv[FPL - 1] = (int32_t)cv;
#pragma unroll
            for (int i = 0; i < FPL; i++) { u[i] = nu[i]; v[i] = nv[i]; }
        }
    }
    // v in (-28p, 28p) -> v + 32p in (4p, 60p), weakly normalised
    constexpr int KI = fp_ki(32);
    FpRaw r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = any ? (uint32_t)v[i] + FP29_KP[KI][i] : 0u;
    fp_carry(r.v);
    return r;
}

// (a R)^-1 R^3 / R = a^-1 R; inv(0) = 0.  Same contract as fe_inv (ff.cuh), lockstep cost.
template <int A> FF_INLINE FpB<2> fe_inv_fast(const FpB<A>& a) {
    const FpB<1> c = fp_canon(a);
    const FpB<64> r = fp_from_raw<64>(fp_inv29_call(FP_PASS(c.v)));
    FpB<1> r3;
#pragma unroll
    for (int i = 0; i < FPL; i++) r3.v[i] = FP29_R3[i];
    return fe_mul(r, r3);
}
// the library's fe_inv for Fp and for a whole Fp2 on one lane (to-affine conversions, key-time window tables): the same
// inversion -- a single lane pays ~25 k instructions either way, 64 lanes at once no longer pay the slowest lane's loop counts
template <int A> FF_INLINE FpB<2> fe_inv(const FpB<A>& a) { return fe_inv_fast(a); }
template <int A> FF_INLINE Fp2B<4> fe_inv(const Fp2B<A>& a) {
    const FpB<2> d = fe_inv_fast(fe_add(fe_sqr(a.c0), fe_sqr(a.c1)));
    return {fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}
// lane-pair Fp2: 1 / (a0 + a1 u) = (a0 - a1 u) / (a0^2 + a1^2); both lanes invert the (shared) norm
template <int A> FF_INLINE Fp2HB<4> fe_inv_fast(const Fp2HB<A>& a) {
    const FpB<2> s = fe_sqr(a.v);
    const FpB<4> norm = fe_add(s, pair_swap(s));
    const FpB<2> ni = fe_inv_fast(norm);
    const FpB<2> m = fe_mul(a.v, ni);
    const FpB<4> neg = fe_neg(m);
    FpB<4> r;
    const bool c1 = pair_comp() != 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = c1 ? neg.v[i] : m.v[i];
    return {r};
}

template <int A> FF_INLINE Fp2HB<4> fe_inv(const Fp2HB<A>& a) { return fe_inv_fast(a); }

}  // namespace zk
