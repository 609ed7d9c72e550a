// Scalar-field arithmetic of the NTT kernels: Fr on 9 limbs of 29 bits, Montgomery R' = 2^261, carry-free product
// columns (18 partial products < 2^58 per column) and lazy reduction, as for Fp (ff.cuh).  The NTT passes are VALU-bound
// (PMC: the SIMDs issue vector instructions all the time, ~545 per butterfly with the 8 x 32-bit multiplier), and this
// multiplier measures 170 G products/s against 128.
//
// Memory keeps the 8-word format of Fe<FrParams> (values x R mod r with R = 2^256, fully reduced); a kernel unpacks at its
// tile load, works on lazily reduced 9-limb integers, and reduces + packs at its tile store.  Factors (twiddles, pointwise
// tables, scale constants) are stored as f R' mod r = 32 * (f R): mul'(x R, f R') = x f R stays in the data's form.
//
// Bounds (value < B r, limbs 0..7 <= 2^29 + 7): the product needs A * B <= 64 (2^261 / r = 70) and returns < 2r;
// a - b adds K r with K >= B_b + 1 a power of two (spread form: no limb borrows, the value stays >= r so the top limb
// is exact after the carry pass).  The NTT stages keep B <= 56 by construction (ntt_lds.cuh).
#pragma once
#include "ff.cuh"
#include "fr29_consts.cuh"

namespace zk {

struct Fr9 {
    uint32_t v[FR29_L];
};
FF_INLINE Fr9 fr9_unpack(const Fr& a) {               // any 256-bit value; 8 words -> 9 limbs
    Fr9 r;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) {
        const int bit = FR29_W * i, k = bit / 32, s = bit % 32;
        uint64_t x = a.v[k] >> s;
        if (k + 1 < 8) x |= (uint64_t)a.v[k + 1] << (32 - s);
        r.v[i] = i < FR29_L - 1 ? ((uint32_t)x & FR29_MASK) : (uint32_t)x;
    }
    return r;
}
FF_INLINE Fr fr9_pack_exact(const Fr9& a) {           // exact limbs, value < 2^256
    Fr r;
#pragma unroll
    for (int j = 0; j < 8; j++) {
        const int bit = 32 * j, k = bit / FR29_W, s = bit % FR29_W;
        uint64_t x = a.v[k] >> s;
        if (k + 1 < FR29_L) x |= (uint64_t)a.v[k + 1] << (FR29_W - s);
        if (2 * FR29_W - s < 32 && k + 2 < FR29_L) x |= (uint64_t)a.v[k + 2] << (2 * FR29_W - s);
        r.v[j] = (uint32_t)x;
    }
    return r;
}
FF_INLINE void fr9_carry(uint32_t* t) {
    uint32_t c[FR29_L - 1];
#pragma unroll
    for (int i = 0; i < FR29_L - 1; i++) c[i] = t[i] >> FR29_W;
    t[0] &= FR29_MASK;
#pragma unroll
    for (int i = 1; i < FR29_L - 1; i++) t[i] = (t[i] & FR29_MASK) + c[i - 1];
    t[FR29_L - 1] += c[FR29_L - 2];
}
FF_INLINE Fr9 fr9_add(const Fr9& a, const Fr9& b) {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) r.v[i] = a.v[i] + b.v[i];
    fr9_carry(r.v);
    return r;
}
// a - b + 2^KI r, 2^KI >= (bound of b) + 1
template <int KI> FF_INLINE Fr9 fr9_sub(const Fr9& a, const Fr9& b) {
    static_assert(KI >= 0 && KI < FR29_NK, "no such multiple of r");
    Fr9 r;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) r.v[i] = a.v[i] + (FR29_KR[KI][i] - b.v[i]);
    fr9_carry(r.v);
    return r;
}
// The same without the carry pass: limbs up to 2^31, only good as the FIRST operand of a product whose second operand
// has exact 29-bit limbs (a factor read from memory): 9 * 2^31 * 2^29 + 9 * 2^58 < 2^64.
template <int KI> FF_INLINE Fr9 fr9_sub_raw(const Fr9& a, const Fr9& b) {
    static_assert(KI >= 0 && KI < FR29_NK, "no such multiple of r");
    Fr9 r;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) r.v[i] = a.v[i] + (FR29_KR[KI][i] - b.v[i]);
    return r;
}
// (a b + m r) / 2^261 < 2r for bounds A * B <= 64.  r = 1 mod 2^29: m_k = -acc mod 2^29 and m_k * r_0 = m_k.
FF_INLINE Fr9 fr9_mul(const Fr9& a, const Fr9& b) {
    uint64_t acc = 0;
    uint32_t m[FR29_L];
    Fr9 r;
#pragma unroll
    for (int k = 0; k < FR29_L; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FR29_MOD[k - i];
        m[k] = (0u - (uint32_t)acc) & FR29_MASK;
        acc += m[k];
        acc >>= FR29_W;
    }
#pragma unroll
    for (int k = FR29_L; k < 2 * FR29_L - 1; k++) {
#pragma unroll
        for (int i = k - FR29_L + 1; i < FR29_L; i++) acc += (uint64_t)a.v[i] * b.v[k - i];
#pragma unroll
        for (int i = k - FR29_L + 1; i < FR29_L; i++) acc += (uint64_t)m[i] * FR29_MOD[k - i];
        r.v[k - FR29_L] = (uint32_t)acc & FR29_MASK;
        acc >>= FR29_W;
    }
    r.v[FR29_L - 1] = (uint32_t)acc;
    return r;
}
// value < 64 r -> the same value mod r, below 2r, exact limbs.  q = floor(top / ((r >> 232) + 1)) never exceeds
// the true quotient and falls short of it by at most one (scripts/proto/fr29_model.py checks the extremes).
FF_INLINE Fr9 fr9_reduce_weak(const Fr9& a) {
    uint32_t t[FR29_L];
#pragma unroll
    for (int i = 0; i < FR29_L; i++) t[i] = a.v[i];
#pragma unroll
    for (int i = 0; i < FR29_L - 1; i++) {            // exact limbs
        t[i + 1] += t[i] >> FR29_W;
        t[i] &= FR29_MASK;
    }
    const uint32_t q = (uint32_t)(((uint64_t)t[FR29_L - 1] * FR29_QMUL) >> 50);
    Fr9 r;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) {
        const int64_t cur = (int64_t)t[i] - (int64_t)((uint64_t)q * FR29_MOD[i]) + cy;
        r.v[i] = i < FR29_L - 1 ? ((uint32_t)cur & FR29_MASK) : (uint32_t)cur;
        cy = cur >> FR29_W;
    }
    return r;
}
// full reduction to [0, r) and the 8-word memory format
FF_INLINE Fr fr9_canon_pack(const Fr9& a) {
    Fr9 t = fr9_reduce_weak(a);                        // < 2r: one conditional subtraction
    {
        uint32_t u[FR29_L];
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < FR29_L; i++) {
            const int32_t d = (int32_t)t.v[i] - (int32_t)FR29_MOD[i] - bw;
            bw = (d >> 31) & 1;
            u[i] = i < FR29_L - 1 ? ((uint32_t)d & FR29_MASK) : (uint32_t)d;
        }
        if (!bw) {
#pragma unroll
            for (int i = 0; i < FR29_L; i++) t.v[i] = u[i];
        }
    }
    return fr9_pack_exact(t);
}
FF_INLINE Fr9 fr9_load(const uint32_t* p) { return fr9_unpack(fe_load<FrParams>(p)); }
FF_INLINE void fr9_store(uint32_t* p, const Fr9& a) { fe_store<FrParams>(p, fr9_canon_pack(a)); }
FF_INLINE Fr9 fr9_zero() {
    Fr9 r;
#pragma unroll
    for (int i = 0; i < FR29_L; i++) r.v[i] = 0;
    return r;
}
template <class LDS> FF_INLINE Fr9 fr9_lds_get(LDS& lds, uint32_t e) {
    Fr9 r;
#pragma unroll
    for (int l = 0; l < FR29_L; l++) r.v[l] = lds[l][e];
    return r;
}
template <class LDS> FF_INLINE void fr9_lds_put(LDS& lds, uint32_t e, const Fr9& a) {
#pragma unroll
    for (int l = 0; l < FR29_L; l++) lds[l][e] = a.v[l];
}

}  // namespace zk
