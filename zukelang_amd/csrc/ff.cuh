// BLS12-381 field arithmetic for gfx950 (CDNA4): Montgomery form on 32-bit limbs held in VGPRs.
//
// CDNA4 has no 64x64 multiplier; the widest integer multiply-add is v_mad_u64_u32
// (32x32 + 64 -> 64), so elements are N x u32 (Fr: N = 8, R = 2^256; Fp: N = 12, R = 2^384)
// and every product row is a chain of v_mad_u64_u32 with the running 32-bit carry folded into
// the 64-bit addend.  Everything is fully unrolled so limbs never leave registers.
//
// Replaces, for the prove path, the Fr / Fq arithmetic the reference obtains from opam
// bls12-381 (src/lib/zk/curve.ml:121-140 Fr ops; G1/G2 coordinates curve.ml:159-191).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FF_INLINE __device__ __forceinline__

#include "ff_mul_gen.cuh"

namespace zk {

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
};
struct FpParams {
    static constexpr int N = 12;
    static constexpr uint32_t INV = 0xfffcfffdu;  // -p^-1 mod 2^32
};

// Constants live in const device arrays; with full unrolling clang folds them to literals.
__device__ static const uint32_t FR_MOD[8] = {0x00000001u, 0xffffffffu, 0xfffe5bfeu, 0x53bda402u,
                                              0x09a1d805u, 0x3339d808u, 0x299d7d48u, 0x73eda753u};
__device__ static const uint32_t FR_R1[8] = {0xfffffffeu, 0x00000001u, 0x00034802u, 0x5884b7fau,
                                             0xecbc4ff5u, 0x998c4fefu, 0xacc5056fu, 0x1824b159u};
__device__ static const uint32_t FR_R2[8] = {0xf3f29c6du, 0xc999e990u, 0x87925c23u, 0x2b6cedcbu,
                                             0x7254398fu, 0x05d31496u, 0x9f59ff11u, 0x0748d9d9u};
__device__ static const uint32_t FP_MOD[12] = {0xffffaaabu, 0xb9feffffu, 0xb153ffffu, 0x1eabfffeu,
                                               0xf6b0f624u, 0x6730d2a0u, 0xf38512bfu, 0x64774b84u,
                                               0x434bacd7u, 0x4b1ba7b6u, 0x397fe69au, 0x1a0111eau};
__device__ static const uint32_t FP_R1[12] = {0x0002fffdu, 0x76090000u, 0xc40c0002u, 0xebf4000bu,
                                              0x53c758bau, 0x5f489857u, 0x70525745u, 0x77ce5853u,
                                              0xa256ec6du, 0x5c071a97u, 0xfa80e493u, 0x15f65ec3u};
__device__ static const uint32_t FP_R2[12] = {0x1c341746u, 0xf4df1f34u, 0x09d104f1u, 0x0a76e6a6u,
                                              0x4c95b6d5u, 0x8de5476cu, 0x939d83c0u, 0x67eb88a9u,
                                              0xb519952du, 0x9a793e85u, 0x92cae3aau, 0x11988fe5u};

__device__ static const uint32_t FR_R3[8] = {0x439b73afu, 0xc62c1807u, 0x8cf06990u, 0x1b3e0d18u,
                                             0xc7b5f418u, 0x73d13c71u, 0xc8db33e9u, 0x6e2a5bb9u};
__device__ static const uint32_t FP_R3[12] = {0xd94ca1e0u, 0xed48ac6bu, 0x03a7adf8u, 0x315f831eu,
                                              0x615e29ddu, 0x9a53352au, 0x921e1761u, 0x34c04e5eu,
                                              0x65724728u, 0x2512d435u, 0x91755d4du, 0x0aa63460u};

template <class P> struct Consts;
template <> struct Consts<FrParams> {
    static FF_INLINE uint32_t mod(int i) { return FR_MOD[i]; }
    static FF_INLINE uint32_t r1(int i) { return FR_R1[i]; }
    static FF_INLINE uint32_t r2(int i) { return FR_R2[i]; }
    static FF_INLINE uint32_t r3(int i) { return FR_R3[i]; }
};
template <> struct Consts<FpParams> {
    static FF_INLINE uint32_t mod(int i) { return FP_MOD[i]; }
    static FF_INLINE uint32_t r1(int i) { return FP_R1[i]; }
    static FF_INLINE uint32_t r2(int i) { return FP_R2[i]; }
    static FF_INLINE uint32_t r3(int i) { return FP_R3[i]; }
};

template <class P> struct Fe {
    static constexpr int N = P::N;
    uint32_t v[P::N];
};
using Fr = Fe<FrParams>;
using Fp = Fe<FpParams>;

template <class P> FF_INLINE Fe<P> fe_zero() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = 0;
    return r;
}
template <class P> FF_INLINE Fe<P> fe_one() {
    Fe<P> r;
#pragma unroll
    for (int i = 0; i < P::N; i++) r.v[i] = Consts<P>::r1(i);
    return r;
}
template <class P> FF_INLINE bool fe_is_zero(const Fe<P>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.v[i];
    return o == 0;
}
template <class P> FF_INLINE bool fe_eq(const Fe<P>& a, const Fe<P>& b) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) o |= a.v[i] ^ b.v[i];
    return o == 0;
}

// r = a - mod if a >= mod (a < 2*mod)
template <class P> FF_INLINE void fe_cond_sub(Fe<P>& a) {
    uint32_t t[P::N];
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t d = (uint64_t)a.v[i] - Consts<P>::mod(i) - bw;
        t[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    if (!bw) {
#pragma unroll
        for (int i = 0; i < P::N; i++) a.v[i] = t[i];
    }
}
template <class P> FF_INLINE Fe<P> fe_add(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)a.v[i] + b.v[i];
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    // moduli are < 2^(32N-1): no carry out of the top limb
    fe_cond_sub(r);
    return r;
}
template <class P> FF_INLINE Fe<P> fe_sub(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t d = (uint64_t)a.v[i] - b.v[i] - bw;
        r.v[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    uint32_t mask = (uint32_t)0 - (uint32_t)bw;
    uint64_t c = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        c += (uint64_t)r.v[i] + (Consts<P>::mod(i) & mask);
        r.v[i] = (uint32_t)c;
        c >>= 32;
    }
    return r;
}
template <class P> FF_INLINE Fe<P> fe_neg(const Fe<P>& a) {
    if (fe_is_zero(a)) return a;
    Fe<P> r;
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t d = (uint64_t)Consts<P>::mod(i) - a.v[i] - bw;
        r.v[i] = (uint32_t)d;
        bw = (d >> 32) & 1;
    }
    return r;
}
template <class P> FF_INLINE Fe<P> fe_dbl(const Fe<P>& a) { return fe_add(a, a); }

// Montgomery product a*b*R^-1 mod p, fully reduced.  The straight-line body is generated
// (scripts/gen_mont_mul.py): finely integrated product scanning, 2N^2 + N partial products at two
// instructions each (v_mad_u64_u32 + v_addc_co_u32), one asm statement per column.
template <class P> FF_INLINE Fe<P> fe_mul_inline(const Fe<P>& a, const Fe<P>& b) {
    Fe<P> r;
    if constexpr (P::N == 8) mont_mul_fr(r.v, a.v, b.v);
    else mont_mul_fp(r.v, a.v, b.v);
    // the product is < 2p < 2^(32N): one conditional subtraction
    fe_cond_sub(r);
    return r;
}
// The 12-limb product is ~0.8k instructions: as a real function (operands and result travel in
// VGPRs under the AMDGPU calling convention) every kernel shares one copy -- the hot loops stay
// inside the instruction cache and the library compiles in minutes instead of an hour.
__device__ __noinline__ static Fe<FpParams> fp_mul_call(Fe<FpParams> a, Fe<FpParams> b) { return fe_mul_inline(a, b); }
template <class P> FF_INLINE Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) {
    if constexpr (P::N == 12) return fp_mul_call(a, b);
    else return fe_mul_inline(a, b);
}
template <class P> FF_INLINE Fe<P> fe_sqr(const Fe<P>& a) { return fe_mul(a, a); }

template <class P> FF_INLINE Fe<P> fe_to_mont(const Fe<P>& a) {
    Fe<P> r2;
#pragma unroll
    for (int i = 0; i < P::N; i++) r2.v[i] = Consts<P>::r2(i);
    return fe_mul(a, r2);
}
template <class P> FF_INLINE Fe<P> fe_from_mont(const Fe<P>& a) {
    Fe<P> one = fe_zero<P>();
    one.v[0] = 1;
    return fe_mul(a, one);
}
// canonical a < mod ?
template <class P> FF_INLINE bool fe_is_canonical(const Fe<P>& a) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t d = (uint64_t)a.v[i] - Consts<P>::mod(i) - bw;
        bw = (d >> 32) & 1;
    }
    return bw != 0;
}
// Modular inverse by the binary extended Euclid on plain limbs (shifts and subtractions only):
// ~2 * bits iterations of a few N-limb operations, an order of magnitude cheaper than Fermat's
// a^(p-2) (~580 Montgomery products) -- it matters because conversions to affine sit on
// single-lane tails.  Input and output in Montgomery form: (aR)^-1 * R^3 * R^-1 = a^-1 R.  inv(0) = 0.
template <class P> __device__ __noinline__ Fe<P> fe_inv(const Fe<P>& a) {
    constexpr int N = P::N;
    if (fe_is_zero(a)) return a;
    uint32_t u[N], v[N], x1[N], x2[N];
#pragma unroll
    for (int i = 0; i < N; i++) { u[i] = a.v[i]; v[i] = Consts<P>::mod(i); x1[i] = 0; x2[i] = 0; }
    x1[0] = 1;
    auto is_one = [](const uint32_t* x) {
        uint32_t o = x[0] ^ 1u;
#pragma unroll
        for (int i = 1; i < N; i++) o |= x[i];
        return o == 0;
    };
    auto shr1 = [](uint32_t* x) {
#pragma unroll
        for (int i = 0; i < N - 1; i++) x[i] = (x[i] >> 1) | (x[i + 1] << 31);
        x[N - 1] >>= 1;
    };
    auto halve_mod = [&](uint32_t* x) {          // x/2 mod p
        if (x[0] & 1) {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < N; i++) { c += (uint64_t)x[i] + Consts<P>::mod(i); x[i] = (uint32_t)c; c >>= 32; }
        }
        shr1(x);
    };
    auto geq = [](const uint32_t* x, const uint32_t* y) {
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)x[i] - y[i] - bw; bw = (d >> 32) & 1; }
        return bw == 0;
    };
    auto sub = [](uint32_t* x, const uint32_t* y) {      // x -= y (x >= y)
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)x[i] - y[i] - bw; x[i] = (uint32_t)d; bw = (d >> 32) & 1; }
    };
    auto sub_mod = [&](uint32_t* x, const uint32_t* y) {  // x = x - y mod p
        uint64_t bw = 0;
#pragma unroll
        for (int i = 0; i < N; i++) { uint64_t d = (uint64_t)x[i] - y[i] - bw; x[i] = (uint32_t)d; bw = (d >> 32) & 1; }
        if (bw) {
            uint64_t c = 0;
#pragma unroll
            for (int i = 0; i < N; i++) { c += (uint64_t)x[i] + Consts<P>::mod(i); x[i] = (uint32_t)c; c >>= 32; }
        }
    };
    for (int guard = 0; guard < 4 * 32 * N && !is_one(u) && !is_one(v); guard++) {
        while (!(u[0] & 1)) { shr1(u); halve_mod(x1); }
        while (!(v[0] & 1)) { shr1(v); halve_mod(x2); }
        if (geq(u, v)) { sub(u, v); sub_mod(x1, x2); }
        else { sub(v, u); sub_mod(x2, x1); }
    }
    Fe<P> r, r3;
    const bool use1 = is_one(u);
#pragma unroll
    for (int i = 0; i < N; i++) { r.v[i] = use1 ? x1[i] : x2[i]; r3.v[i] = Consts<P>::r3(i); }
    return fe_mul(r, r3);
}
template <class P> FF_INLINE Fe<P> fe_from_u32(uint32_t x) {
    Fe<P> r = fe_zero<P>();
    r.v[0] = x;
    return fe_to_mont(r);
}

// 128-bit vector loads / stores of whole elements (32 B Fr = 2 x dwordx4, 48 B Fp = 3 x dwordx4).
template <class P> FF_INLINE Fe<P> fe_load(const void* p) {
    Fe<P> r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) {
        uint4 x = q[i];
        r.v[4 * i] = x.x; r.v[4 * i + 1] = x.y; r.v[4 * i + 2] = x.z; r.v[4 * i + 3] = x.w;
    }
    return r;
}
template <class P> FF_INLINE void fe_store(void* p, const Fe<P>& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < P::N / 4; i++) q[i] = make_uint4(a.v[4 * i], a.v[4 * i + 1], a.v[4 * i + 2], a.v[4 * i + 3]);
}

// ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2 + 1)
struct Fp2 {
    Fp c0, c1;
};
FF_INLINE Fp2 fp2_zero() { return {fe_zero<FpParams>(), fe_zero<FpParams>()}; }
FF_INLINE Fp2 fp2_one() { return {fe_one<FpParams>(), fe_zero<FpParams>()}; }
FF_INLINE bool fe_is_zero(const Fp2& a) { return fe_is_zero(a.c0) && fe_is_zero(a.c1); }
FF_INLINE bool fe_eq(const Fp2& a, const Fp2& b) { return fe_eq(a.c0, b.c0) && fe_eq(a.c1, b.c1); }
FF_INLINE Fp2 fe_add(const Fp2& a, const Fp2& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
FF_INLINE Fp2 fe_sub(const Fp2& a, const Fp2& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
FF_INLINE Fp2 fe_neg(const Fp2& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
FF_INLINE Fp2 fe_dbl(const Fp2& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
// Karatsuba: 3 base multiplications
FF_INLINE Fp2 fe_mul(const Fp2& a, const Fp2& b) {
    Fp t0 = fe_mul(a.c0, b.c0);
    Fp t1 = fe_mul(a.c1, b.c1);
    Fp s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return {fe_sub(t0, t1), fe_sub(fe_sub(s, t0), t1)};
}
// (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u : 2 base multiplications
FF_INLINE Fp2 fe_sqr(const Fp2& a) {
    Fp p = fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1));
    Fp q = fe_mul(a.c0, a.c1);
    return {p, fe_dbl(q)};
}
__device__ __noinline__ inline Fp2 fe_inv(const Fp2& a) {
    Fp n = fe_add(fe_sqr(a.c0), fe_sqr(a.c1));
    Fp d = fe_inv(n);
    return {fe_mul(a.c0, d), fe_neg(fe_mul(a.c1, d))};
}

// ------------------------------------------------------------------ Fp2 split over a lane pair
// Lane 2k holds the c0 component, lane 2k+1 the c1 component of the same Fp2 value; partners trade
// operands with one DPP quad_perm(1,0,3,2) move per limb.  A product costs each lane two base
// multiplications (a0 b0, a1 b1 | a0 b1, a1 b0) instead of Karatsuba's three on one lane: 4/3 of the
// multiplier work, but every lane carries HALF of each value, so a G2 mixed addition has the register
// footprint of a G1 one (2 waves per SIMD, the accumulator never leaves the VGPRs).  Both lanes of a
// pair must follow the same control flow; every predicate below is pair-uniform by construction.
struct Fp2H {
    Fp v;
};
FF_INLINE uint32_t pair_comp() { return threadIdx.x & 1u; }
FF_INLINE Fp pair_swap(const Fp& a) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xF, 0xF, true);
    return r;
}
FF_INLINE Fp fp_select(bool take_b, const Fp& a, const Fp& b) {
    Fp r;
#pragma unroll
    for (int i = 0; i < 12; i++) r.v[i] = take_b ? b.v[i] : a.v[i];
    return r;
}
FF_INLINE bool fe_is_zero(const Fp2H& a) {
    const int z = fe_is_zero(a.v) ? 1 : 0;
    return z && __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true);
}
FF_INLINE bool fe_eq(const Fp2H& a, const Fp2H& b) {
    const int z = fe_eq(a.v, b.v) ? 1 : 0;
    return z && __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true);
}
FF_INLINE Fp2H fe_add(const Fp2H& a, const Fp2H& b) { return {fe_add(a.v, b.v)}; }
FF_INLINE Fp2H fe_sub(const Fp2H& a, const Fp2H& b) { return {fe_sub(a.v, b.v)}; }
FF_INLINE Fp2H fe_neg(const Fp2H& a) { return {fe_neg(a.v)}; }
FF_INLINE Fp2H fe_dbl(const Fp2H& a) { return {fe_dbl(a.v)}; }
FF_INLINE Fp2H fe_mul(const Fp2H& a, const Fp2H& b) {
    const bool c1 = pair_comp() != 0;
    const Fp ao = pair_swap(a.v), bo = pair_swap(b.v);
    const Fp a0 = fp_select(c1, a.v, ao), a1 = fp_select(c1, ao, a.v);
    const Fp m1 = fe_mul(a0, b.v);      // c0 lane: a0 b0   c1 lane: a0 b1
    const Fp m2 = fe_mul(a1, bo);       // c0 lane: a1 b1   c1 lane: a1 b0
    return {c1 ? fe_add(m1, m2) : fe_sub(m1, m2)};
}
FF_INLINE Fp2H fe_sqr(const Fp2H& a) {
    const bool c1 = pair_comp() != 0;
    const Fp ao = pair_swap(a.v);
    const Fp x = c1 ? ao : fe_add(a.v, ao);          // c0 lane: a0 + a1   c1 lane: a0
    const Fp y = c1 ? a.v : fe_sub(a.v, ao);         // c0 lane: a0 - a1   c1 lane: a1
    const Fp m = fe_mul(x, y);
    return {c1 ? fe_dbl(m) : m};
}

}  // namespace zk
