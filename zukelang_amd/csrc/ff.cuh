// BLS12-381 field arithmetic for gfx950 (CDNA4), Montgomery form, limbs held in VGPRs.
//
// CDNA4 has no 64x64 multiplier; the widest integer multiply-add is v_mad_u64_u32
// (32x32 + 64 -> 64, ~5 issue cycles per wave64; plain VOP2 integer ops ~2.5, measured with
// scripts/proto/valu_rate.hip).
//
//  * Fr (255 bits, the NTT / QAP field): 8 x u32, R = 2^256, generated product-scanning code with
//    explicit carries (ff_mul_gen.cuh).  Fr work is bandwidth / launch bound, not multiplier bound.
//  * Fp (381 bits, the curve coordinate field -- >95% of the prover's ALU work): 14 limbs of 29
//    bits, R = 2^406.  A column of the product holds at most 28 partial products < 2^58, which
//    never overflows the 64-bit accumulator: EVERY partial product is exactly one v_mad_u64_u32
//    and there is no carry instruction, no vcc dependency chain (the 12 x 32-bit scheme paid
//    v_mad + v_addc per product and stalled on the carry at the 1-2 waves/SIMD the EC kernels run at:
//    812 instructions and 1.24 us single-wave latency per product against 527 / 0.81 us here).
//    Values are kept LAZILY reduced: additions and subtractions never compare against p, they only
//    re-normalise limbs with one parallel carry pass.  The value bound (a multiple of p) is part of
//    the TYPE, FpB<B>: value < B * p.  fe_sub picks the multiple of p it must add from the bound of
//    its subtrahend, fe_mul static_asserts that the product of its operand bounds fits the
//    Montgomery headroom (2^406 / p ~ 2^25) and returns FpB<2>.  A formula whose bounds do not
//    close does not compile.  Memory keeps the dense format: 12 x u32, fully reduced.
//
// Replaces, for the prove path, the Fr / Fq arithmetic the reference obtains from opam
// bls12-381 (src/lib/zk/curve.ml:121-140 Fr ops; G1/G2 coordinates curve.ml:159-191).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#define FF_INLINE __device__ __forceinline__

#include "ff_mul_gen.cuh"
#include "fp29_consts.cuh"

namespace zk {

struct FrParams {
    static constexpr int N = 8;
    static constexpr uint32_t INV = 0xffffffffu;  // -r^-1 mod 2^32
};
// Fp as 12 dense 32-bit words: the memory format and the operand of the binary inversion
struct FpParams {
    static constexpr int N = 12;
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
__device__ static const uint32_t FR_R3[8] = {0x439b73afu, 0xc62c1807u, 0x8cf06990u, 0x1b3e0d18u,
                                             0xc7b5f418u, 0x73d13c71u, 0xc8db33e9u, 0x6e2a5bb9u};

template <class P> struct Consts;
template <> struct Consts<FrParams> {
    static FF_INLINE uint32_t mod(int i) { return FR_MOD[i]; }
    static FF_INLINE uint32_t r1(int i) { return FR_R1[i]; }
    static FF_INLINE uint32_t r2(int i) { return FR_R2[i]; }
    static FF_INLINE uint32_t r3(int i) { return FR_R3[i]; }
};
template <> struct Consts<FpParams> {
    static FF_INLINE uint32_t mod(int i) { return FP_MOD[i]; }
};

template <class P> struct Fe {
    static constexpr int N = P::N;
    uint32_t v[P::N];
};
using Fr = Fe<FrParams>;
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

// Montgomery product a*b*R^-1 mod r, fully reduced.  The straight-line body is generated
// (scripts/gen_mont_mul.py): finely integrated product scanning, 2N^2 + N partial products at two
// instructions each (v_mad_u64_u32 + v_addc_co_u32), one asm statement per column.
template <class P> FF_INLINE Fe<P> fe_mul_inline(const Fe<P>& a, const Fe<P>& b) {
    static_assert(P::N == 8, "generic Fe<P> arithmetic is the scalar field only");
    Fe<P> r;
    mont_mul_fr(r.v, a.v, b.v);
    // the product is < 2r < 2^(32N): one conditional subtraction
    fe_cond_sub(r);
    return r;
}
template <class P> FF_INLINE Fe<P> fe_mul(const Fe<P>& a, const Fe<P>& b) { return fe_mul_inline(a, b); }
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
// canonical a < mod ?   (dense 32-bit words)
template <class P> FF_INLINE bool words_are_canonical(const uint32_t* a) {
    uint64_t bw = 0;
#pragma unroll
    for (int i = 0; i < P::N; i++) {
        uint64_t d = (uint64_t)a[i] - Consts<P>::mod(i) - bw;
        bw = (d >> 32) & 1;
    }
    return bw != 0;
}
template <class P> FF_INLINE bool fe_is_canonical(const Fe<P>& a) { return words_are_canonical<P>(a.v); }
// Modular inverse by the binary extended Euclid on plain dense words (shifts and subtractions only):
// ~2 * bits iterations of a few N-word operations, an order of magnitude cheaper than Fermat's
// a^(p-2) (~580 Montgomery products) -- it matters because conversions to affine sit on
// single-lane tails.  x -> x^-1 mod p with NO Montgomery correction; inv(0) = 0.
template <class P> __device__ __noinline__ void words_inv(uint32_t* __restrict__ out, const uint32_t* __restrict__ a) {
    constexpr int N = P::N;
    uint32_t u[N], v[N], x1[N], x2[N];
    uint32_t any = 0;
#pragma unroll
    for (int i = 0; i < N; i++) { u[i] = a[i]; v[i] = Consts<P>::mod(i); x1[i] = 0; x2[i] = 0; any |= a[i]; }
    if (!any) {
#pragma unroll
        for (int i = 0; i < N; i++) out[i] = 0;
        return;
    }
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
    const bool use1 = is_one(u);
#pragma unroll
    for (int i = 0; i < N; i++) out[i] = use1 ? x1[i] : x2[i];
}
// Input and output in Montgomery form: (aR)^-1 * R^3 * R^-1 = a^-1 R.
template <class P> FF_INLINE Fe<P> fe_inv(const Fe<P>& a) {
    Fe<P> r, r3;
    words_inv<P>(r.v, a.v);
#pragma unroll
    for (int i = 0; i < P::N; i++) r3.v[i] = Consts<P>::r3(i);
    return fe_mul(r, r3);
}
template <class P> FF_INLINE Fe<P> fe_from_u32(uint32_t x) {
    Fe<P> r = fe_zero<P>();
    r.v[0] = x;
    return fe_to_mont(r);
}

// 128-bit vector loads / stores of whole elements (32 B Fr = 2 x dwordx4).
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

// ================================================================== Fp: 14 limbs x 29 bits, R = 2^406
// Invariants of every FpB<B> held in registers:
//   value < B * p;   limbs 0..12 <= 2^29 + 7 ("weakly normalised": one carry pass over limbs < 2^32 leaves at most 2^29 - 1 + 7), limb 13 holds the rest (small).
// Column bound of the product: 14 (2^29+7)^2 + 14 (2^29)^2 + carry < 2^62.9.
typedef uint32_t u32x8 __attribute__((ext_vector_type(8)));
typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
typedef uint32_t u32x2 __attribute__((ext_vector_type(2)));
static constexpr int FPL = FP29_L;
static constexpr int FP_MAX_BOUND = 8192;            // the largest multiple of p in FP29_KP is 2^13
static constexpr long long FP_MUL_BUDGET = 1ll << 25;  // 2^406 / p = 2^25.3: operand bounds A * B <= this gives a product < 2p

struct FpRaw {
    uint32_t v[FPL];
};
template <int B> struct FpB {
    static_assert(B >= 1 && B <= FP_MAX_BOUND, "Fp value bound out of range");
    static constexpr int BOUND = B;
    uint32_t v[FPL];
    FpB() = default;
    // a looser bound is always valid; a tighter one is a compile error
    template <int A> FF_INLINE FpB(const FpB<A>& o) {
        static_assert(A <= B, "Fp value bound does not fit the destination type");
#pragma unroll
        for (int i = 0; i < FPL; i++) v[i] = o.v[i];
    }
};
static constexpr int FP_REST = 64;          // bound of values "at rest" (struct members, loop-carried accumulators)
using Fp = FpB<FP_REST>;

constexpr int fp_ks(int b) { int k = 2; while (k < b + 1) k <<= 1; return k; }   // multiple of p added by a - b, b < B p
constexpr int fp_ki(int k) { int i = 0; while ((2 << i) < k) i++; return i; }     // its row in FP29_KP

template <int B> FF_INLINE FpB<B> fp_from_raw(const FpRaw& r) {
    FpB<B> x;
#pragma unroll
    for (int i = 0; i < FPL; i++) x.v[i] = r.v[i];
    return x;
}
// UNCHECKED re-typing for values whose tighter bound the caller knows (e.g. just loaded from memory: < p)
template <int T, int A> FF_INLINE FpB<T> fp_assume(const FpB<A>& a) {
    FpB<T> x;
#pragma unroll
    for (int i = 0; i < FPL; i++) x.v[i] = a.v[i];
    return x;
}
FF_INLINE FpB<1> fp_zero() {
    FpB<1> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = 0;
    return r;
}
FF_INLINE FpB<1> fp_one() {
    FpB<1> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = FP29_R1[i];
    return r;
}
// one parallel carry step: limbs < 2^32 in, weakly normalised out; the value is unchanged
FF_INLINE void fp_carry(uint32_t* t) {
    uint32_t c[FPL - 1];
#pragma unroll
    for (int i = 0; i < FPL - 1; i++) c[i] = t[i] >> FP29_W;
    t[0] &= FP29_MASK;
#pragma unroll
    for (int i = 1; i < FPL - 1; i++) t[i] = (t[i] & FP29_MASK) + c[i - 1];
    t[FPL - 1] += c[FPL - 2];
}
template <int A, int B> FF_INLINE FpB<A + B> fe_add(const FpB<A>& a, const FpB<B>& b) {
    FpB<A + B> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = a.v[i] + b.v[i];
    fp_carry(r.v);
    return r;
}
// a - b + K p with K = fp_ks(B) > b / p.  The spread limbs of K p are >= 2^30 - 2 >= any limb of b, so no
// limb borrows; the top limb may wrap below zero before the carry pass and is exact after it because the
// value a + K p - b >= p exceeds what limbs 0..12 can hold.
template <int A, int B> FF_INLINE FpB<A + fp_ks(B)> fe_sub(const FpB<A>& a, const FpB<B>& b) {
    constexpr int KI = fp_ki(fp_ks(B));
    static_assert(KI < FP29_NK, "subtrahend bound too large");
    FpB<A + fp_ks(B)> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = a.v[i] + (FP29_KP[KI][i] - b.v[i]);
    fp_carry(r.v);
    return r;
}
// a - b - 2 c + K p in ONE carry pass, K = fp_ks(B + 2 C): X3 = R^2 - PPP - 2 Q of every group addition (composed from fe_sub / fe_dbl it was three carry
// passes).  The WIDE spread of K p (limbs >= 2^31 - 4) covers b_i + 2 c_i <= 3 (2^29 + 7) without a borrow and keeps every limb below 2^32.
template <int A, int B, int C> FF_INLINE FpB<A + fp_ks(B + 2 * C)> fe_sub_sub_dbl(const FpB<A>& a, const FpB<B>& b, const FpB<C>& c) {
    constexpr int KI = fp_ki(fp_ks(B + 2 * C));
    static_assert(KI >= 1 && KI < FP29_NK, "subtrahend bound out of range");
    FpB<A + fp_ks(B + 2 * C)> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = a.v[i] + (FP29_KPW[KI][i] - b.v[i] - (c.v[i] << 1));
    fp_carry(r.v);
    return r;
}
// K p - a WITHOUT a carry pass (narrow spread: limbs in [1, 2^30]): legal only as the second LEFT factor of the fused double product below, whose
// columns then hold 14 (2^29+7)^2 + 14 (2^30)(2^29+7) + 14 (2^29)(2^29) + carry < 56 * 2^58 * (1 + 2^-20) < 2^64.
template <int K> struct FpLazyNeg {
    static constexpr int BOUND = K;
    uint32_t v[FPL];
};
template <int A> FF_INLINE FpLazyNeg<fp_ks(A)> fe_neg_lazy(const FpB<A>& a) {
    constexpr int KI = fp_ki(fp_ks(A));
    static_assert(KI < FP29_NK, "bound too large");
    FpLazyNeg<fp_ks(A)> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = FP29_KPN[KI][i] - a.v[i];
    return r;
}
template <int A> FF_INLINE FpB<fp_ks(A)> fe_neg(const FpB<A>& a) {
    constexpr int KI = fp_ki(fp_ks(A));
    static_assert(KI < FP29_NK, "bound too large");
    FpB<fp_ks(A)> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = FP29_KP[KI][i] - a.v[i];
    fp_carry(r.v);
    return r;
}
template <int A> FF_INLINE FpB<2 * A> fe_dbl(const FpB<A>& a) {
    FpB<2 * A> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = a.v[i] << 1;
    fp_carry(r.v);
    return r;
}

// ---- Montgomery product (operands weakly normalised, A * B <= FP_MUL_BUDGET): (a b + m p) / 2^406 < 2p.
// Column-wise, quotient digit m_k chosen as soon as column k is complete; the two sums are interleaved by
// the compiler into independent accumulator chains.
FF_INLINE void fp_mul_limbs(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    uint64_t acc = 0;
    uint32_t m[FPL];
#pragma unroll
    for (int k = 0; k < FPL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        m[k] = ((uint32_t)acc * FP29_NINV) & FP29_MASK;
        acc += (uint64_t)m[k] * FP29_MOD[0];
        acc >>= FP29_W;
    }
#pragma unroll
    for (int k = FPL; k < 2 * FPL - 1; k++) {
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)a[i] * b[k - i];
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        r[k - FPL] = (uint32_t)acc & FP29_MASK;
        acc >>= FP29_W;
    }
    r[FPL - 1] = (uint32_t)acc;
}
// square: the off-diagonal products are taken once against the doubled operand (105 + 196 multiply-adds
// instead of 196 + 196)
FF_INLINE void fp_sqr_limbs(uint32_t* __restrict__ r, const uint32_t* __restrict__ a) {
    uint64_t acc = 0;
    uint32_t m[FPL], d[FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) d[i] = a[i] << 1;
#pragma unroll
    for (int k = 0; k < FPL; k++) {
#pragma unroll
        for (int i = 0; 2 * i < k; i++) acc += (uint64_t)a[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        m[k] = ((uint32_t)acc * FP29_NINV) & FP29_MASK;
        acc += (uint64_t)m[k] * FP29_MOD[0];
        acc >>= FP29_W;
    }
#pragma unroll
    for (int k = FPL; k < 2 * FPL - 1; k++) {
#pragma unroll
        for (int i = k - FPL + 1; 2 * i < k; i++) acc += (uint64_t)a[i] * d[k - i];
        if ((k & 1) == 0) acc += (uint64_t)a[k / 2] * a[k / 2];
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        r[k - FPL] = (uint32_t)acc & FP29_MASK;
        acc >>= FP29_W;
    }
    r[FPL - 1] = (uint32_t)acc;
}
// Real functions: ~0.5k instructions each, shared by every caller of a translation unit (the hot loops
// stay inside the instruction cache, the library compiles in a minute).  Operands travel as vector-typed
// arguments, which the AMDGPU calling convention keeps entirely in VGPRs (a by-value struct of 14 words
// is passed through scratch memory instead).
#define FP_ARGS(x) u32x8 x##0, u32x4 x##1, u32x2 x##2
#define FP_UNPACK_ARGS(t, x)                                     \
    uint32_t t[FPL];                                             \
    _Pragma("unroll") for (int i = 0; i < 8; i++) t[i] = x##0[i]; \
    _Pragma("unroll") for (int i = 0; i < 4; i++) t[8 + i] = x##1[i]; \
    t[12] = x##2[0];                                             \
    t[13] = x##2[1];
#define FP_PASS(a)                                                                                      \
    u32x8{(a)[0], (a)[1], (a)[2], (a)[3], (a)[4], (a)[5], (a)[6], (a)[7]}, u32x4{(a)[8], (a)[9], (a)[10], (a)[11]}, \
        u32x2 { (a)[12], (a)[13] }
__device__ __noinline__ static FpRaw fp_mul_call(FP_ARGS(a), FP_ARGS(b)) {
    FP_UNPACK_ARGS(x, a)
    FP_UNPACK_ARGS(y, b)
    FpRaw r;
    fp_mul_limbs(r.v, x, y);
    return r;
}
__device__ __noinline__ static FpRaw fp_sqr_call(FP_ARGS(a)) {
    FP_UNPACK_ARGS(x, a)
    FpRaw r;
    fp_sqr_limbs(r.v, x);
    return r;
}
// A translation unit that defines ZK_FP_INLINE_MUL before including this header gets the products
// expanded in place (the G1 bucket-accumulation loop: no call boundary, so the loads of the next point
// stay in flight across the whole mixed addition).
template <int A, int B> FF_INLINE FpB<2> fe_mul(const FpB<A>& a, const FpB<B>& b) {
    static_assert((long long)A * B <= FP_MUL_BUDGET, "operand bounds exceed the Montgomery headroom");
#ifdef ZK_FP_INLINE_MUL
    FpB<2> r;
    fp_mul_limbs(r.v, a.v, b.v);
    return r;
#else
    return fp_from_raw<2>(fp_mul_call(FP_PASS(a.v), FP_PASS(b.v)));
#endif
}
template <int A> FF_INLINE FpB<2> fe_sqr(const FpB<A>& a) {
    static_assert((long long)A * A <= FP_MUL_BUDGET, "operand bound exceeds the Montgomery headroom");
#ifdef ZK_FP_INLINE_MUL
    FpB<2> r;
    fp_sqr_limbs(r.v, a.v);
    return r;
#else
    return fp_from_raw<2>(fp_sqr_call(FP_PASS(a.v)));
#endif
}
template <int A, int B> FF_INLINE FpB<2> fe_mul_inline(const FpB<A>& a, const FpB<B>& b) {
    static_assert((long long)A * B <= FP_MUL_BUDGET, "operand bounds exceed the Montgomery headroom");
    FpB<2> r;
    fp_mul_limbs(r.v, a.v, b.v);
    return r;
}

// ---- full reduction: any value < 8192 p -> the unique representative < p with exact 29-bit limbs
__device__ __noinline__ static FpRaw fp_canon_call(FP_ARGS(a)) {
    FP_UNPACK_ARGS(t, a)
#pragma unroll
    for (int i = 0; i < FPL - 1; i++) {           // exact limbs
        t[i + 1] += t[i] >> FP29_W;
        t[i] &= FP29_MASK;
    }
    // quotient estimate from the top two limbs against (p >> 348) + 1, minus one: q in {Q-2, Q-1, Q}
    // for Q = floor(value / p) (float error < 2^-9 at Q < 2^13)
    const float xf = (float)t[13] * 536870912.0f + (float)t[12];
    uint32_t q = (uint32_t)(xf * FP29_PTOP_INV);
    q = q ? q - 1 : 0;
    int64_t cy = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        const int64_t cur = (int64_t)t[i] - (int64_t)((uint64_t)q * FP29_MOD[i]) + cy;
        t[i] = i < FPL - 1 ? ((uint32_t)cur & FP29_MASK) : (uint32_t)cur;
        cy = cur >> FP29_W;
    }
#pragma unroll
    for (int rep = 0; rep < 2; rep++) {            // the remainder is < 3p
        uint32_t u[FPL];
        int32_t bw = 0;
#pragma unroll
        for (int i = 0; i < FPL; i++) {
            const int32_t d = (int32_t)t[i] - (int32_t)FP29_MOD[i] - bw;
            bw = (d >> 31) & 1;
            u[i] = i < FPL - 1 ? ((uint32_t)d & FP29_MASK) : (uint32_t)d;
        }
        if (!bw) {
#pragma unroll
            for (int i = 0; i < FPL; i++) t[i] = u[i];
        }
    }
    FpRaw r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = t[i];
    return r;
}
template <int A> FF_INLINE FpB<1> fp_canon(const FpB<A>& a) { return fp_from_raw<1>(fp_canon_call(FP_PASS(a.v))); }
// The same reduction expanded in place, for the slow path of the zero test inside loops that must not contain a call (a call pins every live value
// to the callee-saved half of the register file and drains the loads in flight; the slow path itself runs once in ~2^23 tests)
FF_INLINE bool fp_is_zero_mod_p_inline(const uint32_t* __restrict__ a) {
    uint32_t t[FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) t[i] = a[i];
#pragma unroll
    for (int i = 0; i < FPL - 1; i++) {
        t[i + 1] += t[i] >> FP29_W;
        t[i] &= FP29_MASK;
    }
    // value = k p exactly  <=>  every limb of value - k p is zero, k = limb0 / p mod 2^29 (the only candidate)
    const uint32_t k = (t[0] * FP29_PINV) & FP29_MASK;
    int64_t cy = 0;
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        const int64_t cur = (int64_t)t[i] - (int64_t)((uint64_t)k * FP29_MOD[i]) + cy;
        o |= i < FPL - 1 ? ((uint32_t)cur & FP29_MASK) : (uint32_t)cur;
        cy = cur >> FP29_W;
    }
    return o == 0 && cy == 0;
}

// value == 0 mod p ?  Exact zeros (how the identity is encoded) are caught first; otherwise a multiple
// k p, k < B, must have k = limb0 * p^-1 mod 2^29 < B: everything else (all but B in 2^29 values) is
// rejected by one multiply, the rare survivor is fully reduced.
template <int B> FF_INLINE bool fe_is_zero(const FpB<B>& a) {
    uint32_t o = 0;
#pragma unroll
    for (int i = 0; i < FPL; i++) o |= a.v[i];
    if (o == 0) return true;
    if constexpr (B == 1) return false;
    else {
        const uint32_t k = (a.v[0] * FP29_PINV) & FP29_MASK;
        if (k >= (uint32_t)B) return false;
#ifdef ZK_FP_INLINE_MUL
        return fp_is_zero_mod_p_inline(a.v);
#else
        const FpB<1> c = fp_canon(a);
        uint32_t z = 0;
#pragma unroll
        for (int i = 0; i < FPL; i++) z |= c.v[i];
        return z == 0;
#endif
    }
}
template <int A, int B> FF_INLINE bool fe_eq(const FpB<A>& a, const FpB<B>& b) { return fe_is_zero(fe_sub(a, b)); }

// ---- dense 12 x u32 memory format <-> limbs
struct FpWords {
    uint32_t w[12];
};
FF_INLINE FpWords fp_pack(const FpB<1>& a) {          // a fully reduced, exact limbs
    FpWords r;
#pragma unroll
    for (int j = 0; j < 12; j++) {
        const int bit = 32 * j, k = bit / FP29_W, s = bit % FP29_W;
        uint64_t x = a.v[k] >> s;
        x |= (uint64_t)a.v[k + 1] << (FP29_W - s);
        if (2 * FP29_W - s < 32 && k + 2 < FPL) x |= (uint64_t)a.v[k + 2] << (2 * FP29_W - s);
        r.w[j] = (uint32_t)x;
    }
    return r;
}
FF_INLINE FpB<1> fp_unpack(const FpWords& a) {        // a < p
    FpB<1> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        const int bit = FP29_W * i, k = bit / 32, s = bit % 32;
        uint64_t x = a.w[k] >> s;
        if (k + 1 < 12) x |= (uint64_t)a.w[k + 1] << (32 - s);
        r.v[i] = (uint32_t)x & FP29_MASK;
    }
    return r;
}
FF_INLINE FpWords fpw_load(const void* p) {
    FpWords r;
    const uint4* q = reinterpret_cast<const uint4*>(p);
#pragma unroll
    for (int i = 0; i < 3; i++) {
        uint4 x = q[i];
        r.w[4 * i] = x.x; r.w[4 * i + 1] = x.y; r.w[4 * i + 2] = x.z; r.w[4 * i + 3] = x.w;
    }
    return r;
}
FF_INLINE void fpw_store(void* p, const FpWords& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
#pragma unroll
    for (int i = 0; i < 3; i++) q[i] = make_uint4(a.w[4 * i], a.w[4 * i + 1], a.w[4 * i + 2], a.w[4 * i + 3]);
}
FF_INLINE FpB<1> fp_load(const void* p) { return fp_unpack(fpw_load(p)); }
template <int A> FF_INLINE void fp_store(void* p, const FpB<A>& a) { fpw_store(p, fp_pack(fp_canon(a))); }

// plain integer x < p (dense words) <-> Montgomery form
FF_INLINE FpB<2> fp_to_mont(const FpWords& x) {
    FpB<1> r2;
#pragma unroll
    for (int i = 0; i < FPL; i++) r2.v[i] = FP29_R2[i];
    return fe_mul(fp_unpack(x), r2);
}
template <int A> FF_INLINE FpWords fp_from_mont(const FpB<A>& a) {
    FpB<1> one = fp_zero();
    one.v[0] = 1;
    return fp_pack(fp_canon(fe_mul(a, one)));
}
// fe_inv(FpB): see fp_inv.cuh (included at the end of this header) -- the lockstep inversion serves every caller.
template <int A> FF_INLINE FpB<2> fe_inv(const FpB<A>& a);

// ------------------------------------------------------------------ Fp2 = Fp[u]/(u^2 + 1)
template <int B> struct Fp2B {
    static constexpr int BOUND = B;
    FpB<B> c0, c1;
    Fp2B() = default;
    template <int A0, int A1> FF_INLINE Fp2B(const FpB<A0>& a, const FpB<A1>& b) : c0(a), c1(b) {}
    template <int A> FF_INLINE Fp2B(const Fp2B<A>& o) : c0(o.c0), c1(o.c1) {}
};
using Fp2 = Fp2B<FP_REST>;
template <int T, int A> FF_INLINE Fp2B<T> fp_assume(const Fp2B<A>& a) { return {fp_assume<T>(a.c0), fp_assume<T>(a.c1)}; }
FF_INLINE Fp2B<1> fp2_zero() { return {fp_zero(), fp_zero()}; }
FF_INLINE Fp2B<1> fp2_one() { return {fp_one(), fp_zero()}; }
template <int B> FF_INLINE bool fe_is_zero(const Fp2B<B>& a) { return fe_is_zero(a.c0) && fe_is_zero(a.c1); }
template <int A, int B> FF_INLINE bool fe_eq(const Fp2B<A>& a, const Fp2B<B>& b) { return fe_eq(a.c0, b.c0) && fe_eq(a.c1, b.c1); }
template <int A, int B> FF_INLINE Fp2B<A + B> fe_add(const Fp2B<A>& a, const Fp2B<B>& b) { return {fe_add(a.c0, b.c0), fe_add(a.c1, b.c1)}; }
template <int A, int B> FF_INLINE Fp2B<A + fp_ks(B)> fe_sub(const Fp2B<A>& a, const Fp2B<B>& b) { return {fe_sub(a.c0, b.c0), fe_sub(a.c1, b.c1)}; }
template <int A> FF_INLINE Fp2B<fp_ks(A)> fe_neg(const Fp2B<A>& a) { return {fe_neg(a.c0), fe_neg(a.c1)}; }
template <int A, int B, int C> FF_INLINE Fp2B<A + fp_ks(B + 2 * C)> fe_sub_sub_dbl(const Fp2B<A>& a, const Fp2B<B>& b, const Fp2B<C>& c) {
    return {fe_sub_sub_dbl(a.c0, b.c0, c.c0), fe_sub_sub_dbl(a.c1, b.c1, c.c1)};
}
template <int A> FF_INLINE Fp2B<2 * A> fe_dbl(const Fp2B<A>& a) { return {fe_dbl(a.c0), fe_dbl(a.c1)}; }
// Karatsuba: 3 base multiplications.  c0 = t0 - t1 < 6p, c1 = s - (t0 + t1) < 10p
template <int A, int B> FF_INLINE Fp2B<10> fe_mul(const Fp2B<A>& a, const Fp2B<B>& b) {
    const FpB<2> t0 = fe_mul(a.c0, b.c0);
    const FpB<2> t1 = fe_mul(a.c1, b.c1);
    const FpB<2> s = fe_mul(fe_add(a.c0, a.c1), fe_add(b.c0, b.c1));
    return {fe_sub(t0, t1), fe_sub(s, fe_add(t0, t1))};
}
// (a0 + a1 u)^2 = (a0+a1)(a0-a1) + 2 a0 a1 u : 2 base multiplications
template <int A> FF_INLINE Fp2B<4> fe_sqr(const Fp2B<A>& a) {
    const FpB<2> p = fe_mul(fe_add(a.c0, a.c1), fe_sub(a.c0, a.c1));
    const FpB<2> q = fe_mul(a.c0, a.c1);
    return {p, fe_dbl(q)};
}
template <int A> FF_INLINE Fp2B<4> fe_inv(const Fp2B<A>& a);      // fp_inv.cuh

// ------------------------------------------------------------------ Fp2 split over a lane pair
// Lane 2k holds the c0 component, lane 2k+1 the c1 component of the same Fp2 value; partners trade
// operands with one DPP quad_perm(1,0,3,2) move per limb.  A product costs each lane two base
// multiplications (a0 b0, a1 b1 | a0 b1, a1 b0) instead of Karatsuba's three on one lane: 4/3 of the
// multiplier work, but every lane carries HALF of each value, so a G2 mixed addition has the register
// footprint of a G1 one (the accumulator never leaves the VGPRs).  Both lanes of a pair must follow
// the same control flow; every predicate below is pair-uniform by construction.
template <int B> struct Fp2HB {
    static constexpr int BOUND = B;
    FpB<B> v;
    Fp2HB() = default;
    template <int A> FF_INLINE Fp2HB(const FpB<A>& a) : v(a) {}
    template <int A> FF_INLINE Fp2HB(const Fp2HB<A>& o) : v(o.v) {}
};
using Fp2H = Fp2HB<FP_REST>;
template <int T, int A> FF_INLINE Fp2HB<T> fp_assume(const Fp2HB<A>& a) { return {fp_assume<T>(a.v)}; }
FF_INLINE uint32_t pair_comp() { return threadIdx.x & 1u; }
template <int B> FF_INLINE FpB<B> pair_swap(const FpB<B>& a) {
    FpB<B> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)a.v[i], 0xB1, 0xF, 0xF, true);
    return r;
}
template <int B> FF_INLINE FpB<B> fp_select(bool take_b, const FpB<B>& a, const FpB<B>& b) {
    FpB<B> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = take_b ? b.v[i] : a.v[i];
    return r;
}
template <int B> FF_INLINE bool fe_is_zero(const Fp2HB<B>& a) {
    const int z = fe_is_zero(a.v) ? 1 : 0;
    return z && __builtin_amdgcn_mov_dpp(z, 0xB1, 0xF, 0xF, true);
}
template <int A, int B> FF_INLINE Fp2HB<A + B> fe_add(const Fp2HB<A>& a, const Fp2HB<B>& b) { return {fe_add(a.v, b.v)}; }
template <int A, int B> FF_INLINE Fp2HB<A + fp_ks(B)> fe_sub(const Fp2HB<A>& a, const Fp2HB<B>& b) { return {fe_sub(a.v, b.v)}; }
template <int A, int B> FF_INLINE bool fe_eq(const Fp2HB<A>& a, const Fp2HB<B>& b) { return fe_is_zero(fe_sub(a, b)); }
template <int A> FF_INLINE Fp2HB<fp_ks(A)> fe_neg(const Fp2HB<A>& a) { return {fe_neg(a.v)}; }
template <int A, int B, int C> FF_INLINE Fp2HB<A + fp_ks(B + 2 * C)> fe_sub_sub_dbl(const Fp2HB<A>& a, const Fp2HB<B>& b, const Fp2HB<C>& c) { return {fe_sub_sub_dbl(a.v, b.v, c.v)}; }
template <int A> FF_INLINE Fp2HB<2 * A> fe_dbl(const Fp2HB<A>& a) { return {fe_dbl(a.v)}; }
// Fp2 product on a lane pair as ONE fused double product per lane, (x1 y1 + x2 y2) / R with a single
// Montgomery reduction (28 + 14 partial products per column still fit 64 bits):
//   c0 lane: a0 b0 + (256p - a1) b1          c1 lane: a0 b1 + a1 b0
// 588 multiply-adds per lane instead of two full products (784) plus a combining add/sub, and the result
// is < 2p.  The partner's operands arrive by DPP inside the function, so the call still passes 28 registers.
FF_INLINE void fp_mul2_limbs(uint32_t* __restrict__ r, const uint32_t* __restrict__ x1, const uint32_t* __restrict__ y1,
                             const uint32_t* __restrict__ x2, const uint32_t* __restrict__ y2) {
    uint64_t acc = 0;
    uint32_t m[FPL];
#pragma unroll
    for (int k = 0; k < FPL; k++) {
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x1[i] * y1[k - i];
#pragma unroll
        for (int i = 0; i <= k; i++) acc += (uint64_t)x2[i] * y2[k - i];
#pragma unroll
        for (int i = 0; i < k; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        m[k] = ((uint32_t)acc * FP29_NINV) & FP29_MASK;
        acc += (uint64_t)m[k] * FP29_MOD[0];
        acc >>= FP29_W;
    }
#pragma unroll
    for (int k = FPL; k < 2 * FPL - 1; k++) {
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)x1[i] * y1[k - i];
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)x2[i] * y2[k - i];
#pragma unroll
        for (int i = k - FPL + 1; i < FPL; i++) acc += (uint64_t)m[i] * FP29_MOD[k - i];
        r[k - FPL] = (uint32_t)acc & FP29_MASK;
        acc >>= FP29_W;
    }
    r[FPL - 1] = (uint32_t)acc;
}
static constexpr int FP2H_NEG_K = 256;        // the c0 lane negates a1 as 256p - a1
FF_INLINE void fp2h_mul_body(uint32_t* __restrict__ r, const uint32_t* __restrict__ a, const uint32_t* __restrict__ b) {
    const bool c1 = pair_comp() != 0;
    constexpr int KI = fp_ki(FP2H_NEG_K);
    uint32_t x1[FPL], x2[FPL], bo[FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        const uint32_t ao = (uint32_t)__builtin_amdgcn_mov_dpp((int)a[i], 0xB1, 0xF, 0xF, true);
        bo[i] = (uint32_t)__builtin_amdgcn_mov_dpp((int)b[i], 0xB1, 0xF, 0xF, true);
        x1[i] = c1 ? ao : a[i];
        x2[i] = c1 ? a[i] : FP29_KPN[KI][i] - ao;          // 256 p - a1, narrow spread, NO carry pass: a lazy left factor (see fe_neg_lazy), limbs <= 2^30
    }
    fp_mul2_limbs(r, x1, b, x2, bo);
}
__device__ __noinline__ static FpRaw fp2h_mul_call(FP_ARGS(a), FP_ARGS(b)) {
    FP_UNPACK_ARGS(x, a)
    FP_UNPACK_ARGS(y, b)
    FpRaw r;
    fp2h_mul_body(r.v, x, y);
    return r;
}
template <int A, int B> FF_INLINE Fp2HB<2> fe_mul(const Fp2HB<A>& a, const Fp2HB<B>& b) {
    static_assert(A < FP2H_NEG_K, "operand bound too large for the in-product negation");
    static_assert(((long long)A + FP2H_NEG_K) * B <= FP_MUL_BUDGET, "operand bounds exceed the Montgomery headroom");
#ifdef ZK_FP_INLINE_MUL
    FpB<2> r;
    fp2h_mul_body(r.v, a.v.v, b.v.v);
    return {r};
#else
    return {fp_from_raw<2>(fp2h_mul_call(FP_PASS(a.v.v), FP_PASS(b.v.v)))};
#endif
}
// c0 lane: (a0 + a1)(a0 - a1)     c1 lane: 2 a0 a1
// The difference a0 - a1 stays LAZY (narrow spread of K p, no carry pass: limbs <= 2^29 + 7 + 2^30): it is one factor of a plain product whose other
// factor is weakly normalised -- 14 (2^29+7)(1.5 * 2^30 + 5) + 14 * 2^58 + carry < 56 * 2^58 * (1 + 2^-20) < 2^64 per column.
template <int A> FF_INLINE Fp2HB<4> fe_sqr(const Fp2HB<A>& a) {
    const bool c1 = pair_comp() != 0;
    const FpB<A> ao = pair_swap(a.v);
    constexpr int KI = fp_ki(fp_ks(A));
    static_assert(KI < FP29_NK, "bound too large");
    FpB<2 * A> x;                            // c0 lane: a0 + a1   c1 lane: a0 (= ao)
    FpB<A + fp_ks(A)> y;                     // c0 lane: a0 - a1   c1 lane: a1 (= a)
#pragma unroll
    for (int i = 0; i < FPL; i++) {
        x.v[i] = c1 ? ao.v[i] : a.v.v[i] + ao.v[i];
        y.v[i] = c1 ? a.v.v[i] : a.v.v[i] + (FP29_KPN[KI][i] - ao.v[i]);
    }
    fp_carry(x.v);
    const FpB<2> m = fe_mul(x, y);
    FpB<4> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = c1 ? m.v[i] << 1 : m.v[i];
    fp_carry(r.v);
    return {r};
}

// a b - c d.  For Fp as ONE fused double product a b + (K p - c) d (one Montgomery reduction instead of
// two and no subtraction afterwards); for the extension types as the plain composition.
__device__ __noinline__ static FpRaw fp_mul2_call(FP_ARGS(a), FP_ARGS(b), const uint32_t* __restrict__ cd) {
    FP_UNPACK_ARGS(x, a)
    FP_UNPACK_ARGS(y, b)
    FpRaw r;
    fp_mul2_limbs(r.v, x, y, cd, cd + FPL);
    return r;
}
template <int A, int B, int C, int D> FF_INLINE FpB<2> fe_mul_sub(const FpB<A>& a, const FpB<B>& b, const FpB<C>& c, const FpB<D>& d) {
    static_assert((long long)A * B + (long long)fp_ks(C) * D <= FP_MUL_BUDGET, "operand bounds exceed the Montgomery headroom");
    const auto nc = fe_neg_lazy(c);          // no carry pass: the fused product's columns have room for ONE factor of limbs <= 2^30
#ifdef ZK_FP_INLINE_MUL
    FpB<2> r;
    fp_mul2_limbs(r.v, a.v, b.v, nc.v, d.v);
    return r;
#else
    uint32_t cd[2 * FPL];
#pragma unroll
    for (int i = 0; i < FPL; i++) { cd[i] = nc.v[i]; cd[FPL + i] = d.v[i]; }
    return fp_from_raw<2>(fp_mul2_call(FP_PASS(a.v), FP_PASS(b.v), cd));
#endif
}
template <class X, class Y, class Z, class W> FF_INLINE auto fe_mul_sub(const X& a, const Y& b, const Z& c, const W& d) { return fe_sub(fe_mul(a, b), fe_mul(c, d)); }

}  // namespace zk

#include "fp_inv.cuh"
