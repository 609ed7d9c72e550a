// Points of BLS12-381 G1 / G2 on their way into and out of the multi-scalar multiplications of msm.hip: the byte <-> Montgomery conversions of
// the ZCash encoding, the checks the reference applies to incoming points (of_bytes_exn, src/lib/zk/curve.ml:199-212: encoding, curve equation,
// prime-order subgroup), the resident window tables 2^(c j) P_i of a base set, the fixed-base products s_i G of keygen (curve.ml:106-109,180)
// and the sum of partial results across devices.  Split off msm.hip in round 5 (one translation unit per concern: points / sort / dispatch).
#include "ec.cuh"
#include "msm.cuh"

#include <stdlib.h>
#include <string.h>

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

// ------------------------------------------------------------------ of_compressed_bytes_exn over whole lists (curve.ml:199-212; round 5)
// The reference's JSON holds every key point COMPRESSED (Bls12_381.G1/G2.to_compressed_bytes): a 2^20-constraint key is five million square roots and
// subgroup checks on the way in -- half an hour of one host core through zk_g1/g2_decompress, a second here.  One lane per point: x from its 48 / 96
// big-endian bytes, y = sqrt(x^3 + b) by the power (p + 1) / 4 (p = 3 mod 4; in Fp2 through the norm), the sign bit's choice of root, then the
// subgroup check and the encoder above.  Same verdicts as the host functions: flag 2 = bad encoding (compression bit missing, coordinate >= p),
// 1 = x is not the abscissa of a curve point.
__device__ static const uint32_t FP_SQRT_EXP[12] = {0xffffeaabu, 0xee7fbfffu, 0xac54ffffu, 0x07aaffffu, 0x3dac3d89u, 0xd9cc34a8u,
                                                    0x3ce144afu, 0xd91dd2e1u, 0x90d2eb35u, 0x92c6e9edu, 0x8e5ff9a6u, 0x0680447au};      // (p + 1) / 4, 379 bits
__device__ static const uint32_t FP_HALF_PM1[12] = {0xffffd555u, 0xdcff7fffu, 0x58a9ffffu, 0x0f55ffffu, 0x7b587b12u, 0xb3986950u,
                                                    0x79c2895fu, 0xb23ba5c2u, 0x21a5d66bu, 0x258dd3dbu, 0x1cbff34du, 0x0d0088f5u};      // (p - 1) / 2
template <int A> FF_INLINE FpB<2> fp_red2(const FpB<A>& a) { return fe_mul(a, fp_one()); }          // the same value below 2 p
__device__ __noinline__ FpB<2> fp_pow_sqrt(const FpB<2>& a) {          // a^((p + 1) / 4): the exponent is a constant, every lane takes the same branches
    FpB<2> acc = fp_one();
    for (int i = 378; i >= 0; i--) {
        acc = fe_sqr(acc);
        if ((FP_SQRT_EXP[i >> 5] >> (i & 31)) & 1u) acc = fe_mul(acc, a);
    }
    return acc;
}
FF_INLINE bool fp_sqrt_dev(FpB<2>& out, const FpB<2>& a) {
    out = fp_pow_sqrt(a);
    return fe_eq(fe_sqr(out), a);
}
template <int A> FF_INLINE bool fp_is_large(const FpB<A>& y) {          // canonical integer of y > (p - 1) / 2: the ZCash sign of a coordinate
    const FpWords w = fp_from_mont(y);
    bool gt = false, decided = false;
#pragma unroll
    for (int k = 11; k >= 0; k--)
        if (!decided && w.w[k] != FP_HALF_PM1[k]) { gt = w.w[k] > FP_HALF_PM1[k]; decided = true; }
    return gt;
}
// a square root of a in Fp2 = Fp[u] / (u^2 + 1) (either one: the caller fixes the sign); false: a is not a square
FF_INLINE bool fp2_sqrt_dev(Fp2B<2>& out, const Fp2B<2>& a) {
    if (fe_is_zero(a.c1)) {
        FpB<2> r;
        if (fp_sqrt_dev(r, a.c0)) { out = {r, fp_zero()}; return true; }
        if (fp_sqrt_dev(r, fp_red2(fe_neg(a.c0)))) { out = {fp_zero(), r}; return true; }          // (r u)^2 = -r^2
        return false;
    }
    FpB<2> s;
    if (!fp_sqrt_dev(s, fp_red2(fe_add(fe_sqr(a.c0), fe_sqr(a.c1))))) return false;                 // the norm of a square is a square
    const FpB<2> half = fe_inv(fe_dbl(fp_one()));
    FpB<2> x0;
    if (!fp_sqrt_dev(x0, fe_mul(fe_add(a.c0, s), half)) && !fp_sqrt_dev(x0, fe_mul(fe_sub(a.c0, s), half))) return false;
    const FpB<2> x1 = fe_mul(a.c1, fe_inv(fe_dbl(x0)));
    out = {x0, x1};
    const Fp2B<4> sq = fe_sqr(out);
    return fe_eq(sq.c0, a.c0) && fe_eq(sq.c1, a.c1);
}
__global__ __launch_bounds__(128) void k_decompress_g1(uint8_t* __restrict__ dense, const uint8_t* __restrict__ in, uint64_t n, int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    alignas(16) uint8_t xb[48];
    for (int k = 0; k < 48; k++) xb[k] = in[48 * i + k];
    const uint8_t f0 = xb[0];
    xb[0] &= 0x1f;
    Aff<Fp> out = aff_inf<Fp>();
    if (!(f0 & 0x80)) atomicOr(flag, 2);
    else if (!(f0 & 0x40)) {
        const FpWords xw = fpw_from_be(xb);
        if (!fpw_canonical(xw)) atomicOr(flag, 2);
        else {
            const FpB<2> x = fp_to_mont(xw);
            const FpB<2> rhs = fp_red2(fe_add(fe_mul(fe_sqr(x), x), FieldOps<Fp>::curve_b()));
            FpB<2> y;
            if (!fp_sqrt_dev(y, rhs)) atomicOr(flag, 1);
            else {
                if (fp_is_large(y) != ((f0 & 0x20) != 0)) y = fp_red2(fe_neg(y));
                out = {x, y};
            }
        }
    }
    aff_store<Fp>(dense + 96 * i, out);
}
__global__ __launch_bounds__(128) void k_decompress_g2(uint8_t* __restrict__ dense, const uint8_t* __restrict__ in, uint64_t n, int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    alignas(16) uint8_t xb[96];          // x.c1 (with the flags) | x.c0
    for (int k = 0; k < 96; k++) xb[k] = in[96 * i + k];
    const uint8_t f0 = xb[0];
    xb[0] &= 0x1f;
    Aff<Fp2> out = aff_inf<Fp2>();
    if (!(f0 & 0x80)) atomicOr(flag, 2);
    else if (!(f0 & 0x40)) {
        const FpWords x1w = fpw_from_be(xb), x0w = fpw_from_be(xb + 48);
        if (!fpw_canonical(x1w) || !fpw_canonical(x0w)) atomicOr(flag, 2);
        else {
            const Fp2B<2> x = {fp_to_mont(x0w), fp_to_mont(x1w)};
            const auto cube = fe_mul(fe_sqr(x), x);
            const Fp2 b = FieldOps<Fp2>::curve_b();
            const Fp2B<2> rhs = {fp_red2(fe_add(cube.c0, b.c0)), fp_red2(fe_add(cube.c1, b.c1))};
            Fp2B<2> y;
            if (!fp2_sqrt_dev(y, rhs)) atomicOr(flag, 1);
            else {
                const bool large = fe_is_zero(y.c1) ? fp_is_large(y.c0) : fp_is_large(y.c1);
                if (large != ((f0 & 0x20) != 0)) y = {fp_red2(fe_neg(y.c0)), fp_red2(fe_neg(y.c1))};
                out = {Fp2(x), Fp2(y)};
            }
        }
    }
    aff_store<Fp2>(dense + 192 * i, out);
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

// acc[i] = sum_j parts[j * npoints + i] on dense XYZZ points: the sum of the ranks' / devices' partial sums of a proof.  One lane per point in G1, a lane PAIR
// in G2 (F = Fp2H), additions expanded in place on the lane's registers: round 3's form (a whole Fp2 point per lane through the out-of-line addition)
// carried 3 KiB of private memory per lane -- 1.6 GiB of scratch reserved on every queue the kernel was dispatched on (DESIGN A.2).
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
// ================================================================== host side
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
// n compressed points (host) -> n uncompressed points (host), every one decoded, checked (curve, subgroup) and re-encoded on the device
int points_decompress(Curve curve, const uint8_t* in, uint64_t n, uint8_t* out, hipStream_t s) {
    if (!n) return ZK_OK;
    const size_t cb = aff_bytes(curve) / 2, ub = aff_bytes(curve);
    DevBuf din, dense, dout, flag;
    ZKCHK(din.alloc(cb * n));
    ZKCHK(dense.alloc(ub * n));
    ZKCHK(dout.alloc(ub * n));
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
    HIPCHK(hipMemcpyAsync(din.p, in, cb * n, hipMemcpyHostToDevice, s));
    if (curve == CURVE_G1) {
        hipLaunchKernelGGL(k_decompress_g1, grid_for(n, 128), dim3(128), 0, s, dense.as<uint8_t>(), (const uint8_t*)din.as<uint8_t>(), n, flag.as<int>());
        hipLaunchKernelGGL(k_subgroup_check<Fp>, grid_for(n, 128), dim3(128), 0, s, (const uint8_t*)dense.as<uint8_t>(), n, flag.as<int>());
    } else {
        hipLaunchKernelGGL(k_decompress_g2, grid_for(n, 128), dim3(128), 0, s, dense.as<uint8_t>(), (const uint8_t*)din.as<uint8_t>(), n, flag.as<int>());
        hipLaunchKernelGGL(k_subgroup_check<Fp2>, grid_for(n, 128), dim3(128), 0, s, (const uint8_t*)dense.as<uint8_t>(), n, flag.as<int>());
    }
    HIPCHK(hipGetLastError());
    ZKCHK(points_affine_to_bytes(curve, dout.p, dense.p, n, s));
    int h = 0;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipMemcpyAsync(out, dout.p, ub * n, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    if (h & 2) ZK_FAIL(ZK_ERR_ARG, "decompress: a point's compression flag is not set or a coordinate is >= p");
    if (h & 1) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "decompress: an abscissa is not on the curve");
    if (h & 4) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "decompress: a point is on the curve but outside the prime-order subgroup");
    return ZK_OK;
}

// ---- may one product read another's sort?  (msm.cuh: msm_accumulate_sorted)
__global__ void k_bytes_differ(const uint8_t* __restrict__ a, const uint8_t* __restrict__ b, uint64_t n, int* flag) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n && a[i] != b[i]) *flag = 1;
}
int msm_bases_same_geometry(const MsmBases& a, const MsmBases& b, bool* same, hipStream_t s) {
    *same = false;
    if (a.n != b.n || a.c != b.c || a.nw != b.nw || a.precomp != b.precomp || a.fold != b.fold) return ZK_OK;
    DevBuf flag;
    ZKCHK(flag.alloc(4));
    HIPCHK(hipMemsetAsync(flag.p, 0, 4, s));
    hipLaunchKernelGGL(k_bytes_differ, grid_for(a.n, 256), dim3(256), 0, s, (const uint8_t*)a.ident.as<uint8_t>(), (const uint8_t*)b.ident.as<uint8_t>(), a.n, flag.as<int>());
    HIPCHK(hipGetLastError());
    int h = 1;
    HIPCHK(hipMemcpyAsync(&h, flag.p, 4, hipMemcpyDeviceToHost, s));
    HIPCHK(hipStreamSynchronize(s));
    *same = h == 0;
    return ZK_OK;
}

}  // namespace zk
