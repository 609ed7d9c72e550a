// Lagrange-form bases DERIVED from a proving key in the reference's format (tau powers), on the device, without tau.
//
// A key an OCaml `keygen` emits (src/groth16/groth16.ml:45-108) holds [tau^k]_1, [tau^k]_2 and [tau^k Z(tau)/delta]_1; the
// prover then needs the MONOMIAL coefficients of v, w, h and pays the O(n log^2 n) basis conversion per proof (frstage.hip).
// With [l_i(tau)]_1, [l_i(tau)]_2 (l_i: Lagrange basis of the QAP's points 0..n-1, QAP.ml:84,92) and [lambda_t(tau) Z(tau)/delta]_1
// (lambda_t: basis of n..2n-2) it needs only VALUES (scope row f4: three convolutions per proof).  Those bases are a linear
// image of the key's own points: with V_ik = i^k, sum_k a_k [tau^k] = sum_i y_i [l_i(tau)] for every polynomial forces
//     [tau^k] = sum_i i^k [l_i(tau)],   i.e.   L = V^-T P .
// The prover's Fr stage factors V^-1 = T . E (values -> Newton coefficients E = Conv_alt . D(1/i!); Newton -> monomial T over
// the subproduct tree), hence
//     L = D(1/i!) . Conv_alt^T . T^T . P :
// the TRANSPOSED tree top-down (every node of 2^l points: upper half <- middle product of the node with the subproduct of its
// left half, lower half unchanged), one correlation with alt[j] = (-1)^j / j!, one scaling -- the very polynomial products of the
// basis conversion, applied to GROUP ELEMENTS: every product is a radix-2 NTT "in the exponent" whose butterflies multiply a point
// by a 255-bit twiddle.  O(n log^2 n) scalar multiplications, ONCE per key: seconds at 2^16, minutes at 2^20.  The h bases use the
// same code on the tree of the shifted points n..2n-2.  scripts/proto/lagrange_derive_model.py is the integer model of the algebra;
// tests compare the derived pools byte for byte with what a keygen that knows tau emits (Groth16.keygen(..., lagrange=True)).
#include "ec.cuh"
#include "endo_consts.cuh"
#include "frstage.cuh"
#include "msm.cuh"

#include <type_traits>

namespace zk {

static inline dim3 g1d(uint64_t n, unsigned t = 256) { return dim3((unsigned)((n + t - 1) / t)); }
template <class T> struct LaneCount { static constexpr uint32_t N = RawLayout<T>::LANES; };

// ---- scalar (canonical Fr, 8 words in memory) times point, complete formulas (identity, equal operands).  Fixed 4-bit windows: the
// multiples 1 P .. 15 P go to a per-lane table in device memory (7 doublings + 7 additions), then every window costs 4 doublings + ONE
// addition of the looked-up multiple per sub-scalar (with per-lane scalars the conditional addition of a bitwise ladder would run in nearly
// every step of a wave).  Round-2 history: bitwise 5 600 field products per multiplication, one 255-bit scalar in 64 windows 3 300
// (4.2 s / 15.8 s / 77 s at 2^16 / 2^18 / 2^20 for a key's three sets), split through the endomorphisms 2 300 on G1 and 1 700 Fp2
// products on G2 (2.7 s / 9.5 s / 45 s).
// The scalar is split through the curve's endomorphisms (Gallant-Lambert-Vanstone on G1, Galbraith-Lin-Scott on G2): the doublings are
// what a 255-bit scalar costs (256 x 9 of 3 300 field products), and they are shared between the sub-scalars of
//   G1:  k = q z^2 + t  =>  k P = (t + q) P + q phi(P),   phi(x, y) = (beta x, y) = [z^2 - 1] (x, y)          2 scalars of <= 129 bits
//   G2:  k = sum_i k_i |z|^i  =>  k P = sum_i (-1)^i k_i psi^i(P),   psi(x, y) = (cx conj x, cy conj y) = [z] (x, y)      4 scalars of 64 bits
// (z = -0xd201000000010000; r = z^4 - z^2 + 1, p = z mod r).  ONE table of the multiples 1 P .. 15 P as before; the image of an entry under
// phi / psi^i costs one / two products by constants (scripts/gen_endo_consts.py derives and CHECKS them against first-principles
// arithmetic).  132 doublings + ~62 additions on G1, 64 + ~60 on G2.  The split is a bitwise long division per lane (~600 steps of a few
// integer instructions: the cost of a handful of field products).
FF_INLINE void glv_split_g1(const uint32_t* __restrict__ k, uint32_t a[5], uint32_t b[4]) {
    const uint64_t d0 = (uint64_t)ENDO_Z2[0] | ((uint64_t)ENDO_Z2[1] << 32), d1 = (uint64_t)ENDO_Z2[2] | ((uint64_t)ENDO_Z2[3] << 32);
    uint64_t r0 = 0, r1 = 0;
    uint32_t q[8];
#pragma unroll 1
    for (int w = 7; w >= 0; w--) {
        const uint32_t bits = k[w];
        uint32_t qw = 0;
#pragma unroll 1
        for (int bt = 31; bt >= 0; bt--) {
            const uint64_t top = r1 >> 63;
            r1 = (r1 << 1) | (r0 >> 63);
            r0 = (r0 << 1) | ((bits >> bt) & 1u);
            const bool ge = top || r1 > d1 || (r1 == d1 && r0 >= d0);
            if (ge) {
                const uint64_t br = r0 < d0 ? 1u : 0u;
                r0 -= d0;
                r1 = r1 - d1 - br;
            }
            qw = (qw << 1) | (ge ? 1u : 0u);
        }
        q[w] = qw;
    }
    // k < r < z^4: the quotient has 128 bits; a = t + q has at most 129
    uint64_t c = 0;
    const uint32_t t[4] = {(uint32_t)r0, (uint32_t)(r0 >> 32), (uint32_t)r1, (uint32_t)(r1 >> 32)};
#pragma unroll
    for (int i = 0; i < 4; i++) {
        b[i] = q[i];
        c += (uint64_t)t[i] + q[i];
        a[i] = (uint32_t)c;
        c >>= 32;
    }
    a[4] = (uint32_t)c;
}
FF_INLINE void gls_split_g2(const uint32_t* __restrict__ k, uint64_t d[4]) {
    const uint64_t z = (uint64_t)ENDO_Z[0] | ((uint64_t)ENDO_Z[1] << 32);
    uint32_t cur[8];
#pragma unroll
    for (int i = 0; i < 8; i++) cur[i] = k[i];
#pragma unroll 1
    for (int it = 0; it < 3; it++) {
        uint64_t r = 0;
#pragma unroll 1
        for (int w = 7 - 2 * it; w >= 0; w--) {          // the dividend loses 64 bits per round
            const uint32_t bits = cur[w];
            uint32_t qw = 0;
#pragma unroll 1
            for (int bt = 31; bt >= 0; bt--) {
                const uint64_t top = r >> 63;
                r = (r << 1) | ((bits >> bt) & 1u);
                const bool ge = top || r >= z;
                if (ge) r -= z;
                qw = (qw << 1) | (ge ? 1u : 0u);
            }
            cur[w] = qw;
        }
        d[it] = r;
    }
    d[3] = (uint64_t)cur[0] | ((uint64_t)cur[1] << 32);
}
FF_INLINE FpB<1> endo_limbs(const uint32_t* __restrict__ c) {
    FpB<1> r;
#pragma unroll
    for (int i = 0; i < FPL; i++) r.v[i] = c[i];
    return r;
}
// the table entry under phi
FF_INLINE void endo_apply_g1(Xyzz<Fp>& q) { q.x = Fp(fe_mul(q.x, endo_limbs(ENDO_BETA))); }
// the table entry under (-1)^i psi^i, i = 1..3 (lane pair: this lane holds component pair_comp() of every coordinate)
FF_INLINE void endo_apply_g2(Xyzz<Fp2H>& q, int i) {
    const uint32_t comp = pair_comp();
    const bool flip = (i & 1) && comp;                   // conj^i negates the c1 component for odd i
    const Fp2HB<1> cx{endo_limbs(ENDO_PSI_X[i - 1][comp])}, cy{endo_limbs(ENDO_PSI_Y[i - 1][comp])};
    auto conj = [&](const Fp2H& a) -> Fp2HB<128> {
        const FpB<128> n = fe_neg(a.v), p = a.v;
        return {fp_select(flip, p, n)};
    };
    q.x = Fp2H(fe_mul(conj(q.x), cx));
    q.y = Fp2H(fe_mul(conj(q.y), cy));
    if (i & 1) {
        q.zz = Fp2H(fp_canon(conj(q.zz).v));
        q.zzz = Fp2H(fp_canon(conj(q.zzz).v));
    }
}
template <class T> FF_INLINE void window_table(const Xyzz<T>& p, uint8_t* __restrict__ tab) {
    constexpr int XB = RawLayout<T>::XYZZ;
    xyzz_store_raw<T>(tab + XB * 1, p);
    for (uint32_t d = 2; d < 16; d++) {
        Xyzz<T> q;
        if (d & 1) {
            q = xyzz_load_raw<T>(tab + XB * (d - 1));
            xyzz_add_impl(q, p);
        } else {
            q = xyzz_dbl_impl(xyzz_load_raw<T>(tab + XB * (d >> 1)));
        }
        xyzz_store_raw<T>(tab + XB * d, q);
    }
}
FF_INLINE Xyzz<Fp> xyzz_mul_scalar_endo(const Xyzz<Fp>& p, const uint32_t* __restrict__ k, uint8_t* __restrict__ tab) {
    constexpr int XB = RawLayout<Fp>::XYZZ;
    window_table<Fp>(p, tab);
    uint32_t a[5], b[4];
    glv_split_g1(k, a, b);
    Xyzz<Fp> acc = xyzz_inf<Fp>();
    bool started = false;
#pragma unroll 1
    for (int w = 32; w >= 0; w--) {
        const uint32_t da = (a[w >> 3] >> ((w & 7) * 4)) & 15u, db = w < 32 ? (b[w >> 3] >> ((w & 7) * 4)) & 15u : 0u;
        if (!started) {
            if (__ballot((da | db) != 0) == 0) continue;          // wave-uniform leading zero windows
        } else {
#pragma unroll 1
            for (int r = 0; r < 4; r++) acc = xyzz_dbl_impl(acc);
        }
        started = true;
        if (da) {
            const Xyzz<Fp> q = xyzz_load_raw<Fp>(tab + XB * da);
            xyzz_add_impl(acc, q);
        }
        if (db) {
            Xyzz<Fp> q = xyzz_load_raw<Fp>(tab + XB * db);
            endo_apply_g1(q);
            xyzz_add_impl(acc, q);
        }
    }
    return acc;
}
FF_INLINE Xyzz<Fp2H> xyzz_mul_scalar_endo(const Xyzz<Fp2H>& p, const uint32_t* __restrict__ k, uint8_t* __restrict__ tab) {
    constexpr int XB = RawLayout<Fp2H>::XYZZ;
    window_table<Fp2H>(p, tab);
    uint64_t d[4];
    gls_split_g2(k, d);
    Xyzz<Fp2H> acc = xyzz_inf<Fp2H>();
    bool started = false;
#pragma unroll 1
    for (int w = 15; w >= 0; w--) {
        uint32_t dg[4];
#pragma unroll
        for (int i = 0; i < 4; i++) dg[i] = (uint32_t)(d[i] >> (4 * w)) & 15u;
        if (!started) {
            if (__ballot((dg[0] | dg[1] | dg[2] | dg[3]) != 0) == 0) continue;
        } else {
#pragma unroll 1
            for (int r = 0; r < 4; r++) acc = xyzz_dbl_impl(acc);
        }
        started = true;
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            if (dg[i]) {                                          // pair-uniform: both lanes of a pair hold the same scalar
                Xyzz<Fp2H> q = xyzz_load_raw<Fp2H>(tab + XB * dg[i]);
                if (i) endo_apply_g2(q, i);
                xyzz_add_impl(acc, q);
            }
        }
    }
    return acc;
}
FF_INLINE Fp neg_coord(const Fp& y) { return Fp(fp_canon(fe_neg(y))); }
FF_INLINE Fp2H neg_coord(const Fp2H& y) { return Fp2H(fp_canon(fe_neg(y).v)); }

// ---- conversions between the key's dense affine points and the raw XYZZ working arrays
template <class T> __global__ void k_aff_to_raw(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t count, uint64_t total) {
    constexpr int AB = FieldOps<T>::WORDS * 8, XB = RawLayout<T>::XYZZ;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N;
    if (i >= total) return;
    Xyzz<T> q = xyzz_inf<T>();
    if (i < count) q = xyzz_from_aff(aff_load<T>(src + AB * i));
    xyzz_store_raw<T>(dst + XB * i, q);
}
template <class T> __global__ void k_raw_to_aff(uint8_t* __restrict__ dst, const uint8_t* __restrict__ src, uint64_t count) {
    constexpr int AB = FieldOps<T>::WORDS * 8, XB = RawLayout<T>::XYZZ;
    const uint64_t i = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N;
    if (i >= count) return;
    aff_store<T>(dst + AB * i, xyzz_to_aff(xyzz_load_raw<T>(src + XB * i)));
}

// ---- one radix-2 stage over the whole array (independent blocks of 2h points): forward = DIF (natural -> bit-reversed),
// inverse = DIT (bit-reversed -> natural, unscaled) -- the conventions of ntt.hip, so the Fr tables line up.  tw[h + j] = w_2h^(+-j).
// Two waves per SIMD for G1 (256 registers: 16-122 spilled ones cost less than the second wave brings -- 2^20: 44.4 -> 41.6 s for a key's three
// sets); the lane-pair G2 form needs 340-400 registers and spills hundreds at 256: one wave per SIMD there.
template <class T, bool INVERSE>
__global__ __launch_bounds__(128, (std::is_same<T, Fp>::value ? 2 : 1)) void k_gntt_stage(uint8_t* __restrict__ pts, const uint32_t* __restrict__ tw, uint32_t log_h, uint64_t b0, uint64_t pairs, uint8_t* __restrict__ scratch) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint64_t loc = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N, b = b0 + loc;      // a slab of butterflies per launch
    if (b >= pairs) return;
    uint8_t* tab = scratch + (uint64_t)16 * XB * loc;
    const uint64_t h = (uint64_t)1 << log_h, j = b & (h - 1), e = ((b >> log_h) << (log_h + 1)) | j;
    const uint32_t* w = tw + 8 * (h + j);
    // nothing but the multiplicand stays live across the scalar multiplication (a second point in registers there costs hundreds of spills)
    if (INVERSE) {
        Xyzz<T> v = xyzz_load_raw<T>(pts + XB * (e + h));
        if (log_h) v = xyzz_mul_scalar_endo(v, w, tab);      // span 2: the twiddle is 1 (wave-uniform test)
        Xyzz<T> u = xyzz_load_raw<T>(pts + XB * e);
        Xyzz<T> x = u;
        xyzz_add_impl(x, v);
        xyzz_store_raw<T>(pts + XB * e, x);
        v.y = neg_coord(v.y);
        xyzz_add_impl(u, v);
        xyzz_store_raw<T>(pts + XB * (e + h), u);
    } else {
        Xyzz<T> u = xyzz_load_raw<T>(pts + XB * e), v = xyzz_load_raw<T>(pts + XB * (e + h));
        {
            Xyzz<T> x = u;
            xyzz_add_impl(x, v);
            xyzz_store_raw<T>(pts + XB * e, x);
        }
        v.y = neg_coord(v.y);
        xyzz_add_impl(u, v);                                // u - v
        if (log_h) u = xyzz_mul_scalar_endo(u, w, tab);
        xyzz_store_raw<T>(pts + XB * (e + h), u);
    }
}
// pts[i] <- tab[i] * pts[i]
template <class T> __global__ __launch_bounds__(128, (std::is_same<T, Fp>::value ? 2 : 1)) void k_g_tabmul(uint8_t* __restrict__ pts, const uint32_t* __restrict__ tab, uint64_t i0, uint64_t total, uint8_t* __restrict__ scratch) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint64_t loc = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N, i = i0 + loc;
    if (i >= total) return;
    const Xyzz<T> p = xyzz_load_raw<T>(pts + XB * i);
    xyzz_store_raw<T>(pts + XB * i, xyzz_mul_scalar_endo(p, tab + 8 * i, scratch + (uint64_t)16 * XB * loc));
}
// scalar multiplications per launch: bounds the scratch of the window tables (16 multiples each) to 1 (G1) / 2 (G2) GiB
static constexpr uint64_t DERIVE_SLAB = (uint64_t)1 << 18;
template <class T> static int g_tabmul(uint8_t* pts, const uint32_t* tab, uint64_t total, uint8_t* scratch, hipStream_t s) {
    for (uint64_t i0 = 0; i0 < total; i0 += DERIVE_SLAB) {
        const uint64_t cnt = total - i0 < DERIVE_SLAB ? total - i0 : DERIVE_SLAB;
        hipLaunchKernelGGL(k_g_tabmul<T>, g1d(cnt * LaneCount<T>::N, 128), dim3(128), 0, s, pts, tab, i0, total, scratch);
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
// the transposed tree step of a level: arr[node * N + h + j] = work[node * N + j], j < h = N / 2 (the lower halves stay)
__global__ void k_g_upper_from(uint4* __restrict__ arr, const uint4* __restrict__ work, uint32_t log_N, uint64_t total_vec, uint32_t vec_per_point) {
    const uint64_t g = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total_vec) return;
    const uint64_t p = g / vec_per_point, N = (uint64_t)1 << log_N, h = N >> 1;
    if ((p & (N - 1)) < h) return;
    arr[g] = work[g - h * vec_per_point];
}

// ---- Fr-side tables (canonical integers: they multiply points, not field elements)
FF_INLINE uint32_t bitrev(uint32_t x, uint32_t bits) { return bits ? __brev(x) >> (32 - bits) : 0; }
// out[p] = canonical( tab[base + br(N - br(q))] * scale ),  p = base + q, q < N = 2^log_N: the table at the NEGATED frequency
// (a correlation is a product with the transform at the inverse roots), in the transform-domain order of the data
__global__ void k_tab_neg_canon(uint32_t* __restrict__ out, const uint32_t* __restrict__ tab, uint32_t log_N, uint64_t total, const uint32_t* __restrict__ scale) {
    const uint64_t p = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (p >= total) return;
    const uint32_t N = 1u << log_N, q = (uint32_t)(p & (N - 1));
    const uint32_t k = bitrev(q, log_N), kn = (N - k) & (N - 1), qn = bitrev(kn, log_N);
    Fr x = fe_load<FrParams>(tab + 8 * (p - q + qn));
    if (scale) x = fe_mul(x, fe_load<FrParams>(scale));
    fe_store<FrParams>(out + 8 * p, fe_from_mont(x));
}
// alt[j] = (-1)^j / j! for j < n2, 0 up to total (Montgomery)
__global__ void k_alt_kernel(uint32_t* __restrict__ out, const uint32_t* __restrict__ invfact, uint32_t n2, uint64_t total) {
    const uint64_t j = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= total) return;
    Fr x = fe_zero<FrParams>();
    if (j < n2) {
        x = fe_load<FrParams>(invfact + 8 * j);
        if (j & 1) x = fe_neg(x);
    }
    fe_store<FrParams>(out + 8 * j, x);
}
__global__ void k_inv_pow2_mont(uint32_t* out, uint32_t k) {   // 2^-k (Montgomery)
    if (blockIdx.x || threadIdx.x) return;
    Fr half = fe_inv(fe_from_u32<FrParams>(2)), acc = fe_one<FrParams>();
    for (uint32_t i = 0; i < k; i++) acc = fe_mul(acc, half);
    fe_store<FrParams>(out, acc);
}
// canonical twiddle heap: tw[2^(k-1) + j] = w_{2^k}^(+-j) (as k_gen_twiddles of ntt.hip, plain integers)
__device__ static const uint32_t LD_OMEGA_MONT[8] = {0x0c17f47cu, 0x9cab6d5cu, 0xfd4b71e5u, 0x1ce1e93du, 0x471dd505u, 0x0d6db230u, 0x743a3b6au, 0x3f0ee990u};
__device__ static const uint32_t LD_OMEGA_INV_MONT[8] = {0xb3082d19u, 0x55a9e082u, 0xc7dc4a13u, 0x082f90b2u, 0xc76b052cu, 0x76ce3accu, 0x6e54185du, 0x15c39d95u};
__global__ void k_twiddles_canon(uint32_t* tw, uint32_t log_k, int inverse) {
    const uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x, total = (uint64_t)1 << log_k;
    if (i >= total || i == 0) return;
    const uint32_t k = 64 - __builtin_clzll(i);
    const uint32_t j = (uint32_t)(i - ((uint64_t)1 << (k - 1)));
    const uint32_t e = j << (32 - k);
    Fr base, acc = fe_one<FrParams>();
#pragma unroll
    for (int l = 0; l < 8; l++) base.v[l] = inverse ? LD_OMEGA_INV_MONT[l] : LD_OMEGA_MONT[l];
    for (int b = 0; b < 32; b++) {
        if ((e >> b) & 1) acc = fe_mul(acc, base);
        base = fe_sqr(base);
    }
    fe_store<FrParams>(tw + 8 * i, fe_from_mont(acc));
}

// ---- the derivation of one set of bases
struct DeriveTables {
    uint32_t n2 = 0, log_n2 = 0;
    DevBuf tree;          // log_n2 levels x n2 canonical scalars, negated frequency, level l (node size 2^l) at (l - 1) * n2
    DevBuf alt;           // 2 n2 canonical scalars: NTT of alt at the negated frequency, times 1 / (2 n2)
    DevBuf invfact;       // n2 canonical scalars 1 / i!
    DevBuf tw_f, tw_i;    // canonical twiddle heaps up to 2 n2
};
static int derive_tables_build(DeriveTables& t, const FrStage& f, uint32_t offset, hipStream_t s) {
    const uint32_t n2 = f.n2, lg = f.log_n2, S = 2 * n2, lgS = lg + 1;
    t.n2 = n2; t.log_n2 = lg;
    ZKCHK(t.tree.alloc(32 * (size_t)n2 * (lg ? lg : 1)));
    ZKCHK(t.alt.alloc(32 * (size_t)S));
    ZKCHK(t.invfact.alloc(32 * (size_t)n2));
    ZKCHK(t.tw_f.alloc(32 * (size_t)S));
    ZKCHK(t.tw_i.alloc(32 * (size_t)S));
    {
        DevBuf mont, q;
        ZKCHK(mont.alloc(32 * (size_t)n2 * (lg ? lg : 1)));
        ZKCHK(frstage_tree_tables(n2, lg, offset, mont.p, q, s));
        for (uint32_t l = 1; l <= lg; l++)
            hipLaunchKernelGGL(k_tab_neg_canon, g1d(n2), dim3(256), 0, s, t.tree.as<uint32_t>() + 8 * (uint64_t)(l - 1) * n2,
                               (const uint32_t*)(mont.as<uint32_t>() + 8 * (uint64_t)(l - 1) * n2), l, (uint64_t)n2, (const uint32_t*)nullptr);
        HIPCHK(hipStreamSynchronize(s));
    }
    {
        DevBuf a, sc;
        ZKCHK(a.alloc(32 * (size_t)S));
        ZKCHK(sc.alloc(32));
        hipLaunchKernelGGL(k_alt_kernel, g1d(S), dim3(256), 0, s, a.as<uint32_t>(), (const uint32_t*)f.invfact.as<uint32_t>(), n2, (uint64_t)S);
        ZKCHK(ntt_forward(a.p, S, lgS, s));
        hipLaunchKernelGGL(k_inv_pow2_mont, dim3(1), dim3(64), 0, s, sc.as<uint32_t>(), lgS);
        hipLaunchKernelGGL(k_tab_neg_canon, g1d(S), dim3(256), 0, s, t.alt.as<uint32_t>(), (const uint32_t*)a.as<uint32_t>(), lgS, (uint64_t)S, (const uint32_t*)sc.as<uint32_t>());
        ZKCHK(fr_from_mont(t.invfact.p, f.invfact.p, n2, s));
        hipLaunchKernelGGL(k_twiddles_canon, g1d(S), dim3(256), 0, s, t.tw_f.as<uint32_t>(), lgS, 0);
        hipLaunchKernelGGL(k_twiddles_canon, g1d(S), dim3(256), 0, s, t.tw_i.as<uint32_t>(), lgS, 1);
        HIPCHK(hipGetLastError());
        HIPCHK(hipStreamSynchronize(s));
    }
    return ZK_OK;
}

template <class T> static int gntt(uint8_t* pts, uint64_t total, uint32_t log_len, bool inverse, const DeriveTables& t, uint8_t* scratch, hipStream_t s) {
    const uint64_t pairs = total / 2;
    for (uint32_t st = 0; st < log_len; st++) {
        const uint32_t log_h = inverse ? st : log_len - 1 - st;
        for (uint64_t b0 = 0; b0 < pairs; b0 += DERIVE_SLAB) {
            const uint64_t cnt = pairs - b0 < DERIVE_SLAB ? pairs - b0 : DERIVE_SLAB;
            const dim3 g = g1d(cnt * LaneCount<T>::N, 128);
            if (inverse) hipLaunchKernelGGL((k_gntt_stage<T, true>), g, dim3(128), 0, s, pts, (const uint32_t*)t.tw_i.as<uint32_t>(), log_h, b0, pairs, scratch);
            else hipLaunchKernelGGL((k_gntt_stage<T, false>), g, dim3(128), 0, s, pts, (const uint32_t*)t.tw_f.as<uint32_t>(), log_h, b0, pairs, scratch);
        }
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
// src: n dense affine points [x^k], k < n (device);  dst: n dense affine points, the Lagrange-form bases of the tables' points
template <class T> static int derive_set(const DeriveTables& t, const uint8_t* d_src, uint32_t n, uint8_t* d_dst, hipStream_t s) {
    constexpr size_t XB = RawLayout<T>::XYZZ;
    constexpr uint32_t LP = LaneCount<T>::N;
    const uint32_t n2 = t.n2, S = 2 * n2;
    DevBuf A, W, scr;
    ZKCHK(A.alloc(XB * S));
    ZKCHK(W.alloc(XB * (size_t)n2));
    ZKCHK(scr.alloc((size_t)16 * XB * (S < DERIVE_SLAB ? S : DERIVE_SLAB)));      // one table of 16 multiples per scalar multiplication of a launch
    uint8_t* sc = scr.as<uint8_t>();
    hipLaunchKernelGGL(k_aff_to_raw<T>, g1d((uint64_t)n2 * LP), dim3(256), 0, s, A.as<uint8_t>(), d_src, (uint64_t)n, (uint64_t)n2);
    // T^T: top level first
    for (uint32_t l = t.log_n2; l >= 1; l--) {
        HIPCHK(hipMemcpyAsync(W.p, A.p, XB * (size_t)n2, hipMemcpyDeviceToDevice, s));
        ZKCHK(gntt<T>(W.as<uint8_t>(), n2, l, false, t, sc, s));
        ZKCHK(g_tabmul<T>(W.as<uint8_t>(), (const uint32_t*)(t.tree.as<uint32_t>() + 8 * (uint64_t)(l - 1) * n2), (uint64_t)n2, sc, s));
        ZKCHK(gntt<T>(W.as<uint8_t>(), n2, l, true, t, sc, s));
        const uint32_t vpp = (uint32_t)(XB / 16);
        hipLaunchKernelGGL(k_g_upper_from, g1d((uint64_t)n2 * vpp), dim3(256), 0, s, A.as<uint4>(), (const uint4*)W.as<uint4>(), l, (uint64_t)n2 * vpp, vpp);
    }
    // the n-point problem is the leading block: entries from n on drop out; zero padding to the convolution size
    HIPCHK(hipMemsetAsync(A.as<uint8_t>() + XB * (size_t)n, 0, XB * (size_t)(S - n), s));
    // Conv_alt^T: correlation with alt through a cyclic transform of size 2 n2
    ZKCHK(gntt<T>(A.as<uint8_t>(), S, t.log_n2 + 1, false, t, sc, s));
    ZKCHK(g_tabmul<T>(A.as<uint8_t>(), (const uint32_t*)t.alt.as<uint32_t>(), (uint64_t)S, sc, s));
    ZKCHK(gntt<T>(A.as<uint8_t>(), S, t.log_n2 + 1, true, t, sc, s));
    // D(1/i!)
    ZKCHK(g_tabmul<T>(A.as<uint8_t>(), (const uint32_t*)t.invfact.as<uint32_t>(), (uint64_t)n, sc, s));
    hipLaunchKernelGGL(k_raw_to_aff<T>, g1d((uint64_t)n * LP, 128), dim3(128), 0, s, d_dst, (const uint8_t*)A.as<uint8_t>(), (uint64_t)n);
    HIPCHK(hipGetLastError());
    HIPCHK(hipStreamSynchronize(s));
    return ZK_OK;
}

// Pools of a key in the reference's layout (device, dense affine)  g1 = a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid,  g2 = b2 | d2 | ti2[n+2]
// -> the Lagrange-form pools  g1' = a | d1 | b1 | [l_i]_1 (n) | [lambda_t Z/delta]_1 (n-1) | ltd_mid,  g2' = b2 | d2 | [l_i]_2 (n)  (device, dense affine).
// The three derived sets are independent of one another: `sets` selects them (bit 0: [l_i]_1, bit 1: [l_i]_2, bit 2: the h bases) -- N ranks of
// a node derive one set each and broadcast it instead of N redundant derivations (zk_groth16_pk_derive_lagrange_sets).  The copied parts
// (a | d1 | b1, ltd_mid, b2 | d2) are always written; a set that is not selected leaves its region of the output untouched.
int groth16_derive_lagrange_pools(const FrStage& f, const uint8_t* d_g1, uint64_t n_mid, const uint8_t* d_g2, uint8_t* out_g1, uint8_t* out_g2, uint32_t sets, hipStream_t s) {
    const uint32_t n = f.n;
    const uint64_t o_ti = 3, o_tz = 3 + ((uint64_t)n + 2), o_lt = o_tz + (n - 1);
    HIPCHK(hipMemcpyAsync(out_g1, d_g1, 96 * 3, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(out_g1 + 96 * (3 + (uint64_t)n + (n - 1)), d_g1 + 96 * o_lt, 96 * n_mid, hipMemcpyDeviceToDevice, s));
    HIPCHK(hipMemcpyAsync(out_g2, d_g2, 192 * 2, hipMemcpyDeviceToDevice, s));
    if (n == 1) {      // one gate: l_0 = 1, the Lagrange point IS [tau^0]
        if (sets & 1) HIPCHK(hipMemcpyAsync(out_g1 + 96 * 3, d_g1 + 96 * o_ti, 96, hipMemcpyDeviceToDevice, s));
        if (sets & 2) HIPCHK(hipMemcpyAsync(out_g2 + 192 * 2, d_g2 + 192 * 2, 192, hipMemcpyDeviceToDevice, s));
        HIPCHK(hipStreamSynchronize(s));
        return ZK_OK;
    }
    if (sets & 3) {
        ScopedTimer tm("lagrange_derive", s);
        DeriveTables t0;
        ZKCHK(derive_tables_build(t0, f, 0, s));
        if (sets & 1) ZKCHK(derive_set<Fp>(t0, d_g1 + 96 * o_ti, n, out_g1 + 96 * 3, s));
        if (sets & 2) ZKCHK(derive_set<Fp2H>(t0, d_g2 + 192 * 2, n, out_g2 + 192 * 2, s));
    }
    if (sets & 4) {
        DeriveTables tn;                       // the points n .. 2n-2 of the h values
        ZKCHK(derive_tables_build(tn, f, n, s));
        ZKCHK(derive_set<Fp>(tn, d_g1 + 96 * o_tz, n - 1, out_g1 + 96 * (3 + (uint64_t)n), s));
    }
    HIPCHK(hipStreamSynchronize(s));
    return ZK_OK;
}

// Pinocchio (src/pinocchio/pinocchio.ml:450: h against the powers si): d_si holds [s^k]_1, k < n - 1 (dense affine, device);
// d_out receives [lambda_t(s)]_1, t < n - 1, lambda_t the Lagrange basis of the points n .. 2n-2 -- the bases the h VALUES multiply.
int derive_shifted_bases_g1(const FrStage& f, const uint8_t* d_si, uint8_t* d_out, hipStream_t s) {
    if (f.n < 2) return ZK_OK;
    ScopedTimer tm("lagrange_derive", s);
    DeriveTables tn;
    ZKCHK(derive_tables_build(tn, f, f.n, s));
    return derive_set<Fp>(tn, d_si, f.n - 1, d_out, s);
}

}  // namespace zk
