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
// Waves per SIMD of the lane-pair (G2) butterflies and table products.  Round 3 ran them at ONE wave (340-400 registers, no spill); with the Jacobian /
// affine-table multiplication of round 4 the table product fits 256 registers with 20 spilled and the butterflies spill 146 / 227 -- outside the
// multiplication's loop -- and TWO waves are faster: 7.39 -> 6.70 s at 2^18, 34.8 -> 31.2 s at 2^20 for a key's three sets (same box, alternating).
#ifndef ZK_DERIVE_G2_WAVES
#define ZK_DERIVE_G2_WAVES 2
#endif
template <class T> struct LaneCount { static constexpr uint32_t N = RawLayout<T>::LANES; };

// ---- scalar (canonical Fr, 8 words in memory) times point, complete formulas (identity, equal operands).
// History of the per-multiplication cost (field products; a key's three sets at 2^16 / 2^18 / 2^20): bitwise ladder 5 600; one 255-bit scalar in 64
// fixed 4-bit windows over a per-lane table of 1 P .. 15 P in device memory 3 300 (4.2 s / 15.8 s / 77 s); split through the endomorphisms 2 300 on G1,
// 1 700 Fp2 products on G2 (2.7 s / 9.5 s / 45 s, round 2; 41 s at 2^20 in round 3); Jacobian accumulator + affine table + signed windows (round 4, below)
// ~-16 % instructions: 34 s at 2^20, 160 s at 2^22.
// The scalar is split through the curve's endomorphisms (Gallant-Lambert-Vanstone on G1, Galbraith-Lin-Scott on G2): the doublings are
// what a 255-bit scalar costs, and they are shared between the sub-scalars of
//   G1:  k = q z^2 + t  =>  k P = (t + q) P + q phi(P),   phi(x, y) = (beta x, y) = [z^2 - 1] (x, y)          2 scalars of <= 129 bits
//   G2:  k = sum_i k_i |z|^i  =>  k P = sum_i (-1)^i k_i psi^i(P),   psi(x, y) = (cx conj x, cy conj y) = [z] (x, y)      4 scalars of 64 bits
// (z = -0xd201000000010000; r = z^4 - z^2 + 1, p = z mod r).  ONE table of multiples of P; the image of an entry under phi / psi^i costs one / two
// products by constants (scripts/gen_endo_consts.py derives and CHECKS them against first-principles arithmetic).  The split is a bitwise long
// division per lane (~600 steps of a few integer instructions: the cost of a handful of field products).
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
// ---- round 4: the scalar multiplication itself.  Every butterfly of the transforms in the exponent is ONE of these, and the kernels already run at
// ~90 % of the chip's field-multiplier rate (29 M G1 multiplications/s x 2 300 products): only fewer products per multiplication make the derivation
// shorter.  Three changes, same group elements:
//   * the accumulator is JACOBIAN (X, Y, Z) instead of XYZZ: a doubling is 3 M + 4 S (6.1 product-equivalents) instead of 4 M + 3 S + a fused
//     double product (7.8) -- and doublings are what a 129-bit half-scalar is made of;
//   * the window table is AFFINE: the multiples 1 P .. 8 P are built in Jacobian coordinates (4 doublings + 3 general additions), their Z are inverted
//     together (Montgomery's trick: 21 products + ONE lockstep inversion, ~64 product-equivalents, fp_inv.cuh) and every window addition is a MIXED
//     one -- 8 M + 3 S instead of the 12 M + 2 S of two projective points;
//   * SIGNED 4-bit windows: k + 0x88..8 read nibble by nibble gives digits m - 8 in [-8, 7], so eight table entries serve sixteen digit values (the
//     table costs half as much to build and to invert) and -P is a conditional negation of y.
// Per G1 multiplication ~1 630 products instead of ~2 300; per G2 multiplication (lane pairs: a product 1.5, a square 1.0) ~1 800 instead of ~2 330.
template <class T> struct Jac {
    T x, y, z;          // z = 0: the identity (x, y arbitrary field elements)
};
// dbl-2009-l for a = 0 with 8 Y^4 taken as 2 (2 Y^2)^2 (keeps every bound inside the at-rest type for the lane-pair field too).  Complete: the
// identity (Z = 0) and a point of order two (Y = 0) both give Z3 = 0.
template <class T> FF_INLINE Jac<T> jac_dbl(const Jac<T>& p) {
    const auto A = fe_sqr(p.x);
    const auto B2 = fe_dbl(fe_sqr(p.y));                      // 2 Y^2
    const auto C8 = fe_dbl(fe_sqr(B2));                       // 8 Y^4
    const auto D = fe_dbl(fe_mul(p.x, B2));                   // 4 X Y^2
    const auto E = fe_add(fe_dbl(A), A);                      // 3 X^2
    const auto X3 = fe_sub(fe_sqr(E), fe_dbl(D));
    const auto Y3 = fe_sub(fe_mul(E, fe_sub(D, X3)), C8);
    const auto Z3 = fe_dbl(fe_mul(p.y, p.z));
    return {T(X3), T(Y3), T(Z3)};
}
// acc += (qx, qy), an AFFINE point that is not the identity (madd-2004-hmv shape: 8 M + 3 S).  Complete: identity accumulator, equal and opposite points.
template <class T, class QX, class QY> FF_INLINE void jac_madd(Jac<T>& acc, const QX& qx, const QY& qy) {
    if (fe_is_zero(acc.z)) {
        acc = {T(qx), T(qy), T(FieldOps<T>::one())};
        return;
    }
    const auto Z2 = fe_sqr(acc.z);
    const auto U2 = fe_mul(qx, Z2);
    const auto S2 = fe_mul(fe_mul(qy, acc.z), Z2);
    const auto H = fe_sub(U2, acc.x);
    const auto R = fe_sub(S2, acc.y);
    if (fe_is_zero(H)) {                                      // same x: the same point (double it) or its negative (identity)
        if (fe_is_zero(R)) acc = jac_dbl(acc);
        else acc.z = T(FieldOps<T>::zero());
        return;
    }
    const auto HH = fe_sqr(H);
    const auto HHH = fe_mul(H, HH);
    const auto V = fe_mul(acc.x, HH);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), HHH, V);        // R^2 - H^3 - 2 V
    const auto Y3 = fe_mul_sub(R, fe_sub(V, X3), acc.y, HHH);
    acc.z = T(fe_mul(acc.z, H));
    acc.x = T(X3);
    acc.y = T(Y3);
}
// acc += q, both Jacobian (the three general additions of the table build: 12 M + 4 S).  Complete.
template <class T> FF_INLINE void jac_add(Jac<T>& acc, const Jac<T>& q) {
    if (fe_is_zero(q.z)) return;
    if (fe_is_zero(acc.z)) {
        acc = q;
        return;
    }
    const auto Z1Z1 = fe_sqr(acc.z), Z2Z2 = fe_sqr(q.z);
    const auto U1 = fe_mul(acc.x, Z2Z2), U2 = fe_mul(q.x, Z1Z1);
    const auto S1 = fe_mul(fe_mul(acc.y, q.z), Z2Z2), S2 = fe_mul(fe_mul(q.y, acc.z), Z1Z1);
    const auto H = fe_sub(U2, U1);
    const auto R = fe_sub(S2, S1);
    if (fe_is_zero(H)) {
        if (fe_is_zero(R)) acc = jac_dbl(acc);
        else acc.z = T(FieldOps<T>::zero());
        return;
    }
    const auto HH = fe_sqr(H);
    const auto HHH = fe_mul(H, HH);
    const auto V = fe_mul(U1, HH);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), HHH, V);
    const auto Y3 = fe_mul_sub(R, fe_sub(V, X3), S1, HHH);
    acc.z = T(fe_mul(fe_mul(acc.z, q.z), H));
    acc.x = T(X3);
    acc.y = T(Y3);
}
// Per-multiplication scratch (one lane for G1, one lane PAIR for G2; 16 XYZZ points = 64 raw field elements were reserved per multiplication):
//   [0, 16 E)   the affine table: entry d = 1..8 as x | y at (d - 1) * 2 E
//   [16 E, 40 E) the Jacobian multiples d P as x | y | z at 16 E + (d - 1) * 3 E, while the table is being built
//   [40 E, 48 E) the running products Z_1 .. Z_1 Z_2 .. Z_d of the shared inversion
template <class T> struct TabLayout {
    static constexpr int E = RawLayout<T>::ELEM;
    static constexpr int AFF = 0, JAC = 16 * E, PRE = 40 * E;
    static_assert(48 * E <= 16 * RawLayout<T>::XYZZ, "window-table scratch outgrew its reservation");
};
template <class T> FF_INLINE T ld_raw(const uint8_t* p) { return load_raw_f((const T*)nullptr, p); }
// the multiples 1 P .. 8 P of a point that is NOT the identity, as affine points in the lane's scratch
template <class T> FF_INLINE void window_table_affine(const Xyzz<T>& p, uint8_t* __restrict__ tab) {
    using L = TabLayout<T>;
    constexpr int E = L::E;
    uint8_t* jac = tab + L::JAC;
    uint8_t* pre = tab + L::PRE;
    auto put = [&](uint32_t d, const Jac<T>& q) {
        store_raw_f(jac + (d - 1) * 3 * E, q.x);
        store_raw_f(jac + (d - 1) * 3 * E + E, q.y);
        store_raw_f(jac + (d - 1) * 3 * E + 2 * E, q.z);
    };
    auto get = [&](uint32_t d) -> Jac<T> { return {ld_raw<T>(jac + (d - 1) * 3 * E), ld_raw<T>(jac + (d - 1) * 3 * E + E), ld_raw<T>(jac + (d - 1) * 3 * E + 2 * E)}; };
    // (X, Y, ZZ, ZZZ) is the Jacobian point (X ZZ, Y ZZZ, ZZ): x = X ZZ / ZZ^2, y = Y ZZZ / ZZ^3 with ZZ^3 = ZZZ^2
    const Jac<T> p1{T(fe_mul(p.x, p.zz)), T(fe_mul(p.y, p.zzz)), p.zz};
    put(1, p1);
#pragma unroll 1
    for (uint32_t d = 2; d <= 8; d++) {
        Jac<T> q;
        if (d & 1) {
            q = get(d - 1);
            jac_add(q, p1);
        } else {
            q = jac_dbl(get(d >> 1));
        }
        put(d, q);
    }
    // running products of the Z, one inversion, back substitution: zi = 1 / Z_d
    {
        T run = ld_raw<T>(jac + 2 * E);
        store_raw_f(pre, run);
#pragma unroll 1
        for (uint32_t d = 2; d <= 8; d++) {
            run = T(fe_mul(run, ld_raw<T>(jac + (d - 1) * 3 * E + 2 * E)));
            store_raw_f(pre + (d - 1) * E, run);
        }
        T inv = T(fe_inv_fast(run));
#pragma unroll 1
        for (uint32_t d = 8; d >= 1; d--) {
            const T z = ld_raw<T>(jac + (d - 1) * 3 * E + 2 * E);
            T zi = inv;
            if (d > 1) {
                zi = T(fe_mul(inv, ld_raw<T>(pre + (d - 2) * E)));
                inv = T(fe_mul(inv, z));
            }
            const auto zi2 = fe_sqr(zi);
            const auto x = fe_mul(ld_raw<T>(jac + (d - 1) * 3 * E), zi2);
            const auto y = fe_mul(fe_mul(ld_raw<T>(jac + (d - 1) * 3 * E + E), zi), zi2);
            store_raw_f(tab + L::AFF + (d - 1) * 2 * E, T(x));
            store_raw_f(tab + L::AFF + (d - 1) * 2 * E + E, T(y));
        }
    }
}
// -P as a conditional negation of y (both branches in one type)
FF_INLINE FpB<4> cond_neg(bool neg, const FpB<2>& y) { return fp_select(neg, FpB<4>(y), FpB<4>(fe_neg(y))); }
FF_INLINE Fp2HB<4> cond_neg(bool neg, const Fp2HB<2>& y) { return {cond_neg(neg, y.v)}; }
template <class T> FF_INLINE Xyzz<T> jac_to_xyzz(const Jac<T>& a) {
    const auto zz = fe_sqr(a.z);
    return {a.x, a.y, T(zz), T(fe_mul(a.z, zz))};
}
FF_INLINE Xyzz<Fp> xyzz_mul_scalar_endo(const Xyzz<Fp>& p, const uint32_t* __restrict__ k, uint8_t* __restrict__ tab) {
    using L = TabLayout<Fp>;
    constexpr int E = L::E;
    if (xyzz_is_inf(p)) return xyzz_inf<Fp>();
    window_table_affine<Fp>(p, tab);
    uint32_t a[5], b[5];
    glv_split_g1(k, a, b);
    b[4] = 0;
    {   // signed digits: nibble w of k + 0x8..8 (33 nibbles) minus 8
        uint64_t ca = 0, cb = 0;
#pragma unroll
        for (int i = 0; i < 5; i++) {
            const uint32_t add = i < 4 ? 0x88888888u : 0x8u;
            ca += (uint64_t)a[i] + add;
            a[i] = (uint32_t)ca;
            ca >>= 32;
            cb += (uint64_t)b[i] + add;
            b[i] = (uint32_t)cb;
            cb >>= 32;
        }
    }
    Jac<Fp> acc;
    acc.x = acc.y = acc.z = Fp(fp_zero());
#pragma unroll 1
    for (int w = 32; w >= 0; w--) {
        if (w != 32) {
#pragma unroll 1
            for (int r = 0; r < 4; r++) acc = jac_dbl(acc);
        }
        const int da = (int)((a[w >> 3] >> ((w & 7) * 4)) & 15u) - 8, db = (int)((b[w >> 3] >> ((w & 7) * 4)) & 15u) - 8;
        if (da) {
            const uint8_t* e = tab + L::AFF + ((da < 0 ? -da : da) - 1) * 2 * E;
            const FpB<2> x = fp_assume<2>(ld_raw<Fp>(e)), y = fp_assume<2>(ld_raw<Fp>(e + E));
            jac_madd(acc, x, cond_neg(da < 0, y));
        }
        if (db) {                                             // the same entry under phi: (beta x, y)
            const uint8_t* e = tab + L::AFF + ((db < 0 ? -db : db) - 1) * 2 * E;
            const FpB<2> x = fe_mul(fp_assume<2>(ld_raw<Fp>(e)), endo_limbs(ENDO_BETA)), y = fp_assume<2>(ld_raw<Fp>(e + E));
            jac_madd(acc, x, cond_neg(db < 0, y));
        }
    }
    return jac_to_xyzz(acc);
}
// the affine table entry under (-1)^i psi^i, i = 1..3 (lane pair: this lane holds component pair_comp() of each coordinate)
FF_INLINE void endo_apply_g2_aff(Fp2HB<2>& x, Fp2HB<2>& y, int i) {
    const uint32_t comp = pair_comp();
    const bool flip = (i & 1) && comp;                   // conj^i negates the c1 component for odd i
    const Fp2HB<1> cx{endo_limbs(ENDO_PSI_X[i - 1][comp])}, cy{endo_limbs(ENDO_PSI_Y[i - 1][comp])};
    auto conj = [&](const Fp2HB<2>& a) -> Fp2HB<4> { return {fp_select(flip, FpB<4>(a.v), FpB<4>(fe_neg(a.v)))}; };
    x = fe_mul(conj(x), cx);
    y = fe_mul(conj(y), cy);
}
FF_INLINE Xyzz<Fp2H> xyzz_mul_scalar_endo(const Xyzz<Fp2H>& p, const uint32_t* __restrict__ k, uint8_t* __restrict__ tab) {
    using L = TabLayout<Fp2H>;
    constexpr int E = L::E;
    if (xyzz_is_inf(p)) return xyzz_inf<Fp2H>();
    window_table_affine<Fp2H>(p, tab);
    uint64_t d[4];
    gls_split_g2(k, d);
    uint32_t top = 0;                                         // nibble 16 of d[i] + 0x8..8 (17 nibbles), 4 bits per sub-scalar
#pragma unroll
    for (int i = 0; i < 4; i++) {
        const uint64_t lo = d[i] + 0x8888888888888888ull;
        top |= (8u + (lo < d[i] ? 1u : 0u)) << (4 * i);
        d[i] = lo;
    }
    Jac<Fp2H> acc;
    acc.x = acc.y = acc.z = Fp2H(fp_zero());
#pragma unroll 1
    for (int w = 16; w >= 0; w--) {
        if (w != 16) {
#pragma unroll 1
            for (int r = 0; r < 4; r++) acc = jac_dbl(acc);
        }
#pragma unroll 1
        for (int i = 0; i < 4; i++) {
            const int dg = (int)(w == 16 ? (top >> (4 * i)) & 15u : (uint32_t)(d[i] >> (4 * w)) & 15u) - 8;
            if (dg) {                                          // pair-uniform: both lanes of a pair hold the same scalar
                const uint8_t* e = tab + L::AFF + ((dg < 0 ? -dg : dg) - 1) * 2 * E;
                Fp2HB<2> x = fp_assume<2>(ld_raw<Fp2H>(e)), y = fp_assume<2>(ld_raw<Fp2H>(e + E));
                if (i) endo_apply_g2_aff(x, y, i);
                jac_madd(acc, x, cond_neg(dg < 0, y));
            }
        }
    }
    return jac_to_xyzz(acc);
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
// Two waves per SIMD for both curves (G1: 256 registers, 0 / 69 spilled; G2: see ZK_DERIVE_G2_WAVES above).
template <class T, bool INVERSE>
__global__ __launch_bounds__(128, (std::is_same<T, Fp>::value ? 2 : ZK_DERIVE_G2_WAVES)) void k_gntt_stage(uint8_t* __restrict__ pts, const uint32_t* __restrict__ tw, uint32_t log_h, uint64_t b0, uint64_t pairs, uint32_t log_pairs, uint8_t* __restrict__ scratch) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint64_t loc = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N, b = b0 + loc;      // a slab of butterflies per launch
    if (b >= pairs) return;
    uint8_t* tab = scratch + (uint64_t)16 * XB * loc;
    // butterfly b = (j, q) in j-MAJOR order (q = the block of 2h points, q < pairs / h): the butterflies with twiddle w^0 = 1 -- one in h, a tenth of a
    // derivation's scalar multiplications over the levels of the tree -- then fill whole waves, which skip the multiplication (a wave-uniform test;
    // in block-major order they sat in every h-th lane and saved nothing).  These kernels are bound by their arithmetic: the stride costs nothing.
    const uint64_t h = (uint64_t)1 << log_h, nq = pairs >> log_h, q = b & (nq - 1), j = b >> (log_pairs - log_h), e = (q << (log_h + 1)) | j;
    const uint32_t* w = tw + 8 * (h + j);
    // nothing but the multiplicand stays live across the scalar multiplication (a second point in registers there costs hundreds of spills)
    if (INVERSE) {
        Xyzz<T> v = xyzz_load_raw<T>(pts + XB * (e + h));
        if (j) v = xyzz_mul_scalar_endo(v, w, tab);          // j = 0: the twiddle is 1
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
        if (j) u = xyzz_mul_scalar_endo(u, w, tab);
        xyzz_store_raw<T>(pts + XB * (e + h), u);
    }
}
// pts[i] <- tab[i] * pts[i]
template <class T> __global__ __launch_bounds__(128, (std::is_same<T, Fp>::value ? 2 : ZK_DERIVE_G2_WAVES)) void k_g_tabmul(uint8_t* __restrict__ pts, const uint32_t* __restrict__ tab, uint64_t i0, uint64_t total, uint8_t* __restrict__ scratch) {
    constexpr int XB = RawLayout<T>::XYZZ;
    const uint64_t loc = ((uint64_t)blockIdx.x * blockDim.x + threadIdx.x) / LaneCount<T>::N, i = i0 + loc;
    if (i >= total) return;
    const Xyzz<T> p = xyzz_load_raw<T>(pts + XB * i);
    xyzz_store_raw<T>(pts + XB * i, xyzz_mul_scalar_endo(p, tab + 8 * i, scratch + (uint64_t)16 * XB * loc));
}
// scalar multiplications per launch: bounds the scratch of the window tables (16 multiples each) to 1 (G1) / 2 (G2) GiB
static constexpr uint64_t DERIVE_SLAB = (uint64_t)1 << 18;
static constexpr uint32_t DERIVE_SIDE_BY_SIDE_MAX_LOG = 22;          // measured: see groth16_derive_lagrange_pools
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
    if (log_len == 0) return ZK_OK;
    if (pairs == 0 || (pairs & (pairs - 1))) ZK_FAIL(ZK_ERR_ARG, "derive: a transform array must hold a power of two of points");
    uint32_t log_pairs = 0;
    while (((uint64_t)1 << log_pairs) < pairs) log_pairs++;
    for (uint32_t st = 0; st < log_len; st++) {
        const uint32_t log_h = inverse ? st : log_len - 1 - st;
        for (uint64_t b0 = 0; b0 < pairs; b0 += DERIVE_SLAB) {
            const uint64_t cnt = pairs - b0 < DERIVE_SLAB ? pairs - b0 : DERIVE_SLAB;
            const dim3 g = g1d(cnt * LaneCount<T>::N, 128);
            if (inverse) hipLaunchKernelGGL((k_gntt_stage<T, true>), g, dim3(128), 0, s, pts, (const uint32_t*)t.tw_i.as<uint32_t>(), log_h, b0, pairs, log_pairs, scratch);
            else hipLaunchKernelGGL((k_gntt_stage<T, false>), g, dim3(128), 0, s, pts, (const uint32_t*)t.tw_f.as<uint32_t>(), log_h, b0, pairs, log_pairs, scratch);
        }
    }
    HIPCHK(hipGetLastError());
    return ZK_OK;
}
// src: n dense affine points [x^k], k < n (device);  dst: n dense affine points, the Lagrange-form bases of the tables' points
// working arrays of one set: they outlive the enqueue (the sets of a key run side by side on their own streams)
struct DeriveWork {
    DevBuf A, W, scr;
};
template <class T> static int derive_set_enqueue(DeriveWork& wk, const DeriveTables& t, const uint8_t* d_src, uint32_t n, uint8_t* d_dst, hipStream_t s) {
    constexpr size_t XB = RawLayout<T>::XYZZ;
    constexpr uint32_t LP = LaneCount<T>::N;
    const uint32_t n2 = t.n2, S = 2 * n2;
    DevBuf &A = wk.A, &W = wk.W, &scr = wk.scr;
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
    return ZK_OK;
}
template <class T> static int derive_set(const DeriveTables& t, const uint8_t* d_src, uint32_t n, uint8_t* d_dst, hipStream_t s) {
    DeriveWork wk;
    ZKCHK(derive_set_enqueue<T>(wk, t, d_src, n, d_dst, s));
    HIPCHK(hipStreamSynchronize(s));
    return ZK_OK;
}
// streams of the sets that run beside the caller's
struct SideStreams {
    hipStream_t st[2] = {nullptr, nullptr};
    hipEvent_t ev[3] = {nullptr, nullptr, nullptr};
    ~SideStreams() {
        for (hipStream_t x : st) if (x) { (void)hipStreamSynchronize(x); (void)hipStreamDestroy(x); }
        for (hipEvent_t e : ev) if (e) (void)hipEventDestroy(e);
    }
};

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
    // The sets are independent, and a stage of one set is n2 / 2 scalar multiplications of ~2 ms (G1) / ~4 ms (G2) each: below 2^19 gates that is fewer
    // waves than the chip has SIMDs (2^16: 512 of 1024 for a G1 set), and a stage takes one multiplication's latency however few waves it has.  Up to
    // DERIVE_SIDE_BY_SIDE_MAX_LOG the selected sets therefore run side by side on their own streams (ZK_DERIVE_SIDE_BY_SIDE=0 / 1 overrides).
    const char* e_sbs = ::zk::opt("ZK_DERIVE_SIDE_BY_SIDE");          // per derivation, not per proof: read every time (tests run both orders in one process)
    const bool several = (sets & (sets - 1)) != 0;
    const bool side_by_side = several && (e_sbs ? atoi(e_sbs) != 0 : f.log_n2 <= DERIVE_SIDE_BY_SIDE_MAX_LOG);
    ScopedTimer tm("lagrange_derive", s);
    DeriveTables t0, tn;
    if (sets & 3) ZKCHK(derive_tables_build(t0, f, 0, s));
    if (sets & 4) ZKCHK(derive_tables_build(tn, f, n, s));          // the points n .. 2n-2 of the h values
    if (!side_by_side) {
        if (sets & 1) ZKCHK(derive_set<Fp>(t0, d_g1 + 96 * o_ti, n, out_g1 + 96 * 3, s));
        if (sets & 2) ZKCHK(derive_set<Fp2H>(t0, d_g2 + 192 * 2, n, out_g2 + 192 * 2, s));
        if (sets & 4) ZKCHK(derive_set<Fp>(tn, d_g1 + 96 * o_tz, n - 1, out_g1 + 96 * (3 + (uint64_t)n), s));
        HIPCHK(hipStreamSynchronize(s));
        return ZK_OK;
    }
    DeriveWork wk[3];                          // declared before the streams: the streams drain (SideStreams' destructor) before the arrays go
    SideStreams side;
    for (hipStream_t& x : side.st) HIPCHK(hipStreamCreateWithFlags(&x, hipStreamNonBlocking));
    for (hipEvent_t& e : side.ev) HIPCHK(hipEventCreateWithFlags(&e, hipEventDisableTiming));
    HIPCHK(hipEventRecord(side.ev[0], s));     // tables built, copies enqueued: the side streams start from here
    uint32_t used = 0;
    int rc = ZK_OK;
    auto stream_of = [&](uint32_t k) -> hipStream_t { return k == 0 ? s : side.st[k - 1]; };
    auto start = [&](uint32_t k) -> int {
        if (k) HIPCHK(hipStreamWaitEvent(side.st[k - 1], side.ev[0], 0));
        return ZK_OK;
    };
    // the G2 set first: it is the longest
    if (rc == ZK_OK && (sets & 2)) { rc = start(used); if (rc == ZK_OK) rc = derive_set_enqueue<Fp2H>(wk[used], t0, d_g2 + 192 * 2, n, out_g2 + 192 * 2, stream_of(used)); used++; }
    if (rc == ZK_OK && (sets & 1)) { rc = start(used); if (rc == ZK_OK) rc = derive_set_enqueue<Fp>(wk[used], t0, d_g1 + 96 * o_ti, n, out_g1 + 96 * 3, stream_of(used)); used++; }
    if (rc == ZK_OK && (sets & 4)) { rc = start(used); if (rc == ZK_OK) rc = derive_set_enqueue<Fp>(wk[used], tn, d_g1 + 96 * o_tz, n - 1, out_g1 + 96 * (3 + (uint64_t)n), stream_of(used)); used++; }
    for (uint32_t k = 1; k < used; k++) {      // the caller's stream ends after every set (also on the way out of a failed enqueue)
        if (hipEventRecord(side.ev[k], side.st[k - 1]) == hipSuccess) (void)hipStreamWaitEvent(s, side.ev[k], 0);
    }
    const hipError_t e_sync = hipStreamSynchronize(s);
    if (rc != ZK_OK) return rc;
    HIPCHK(e_sync);
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
