// BLS12-381 G1 / G2 group law for gfx950, generic over the coordinate field (Fp or Fp2).
//
// Accumulators use extended Jacobian "XYZZ" coordinates (x = X/ZZ, y = Y/ZZZ, ZZ^3 = ZZZ^2):
// the mixed addition XYZZ += affine costs 8M + 2S with no inversion, which is what the Pippenger
// bucket loop executes n * windows times.  Every special case the bucket sums hit (identity
// operands, P + P, P + (-P)) is branched explicitly -- bit-exact results depend on it
// (SURVEY.md 7.2 hard part 5).  ZZ = 0 encodes the identity, so zero-filled memory is a valid
// array of identities.  Affine (0, 0) encodes the identity (not on either curve since b != 0).
//
// Computes the same group elements as the reference's G.add / G.mul / G.negate
// (src/lib/zk/curve.ml:159-191, delegated to opam bls12-381).
#pragma once
#include "ff.cuh"

#include <type_traits>

namespace zk {

// F is the AT-REST coordinate type (Fp = FpB<64>, Fp2, Fp2H: value bound 64 p, see ff.cuh): struct members
// and loop-carried accumulators have it; temporaries inside a formula are `auto` and carry their own
// bound in the type, and assigning one back to an F member checks (at compile time) that it fits.
template <class F> struct FieldOps;
template <> struct FieldOps<Fp> {
    static FF_INLINE Fp zero() { return fp_zero(); }
    static FF_INLINE Fp one() { return fp_one(); }
    static FF_INLINE Fp curve_b() { return fe_dbl(fe_dbl(fp_one())); }   // 4
    static constexpr int WORDS = 12;
};
template <> struct FieldOps<Fp2> {
    static FF_INLINE Fp2 zero() { return fp2_zero(); }
    static FF_INLINE Fp2 one() { return fp2_one(); }
    static FF_INLINE Fp2 curve_b() { const FpB<4> f = fe_dbl(fe_dbl(fp_one())); return {f, f}; }  // 4 + 4u
    static constexpr int WORDS = 24;
};

template <> struct FieldOps<Fp2H> {
    static FF_INLINE Fp2H zero() { return {fp_zero()}; }
    static FF_INLINE Fp2H one() { return {pair_comp() ? fp_zero() : fp_one()}; }
    static constexpr int WORDS = 24;     // memory layout is the one of Fp2: c0 | c1
};

template <class F> struct Aff {
    F x, y;
};
template <class F> struct Xyzz {
    F x, y, zz, zzz;
};
using G1Aff = Aff<Fp>;
using G2Aff = Aff<Fp2>;
using G1Xyzz = Xyzz<Fp>;
using G2Xyzz = Xyzz<Fp2>;

template <class F> FF_INLINE bool aff_is_inf(const Aff<F>& p) { return fe_is_zero(p.x) && fe_is_zero(p.y); }
template <class F> FF_INLINE Aff<F> aff_inf() { return {FieldOps<F>::zero(), FieldOps<F>::zero()}; }
template <class F> FF_INLINE bool xyzz_is_inf(const Xyzz<F>& p) { return fe_is_zero(p.zz); }
template <class F> FF_INLINE Xyzz<F> xyzz_inf() {
    F z = FieldOps<F>::zero();
    return {z, z, z, z};
}
template <class F> FF_INLINE Xyzz<F> xyzz_from_aff(const Aff<F>& p) {
    if (aff_is_inf(p)) return xyzz_inf<F>();
    return {p.x, p.y, FieldOps<F>::one(), FieldOps<F>::one()};
}

// y^2 == x^3 + b
template <class F> FF_INLINE bool aff_on_curve(const Aff<F>& p) {
    if (aff_is_inf(p)) return true;
    const auto l = fe_sqr(p.y);
    const auto r = fe_add(fe_mul(fe_sqr(p.x), p.x), FieldOps<F>::curve_b());
    return fe_eq(l, r);
}

// dbl-2008-s-1 (a = 0)
template <class F> FF_INLINE Xyzz<F> xyzz_dbl_impl(const Xyzz<F>& p) {
    if (xyzz_is_inf(p) || fe_is_zero(p.y)) return xyzz_inf<F>();
    const auto U = fe_dbl(p.y);
    const auto V = fe_sqr(U);
    const auto W = fe_mul(U, V);
    const auto S = fe_mul(p.x, V);
    const auto X2 = fe_sqr(p.x);
    const auto M = fe_add(fe_dbl(X2), X2);
    const auto X3 = fe_sub(fe_sqr(M), fe_dbl(S));
    const auto Y3 = fe_mul_sub(M, fe_sub(S, X3), W, p.y);
    return {X3, Y3, fe_mul(V, p.zz), fe_mul(W, p.zzz)};
}
// mdbl-2008-s-1: doubling of an affine point
template <class F> FF_INLINE Xyzz<F> xyzz_dbl_aff(const Aff<F>& p) {
    if (aff_is_inf(p) || fe_is_zero(p.y)) return xyzz_inf<F>();
    const auto U = fe_dbl(p.y);
    const auto V = fe_sqr(U);
    const auto W = fe_mul(U, V);
    const auto S = fe_mul(p.x, V);
    const auto X2 = fe_sqr(p.x);
    const auto M = fe_add(fe_dbl(X2), X2);
    const auto X3 = fe_sub(fe_sqr(M), fe_dbl(S));
    const auto Y3 = fe_mul_sub(M, fe_sub(S, X3), W, p.y);
    return {X3, Y3, V, W};
}
// The equal-x case of a mixed addition (P + P or P + (-P)): a doubling's worth of code that a bucket loop meets once in 2^381 additions unless
// the bases repeat.  Out of line -- ~17 KB of straight-line code with the products expanded in place, in the middle of a loop body that has to
// live in the instruction cache -- and on the ACCUMULATOR, which holds the same point as q here (equal x, equal y) and is live anyway: doubling
// q instead kept the 28 words of the table entry alive across the whole addition, and the compiler parked them in scratch memory in EVERY
// iteration (PMC: 112 B of scratch writes per addition, 5.9 GB per 2^20 proof, for a path that never runs).
template <class F> __device__ __noinline__ void xyzz_madd_equal_x_fn(Xyzz<F>* acc, int same_y) {
    if (same_y) *acc = xyzz_dbl_impl(*acc);
    else *acc = xyzz_inf<F>();
}
template <class F> FF_INLINE void xyzz_madd_equal_x(Xyzz<F>& acc, bool same_y) {
    Xyzz<F> t = acc;
    xyzz_madd_equal_x_fn<F>(&t, same_y ? 1 : 0);
    acc = t;
}
// madd-2008-s: acc += q (q affine).  Q_MAY_BE_INF = false: the caller knows q is a genuine point (entries of the resident base tables: the
// counting sort never files an identity base into a bucket), so the identity test -- 28 limbs OR-ed, a zero test of the lazily reduced y with
// its out-of-line slow path, and the registers that path pins -- stays out of the bucket loop
template <class F, bool Q_MAY_BE_INF = true> FF_INLINE void xyzz_madd_impl(Xyzz<F>& acc, const Aff<F>& q) {
    if constexpr (Q_MAY_BE_INF) {
        if (aff_is_inf(q)) return;
    }
    if (xyzz_is_inf(acc)) {
        acc = {q.x, q.y, FieldOps<F>::one(), FieldOps<F>::one()};
        return;
    }
    const auto U2 = fe_mul(q.x, acc.zz);
    const auto S2 = fe_mul(q.y, acc.zzz);
    const auto P = fe_sub(U2, acc.x);
    const auto R = fe_sub(S2, acc.y);
    if (fe_is_zero(P)) {
        xyzz_madd_equal_x(acc, fe_is_zero(R));
        return;
    }
    const auto PP = fe_sqr(P);
    const auto PPP = fe_mul(P, PP);
    const auto Q = fe_mul(acc.x, PP);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), PPP, Q);          // R^2 - PPP - 2 Q, one carry pass
    const auto Y3 = fe_mul_sub(R, fe_sub(Q, X3), acc.y, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(acc.zz, PP);
    acc.zzz = fe_mul(acc.zzz, PPP);
}
// mmadd-2008-s: the same when acc is known to hold an AFFINE point (zz = zzz = 1) or the identity -- the second entry
// of a bucket run.  6 products instead of 10: U2 = x2, S2 = y2, ZZ3 = PP, ZZZ3 = PPP.
template <class F, bool Q_MAY_BE_INF = true> FF_INLINE void xyzz_mmadd_impl(Xyzz<F>& acc, const Aff<F>& q) {
    if constexpr (Q_MAY_BE_INF) {
        if (aff_is_inf(q)) return;
    }
    if (xyzz_is_inf(acc)) {
        acc = {q.x, q.y, FieldOps<F>::one(), FieldOps<F>::one()};
        return;
    }
    const auto P = fe_sub(q.x, acc.x);
    const auto R = fe_sub(q.y, acc.y);
    if (fe_is_zero(P)) {
        xyzz_madd_equal_x(acc, fe_is_zero(R));
        return;
    }
    const auto PP = fe_sqr(P);
    const auto PPP = fe_mul(P, PP);
    const auto Q = fe_mul(acc.x, PP);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), PPP, Q);          // R^2 - PPP - 2 Q, one carry pass
    const auto Y3 = fe_mul_sub(R, fe_sub(Q, X3), acc.y, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = PP;
    acc.zzz = PPP;
}
// add-2008-s: acc += q (both XYZZ)
template <class F> FF_INLINE void xyzz_add_impl(Xyzz<F>& acc, const Xyzz<F>& q) {
    if (xyzz_is_inf(q)) return;
    if (xyzz_is_inf(acc)) {
        acc = q;
        return;
    }
    const auto U1 = fe_mul(acc.x, q.zz);
    const auto U2 = fe_mul(q.x, acc.zz);
    const auto S1 = fe_mul(acc.y, q.zzz);
    const auto S2 = fe_mul(q.y, acc.zzz);
    const auto P = fe_sub(U2, U1);
    const auto R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) acc = xyzz_dbl_impl(acc);
        else acc = xyzz_inf<F>();
        return;
    }
    const auto PP = fe_sqr(P);
    const auto PPP = fe_mul(P, PP);
    const auto Q = fe_mul(U1, PP);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), PPP, Q);          // R^2 - PPP - 2 Q, one carry pass
    const auto Y3 = fe_mul_sub(R, fe_sub(Q, X3), S1, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(fe_mul(acc.zz, q.zz), PP);
    acc.zzz = fe_mul(fe_mul(acc.zzz, q.zzz), PPP);
}
// The group operations are REAL functions (one copy per coordinate field in each translation
// unit), with operands passed by address: kernels that chain many of them stay small, which keeps
// hipcc's compile time in minutes and its register allocator out of trouble (a fully inlined G2
// Horner loop -- hundreds of thousands of instructions -- was miscompiled by ROCm 7.2's clang).
// The data crosses the call through the wave's private memory: ~1 KB of scratch traffic against
// ~10^4 ALU instructions per operation.
template <class F> __device__ __noinline__ void xyzz_add_fn(Xyzz<F>* acc, const Xyzz<F>* q) { xyzz_add_impl(*acc, *q); }
template <class F> __device__ __noinline__ void xyzz_madd_fn(Xyzz<F>* acc, const Aff<F>* q) { xyzz_madd_impl(*acc, *q); }
template <class F> __device__ __noinline__ void xyzz_dbl_fn(Xyzz<F>* r, const Xyzz<F>* p) { *r = xyzz_dbl_impl(*p); }
template <class F> FF_INLINE void xyzz_add(Xyzz<F>& acc, const Xyzz<F>& q) { xyzz_add_fn<F>(&acc, &q); }
template <class F> FF_INLINE void xyzz_madd(Xyzz<F>& acc, const Aff<F>& q) { xyzz_madd_fn<F>(&acc, &q); }
template <class F> FF_INLINE Xyzz<F> xyzz_dbl(const Xyzz<F>& p) {
    Xyzz<F> r;
    xyzz_dbl_fn<F>(&r, &p);
    return r;
}

// one inversion: 1/(ZZ*ZZZ)
template <class F> FF_INLINE Aff<F> xyzz_to_aff(const Xyzz<F>& p) {
    if (xyzz_is_inf(p)) return aff_inf<F>();
    const auto i = fe_inv(fe_mul(p.zz, p.zzz));
    const auto izz = fe_mul(i, p.zzz);
    const auto izzz = fe_mul(i, p.zz);
    return {fe_mul(p.x, izz), fe_mul(p.y, izzz)};
}
// ---- memory layout: affine points are stored as consecutive Montgomery coordinates
//      (G1: x | y = 96 B; G2: x.c0 | x.c1 | y.c0 | y.c1 = 192 B), XYZZ as x | y | zz | zzz.
//      Memory always holds fully reduced coordinates as 12 dense 32-bit words (48 B).
FF_INLINE Fp load_f(const Fp*, const void* p) { return fp_load(p); }
FF_INLINE Fp2 load_f(const Fp2*, const void* p) { return {fp_load(p), fp_load((const char*)p + 48)}; }
FF_INLINE Fp2H load_f(const Fp2H*, const void* p) { return {fp_load((const char*)p + 48 * pair_comp())}; }
FF_INLINE void store_f(void* p, const Fp2H& a) { fp_store((char*)p + 48 * pair_comp(), a.v); }
FF_INLINE void store_f(void* p, const Fp& a) { fp_store(p, a); }
FF_INLINE void store_f(void* p, const Fp2& a) {
    fp_store(p, a.c0);
    fp_store((char*)p + 48, a.c1);
}
template <class F> FF_INLINE Aff<F> aff_load(const void* p) {
    constexpr int B = FieldOps<F>::WORDS * 4;
    return {load_f((const F*)nullptr, p), load_f((const F*)nullptr, (const char*)p + B)};
}
template <class F> FF_INLINE void aff_store(void* p, const Aff<F>& a) {
    constexpr int B = FieldOps<F>::WORDS * 4;
    store_f(p, a.x);
    store_f((char*)p + B, a.y);
}
template <class F> FF_INLINE Xyzz<F> xyzz_load(const void* p) {
    constexpr int B = FieldOps<F>::WORDS * 4;
    const char* c = (const char*)p;
    return {load_f((const F*)nullptr, c), load_f((const F*)nullptr, c + B), load_f((const F*)nullptr, c + 2 * B),
            load_f((const F*)nullptr, c + 3 * B)};
}
template <class F> FF_INLINE void xyzz_store(void* p, const Xyzz<F>& a) {
    constexpr int B = FieldOps<F>::WORDS * 4;
    char* c = (char*)p;
    store_f(c, a.x);
    store_f(c + B, a.y);
    store_f(c + 2 * B, a.zz);
    store_f(c + 3 * B, a.zzz);
}

// ---- internal ("raw") layout of intermediate XYZZ points (bucket sums and everything downstream of them):
//      every Fp component as its 14 register limbs in a 16-word slot (64 B, 128-bit accesses), lazily reduced
//      exactly as in registers.  Writing and reading back is a plain copy: no reduction, no repacking.
//      G1: x | y | zz | zzz = 256 B;  G2: x.c0 | x.c1 | y.c0 | ... = 512 B.  Zero-filled memory is the identity.
template <class F> struct RawLayout;
template <> struct RawLayout<Fp> { static constexpr int ELEM = 64, XYZZ = 256, LANES = 1; };
template <> struct RawLayout<Fp2> { static constexpr int ELEM = 128, XYZZ = 512, LANES = 1; };
template <> struct RawLayout<Fp2H> { static constexpr int ELEM = 128, XYZZ = 512, LANES = 2; };
FF_INLINE Fp fp_load_raw(const void* p) {
    const uint4* q = reinterpret_cast<const uint4*>(p);
    const uint4 a = q[0], b = q[1], c = q[2], d = q[3];
    Fp r;
    r.v[0] = a.x; r.v[1] = a.y; r.v[2] = a.z; r.v[3] = a.w;
    r.v[4] = b.x; r.v[5] = b.y; r.v[6] = b.z; r.v[7] = b.w;
    r.v[8] = c.x; r.v[9] = c.y; r.v[10] = c.z; r.v[11] = c.w;
    r.v[12] = d.x; r.v[13] = d.y;
    return r;
}
FF_INLINE void fp_store_raw(void* p, const Fp& a) {
    uint4* q = reinterpret_cast<uint4*>(p);
    q[0] = make_uint4(a.v[0], a.v[1], a.v[2], a.v[3]);
    q[1] = make_uint4(a.v[4], a.v[5], a.v[6], a.v[7]);
    q[2] = make_uint4(a.v[8], a.v[9], a.v[10], a.v[11]);
    q[3] = make_uint4(a.v[12], a.v[13], 0u, 0u);
}
FF_INLINE Fp load_raw_f(const Fp*, const void* p) { return fp_load_raw(p); }
FF_INLINE Fp2 load_raw_f(const Fp2*, const void* p) { return {fp_load_raw(p), fp_load_raw((const char*)p + 64)}; }
FF_INLINE Fp2H load_raw_f(const Fp2H*, const void* p) { return {fp_load_raw((const char*)p + 64 * pair_comp())}; }
FF_INLINE void store_raw_f(void* p, const Fp& a) { fp_store_raw(p, a); }
FF_INLINE void store_raw_f(void* p, const Fp2& a) {
    fp_store_raw(p, a.c0);
    fp_store_raw((char*)p + 64, a.c1);
}
FF_INLINE void store_raw_f(void* p, const Fp2H& a) { fp_store_raw((char*)p + 64 * pair_comp(), a.v); }
template <class F> FF_INLINE Xyzz<F> xyzz_load_raw(const void* p) {
    constexpr int B = RawLayout<F>::ELEM;
    const char* c = (const char*)p;
    return {load_raw_f((const F*)nullptr, c), load_raw_f((const F*)nullptr, c + B), load_raw_f((const F*)nullptr, c + 2 * B),
            load_raw_f((const F*)nullptr, c + 3 * B)};
}
template <class F> FF_INLINE void xyzz_store_raw(void* p, const Xyzz<F>& a) {
    constexpr int B = RawLayout<F>::ELEM;
    char* c = (char*)p;
    store_raw_f(c, a.x);
    store_raw_f(c + B, a.y);
    store_raw_f(c + 2 * B, a.zz);
    store_raw_f(c + 3 * B, a.zzz);
}
// add-2008-s with the second operand READ FROM MEMORY (raw layout) coordinate by coordinate, each right before its product, and ZZ / ZZZ read a
// second time for the last two products: q never occupies 4 x 14 registers beside the accumulator and the temporaries of the formula.  The reduction
// kernels that add bucket sums and chunk partials (msm.hip: k_msm_fixup, k_msm_digit_sums) spilled 25-39 registers with q loaded up front.
template <class F> FF_INLINE void xyzz_add_raw_mem(Xyzz<F>& acc, const uint8_t* __restrict__ q) {
    constexpr int B = RawLayout<F>::ELEM;
    {
        const F qzz = load_raw_f((const F*)nullptr, q + 2 * B);
        if (fe_is_zero(qzz)) return;
        if (xyzz_is_inf(acc)) {
            acc = xyzz_load_raw<F>(q);
            return;
        }
    }
    const auto U1 = fe_mul(acc.x, load_raw_f((const F*)nullptr, q + 2 * B));
    const auto U2 = fe_mul(load_raw_f((const F*)nullptr, q), acc.zz);
    const auto S1 = fe_mul(acc.y, load_raw_f((const F*)nullptr, q + 3 * B));
    const auto S2 = fe_mul(load_raw_f((const F*)nullptr, q + B), acc.zzz);
    const auto P = fe_sub(U2, U1);
    const auto R = fe_sub(S2, S1);
    if (fe_is_zero(P)) {
        if (fe_is_zero(R)) acc = xyzz_dbl_impl(acc);
        else acc = xyzz_inf<F>();
        return;
    }
    const auto PP = fe_sqr(P);
    const auto PPP = fe_mul(P, PP);
    const auto Q = fe_mul(U1, PP);
    const auto X3 = fe_sub_sub_dbl(fe_sqr(R), PPP, Q);          // R^2 - PPP - 2 Q, one carry pass
    const auto Y3 = fe_mul_sub(R, fe_sub(Q, X3), S1, PPP);
    acc.x = X3;
    acc.y = Y3;
    acc.zz = fe_mul(fe_mul(acc.zz, load_raw_f((const F*)nullptr, q + 2 * B)), PP);
    acc.zzz = fe_mul(fe_mul(acc.zzz, load_raw_f((const F*)nullptr, q + 3 * B)), PPP);
}
// ---- layout of the RESIDENT BASE TABLES (MsmBases::table): one 128-byte record per lane that reads the entry --
//      x as its 14 register limbs | y as its 14 register limbs | 16 B of padding, canonical (< p, exact 29-bit limbs).
//      G1: one record per point (128 B); G2: two records per point (256 B), record c = the c-th Fp2 component of x and of y, i.e. what
//      lane c of a lane pair keeps.  A gather touches exactly ONE 128-byte cache line per lane (the dense 96-byte entries straddled two
//      lines at every other index: 1.25-1.5 lines per gather) and the limbs go from memory to the product without repacking.
//      The identity is all-zero limbs, as everywhere.
static constexpr int TAB_REC = 128, TAB_REC_WORDS = 2 * FPL;
template <class F> struct TableLayout;
template <> struct TableLayout<Fp> { static constexpr int ENTRY = TAB_REC; };
template <> struct TableLayout<Fp2> { static constexpr int ENTRY = 2 * TAB_REC; };
template <> struct TableLayout<Fp2H> { static constexpr int ENTRY = 2 * TAB_REC; };
struct TabRec {          // one record in registers, still untyped: what the accumulate loop keeps in flight for the NEXT step
    uint32_t w[TAB_REC_WORDS];
};
FF_INLINE TabRec tab_rec_load(const uint8_t* rec) {
    TabRec r;
    const uint4* q = reinterpret_cast<const uint4*>(rec);
#pragma unroll
    for (int i = 0; i < TAB_REC_WORDS / 4; i++) {
        const uint4 x = q[i];
        r.w[4 * i] = x.x; r.w[4 * i + 1] = x.y; r.w[4 * i + 2] = x.z; r.w[4 * i + 3] = x.w;
    }
    return r;
}
FF_INLINE void tab_rec_store(uint8_t* rec, const FpB<1>& x, const FpB<1>& y) {
    uint32_t w[32];
#pragma unroll
    for (int i = 0; i < FPL; i++) { w[i] = x.v[i]; w[FPL + i] = y.v[i]; }
#pragma unroll
    for (int i = TAB_REC_WORDS; i < 32; i++) w[i] = 0;
    uint4* q = reinterpret_cast<uint4*>(rec);
#pragma unroll
    for (int i = 0; i < 8; i++) q[i] = make_uint4(w[4 * i], w[4 * i + 1], w[4 * i + 2], w[4 * i + 3]);
}
FF_INLINE FpB<1> tab_rec_x(const TabRec& r) {
    FpB<1> x;
#pragma unroll
    for (int i = 0; i < FPL; i++) x.v[i] = r.w[i];
    return x;
}
FF_INLINE FpB<1> tab_rec_y(const TabRec& r) {
    FpB<1> y;
#pragma unroll
    for (int i = 0; i < FPL; i++) y.v[i] = r.w[FPL + i];
    return y;
}
// whole entries (key set-up kernels: one lane per point)
FF_INLINE void tab_store(uint8_t* entry, const Aff<Fp>& a) { tab_rec_store(entry, fp_canon(a.x), fp_canon(a.y)); }
FF_INLINE void tab_store(uint8_t* entry, const Aff<Fp2>& a) {
    tab_rec_store(entry, fp_canon(a.x.c0), fp_canon(a.y.c0));
    tab_rec_store(entry + TAB_REC, fp_canon(a.x.c1), fp_canon(a.y.c1));
}
FF_INLINE Aff<Fp> tab_load(const Fp*, const uint8_t* entry) {
    const TabRec r = tab_rec_load(entry);
    return {tab_rec_x(r), tab_rec_y(r)};
}
FF_INLINE Aff<Fp2> tab_load(const Fp2*, const uint8_t* entry) {
    const TabRec r0 = tab_rec_load(entry), r1 = tab_rec_load(entry + TAB_REC);
    return {{tab_rec_x(r0), tab_rec_x(r1)}, {tab_rec_y(r0), tab_rec_y(r1)}};
}

// the 4 x 14 limbs one lane holds of a point (a whole G1 point, or one Fp2 component of a G2 point)
static constexpr int LANE_POINT_WORDS = 4 * FPL;
FF_INLINE void xyzz_to_words(uint32_t* w, const Xyzz<Fp>& a) {
#pragma unroll
    for (int i = 0; i < FPL; i++) { w[i] = a.x.v[i]; w[FPL + i] = a.y.v[i]; w[2 * FPL + i] = a.zz.v[i]; w[3 * FPL + i] = a.zzz.v[i]; }
}
FF_INLINE void xyzz_from_words(Xyzz<Fp>& a, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < FPL; i++) { a.x.v[i] = w[i]; a.y.v[i] = w[FPL + i]; a.zz.v[i] = w[2 * FPL + i]; a.zzz.v[i] = w[3 * FPL + i]; }
}
FF_INLINE void xyzz_to_words(uint32_t* w, const Xyzz<Fp2H>& a) {
#pragma unroll
    for (int i = 0; i < FPL; i++) { w[i] = a.x.v.v[i]; w[FPL + i] = a.y.v.v[i]; w[2 * FPL + i] = a.zz.v.v[i]; w[3 * FPL + i] = a.zzz.v.v[i]; }
}
FF_INLINE void xyzz_from_words(Xyzz<Fp2H>& a, const uint32_t* w) {
#pragma unroll
    for (int i = 0; i < FPL; i++) { a.x.v.v[i] = w[i]; a.y.v.v[i] = w[FPL + i]; a.zz.v.v[i] = w[2 * FPL + i]; a.zzz.v.v[i] = w[3 * FPL + i]; }
}

}  // namespace zk
