// Host-side optimal-ate pairing on BLS12-381 for the `verify` surface (scope row f1):
// Groth16 verify is 3 pairings + a small io product (src/groth16/groth16.ml:163-173), Pinocchio's
// Verify.f 13 pairings (src/pinocchio/pinocchio.ml:254-420).  The reference obtains `Pairing.pairing`
// from opam bls12-381 (curve.ml:77); a handful of pairings per proof is host work there and here --
// nothing in this file runs on the GPU (it is a .hip file only so that the one Makefile rule builds it).
//
// Deliberately the transparent formulation: Fp on 6 x 64-bit Montgomery limbs, the tower
// Fp2 = Fp[u]/(u^2+1), Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v) with schoolbook products,
// the twist point mapped into E(Fp12) (x / w^2, y / w^3), an affine Miller loop over |x| and the final
// exponentiation as one power (p^12 - 1) / r.  ~0.1 s per pairing product; correctness over speed.
// GT encoding (ours; the reference's is defined by its external library): the 12 Fp coefficients
// c0.c0.a, c0.c0.b, c0.c1.a, ... c1.c2.b as 48-byte big-endian integers = 576 B.
#include "../../include/zkmi355x.h"
#include "pairing_consts.h"
#include "zk_err.h"

#include <string.h>
#include <vector>

namespace hp {
typedef unsigned __int128 u128;

struct Fp {
    uint64_t l[6];
};
static inline Fp fp_zero() { Fp r; memset(&r, 0, sizeof r); return r; }
static inline Fp fp_one() { Fp r; memcpy(r.l, HP_ONE, 48); return r; }
static inline bool fp_is_zero(const Fp& a) { uint64_t o = 0; for (int i = 0; i < 6; i++) o |= a.l[i]; return o == 0; }
static inline bool fp_eq(const Fp& a, const Fp& b) { return memcmp(a.l, b.l, 48) == 0; }
static inline bool geq_p(const uint64_t* a) {
    for (int i = 5; i >= 0; i--) {
        if (a[i] > HP_P[i]) return true;
        if (a[i] < HP_P[i]) return false;
    }
    return true;
}
static inline void sub_p(uint64_t* a) {
    u128 bw = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a[i] - HP_P[i] - (uint64_t)bw;
        a[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
}
static inline Fp fp_add(const Fp& a, const Fp& b) {
    Fp r;
    u128 c = 0;
    for (int i = 0; i < 6; i++) { c += (u128)a.l[i] + b.l[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    if (geq_p(r.l)) sub_p(r.l);          // p < 2^381: no carry out of 384 bits
    return r;
}
static inline Fp fp_sub(const Fp& a, const Fp& b) {
    Fp r;
    u128 bw = 0;
    for (int i = 0; i < 6; i++) {
        u128 d = (u128)a.l[i] - b.l[i] - (uint64_t)bw;
        r.l[i] = (uint64_t)d;
        bw = (d >> 64) & 1;
    }
    if (bw) {
        u128 c = 0;
        for (int i = 0; i < 6; i++) { c += (u128)r.l[i] + HP_P[i]; r.l[i] = (uint64_t)c; c >>= 64; }
    }
    return r;
}
static inline Fp fp_neg(const Fp& a) { return fp_sub(fp_zero(), a); }
// CIOS Montgomery product
static Fp fp_mul(const Fp& a, const Fp& b) {
    uint64_t t[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = 0; i < 6; i++) {
        u128 c = 0;
        for (int j = 0; j < 6; j++) {
            c += (u128)a.l[j] * b.l[i] + t[j];
            t[j] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[6] = (uint64_t)c;
        t[7] = (uint64_t)(c >> 64);
        const uint64_t m = t[0] * HP_NINV;
        c = ((u128)m * HP_P[0] + t[0]) >> 64;
        for (int j = 1; j < 6; j++) {
            c += (u128)m * HP_P[j] + t[j];
            t[j - 1] = (uint64_t)c;
            c >>= 64;
        }
        c += t[6];
        t[5] = (uint64_t)c;
        t[6] = t[7] + (uint64_t)(c >> 64);
        t[7] = 0;
    }
    Fp r;
    memcpy(r.l, t, 48);
    if (t[6] || geq_p(r.l)) sub_p(r.l);
    return r;
}
static inline Fp fp_sqr(const Fp& a) { return fp_mul(a, a); }
static Fp fp_pow(const Fp& a, const uint64_t* e, int nlimbs) {
    Fp acc = fp_one();
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        acc = fp_sqr(acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = fp_mul(acc, a);
    }
    return acc;
}
static inline Fp fp_inv(const Fp& a) { return fp_pow(a, HP_PM2, 6); }     // inv(0) = 0
static Fp fp_from_u64(uint64_t x) {
    Fp r = fp_zero(), r2;
    r.l[0] = x;
    memcpy(r2.l, HP_R2, 48);
    return fp_mul(r, r2);
}
// 48 B big-endian canonical integer <-> Montgomery; false if >= p
static bool fp_from_be(Fp& out, const uint8_t* p) {
    Fp r;
    for (int i = 0; i < 6; i++) {
        uint64_t w = 0;
        for (int k = 0; k < 8; k++) w = (w << 8) | p[8 * (5 - i) + k];
        r.l[i] = w;
    }
    if (geq_p(r.l)) return false;
    Fp r2;
    memcpy(r2.l, HP_R2, 48);
    out = fp_mul(r, r2);
    return true;
}
static void fp_to_be(uint8_t* p, const Fp& a) {
    Fp one = fp_zero();
    one.l[0] = 1;
    const Fp r = fp_mul(a, one);
    for (int i = 0; i < 6; i++)
        for (int k = 0; k < 8; k++) p[8 * (5 - i) + k] = (uint8_t)(r.l[i] >> (8 * (7 - k)));
}

struct Fp2 {
    Fp a, b;
};
static inline Fp2 f2(const Fp& a, const Fp& b) { return Fp2{a, b}; }
static inline Fp2 f2_zero() { return f2(fp_zero(), fp_zero()); }
static inline Fp2 f2_one() { return f2(fp_one(), fp_zero()); }
static inline Fp2 f2_add(const Fp2& x, const Fp2& y) { return f2(fp_add(x.a, y.a), fp_add(x.b, y.b)); }
static inline Fp2 f2_sub(const Fp2& x, const Fp2& y) { return f2(fp_sub(x.a, y.a), fp_sub(x.b, y.b)); }
static inline Fp2 f2_neg(const Fp2& x) { return f2(fp_neg(x.a), fp_neg(x.b)); }
static inline Fp2 f2_mul(const Fp2& x, const Fp2& y) {
    return f2(fp_sub(fp_mul(x.a, y.a), fp_mul(x.b, y.b)), fp_add(fp_mul(x.a, y.b), fp_mul(x.b, y.a)));
}
static inline Fp2 f2_mul_xi(const Fp2& x) { return f2(fp_sub(x.a, x.b), fp_add(x.a, x.b)); }      // * (1 + u)
static inline Fp2 f2_inv(const Fp2& x) {
    const Fp d = fp_inv(fp_add(fp_sqr(x.a), fp_sqr(x.b)));
    return f2(fp_mul(x.a, d), fp_neg(fp_mul(x.b, d)));
}
static inline bool f2_is_zero(const Fp2& x) { return fp_is_zero(x.a) && fp_is_zero(x.b); }
static inline bool f2_eq(const Fp2& x, const Fp2& y) { return fp_eq(x.a, y.a) && fp_eq(x.b, y.b); }

struct Fp6 {
    Fp2 c0, c1, c2;
};
static inline Fp6 f6_zero() { return Fp6{f2_zero(), f2_zero(), f2_zero()}; }
static inline Fp6 f6_one() { return Fp6{f2_one(), f2_zero(), f2_zero()}; }
static inline Fp6 f6_add(const Fp6& x, const Fp6& y) { return Fp6{f2_add(x.c0, y.c0), f2_add(x.c1, y.c1), f2_add(x.c2, y.c2)}; }
static inline Fp6 f6_sub(const Fp6& x, const Fp6& y) { return Fp6{f2_sub(x.c0, y.c0), f2_sub(x.c1, y.c1), f2_sub(x.c2, y.c2)}; }
static inline Fp6 f6_neg(const Fp6& x) { return Fp6{f2_neg(x.c0), f2_neg(x.c1), f2_neg(x.c2)}; }
static Fp6 f6_mul(const Fp6& x, const Fp6& y) {
    const Fp2 t0 = f2_add(f2_mul(x.c0, y.c0), f2_mul_xi(f2_add(f2_mul(x.c1, y.c2), f2_mul(x.c2, y.c1))));
    const Fp2 t1 = f2_add(f2_add(f2_mul(x.c0, y.c1), f2_mul(x.c1, y.c0)), f2_mul_xi(f2_mul(x.c2, y.c2)));
    const Fp2 t2 = f2_add(f2_add(f2_mul(x.c0, y.c2), f2_mul(x.c1, y.c1)), f2_mul(x.c2, y.c0));
    return Fp6{t0, t1, t2};
}
static inline Fp6 f6_mul_by_v(const Fp6& x) { return Fp6{f2_mul_xi(x.c2), x.c0, x.c1}; }
static Fp6 f6_inv(const Fp6& x) {
    const Fp2 t0 = f2_sub(f2_mul(x.c0, x.c0), f2_mul_xi(f2_mul(x.c1, x.c2)));
    const Fp2 t1 = f2_sub(f2_mul_xi(f2_mul(x.c2, x.c2)), f2_mul(x.c0, x.c1));
    const Fp2 t2 = f2_sub(f2_mul(x.c1, x.c1), f2_mul(x.c0, x.c2));
    const Fp2 d = f2_inv(f2_add(f2_mul(x.c0, t0), f2_add(f2_mul_xi(f2_mul(x.c2, t1)), f2_mul_xi(f2_mul(x.c1, t2)))));
    return Fp6{f2_mul(t0, d), f2_mul(t1, d), f2_mul(t2, d)};
}
static inline bool f6_eq(const Fp6& x, const Fp6& y) { return f2_eq(x.c0, y.c0) && f2_eq(x.c1, y.c1) && f2_eq(x.c2, y.c2); }

struct Fp12 {
    Fp6 c0, c1;
};
static inline Fp12 f12_one() { return Fp12{f6_one(), f6_zero()}; }
static inline Fp12 f12_add(const Fp12& x, const Fp12& y) { return Fp12{f6_add(x.c0, y.c0), f6_add(x.c1, y.c1)}; }
static inline Fp12 f12_sub(const Fp12& x, const Fp12& y) { return Fp12{f6_sub(x.c0, y.c0), f6_sub(x.c1, y.c1)}; }
static Fp12 f12_mul(const Fp12& x, const Fp12& y) {
    return Fp12{f6_add(f6_mul(x.c0, y.c0), f6_mul_by_v(f6_mul(x.c1, y.c1))), f6_add(f6_mul(x.c0, y.c1), f6_mul(x.c1, y.c0))};
}
static Fp12 f12_inv(const Fp12& x) {
    const Fp6 d = f6_inv(f6_sub(f6_mul(x.c0, x.c0), f6_mul_by_v(f6_mul(x.c1, x.c1))));
    return Fp12{f6_mul(x.c0, d), f6_neg(f6_mul(x.c1, d))};
}
static inline Fp12 f12_conj(const Fp12& x) { return Fp12{x.c0, f6_neg(x.c1)}; }
static inline bool f12_eq(const Fp12& x, const Fp12& y) { return f6_eq(x.c0, y.c0) && f6_eq(x.c1, y.c1); }
static inline Fp12 f12_from_fp(const Fp& a) { return Fp12{Fp6{f2(a, fp_zero()), f2_zero(), f2_zero()}, f6_zero()}; }
static inline Fp12 f12_from_fp2(const Fp2& a) { return Fp12{Fp6{a, f2_zero(), f2_zero()}, f6_zero()}; }
static Fp12 f12_pow(const Fp12& x, const uint64_t* e, int bits) {
    Fp12 acc = f12_one();
    for (int i = bits - 1; i >= 0; i--) {
        acc = f12_mul(acc, acc);
        if ((e[i / 64] >> (i % 64)) & 1) acc = f12_mul(acc, x);
    }
    return acc;
}

// ---- curve points (affine; inf flag)
struct G1 { Fp x, y; bool inf; };
struct G2 { Fp2 x, y; bool inf; };

// generic affine group law over a field given by (add, sub, mul, inv, is_zero, eq)
template <class F> struct Ops;
template <> struct Ops<Fp> {
    static Fp add(const Fp& a, const Fp& b) { return fp_add(a, b); }
    static Fp sub(const Fp& a, const Fp& b) { return fp_sub(a, b); }
    static Fp mul(const Fp& a, const Fp& b) { return fp_mul(a, b); }
    static Fp inv(const Fp& a) { return fp_inv(a); }
    static bool is_zero(const Fp& a) { return fp_is_zero(a); }
    static bool eq(const Fp& a, const Fp& b) { return fp_eq(a, b); }
};
template <> struct Ops<Fp2> {
    static Fp2 add(const Fp2& a, const Fp2& b) { return f2_add(a, b); }
    static Fp2 sub(const Fp2& a, const Fp2& b) { return f2_sub(a, b); }
    static Fp2 mul(const Fp2& a, const Fp2& b) { return f2_mul(a, b); }
    static Fp2 inv(const Fp2& a) { return f2_inv(a); }
    static bool is_zero(const Fp2& a) { return f2_is_zero(a); }
    static bool eq(const Fp2& a, const Fp2& b) { return f2_eq(a, b); }
};
template <class PT, class F> static PT pt_add(const PT& p, const PT& q) {
    typedef Ops<F> O;
    if (p.inf) return q;
    if (q.inf) return p;
    F lam;
    if (O::eq(p.x, q.x)) {
        if (!O::eq(p.y, q.y) || O::is_zero(p.y)) { PT r = p; r.inf = true; return r; }
        const F xx = O::mul(p.x, p.x);
        lam = O::mul(O::add(O::add(xx, xx), xx), O::inv(O::add(p.y, p.y)));
    } else {
        lam = O::mul(O::sub(q.y, p.y), O::inv(O::sub(q.x, p.x)));
    }
    PT r;
    r.inf = false;
    r.x = O::sub(O::sub(O::mul(lam, lam), p.x), q.x);
    r.y = O::sub(O::mul(lam, O::sub(p.x, r.x)), p.y);
    return r;
}
template <class PT, class F> static PT pt_mul(const PT& p, const uint64_t* k, int nlimbs) {
    PT acc = p;
    acc.inf = true;
    for (int i = nlimbs * 64 - 1; i >= 0; i--) {
        acc = pt_add<PT, F>(acc, acc);
        if ((k[i / 64] >> (i % 64)) & 1) acc = pt_add<PT, F>(acc, p);
    }
    return acc;
}
// ZCash uncompressed decoding + curve and subgroup membership; 0 ok, else a ZK_ERR code
static int g1_decode(G1& out, const uint8_t* b) {
    if (b[0] & 0x80) return ZK_ERR_ARG;
    if (b[0] & 0x40) { out.inf = true; out.x = fp_zero(); out.y = fp_zero(); return ZK_OK; }
    if (!fp_from_be(out.x, b) || !fp_from_be(out.y, b + 48)) return ZK_ERR_ARG;
    out.inf = false;
    const Fp rhs = fp_add(fp_mul(fp_sqr(out.x), out.x), fp_from_u64(4));
    if (!fp_eq(fp_sqr(out.y), rhs)) return ZK_ERR_NOT_ON_CURVE;
    if (!pt_mul<G1, Fp>(out, HP_R, 4).inf) return ZK_ERR_NOT_ON_CURVE;
    return ZK_OK;
}
static int g2_decode(G2& out, const uint8_t* b) {
    if (b[0] & 0x80) return ZK_ERR_ARG;
    if (b[0] & 0x40) { out.inf = true; out.x = f2_zero(); out.y = f2_zero(); return ZK_OK; }
    if (!fp_from_be(out.x.b, b) || !fp_from_be(out.x.a, b + 48) || !fp_from_be(out.y.b, b + 96) || !fp_from_be(out.y.a, b + 144)) return ZK_ERR_ARG;
    out.inf = false;
    const Fp four = fp_from_u64(4);
    const Fp2 rhs = f2_add(f2_mul(f2_mul(out.x, out.x), out.x), f2(four, four));
    if (!f2_eq(f2_mul(out.y, out.y), rhs)) return ZK_ERR_NOT_ON_CURVE;
    if (!pt_mul<G2, Fp2>(out, HP_R, 4).inf) return ZK_ERR_NOT_ON_CURVE;
    return ZK_OK;
}

// ---- square roots (p = 3 mod 4) for of_compressed_bytes_exn (curve.ml:199-212)
static bool fp_sqrt(Fp& out, const Fp& a) {
    out = fp_pow(a, HP_PP1D4, 6);
    return fp_eq(fp_sqr(out), a);
}
static bool f2_sqrt(Fp2& out, const Fp2& a) {
    if (fp_is_zero(a.b)) {
        Fp r;
        if (fp_sqrt(r, a.a)) { out = f2(r, fp_zero()); return true; }
        if (fp_sqrt(r, fp_neg(a.a))) { out = f2(fp_zero(), r); return true; }      // (r u)^2 = -r^2
        return false;
    }
    Fp s;
    if (!fp_sqrt(s, fp_add(fp_sqr(a.a), fp_sqr(a.b)))) return false;                 // the norm of a square is a square
    const Fp half = fp_inv(fp_from_u64(2));
    Fp x0;
    if (!fp_sqrt(x0, fp_mul(fp_add(a.a, s), half)) && !fp_sqrt(x0, fp_mul(fp_sub(a.a, s), half))) return false;
    const Fp x1 = fp_mul(a.b, fp_inv(fp_add(x0, x0)));
    out = f2(x0, x1);
    return f2_eq(f2_mul(out, out), a);
}
// canonical integer of a > (p - 1) / 2 ?
static bool fp_is_large(const Fp& a) {
    Fp one = fp_zero();
    one.l[0] = 1;
    const Fp r = fp_mul(a, one);
    for (int i = 5; i >= 0; i--) {
        if (r.l[i] > HP_PM1D2[i]) return true;
        if (r.l[i] < HP_PM1D2[i]) return false;
    }
    return false;
}

// Miller loop f_{|x|,Q}(P) over E(Fp12) with the untwisted Q, conjugated at the end (x < 0)
static Fp12 miller_loop(const G1& p, const G2& q) {
    if (p.inf || q.inf) return f12_one();
    const Fp12 px = f12_from_fp(p.x), py = f12_from_fp(p.y);
    const Fp12 w = Fp12{f6_zero(), f6_one()};
    const Fp12 w2 = f12_mul(w, w), w3 = f12_mul(w2, w);
    const Fp12 qx = f12_mul(f12_from_fp2(q.x), f12_inv(w2)), qy = f12_mul(f12_from_fp2(q.y), f12_inv(w3));
    Fp12 tx = qx, ty = qy, f = f12_one();
    const Fp12 two = f12_from_fp(fp_from_u64(2)), three = f12_from_fp(fp_from_u64(3));
    int top = 63;
    while (!((HP_BLS_X >> top) & 1)) top--;
    for (int i = top - 1; i >= 0; i--) {
        Fp12 lam = f12_mul(f12_mul(three, f12_mul(tx, tx)), f12_inv(f12_mul(two, ty)));
        Fp12 line = f12_sub(f12_sub(py, ty), f12_mul(lam, f12_sub(px, tx)));
        f = f12_mul(f12_mul(f, f), line);
        Fp12 nx = f12_sub(f12_sub(f12_mul(lam, lam), tx), tx);
        Fp12 ny = f12_sub(f12_mul(lam, f12_sub(tx, nx)), ty);
        tx = nx; ty = ny;
        if ((HP_BLS_X >> i) & 1) {
            lam = f12_mul(f12_sub(qy, ty), f12_inv(f12_sub(qx, tx)));
            line = f12_sub(f12_sub(py, ty), f12_mul(lam, f12_sub(px, tx)));
            f = f12_mul(f, line);
            nx = f12_sub(f12_sub(f12_mul(lam, lam), tx), qx);
            ny = f12_sub(f12_mul(lam, f12_sub(tx, nx)), ty);
            tx = nx; ty = ny;
        }
    }
    return f12_conj(f);
}
static inline Fp12 final_exp(const Fp12& f) { return f12_pow(f, HP_FEXP, HP_FEXP_BITS); }

static void gt_to_bytes(uint8_t* out, const Fp12& g) {
    const Fp2* c[6] = {&g.c0.c0, &g.c0.c1, &g.c0.c2, &g.c1.c0, &g.c1.c1, &g.c1.c2};
    for (int i = 0; i < 6; i++) {
        fp_to_be(out + 96 * i, c[i]->a);
        fp_to_be(out + 96 * i + 48, c[i]->b);
    }
}
}  // namespace hp

using namespace zk;
// ---- whole verifiers on the host (scalars: canonical 32-byte little-endian Fr, as everywhere else)
static int fr_limbs(uint64_t out[4], const uint8_t* b) {
    for (int i = 0; i < 4; i++) {
        uint64_t w = 0;
        for (int k = 7; k >= 0; k--) w = (w << 8) | b[8 * i + k];
        out[i] = w;
    }
    for (int i = 3; i >= 0; i--) {
        if (out[i] < HP_R[i]) return ZK_OK;
        if (out[i] > HP_R[i]) break;
    }
    ZK_FAIL(ZK_ERR_SCALAR_RANGE, "verify: public input >= r");
}
static hp::G1 g1_neg(hp::G1 p) { if (!p.inf) p.y = hp::fp_neg(p.y); return p; }
// sum_k c_k * P_k over decoded points (G.dot, curve.ml:91-103, on a handful of public inputs)
template <class PT, class F> static int dot_host(PT& acc, const std::vector<PT>& pts, const uint8_t* scalars) {
    acc.inf = true;
    for (size_t k = 0; k < pts.size(); k++) {
        uint64_t c[4];
        ZKCHK(fr_limbs(c, scalars + 32 * k));
        acc = hp::pt_add<PT, F>(acc, hp::pt_mul<PT, F>(pts[k], c, 4));
    }
    return ZK_OK;
}
#define HP_DECODE1(var, ptr) hp::G1 var; { int rc_ = hp::g1_decode(var, ptr); if (rc_) ZK_FAIL(rc_, "verify: bad G1 point"); }
#define HP_DECODE2(var, ptr) hp::G2 var; { int rc_ = hp::g2_decode(var, ptr); if (rc_) ZK_FAIL(rc_, "verify: bad G2 point"); }

extern "C" {

int zk_pairing_product(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, uint8_t gt_out[576]) {
    if ((n && (!g1_points || !g2_points)) || !gt_out) ZK_FAIL(ZK_ERR_ARG, "zk_pairing_product: null argument");
    hp::Fp12 f = hp::f12_one();
    for (size_t i = 0; i < n; i++) {
        hp::G1 p;
        hp::G2 q;
        int rc = hp::g1_decode(p, g1_points + 96 * i);
        if (rc) ZK_FAIL(rc, "zk_pairing_product: bad G1 point (encoding, curve or subgroup)");
        rc = hp::g2_decode(q, g2_points + 192 * i);
        if (rc) ZK_FAIL(rc, "zk_pairing_product: bad G2 point (encoding, curve or subgroup)");
        f = hp::f12_mul(f, hp::miller_loop(p, q));
    }
    hp::gt_to_bytes(gt_out, hp::final_exp(f));
    return ZK_OK;
}
int zk_pairing_check(const uint8_t* g1_points, const uint8_t* g2_points, size_t n, int* is_one) {
    if (!is_one) ZK_FAIL(ZK_ERR_ARG, "zk_pairing_check: null argument");
    uint8_t gt[576], one[576];
    ZKCHK(zk_pairing_product(g1_points, g2_points, n, gt));
    hp::gt_to_bytes(one, hp::f12_one());
    *is_one = memcmp(gt, one, 576) == 0 ? 1 : 0;
    return ZK_OK;
}

// of_compressed_bytes_exn (curve.ml:199-212): ZCash compressed -> uncompressed, with curve and subgroup checks
int zk_g1_decompress(const uint8_t in[48], uint8_t out[96]) {
    if (!in || !out) ZK_FAIL(ZK_ERR_ARG, "zk_g1_decompress: null");
    if (!(in[0] & 0x80)) ZK_FAIL(ZK_ERR_ARG, "zk_g1_decompress: compression flag not set");
    memset(out, 0, 96);
    if (in[0] & 0x40) { out[0] = 0x40; return ZK_OK; }
    uint8_t xb[48];
    memcpy(xb, in, 48);
    xb[0] &= 0x1f;
    hp::G1 p;
    p.inf = false;
    if (!hp::fp_from_be(p.x, xb)) ZK_FAIL(ZK_ERR_ARG, "zk_g1_decompress: x >= p");
    if (!hp::fp_sqrt(p.y, hp::fp_add(hp::fp_mul(hp::fp_sqr(p.x), p.x), hp::fp_from_u64(4)))) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "zk_g1_decompress: x is not on the curve");
    if (hp::fp_is_large(p.y) != ((in[0] & 0x20) != 0)) p.y = hp::fp_neg(p.y);
    if (!hp::pt_mul<hp::G1, hp::Fp>(p, HP_R, 4).inf) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "zk_g1_decompress: not in the prime-order subgroup");
    hp::fp_to_be(out, p.x);
    hp::fp_to_be(out + 48, p.y);
    return ZK_OK;
}
int zk_g2_decompress(const uint8_t in[96], uint8_t out[192]) {
    if (!in || !out) ZK_FAIL(ZK_ERR_ARG, "zk_g2_decompress: null");
    if (!(in[0] & 0x80)) ZK_FAIL(ZK_ERR_ARG, "zk_g2_decompress: compression flag not set");
    memset(out, 0, 192);
    if (in[0] & 0x40) { out[0] = 0x40; return ZK_OK; }
    uint8_t xb[96];
    memcpy(xb, in, 96);
    xb[0] &= 0x1f;
    hp::G2 p;
    p.inf = false;
    if (!hp::fp_from_be(p.x.b, xb) || !hp::fp_from_be(p.x.a, xb + 48)) ZK_FAIL(ZK_ERR_ARG, "zk_g2_decompress: coordinate >= p");
    const hp::Fp four = hp::fp_from_u64(4);
    if (!hp::f2_sqrt(p.y, hp::f2_add(hp::f2_mul(hp::f2_mul(p.x, p.x), p.x), hp::f2(four, four)))) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "zk_g2_decompress: x is not on the curve");
    const bool large = hp::fp_is_zero(p.y.b) ? hp::fp_is_large(p.y.a) : hp::fp_is_large(p.y.b);
    if (large != ((in[0] & 0x20) != 0)) p.y = hp::f2_neg(p.y);
    if (!hp::pt_mul<hp::G2, hp::Fp2>(p, HP_R, 4).inf) ZK_FAIL(ZK_ERR_NOT_ON_CURVE, "zk_g2_decompress: not in the prime-order subgroup");
    hp::fp_to_be(out, p.x.b);
    hp::fp_to_be(out + 48, p.x.a);
    hp::fp_to_be(out + 96, p.y.b);
    hp::fp_to_be(out + 144, p.y.a);
    return ZK_OK;
}
// groth16.ml:163-173:  e(A, B) = ab * e(sum_k w_k ltgm_io_k, gm) * e(C, d)
int zk_groth16_verify(const uint8_t ab[576], const uint8_t* ltgm_io, const uint8_t* io_scalars, size_t n_io, const uint8_t gm[192],
                      const uint8_t d[192], const uint8_t proof[384], int* ok) {
    if (!ab || !gm || !d || !proof || !ok || (n_io && (!ltgm_io || !io_scalars))) ZK_FAIL(ZK_ERR_ARG, "zk_groth16_verify: null argument");
    *ok = 0;
    HP_DECODE1(A, proof) HP_DECODE2(B, proof + 96) HP_DECODE1(Cc, proof + 288)
    HP_DECODE2(GM, gm) HP_DECODE2(D, d)
    std::vector<hp::G1> pts(n_io);
    for (size_t k = 0; k < n_io; k++) {
        int rc = hp::g1_decode(pts[k], ltgm_io + 96 * k);
        if (rc) ZK_FAIL(rc, "verify: bad G1 point in the key");
    }
    hp::G1 acc;
    ZKCHK((dot_host<hp::G1, hp::Fp>(acc, pts, io_scalars)));
    hp::Fp12 f = hp::miller_loop(A, B);
    f = hp::f12_mul(f, hp::miller_loop(g1_neg(acc), GM));
    f = hp::f12_mul(f, hp::miller_loop(g1_neg(Cc), D));
    uint8_t gt[576];
    hp::gt_to_bytes(gt, hp::final_exp(f));
    *ok = memcmp(gt, ab, 576) == 0 ? 1 : 0;
    return ZK_OK;
}
// Verify.f, pinocchio.ml:254-420.  vk_g1 = one | aw | bgm | vv_io[n_io] | yy_io[n_io];
// vk_g2 = one2 | av | ay | gm2 | bgm2 | yt | ww_io[n_io]  (the verification key of pinocchio.ml:62-75, flattened).
int zk_pinocchio_verify(const uint8_t* vk_g1, const uint8_t* vk_g2, const uint8_t* io_scalars, size_t n_io, const uint8_t proof[960], int* ok) {
    if (!vk_g1 || !vk_g2 || !proof || !ok || (n_io && !io_scalars)) ZK_FAIL(ZK_ERR_ARG, "zk_pinocchio_verify: null argument");
    *ok = 0;
    HP_DECODE1(one, vk_g1) HP_DECODE1(aw, vk_g1 + 96) HP_DECODE1(bgm, vk_g1 + 192)
    HP_DECODE2(one2, vk_g2) HP_DECODE2(av, vk_g2 + 192) HP_DECODE2(ay, vk_g2 + 384) HP_DECODE2(gm2, vk_g2 + 576)
    HP_DECODE2(bgm2, vk_g2 + 768) HP_DECODE2(yt, vk_g2 + 960)
    std::vector<hp::G1> vv_io(n_io), yy_io(n_io);
    std::vector<hp::G2> ww_io(n_io);
    for (size_t k = 0; k < n_io; k++) {
        int rc = hp::g1_decode(vv_io[k], vk_g1 + 96 * (3 + k));
        if (!rc) rc = hp::g1_decode(yy_io[k], vk_g1 + 96 * (3 + n_io + k));
        if (!rc) rc = hp::g2_decode(ww_io[k], vk_g2 + 192 * (6 + k));
        if (rc) ZK_FAIL(rc, "verify: bad point in the key");
    }
    HP_DECODE1(vv, proof) HP_DECODE2(ww, proof + 96) HP_DECODE1(yy, proof + 288) HP_DECODE1(h, proof + 384)
    HP_DECODE1(vavv, proof + 480) HP_DECODE2(waww, proof + 576) HP_DECODE1(yayy, proof + 768) HP_DECODE1(bvwy, proof + 864)
    auto is_one = [](const hp::Fp12& f) { return hp::f12_eq(hp::final_exp(f), hp::f12_one()); };
    using hp::f12_mul; using hp::miller_loop;
    bool good = true;
    good &= is_one(f12_mul(miller_loop(vv, av), miller_loop(g1_neg(vavv), one2)));                      // :285
    good &= is_one(f12_mul(miller_loop(aw, ww), miller_loop(g1_neg(one), waww)));                       // :298
    good &= is_one(f12_mul(miller_loop(yy, ay), miller_loop(g1_neg(yayy), one2)));                      // :311
    good &= is_one(f12_mul(f12_mul(miller_loop(bvwy, gm2), miller_loop(g1_neg(vv), bgm2)),
                           f12_mul(miller_loop(g1_neg(bgm), ww), miller_loop(g1_neg(yy), bgm2))));       // :361-366
    hp::G1 vio, yio;
    hp::G2 wio;
    ZKCHK((dot_host<hp::G1, hp::Fp>(vio, vv_io, io_scalars)));
    ZKCHK((dot_host<hp::G1, hp::Fp>(yio, yy_io, io_scalars)));
    ZKCHK((dot_host<hp::G2, hp::Fp2>(wio, ww_io, io_scalars)));
    const hp::G1 vsum = hp::pt_add<hp::G1, hp::Fp>(vio, vv), ysum = hp::pt_add<hp::G1, hp::Fp>(yio, yy);
    const hp::G2 wsum = hp::pt_add<hp::G2, hp::Fp2>(wio, ww);
    good &= is_one(f12_mul(f12_mul(miller_loop(vsum, wsum), miller_loop(g1_neg(ysum), one2)), miller_loop(g1_neg(h), yt)));   // :418-420
    *ok = good ? 1 : 0;
    return ZK_OK;
}
}
