/* TEST INFRASTRUCTURE ONLY -- see bls12_381.h.  PARITY UNPINNED by the reference. */
#include "bls12_381.h"
#include <string.h>

typedef unsigned __int128 u128;

/* ------------------------------------------------------------------ generic Montgomery (n 64-bit limbs) */
static int ge_n(const uint64_t *a, const uint64_t *b, int n) {
    for (int i = n - 1; i >= 0; i--) {
        if (a[i] > b[i]) return 1;
        if (a[i] < b[i]) return 0;
    }
    return 1;
}
static uint64_t add_n(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    u128 c = 0;
    for (int i = 0; i < n; i++) { c += (u128)a[i] + b[i]; r[i] = (uint64_t)c; c >>= 64; }
    return (uint64_t)c;
}
static uint64_t sub_n(uint64_t *r, const uint64_t *a, const uint64_t *b, int n) {
    uint64_t borrow = 0;
    for (int i = 0; i < n; i++) {
        u128 d = (u128)a[i] - b[i] - borrow;
        r[i] = (uint64_t)d; borrow = (uint64_t)(d >> 64) & 1;
    }
    return borrow;
}
static void mod_add(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *p, int n) {
    uint64_t t[6];
    uint64_t c = add_n(t, a, b, n);
    if (c || ge_n(t, p, n)) sub_n(t, t, p, n);
    memcpy(r, t, 8 * n);
}
static void mod_sub(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *p, int n) {
    uint64_t t[6];
    if (sub_n(t, a, b, n)) add_n(t, t, p, n);
    memcpy(r, t, 8 * n);
}
/* CIOS Montgomery multiplication */
static void mont_mul(uint64_t *r, const uint64_t *a, const uint64_t *b, const uint64_t *p, uint64_t inv, int n) {
    uint64_t t[8] = {0};
    for (int i = 0; i < n; i++) {
        u128 c = 0;
        for (int j = 0; j < n; j++) {
            c += (u128)a[j] * b[i] + t[j];
            t[j] = (uint64_t)c; c >>= 64;
        }
        c += t[n]; t[n] = (uint64_t)c; t[n + 1] = (uint64_t)(c >> 64);
        uint64_t m = t[0] * inv;
        c = (u128)m * p[0] + t[0]; c >>= 64;
        for (int j = 1; j < n; j++) {
            c += (u128)m * p[j] + t[j];
            t[j - 1] = (uint64_t)c; c >>= 64;
        }
        c += t[n]; t[n - 1] = (uint64_t)c;
        t[n] = t[n + 1] + (uint64_t)(c >> 64);
    }
    if (t[n] || ge_n(t, p, n)) sub_n(t, t, p, n);
    memcpy(r, t, 8 * n);
}

/* ------------------------------------------------------------------ Fr */
static const uint64_t FR_P[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
static const uint64_t FR_R2[4] = {0xc999e990f3f29c6dULL, 0x2b6cedcb87925c23ULL, 0x05d314967254398fULL, 0x0748d9d99f59ff11ULL};
#define FR_INV 0xfffffffeffffffffULL
const fr_t FR_ZERO = {{0, 0, 0, 0}};
const fr_t FR_ONE = {{0x00000001fffffffeULL, 0x5884b7fa00034802ULL, 0x998c4fefecbc4ff5ULL, 0x1824b159acc5056fULL}};

void fr_add(fr_t *r, const fr_t *a, const fr_t *b) { mod_add(r->l, a->l, b->l, FR_P, 4); }
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b) { mod_sub(r->l, a->l, b->l, FR_P, 4); }
void fr_neg(fr_t *r, const fr_t *a) { fr_t z = FR_ZERO; mod_sub(r->l, z.l, a->l, FR_P, 4); }
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b) { mont_mul(r->l, a->l, b->l, FR_P, FR_INV, 4); }
int fr_is_zero(const fr_t *a) { return (a->l[0] | a->l[1] | a->l[2] | a->l[3]) == 0; }
int fr_eq(const fr_t *a, const fr_t *b) { return memcmp(a, b, sizeof(fr_t)) == 0; }
void fr_from_bytes(fr_t *r, const uint8_t in[32]) {
    fr_t t, r2;
    for (int i = 0; i < 4; i++) {
        uint64_t v = 0;
        for (int j = 7; j >= 0; j--) v = (v << 8) | in[8 * i + j];
        t.l[i] = v;
    }
    memcpy(r2.l, FR_R2, 32);
    fr_mul(r, &t, &r2);
}
void fr_to_bytes(uint8_t out[32], const fr_t *a) {
    fr_t one = {{1, 0, 0, 0}}, t;
    fr_mul(&t, a, &one);
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 8; j++) out[8 * i + j] = (uint8_t)(t.l[i] >> (8 * j));
}
void fr_from_u64(fr_t *r, uint64_t v) {
    fr_t t = {{v, 0, 0, 0}}, r2;
    memcpy(r2.l, FR_R2, 32);
    fr_mul(r, &t, &r2);
}
static void fr_pow_limbs(fr_t *r, const fr_t *a, const uint64_t *e, int n) {
    fr_t acc = FR_ONE, base = *a;
    for (int i = 0; i < 64 * n; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fr_mul(&acc, &acc, &base);
        fr_mul(&base, &base, &base);
    }
    *r = acc;
}
void fr_pow_u64(fr_t *r, const fr_t *a, uint64_t e) { fr_pow_limbs(r, a, &e, 1); }
void fr_inv(fr_t *r, const fr_t *a) {
    uint64_t e[4];
    uint64_t two[4] = {2, 0, 0, 0};
    sub_n(e, FR_P, two, 4);
    fr_pow_limbs(r, a, e, 4);
}
void fr_omega(fr_t *r) {
    /* (r-1) / 2^32 */
    uint64_t e[4];
    uint64_t one[4] = {1, 0, 0, 0};
    sub_n(e, FR_P, one, 4);
    for (int i = 0; i < 4; i++) e[i] = (e[i] >> 32) | (i < 3 ? e[i + 1] << 32 : 0);
    fr_t five;
    fr_from_u64(&five, 5);
    fr_pow_limbs(r, &five, e, 4);
}

/* ------------------------------------------------------------------ Fp */
static const uint64_t FP_P[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                                 0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
static const uint64_t FP_R2[6] = {0xf4df1f341c341746ULL, 0x0a76e6a609d104f1ULL, 0x8de5476c4c95b6d5ULL,
                                  0x67eb88a9939d83c0ULL, 0x9a793e85b519952dULL, 0x11988fe592cae3aaULL};
static const uint64_t FP_R1[6] = {0x760900000002fffdULL, 0xebf4000bc40c0002ULL, 0x5f48985753c758baULL,
                                  0x77ce585370525745ULL, 0x5c071a97a256ec6dULL, 0x15f65ec3fa80e493ULL};
static const uint64_t FP_PM1H[6] = {0xdcff7fffffffd555ULL, 0x0f55ffff58a9ffffULL, 0xb39869507b587b12ULL,
                                    0xb23ba5c279c2895fULL, 0x258dd3db21a5d66bULL, 0x0d0088f51cbff34dULL};
#define FP_INV 0x89f3fffcfffcfffdULL

void fp_add(fp_t *r, const fp_t *a, const fp_t *b) { mod_add(r->l, a->l, b->l, FP_P, 6); }
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b) { mod_sub(r->l, a->l, b->l, FP_P, 6); }
void fp_neg(fp_t *r, const fp_t *a) { fp_t z; memset(&z, 0, sizeof z); mod_sub(r->l, z.l, a->l, FP_P, 6); }
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b) { mont_mul(r->l, a->l, b->l, FP_P, FP_INV, 6); }
int fp_is_zero(const fp_t *a) { uint64_t o = 0; for (int i = 0; i < 6; i++) o |= a->l[i]; return o == 0; }
int fp_eq(const fp_t *a, const fp_t *b) { return memcmp(a, b, sizeof(fp_t)) == 0; }
static void fp_one(fp_t *r) { memcpy(r->l, FP_R1, 48); }
static void fp_canon(uint64_t out[6], const fp_t *a) {
    fp_t one, t;
    memset(&one, 0, sizeof one); one.l[0] = 1;
    fp_mul(&t, a, &one);
    memcpy(out, t.l, 48);
}
void fp_from_be(fp_t *r, const uint8_t in[48]) {
    fp_t t, r2;
    for (int i = 0; i < 6; i++) {
        uint64_t v = 0;
        for (int j = 0; j < 8; j++) v = (v << 8) | in[48 - 8 * (i + 1) + j];
        t.l[i] = v;
    }
    memcpy(r2.l, FP_R2, 48);
    fp_mul(r, &t, &r2);
}
void fp_to_be(uint8_t out[48], const fp_t *a) {
    uint64_t c[6];
    fp_canon(c, a);
    for (int i = 0; i < 6; i++)
        for (int j = 0; j < 8; j++) out[48 - 8 * (i + 1) + j] = (uint8_t)(c[i] >> (8 * (7 - j)));
}
int fp_is_lex_largest(const fp_t *a) {
    uint64_t c[6];
    fp_canon(c, a);
    /* c > (p-1)/2 */
    return ge_n(c, FP_PM1H, 6) && memcmp(c, FP_PM1H, 48) != 0;
}
void fp_inv(fp_t *r, const fp_t *a) {
    uint64_t e[6], two[6] = {2, 0, 0, 0, 0, 0};
    sub_n(e, FP_P, two, 6);
    fp_t acc, base = *a;
    fp_one(&acc);
    for (int i = 0; i < 384; i++) {
        if ((e[i / 64] >> (i % 64)) & 1) fp_mul(&acc, &acc, &base);
        fp_mul(&base, &base, &base);
    }
    *r = acc;
}

/* ------------------------------------------------------------------ Fp2 */
static void fp2_add(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_add(&r->c0, &a->c0, &b->c0); fp_add(&r->c1, &a->c1, &b->c1); }
static void fp2_sub(fp2_t *r, const fp2_t *a, const fp2_t *b) { fp_sub(&r->c0, &a->c0, &b->c0); fp_sub(&r->c1, &a->c1, &b->c1); }
static void fp2_neg(fp2_t *r, const fp2_t *a) { fp_neg(&r->c0, &a->c0); fp_neg(&r->c1, &a->c1); }
static void fp2_mul(fp2_t *r, const fp2_t *a, const fp2_t *b) {
    fp_t t0, t1, t2, t3;
    fp_mul(&t0, &a->c0, &b->c0);
    fp_mul(&t1, &a->c1, &b->c1);
    fp_mul(&t2, &a->c0, &b->c1);
    fp_mul(&t3, &a->c1, &b->c0);
    fp_sub(&r->c0, &t0, &t1);
    fp_add(&r->c1, &t2, &t3);
}
static int fp2_is_zero(const fp2_t *a) { return fp_is_zero(&a->c0) && fp_is_zero(&a->c1); }
static int fp2_eq(const fp2_t *a, const fp2_t *b) { return fp_eq(&a->c0, &b->c0) && fp_eq(&a->c1, &b->c1); }
static void fp2_inv(fp2_t *r, const fp2_t *a) {
    fp_t n, t, d;
    fp_mul(&n, &a->c0, &a->c0);
    fp_mul(&t, &a->c1, &a->c1);
    fp_add(&n, &n, &t);
    fp_inv(&d, &n);
    fp_mul(&r->c0, &a->c0, &d);
    fp_mul(&t, &a->c1, &d);
    fp_neg(&r->c1, &t);
}

/* ------------------------------------------------------------------ curve law, generic over the field via macros */
#define DEFINE_CURVE(G, F, f_add, f_sub, f_neg, f_mul, f_is_zero, f_eq)                           \
    void G##_set_inf(G##_t *r) { memset(r, 0, sizeof *r); }                                       \
    int G##_is_inf(const G##_t *a) { return f_is_zero(&a->z); }                                   \
    void G##_neg(G##_t *r, const G##_t *a) { *r = *a; f_neg(&r->y, &a->y); }                      \
    void G##_dbl(G##_t *r, const G##_t *p) {                                                      \
        /* a = 0: A=X^2 B=Y^2 C=B^2 D=2((X+B)^2-A-C) E=3A F=E^2 X3=F-2D Y3=E(D-X3)-8C Z3=2YZ */   \
        if (G##_is_inf(p) || f_is_zero(&p->y)) { G##_set_inf(r); return; }                        \
        F A, B, C, D, E, FF, t, X3, Y3, Z3;                                                       \
        f_mul(&A, &p->x, &p->x); f_mul(&B, &p->y, &p->y); f_mul(&C, &B, &B);                      \
        f_add(&t, &p->x, &B); f_mul(&t, &t, &t); f_sub(&t, &t, &A); f_sub(&t, &t, &C);            \
        f_add(&D, &t, &t);                                                                        \
        f_add(&E, &A, &A); f_add(&E, &E, &A);                                                     \
        f_mul(&FF, &E, &E);                                                                       \
        f_sub(&X3, &FF, &D); f_sub(&X3, &X3, &D);                                                 \
        f_sub(&t, &D, &X3); f_mul(&Y3, &E, &t);                                                   \
        f_add(&C, &C, &C); f_add(&C, &C, &C); f_add(&C, &C, &C);                                  \
        f_sub(&Y3, &Y3, &C);                                                                      \
        f_mul(&Z3, &p->y, &p->z); f_add(&Z3, &Z3, &Z3);                                           \
        r->x = X3; r->y = Y3; r->z = Z3;                                                          \
    }                                                                                             \
    void G##_add(G##_t *r, const G##_t *p, const G##_t *q) {                                      \
        if (G##_is_inf(p)) { *r = *q; return; }                                                   \
        if (G##_is_inf(q)) { *r = *p; return; }                                                   \
        F Z1Z1, Z2Z2, U1, U2, S1, S2, H, RR, t, HH, HHH, V, X3, Y3, Z3;                           \
        f_mul(&Z1Z1, &p->z, &p->z); f_mul(&Z2Z2, &q->z, &q->z);                                   \
        f_mul(&U1, &p->x, &Z2Z2); f_mul(&U2, &q->x, &Z1Z1);                                       \
        f_mul(&t, &q->z, &Z2Z2); f_mul(&S1, &p->y, &t);                                           \
        f_mul(&t, &p->z, &Z1Z1); f_mul(&S2, &q->y, &t);                                           \
        f_sub(&H, &U2, &U1); f_sub(&RR, &S2, &S1);                                                \
        if (f_is_zero(&H)) {                                                                      \
            if (f_is_zero(&RR)) { G##_dbl(r, p); return; }                                        \
            G##_set_inf(r); return;                                                               \
        }                                                                                         \
        f_mul(&HH, &H, &H); f_mul(&HHH, &H, &HH); f_mul(&V, &U1, &HH);                            \
        f_mul(&X3, &RR, &RR); f_sub(&X3, &X3, &HHH); f_sub(&X3, &X3, &V); f_sub(&X3, &X3, &V);    \
        f_sub(&t, &V, &X3); f_mul(&Y3, &RR, &t); f_mul(&t, &S1, &HHH); f_sub(&Y3, &Y3, &t);       \
        f_mul(&Z3, &p->z, &q->z); f_mul(&Z3, &Z3, &H);                                            \
        r->x = X3; r->y = Y3; r->z = Z3;                                                          \
    }                                                                                             \
    void G##_mul(G##_t *r, const G##_t *a, const fr_t *k) {                                       \
        uint8_t kb[32];                                                                           \
        fr_to_bytes(kb, k);                                                                       \
        G##_t acc, base = *a;                                                                     \
        G##_set_inf(&acc);                                                                        \
        for (int i = 254; i >= 0; i--) {                                                          \
            G##_dbl(&acc, &acc);                                                                  \
            if ((kb[i / 8] >> (i % 8)) & 1) G##_add(&acc, &acc, &base);                           \
        }                                                                                         \
        *r = acc;                                                                                 \
    }                                                                                             \
    int G##_eq(const G##_t *p, const G##_t *q) {                                                  \
        if (G##_is_inf(p) || G##_is_inf(q)) return G##_is_inf(p) && G##_is_inf(q);               \
        F Z1Z1, Z2Z2, a, b, t;                                                                    \
        f_mul(&Z1Z1, &p->z, &p->z); f_mul(&Z2Z2, &q->z, &q->z);                                   \
        f_mul(&a, &p->x, &Z2Z2); f_mul(&b, &q->x, &Z1Z1);                                         \
        if (!f_eq(&a, &b)) return 0;                                                              \
        f_mul(&t, &q->z, &Z2Z2); f_mul(&a, &p->y, &t);                                            \
        f_mul(&t, &p->z, &Z1Z1); f_mul(&b, &q->y, &t);                                            \
        return f_eq(&a, &b);                                                                      \
    }

DEFINE_CURVE(g1, fp_t, fp_add, fp_sub, fp_neg, fp_mul, fp_is_zero, fp_eq)
DEFINE_CURVE(g2, fp2_t, fp2_add, fp2_sub, fp2_neg, fp2_mul, fp2_is_zero, fp2_eq)

/* ------------------------------------------------------------------ generators + serialization */
static void hex_to_be48(uint8_t out[48], const char *hex) {
    for (int i = 0; i < 48; i++) {
        unsigned v = 0;
        for (int j = 0; j < 2; j++) {
            char c = hex[2 * i + j];
            v = v * 16 + (unsigned)(c <= '9' ? c - '0' : c - 'a' + 10);
        }
        out[i] = (uint8_t)v;
    }
}
static const char *G1X = "17f1d3a73197d7942695638c4fa9ac0fc3688c4f9774b905a14e3a3f171bac586c55e83ff97a1aeffb3af00adb22c6bb";
static const char *G1Y = "08b3f481e3aaa0f1a09e30ed741d8ae4fcf5e095d5d00af600db18cb2c04b3edd03cc744a2888ae40caa232946c5e7e1";
static const char *G2X0 = "024aa2b2f08f0a91260805272dc51051c6e47ad4fa403b02b4510b647ae3d1770bac0326a805bbefd48056c8c121bdb8";
static const char *G2X1 = "13e02b6052719f607dacd3a088274f65596bd0d09920b61ab5da61bbdc7f5049334cf11213945d57e5ac7d055d042b7e";
static const char *G2Y0 = "0ce5d527727d6e118cc9cdc6da2e351aadfd9baa8cbdd3a76d429a695160d12c923ac9cc3baca289e193548608b82801";
static const char *G2Y1 = "0606c4a02ea734cc32acd2b02bc28b99cb3e287e85a763af267492ab572e99ab3f370d275cec1da1aaa9075ff05f79be";

void g1_generator(g1_t *r) {
    uint8_t b[48];
    hex_to_be48(b, G1X); fp_from_be(&r->x, b);
    hex_to_be48(b, G1Y); fp_from_be(&r->y, b);
    fp_one(&r->z);
}
void g2_generator(g2_t *r) {
    uint8_t b[48];
    hex_to_be48(b, G2X0); fp_from_be(&r->x.c0, b);
    hex_to_be48(b, G2X1); fp_from_be(&r->x.c1, b);
    hex_to_be48(b, G2Y0); fp_from_be(&r->y.c0, b);
    hex_to_be48(b, G2Y1); fp_from_be(&r->y.c1, b);
    fp_one(&r->z.c0); memset(&r->z.c1, 0, sizeof(fp_t));
}

static void g1_affine(fp_t *x, fp_t *y, const g1_t *a) {
    fp_t zi, zi2, zi3;
    fp_inv(&zi, &a->z);
    fp_mul(&zi2, &zi, &zi); fp_mul(&zi3, &zi2, &zi);
    fp_mul(x, &a->x, &zi2); fp_mul(y, &a->y, &zi3);
}
static void g2_affine(fp2_t *x, fp2_t *y, const g2_t *a) {
    fp2_t zi, zi2, zi3;
    fp2_inv(&zi, &a->z);
    fp2_mul(&zi2, &zi, &zi); fp2_mul(&zi3, &zi2, &zi);
    fp2_mul(x, &a->x, &zi2); fp2_mul(y, &a->y, &zi3);
}
void g1_to_bytes(uint8_t out[96], const g1_t *a) {
    if (g1_is_inf(a)) { memset(out, 0, 96); out[0] = 0x40; return; }
    fp_t x, y;
    g1_affine(&x, &y, a);
    fp_to_be(out, &x); fp_to_be(out + 48, &y);
}
void g1_compress(uint8_t out[48], const g1_t *a) {
    if (g1_is_inf(a)) { memset(out, 0, 48); out[0] = 0xC0; return; }
    fp_t x, y;
    g1_affine(&x, &y, a);
    fp_to_be(out, &x);
    out[0] |= 0x80;
    if (fp_is_lex_largest(&y)) out[0] |= 0x20;
}
int g1_from_bytes(g1_t *r, const uint8_t in[96]) {
    if (in[0] & 0x40) { g1_set_inf(r); return 0; }
    fp_from_be(&r->x, in); fp_from_be(&r->y, in + 48); fp_one(&r->z);
    fp_t l, rr, four;
    fp_mul(&l, &r->y, &r->y);
    fp_mul(&rr, &r->x, &r->x); fp_mul(&rr, &rr, &r->x);
    fp_one(&four); fp_add(&four, &four, &four); fp_add(&four, &four, &four);
    fp_add(&rr, &rr, &four);
    return fp_eq(&l, &rr) ? 0 : -1;
}
void g2_to_bytes(uint8_t out[192], const g2_t *a) {
    if (g2_is_inf(a)) { memset(out, 0, 192); out[0] = 0x40; return; }
    fp2_t x, y;
    g2_affine(&x, &y, a);
    fp_to_be(out, &x.c1); fp_to_be(out + 48, &x.c0);
    fp_to_be(out + 96, &y.c1); fp_to_be(out + 144, &y.c0);
}
void g2_compress(uint8_t out[96], const g2_t *a) {
    if (g2_is_inf(a)) { memset(out, 0, 96); out[0] = 0xC0; return; }
    fp2_t x, y;
    g2_affine(&x, &y, a);
    fp_to_be(out, &x.c1); fp_to_be(out + 48, &x.c0);
    out[0] |= 0x80;
    /* y lexicographically larger than -y: compare c1 first, then c0 */
    int larger = fp_is_zero(&y.c1) ? fp_is_lex_largest(&y.c0) : fp_is_lex_largest(&y.c1);
    if (larger) out[0] |= 0x20;
}
int g2_from_bytes(g2_t *r, const uint8_t in[192]) {
    if (in[0] & 0x40) { g2_set_inf(r); return 0; }
    fp_from_be(&r->x.c1, in); fp_from_be(&r->x.c0, in + 48);
    fp_from_be(&r->y.c1, in + 96); fp_from_be(&r->y.c0, in + 144);
    fp_one(&r->z.c0); memset(&r->z.c1, 0, sizeof(fp_t));
    fp2_t l, rr, b;
    fp2_mul(&l, &r->y, &r->y);
    fp2_mul(&rr, &r->x, &r->x); fp2_mul(&rr, &rr, &r->x);
    fp_one(&b.c0); fp_add(&b.c0, &b.c0, &b.c0); fp_add(&b.c0, &b.c0, &b.c0); b.c1 = b.c0;
    fp2_add(&rr, &rr, &b);
    return fp2_eq(&l, &rr) ? 0 : -1;
}
