/* TEST INFRASTRUCTURE ONLY -- the "fast CPU path (context)" of BASELINE.md 3.3 / SURVEY.md 8d item 3.
 *
 * NOT the reference's algorithm (that is zk_oracle.c: orc_groth16_prove, the per-variable fold with double-and-add
 * scalar multiplications and schoolbook polynomials) and NOT the product: a multi-threaded CPU prover that has the same
 * algorithmic freedom as the GPU path -- NTT convolutions for the witness polynomials, Pippenger bucket sums for the
 * three products -- so that a bench line can say how much of the GPU/CPU ratio is algorithm and how much is the MI355X.
 * It produces the proof of groth16.ml:123-161 (same bytes as the literal oracle, the trapdoor evaluator and the GPU)
 * from the Lagrange-form key (scope row f4; the key the GPU derives with zk_groth16_pk_derive_lagrange):
 *   g1 = alpha | delta | beta | [l_i(tau)] (n) | [lambda_t(tau) Z(tau)/delta] (n-1) | ltd_mid (n_mid)
 *   g2 = beta | delta | [l_i(tau)] (n)
 *
 *   a = L w, b = R w, c = O w                     values of v, w, y at 0..n-1            (QAP.ml:121-131)
 *   values at n..2n-2 through the Newton basis:   d = (y_i / i!) * ((-1)^j / j!),  p(t) = t! (d * (1/j!))_t
 *   h(t) = (a(t) b(t) - c(t)) / Z(t),  Z(t) = t! / (t-n)!                               (QAP.ml:132-135)
 *   A = <[1, r, 0, a...]; g1>,  B = <[1, s, b...]; g2>,  C = <[s, rs, r, s a + r b, h, w_mid]; g1>
 *
 * Field arithmetic: the Montgomery forms of bls12_381.c (so its (de)serialisers and inversions are reused), with
 * fixed-size fully unrolled multiplications; group law: XYZZ coordinates with mixed additions (the formulas the GPU
 * kernels use, EFD madd-2008-s / add-2008-s / dbl-2008-s-1); threads: pthreads, one slice of the non-zero scalars per
 * thread for every window (signed digits), butterfly ranges per thread with a barrier between NTT stages.
 * About 1.5-2x slower per field product than blst's assembly: a context figure, stated as such wherever it is printed.
 *
 * Only tests/ and bench.py's cpu_baseline leg may load this.  PARITY UNPINNED by the reference (see bls12_381.h).
 */
#define _GNU_SOURCE            /* pthread barriers under -std=c11 */
#include "bls12_381.h"
#include <pthread.h>
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))
#define INL static inline __attribute__((always_inline))
typedef unsigned __int128 u128;

static const uint64_t QP[6] = {0xb9feffffffffaaabULL, 0x1eabfffeb153ffffULL, 0x6730d2a0f6b0f624ULL,
                               0x64774b84f38512bfULL, 0x4b1ba7b6434bacd7ULL, 0x1a0111ea397fe69aULL};
#define QP_INV 0x89f3fffcfffcfffdULL
static const uint64_t RP[4] = {0xffffffff00000001ULL, 0x53bda402fffe5bfeULL, 0x3339d80809a1d805ULL, 0x73eda753299d7d48ULL};
#define RP_INV 0xfffffffeffffffffULL

/* ------------------------------------------------------------------ fixed-size Montgomery arithmetic */
#define DEFINE_FIELD(PFX, N, MOD, INV)                                                                        \
    INL int PFX##_geq(const uint64_t *a) {                                                                    \
        for (int i = N - 1; i >= 0; i--) { if (a[i] > MOD[i]) return 1; if (a[i] < MOD[i]) return 0; }        \
        return 1;                                                                                             \
    }                                                                                                         \
    INL void PFX##_subp(uint64_t *a) {                                                                        \
        uint64_t bw = 0;                                                                                      \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) {                                                 \
            u128 d = (u128)a[i] - MOD[i] - bw; a[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;              \
        }                                                                                                     \
    }                                                                                                         \
    INL void PFX##_add(uint64_t *r, const uint64_t *a, const uint64_t *b) {                                   \
        u128 c = 0; uint64_t t[N];                                                                            \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) { c += (u128)a[i] + b[i]; t[i] = (uint64_t)c; c >>= 64; } \
        if (PFX##_geq(t)) PFX##_subp(t);                     /* both moduli leave the top bit of the top limb free */ \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) r[i] = t[i];                                      \
    }                                                                                                         \
    INL void PFX##_sub(uint64_t *r, const uint64_t *a, const uint64_t *b) {                                   \
        uint64_t bw = 0, t[N];                                                                                \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) {                                                 \
            u128 d = (u128)a[i] - b[i] - bw; t[i] = (uint64_t)d; bw = (uint64_t)(d >> 64) & 1;                \
        }                                                                                                     \
        if (bw) { u128 c = 0;                                                                                 \
            _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) { c += (u128)t[i] + MOD[i]; t[i] = (uint64_t)c; c >>= 64; } } \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) r[i] = t[i];                                      \
    }                                                                                                         \
    INL __attribute__((unused)) int PFX##_is0(const uint64_t *a) { uint64_t o = 0; for (int i = 0; i < N; i++) o |= a[i]; return o == 0; } \
    INL void PFX##_mul(uint64_t *r, const uint64_t *a, const uint64_t *b) {          /* CIOS */               \
        uint64_t t[N + 2];                                                                                    \
        _Pragma("GCC unroll 8") for (int i = 0; i < N + 2; i++) t[i] = 0;                                     \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) {                                                 \
            u128 c = 0;                                                                                       \
            _Pragma("GCC unroll 6") for (int j = 0; j < N; j++) { c += (u128)a[j] * b[i] + t[j]; t[j] = (uint64_t)c; c >>= 64; } \
            c += t[N]; t[N] = (uint64_t)c; t[N + 1] = (uint64_t)(c >> 64);                                    \
            const uint64_t m = t[0] * INV;                                                                    \
            c = (u128)m * MOD[0] + t[0]; c >>= 64;                                                            \
            _Pragma("GCC unroll 6") for (int j = 1; j < N; j++) { c += (u128)m * MOD[j] + t[j]; t[j - 1] = (uint64_t)c; c >>= 64; } \
            c += t[N]; t[N - 1] = (uint64_t)c; t[N] = t[N + 1] + (uint64_t)(c >> 64);                         \
        }                                                                                                     \
        if (t[N] || PFX##_geq(t)) PFX##_subp(t);                                                              \
        _Pragma("GCC unroll 6") for (int i = 0; i < N; i++) r[i] = t[i];                                      \
    }

DEFINE_FIELD(q, 6, QP, QP_INV)
DEFINE_FIELD(rr, 4, RP, RP_INV)

/* Fr on fr_t */
INL void xr_mul(fr_t *r, const fr_t *a, const fr_t *b) { rr_mul(r->l, a->l, b->l); }
INL void xr_add(fr_t *r, const fr_t *a, const fr_t *b) { rr_add(r->l, a->l, b->l); }
INL void xr_sub(fr_t *r, const fr_t *a, const fr_t *b) { rr_sub(r->l, a->l, b->l); }

/* Fp / Fp2 behind one set of names per curve */
typedef fp_t F1;
static fp_t Q_ONE;                                   /* Montgomery one, set by init_one() before any thread starts */
static void init_one(void) { uint8_t ob[48] = {0}; ob[47] = 1; fp_from_be(&Q_ONE, ob); }
INL void F1_one(F1 *r) { *r = Q_ONE; }
INL void F1_mul(F1 *r, const F1 *a, const F1 *b) { q_mul(r->l, a->l, b->l); }
INL void F1_sqr(F1 *r, const F1 *a) { q_mul(r->l, a->l, a->l); }
INL void F1_add(F1 *r, const F1 *a, const F1 *b) { q_add(r->l, a->l, b->l); }
INL void F1_sub(F1 *r, const F1 *a, const F1 *b) { q_sub(r->l, a->l, b->l); }
INL void F1_neg(F1 *r, const F1 *a) { F1 z; memset(&z, 0, sizeof z); q_sub(r->l, z.l, a->l); }
INL int F1_is0(const F1 *a) { return q_is0(a->l); }
typedef fp2_t F2;
INL void F2_one(F2 *r) { memset(r, 0, sizeof *r); r->c0 = Q_ONE; }
INL void F2_add(F2 *r, const F2 *a, const F2 *b) { F1_add(&r->c0, &a->c0, &b->c0); F1_add(&r->c1, &a->c1, &b->c1); }
INL void F2_sub(F2 *r, const F2 *a, const F2 *b) { F1_sub(&r->c0, &a->c0, &b->c0); F1_sub(&r->c1, &a->c1, &b->c1); }
INL void F2_neg(F2 *r, const F2 *a) { F1_neg(&r->c0, &a->c0); F1_neg(&r->c1, &a->c1); }
INL int F2_is0(const F2 *a) { return F1_is0(&a->c0) && F1_is0(&a->c1); }
INL void F2_mul(F2 *r, const F2 *a, const F2 *b) {            /* Karatsuba, u^2 = -1 */
    F1 t0, t1, sa, sb, m;
    F1_mul(&t0, &a->c0, &b->c0); F1_mul(&t1, &a->c1, &b->c1);
    F1_add(&sa, &a->c0, &a->c1); F1_add(&sb, &b->c0, &b->c1); F1_mul(&m, &sa, &sb);
    F1_sub(&r->c0, &t0, &t1);
    F1_sub(&m, &m, &t0); F1_sub(&r->c1, &m, &t1);
}
INL void F2_sqr(F2 *r, const F2 *a) {                         /* (a0+a1)(a0-a1) + 2 a0 a1 u */
    F1 s, d, m;
    F1_add(&s, &a->c0, &a->c1); F1_sub(&d, &a->c0, &a->c1); F1_mul(&m, &a->c0, &a->c1);
    F1_mul(&r->c0, &s, &d); F1_add(&r->c1, &m, &m);
}

/* ------------------------------------------------------------------ XYZZ group law, generic over the field */
#define DEFINE_XYZZ(G, F)                                                                                     \
    typedef struct { F x, y; int inf; } G##aff_t;                                                             \
    typedef struct { F x, y, zz, zzz; } G##xyzz_t;              /* zz = 0: identity */                        \
    INL void G##x_set_inf(G##xyzz_t *r) { memset(r, 0, sizeof *r); }                                          \
    INL int G##x_is_inf(const G##xyzz_t *a) { return F##_is0(&a->zz); }                                       \
    static void G##x_dbl_aff(G##xyzz_t *r, const F *x, const F *y) {                                          \
        F U, V, W, S, M, t;                                                                                   \
        F##_add(&U, y, y); F##_sqr(&V, &U); F##_mul(&W, &U, &V); F##_mul(&S, x, &V);                          \
        F##_sqr(&M, x); F##_add(&t, &M, &M); F##_add(&M, &t, &M);                                             \
        F##_sqr(&r->x, &M); F##_sub(&r->x, &r->x, &S); F##_sub(&r->x, &r->x, &S);                             \
        F##_sub(&t, &S, &r->x); F##_mul(&t, &M, &t); F##_mul(&U, &W, y); F##_sub(&r->y, &t, &U);              \
        r->zz = V; r->zzz = W;                                                                                \
    }                                                                                                         \
    static void G##x_dbl(G##xyzz_t *r, const G##xyzz_t *p) {                                                  \
        if (G##x_is_inf(p)) { *r = *p; return; }                                                              \
        F U, V, W, S, M, t, X3, Y3;                                                                           \
        F##_add(&U, &p->y, &p->y); F##_sqr(&V, &U); F##_mul(&W, &U, &V); F##_mul(&S, &p->x, &V);              \
        F##_sqr(&M, &p->x); F##_add(&t, &M, &M); F##_add(&M, &t, &M);                                         \
        F##_sqr(&X3, &M); F##_sub(&X3, &X3, &S); F##_sub(&X3, &X3, &S);                                       \
        F##_sub(&t, &S, &X3); F##_mul(&t, &M, &t); F##_mul(&U, &W, &p->y); F##_sub(&Y3, &t, &U);              \
        F##_mul(&r->zz, &V, &p->zz); F##_mul(&r->zzz, &W, &p->zzz); r->x = X3; r->y = Y3;                     \
    }                                                                                                         \
    /* r += (x2, +-y2) */                                                                                     \
    INL void G##x_madd(G##xyzz_t *r, const F *x2, const F *y2in, int negate) {                                \
        F y2 = *y2in;                                                                                         \
        if (negate) F##_neg(&y2, y2in);                                                                       \
        if (G##x_is_inf(r)) { r->x = *x2; r->y = y2; F##_one(&r->zz); F##_one(&r->zzz); return; }                   \
        F U2, S2, P, R, PP, PPP, Q, t;                                                                        \
        F##_mul(&U2, x2, &r->zz); F##_mul(&S2, &y2, &r->zzz);                                                 \
        F##_sub(&P, &U2, &r->x); F##_sub(&R, &S2, &r->y);                                                     \
        if (F##_is0(&P)) {                                                                                    \
            if (F##_is0(&R)) G##x_dbl_aff(r, x2, &y2); else G##x_set_inf(r);                                  \
            return;                                                                                           \
        }                                                                                                     \
        F##_sqr(&PP, &P); F##_mul(&PPP, &P, &PP); F##_mul(&Q, &r->x, &PP);                                    \
        F##_sqr(&t, &R); F##_sub(&t, &t, &PPP); F##_sub(&t, &t, &Q); F##_sub(&t, &t, &Q);                     \
        F##_mul(&U2, &r->y, &PPP);                                                                            \
        r->x = t;                                                                                             \
        F##_sub(&Q, &Q, &t); F##_mul(&Q, &R, &Q); F##_sub(&r->y, &Q, &U2);                                    \
        F##_mul(&r->zz, &r->zz, &PP); F##_mul(&r->zzz, &r->zzz, &PPP);                                        \
    }                                                                                                         \
    static void G##x_add(G##xyzz_t *r, const G##xyzz_t *a, const G##xyzz_t *b) {                              \
        if (G##x_is_inf(a)) { *r = *b; return; }                                                              \
        if (G##x_is_inf(b)) { *r = *a; return; }                                                              \
        F U1, U2, S1, S2, P, R, PP, PPP, Q, t, X3;                                                            \
        F##_mul(&U1, &a->x, &b->zz); F##_mul(&U2, &b->x, &a->zz);                                             \
        F##_mul(&S1, &a->y, &b->zzz); F##_mul(&S2, &b->y, &a->zzz);                                           \
        F##_sub(&P, &U2, &U1); F##_sub(&R, &S2, &S1);                                                         \
        if (F##_is0(&P)) {                                                                                    \
            if (F##_is0(&R)) G##x_dbl(r, a); else G##x_set_inf(r);                                            \
            return;                                                                                           \
        }                                                                                                     \
        F##_sqr(&PP, &P); F##_mul(&PPP, &P, &PP); F##_mul(&Q, &U1, &PP);                                      \
        F##_sqr(&X3, &R); F##_sub(&X3, &X3, &PPP); F##_sub(&X3, &X3, &Q); F##_sub(&X3, &X3, &Q);              \
        F##_sub(&t, &Q, &X3); F##_mul(&t, &R, &t); F##_mul(&S1, &S1, &PPP);                                   \
        F##_mul(&U1, &a->zz, &b->zz); F##_mul(&U2, &a->zzz, &b->zzz);                                         \
        F##_sub(&r->y, &t, &S1); r->x = X3;                                                                   \
        F##_mul(&r->zz, &U1, &PP); F##_mul(&r->zzz, &U2, &PPP);                                               \
    }

DEFINE_XYZZ(g1, F1)
DEFINE_XYZZ(g2, F2)

/* ------------------------------------------------------------------ threads */
typedef void (*par_fn)(int tid, int nth, void *arg);
typedef struct { par_fn fn; int tid, nth; void *arg; } par_item;
static void *par_tramp(void *p) { par_item *it = p; it->fn(it->tid, it->nth, it->arg); return NULL; }
static void run_par(int nth, par_fn fn, void *arg) {
    if (nth <= 1) { fn(0, 1, arg); return; }
    pthread_t th[256]; par_item it[256];
    if (nth > 256) nth = 256;
    for (int i = 0; i < nth; i++) { it[i] = (par_item){fn, i, nth, arg}; }
    for (int i = 1; i < nth; i++) pthread_create(&th[i], NULL, par_tramp, &it[i]);
    fn(0, nth, arg);
    for (int i = 1; i < nth; i++) pthread_join(th[i], NULL);
}
static void slice(size_t total, int tid, int nth, size_t *lo, size_t *hi) {
    *lo = total * (size_t)tid / (size_t)nth; *hi = total * (size_t)(tid + 1) / (size_t)nth;
}

/* ------------------------------------------------------------------ Pippenger, one slice of the points per thread */
#define DEFINE_MSM(G, F)                                                                                      \
    typedef struct { const G##aff_t *pts; const uint8_t *scal; const uint32_t *idx; size_t nnz; int c; G##xyzz_t *outs; } G##msm_job; \
    static void G##msm_worker(int tid, int nth, void *argp) {                                                 \
        G##msm_job *J = argp;                                                                                 \
        size_t lo, hi; slice(J->nnz, tid, nth, &lo, &hi);                                                     \
        const int c = J->c, nw = (256 + c - 1) / c;                                                           \
        const size_t cnt = hi - lo, nb = (size_t)1 << (c - 1);                                                \
        G##xyzz_t total; G##x_set_inf(&total);                                                                \
        if (cnt) {                                                                                            \
            int32_t *dig = malloc(sizeof(int32_t) * cnt * (size_t)nw);                                        \
            for (size_t i = 0; i < cnt; i++) {              /* signed digits in [-2^(c-1), 2^(c-1)] */        \
                const uint8_t *k = J->scal + 32 * (size_t)J->idx[lo + i];                                     \
                int carry = 0;                                                                                \
                for (int j = 0; j < nw; j++) {                                                                \
                    const int bit = j * c; uint32_t v = 0;                                                    \
                    for (int b = 0; b < c && bit + b < 256; b++) v |= (uint32_t)((k[(bit + b) >> 3] >> ((bit + b) & 7)) & 1) << b; \
                    int32_t d = (int32_t)v + carry;                                                           \
                    if (d > (int32_t)nb) { d -= (int32_t)(2 * nb); carry = 1; } else carry = 0;               \
                    dig[i * (size_t)nw + j] = d;                                                              \
                }                                                                                             \
            }                                                                                                 \
            G##xyzz_t *bk = malloc(sizeof(G##xyzz_t) * nb);                                                   \
            for (int j = nw - 1; j >= 0; j--) {                                                               \
                memset(bk, 0, sizeof(G##xyzz_t) * nb);                                                        \
                for (size_t i = 0; i < cnt; i++) {                                                            \
                    const int32_t d = dig[i * (size_t)nw + j];                                                \
                    if (!d) continue;                                                                         \
                    const G##aff_t *p = &J->pts[J->idx[lo + i]];                                              \
                    if (p->inf) continue;                                                                     \
                    G##x_madd(&bk[(d > 0 ? d : -d) - 1], &p->x, &p->y, d < 0);                                \
                }                                                                                             \
                G##xyzz_t acc, sum; G##x_set_inf(&acc); G##x_set_inf(&sum);                                   \
                for (size_t b = nb; b-- > 0;) { G##x_add(&acc, &acc, &bk[b]); G##x_add(&sum, &sum, &acc); }   \
                for (int k = 0; k < c; k++) G##x_dbl(&total, &total);                                         \
                G##x_add(&total, &total, &sum);                                                               \
            }                                                                                                 \
            free(bk); free(dig);                                                                              \
        }                                                                                                     \
        J->outs[tid] = total;                                                                                 \
    }                                                                                                         \
    /* scal: n canonical 32-byte scalars; points with zero scalars are skipped before the split */           \
    static void G##msm(G##xyzz_t *out, const G##aff_t *pts, const uint8_t *scal, size_t n, int nth) {         \
        uint32_t *idx = malloc(sizeof(uint32_t) * (n ? n : 1)); size_t nnz = 0;                               \
        for (size_t i = 0; i < n; i++) {                                                                      \
            const uint64_t *k = (const uint64_t *)(scal + 32 * i);                                            \
            if ((k[0] | k[1] | k[2] | k[3]) && !pts[i].inf) idx[nnz++] = (uint32_t)i;                         \
        }                                                                                                     \
        if ((size_t)nth > nnz) nth = nnz ? (int)nnz : 1;                                                      \
        const size_t per = nnz / (size_t)nth + 1;                                                             \
        int best = 2; double bc = 1e300;                                                                      \
        for (int c = 2; c <= 16; c++) {                                                                       \
            const double cost = (double)((256 + c - 1) / c) * ((double)per + 2.8 * (double)((size_t)1 << (c - 1)) + c); \
            if (cost < bc) { bc = cost; best = c; }                                                           \
        }                                                                                                     \
        G##xyzz_t *outs = malloc(sizeof(G##xyzz_t) * (size_t)nth);                                            \
        G##msm_job J = {pts, scal, idx, nnz, best, outs};                                                     \
        run_par(nth, G##msm_worker, &J);                                                                      \
        G##x_set_inf(out);                                                                                    \
        for (int t = 0; t < nth; t++) G##x_add(out, out, &outs[t]);                                           \
        free(outs); free(idx);                                                                                \
    }

DEFINE_MSM(g1, F1)
DEFINE_MSM(g2, F2)

static void g1_out(uint8_t out[96], const g1xyzz_t *p) {
    g1_t j; g1_set_inf(&j);
    if (!g1x_is_inf(p)) {             /* Jacobian (X zz, Y zzz^... ) : x = X/zz, y = Y/zzz  ->  (x, y, 1) */
        fp_t zi;
        fp_inv(&zi, &p->zz); F1_mul(&j.x, &p->x, &zi);
        fp_inv(&zi, &p->zzz); F1_mul(&j.y, &p->y, &zi);
        j.z = Q_ONE;
    }
    g1_to_bytes(out, &j);
}
static void g2_out(uint8_t out[192], const g2xyzz_t *p) {
    g2_t j; g2_set_inf(&j);
    if (!g2x_is_inf(p)) {
        fp2_t one; F2_one(&one);
        /* 1/(a + bu) = (a - bu)/(a^2 + b^2) */
        const fp2_t *zs[2] = {&p->zz, &p->zzz}; const fp2_t *ns[2] = {&p->x, &p->y}; fp2_t *os[2] = {&j.x, &j.y};
        for (int k = 0; k < 2; k++) {
            fp_t n0, n1, d; fp2_t inv;
            F1_sqr(&n0, &zs[k]->c0); F1_sqr(&n1, &zs[k]->c1); F1_add(&n0, &n0, &n1); fp_inv(&d, &n0);
            F1_mul(&inv.c0, &zs[k]->c0, &d); F1_mul(&n1, &zs[k]->c1, &d); F1_neg(&inv.c1, &n1);
            F2_mul(os[k], ns[k], &inv);
        }
        j.z = one;
    }
    g2_to_bytes(out, &j);
}

/* ------------------------------------------------------------------ NTT over Fr, threads share every stage */
typedef struct {
    fr_t *a; const fr_t *tw; uint32_t logn; pthread_barrier_t *bar;
} ntt_job;
static void ntt_worker(int tid, int nth, void *argp) {
    ntt_job *J = argp;
    const size_t N = (size_t)1 << J->logn;
    size_t lo, hi;
    slice(N, tid, nth, &lo, &hi);
    for (size_t i = lo; i < hi; i++) {                         /* bit reversal: the owner of the smaller index swaps */
        size_t r = 0;
        for (uint32_t b = 0; b < J->logn; b++) r |= ((i >> b) & 1) << (J->logn - 1 - b);
        if (i < r) { fr_t t = J->a[i]; J->a[i] = J->a[r]; J->a[r] = t; }
    }
    if (nth > 1) pthread_barrier_wait(J->bar);
    slice(N / 2, tid, nth, &lo, &hi);
    for (uint32_t s = 0; s < J->logn; s++) {
        const size_t half = (size_t)1 << s, stride = (N / 2) >> s;
        for (size_t b = lo; b < hi; b++) {
            const size_t pos = b & (half - 1), base = ((b >> s) << (s + 1)) + pos;
            fr_t t; xr_mul(&t, &J->a[base + half], &J->tw[pos * stride]);
            xr_sub(&J->a[base + half], &J->a[base], &t);
            xr_add(&J->a[base], &J->a[base], &t);
        }
        if (nth > 1) pthread_barrier_wait(J->bar);
    }
}
static void ntt(fr_t *a, const fr_t *tw, uint32_t logn, int nth) {
    if (((size_t)1 << logn) < 4096) nth = 1;
    pthread_barrier_t bar;
    if (nth > 1) pthread_barrier_init(&bar, NULL, (unsigned)nth);
    ntt_job J = {a, tw, logn, &bar};
    run_par(nth, ntt_worker, &J);
    if (nth > 1) pthread_barrier_destroy(&bar);
}

/* ------------------------------------------------------------------ the prover */
typedef struct { const uint32_t *ptr, *col; const uint8_t *val; } csr_t;
typedef struct {
    uint32_t n, m, n_mid, logN;
    size_t p1, p2, N;
    int nth;
    uint32_t *l_ptr, *l_col, *r_ptr, *r_col, *o_ptr, *o_col, *mid_idx;
    fr_t *l_val, *r_val, *o_val;                 /* Montgomery */
    g1aff_t *g1; g2aff_t *g2;
    fr_t *tw_f, *tw_i;                           /* w^k, w^-k, k < N/2 */
    fr_t *alt_hat, *e_hat;                       /* transforms of (-1)^j/j! (j < n) and 1/j! (j < N), both times 1/N */
    fr_t *fact, *ifact;                          /* k!, 1/k!, k < 2n */
} fast_ctx;

static void csr_copy(uint32_t n, const uint32_t *ptr, const uint32_t *col, const uint8_t *val, uint32_t **p, uint32_t **c, fr_t **v) {
    const uint32_t nnz = ptr[n];
    *p = malloc(4 * ((size_t)n + 1)); memcpy(*p, ptr, 4 * ((size_t)n + 1));
    *c = malloc(4 * (size_t)(nnz ? nnz : 1)); memcpy(*c, col, 4 * (size_t)nnz);
    *v = malloc(sizeof(fr_t) * (size_t)(nnz ? nnz : 1));
    for (uint32_t e = 0; e < nnz; e++) fr_from_bytes(&(*v)[e], val + 32 * (size_t)e);
}

API void *orc_fast_groth16_new(uint32_t n, uint32_t m,
                               const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                               const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                               const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val,
                               const uint8_t *mid, const uint8_t *lag_g1, size_t p1, const uint8_t *lag_g2, size_t p2, int threads) {
    if (n < 2) return NULL;
    init_one();
    fast_ctx *C = calloc(1, sizeof *C);
    C->n = n; C->m = m; C->nth = threads < 1 ? 1 : threads;
    for (uint32_t k = 0; k < m; k++) C->n_mid += mid[k] ? 1 : 0;
    C->p1 = 3 + (size_t)n + (n - 1) + C->n_mid; C->p2 = 2 + (size_t)n;
    if (p1 != C->p1 || p2 != C->p2) { free(C); return NULL; }
    C->mid_idx = malloc(4 * (size_t)(C->n_mid ? C->n_mid : 1));
    for (uint32_t k = 0, j = 0; k < m; k++) if (mid[k]) C->mid_idx[j++] = k;
    csr_copy(n, l_ptr, l_col, l_val, &C->l_ptr, &C->l_col, &C->l_val);
    csr_copy(n, r_ptr, r_col, r_val, &C->r_ptr, &C->r_col, &C->r_val);
    csr_copy(n, o_ptr, o_col, o_val, &C->o_ptr, &C->o_col, &C->o_val);
    C->g1 = malloc(sizeof(g1aff_t) * p1); C->g2 = malloc(sizeof(g2aff_t) * p2);
    for (size_t i = 0; i < p1; i++) {
        g1_t P;
        if (g1_from_bytes(&P, lag_g1 + 96 * i)) { free(C); return NULL; }
        C->g1[i].x = P.x; C->g1[i].y = P.y; C->g1[i].inf = g1_is_inf(&P);
    }
    for (size_t i = 0; i < p2; i++) {
        g2_t P;
        if (g2_from_bytes(&P, lag_g2 + 192 * i)) { free(C); return NULL; }
        C->g2[i].x = P.x; C->g2[i].y = P.y; C->g2[i].inf = g2_is_inf(&P);
    }
    /* transform size: N >= 2n - 1 (the first convolution's outputs 0..n-1 see no wrap-around; the second's outputs n..2n-2
       are met only by wrapped indices t + N <= 3n - 3, i.e. t <= n - 3: outside the range that is read) */
    C->logN = 1;
    while (((size_t)1 << C->logN) < 2 * (size_t)n - 1) C->logN++;
    C->N = (size_t)1 << C->logN;
    const size_t N = C->N;
    fr_t w, wi, x;
    fr_omega(&w);
    for (uint32_t k = C->logN; k < 32; k++) xr_mul(&w, &w, &w);            /* w_N = omega^(2^32 / N), FFT.ml:208-232 */
    fr_inv(&wi, &w);
    C->tw_f = malloc(sizeof(fr_t) * (N / 2)); C->tw_i = malloc(sizeof(fr_t) * (N / 2));
    C->tw_f[0] = FR_ONE; C->tw_i[0] = FR_ONE;
    for (size_t k = 1; k < N / 2; k++) { xr_mul(&C->tw_f[k], &C->tw_f[k - 1], &w); xr_mul(&C->tw_i[k], &C->tw_i[k - 1], &wi); }
    const size_t F = 2 * (size_t)n > N ? 2 * (size_t)n : N;
    C->fact = malloc(sizeof(fr_t) * F); C->ifact = malloc(sizeof(fr_t) * F);
    C->fact[0] = FR_ONE;
    for (size_t k = 1; k < F; k++) { fr_from_u64(&x, k); xr_mul(&C->fact[k], &C->fact[k - 1], &x); }
    fr_inv(&C->ifact[F - 1], &C->fact[F - 1]);
    for (size_t k = F - 1; k > 0; k--) { fr_from_u64(&x, k); xr_mul(&C->ifact[k - 1], &C->ifact[k], &x); }
    fr_t ninv; fr_from_u64(&x, N); fr_inv(&ninv, &x);
    C->alt_hat = calloc(N, sizeof(fr_t)); C->e_hat = calloc(N, sizeof(fr_t));
    for (size_t j = 0; j < n; j++) { if (j & 1) fr_neg(&C->alt_hat[j], &C->ifact[j]); else C->alt_hat[j] = C->ifact[j]; }
    for (size_t j = 0; j < N && j < 2 * (size_t)n - 1; j++) C->e_hat[j] = C->ifact[j];
    ntt(C->alt_hat, C->tw_f, C->logN, C->nth); ntt(C->e_hat, C->tw_f, C->logN, C->nth);
    for (size_t j = 0; j < N; j++) { xr_mul(&C->alt_hat[j], &C->alt_hat[j], &ninv); xr_mul(&C->e_hat[j], &C->e_hat[j], &ninv); }
    return C;
}
API void orc_fast_groth16_free(void *h) {
    fast_ctx *C = h;
    if (!C) return;
    free(C->l_ptr); free(C->l_col); free(C->l_val); free(C->r_ptr); free(C->r_col); free(C->r_val);
    free(C->o_ptr); free(C->o_col); free(C->o_val); free(C->mid_idx); free(C->g1); free(C->g2);
    free(C->tw_f); free(C->tw_i); free(C->alt_hat); free(C->e_hat); free(C->fact); free(C->ifact);
    free(C);
}

typedef struct { const fast_ctx *C; const fr_t *w; fr_t *a, *b, *c; } spmv_job;
static void spmv_worker(int tid, int nth, void *argp) {
    spmv_job *J = argp; const fast_ctx *C = J->C;
    size_t lo, hi; slice(C->n, tid, nth, &lo, &hi);
    const uint32_t *ptrs[3] = {C->l_ptr, C->r_ptr, C->o_ptr}, *cols[3] = {C->l_col, C->r_col, C->o_col};
    const fr_t *vals[3] = {C->l_val, C->r_val, C->o_val}; fr_t *outs[3] = {J->a, J->b, J->c};
    for (int q = 0; q < 3; q++)
        for (size_t g = lo; g < hi; g++) {
            fr_t acc = FR_ZERO, t;
            for (uint32_t e = ptrs[q][g]; e < ptrs[q][g + 1]; e++) { xr_mul(&t, &vals[q][e], &J->w[cols[q][e]]); xr_add(&acc, &acc, &t); }
            outs[q][g] = acc;
        }
}
typedef struct { fr_t *x; const fr_t *k; size_t lo, hi; } pw_job;          /* x[i] *= k[i], i in [lo, hi); x[i] = 0 beyond hi up to N handled by the caller */
static void pw_worker(int tid, int nth, void *argp) {
    pw_job *J = argp; size_t lo, hi; slice(J->hi - J->lo, tid, nth, &lo, &hi);
    for (size_t i = J->lo + lo; i < J->lo + hi; i++) xr_mul(&J->x[i], &J->x[i], &J->k[i]);
}
static void pointwise(fr_t *x, const fr_t *k, size_t lo, size_t hi, int nth) { pw_job J = {x, k, lo, hi}; run_par(nth, pw_worker, &J); }

/* values at 0..n-1 (in x[0..n-1]) -> x[t] = p(t) / t! for t in n..2n-2 (other entries: scratch) */
static void extrapolate(const fast_ctx *C, fr_t *x) {
    const size_t n = C->n, N = C->N;
    pointwise(x, C->ifact, 0, n, C->nth);
    memset(x + n, 0, sizeof(fr_t) * (N - n));
    ntt(x, C->tw_f, C->logN, C->nth); pointwise(x, C->alt_hat, 0, N, C->nth); ntt(x, C->tw_i, C->logN, C->nth);
    memset(x + n, 0, sizeof(fr_t) * (N - n));                                   /* d_k = Delta^k p(0) / k!, k < n */
    ntt(x, C->tw_f, C->logN, C->nth); pointwise(x, C->e_hat, 0, N, C->nth); ntt(x, C->tw_i, C->logN, C->nth);
}

/* 0 ok; 1 = the witness does not satisfy the circuit (QAP.ml:134 "Polynomial.is_zero rem"); 2 = bad argument */
API int orc_fast_groth16_prove(void *h, const uint8_t *sol, const uint8_t r_[32], const uint8_t s_[32], uint8_t proof[384]) {
    fast_ctx *C = h;
    if (!C) return 2;
    const size_t n = C->n, N = C->N;
    fr_t *w = malloc(sizeof(fr_t) * C->m);
    for (uint32_t k = 0; k < C->m; k++) fr_from_bytes(&w[k], sol + 32 * (size_t)k);
    fr_t *a = calloc(N, sizeof(fr_t)), *b = calloc(N, sizeof(fr_t)), *c = calloc(N, sizeof(fr_t));
    spmv_job SJ = {C, w, a, b, c};
    run_par(C->nth, spmv_worker, &SJ);
    int bad = 0;
    for (size_t i = 0; i < n; i++) { fr_t t; xr_mul(&t, &a[i], &b[i]); if (!fr_eq(&t, &c[i])) bad = 1; }
    fr_t r, s, rs, t, u;
    fr_from_bytes(&r, r_); fr_from_bytes(&s, s_); xr_mul(&rs, &r, &s);
    uint8_t *sA = calloc(C->p1, 32), *sC = calloc(C->p1, 32), *sB = calloc(C->p2, 32);
    fr_to_bytes(sA, &FR_ONE); fr_to_bytes(sA + 32, &r);
    fr_to_bytes(sC, &s); fr_to_bytes(sC + 32, &rs); fr_to_bytes(sC + 64, &r);
    fr_to_bytes(sB, &FR_ONE); fr_to_bytes(sB + 32, &s);
    for (size_t k = 0; k < n; k++) {
        fr_to_bytes(sA + 32 * (3 + k), &a[k]);
        fr_to_bytes(sB + 32 * (2 + k), &b[k]);
        xr_mul(&t, &s, &a[k]); xr_mul(&u, &r, &b[k]); xr_add(&t, &t, &u);
        fr_to_bytes(sC + 32 * (3 + k), &t);
    }
    if (!bad) {
        extrapolate(C, a); extrapolate(C, b); extrapolate(C, c);
        for (size_t tt = n; tt <= 2 * n - 2; tt++) {                    /* h(t) = (t! A_t B_t - C_t) (t-n)! */
            xr_mul(&t, &a[tt], &b[tt]); xr_mul(&t, &t, &C->fact[tt]); xr_sub(&t, &t, &c[tt]); xr_mul(&t, &t, &C->fact[tt - n]);
            fr_to_bytes(sC + 32 * (3 + n + (tt - n)), &t);
        }
        for (uint32_t j = 0; j < C->n_mid; j++) fr_to_bytes(sC + 32 * (3 + n + (n - 1) + j), &w[C->mid_idx[j]]);
        g1xyzz_t A, Cc; g2xyzz_t B;
        g1msm(&A, C->g1, sA, C->p1, C->nth);
        g2msm(&B, C->g2, sB, C->p2, C->nth);
        g1msm(&Cc, C->g1, sC, C->p1, C->nth);
        g1_out(proof, &A); g2_out(proof + 96, &B); g1_out(proof + 288, &Cc);
    }
    free(w); free(a); free(b); free(c); free(sA); free(sC); free(sB);
    return bad;
}

/* a bare product for tests: out = sum_i scalars[i] * bases[i] (96-byte affine points), threads as given */
API int orc_fast_g1_msm(uint8_t out[96], const uint8_t *bases, const uint8_t *scalars, size_t n, int threads) {
    init_one();
    g1aff_t *p = malloc(sizeof(g1aff_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        g1_t P;
        if (g1_from_bytes(&P, bases + 96 * i)) { free(p); return -1; }
        p[i].x = P.x; p[i].y = P.y; p[i].inf = g1_is_inf(&P);
    }
    g1xyzz_t R; g1msm(&R, p, scalars, n, threads); g1_out(out, &R); free(p); return 0;
}
API int orc_fast_g2_msm(uint8_t out[192], const uint8_t *bases, const uint8_t *scalars, size_t n, int threads) {
    init_one();
    g2aff_t *p = malloc(sizeof(g2aff_t) * (n ? n : 1));
    for (size_t i = 0; i < n; i++) {
        g2_t P;
        if (g2_from_bytes(&P, bases + 192 * i)) { free(p); return -1; }
        p[i].x = P.x; p[i].y = P.y; p[i].inf = g2_is_inf(&P);
    }
    g2xyzz_t R; g2msm(&R, p, scalars, n, threads); g2_out(out, &R); free(p); return 0;
}
