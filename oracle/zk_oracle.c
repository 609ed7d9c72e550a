/* TEST INFRASTRUCTURE ONLY -- CPU restatement of zukelang's prove path.
 *
 * Each function cites the reference file:line it follows (paths relative to
 * /root/reference).  PARITY UNPINNED by the reference (no golden vectors,
 * SURVEY.md 8c); pinned by oracle/pyref.py and the hand-derived README-circuit
 * fixture tests/golden/readme_circuit.json.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load
 * the shared object built from this file.  The product (zukelang_amd/) never
 * links, imports or executes it.
 *
 * Boundary conventions (same as include/zkmi355x.h): Fr = 32 B little-endian
 * canonical; G1 = 96 B, G2 = 192 B uncompressed ZCash big-endian.  Variables
 * are dense indices 0..m-1 (the OCaml side's Var.t Map keys in Var.compare
 * order, src/lib/zk/var.ml:42).  R1CS gates `lhs = l * r`
 * (src/lib/zk/circuit.ml:73-75) are three CSR matrices over gate ids 0..n-1
 * in Gate.Set order (src/lib/zk/QAP.ml:22).
 */
#include "bls12_381.h"
#include <stdlib.h>
#include <string.h>

#define API __attribute__((visibility("default")))

/* ================================================================== field / group one-liners (for cross-checks) */
API void orc_fr_mul(uint8_t out[32], const uint8_t a[32], const uint8_t b[32]) {
    fr_t x, y; fr_from_bytes(&x, a); fr_from_bytes(&y, b); fr_mul(&x, &x, &y); fr_to_bytes(out, &x);
}
API void orc_fr_inv(uint8_t out[32], const uint8_t a[32]) {
    fr_t x; fr_from_bytes(&x, a); fr_inv(&x, &x); fr_to_bytes(out, &x);
}
API void orc_fr_omega(uint8_t out[32]) { fr_t w; fr_omega(&w); fr_to_bytes(out, &w); }
API int orc_g1_add(uint8_t out[96], const uint8_t a[96], const uint8_t b[96]) {
    g1_t p, q; if (g1_from_bytes(&p, a) || g1_from_bytes(&q, b)) return -1;
    g1_add(&p, &p, &q); g1_to_bytes(out, &p); return 0;
}
API int orc_g1_mul(uint8_t out[96], const uint8_t a[96], const uint8_t k[32]) {
    g1_t p; fr_t s; if (g1_from_bytes(&p, a)) return -1;
    fr_from_bytes(&s, k); g1_mul(&p, &p, &s); g1_to_bytes(out, &p); return 0;
}
API int orc_g2_add(uint8_t out[192], const uint8_t a[192], const uint8_t b[192]) {
    g2_t p, q; if (g2_from_bytes(&p, a) || g2_from_bytes(&q, b)) return -1;
    g2_add(&p, &p, &q); g2_to_bytes(out, &p); return 0;
}
API int orc_g2_mul(uint8_t out[192], const uint8_t a[192], const uint8_t k[32]) {
    g2_t p; fr_t s; if (g2_from_bytes(&p, a)) return -1;
    fr_from_bytes(&s, k); g2_mul(&p, &p, &s); g2_to_bytes(out, &p); return 0;
}
API void orc_g1_generator(uint8_t out[96]) { g1_t g; g1_generator(&g); g1_to_bytes(out, &g); }
API void orc_g2_generator(uint8_t out[192]) { g2_t g; g2_generator(&g); g2_to_bytes(out, &g); }
API int orc_g1_compress(uint8_t out[48], const uint8_t a[96]) {
    g1_t p; if (g1_from_bytes(&p, a)) return -1; g1_compress(out, &p); return 0;
}
API int orc_g2_compress(uint8_t out[96], const uint8_t a[192]) {
    g2_t p; if (g2_from_bytes(&p, a)) return -1; g2_compress(out, &p); return 0;
}

/* ================================================================== G.powers / apply_powers / dot
 * src/lib/zk/curve.ml:106-109  powers d s = [ of_Fr (s ** i) | i <- 0..d ]      (d+1 items)
 * src/lib/zk/curve.ml:112-118  apply_powers cs xis = fold (x * c + acc), invalid_arg if points run out
 * src/lib/zk/curve.ml:94-103   dot m c = sum_map m (fun k mk -> mk * c_k)
 * All are left folds of one scalar multiplication and one addition per term. */
API void orc_g1_powers(uint8_t *out /* (d+1)*96 */, uint32_t d, const uint8_t s[32]) {
    g1_t g, p; fr_t sc, si = FR_ONE;
    g1_generator(&g); fr_from_bytes(&sc, s);
    for (uint32_t i = 0; i <= d; i++) {
        g1_mul(&p, &g, &si); g1_to_bytes(out + 96 * (size_t)i, &p);
        fr_mul(&si, &si, &sc);
    }
}
API void orc_g2_powers(uint8_t *out /* (d+1)*192 */, uint32_t d, const uint8_t s[32]) {
    g2_t g, p; fr_t sc, si = FR_ONE;
    g2_generator(&g); fr_from_bytes(&sc, s);
    for (uint32_t i = 0; i <= d; i++) {
        g2_mul(&p, &g, &si); g2_to_bytes(out + 192 * (size_t)i, &p);
        fr_mul(&si, &si, &sc);
    }
}
/* returns -2 for the reference's invalid_arg "apply_powers" (fewer points than coefficients) */
API int orc_g1_msm_naive(uint8_t out[96], const uint8_t *bases, size_t nbases, const uint8_t *scalars, size_t nscalars) {
    if (nscalars > nbases) return -2;
    g1_t acc, p; fr_t c;
    g1_set_inf(&acc);
    for (size_t i = 0; i < nscalars; i++) {
        if (g1_from_bytes(&p, bases + 96 * i)) return -1;
        fr_from_bytes(&c, scalars + 32 * i);
        g1_mul(&p, &p, &c); g1_add(&acc, &p, &acc);
    }
    g1_to_bytes(out, &acc); return 0;
}
API int orc_g2_msm_naive(uint8_t out[192], const uint8_t *bases, size_t nbases, const uint8_t *scalars, size_t nscalars) {
    if (nscalars > nbases) return -2;
    g2_t acc, p; fr_t c;
    g2_set_inf(&acc);
    for (size_t i = 0; i < nscalars; i++) {
        if (g2_from_bytes(&p, bases + 192 * i)) return -1;
        fr_from_bytes(&c, scalars + 32 * i);
        g2_mul(&p, &p, &c); g2_add(&acc, &p, &acc);
    }
    g2_to_bytes(out, &acc); return 0;
}

/* ================================================================== FFT.ml:29-67  gen_fft (recursive radix-2 DIT)
 * zeta n i = w^((2^32/n) * i), w = 5^((r-1)/2^32)  (FFT.ml:208-232).
 * out[k]       = a0'[k] + zeta_n'(k')      * a1'[k],   k <  n'/2
 * out[k+n'/2]  = a0'[k] + zeta_n'(k'+n'/2) * a1'[k]    with k' = -k when inverse (FFT.ml:55-62)
 * inverse divides every entry by n (FFT.ml:64-66). */
static void fft_rec(fr_t *a, size_t n, const fr_t *zeta /* zeta[i] = w_N^i, i<N */, size_t N, size_t m, int inv) {
    if (n <= 1) return;
    size_t h = n / 2;
    fr_t *a0 = malloc(sizeof(fr_t) * h), *a1 = malloc(sizeof(fr_t) * h);
    for (size_t i = 0; i < h; i++) { a0[i] = a[2 * i]; a1[i] = a[2 * i + 1]; }
    fft_rec(a0, h, zeta, N, m * 2, inv);
    fft_rec(a1, h, zeta, N, m * 2, inv);
    for (size_t k = 0; k < h; k++) {
        /* zeta_n' i = zeta_N (i*m); negative indices wrap mod N */
        size_t e0 = inv ? (N - (k * m) % N) % N : (k * m) % N;
        size_t e1 = inv ? (N + h * m - (k * m) % N) % N : ((k + h) * m) % N;
        fr_t t;
        fr_mul(&t, &zeta[e0], &a1[k]); fr_add(&a[k], &a0[k], &t);
        fr_mul(&t, &zeta[e1], &a1[k]); fr_add(&a[k + h], &a0[k], &t);
    }
    free(a0); free(a1);
}
API int orc_fr_ntt(uint8_t *io, uint32_t log_n, int inverse) {
    if (log_n > 32) return -1;   /* FFT.ml:230 invalid_arg "Fr.zeta" */
    size_t n = (size_t)1 << log_n;
    fr_t *a = malloc(sizeof(fr_t) * n), *zeta = malloc(sizeof(fr_t) * n);
    fr_t w, wn;
    fr_omega(&w);
    wn = w;
    for (uint32_t i = log_n; i < 32; i++) fr_mul(&wn, &wn, &wn);  /* w^(2^32/n) */
    zeta[0] = FR_ONE;
    for (size_t i = 1; i < n; i++) fr_mul(&zeta[i], &zeta[i - 1], &wn);
    for (size_t i = 0; i < n; i++) fr_from_bytes(&a[i], io + 32 * i);
    fft_rec(a, n, zeta, n, 1, inverse);
    if (inverse) {
        fr_t ninv; fr_from_u64(&ninv, (uint64_t)n); fr_inv(&ninv, &ninv);
        for (size_t i = 0; i < n; i++) fr_mul(&a[i], &a[i], &ninv);
    }
    for (size_t i = 0; i < n; i++) fr_to_bytes(io + 32 * i, &a[i]);
    free(a); free(zeta);
    return 0;
}

/* ================================================================== polynomial.ml (dense, low -> high, normalized) */
static size_t poly_normalize(const fr_t *p, size_t n) {   /* polynomial.ml:100-107 */
    while (n && fr_is_zero(&p[n - 1])) n--;
    return n;
}
/* polynomial.ml:124-131  mul = sum_i (x^i * a_i * p2): schoolbook */
static size_t poly_mul(fr_t *out, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    if (!na || !nb) return 0;
    for (size_t i = 0; i < na + nb - 1; i++) out[i] = FR_ZERO;
    for (size_t i = 0; i < na; i++) {
        if (fr_is_zero(&a[i])) continue;       /* mul_scalar zero -> [] (polynomial.ml:119-120) */
        for (size_t j = 0; j < nb; j++) {
            fr_t t; fr_mul(&t, &a[i], &b[j]); fr_add(&out[i + j], &out[i + j], &t);
        }
    }
    return poly_normalize(out, na + nb - 1);
}
/* polynomial.ml:142-169  div_rem: long division from the top coefficient, d = a1 / rp2hd per step */
static void poly_divrem(fr_t *q, size_t *nq, fr_t *rem, size_t *nrem, const fr_t *a, size_t na, const fr_t *b, size_t nb) {
    na = poly_normalize(a, na); nb = poly_normalize(b, nb);
    memcpy(rem, a, sizeof(fr_t) * na);
    if (na < nb) { *nq = 0; *nrem = na; return; }
    fr_t binv; fr_inv(&binv, &b[nb - 1]);
    size_t qlen = na - nb + 1;
    for (size_t k = qlen; k-- > 0;) {
        fr_t d; fr_mul(&d, &rem[k + nb - 1], &binv);
        q[k] = d;
        for (size_t j = 0; j < nb; j++) {
            fr_t t; fr_mul(&t, &d, &b[j]); fr_sub(&rem[k + j], &rem[k + j], &t);
        }
    }
    *nq = qlen;              /* List.rev ds keeps every quotient coefficient (no normalize) */
    *nrem = poly_normalize(rem, nb - 1);
}
API size_t orc_poly_mul(uint8_t *out, const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    fr_t *x = malloc(sizeof(fr_t) * (na + 1)), *y = malloc(sizeof(fr_t) * (nb + 1)), *o = malloc(sizeof(fr_t) * (na + nb + 1));
    for (size_t i = 0; i < na; i++) fr_from_bytes(&x[i], a + 32 * i);
    for (size_t i = 0; i < nb; i++) fr_from_bytes(&y[i], b + 32 * i);
    size_t n = poly_mul(o, x, na, y, nb);
    for (size_t i = 0; i < n; i++) fr_to_bytes(out + 32 * i, &o[i]);
    free(x); free(y); free(o);
    return n;
}
API void orc_poly_divrem(uint8_t *q, size_t *nq, uint8_t *rem, size_t *nrem, const uint8_t *a, size_t na, const uint8_t *b, size_t nb) {
    fr_t *x = malloc(sizeof(fr_t) * (na + 1)), *y = malloc(sizeof(fr_t) * (nb + 1));
    fr_t *qq = malloc(sizeof(fr_t) * (na + 1)), *rr = malloc(sizeof(fr_t) * (na + 1));
    for (size_t i = 0; i < na; i++) fr_from_bytes(&x[i], a + 32 * i);
    for (size_t i = 0; i < nb; i++) fr_from_bytes(&y[i], b + 32 * i);
    poly_divrem(qq, nq, rr, nrem, x, na, y, nb);
    for (size_t i = 0; i < *nq; i++) fr_to_bytes(q + 32 * i, &qq[i]);
    for (size_t i = 0; i < *nrem; i++) fr_to_bytes(rem + 32 * i, &rr[i]);
    free(x); free(y); free(qq); free(rr);
}

/* ================================================================== QAP.ml:18-94  build
 * Points are F.of_int rg for rg = 0..n-1 (QAP.ml:84); target = prod (x - rg) (QAP.ml:92,
 * polynomial.ml:248-251).  lagrange_basis (polynomial.ml:212-226) builds
 * l_j = prod_{i != j} (x - x_i)/(x_j - x_i); here l_j = Z/(x - j) / Z'(j), the same polynomial
 * (exact field arithmetic), computed once and shared by every variable instead of once per
 * `interpolate` call -- the reference recomputes it 3m times with identical results. */
typedef struct {
    uint32_t n, m;
    fr_t *v, *w, *y;   /* m * n dense coefficients each, row k = variable k */
    fr_t *target;      /* n + 1 */
} qap_t;

static void z_poly(fr_t *z, uint32_t n) {      /* polynomial.ml:248-251 */
    z[0] = FR_ONE;
    for (uint32_t i = 0; i < n; i++) {
        fr_t fi; fr_from_u64(&fi, i);
        z[i + 1] = z[i];
        for (uint32_t k = i; k >= 1; k--) {
            fr_t t; fr_mul(&t, &z[k], &fi); fr_sub(&z[k], &z[k - 1], &t);
        }
        fr_t t; fr_mul(&t, &z[0], &fi); fr_neg(&z[0], &t);
    }
}
static fr_t *lagrange_basis_int(uint32_t n, const fr_t *z) {
    fr_t *L = malloc(sizeof(fr_t) * (size_t)n * n);
    for (uint32_t j = 0; j < n; j++) {
        fr_t fj; fr_from_u64(&fj, j);
        fr_t *l = L + (size_t)j * n;
        /* synthetic division of monic Z by (x - j) */
        l[n - 1] = z[n];
        for (uint32_t k = n - 1; k >= 1; k--) {
            fr_t t; fr_mul(&t, &l[k], &fj); fr_add(&l[k - 1], &z[k], &t);
        }
        fr_t den = FR_ONE;
        for (uint32_t i = 0; i < n; i++) {
            if (i == j) continue;
            fr_t fi, d; fr_from_u64(&fi, i); fr_sub(&d, &fj, &fi); fr_mul(&den, &den, &d);
        }
        fr_inv(&den, &den);
        for (uint32_t k = 0; k < n; k++) fr_mul(&l[k], &l[k], &den);
    }
    return L;
}
static void csr_to_polys(fr_t *out, uint32_t n, uint32_t m, const fr_t *L,
                         const uint32_t *rowptr, const uint32_t *col, const uint8_t *val) {
    memset(out, 0, sizeof(fr_t) * (size_t)m * n);
    for (uint32_t g = 0; g < n; g++)
        for (uint32_t e = rowptr[g]; e < rowptr[g + 1]; e++) {
            fr_t c; fr_from_bytes(&c, val + 32 * (size_t)e);
            fr_t *p = out + (size_t)col[e] * n;
            const fr_t *l = L + (size_t)g * n;
            for (uint32_t k = 0; k < n; k++) { fr_t t; fr_mul(&t, &c, &l[k]); fr_add(&p[k], &p[k], &t); }
        }
}
API void *orc_qap_build(uint32_t n, uint32_t m,
                        const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                        const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                        const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val) {
    qap_t *q = malloc(sizeof *q);
    q->n = n; q->m = m;
    q->target = malloc(sizeof(fr_t) * (n + 1));
    z_poly(q->target, n);
    fr_t *L = lagrange_basis_int(n, q->target);
    q->v = malloc(sizeof(fr_t) * (size_t)m * n);
    q->w = malloc(sizeof(fr_t) * (size_t)m * n);
    q->y = malloc(sizeof(fr_t) * (size_t)m * n);
    csr_to_polys(q->v, n, m, L, l_ptr, l_col, l_val);   /* QAP.ml:30-36  v: left operand  */
    csr_to_polys(q->w, n, m, L, r_ptr, r_col, r_val);   /* QAP.ml:38-44  w: right operand */
    csr_to_polys(q->y, n, m, L, o_ptr, o_col, o_val);   /* QAP.ml:46-52  y: lhs           */
    free(L);
    return q;
}
API void orc_qap_free(void *h) {
    qap_t *q = h; free(q->v); free(q->w); free(q->y); free(q->target); free(q);
}
API void orc_qap_get(void *h, int which, uint32_t k, uint8_t *out /* n*32, or (n+1)*32 for target */) {
    qap_t *q = h;
    if (which == 3) { for (uint32_t i = 0; i <= q->n; i++) fr_to_bytes(out + 32 * i, &q->target[i]); return; }
    fr_t *src = (which == 0 ? q->v : which == 1 ? q->w : q->y) + (size_t)k * q->n;
    for (uint32_t i = 0; i < q->n; i++) fr_to_bytes(out + 32 * i, &src[i]);
}

/* QAP.ml:120-135 eval: v = sum_k sol_k * v_k (mul_scalar + sum), p = v*w - y, (h, rem) = p /% target,
 * assert rem = 0.  Returns 0, or -3 when the remainder is not zero (the reference's assert). */
static int qap_eval(const qap_t *q, const fr_t *sol, fr_t *v, fr_t *w, fr_t *y, fr_t *p, size_t *np, fr_t *h, size_t *nh) {
    uint32_t n = q->n, m = q->m;
    for (uint32_t i = 0; i < n; i++) v[i] = w[i] = y[i] = FR_ZERO;
    for (uint32_t k = 0; k < m; k++)
        for (uint32_t i = 0; i < n; i++) {
            fr_t t;
            fr_mul(&t, &sol[k], &q->v[(size_t)k * n + i]); fr_add(&v[i], &v[i], &t);
            fr_mul(&t, &sol[k], &q->w[(size_t)k * n + i]); fr_add(&w[i], &w[i], &t);
            fr_mul(&t, &sol[k], &q->y[(size_t)k * n + i]); fr_add(&y[i], &y[i], &t);
        }
    size_t nv = poly_normalize(v, n), nw = poly_normalize(w, n), ny = poly_normalize(y, n);
    size_t npr = poly_mul(p, v, nv, w, nw);
    size_t len = npr > ny ? npr : ny;
    for (size_t i = npr; i < len; i++) p[i] = FR_ZERO;
    for (size_t i = 0; i < ny; i++) fr_sub(&p[i], &p[i], &y[i]);
    *np = poly_normalize(p, len);
    fr_t *rem = malloc(sizeof(fr_t) * (2 * (size_t)n + 2));
    size_t nrem;
    poly_divrem(h, nh, rem, &nrem, p, *np, q->target, n + 1);
    free(rem);
    return nrem == 0 ? 0 : -3;
}
API int orc_qap_eval(void *hq, const uint8_t *sol, uint8_t *p_out, size_t *np, uint8_t *h_out, size_t *nh) {
    qap_t *q = hq; uint32_t n = q->n;
    fr_t *s = malloc(sizeof(fr_t) * q->m), *v = malloc(sizeof(fr_t) * n * 3);
    fr_t *p = malloc(sizeof(fr_t) * (2 * (size_t)n + 2)), *h = malloc(sizeof(fr_t) * (2 * (size_t)n + 2));
    for (uint32_t k = 0; k < q->m; k++) fr_from_bytes(&s[k], sol + 32 * (size_t)k);
    int rc = qap_eval(q, s, v, v + n, v + 2 * n, p, np, h, nh);
    for (size_t i = 0; i < *np; i++) fr_to_bytes(p_out + 32 * i, &p[i]);
    for (size_t i = 0; i < *nh; i++) fr_to_bytes(h_out + 32 * i, &h[i]);
    free(s); free(v); free(p); free(h);
    return rc;
}
/* v, w coefficient vectors of QAP.eval (local at QAP.ml:129-130), exposed for stage-level parity */
API void orc_qap_eval_vw(void *hq, const uint8_t *sol, uint8_t *v_out, uint8_t *w_out, uint8_t *y_out) {
    qap_t *q = hq; uint32_t n = q->n;
    fr_t *s = malloc(sizeof(fr_t) * q->m), *v = malloc(sizeof(fr_t) * n * 3);
    fr_t *p = malloc(sizeof(fr_t) * (2 * (size_t)n + 2)), *h = malloc(sizeof(fr_t) * (2 * (size_t)n + 2));
    size_t np, nh;
    for (uint32_t k = 0; k < q->m; k++) fr_from_bytes(&s[k], sol + 32 * (size_t)k);
    qap_eval(q, s, v, v + n, v + 2 * n, p, &np, h, &nh);
    for (uint32_t i = 0; i < n; i++) {
        fr_to_bytes(v_out + 32 * i, &v[i]); fr_to_bytes(w_out + 32 * i, &v[n + i]); fr_to_bytes(y_out + 32 * i, &v[2 * n + i]);
    }
    free(s); free(v); free(p); free(h);
}

/* polynomial.ml:87-92 apply (Horner from the low end with running power) */
static void poly_apply(fr_t *r, const fr_t *f, size_t n, const fr_t *x) {
    fr_t acc = FR_ZERO, xi = FR_ONE;
    for (size_t i = 0; i < n; i++) {
        fr_t t; fr_mul(&t, &f[i], &xi); fr_add(&acc, &acc, &t); fr_mul(&xi, &xi, x);
    }
    *r = acc;
}

/* ================================================================== groth16.ml:45-108  setup
 * toxic = alpha, beta, gamma, delta, tau in the order Fr.gen is called (groth16.ml:51-55).
 * mid[k] != 0 marks k in circuit.mids; the others are v_io (groth16.ml:231).
 * Layout of pk_g1: a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid[#mid in index order]
 *           pk_g2: b2 | d2 | ti2[n+2]
 *           vk_g1: one1 | ltgm_io[#io]          vk_g2: one2 | gm | d      (vkey.ab needs the pairing: Python side) */
API void orc_groth16_setup(void *hq, const uint8_t toxic[5 * 32], const uint8_t *mid,
                           uint8_t *pk_g1, uint8_t *pk_g2, uint8_t *vk_g1, uint8_t *vk_g2) {
    qap_t *q = hq; uint32_t n = q->n, m = q->m;
    fr_t a, b, gm, d, t;
    fr_from_bytes(&a, toxic); fr_from_bytes(&b, toxic + 32); fr_from_bytes(&gm, toxic + 64);
    fr_from_bytes(&d, toxic + 96); fr_from_bytes(&t, toxic + 128);
    g1_t g1, P1; g2_t g2, P2;
    g1_generator(&g1); g2_generator(&g2);
    size_t o1 = 0, o2 = 0;
    g1_mul(&P1, &g1, &a); g1_to_bytes(pk_g1 + 96 * o1++, &P1);            /* a  = g1 alpha   :71 */
    g1_mul(&P1, &g1, &d); g1_to_bytes(pk_g1 + 96 * o1++, &P1);            /* d1 = g1 delta   :72 */
    g1_mul(&P1, &g1, &b); g1_to_bytes(pk_g1 + 96 * o1++, &P1);            /* b1 = g1 beta    :84 */
    orc_g1_powers(pk_g1 + 96 * o1, n + 1, toxic + 128); o1 += n + 2;      /* ti1 = G1.powers (n+1) tau  :73 */
    fr_t dinv, ginv, zt, ztd;
    fr_inv(&dinv, &d); fr_inv(&ginv, &gm);
    poly_apply(&zt, q->target, n + 1, &t); fr_mul(&ztd, &zt, &dinv);      /* Z(tau)/delta    :82 */
    fr_t ti = FR_ONE;
    for (uint32_t i = 0; i + 1 < n; i++) {                                /* tiztd, i in [0, n-2]  :83 */
        fr_t s; fr_mul(&s, &ti, &ztd);
        g1_mul(&P1, &g1, &s); g1_to_bytes(pk_g1 + 96 * o1++, &P1);
        fr_mul(&ti, &ti, &t);
    }
    size_t ov = 0;
    g1_to_bytes(vk_g1 + 96 * ov++, &g1);                                   /* one1 :93 */
    for (uint32_t k = 0; k < m; k++) {
        /* L_k = beta*A_k + alpha*B_k + C_k  :59-68, evaluated at tau */
        fr_t va, vb, vc, l, s;
        poly_apply(&va, q->v + (size_t)k * n, n, &t);
        poly_apply(&vb, q->w + (size_t)k * n, n, &t);
        poly_apply(&vc, q->y + (size_t)k * n, n, &t);
        fr_mul(&va, &va, &b); fr_mul(&vb, &vb, &a);
        fr_add(&l, &va, &vb); fr_add(&l, &l, &vc);
        if (mid[k]) { fr_mul(&s, &l, &dinv); g1_mul(&P1, &g1, &s); g1_to_bytes(pk_g1 + 96 * o1++, &P1); }   /* :74-79 */
        else        { fr_mul(&s, &l, &ginv); g1_mul(&P1, &g1, &s); g1_to_bytes(vk_g1 + 96 * ov++, &P1); }   /* :94-99 */
    }
    g2_mul(&P2, &g2, &b); g2_to_bytes(pk_g2 + 192 * o2++, &P2);           /* b2 :85 */
    g2_mul(&P2, &g2, &d); g2_to_bytes(pk_g2 + 192 * o2++, &P2);           /* d2 :86 */
    orc_g2_powers(pk_g2 + 192 * o2, n + 1, toxic + 128);                  /* ti2 :87 */
    g2_to_bytes(vk_g2, &g2);                                               /* one2 :100 */
    g2_mul(&P2, &g2, &gm); g2_to_bytes(vk_g2 + 192, &P2);                 /* gm   :101 */
    g2_mul(&P2, &g2, &d); g2_to_bytes(vk_g2 + 384, &P2);                  /* d    :102 */
}

/* ================================================================== groth16.ml:116-161  prove (LITERAL)
 * sum_apply_powers = fold over variables of (apply_powers p_k ti) * w_k  -- O(m*n) scalar muls. */
static void g1_apply_powers(g1_t *r, const fr_t *cs, size_t nc, const g1_t *xs) {
    g1_t acc, p; g1_set_inf(&acc);
    for (size_t i = 0; i < nc; i++) { g1_mul(&p, &xs[i], &cs[i]); g1_add(&acc, &p, &acc); }
    *r = acc;
}
static void g2_apply_powers(g2_t *r, const fr_t *cs, size_t nc, const g2_t *xs) {
    g2_t acc, p; g2_set_inf(&acc);
    for (size_t i = 0; i < nc; i++) { g2_mul(&p, &xs[i], &cs[i]); g2_add(&acc, &p, &acc); }
    *r = acc;
}
/* literal = 1: per-variable apply_powers exactly as groth16.ml:116-121.
 * literal = 0: same group elements via apply_powers on the summed coefficient vectors
 *              (sum_k w_k * apply_powers(A_k, ti) = apply_powers(sum_k w_k A_k, ti)). */
API int orc_groth16_prove(void *hq, const uint8_t *pk_g1, const uint8_t *pk_g2, const uint8_t *mid,
                          const uint8_t *sol, const uint8_t r_[32], const uint8_t s_[32], int literal,
                          uint8_t proof_a[96], uint8_t proof_b[192], uint8_t proof_c[96]) {
    qap_t *q = hq; uint32_t n = q->n, m = q->m;
    fr_t *w = malloc(sizeof(fr_t) * m);
    for (uint32_t k = 0; k < m; k++) fr_from_bytes(&w[k], sol + 32 * (size_t)k);
    /* groth16.ml:235-237: (_p, h) = QAP.eval sol qap */
    fr_t *vv = malloc(sizeof(fr_t) * n * 3), *p = malloc(sizeof(fr_t) * (2 * (size_t)n + 2)), *h = malloc(sizeof(fr_t) * (2 * (size_t)n + 2));
    size_t np, nh;
    int rc = qap_eval(q, w, vv, vv + n, vv + 2 * n, p, &np, h, &nh);
    if (rc) { free(w); free(vv); free(p); free(h); return rc; }
    fr_t r, s; fr_from_bytes(&r, r_); fr_from_bytes(&s, s_);
    /* decode the proving key */
    g1_t pa, pd1, pb1, *ti1 = malloc(sizeof(g1_t) * (n + 2)), *tiztd = malloc(sizeof(g1_t) * n);
    g2_t pb2, pd2, *ti2 = malloc(sizeof(g2_t) * (n + 2));
    size_t o = 0;
    g1_from_bytes(&pa, pk_g1 + 96 * o++); g1_from_bytes(&pd1, pk_g1 + 96 * o++); g1_from_bytes(&pb1, pk_g1 + 96 * o++);
    for (uint32_t i = 0; i < n + 2; i++) g1_from_bytes(&ti1[i], pk_g1 + 96 * o++);
    for (uint32_t i = 0; i + 1 < n; i++) g1_from_bytes(&tiztd[i], pk_g1 + 96 * o++);
    const uint8_t *ltd = pk_g1 + 96 * o;
    g2_from_bytes(&pb2, pk_g2); g2_from_bytes(&pd2, pk_g2 + 192);
    for (uint32_t i = 0; i < n + 2; i++) g2_from_bytes(&ti2[i], pk_g2 + 192 * (2 + (size_t)i));

    g1_t SA, SB1, t1; g2_t SB, t2;
    if (literal) {
        g1_set_inf(&SA); g1_set_inf(&SB1); g2_set_inf(&SB);
        for (uint32_t k = 0; k < m; k++) {       /* Var.Map.fold in key order, groth16.ml:117-121 */
            size_t na = poly_normalize(q->v + (size_t)k * n, n), nb = poly_normalize(q->w + (size_t)k * n, n);
            g1_apply_powers(&t1, q->v + (size_t)k * n, na, ti1); g1_mul(&t1, &t1, &w[k]); g1_add(&SA, &t1, &SA);
            g1_apply_powers(&t1, q->w + (size_t)k * n, nb, ti1); g1_mul(&t1, &t1, &w[k]); g1_add(&SB1, &t1, &SB1);
            g2_apply_powers(&t2, q->w + (size_t)k * n, nb, ti2); g2_mul(&t2, &t2, &w[k]); g2_add(&SB, &t2, &SB);
        }
    } else {
        g1_apply_powers(&SA, vv, n, ti1);
        g1_apply_powers(&SB1, vv + n, n, ti1);
        g2_apply_powers(&SB, vv + n, n, ti2);
    }
    g1_t A, B1, C; g2_t B;
    g1_add(&A, &pa, &SA); g1_mul(&t1, &pd1, &r); g1_add(&A, &A, &t1);             /* :128-134 */
    g2_add(&B, &pb2, &SB); g2_mul(&t2, &pd2, &s); g2_add(&B, &B, &t2);            /* :135-141 */
    g1_add(&B1, &pb1, &SB1); g1_mul(&t1, &pd1, &s); g1_add(&B1, &B1, &t1);        /* :146-150 */
    if (nh + 1 > n) { free(w); free(vv); free(p); free(h); free(ti1); free(tiztd); free(ti2); return -2; }  /* invalid_arg "apply_powers" */
    g1_t H; g1_apply_powers(&H, h, nh, tiztd);                                    /* :151 */
    g1_set_inf(&C);
    size_t j = 0;
    for (uint32_t k = 0; k < m; k++)                                              /* :154 dot ltd_mid w|mid */
        if (mid[k]) { g1_from_bytes(&t1, ltd + 96 * j++); g1_mul(&t1, &t1, &w[k]); g1_add(&C, &t1, &C); }
    g1_add(&C, &C, &H);
    g1_mul(&t1, &A, &s); g1_add(&C, &C, &t1);                                     /* :157 */
    g1_mul(&t1, &B1, &r); g1_add(&C, &C, &t1);                                    /* :158 */
    fr_t rs; fr_mul(&rs, &r, &s);
    g1_mul(&t1, &pd1, &rs); g1_neg(&t1, &t1); g1_add(&C, &C, &t1);                /* :159 */
    g1_to_bytes(proof_a, &A); g2_to_bytes(proof_b, &B); g1_to_bytes(proof_c, &C);
    free(w); free(vv); free(p); free(h); free(ti1); free(tiztd); free(ti2);
    return 0;
}

/* ================================================================== trapdoor evaluation (size-independent exact check)
 * NOT a reference algorithm: with the toxic waste known (test keys only) every proof element is a
 * known multiple of the generator, so the expected bytes at ANY n cost O(nnz + n) field operations:
 *   v(tau) = sum_i (L w)_i * l_i(tau),  l_i(tau) = Z(tau) / ((tau - i) * Z'(i)),  Z'(i) = (-1)^(n-1-i) i! (n-1-i)!
 *   A = [alpha + v(tau) + r delta]_1,  B = [beta + w(tau) + s delta]_2,
 *   C = [ (sum_{k in mid} w_k L_k(tau) + h(tau) Z(tau)) / delta + s a + r b - r s delta ]_1,  h(tau) = (v w - y)(tau) / Z(tau)
 * Group elements equal to groth16.ml:123-161 evaluated on the same inputs. */
typedef struct { fr_t *lag; fr_t zt; } lagtab_t;
static void lagrange_at(lagtab_t *T, uint32_t n, const fr_t *tau) {
    T->lag = malloc(sizeof(fr_t) * n);
    fr_t *den = malloc(sizeof(fr_t) * n), *pre = malloc(sizeof(fr_t) * (n + 1));
    fr_t *fact = malloc(sizeof(fr_t) * (n + 1));
    fact[0] = FR_ONE;
    for (uint32_t i = 1; i <= n; i++) { fr_t fi; fr_from_u64(&fi, i); fr_mul(&fact[i], &fact[i - 1], &fi); }
    fr_t zt = FR_ONE;
    for (uint32_t i = 0; i < n; i++) {
        fr_t fi, d; fr_from_u64(&fi, i); fr_sub(&d, tau, &fi); fr_mul(&zt, &zt, &d);
        fr_mul(&den[i], &d, &fact[i]); fr_mul(&den[i], &den[i], &fact[n - 1 - i]);
        if ((n - 1 - i) & 1) fr_neg(&den[i], &den[i]);
    }
    T->zt = zt;
    /* batch inversion */
    pre[0] = FR_ONE;
    for (uint32_t i = 0; i < n; i++) fr_mul(&pre[i + 1], &pre[i], &den[i]);
    fr_t inv; fr_inv(&inv, &pre[n]);
    for (uint32_t i = n; i-- > 0;) {
        fr_t t; fr_mul(&t, &inv, &pre[i]); fr_mul(&inv, &inv, &den[i]);
        fr_mul(&T->lag[i], &t, &zt);
    }
    free(den); free(pre); free(fact);
}
typedef struct { const uint32_t *ptr, *col; const uint8_t *val; } csr_t;
static void csr_at_tau(fr_t *total, fr_t *mid_part, const csr_t *M, uint32_t n, const fr_t *w, const uint8_t *mid, const fr_t *lag) {
    *total = FR_ZERO; *mid_part = FR_ZERO;
    for (uint32_t g = 0; g < n; g++)
        for (uint32_t e = M->ptr[g]; e < M->ptr[g + 1]; e++) {
            fr_t c, t; fr_from_bytes(&c, M->val + 32 * (size_t)e);
            fr_mul(&t, &c, &w[M->col[e]]); fr_mul(&t, &t, &lag[g]);
            fr_add(total, total, &t);
            if (mid[M->col[e]]) fr_add(mid_part, mid_part, &t);
        }
}
API void orc_groth16_prove_trapdoor(uint32_t n, uint32_t m,
                                    const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                                    const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                                    const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val,
                                    const uint8_t *mid, const uint8_t *sol, const uint8_t toxic[5 * 32],
                                    const uint8_t r_[32], const uint8_t s_[32],
                                    uint8_t proof_a[96], uint8_t proof_b[192], uint8_t proof_c[96]) {
    fr_t a, b, d, t, r, s;
    fr_from_bytes(&a, toxic); fr_from_bytes(&b, toxic + 32); fr_from_bytes(&d, toxic + 96); fr_from_bytes(&t, toxic + 128);
    fr_from_bytes(&r, r_); fr_from_bytes(&s, s_);
    fr_t *w = malloc(sizeof(fr_t) * m);
    for (uint32_t k = 0; k < m; k++) fr_from_bytes(&w[k], sol + 32 * (size_t)k);
    lagtab_t T; lagrange_at(&T, n, &t);
    csr_t L = {l_ptr, l_col, l_val}, Rm = {r_ptr, r_col, r_val}, O = {o_ptr, o_col, o_val};
    fr_t vt, vm, wt, wm, yt, ym;
    csr_at_tau(&vt, &vm, &L, n, w, mid, T.lag);
    csr_at_tau(&wt, &wm, &Rm, n, w, mid, T.lag);
    csr_at_tau(&yt, &ym, &O, n, w, mid, T.lag);
    fr_t ea, eb, ec, x, dinv;
    fr_mul(&x, &r, &d); fr_add(&ea, &a, &vt); fr_add(&ea, &ea, &x);          /* alpha + v(tau) + r delta */
    fr_mul(&x, &s, &d); fr_add(&eb, &b, &wt); fr_add(&eb, &eb, &x);          /* beta + w(tau) + s delta  */
    fr_inv(&dinv, &d);
    fr_t lm, hz;                                                             /* sum_mid w_k L_k(tau), h(tau) Z(tau) */
    fr_mul(&lm, &b, &vm); fr_mul(&x, &a, &wm); fr_add(&lm, &lm, &x); fr_add(&lm, &lm, &ym);
    fr_mul(&hz, &vt, &wt); fr_sub(&hz, &hz, &yt);
    fr_add(&ec, &lm, &hz); fr_mul(&ec, &ec, &dinv);
    fr_mul(&x, &s, &ea); fr_add(&ec, &ec, &x);
    fr_mul(&x, &r, &eb); fr_add(&ec, &ec, &x);
    fr_mul(&x, &r, &s); fr_mul(&x, &x, &d); fr_sub(&ec, &ec, &x);
    g1_t g1, P1; g2_t g2, P2;
    g1_generator(&g1); g2_generator(&g2);
    g1_mul(&P1, &g1, &ea); g1_to_bytes(proof_a, &P1);
    g2_mul(&P2, &g2, &eb); g2_to_bytes(proof_b, &P2);
    g1_mul(&P1, &g1, &ec); g1_to_bytes(proof_c, &P1);
    free(w); free(T.lag);
}

/* Proving-key scalars for large synthetic keys (the points are then produced by a fixed-base
 * kernel and spot-checked against orc_g1_mul): exponents of every pkey element in the layout of
 * orc_groth16_setup.  ex_g1: (3 + (n+2) + (n-1) + #mid) * 32 B;  ex_g2: (2 + (n+2)) * 32 B. */
API void orc_groth16_setup_exponents(uint32_t n, uint32_t m,
                                     const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                                     const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                                     const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val,
                                     const uint8_t *mid, const uint8_t toxic[5 * 32],
                                     uint8_t *ex_g1, uint8_t *ex_g2, uint8_t *ex_vk_io /* #io * 32 or NULL */) {
    fr_t a, b, gm, d, t;
    fr_from_bytes(&a, toxic); fr_from_bytes(&b, toxic + 32); fr_from_bytes(&gm, toxic + 64);
    fr_from_bytes(&d, toxic + 96); fr_from_bytes(&t, toxic + 128);
    lagtab_t T; lagrange_at(&T, n, &t);
    fr_t dinv, ginv; fr_inv(&dinv, &d); fr_inv(&ginv, &gm);
    size_t o = 0;
    fr_to_bytes(ex_g1 + 32 * o++, &a); fr_to_bytes(ex_g1 + 32 * o++, &d); fr_to_bytes(ex_g1 + 32 * o++, &b);
    fr_t ti = FR_ONE;
    for (uint32_t i = 0; i < n + 2; i++) { fr_to_bytes(ex_g1 + 32 * o++, &ti); fr_mul(&ti, &ti, &t); }
    fr_t ztd; fr_mul(&ztd, &T.zt, &dinv);
    ti = FR_ONE;
    for (uint32_t i = 0; i + 1 < n; i++) { fr_t x; fr_mul(&x, &ti, &ztd); fr_to_bytes(ex_g1 + 32 * o++, &x); fr_mul(&ti, &ti, &t); }
    /* L_k(tau) = beta A_k(tau) + alpha B_k(tau) + C_k(tau), columns gathered from the CSR rows */
    fr_t *Lk = calloc(m, sizeof(fr_t));
    const csr_t Ms[3] = {{l_ptr, l_col, l_val}, {r_ptr, r_col, r_val}, {o_ptr, o_col, o_val}};
    for (int q = 0; q < 3; q++)
        for (uint32_t g = 0; g < n; g++)
            for (uint32_t e = Ms[q].ptr[g]; e < Ms[q].ptr[g + 1]; e++) {
                fr_t c, x; fr_from_bytes(&c, Ms[q].val + 32 * (size_t)e);
                fr_mul(&x, &c, &T.lag[g]);
                if (q == 0) fr_mul(&x, &x, &b); else if (q == 1) fr_mul(&x, &x, &a);
                fr_add(&Lk[Ms[q].col[e]], &Lk[Ms[q].col[e]], &x);
            }
    size_t oi = 0;
    for (uint32_t k = 0; k < m; k++) {
        fr_t x;
        if (mid[k]) { fr_mul(&x, &Lk[k], &dinv); fr_to_bytes(ex_g1 + 32 * o++, &x); }
        else if (ex_vk_io) { fr_mul(&x, &Lk[k], &ginv); fr_to_bytes(ex_vk_io + 32 * oi++, &x); }
    }
    o = 0;
    fr_to_bytes(ex_g2 + 32 * o++, &b); fr_to_bytes(ex_g2 + 32 * o++, &d);
    ti = FR_ONE;
    for (uint32_t i = 0; i < n + 2; i++) { fr_to_bytes(ex_g2 + 32 * o++, &ti); fr_mul(&ti, &ti, &t); }
    free(Lk); free(T.lag);
}

/* sum_i s_i * k_i mod r : expected exponent of an MSM over bases k_i * G (size-independent MSM check) */
API void orc_fr_dot(uint8_t out[32], const uint8_t *a, const uint8_t *b, size_t n) {
    fr_t acc = FR_ZERO, x, y;
    for (size_t i = 0; i < n; i++) {
        fr_from_bytes(&x, a + 32 * i); fr_from_bytes(&y, b + 32 * i);
        fr_mul(&x, &x, &y); fr_add(&acc, &acc, &x);
    }
    fr_to_bytes(out, &acc);
}

/* R1CS products (L w)_i, (R w)_i, (O w)_i : the values at X = i of QAP.eval's v, w, y (QAP.ml:121-131) */
API void orc_r1cs_spmv(uint32_t n, const uint32_t *ptr, const uint32_t *col, const uint8_t *val,
                       const uint8_t *sol, uint8_t *out) {
    for (uint32_t g = 0; g < n; g++) {
        fr_t acc = FR_ZERO;
        for (uint32_t e = ptr[g]; e < ptr[g + 1]; e++) {
            fr_t c, w; fr_from_bytes(&c, val + 32 * (size_t)e); fr_from_bytes(&w, sol + 32 * (size_t)col[e]);
            fr_mul(&c, &c, &w); fr_add(&acc, &acc, &c);
        }
        fr_to_bytes(out + 32 * (size_t)g, &acc);
    }
}

/* ================================================================== Pinocchio Protocol 2
 * src/pinocchio/pinocchio.ml.  toxic = rv, rw, s, av, aw, ay, b, gm in the order KeyGen.generate draws
 * them (:83-91); ry = rv*rw (:93).  Every key element is a known multiple of a generator, so the key is
 * returned as exponents (the points are `G.of_Fr` of them: oracle g1_mul at small n, the fixed-base
 * kernel at large n -- spot-checked against the oracle).
 *
 * literal = 1: u_k(s) by Poly.apply on the dense QAP polynomials, as map_apply_s does (:104-109);
 * literal = 0: u_k(s) = sum_g M[g,k] * l_g(s) through the Lagrange basis of the integer domain
 *              (same field element, O(nnz + n), any n).
 *
 * G1 exponent layout (n_mid = |mids|, in variable order):
 *   vv[n_mid] | yy[n_mid] | vav[n_mid] | yay[n_mid] | bvwy[n_mid] | si[n+1] | v_all[m] | w_all[m] |
 *   vt | yt | vavt | yayt | vbt | wbt | ybt
 * G2: ww[n_mid] | waw[n_mid] | si2[n+1] | wt | wawt
 * vkey G1: one | aw | bgm | vv_io[n_io] | yy_io[n_io]        vkey G2: one2 | av | ay | gm2 | bgm2 | yt | ww_io[n_io] */
static void pin_uks(fr_t *vk, fr_t *wk, fr_t *yk, fr_t *t_at_s, uint32_t n, uint32_t m, const qap_t *q,
                    const csr_t *Ms, const fr_t *s, int literal) {
    if (literal) {
        for (uint32_t k = 0; k < m; k++) {
            poly_apply(&vk[k], q->v + (size_t)k * n, n, s);
            poly_apply(&wk[k], q->w + (size_t)k * n, n, s);
            poly_apply(&yk[k], q->y + (size_t)k * n, n, s);
        }
        poly_apply(t_at_s, q->target, n + 1, s);
        return;
    }
    lagtab_t T; lagrange_at(&T, n, s);
    *t_at_s = T.zt;
    fr_t *outs[3] = {vk, wk, yk};
    for (int qi = 0; qi < 3; qi++) {
        for (uint32_t k = 0; k < m; k++) outs[qi][k] = FR_ZERO;
        for (uint32_t g = 0; g < n; g++)
            for (uint32_t e = Ms[qi].ptr[g]; e < Ms[qi].ptr[g + 1]; e++) {
                fr_t c, x; fr_from_bytes(&c, Ms[qi].val + 32 * (size_t)e);
                fr_mul(&x, &c, &T.lag[g]);
                fr_add(&outs[qi][Ms[qi].col[e]], &outs[qi][Ms[qi].col[e]], &x);
            }
    }
    free(T.lag);
}
API void orc_pinocchio_keygen_exponents(void *hq /* may be NULL when literal = 0 */, uint32_t n, uint32_t m,
                                        const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                                        const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                                        const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val,
                                        const uint8_t *mid, const uint8_t toxic[8 * 32], int literal,
                                        uint8_t *pk_g1, uint8_t *pk_g2, uint8_t *vk_g1, uint8_t *vk_g2) {
    fr_t rv, rw, s, av, aw, ay, b, gm, ry;
    fr_from_bytes(&rv, toxic); fr_from_bytes(&rw, toxic + 32); fr_from_bytes(&s, toxic + 64);
    fr_from_bytes(&av, toxic + 96); fr_from_bytes(&aw, toxic + 128); fr_from_bytes(&ay, toxic + 160);
    fr_from_bytes(&b, toxic + 192); fr_from_bytes(&gm, toxic + 224);
    fr_mul(&ry, &rv, &rw);                                                   /* :93 */
    const csr_t Ms[3] = {{l_ptr, l_col, l_val}, {r_ptr, r_col, r_val}, {o_ptr, o_col, o_val}};
    fr_t *vk = malloc(sizeof(fr_t) * m), *wk = malloc(sizeof(fr_t) * m), *yk = malloc(sizeof(fr_t) * m), t;
    pin_uks(vk, wk, yk, &t, n, m, (const qap_t *)hq, Ms, &s, literal);
    uint32_t n_mid = 0;
    for (uint32_t k = 0; k < m; k++) n_mid += mid[k] ? 1 : 0;
    uint8_t *o = pk_g1;
#define PUT(ptr, val) do { fr_to_bytes(ptr, &(val)); ptr += 32; } while (0)
    fr_t x, y, z;
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &rv, &vk[k]); PUT(o, x); }                       /* vv   :113 */
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &ry, &yk[k]); PUT(o, x); }                       /* yy   :118 */
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &rv, &vk[k]); fr_mul(&x, &x, &av); PUT(o, x); }  /* vav  :126 */
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &ry, &yk[k]); fr_mul(&x, &x, &ay); PUT(o, x); }  /* yay  :130 */
    for (uint32_t k = 0; k < m; k++) if (mid[k]) {                                                            /* bvwy :137-140 */
        fr_mul(&x, &rv, &vk[k]); fr_mul(&y, &rw, &wk[k]); fr_mul(&z, &ry, &yk[k]);
        fr_add(&x, &x, &y); fr_add(&x, &x, &z); fr_mul(&x, &x, &b); PUT(o, x);
    }
    fr_t si = FR_ONE;
    for (uint32_t i = 0; i <= n; i++) { PUT(o, si); fr_mul(&si, &si, &s); }                                  /* si = powers d s, d = n :133 */
    for (uint32_t k = 0; k < m; k++) PUT(o, vk[k]);                                                          /* v_all :153 */
    for (uint32_t k = 0; k < m; k++) PUT(o, wk[k]);                                                          /* w_all :156 */
    fr_t vt, wt, yt;
    fr_mul(&vt, &rv, &t); fr_mul(&wt, &rw, &t); fr_mul(&yt, &ry, &t);
    PUT(o, vt);                                                                                              /* vt   :142 */
    PUT(o, yt);                                                                                              /* yt   :144 */
    fr_mul(&x, &vt, &av); PUT(o, x);                                                                         /* vavt :145 */
    fr_mul(&x, &yt, &ay); PUT(o, x);                                                                         /* yayt :147 */
    fr_mul(&x, &vt, &b); PUT(o, x);                                                                          /* vbt  :148 */
    fr_mul(&x, &wt, &b); PUT(o, x);                                                                          /* wbt  :149 */
    fr_mul(&x, &yt, &b); PUT(o, x);                                                                          /* ybt  :150 */
    o = pk_g2;
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &rw, &wk[k]); PUT(o, x); }                       /* ww   :116 */
    for (uint32_t k = 0; k < m; k++) if (mid[k]) { fr_mul(&x, &rw, &wk[k]); fr_mul(&x, &x, &aw); PUT(o, x); }  /* waw  :128 */
    si = FR_ONE;
    for (uint32_t i = 0; i <= n; i++) { PUT(o, si); fr_mul(&si, &si, &s); }                                  /* si2  :134 */
    PUT(o, wt);                                                                                              /* wt   :143 */
    fr_mul(&x, &wt, &aw); PUT(o, x);                                                                         /* wawt :146 */
    if (vk_g1 && vk_g2) {
        o = vk_g1;
        fr_t one = FR_ONE, bg; fr_mul(&bg, &gm, &b);
        PUT(o, one); PUT(o, aw); PUT(o, bg);                                                                 /* one, aw, bgm :163-169 */
        for (uint32_t k = 0; k < m; k++) if (!mid[k]) { fr_mul(&x, &rv, &vk[k]); PUT(o, x); }                  /* vv_io :172 */
        for (uint32_t k = 0; k < m; k++) if (!mid[k]) { fr_mul(&x, &ry, &yk[k]); PUT(o, x); }                  /* yy_io :174 */
        o = vk_g2;
        PUT(o, one); PUT(o, av); PUT(o, ay); PUT(o, gm); PUT(o, bg); PUT(o, yt);                             /* one2, av, ay, gm2, bgm2, yt */
        for (uint32_t k = 0; k < m; k++) if (!mid[k]) { fr_mul(&x, &rw, &wk[k]); PUT(o, x); }                  /* ww_io :173 */
    }
#undef PUT
    free(vk); free(wk); free(yk);
}

/* ZKCompute.f (pinocchio.ml:427-514), LITERAL: every dot / apply_powers is a left fold of single
 * scalar multiplications (curve.ml:91-118).  Compute.f (:210-248) is the same with dv = dw = dy = 0
 * and without the `t` term (identical group elements).  Key points in the layout above.
 * proof = vv | ww(G2) | yy | h | vavv | waww(G2) | yayy | bvwy  (Compute.proof field order, :195-208). */
API int orc_pinocchio_prove(void *hq, const uint8_t *pk_g1, const uint8_t *pk_g2, const uint8_t *mid,
                            const uint8_t *sol, const uint8_t dv_[32], const uint8_t dw_[32], const uint8_t dy_[32],
                            uint8_t proof[960]) {
    qap_t *q = hq; uint32_t n = q->n, m = q->m;
    uint32_t n_mid = 0;
    for (uint32_t k = 0; k < m; k++) n_mid += mid[k] ? 1 : 0;
    fr_t *c = malloc(sizeof(fr_t) * m);
    for (uint32_t k = 0; k < m; k++) fr_from_bytes(&c[k], sol + 32 * (size_t)k);
    fr_t *vv_ = malloc(sizeof(fr_t) * n * 3), *p = malloc(sizeof(fr_t) * (2 * (size_t)n + 2)), *h = malloc(sizeof(fr_t) * (2 * (size_t)n + 2));
    size_t np, nh;
    int rc = qap_eval(q, c, vv_, vv_ + n, vv_ + 2 * n, p, &np, h, &nh);          /* :560 */
    if (rc) { free(c); free(vv_); free(p); free(h); return rc; }
    fr_t dv, dw, dy; fr_from_bytes(&dv, dv_); fr_from_bytes(&dw, dw_); fr_from_bytes(&dy, dy_);
    const uint8_t *VV = pk_g1, *YY = VV + 96 * (size_t)n_mid, *VAV = YY + 96 * (size_t)n_mid, *YAY = VAV + 96 * (size_t)n_mid,
                  *BV = YAY + 96 * (size_t)n_mid, *SI = BV + 96 * (size_t)n_mid, *VALL = SI + 96 * (size_t)(n + 1),
                  *WALL = VALL + 96 * (size_t)m, *ONES = WALL + 96 * (size_t)m;
    const uint8_t *WW = pk_g2, *WAW = WW + 192 * (size_t)n_mid, *ONES2 = WAW + 192 * (size_t)n_mid + 192 * (size_t)(n + 1);
    g1_t acc, P1, T; g2_t acc2, P2;
#define DOT1(dst, base)                                                                                    \
    do { g1_set_inf(&acc); size_t j = 0;                                                                   \
         for (uint32_t k = 0; k < m; k++) if (mid[k]) { g1_from_bytes(&P1, (base) + 96 * j++); g1_mul(&P1, &P1, &c[k]); g1_add(&acc, &P1, &acc); } \
         dst = acc; } while (0)
#define DOT2(dst, base)                                                                                    \
    do { g2_set_inf(&acc2); size_t j = 0;                                                                  \
         for (uint32_t k = 0; k < m; k++) if (mid[k]) { g2_from_bytes(&P2, (base) + 192 * j++); g2_mul(&P2, &P2, &c[k]); g2_add(&acc2, &P2, &acc2); } \
         dst = acc2; } while (0)
    g1_t vv, yy, hh, vavv, yayy, bvwy, vall, wall, tt; g2_t ww, waww;
    /* t = apply_powers target si (:431) */
    g1_set_inf(&tt);
    for (uint32_t i = 0; i <= n; i++) { g1_from_bytes(&P1, SI + 96 * (size_t)i); g1_mul(&P1, &P1, &q->target[i]); g1_add(&tt, &P1, &tt); }
    DOT1(vv, VV);   g1_from_bytes(&P1, ONES + 96 * 0); g1_mul(&P1, &P1, &dv); g1_add(&vv, &vv, &P1);          /* :438-439 */
    DOT2(ww, WW);   g2_from_bytes(&P2, ONES2);         g2_mul(&P2, &P2, &dw); g2_add(&ww, &ww, &P2);          /* :442-443 */
    DOT1(yy, YY);   g1_from_bytes(&P1, ONES + 96 * 1); g1_mul(&P1, &P1, &dy); g1_add(&yy, &yy, &P1);          /* :446-447 */
    g1_set_inf(&hh);                                                                                           /* :450 */
    if (nh > n + 1) { rc = -2; goto out; }
    for (size_t i = 0; i < nh; i++) { g1_from_bytes(&P1, SI + 96 * i); g1_mul(&P1, &P1, &h[i]); g1_add(&hh, &P1, &hh); }
    g1_set_inf(&vall); g1_set_inf(&wall);                                                                      /* :483-484 dot over all of c */
    for (uint32_t k = 0; k < m; k++) {
        g1_from_bytes(&P1, VALL + 96 * (size_t)k); g1_mul(&P1, &P1, &c[k]); g1_add(&vall, &P1, &vall);
        g1_from_bytes(&P1, WALL + 96 * (size_t)k); g1_mul(&P1, &P1, &c[k]); g1_add(&wall, &P1, &wall);
    }
    g1_mul(&T, &vall, &dw); g1_add(&hh, &hh, &T);                                                              /* :485 */
    g1_mul(&T, &wall, &dv); g1_add(&hh, &hh, &T);
    g1_mul(&T, &tt, &dv); g1_mul(&T, &T, &dw); g1_add(&hh, &hh, &T);
    g1_generator(&T); g1_mul(&T, &T, &dy); g1_neg(&T, &T); g1_add(&hh, &hh, &T);
    DOT1(vavv, VAV); g1_from_bytes(&P1, ONES + 96 * 2); g1_mul(&P1, &P1, &dv); g1_add(&vavv, &vavv, &P1);      /* :489-490 */
    DOT2(waww, WAW); g2_from_bytes(&P2, ONES2 + 192);   g2_mul(&P2, &P2, &dw); g2_add(&waww, &waww, &P2);      /* :493-494 */
    DOT1(yayy, YAY); g1_from_bytes(&P1, ONES + 96 * 3); g1_mul(&P1, &P1, &dy); g1_add(&yayy, &yayy, &P1);      /* :497-498 */
    DOT1(bvwy, BV);                                                                                            /* :500-505 */
    g1_from_bytes(&P1, ONES + 96 * 4); g1_mul(&P1, &P1, &dv); g1_add(&bvwy, &bvwy, &P1);
    g1_from_bytes(&P1, ONES + 96 * 5); g1_mul(&P1, &P1, &dw); g1_add(&bvwy, &bvwy, &P1);
    g1_from_bytes(&P1, ONES + 96 * 6); g1_mul(&P1, &P1, &dy); g1_add(&bvwy, &bvwy, &P1);
    g1_to_bytes(proof, &vv); g2_to_bytes(proof + 96, &ww); g1_to_bytes(proof + 288, &yy); g1_to_bytes(proof + 384, &hh);
    g1_to_bytes(proof + 480, &vavv); g2_to_bytes(proof + 576, &waww); g1_to_bytes(proof + 768, &yayy); g1_to_bytes(proof + 864, &bvwy);
out:
    free(c); free(vv_); free(p); free(h);
    return rc;
#undef DOT1
#undef DOT2
}

/* Trapdoor evaluation of the ZK proof (size-independent exact check, NOT a reference algorithm):
 *   vv' = [rv (v_mid(s) + dv t)]_1, ww' = [rw (w_mid(s) + dw t)]_2, yy' = [ry (y_mid(s) + dy t)]_1,
 *   h'  = [h(s) + dw v(s) + dv w(s) + dv dw t - dy]_1,  h(s) = (v(s) w(s) - y(s)) / t,
 *   vavv' = av vv', waww' = aw ww', yayy' = ay yy',
 *   bvwy' = [b (rv v_mid + rw w_mid + ry y_mid) + b t (rv dv + rw dw + ry dy)]_1 */
API void orc_pinocchio_prove_trapdoor(uint32_t n, uint32_t m,
                                      const uint32_t *l_ptr, const uint32_t *l_col, const uint8_t *l_val,
                                      const uint32_t *r_ptr, const uint32_t *r_col, const uint8_t *r_val,
                                      const uint32_t *o_ptr, const uint32_t *o_col, const uint8_t *o_val,
                                      const uint8_t *mid, const uint8_t *sol, const uint8_t toxic[8 * 32],
                                      const uint8_t dv_[32], const uint8_t dw_[32], const uint8_t dy_[32], uint8_t proof[960]) {
    fr_t rv, rw, s, av, aw, ay, b, ry, dv, dw, dy;
    fr_from_bytes(&rv, toxic); fr_from_bytes(&rw, toxic + 32); fr_from_bytes(&s, toxic + 64);
    fr_from_bytes(&av, toxic + 96); fr_from_bytes(&aw, toxic + 128); fr_from_bytes(&ay, toxic + 160);
    fr_from_bytes(&b, toxic + 192);
    fr_mul(&ry, &rv, &rw);
    fr_from_bytes(&dv, dv_); fr_from_bytes(&dw, dw_); fr_from_bytes(&dy, dy_);
    fr_t *c = malloc(sizeof(fr_t) * m);
    for (uint32_t k = 0; k < m; k++) fr_from_bytes(&c[k], sol + 32 * (size_t)k);
    lagtab_t T; lagrange_at(&T, n, &s);
    csr_t L = {l_ptr, l_col, l_val}, Rm = {r_ptr, r_col, r_val}, O = {o_ptr, o_col, o_val};
    fr_t vs, vm, ws, wm, ys, ym, t = T.zt;
    csr_at_tau(&vs, &vm, &L, n, c, mid, T.lag);
    csr_at_tau(&ws, &wm, &Rm, n, c, mid, T.lag);
    csr_at_tau(&ys, &ym, &O, n, c, mid, T.lag);
    fr_t e_vv, e_ww, e_yy, e_h, x, y, tinv;
    fr_mul(&x, &dv, &t); fr_add(&x, &x, &vm); fr_mul(&e_vv, &rv, &x);
    fr_mul(&x, &dw, &t); fr_add(&x, &x, &wm); fr_mul(&e_ww, &rw, &x);
    fr_mul(&x, &dy, &t); fr_add(&x, &x, &ym); fr_mul(&e_yy, &ry, &x);
    fr_inv(&tinv, &t);
    fr_mul(&e_h, &vs, &ws); fr_sub(&e_h, &e_h, &ys); fr_mul(&e_h, &e_h, &tinv);
    fr_mul(&x, &dw, &vs); fr_add(&e_h, &e_h, &x);
    fr_mul(&x, &dv, &ws); fr_add(&e_h, &e_h, &x);
    fr_mul(&x, &dv, &dw); fr_mul(&x, &x, &t); fr_add(&e_h, &e_h, &x);
    fr_sub(&e_h, &e_h, &dy);
    fr_t e_b;
    fr_mul(&x, &rv, &vm); fr_mul(&y, &rw, &wm); fr_add(&x, &x, &y); fr_mul(&y, &ry, &ym); fr_add(&x, &x, &y); fr_mul(&e_b, &b, &x);
    fr_mul(&x, &rv, &dv); fr_mul(&y, &rw, &dw); fr_add(&x, &x, &y); fr_mul(&y, &ry, &dy); fr_add(&x, &x, &y);
    fr_mul(&x, &x, &t); fr_mul(&x, &x, &b); fr_add(&e_b, &e_b, &x);
    g1_t g1, P1; g2_t g2, P2;
    g1_generator(&g1); g2_generator(&g2);
    g1_mul(&P1, &g1, &e_vv); g1_to_bytes(proof, &P1);
    g2_mul(&P2, &g2, &e_ww); g2_to_bytes(proof + 96, &P2);
    g1_mul(&P1, &g1, &e_yy); g1_to_bytes(proof + 288, &P1);
    g1_mul(&P1, &g1, &e_h); g1_to_bytes(proof + 384, &P1);
    fr_mul(&x, &e_vv, &av); g1_mul(&P1, &g1, &x); g1_to_bytes(proof + 480, &P1);
    fr_mul(&x, &e_ww, &aw); g2_mul(&P2, &g2, &x); g2_to_bytes(proof + 576, &P2);
    fr_mul(&x, &e_yy, &ay); g1_mul(&P1, &g1, &x); g1_to_bytes(proof + 768, &P1);
    g1_mul(&P1, &g1, &e_b); g1_to_bytes(proof + 864, &P1);
    free(c); free(T.lag);
}
