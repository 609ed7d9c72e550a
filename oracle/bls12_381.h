/* TEST INFRASTRUCTURE ONLY -- CPU oracle for the zukelang hot path.
 *
 * Plain C (gcc, unsigned __int128) BLS12-381 arithmetic: Fr, Fp, Fp2, G1, G2,
 * ZCash-style serialization.  The reference delegates all of this to opam
 * bls12-381 = 6.1.0 (blst binding; zukelang.opam:15) whose source is NOT under
 * /root/reference; this file restates the published algorithms (Montgomery
 * CIOS, Jacobian short-Weierstrass group law for a = 0) and is anchored on the
 * reference's call sites: src/lib/zk/curve.ml:123-140 (Fr ops), :159-191
 * (group ops), :199-219 (serialization).
 *
 * PARITY UNPINNED by the reference: it holds no golden vector.
 * Pinned instead by oracle/pyref.py (Python big ints) in tests/test_oracle.py.
 *
 * Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this; zukelang_amd/ (the product) never does.
 */
#ifndef ZK_ORACLE_BLS12_381_H
#define ZK_ORACLE_BLS12_381_H
#include <stdint.h>
#include <stddef.h>

typedef struct { uint64_t l[4]; } fr_t;   /* Montgomery form, R = 2^256 */
typedef struct { uint64_t l[6]; } fp_t;   /* Montgomery form, R = 2^384 */
typedef struct { fp_t c0, c1; } fp2_t;    /* c0 + c1*u, u^2 = -1 */

typedef struct { fp_t x, y, z; } g1_t;    /* Jacobian; z = 0 is infinity */
typedef struct { fp2_t x, y, z; } g2_t;

/* Fr */
void fr_from_bytes(fr_t *r, const uint8_t in[32]);      /* 32 B little-endian canonical */
void fr_to_bytes(uint8_t out[32], const fr_t *a);
void fr_from_u64(fr_t *r, uint64_t v);
void fr_add(fr_t *r, const fr_t *a, const fr_t *b);
void fr_sub(fr_t *r, const fr_t *a, const fr_t *b);
void fr_neg(fr_t *r, const fr_t *a);
void fr_mul(fr_t *r, const fr_t *a, const fr_t *b);
void fr_inv(fr_t *r, const fr_t *a);
void fr_pow_u64(fr_t *r, const fr_t *a, uint64_t e);
int  fr_is_zero(const fr_t *a);
int  fr_eq(const fr_t *a, const fr_t *b);
extern const fr_t FR_ZERO, FR_ONE;
void fr_omega(fr_t *r);   /* 5^((r-1)/2^32), FFT.ml:208-219 */

/* Fp / Fp2 */
void fp_from_be(fp_t *r, const uint8_t in[48]);
void fp_to_be(uint8_t out[48], const fp_t *a);
void fp_add(fp_t *r, const fp_t *a, const fp_t *b);
void fp_sub(fp_t *r, const fp_t *a, const fp_t *b);
void fp_neg(fp_t *r, const fp_t *a);
void fp_mul(fp_t *r, const fp_t *a, const fp_t *b);
void fp_inv(fp_t *r, const fp_t *a);
int  fp_is_zero(const fp_t *a);
int  fp_eq(const fp_t *a, const fp_t *b);
int  fp_is_lex_largest(const fp_t *a);  /* a > (p-1)/2 */

/* G1 / G2 */
void g1_set_inf(g1_t *r);
void g1_generator(g1_t *r);
int  g1_is_inf(const g1_t *a);
void g1_add(g1_t *r, const g1_t *a, const g1_t *b);
void g1_dbl(g1_t *r, const g1_t *a);
void g1_neg(g1_t *r, const g1_t *a);
void g1_mul(g1_t *r, const g1_t *a, const fr_t *k);     /* double-and-add on canonical bits */
int  g1_eq(const g1_t *a, const g1_t *b);
int  g1_from_bytes(g1_t *r, const uint8_t in[96]);     /* 0 ok, -1 not on curve */
void g1_to_bytes(uint8_t out[96], const g1_t *a);
void g1_compress(uint8_t out[48], const g1_t *a);

void g2_set_inf(g2_t *r);
void g2_generator(g2_t *r);
int  g2_is_inf(const g2_t *a);
void g2_add(g2_t *r, const g2_t *a, const g2_t *b);
void g2_dbl(g2_t *r, const g2_t *a);
void g2_neg(g2_t *r, const g2_t *a);
void g2_mul(g2_t *r, const g2_t *a, const fr_t *k);
int  g2_eq(const g2_t *a, const g2_t *b);
int  g2_from_bytes(g2_t *r, const uint8_t in[192]);
void g2_to_bytes(uint8_t out[192], const g2_t *a);
void g2_compress(uint8_t out[96], const g2_t *a);

#endif
