/* A plain C99 host of the C-ABI (include/zkmi355x.h): what a cgo / ctypes / OCaml-ctypes binding sees.  No GPU is needed for what it
 * calls: the error strings, the handle check and the pure host arithmetic of the sharding rule.  tests/test_abi.py compiles it with
 * `gcc -std=c99 -pedantic`, links it against zukelang_amd/libzkmi355x.so and runs it.
 * With a GPU the same program would go on: zk_init(0); zk_groth16_pk_upload(n, m, &L, &R, &O, mid, pk_g1, p1, pk_g2, p2, &h);
 * zk_groth16_pk_derive_lagrange(h); zk_groth16_prove(h, witness, r, s, proof);  (groth16.ml:235-237). */
#include <stdint.h>
#include <stdio.h>
#include <string.h>

#include "zkmi355x.h"

int main(void) {
    uint64_t lo = 0, hi = 0, covered = 0;
    unsigned rank;
    if (zk_groth16_pk_free(12345) != ZK_ERR_HANDLE) return 1;                 /* unknown handle */
    if (strlen(zk_strerror(ZK_ERR_HANDLE)) == 0 || strlen(zk_last_error()) == 0) return 2;
    /* the G1 pool of a 2^16-constraint key over 8 ranks: 3 + (n + 2) + (n - 1) + n_mid points, the first 3 + (n + 2) serve two products */
    for (rank = 0; rank < 8; rank++) {
        if (zk_groth16_shard_range(3 + 65538 + 65535 + 65537, 3 + 65538, rank, 8, &lo, &hi) != ZK_OK) return 3;
        if (lo != covered || hi <= lo) return 4;
        covered = hi;
    }
    if (covered != 3 + 65538 + 65535 + 65537) return 5;
    if (zk_groth16_shard_range(10, 0, 8, 8, &lo, &hi) != ZK_ERR_ARG) return 6;
    printf("c-host ok: %s\n", zk_strerror(ZK_OK));
    return 0;
}
