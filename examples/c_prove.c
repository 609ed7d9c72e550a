/* The whole Groth16 prove path from a plain C99 host: upload the README circuit (`x*x*x + x + 3`, README.md:49) and its proving key in
 * the reference's layout (groth16.ml:24-34), prove (groth16.ml:235-237), derive the Lagrange form on the device, read the pools back,
 * prove again -- every output compared with the first-principles bytes of examples/readme_fixture.h (tests/golden/readme_groth16_key.json).
 * With arguments -- HIP device indices, e.g. `c_prove 0 1` -- the SAME calls run on a multi-device key: zk_set_device_list shards the key over
 * the listed devices behind the one handle (an index may repeat: `c_prove 0 0` puts two shards on one card, which is how a one-GPU box runs it).
 * Needs a GPU; tests/test_golden_key.py builds and runs it on the GPU box. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "readme_fixture.h"
#include "zkmi355x.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != ZK_OK) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, zk_last_error()); return 1; } \
    } while (0)

static void fr_small(uint8_t out[32], uint32_t v) { memset(out, 0, 32); out[0] = (uint8_t)v; }

int main(int argc, char** argv) {
    /* gates (Gate.compare order): c4 = input*input ; c5 = c4*input ; v6 = (c5 + input + 3 ONE) * (1 ONE); variables ONE, c4, c5, input, v6 */
    static const uint32_t l_ptr[4] = {0, 1, 2, 5}, l_col[5] = {3, 1, 0, 2, 3};
    static const uint32_t r_ptr[4] = {0, 1, 2, 3}, r_col[3] = {3, 3, 0};
    static const uint32_t o_ptr[4] = {0, 1, 2, 3}, o_col[3] = {1, 2, 4};
    uint8_t l_val[5 * 32], r_val[3 * 32], o_val[3 * 32], proof[384], pool1[11 * 96], pool2[5 * 192];
    const uint32_t lc[5] = {1, 1, 3, 1, 1};
    zk_csr L, R, O;
    uint64_t h = 0;
    size_t cnt = 0;
    int i;
    for (i = 0; i < 5; i++) fr_small(l_val + 32 * i, lc[i]);
    for (i = 0; i < 3; i++) { fr_small(r_val + 32 * i, 1); fr_small(o_val + 32 * i, 1); }
    L.row_ptr = l_ptr; L.col = l_col; L.val = l_val;
    R.row_ptr = r_ptr; R.col = r_col; R.val = r_val;
    O.row_ptr = o_ptr; O.col = o_col; O.val = o_val;
    if (argc > 1) {          /* N GPUs behind the one handle: the only line an OCaml host adds */
        int32_t devs[16];
        uint32_t nd = 0;
        for (i = 1; i < argc && nd < 16; i++) devs[nd++] = (int32_t)atoi(argv[i]);
        CHECK(zk_set_device_list(devs, nd));
    } else
        CHECK(zk_init(0));
    CHECK(zk_groth16_pk_upload(3, 5, &L, &R, &O, FIX_MID, FIX_PK_G1, sizeof FIX_PK_G1 / 96, FIX_PK_G2, sizeof FIX_PK_G2 / 192, &h));
    CHECK(zk_groth16_prove(h, FIX_WITNESS, FIX_R, FIX_S, proof));
    if (memcmp(proof, FIX_PROOF, 384)) { fprintf(stderr, "proof from the uploaded key differs from the fixture\n"); return 2; }
    CHECK(zk_groth16_pk_derive_lagrange(h));
    CHECK(zk_groth16_pool_points(h, 1, pool1, sizeof pool1 / 96, &cnt));
    if (cnt != sizeof FIX_LAG_G1 / 96 || memcmp(pool1, FIX_LAG_G1, sizeof FIX_LAG_G1)) { fprintf(stderr, "derived G1 pool differs\n"); return 3; }
    CHECK(zk_groth16_pool_points(h, 2, pool2, sizeof pool2 / 192, &cnt));
    if (cnt != sizeof FIX_LAG_G2 / 192 || memcmp(pool2, FIX_LAG_G2, sizeof FIX_LAG_G2)) { fprintf(stderr, "derived G2 pool differs\n"); return 4; }
    memset(proof, 0, sizeof proof);
    CHECK(zk_groth16_prove(h, FIX_WITNESS, FIX_R, FIX_S, proof));
    if (memcmp(proof, FIX_PROOF, 384)) { fprintf(stderr, "proof from the derived key differs from the fixture\n"); return 5; }
    CHECK(zk_groth16_pk_free(h));
    printf("c-prove ok (%d device entr%s): proofs from the uploaded and the derived key equal the first-principles fixture\n", argc > 1 ? argc - 1 : 1, argc > 2 ? "ies" : "y");
    return 0;
}
