/* Pinocchio Protocol 2 from a plain C99 host, in exactly the call sequence the OCaml shim ocaml/pinocchio_mi355x.ml makes
 * (Pinocchio.Make(C).{NonZK, ZK} : Protocol.S, src/pinocchio/pinocchio.mli:3-15):
 *   keygen  (pinocchio.ml:77-189)  the exponents are host integers; ALL G1 points of both keys come from one zk_g1_of_fr call, all G2 points
 *                                  from one zk_g2_of_fr call; then zk_pinocchio_pk_upload registers circuit + evaluation key
 *   ZK.prove    (:559-561, ZKCompute.f :427-514)  zk_pinocchio_prove with dv, dw, dy
 *   NonZK.prove (:536-538, Compute.f :210-248)    the same call with dv = dw = dy = 0
 *   verify  (Verify.f :254-420)    zk_pinocchio_verify on the flattened verification key; a wrong public input is rejected
 *   a long-lived key: zk_pinocchio_pk_derive_lagrange, after which the same calls give the same bytes
 * With arguments -- HIP device indices, e.g. `c_pinocchio 0 1` -- the SAME calls run on a multi-device key: zk_set_device_list cuts every pool of the
 * key over the listed devices behind the one handle (an index may repeat: several shards on one card), and every byte stays the same.
 * on the README circuit `x*x*x + x + 3` (README.md:49).  Every output is compared with the first-principles bytes of
 * examples/readme_pinocchio_fixture.h (tests/golden/readme_pinocchio_key.json, written by tests/golden/make_readme_pinocchio.py from Python
 * integers).  Needs a GPU; tests/test_golden_key.py builds it with -std=c99 -pedantic -Werror and runs it on the GPU box. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "readme_pinocchio_fixture.h"
#include "zkmi355x.h"

#define CHECK(call)                                                                                   \
    do {                                                                                              \
        int rc_ = (call);                                                                             \
        if (rc_ != ZK_OK) { fprintf(stderr, "%s -> %d (%s)\n", #call, rc_, zk_last_error()); return 1; } \
    } while (0)
#define SAME(got, want, what)                                                        \
    do {                                                                             \
        if (memcmp((got), (want), sizeof(want))) { fprintf(stderr, "%s differs from the fixture\n", what); return 2; } \
    } while (0)

enum { N = 3, M = 5, N_MID = 3, N_IO = 2, PK1 = 5 * N_MID + (N + 1) + 2 * M + 7, PK2 = 2 * N_MID + (N + 1) + 2, VK1 = 3 + 2 * N_IO, VK2 = 6 + N_IO };

static void fr_small(uint8_t out[32], uint32_t v) { memset(out, 0, 32); out[0] = (uint8_t)v; }

int main(int argc, char** argv) {
    /* gates (Gate.compare order): c4 = input*input ; c5 = c4*input ; v6 = (c5 + input + 3 ONE) * (1 ONE); variables ONE, c4, c5, input, v6 */
    static const uint32_t l_ptr[4] = {0, 1, 2, 5}, l_col[5] = {3, 1, 0, 2, 3};
    static const uint32_t r_ptr[4] = {0, 1, 2, 3}, r_col[3] = {3, 3, 0};
    static const uint32_t o_ptr[4] = {0, 1, 2, 3}, o_col[3] = {1, 2, 4};
    static const uint32_t lc[5] = {1, 1, 3, 1, 1};
    static uint8_t g1[(PK1 + VK1 - 1) * 96], g2[(PK2 + VK2 - 1) * 192], vk1[VK1 * 96], vk2[VK2 * 192], pool[(N - 1 + 2 + 2 * M) * 96];
    uint8_t l_val[5 * 32], r_val[3 * 32], o_val[3 * 32], proof[960], zero[32], wrong_io[N_IO * 32];
    zk_csr L, R, O;
    uint64_t h = 0;
    size_t cnt = 0;
    int i, ok = 0;
    for (i = 0; i < 5; i++) fr_small(l_val + 32 * i, lc[i]);
    for (i = 0; i < 3; i++) { fr_small(r_val + 32 * i, 1); fr_small(o_val + 32 * i, 1); }
    L.row_ptr = l_ptr; L.col = l_col; L.val = l_val;
    R.row_ptr = r_ptr; R.col = r_col; R.val = r_val;
    O.row_ptr = o_ptr; O.col = o_col; O.val = o_val;
    memset(zero, 0, sizeof zero);
    if (argc > 1) {          /* N GPUs behind the one handle: the only line a host adds */
        int32_t devs[16];
        uint32_t nd = 0;
        for (i = 1; i < argc && nd < 16; i++) devs[nd++] = (int32_t)atoi(argv[i]);
        CHECK(zk_set_device_list(devs, nd));
    } else
        CHECK(zk_init(0));

    /* keygen: exponents -> points, evaluation key first, then the verification key's points (its `one` is the generator, not a product) */
    CHECK(zk_g1_of_fr(PFIX_PK_EXP_G1, sizeof PFIX_PK_EXP_G1 / 32, g1));
    CHECK(zk_g2_of_fr(PFIX_PK_EXP_G2, sizeof PFIX_PK_EXP_G2 / 32, g2));
    if (memcmp(g1, PFIX_PK_G1, sizeof PFIX_PK_G1) || memcmp(g2, PFIX_PK_G2, sizeof PFIX_PK_G2)) { fprintf(stderr, "evaluation key differs from the fixture\n"); return 2; }
    memcpy(vk1, PFIX_VK_G1, 96);                     /* G1.one */
    memcpy(vk1 + 96, g1 + PK1 * 96, (VK1 - 1) * 96);
    memcpy(vk2, PFIX_VK_G2, 192);                    /* G2.one */
    memcpy(vk2 + 192, g2 + PK2 * 192, (VK2 - 1) * 192);
    SAME(vk1, PFIX_VK_G1, "verification key (G1)");
    SAME(vk2, PFIX_VK_G2, "verification key (G2)");

    CHECK(zk_pinocchio_pk_upload(N, M, &L, &R, &O, PFIX_MID, g1, PK1, g2, PK2, &h));

    /* ZK.prove, NonZK.prove, verify */
    CHECK(zk_pinocchio_prove(h, PFIX_WITNESS, PFIX_DELTAS, PFIX_DELTAS + 32, PFIX_DELTAS + 64, proof));
    SAME(proof, PFIX_PROOF, "ZK proof");
    CHECK(zk_pinocchio_verify(vk1, vk2, PFIX_IO, N_IO, proof, &ok));
    if (!ok) { fprintf(stderr, "the ZK proof does not verify\n"); return 3; }
    memcpy(wrong_io, PFIX_IO, sizeof wrong_io);
    wrong_io[32] ^= 1;                               /* the output v6, off by one */
    CHECK(zk_pinocchio_verify(vk1, vk2, wrong_io, N_IO, proof, &ok));
    if (ok) { fprintf(stderr, "a wrong public input verified\n"); return 3; }
    CHECK(zk_pinocchio_prove(h, PFIX_WITNESS, zero, zero, zero, proof));
    SAME(proof, PFIX_PROOF_NONZK, "NonZK proof");
    CHECK(zk_pinocchio_verify(vk1, vk2, PFIX_IO, N_IO, proof, &ok));
    if (!ok) { fprintf(stderr, "the NonZK proof does not verify\n"); return 3; }

    /* the h bases derived on the device (the shim's `derive_lagrange_on_upload`): same proofs */
    CHECK(zk_pinocchio_pk_derive_lagrange(h));
    CHECK(zk_pinocchio_pool_points(h, 5, pool, sizeof pool / 96, &cnt));
    if (cnt != sizeof PFIX_DERIVED_H_POOL / 96) { fprintf(stderr, "derived h pool has %lu points\n", (unsigned long)cnt); return 4; }
    SAME(pool, PFIX_DERIVED_H_POOL, "derived h pool");
    memset(proof, 0, sizeof proof);
    CHECK(zk_pinocchio_prove(h, PFIX_WITNESS, PFIX_DELTAS, PFIX_DELTAS + 32, PFIX_DELTAS + 64, proof));
    SAME(proof, PFIX_PROOF, "ZK proof from the derived key");

    /* pipelined form: resident witness, two proofs in flight */
    CHECK(zk_pinocchio_reserve_slots(h, 2));
    CHECK(zk_pinocchio_set_witness(h, PFIX_WITNESS));
    CHECK(zk_pinocchio_prove_async(h, NULL, PFIX_DELTAS, PFIX_DELTAS + 32, PFIX_DELTAS + 64, 0));
    CHECK(zk_pinocchio_prove_async(h, NULL, zero, zero, zero, 1));
    CHECK(zk_pinocchio_prove_wait(h, 0, proof));
    SAME(proof, PFIX_PROOF, "ZK proof (slot 0)");
    CHECK(zk_pinocchio_prove_wait(h, 1, proof));
    SAME(proof, PFIX_PROOF_NONZK, "NonZK proof (slot 1)");
    CHECK(zk_pinocchio_pk_free(h));
    printf("c-pinocchio ok (%d device entr%s): keygen, ZK / NonZK proofs (uploaded and derived key, blocking and pipelined) and verify equal the first-principles fixture\n", argc > 1 ? argc - 1 : 1, argc > 2 ? "ies" : "y");
    return 0;
}
