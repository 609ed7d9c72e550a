#!/usr/bin/env python3
"""Randomised parity soak on the GPU box (beyond the fixed cases of tests/): MSMs with adversarial scalar /
point patterns against the oracle's naive fold, Groth16 proofs of random sizes against the trapdoor evaluation
(power-form and Lagrange-form keys).  Test infrastructure (it uses the oracle); not collected by pytest.
Usage: python tests/soak_parity.py [seconds] [seed]"""
import os, random, sys, time
os.environ.setdefault("ZK_TEST_FORMS", "1")      # before the library loads: kernel-form switches (ZK_FR_RNS ...) are read per call, not cached (csrc/zk_common.h)
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))      # ROOT = repo root (this file lives in tests/)
import numpy as np
import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC
from zukelang_amd.curve import G1, G2
from zukelang_amd.groth16 import Groth16
from zukelang_amd import pinocchio as PIN
from zukelang_amd import _lib

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rnd = random.Random(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
frb = P.fr_to_bytes
t_end = time.time() + budget
n_msm = n_g16 = n_pin = n_der = n_ba = n_multi = n_rns = n_sort = n_pin_multi = n_dec = 0
_lib.check(_lib.lib().zk_init(0))
while time.time() < t_end:
    # ---- MSM with duplicates, negations, identity, tiny / huge / zero scalars
    G, naive, gen, mul = rnd.choice([(G1, O.g1_msm_naive, O.g1_generator, O.g1_mul), (G2, O.g2_msm_naive, O.g2_generator, O.g2_mul)])
    n = rnd.choice([1, 2, 3, 17, 64, 255, 256, 700, 1500])
    B = G.POINT_BYTES
    uniq = [mul(gen(), frb(rnd.randrange(1, P.R))) for _ in range(min(n, 6))]
    inf = bytes([0x40]) + bytes(B - 1)
    pts, scs = [], []
    for i in range(n):
        pts.append(rnd.choice(uniq + [inf]) if rnd.random() < 0.7 else mul(gen(), frb(rnd.randrange(1, 1 << 64))))
        kind = rnd.random()
        scs.append(0 if kind < 0.15 else 1 if kind < 0.3 else P.R - 1 if kind < 0.4 else rnd.randrange(1 << 16) if kind < 0.55 else
                   (P.R - rnd.randrange(1, 1 << 20)) if kind < 0.7 else rnd.randrange(P.R))
    bases, scalars = b"".join(pts), b"".join(frb(s) for s in scs)
    rc, ref = naive(bases, scalars)
    assert rc == 0
    for c in (0, rnd.choice([2, 5, 9, 13, 16])):
        got = bytes(G.apply_powers(scalars, np.frombuffer(bases, dtype=np.uint8), c))
        assert got == ref, ("MSM mismatch", G.__name__, n, c)
    if n_msm % 2 == 0:
        # round 2: the same adversarial base set through the resident-key machinery (window tables, one bucket set) with batch-affine rounds
        os.environ["ZK_MSM_API_PRECOMP"] = "1"
        os.environ["ZK_MSM_BA_ROUNDS"] = str(rnd.choice([0, 1, 3, 6]))
        try:
            got = bytes(G.apply_powers(scalars, np.frombuffer(bases, dtype=np.uint8), rnd.choice([3, 4, 8])))
        finally:
            del os.environ["ZK_MSM_API_PRECOMP"], os.environ["ZK_MSM_BA_ROUNDS"]
        assert got == ref, ("resident-path MSM mismatch", G.__name__, n)
    n_msm += 1
    # ---- Groth16 at a random size, both key forms
    n = 2 * rnd.randrange(1, 1500)
    cs, w = RC.iterated_cubic(n, rnd.randrange(1, P.R))
    toxic = [rnd.randrange(1, P.R) for _ in range(5)]
    it = iter(toxic)
    pk, vk = Groth16.keygen(lambda: next(it), cs, lagrange=True)
    r, s = rnd.randrange(P.R), rnd.randrange(P.R)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, b"".join(frb(x) for x in w), b"".join(frb(x) for x in toxic), frb(r), frb(s))
    for lag in (False, True):
        pr = Groth16(cs, pk, lagrange=lag)
        p = pr.prove_rs(w, r, s)
        assert (p.a, p.b, p.c) == exp, ("Groth16 mismatch", n, lag)
        if not lag and n_g16 % 3 == 0:
            # round 2: the Lagrange form DERIVED on the device from the tau-power key must be the keygen's, byte for byte, and prove the same
            pr.derive_lagrange()
            assert bytes(pr.pool_points(1)) == bytes(pk.lag_g1) and bytes(pr.pool_points(2)) == bytes(pk.lag_g2), ("derived pools differ", n)
            p = pr.prove_rs(w, r, s)
            assert (p.a, p.b, p.c) == exp, ("Groth16 mismatch after the derivation", n)
            n_der += 1
        pr.close()
    if n_g16 % 5 == 0:
        # round 2: the optional batch-affine accumulation, forced rounds (small keys), both curves
        os.environ["ZK_MSM_BA_ROUNDS"] = str(rnd.choice([1, 2, 4]))
        pr = Groth16(cs, pk)
        del os.environ["ZK_MSM_BA_ROUNDS"]
        p = pr.prove_rs(w, r, s)
        assert (p.a, p.b, p.c) == exp, ("Groth16 mismatch with batch-affine rounds", n)
        pr.close()
        n_ba += 1
    if n_g16 % 4 == 1:
        # round 4: the same key behind ONE handle over a device list (the card listed 2 .. 4 times: csrc/groth16_multi.hip), a few proofs in flight with
        # the owner of the Fr stage rotating, then the multi-device derivation -- bytes as on one device
        devs = [0] * rnd.choice([2, 3, 4])
        _lib.set_device_list(devs)
        try:
            pr = Groth16(cs, pk)
            pr.set_witness(w)
            k = rnd.choice([1, 2, 3])
            rss = [(r, s)] + [(rnd.randrange(P.R), rnd.randrange(P.R)) for _ in range(k - 1)]
            for slot, (r_, s_) in enumerate(rss):
                pr.prove_async(None, r_, s_, slot)
            for slot, (r_, s_) in enumerate(rss):
                p = pr.prove_wait(slot)
                e_ = exp if slot == 0 else O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, b"".join(frb(x) for x in w), b"".join(frb(x) for x in toxic), frb(r_), frb(s_))
                assert (p.a, p.b, p.c) == e_, ("multi-device Groth16 mismatch", n, devs, slot)
            if n_multi % 2 == 0:
                pr.derive_lagrange()
                assert bytes(pr.pool_points(1)) == bytes(pk.lag_g1) and bytes(pr.pool_points(2)) == bytes(pk.lag_g2), ("multi-device derived pools differ", n, devs)
                p = pr.prove_rs(w, r, s)
                assert (p.a, p.b, p.c) == exp, ("multi-device Groth16 mismatch after the derivation", n, devs)
            pr.close()
        finally:
            _lib.set_device_list([0])
        n_multi += 1
    if n_g16 % 4 == 3:
        # round 4: the Fr stage's products through the residue number system (csrc/rns_ntt.hip, an option read at upload), both key forms
        os.environ["ZK_FR_RNS"] = "1"
        try:
            for lag in (False, True):
                pr = Groth16(cs, pk, lagrange=lag)
                p = pr.prove_rs(w, r, s)
                assert (p.a, p.b, p.c) == exp, ("Groth16 mismatch through the residue number system", n, lag)
                pr.close()
        finally:
            del os.environ["ZK_FR_RNS"]
        n_rns += 1
    if n_g16 % 4 == 2:
        # round 4: the two-level counting sort forced on at this size (ZK_SORT_TWO_LEVEL_MIN, ZK_MSM_WINDOW: read when the key is built) with its two
        # scatters plain or staged through LDS (ZK_SORT_COARSE_STAGED, ZK_SORT_FINE_STAGED: read per call), proofs in flight on several slots
        os.environ["ZK_SORT_TWO_LEVEL_MIN"], os.environ["ZK_MSM_WINDOW"] = "8", str(rnd.choice([15, 16, 17, 20, 22]))          # 15, 17 divide 255: digits of min(s, r - s), one window fewer
        try:
            pr = Groth16(cs, pk, lagrange=rnd.random() < 0.5)
            pr.set_witness(w)
            for _ in range(2):
                os.environ["ZK_SORT_COARSE_STAGED"], os.environ["ZK_SORT_FINE_STAGED"] = rnd.choice("01"), rnd.choice("01")
                for slot in range(3):
                    pr.prove_async(None, r, s, slot)
                for slot in range(3):
                    p = pr.prove_wait(slot)
                    assert (p.a, p.b, p.c) == exp, ("Groth16 mismatch with the two-level sort", n, dict((k, os.environ[k]) for k in os.environ if k.startswith("ZK_SORT") or k == "ZK_MSM_WINDOW"))
            pr.close()
        finally:
            for k in ("ZK_SORT_TWO_LEVEL_MIN", "ZK_MSM_WINDOW", "ZK_SORT_COARSE_STAGED", "ZK_SORT_FINE_STAGED"):
                os.environ.pop(k, None)
        n_sort += 1
    n_g16 += 1
    # ---- every fourth round: Pinocchio ZK prove at a random size against the trapdoor evaluation, then the product's verifier
    if n_g16 % 4 == 0:
        if rnd.random() < 0.5:
            n = 2 * rnd.randrange(1, 400)
            cs, w = RC.iterated_cubic(n, rnd.randrange(1, P.R))
        else:          # round 5: the general family too (multi-term rows, unused variables = identity key points, special witness values, with / without ONE)
            n = rnd.randrange(1, 300)
            hi = rnd.choice([1, 2, 5, 9])
            cs, w = RC.random_r1cs(n, rnd.randrange(hi + 2, hi + 2 + 2 * n), rnd.randrange(1 << 30), nnz=(1, hi), one=rnd.random() < 0.7)
        tox = [rnd.randrange(1, P.R) for _ in range(11)]
        it = iter(tox)
        pk, vk = PIN.ZK.keygen(lambda: next(it), cs)
        # round 5: the compact h pool and the shared sorts are the default; every third case switches one of them off, every fifth runs the Fr stage through the RNS
        opts = {"ZK_PIN_COMPACT_H": "0" if n_pin % 3 == 1 else None, "ZK_PIN_SHARED_SORT": "0" if n_pin % 3 == 2 else None}
        for k_, v_ in opts.items():
            _lib.check(_lib.lib().zk_set_option(k_.encode(), None if v_ is None else v_.encode()))
        if n_pin % 5 == 4:
            os.environ["ZK_FR_RNS"] = "1"
        pin_devs = [0] * rnd.choice([2, 3, 4]) if n_pin % 4 == 3 else None          # round 5: every fourth case behind a device list (csrc/pinocchio.hip, PinGroup)
        if pin_devs:
            _lib.set_device_list(pin_devs)
        try:
            prover = PIN.ZK(cs, pk)
        finally:
            for k_ in opts:
                _lib.check(_lib.lib().zk_set_option(k_.encode(), None))
        assert prover.pool_size(5) == (cs.n + 1 if opts["ZK_PIN_COMPACT_H"] is None else cs.n + 1 + 2 * cs.m), ("h pool form", n)
        if pin_devs:
            n_pin_multi += 1
            prover.set_witness(w)
            for slot in range(3):                          # rotating owners
                prover.prove_async(*tox[8:], slot)
            for slot in range(3):
                assert prover.prove_wait(slot).to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, *[O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)], cs.mid,
                                                                                        b"".join(frb(x) for x in w), b"".join(frb(x) for x in tox[:8]), *(frb(x) for x in tox[8:])), ("multi-device Pinocchio mismatch", n, pin_devs, slot)
        proof = prover.prove(lambda: next(it), w)
        csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
        exp = O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, b"".join(frb(x) for x in w), b"".join(frb(x) for x in tox[:8]), *(frb(x) for x in tox[8:]))
        assert proof.to_bytes() == exp, ("Pinocchio mismatch", n)
        if n_pin % 2 == 0:
            # round 2: the h pool rewritten on the device ([lambda_t(s)], zk_pinocchio_pk_derive_lagrange) proves the same bytes
            prover.derive_lagrange()
            assert prover.prove_with(w, *tox[8:]).to_bytes() == exp, ("Pinocchio mismatch after the derivation", n)
        if n <= 40:
            assert PIN.ZK.verify([w[k] for k in range(cs.m) if not cs.mid[k]], vk, proof)
        prover.close()
        if pin_devs:
            _lib.set_device_list([0])
        os.environ.pop("ZK_FR_RNS", None)          # set for every fifth case from the upload to the last proof (the switch is read per call under ZK_TEST_FORMS)
        n_pin += 1
    # ---- every eighth round: a list of compressed points (valid, the identity, sometimes one bad one) through the GPU decompression against the host function
    if n_msm % 8 == 0:
        Gd = rnd.choice([G1, G2])
        cnt = rnd.choice([1, 3, 64, 257, 700])
        ptsd = Gd.of_Fr(RC.fr_bytes([rnd.randrange(P.R) if rnd.random() > 0.05 else 0 for _ in range(cnt)]))
        Bd, Cd = Gd.POINT_BYTES, Gd.COMPRESSED_BYTES
        comps = [Gd.to_compressed_bytes(ptsd[Bd * i:Bd * (i + 1)]) for i in range(cnt)]
        bad_at = rnd.randrange(cnt) if rnd.random() < 0.3 else None
        if bad_at is not None:
            comps[bad_at] = rnd.choice([bytes([comps[bad_at][0] & 0x7F]) + comps[bad_at][1:], bytes([0x9F]) + bytes([0xFF]) * (Cd - 1),
                                        bytes([0x80]) + rnd.randrange(1, 1 << 64).to_bytes(Cd - 1, "big")])
        import ctypes as _C
        outd = np.zeros(cnt * Bd, dtype=np.uint8)
        bufd = np.frombuffer(b"".join(comps), dtype=np.uint8)
        rcd = getattr(_lib.lib(), "zk_g1_decompress_batch" if Gd is G1 else "zk_g2_decompress_batch")(bufd.ctypes.data_as(_C.c_void_p), _C.c_size_t(cnt), outd.ctypes.data_as(_C.c_void_p))
        one = getattr(_lib.lib(), "zk_g1_decompress" if Gd is G1 else "zk_g2_decompress")
        refs, rcs = [], []
        for cpt in comps:
            o1 = _C.create_string_buffer(Bd)
            rcs.append(one(bytes(cpt), o1))
            refs.append(o1.raw)
        if all(r == 0 for r in rcs):
            assert rcd == 0 and bytes(outd) == b"".join(refs), ("batched decompression differs from the host function", Gd.__name__, cnt)
        else:
            assert rcd != 0 and rcd in rcs, ("batched decompression accepted a list the host function rejects", Gd.__name__, cnt, rcd, rcs[bad_at])
        n_dec += 1
    if (n_msm % 10) == 0:
        print("soak: %d MSM cases, %d Groth16 cases (%d multi-device, %d RNS), %d Pinocchio cases ok" % (n_msm, n_g16, n_multi, n_rns, n_pin), flush=True)
print("SOAK-OK msm=%d groth16=%d (of them %d with the derived Lagrange form, %d with batch-affine rounds, %d behind a multi-device handle, %d through the residue number system, %d through the forced two-level sort in its plain / staged forms) pinocchio=%d (%d of them behind a device list) decompressed_lists=%d"
      % (n_msm, n_g16, n_der, n_ba, n_multi, n_rns, n_sort, n_pin, n_pin_multi, n_dec))
