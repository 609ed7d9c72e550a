"""GPU parity: Pinocchio Protocol 2 prove (HIP) == the oracle's literal restatement of
pinocchio.ml:427-514 (bit-exact), plus Verify.f (:254-420) with the oracle pairing."""
import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC
from zukelang_amd import pinocchio as PIN
from zukelang_amd.curve import G1, G2

pytestmark = pytest.mark.gpu


def frs(xs):
    return b"".join(P.fr_to_bytes(x) for x in xs)


def csrs(cs):
    return [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]


@pytest.mark.parametrize("maker,literal", [(lambda: RC.readme_circuit(3), True), (lambda: RC.iterated_cubic(6, 9), True),
                                           (lambda: RC.iterated_cubic(64, 10), False), (lambda: RC.iterated_cubic(1000, 11), False)])
def test_pinocchio_zk_and_nonzk_match_oracle(maker, literal):
    cs, w = maker()
    L, R_, Oo = csrs(cs)
    st = P.fr_stream(0x5EED0003)
    tox = [next(st) for _ in range(11)]
    toxic = frs(tox[:8])
    it = iter(tox)
    rng = lambda: next(it)
    pk, vk = PIN.ZK.keygen(rng, cs)                               # draws the 8 key scalars
    ex = O.pinocchio_keygen_exponents(None, cs.n, cs.m, L, R_, Oo, cs.mid, toxic, False)
    if cs.n <= 64:
        assert bytes(pk.g1) == O.points_of_exponents_g1(ex[0]) and bytes(pk.g2) == O.points_of_exponents_g2(ex[1])
        assert bytes(vk.g1) == O.points_of_exponents_g1(ex[2]) and bytes(vk.g2) == O.points_of_exponents_g2(ex[3])
    prover = PIN.ZK(cs, pk)
    proof = prover.prove(rng, w)                                  # draws dv, dw, dy = tox[8:11]
    dv, dw, dy = (P.fr_to_bytes(x) for x in tox[8:])
    assert proof.to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, dv, dw, dy)
    if literal:
        q = O.QAP(cs.n, cs.m, L, R_, Oo)
        rc, ref = O.pinocchio_prove(q, bytes(pk.g1), bytes(pk.g2), cs.mid, frs(w), dv, dw, dy)
        assert rc == 0 and proof.to_bytes() == ref
        io = [w[k] for k in range(cs.m) if not cs.mid[k]]
        assert O.pinocchio_verify(bytes(vk.g1), bytes(vk.g2), io, proof.to_bytes())
        assert PIN.ZK.verify(io, vk, proof)                       # the product's own verifier (host pairing)
        assert not PIN.ZK.verify([(x + 1) % RC.FR_MODULUS for x in io], vk, proof)
    # pipelined form: resident witness, three proofs in flight with different blinding, collected in order
    prover.set_witness(w)
    ds = [[next(st) for _ in range(3)] for _ in range(3)]
    for slot, d in enumerate(ds):
        prover.prove_async(*d, slot)
    for slot, d in enumerate(ds):
        got = prover.prove_wait(slot)
        assert got.to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(x) for x in d))
    prover.close()
    nz = PIN.NonZK(cs, pk)
    p0 = nz.prove(None, w)
    z = P.fr_to_bytes(0)
    assert p0.to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, z, z, z)
    w_bad = list(w)
    w_bad[2] = (w_bad[2] + 1) % RC.FR_MODULUS
    with pytest.raises(AssertionError):
        nz.prove(None, w_bad)
    nz.close()


def test_pinocchio_2_18_trapdoor_and_verify():
    """BASELINE config 5: Pinocchio Protocol-2 prove, 2^18 constraints."""
    n = 1 << 18
    cs, w = RC.iterated_cubic(n, next(P.fr_stream(0x5EED0001)))
    L, R_, Oo = csrs(cs)
    st = P.fr_stream(0x5EED0003)
    tox = [next(st) for _ in range(11)]
    toxic = frs(tox[:8])
    e1, e2, v1, v2 = O.pinocchio_keygen_exponents(None, cs.n, cs.m, L, R_, Oo, cs.mid, toxic, False)
    pk = PIN.PKey(G1.of_Fr(e1), G2.of_Fr(e2))
    for i in (0, n, len(e1) // 32 - 1):
        assert bytes(pk.g1[96 * i:96 * i + 96]) == O.g1_mul(O.g1_generator(), e1[32 * i:32 * i + 32])
    prover = PIN.ZK(cs, pk)
    dv, dw, dy = tox[8:]
    proof = prover.prove_with(w, dv, dw, dy)
    assert proof.to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(x) for x in (dv, dw, dy)))
    io = [w[k] for k in range(cs.m) if not cs.mid[k]]
    assert O.pinocchio_verify(bytes(G1.of_Fr(v1)), bytes(G2.of_Fr(v2)), io, proof.to_bytes())
    assert len(proof.to_compressed()) == 480
    # ... and the path bench.py times at this size: the h bases derived on the device (zk_pinocchio_pk_derive_lagrange), proofs one
    # at a time and pipelined, against the trapdoor oracle of pinocchio.ml:427-514
    prover.derive_lagrange()
    ds = [[next(st) for _ in range(3)] for _ in range(3)]
    exp = [O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(x) for x in d)) for d in ds]
    assert prover.prove_with(w, dv, dw, dy).to_bytes() == proof.to_bytes()
    assert prover.prove_with(w, *ds[0]).to_bytes() == exp[0]
    prover.set_witness(w)
    for slot, d in enumerate(ds):
        prover.prove_async(*d, slot)
    for slot in range(len(ds)):
        assert prover.prove_wait(slot).to_bytes() == exp[slot], "slot %d" % slot
    prover.close()


@pytest.mark.parametrize("maker", [lambda: RC.readme_circuit(3), lambda: RC.iterated_cubic(2, 4), lambda: RC.iterated_cubic(6, 9), lambda: RC.iterated_cubic(64, 10),
                                   lambda: RC.iterated_cubic(1000, 11), lambda: RC.iterated_cubic(1 << 14, 12)])
def test_pinocchio_derived_h_bases_give_the_same_proofs(maker):
    """zk_pinocchio_pk_derive_lagrange: the powers si of an uploaded evaluation key (pinocchio.ml:37-60) become the Lagrange basis of the points
    n .. 2n-2 in the exponent, h enters through its values (no basis conversion per proof).  ZK and NonZK proofs must keep their bytes: against
    the same prover before the derivation and against the trapdoor oracle; an unsatisfied witness still raises."""
    cs, w = maker()
    L, R_, Oo = csrs(cs)
    st = P.fr_stream(0x5EED0D03)
    tox = [next(st) for _ in range(8)]
    toxic = frs(tox)
    it = iter(tox)
    pk, vk = PIN.ZK.keygen(lambda: next(it), cs)
    prover = PIN.ZK(cs, pk)
    ds = [[next(st) for _ in range(3)] for _ in range(3)]
    before = [prover.prove_with(w, *d).to_bytes() for d in ds]
    prover.derive_lagrange()
    prover.derive_lagrange()                                   # idempotent
    for d, b in zip(ds, before):
        got = prover.prove_with(w, *d).to_bytes()
        assert got == b == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(x) for x in d))
    assert prover.prove_with(w, 0, 0, 0).to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(0),) * 3)
    prover.set_witness(w)
    for slot, d in enumerate(ds):
        prover.prove_async(*d, slot)
    for slot, b in enumerate(before):
        assert prover.prove_wait(slot).to_bytes() == b
    w_bad = list(w)
    w_bad[2] = (w_bad[2] + 1) % RC.FR_MODULUS
    with pytest.raises(AssertionError):
        prover.prove_with(w_bad, *ds[0])
    prover.close()


def _set_option(name, value):
    from zukelang_amd import _lib
    _lib.check(_lib.lib().zk_set_option(name.encode(), None if value is None else str(value).encode()))


@pytest.mark.parametrize("maker", [lambda: RC.readme_circuit(3), lambda: RC.random_r1cs(1, 6, 76, nnz=(1, 2)), lambda: RC.random_r1cs(2, 5, 75, nnz=(1, 2)), lambda: RC.iterated_cubic(6, 9), lambda: RC.random_r1cs(24, 40, 77),
                                   lambda: RC.random_r1cs(48, 30, 78, one=False), lambda: RC.iterated_cubic(300, 14)])
def test_compact_h_pool_its_switch_and_its_fallback(maker):
    """Round 5 (csrc/pinocchio.hip): the h product sum_k (dw c_k) [v_k(s)] + sum_k (dv c_k) [w_k(s)] over v_all | w_all (pinocchio.ml:481-486) is
    dw [v(s)] + dv [w(s)] and rides on the powers si (n + 1 points instead of n + 1 + 2 m; derived key: n + 2) -- for keys whose v_all / w_all
    pass the upload's consistency check against si.  Three keys: the default, the switch off (ZK_PIN_COMPACT_H=0: the full pool), and a key with
    ONE v_all point replaced by another subgroup point (the check fails, the full pool stays, and the proof is what ZKCompute.f computes FROM
    THAT KEY point by point: the literal oracle on the tampered bytes).  Every proof, tau-power and derived form, against the oracles."""
    cs, w = maker()
    n, m = cs.n, cs.m
    L, R_, Oo = csrs(cs)
    st = P.fr_stream(0x5EED0C0B)
    tox = [next(st) for _ in range(8)]
    toxic = frs(tox)
    it = iter(tox)
    pk, _vk = PIN.ZK.keygen(lambda: next(it), cs)
    ds = [[next(st) for _ in range(3)] for _ in range(2)] + [[0, 0, 0]]
    exp = [O.pinocchio_prove_trapdoor(n, m, L, R_, Oo, cs.mid, frs(w), toxic, *(P.fr_to_bytes(x) for x in d)) for d in ds]
    q = O.QAP(n, m, L, R_, Oo)
    if n <= 64:
        for d, e in zip(ds, exp):
            rc, ref = O.pinocchio_prove(q, bytes(pk.g1), bytes(pk.g2), cs.mid, frs(w), *(P.fr_to_bytes(x) for x in d))
            assert rc == 0 and ref == e
    nm = sum(1 for k in range(m) if cs.mid[k])
    si0 = 5 * nm                                                       # first point of si in the flattened G1 key
    full = bytes(pk.g1[96 * si0:96 * (si0 + n + 1 + 2 * m)])

    def run(prover, want, tag):
        got = [prover.prove_with(w, *d).to_bytes() for d in ds]
        assert got == want, tag
        prover.derive_lagrange()
        got = [prover.prove_with(w, *d).to_bytes() for d in ds]
        assert got == want, tag + ", derived"
        prover.set_witness(w)
        for slot, d in enumerate(ds):
            prover.prove_async(*d, slot)
        for slot in range(len(ds)):
            assert prover.prove_wait(slot).to_bytes() == want[slot], tag + ", pipelined"

    # the default: compact
    prover = PIN.ZK(cs, pk)
    assert bytes(prover.pool_points(5)) == full[:96 * (n + 1)]
    run(prover, exp, "compact")
    pool = bytes(prover.pool_points(5))
    assert len(pool) == 96 * (n + 2) and pool[96 * n:96 * (n + 1)] == full[:96] and pool[96 * (n + 1):] == full[96 * (n - 1):96 * n]      # ... | [1] | [s^(n-1)]
    prover.close()
    # the switch
    _set_option("ZK_PIN_COMPACT_H", 0)
    try:
        prover = PIN.ZK(cs, pk)
    finally:
        _set_option("ZK_PIN_COMPACT_H", None)
    assert bytes(prover.pool_points(5)) == full
    run(prover, exp, "full pool")
    assert len(prover.pool_points(5)) == 96 * (n + 1 + 2 * m)
    prover.close()
    # a key whose v_all is not the image of its si: variable k's point replaced by [7] G1 (on the curve, in the subgroup -- only the relation to si is broken)
    k = next(k for k in range(m) if w[k] % RC.FR_MODULUS != 0)
    bad = np.array(pk.g1, copy=True)
    off = 96 * (si0 + n + 1 + k)
    bad[off:off + 96] = np.frombuffer(O.g1_mul(O.g1_generator(), P.fr_to_bytes(7)), dtype=np.uint8)
    bad_key = PIN.PKey(bad, pk.g2)
    prover = PIN.ZK(cs, bad_key)
    assert bytes(prover.pool_points(5)) == bytes(bad[96 * si0:96 * (si0 + n + 1 + 2 * m)]), "an inconsistent key keeps its full h pool"
    if n <= 64:
        want = []
        for d in ds:
            rc, ref = O.pinocchio_prove(q, bytes(bad), bytes(pk.g2), cs.mid, frs(w), *(P.fr_to_bytes(x) for x in d))
            assert rc == 0
            want.append(ref)
        assert want[0] != exp[0] and want[2] == exp[2]                 # the blinding terms see the replaced point, the NonZK proof does not
        run(prover, want, "inconsistent key")
    prover.close()
    # One sort per distinct scalar vector (vv / vav, yy / yay, ww / waw share theirs): switched off -- same proofs; and a key in which vav holds the identity
    # where vv holds a point (never the case in a key KeyGen.generate made): the identity flags differ, the pair must NOT share, the proof follows the key
    _set_option("ZK_PIN_SHARED_SORT", 0)
    try:
        prover = PIN.ZK(cs, pk)
    finally:
        _set_option("ZK_PIN_SHARED_SORT", None)
    run(prover, exp, "own sorts")
    prover.close()
    mids = [v for v in range(m) if cs.mid[v]]
    ident = bytes([0x40]) + bytes(95)
    j = next((j for j, v in enumerate(mids) if w[v] % RC.FR_MODULUS != 0 and bytes(pk.g1[96 * j:96 * j + 96]) != ident), None)
    if j is not None and n <= 64:
        bad = np.array(pk.g1, copy=True)
        bad[96 * (2 * nm + j):96 * (2 * nm + j + 1)] = np.frombuffer(ident, dtype=np.uint8)          # vav[j] := O
        prover = PIN.ZK(cs, PIN.PKey(bad, pk.g2))
        want = []
        for d in ds:
            rc, ref = O.pinocchio_prove(q, bytes(bad), bytes(pk.g2), cs.mid, frs(w), *(P.fr_to_bytes(x) for x in d))
            assert rc == 0
            want.append(ref)
        assert want[0] != exp[0]
        run(prover, want, "vav with an identity of its own")
        prover.close()
