"""Degenerate variable partitions on both protocols: a circuit with NO intermediate variable (every I_mid pool of the key is empty: the MSMs
run over the appended single points alone) and one where everything but ONE is intermediate -- as uploaded and after the on-device
derivation of the Lagrange-form bases, against the oracle's trapdoor evaluation and the product's own verifier."""
import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import pinocchio as PIN
from zukelang_amd import r1cs as RC
from zukelang_amd.groth16 import Groth16

pytestmark = pytest.mark.gpu


def frs(xs):
    return b"".join(P.fr_to_bytes(x) for x in xs)


@pytest.mark.parametrize("n", [2, 6, 8])
@pytest.mark.parametrize("partition", ["no_mid", "all_but_one_mid"])
def test_degenerate_mid_sets(partition, n):
    cs, w = RC.iterated_cubic(n, 0x99)
    cs.mid = np.zeros(cs.m, dtype=np.uint8) if partition == "no_mid" else np.array([0] + [1] * (cs.m - 1), dtype=np.uint8)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    io = [w[k] for k in range(cs.m) if not cs.mid[k]]
    st = P.fr_stream(77 + n)
    tox = [next(st) for _ in range(5)]
    r, s = next(st), next(st)
    it = iter(tox)
    pk, vk = Groth16.keygen(lambda: next(it), cs)
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox), P.fr_to_bytes(r), P.fr_to_bytes(s))
    pr = Groth16(cs, pk)
    p = pr.prove_rs(w, r, s)
    assert (p.a, p.b, p.c) == exp
    pr.derive_lagrange()
    p = pr.prove_rs(w, r, s)
    assert (p.a, p.b, p.c) == exp
    assert Groth16.verify(io, vk, p)
    pr.close()
    tox = [next(st) for _ in range(11)]
    it = iter(tox)
    pk, vk = PIN.ZK.keygen(lambda: next(it), cs)
    pp = PIN.ZK(cs, pk)
    e2 = O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox[:8]), *(P.fr_to_bytes(x) for x in tox[8:]))
    proof = pp.prove_with(w, *tox[8:])
    assert proof.to_bytes() == e2
    assert PIN.ZK.verify(io, vk, proof)
    pp.derive_lagrange()
    assert pp.prove_with(w, *tox[8:]).to_bytes() == e2
    pp.close()


def test_single_gate_circuit():
    """n = 1: out = x * x.  v, w, y are constants, Z = X, h has no coefficient at all (QAP.ml:132-135 divides the zero polynomial); the
    key's tiztd list is empty and its Lagrange form is the power form's first point."""
    ONE, OUT, X = range(3)
    Mx = RC.Matrix.from_rows
    cs = RC.R1CS(1, 3, Mx([{X: 1}]), Mx([{X: 1}]), Mx([{OUT: 1}]), np.array([0, 0, 1], dtype=np.uint8))
    x = 0x1234567
    w = [1, x * x % P.R, x]
    assert cs.check(w)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    io = [w[ONE], w[OUT]]
    st = P.fr_stream(4242)
    tox = [next(st) for _ in range(5)]
    r, s = next(st), next(st)
    it = iter(tox)
    pk, vk = Groth16.keygen(lambda: next(it), cs, lagrange=True)
    assert len(pk.g1) == 96 * (3 + 3 + 0 + 1) and len(pk.g2) == 192 * (2 + 3)
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox), P.fr_to_bytes(r), P.fr_to_bytes(s))
    pr = Groth16(cs, pk)
    p = pr.prove_rs(w, r, s)
    assert (p.a, p.b, p.c) == exp
    assert Groth16.verify(io, vk, p)
    v, ww, h = pr.qap_eval(w)
    assert RC.fr_ints(v) == [x] and RC.fr_ints(ww) == [x] and len(h) == 0
    with pytest.raises(AssertionError):
        pr.prove_rs([1, (x * x + 1) % P.R, x], r, s)          # QAP.ml:134
    pr.derive_lagrange()
    assert bytes(pr.pool_points(1)) == bytes(pk.lag_g1) and bytes(pr.pool_points(2)) == bytes(pk.lag_g2)
    p = pr.prove_rs(w, r, s)
    assert (p.a, p.b, p.c) == exp
    pr.close()
    pl = Groth16(cs, pk, lagrange=True)
    p = pl.prove_rs(w, r, s)
    assert (p.a, p.b, p.c) == exp
    pl.close()
    tox = [next(st) for _ in range(11)]
    it = iter(tox)
    pk, vk = PIN.ZK.keygen(lambda: next(it), cs)
    pp = PIN.ZK(cs, pk)
    e2 = O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox[:8]), *(P.fr_to_bytes(t) for t in tox[8:]))
    proof = pp.prove_with(w, *tox[8:])
    assert proof.to_bytes() == e2
    assert PIN.ZK.verify(io, vk, proof)
    pp.derive_lagrange()
    assert pp.prove_with(w, *tox[8:]).to_bytes() == e2
    pp.close()
