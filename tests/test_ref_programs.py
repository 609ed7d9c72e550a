"""The reference's own acceptance programs (src/lib/test/test.ml:194-276: 13 DSL programs pushed through compile -> QAP -> keygen ->
prove -> verify by `dune runtest`) under parity.  tests/golden/ref_programs.json holds their R1CS rows (oracle/comp_ref.py restates
Comp.compile), witnesses for both branches of every `if` / `==` / `case`, and proofs computed from first principles.

CPU: the C oracle's LITERAL restatement (dense QAP, schoolbook polynomials, per-variable apply_powers folds) == its trapdoor form ==
the golden bytes, for Groth16 and both Pinocchio variants; the proofs verify through the Python pairing.
GPU (-m gpu): the HIP prover on the same circuits -- tau-power key and derived Lagrange-form key, Groth16 and Pinocchio -- gives the
same bytes; verify accepts them and rejects a wrong public input.

What these circuits exercise that the benchmark family does not: no $ONE and an empty inputs_public (square_no_one, complex_pair,
uint32_add), n = 1 (no tiztd point, h = 0), gates with an EMPTY lhs (the `==` gadget), multi-term l rows and lhs rows, an EXPLICIT zero
coefficient (either: `c = x * (0 x)`), two gates with identical l and r (complex_pair), several outputs, io sets of size 1..3."""
import numpy as np
import pytest

import oracle_lib as O
import ref_programs as RP
from oracle import comp_ref as CR
from oracle import pyref as P
from zukelang_amd import r1cs as RC

frb = P.fr_to_bytes
frs = lambda xs: b"".join(frb(x) for x in xs)
csrs = lambda cs: [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
NAMES = RP.names()


def test_fixture_is_what_the_generator_writes_for_the_readme_circuit():
    """The first program is the README circuit; SURVEY.md 8c derived its compilation by hand and zukelang_amd/r1cs.py carries that."""
    p = RP.program("cubic")
    cs = RP.circuit(p)
    ref, w3 = RC.readme_circuit(3)
    for a, b in ((cs.L, ref.L), (cs.R, ref.R), (cs.O, ref.O)):
        assert list(a.ptr) == list(b.ptr) and list(a.col) == list(b.col) and bytes(a.val) == bytes(b.val)
    assert list(cs.mid) == list(ref.mid) and RP.witnesses(p)[0] == w3
    assert p["vars"] == [["ONE", 1], ["c", 4], ["c", 5], ["input", 3], ["v", 6]]


def test_compiler_restatement_small_cases():
    """oracle/comp_ref.py against facts read off comp.ml: constant folding makes no gate (:233-239), `x + !1` alone needs the
    fix_output gate (:467-473), Gate.Set drops a duplicate gate, Affine.add keeps a zero coefficient (circuit.ml:39)."""
    vg = CR.VarGen(1); L = CR.Lang(vg)
    c = CR.Comp(vg)
    assert c.compile([], L.mul(L.const(3), L.const(4))) == [{CR.ONE: 12}] and c.gates == []
    with pytest.raises(AssertionError):                               # a constant output is `assert false` in the reference too (comp.ml:513)
        CR.compile_program(L.mul(L.const(3), L.const(4)), CR.VarGen(1))
    vg = CR.VarGen(1); L = CR.Lang(vg)
    c = CR.compile_program(L.let_(L.input("input", "secret", CR.FIELD), lambda x: L.mul(L.mul(x, L.const(2)), L.const(5))), vg)
    assert len(c.gates) == 1 and c.gates[0][1] == {("input", 3): 10} and c.gates[0][2] == {CR.ONE: 1}          # weighted variable -> `v = (10 x) * 1`
    assert CR.aff_add({("a", 2): 1}, {("a", 2): CR.R - 1}) == {("a", 2): 0}
    assert CR.aff_is_const({("a", 2): 0}) is None and CR.aff_is_const({}) == 0 and CR.aff_is_const({CR.ONE: 7}) == 7


@pytest.mark.parametrize("name", NAMES)
def test_oracle_literal_equals_trapdoor_equals_golden(name):
    p = RP.program(name)
    cs = RP.circuit(p)
    csr = csrs(cs)
    n, m = cs.n, cs.m
    q = O.QAP(n, m, *csr)
    tox, r, s = RP.groth16_params(p)
    pk1, pk2, vk1, vk2 = q.groth16_setup(frs(tox), cs.mid)
    ptox, (dv, dw, dy) = RP.pinocchio_params(p)
    lit = O.pinocchio_keygen_exponents(q, n, m, *csr, cs.mid, frs(ptox), True)
    assert lit == O.pinocchio_keygen_exponents(None, n, m, *csr, cs.mid, frs(ptox), False)
    ppk1, ppk2 = O.points_of_exponents_g1(lit[0]), O.points_of_exponents_g2(lit[1])
    pvk1, pvk2 = O.points_of_exponents_g1(lit[2]), O.points_of_exponents_g2(lit[3])
    for i, (w, fix) in enumerate(zip(RP.witnesses(p), p["witnesses"])):
        assert cs.check(w)
        sol = frs(w)
        gold = RP.groth16_proof(fix)
        rc, a, b, c = q.groth16_prove(pk1, pk2, cs.mid, sol, frb(r), frb(s), 1)               # literal groth16.ml:116-161 on QAP.eval's h
        assert rc == 0 and (a, b, c) == gold, (name, i)
        assert O.groth16_prove_trapdoor(n, m, *csr, cs.mid, sol, frs(tox), frb(r), frb(s)) == gold
        for zk, d in ((True, (dv, dw, dy)), (False, (0, 0, 0))):
            pg = RP.pinocchio_proof(fix, zk)
            rc, pr = O.pinocchio_prove(q, ppk1, ppk2, cs.mid, sol, *(frb(x) for x in d))     # literal ZKCompute.f / Compute.f
            assert rc == 0 and pr == pg, (name, i, zk)
            assert O.pinocchio_prove_trapdoor(n, m, *csr, cs.mid, sol, frs(ptox), *(frb(x) for x in d)) == pg
        if i == 0:                                                                            # test.ml:178: `assert (Protocol.verify public vkey proof)`
            io = [w[k] for k in range(m) if not cs.mid[k]]
            A, B, Cc = P.g1_from_bytes(gold[0]), P.g2_from_bytes(gold[1]), P.g1_from_bytes(gold[2])
            acc = None
            for j, x in enumerate(io):
                acc = P.pt_add(acc, P.pt_mul(P.g1_from_bytes(vk1[96 * (1 + j):96 * (2 + j)]), x))
            pairs = [(A, B), (P.pt_neg(P.g1_from_bytes(pk1[:96])), P.g2_from_bytes(pk2[:192])), (P.pt_neg(Cc), P.g2_from_bytes(vk2[384:576]))]
            if acc is not None:
                pairs.append((P.pt_neg(acc), P.g2_from_bytes(vk2[192:384])))
            assert P.pairing_product_is_one(pairs)
            assert O.pinocchio_verify(pvk1, pvk2, io, RP.pinocchio_proof(fix, True))
    # an unsatisfied witness: QAP.ml:134
    w = list(RP.witnesses(p)[0])
    w[m - 1] = (w[m - 1] + 1) % P.R
    if not cs.check(w):
        assert q.groth16_prove(pk1, pk2, cs.mid, frs(w), frb(r), frb(s), 1)[0] != 0


# ------------------------------------------------------------------------------------------------------------------ GPU
@pytest.mark.gpu
@pytest.mark.parametrize("name", NAMES)
def test_gpu_proves_the_reference_programs(name):
    from zukelang_amd.groth16 import Groth16
    from zukelang_amd import pinocchio as PIN
    p = RP.program(name)
    cs = RP.circuit(p)
    m = cs.m
    tox, r, s = RP.groth16_params(p)
    it = iter(tox)
    pk, vk = Groth16.keygen(lambda: next(it), cs)
    q = O.QAP(cs.n, m, *csrs(cs))
    pk1, pk2, vk1, _vk2 = q.groth16_setup(frs(tox), cs.mid)
    assert bytes(pk.g1) == pk1 and bytes(pk.g2) == pk2 and bytes(vk.ltgm_io) == vk1[96:]       # GPU keygen == literal setup (identity points included)
    pr = Groth16(cs, pk)
    ws = RP.witnesses(p)
    for derived in (False, True):
        if derived:
            pr.derive_lagrange()
        for w, fix in zip(ws, p["witnesses"]):
            got = pr.prove_rs(w, r, s)
            assert (got.a, got.b, got.c) == RP.groth16_proof(fix), (name, derived)
            io = [w[k] for k in range(m) if not cs.mid[k]]
            assert Groth16.verify(io, vk, got)
            if io:
                assert not Groth16.verify([(io[0] + 1) % RC.FR_MODULUS] + io[1:], vk, got)
        if cs.n > 1:
            v, ww, h = pr.qap_eval(ws[0])
            ov, ow, _oy = q.eval_vwy(frs(ws[0]))
            rc, _p, oh = q.eval(frs(ws[0]))
            assert rc == 0 and bytes(v) == ov and bytes(ww) == ow and bytes(h).rstrip(b"\0") == oh.rstrip(b"\0")
    bad = list(ws[0]); bad[m - 1] = (bad[m - 1] + 1) % RC.FR_MODULUS
    if not cs.check(bad):
        with pytest.raises(AssertionError):
            pr.prove_rs(bad, r, s)
    pr.close()
    # Pinocchio Protocol 2, ZK and NonZK, as uploaded and with the h bases derived on the device
    ptox, (dv, dw, dy) = RP.pinocchio_params(p)
    it = iter(ptox)
    ppk, pvk = PIN.ZK.keygen(lambda: next(it), cs)
    ex = O.pinocchio_keygen_exponents(None, cs.n, m, *csrs(cs), cs.mid, frs(ptox), False)
    assert bytes(ppk.g1) == O.points_of_exponents_g1(ex[0]) and bytes(ppk.g2) == O.points_of_exponents_g2(ex[1])
    pp = PIN.ZK(cs, ppk)
    for derived in (False, True):
        if derived:
            pp.derive_lagrange()
        for w, fix in zip(ws, p["witnesses"]):
            got = pp.prove_with(w, dv, dw, dy)
            assert got.to_bytes() == RP.pinocchio_proof(fix, True), (name, derived)
            assert pp.prove_with(w, 0, 0, 0).to_bytes() == RP.pinocchio_proof(fix, False)
            io = [w[k] for k in range(m) if not cs.mid[k]]
            assert PIN.ZK.verify(io, pvk, got)
            if io:
                assert not PIN.ZK.verify([(io[0] + 1) % RC.FR_MODULUS] + io[1:], pvk, got)
    pp.close()
