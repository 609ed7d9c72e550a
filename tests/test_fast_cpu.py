"""The multi-threaded CPU context prover (oracle/fast_cpu.c: Pippenger + NTT over the Lagrange-form key) against the oracle's
other two forms: the naive MSM fold and the trapdoor evaluation of groth16.ml:123-161.  Test infrastructure checking test
infrastructure: bench.py prints this path's rate beside the GPU's, so it has to produce the same bytes."""
import random

import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC
from zukelang_amd.groth16 import _lagrange_at

rnd = random.Random(77001)
frb = P.fr_to_bytes


def frs(xs):
    return b"".join(frb(x) for x in xs)


@pytest.mark.parametrize("threads", [1, 3])
def test_fast_msm_equals_naive_fold(threads):
    for G, n in ((1, 1), (1, 37), (1, 300), (2, 1), (2, 41)):
        gen, mul, naive, fast, B = ((O.g1_generator, O.g1_mul, O.g1_msm_naive, O.fast_g1_msm, 96) if G == 1 else
                                    (O.g2_generator, O.g2_mul, O.g2_msm_naive, O.fast_g2_msm, 192))
        uniq = [mul(gen(), frb(rnd.randrange(1, P.R))) for _ in range(5)]
        inf = bytes([0x40]) + bytes(B - 1)
        pts, scs = [], []
        for i in range(n):
            # duplicates, negations via r - k, the identity, zero / one / r - 1 / short / full-width scalars
            pts.append(rnd.choice(uniq + [inf]) if rnd.random() < 0.8 else mul(gen(), frb(rnd.randrange(1, 1 << 40))))
            kind = rnd.random()
            scs.append(0 if kind < 0.1 else 1 if kind < 0.2 else P.R - 1 if kind < 0.3 else rnd.randrange(1 << 16) if kind < 0.45 else
                       (P.R - rnd.randrange(1, 1 << 20)) if kind < 0.6 else rnd.randrange(P.R))
        bases, scalars = b"".join(pts), frs(scs)
        rc, ref = naive(bases, scalars)
        rc2, got = fast(bases, scalars, threads)
        assert rc == 0 and rc2 == 0 and got == ref, (G, n, threads)


def _lagrange_key(cs, tox):
    """The Lagrange-form pools of Groth16.keygen(..., lagrange=True) (groth16.py) with the points from the oracle's scalar multiplication."""
    a, b, gm, d, t = tox
    n, m, Pm = cs.n, cs.m, P.R
    lag, zt = _lagrange_at(n, t)
    dinv = pow(d, Pm - 2, Pm)
    Lk = [0] * m
    for M, mult in ((cs.L, b), (cs.R, a), (cs.O, 1)):
        vals = bytes(M.val)
        for g in range(n):
            for e in range(M.ptr[g], M.ptr[g + 1]):
                Lk[M.col[e]] = (Lk[M.col[e]] + int.from_bytes(vals[32 * e:32 * e + 32], "little") * lag[g] % Pm * mult) % Pm
    lam, _ = _lagrange_at(n - 1, (t - n) % Pm)
    ztd = zt * dinv % Pm
    lx1 = [a, d, b] + lag + [lam[i] * ztd % Pm for i in range(n - 1)] + [Lk[k] * dinv % Pm for k in range(m) if cs.mid[k]]
    lx2 = [b, d] + lag
    g1, g2 = O.g1_generator(), O.g2_generator()
    return b"".join(O.g1_mul(g1, frb(x)) for x in lx1), b"".join(O.g2_mul(g2, frb(x)) for x in lx2)


@pytest.mark.parametrize("maker,threads", [(lambda: RC.readme_circuit(3), 1), (lambda: RC.iterated_cubic(2, 5), 2),
                                           (lambda: RC.iterated_cubic(10, 0x1234567), 1), (lambda: RC.iterated_cubic(48, 99), 4)])
def test_fast_prover_equals_trapdoor_evaluation(maker, threads):
    cs, w = maker()
    st = P.fr_stream(0x5EED0077)
    tox = [next(st) for _ in range(5)]
    r, s = next(st), next(st)
    L, R_, Oo = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    lg1, lg2 = _lagrange_key(cs, tox)
    fp = O.FastGroth16(cs.n, cs.m, L, R_, Oo, cs.mid, lg1, lg2, threads)
    rc, a, b, c = fp.prove(frs(w), frb(r), frb(s))
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), frs(tox), frb(r), frb(s))
    assert rc == 0 and (a, b, c) == exp
    # a witness that does not satisfy the circuit: QAP.ml:134
    bad = list(w)
    bad[-1] = (bad[-1] + 1) % P.R
    if not cs.check(bad):
        assert fp.prove(frs(bad), frb(r), frb(s))[0] == 1
    fp.close()
