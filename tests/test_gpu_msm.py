"""GPU parity: Pippenger MSM (HIP) == the oracle's left fold of scalar multiplications
(curve.ml:91-118), bit-exact on the uncompressed encodings; G.of_Fr / G.powers likewise."""
import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC
from zukelang_amd.curve import G1, G2

pytestmark = pytest.mark.gpu


def frb(x):
    return P.fr_to_bytes(x)


@pytest.mark.parametrize("G,gen,mul", [(G1, O.g1_generator, O.g1_mul), (G2, O.g2_generator, O.g2_mul)])
def test_of_fr_and_powers_match_oracle(G, gen, mul):
    ks = [0, 1, 2, P.R - 1, 0x1234567890ABCDEF, P.R // 3] + [int.from_bytes(bytes(RC.random_fr_bytes(1, 40 + i)), "little") for i in range(10)]
    got = bytes(G.of_Fr(b"".join(frb(k) for k in ks)))
    B = G.POINT_BYTES
    for i, k in enumerate(ks):
        assert got[B * i:B * (i + 1)] == mul(gen(), frb(k)), i
    s = 0x0BADC0FFEE0DDF00D
    pw = bytes(G.powers(9, frb(s)))
    ref = (O.g1_powers if G is G1 else O.g2_powers)(9, frb(s))
    assert pw == ref


@pytest.mark.parametrize("n,c", [(1, 0), (2, 4), (7, 3), (33, 5), (100, 8), (257, 0), (600, 11)])
def test_msm_g1_matches_naive_oracle(n, c):
    ks = RC.random_fr_bytes(n, 1000 + n)
    bases = G1.of_Fr(ks)
    scalars = RC.random_fr_bytes(n, 2000 + n)
    rc, ref = O.g1_msm_naive(bytes(bases), bytes(scalars))
    assert rc == 0
    assert bytes(G1.apply_powers(scalars, bases, c)) == ref


@pytest.mark.parametrize("n,c", [(1, 0), (5, 3), (64, 6), (300, 0)])
def test_msm_g2_matches_naive_oracle(n, c):
    ks = RC.random_fr_bytes(n, 3000 + n)
    bases = G2.of_Fr(ks)
    scalars = RC.random_fr_bytes(n, 4000 + n)
    rc, ref = O.g2_msm_naive(bytes(bases), bytes(scalars))
    assert rc == 0
    assert bytes(G2.apply_powers(scalars, bases, c)) == ref


def test_msm_edge_cases():
    """Adversarial inputs for the bucket sums: duplicates (P+P), negations (P+(-P)), zero and
    boundary scalars, infinity among the bases (SURVEY 7.2 hard part 5)."""
    g = O.g1_generator()
    p = O.g1_mul(g, frb(5))
    inf = bytes([0x40]) + bytes(95)
    bases = p * 6 + inf + g
    scalars = [7, 7, P.R - 7, 0, 1, P.R - 1, 12345, 0x8000]      # same digit -> same bucket: doubling path
    rc, ref = O.g1_msm_naive(bases, b"".join(frb(s) for s in scalars))
    for c in (2, 3, 4, 8, 13, 16):
        assert bytes(G1.apply_powers(b"".join(frb(s) for s in scalars), bases, c)) == ref, c
    # everything cancels -> infinity
    sc = [3, P.R - 3]
    assert bytes(G1.apply_powers(b"".join(frb(s) for s in sc), p * 2, 0)) == inf
    # curve.ml:116 invalid_arg "apply_powers"
    with pytest.raises(ValueError):
        G1.apply_powers(frb(1) * 3, p * 2)
    # curve.ml:115: fewer coefficients than points is fine
    assert bytes(G1.apply_powers(frb(2), p * 2)) == O.g1_mul(p, frb(2))
    assert bytes(G1.apply_powers(b"", b"")) == inf
    # dot over different key sets: curve.ml:96-100 assert false
    with pytest.raises(AssertionError):
        G1.dot({1: p, 2: p}, {1: frb(1), 3: frb(1)})
    assert bytes(G1.dot({1: p, 2: g}, {2: frb(3), 1: frb(4)})) == O.g1_add(O.g1_mul(p, frb(4)), O.g1_mul(g, frb(3)))


def test_msm_rejects_bad_inputs():
    from zukelang_amd._lib import ZkError
    g = bytearray(O.g1_generator())
    g[95] ^= 1
    with pytest.raises(ZkError) as e:
        G1.apply_powers(frb(1), bytes(g))
    assert e.value.code == -2
    with pytest.raises(ZkError) as e:
        G1.apply_powers((P.R).to_bytes(32, "little"), O.g1_generator())
    assert e.value.code == -3


@pytest.mark.parametrize("G,mul,gen,logn", [(G1, O.g1_mul, O.g1_generator, 16), (G2, O.g2_mul, O.g2_generator, 14), (G1, O.g1_mul, O.g1_generator, 20)])
def test_msm_large_trapdoor(G, mul, gen, logn):
    """Full-size exact check: bases k_i*G with known k_i, so MSM = (sum s_i k_i) * G."""
    n = 1 << logn
    ks = RC.random_fr_bytes(n, 77)
    scalars = RC.random_fr_bytes(n, 78)
    bases = G.of_Fr(ks)
    # spot-check the fixed-base kernel against the oracle
    B = G.POINT_BYTES
    for i in (0, 1, n // 2, n - 1):
        assert bytes(bases[B * i:B * (i + 1)]) == mul(gen(), bytes(ks[32 * i:32 * i + 32]))
    expect = mul(gen(), O.fr_dot(bytes(ks), bytes(scalars)))
    assert bytes(G.apply_powers(scalars, bases, 0)) == expect


@pytest.mark.parametrize("kind", ["boolean_heavy", "all_equal", "two_values"])
def test_msm_skewed_bucket_loads(kind):
    """Digit distributions that pile most points into one bucket (boolean-heavy witnesses, SURVEY 7.2
    hard part 4): the chunked accumulate + worklist fix-up must stay exact (and fast)."""
    n = 1 << 15
    ks = RC.random_fr_bytes(n, 501)
    bases = G1.of_Fr(ks)
    rng = np.random.Generator(np.random.PCG64(7))
    if kind == "boolean_heavy":
        vals = np.zeros((n, 32), dtype=np.uint8)
        vals[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)          # 0 / 1
        full = np.frombuffer(bytes(RC.random_fr_bytes(n, 502)), dtype=np.uint8).reshape(n, 32)
        pick = rng.random(n) < 0.1
        vals[pick] = full[pick]
    elif kind == "all_equal":
        vals = np.tile(np.frombuffer(P.fr_to_bytes(0x1234567890ABCDEF1234567890ABCDEF), dtype=np.uint8), (n, 1))
    else:
        a = np.frombuffer(P.fr_to_bytes(P.R - 1), dtype=np.uint8)
        b = np.frombuffer(P.fr_to_bytes(3), dtype=np.uint8)
        vals = np.where(rng.random(n)[:, None] < 0.5, a[None, :], b[None, :]).astype(np.uint8)
    scalars = np.ascontiguousarray(vals).reshape(-1)
    expect = O.g1_mul(O.g1_generator(), O.fr_dot(bytes(ks), bytes(scalars)))
    for c in (0, 8, 15):
        assert bytes(G1.apply_powers(scalars, bases, c)) == expect, (kind, c)


# ---- the resident-key machinery (window tables, one bucket set, BATCH-AFFINE halving rounds) driven through zk_msm_g1/g2
@pytest.fixture
def resident_path(monkeypatch):
    """ZK_MSM_API_PRECOMP=1 sends zk_msm_* through the kernels a proving key uses; ZK_MSM_BA_ROUNDS forces the number of
    batch-affine rounds (auto would pick 0 for base sets this small)."""
    def use(rounds):
        monkeypatch.setenv("ZK_MSM_API_PRECOMP", "1")
        monkeypatch.setenv("ZK_MSM_BA_ROUNDS", str(rounds))
    return use


@pytest.mark.parametrize("rounds", [1, 2, 3, 6])
def test_batch_affine_edge_cases(resident_path, rounds):
    """dx = 0 in every flavour (csrc/msm_ba.cuh): P + P (doubling inside the shared inversion), P + (-P) (identity, then identity
    operands in the next round), duplicated and negated bases under equal digits, identity bases (filtered by the sort), odd runs."""
    resident_path(rounds)
    g = O.g1_generator()
    p5 = O.g1_mul(g, frb(5))
    m5 = O.g1_mul(g, frb(P.R - 5))
    inf = bytes([0x40]) + bytes(95)
    cases = [
        (p5 * 8, [7] * 8),                                   # all equal: doubling chain 2P, 4P, 8P
        (p5 * 7, [7] * 7),                                   # ... with odd leftovers
        ((p5 + m5) * 4, [9] * 8),                            # P, -P under the same digit: everything cancels
        ((p5 + m5) * 3 + p5, [9] * 7),                       # cancels except one
        (p5 * 3 + m5 * 3 + g + inf + p5, [3, 3, P.R - 3, 3, 3, P.R - 3, 11, 5, 3]),      # sign of the digit against sign of the base
        (p5 * 6 + inf + g, [7, 7, P.R - 7, 0, 1, P.R - 1, 12345, 0x8000]),
        (inf * 5 + g, [1, 2, 3, 4, 5, 6]),                   # identity bases never enter a bucket
        (p5 + p5, [1, P.R - 1]),                             # P + (-P) through the recoded digits
    ]
    for bases, scalars in cases:
        sc = b"".join(frb(s) for s in scalars)
        rc, ref = O.g1_msm_naive(bases, sc)
        assert rc == 0
        for c in (3, 4, 8):
            assert bytes(G1.apply_powers(sc, bases, c)) == ref, (rounds, c, scalars)


@pytest.mark.parametrize("G,naive,n,c,rounds", [(G1, O.g1_msm_naive, 600, 4, 5), (G1, O.g1_msm_naive, 257, 3, 9), (G1, O.g1_msm_naive, 1000, 8, 2),
                                                (G2, O.g2_msm_naive, 200, 4, 4), (G2, O.g2_msm_naive, 65, 3, 7)])
def test_batch_affine_matches_naive_oracle(resident_path, G, naive, n, c, rounds):
    resident_path(rounds)
    ks = RC.random_fr_bytes(n, 7000 + n)
    bases = G.of_Fr(ks)
    scalars = RC.random_fr_bytes(n, 8000 + n)
    rc, ref = naive(bytes(bases), bytes(scalars))
    assert rc == 0
    assert bytes(G.apply_powers(scalars, bases, c)) == ref


def test_batch_affine_g2_edge_cases(resident_path):
    resident_path(3)
    g = O.g2_generator()
    q = O.g2_mul(g, frb(9))
    mq = O.g2_mul(g, frb(P.R - 9))
    inf = bytes([0x40]) + bytes(191)
    for bases, scalars in ((q * 8, [5] * 8), ((q + mq) * 4, [5] * 8), (q * 3 + mq * 2 + inf + g, [3, 3, 3, 3, P.R - 3, 9, 2])):
        sc = b"".join(frb(s) for s in scalars)
        rc, ref = O.g2_msm_naive(bases, sc)
        assert rc == 0
        for c in (3, 5):
            assert bytes(G2.apply_powers(sc, bases, c)) == ref, (c, scalars)


@pytest.mark.parametrize("kind", ["boolean_heavy", "all_equal"])
def test_batch_affine_skewed_bucket_loads(resident_path, kind):
    """One bucket swallowing most of the digits: the rounds halve it like any other run and the XYZZ finisher (worklist
    fix-up) takes what the fixed number of rounds leaves."""
    resident_path(4)
    n = 1 << 13
    ks = RC.random_fr_bytes(n, 601)
    bases = G1.of_Fr(ks)
    rng = np.random.Generator(np.random.PCG64(9))
    if kind == "boolean_heavy":
        vals = np.zeros((n, 32), dtype=np.uint8)
        vals[:, 0] = rng.integers(0, 2, size=n, dtype=np.uint8)
        full = np.frombuffer(bytes(RC.random_fr_bytes(n, 602)), dtype=np.uint8).reshape(n, 32)
        pick = rng.random(n) < 0.1
        vals[pick] = full[pick]
    else:
        vals = np.tile(np.frombuffer(P.fr_to_bytes(0x1234567890ABCDEF1234567890ABCDEF), dtype=np.uint8), (n, 1))
    scalars = np.ascontiguousarray(vals).reshape(-1)
    expect = O.g1_mul(O.g1_generator(), O.fr_dot(bytes(ks), bytes(scalars)))
    for c in (8, 13):
        assert bytes(G1.apply_powers(scalars, bases, c)) == expect, (kind, c)
