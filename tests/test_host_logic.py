"""CPU-only tests of the host-side mirror: circuit formats, key layout, error behaviour."""
import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import r1cs as RC
from zukelang_amd.groth16 import _lagrange_at, shard_bounds


def test_fr_bytes_roundtrip():
    xs = [0, 1, P.R - 1, 2 ** 200 + 17]
    assert RC.fr_ints(RC.fr_bytes(xs)) == xs
    assert RC.fr_ints(RC.fr_bytes([P.R + 5, -1])) == [5, P.R - 1]


def test_readme_circuit_shape_and_witness():
    cs, w = RC.readme_circuit(3)
    assert (cs.n, cs.m, cs.n_mid) == (3, 5, 3)
    assert w == [1, 9, 27, 3, 33] and cs.check(w)
    w[4] = 34
    assert not cs.check(w)


@pytest.mark.parametrize("n", [2, 8, 64, 1000])
def test_iterated_cubic_is_satisfied_and_sized_as_survey_says(n):
    cs, w = RC.iterated_cubic(n, 12345)
    assert cs.m == n + 2 and cs.n_mid == n
    assert cs.L.ptr[-1] == n and cs.R.ptr[-1] == n and cs.O.ptr[-1] == n + 2 * (n // 2)
    if n <= 64:
        assert cs.check(w)
    # spmv through the oracle: a*b == c on every gate
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    sol = bytes(RC.fr_bytes(w))
    a, b, c = (RC.fr_ints(O.r1cs_spmv(n, M, sol)) for M in csr)
    assert all(x * y % P.R == z for x, y, z in zip(a, b, c))


def test_lagrange_at_matches_interpolation():
    n, tau = 7, 0x1234567
    lag, zt = _lagrange_at(n, tau)
    assert zt == P.poly_eval(P.z_poly(n), tau)
    for i in range(n):
        basis = P.interpolate_int_domain([1 if j == i else 0 for j in range(n)])
        assert lag[i] == P.poly_eval(basis, tau)


def test_shard_bounds_partition_the_pools():
    for size in (13, 65545, 196612):
        for world in (1, 2, 3, 8):
            cuts = [shard_bounds(size, r, world) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == size
            assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
            assert max(hi - lo for lo, hi in cuts) - min(hi - lo for lo, hi in cuts) <= 1
            # the G1 rule: the first `heavy` points count twice (products A and C), the cuts balance points + heavy prefix
            for heavy in (0, 5, size // 4, size // 2, size):
                cuts = [shard_bounds(size, r, world, heavy) for r in range(world)]
                assert cuts[0][0] == 0 and cuts[-1][1] == size
                assert all(cuts[i][1] == cuts[i + 1][0] for i in range(world - 1))
                work = [(hi - lo) + max(0, min(hi, heavy) - min(lo, heavy)) for lo, hi in cuts]
                assert sum(work) == size + heavy and max(work) - min(work) <= 3          # one heavy point = two units


def test_splitmix_stream_is_deterministic():
    a = RC.fr_stream(0x5EED0002)
    b = P.fr_stream(0x5EED0002)
    assert [next(a) for _ in range(5)] == [next(b) for _ in range(5)]
    assert RC.random_fr_bytes(4, 1).tobytes() == RC.random_fr_bytes(4, 1).tobytes()
    assert all(x < P.R for x in RC.fr_ints(RC.random_fr_bytes(64, 9)))
