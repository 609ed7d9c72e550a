"""GPU parity: zk_fr_ntt (HIP) == the oracle's restatement of FFT.ml:29-86, bit-exact."""
import random

import numpy as np
import pytest

import oracle_lib as O
from zukelang_amd import r1cs as RC
from zukelang_amd.curve import FFT_Fr

pytestmark = pytest.mark.gpu


def _rand_fr(n, seed):
    return bytes(RC.random_fr_bytes(n, seed))


@pytest.mark.parametrize("log_n", [0, 1, 2, 3, 5, 8, 10, 11, 12, 13, 16])
def test_fft_matches_oracle(log_n):
    data = _rand_fr(1 << log_n, 100 + log_n)
    assert bytes(FFT_Fr.fft(data, log_n)) == O.fr_ntt(data, log_n, False)
    assert bytes(FFT_Fr.ifft(data)) == O.fr_ntt(data, log_n, True)


@pytest.mark.parametrize("log_n", [4, 10, 12, 14])
def test_fft_extreme_values(log_n):
    """Largest residues everywhere (and sparse mixes of 0, 1, r - 1): the lazily reduced 29-bit-limb butterflies
    (csrc/fr29.cuh) run at the top of their value bounds."""
    n = 1 << log_n
    top = (RC.FR_MODULUS - 1).to_bytes(32, "little")
    rnd = random.Random(7 + log_n)
    pool = [top, bytes(32), (1).to_bytes(32, "little"), (RC.FR_MODULUS - 2).to_bytes(32, "little"), ((1 << 254) + 5).to_bytes(32, "little")]
    for data in (top * n, b"".join(rnd.choice(pool) for _ in range(n)), b"".join(top if i & 1 else bytes(32) for i in range(n))):
        assert bytes(FFT_Fr.fft(data, log_n)) == O.fr_ntt(data, log_n, False)
        assert bytes(FFT_Fr.ifft(data)) == O.fr_ntt(data, log_n, True)


@pytest.mark.parametrize("log_n", [18, 20, 22])
def test_fft_roundtrip_and_linearity_large(log_n):
    """Full benchmark sizes: fft o ifft = id (FFT.ml:88-96) and linearity -- size-independent."""
    n = 1 << log_n
    a = RC.random_fr_bytes(n, 7)
    b = RC.random_fr_bytes(n, 8)
    fa = FFT_Fr.fft(a, log_n)
    assert bytes(FFT_Fr.ifft(fa)) == bytes(a)
    # spot-check 4 outputs against the definition sum_j a_j w^(jk) using the oracle on a folded input:
    # out[k] for k = 0 is the plain sum of the inputs
    ints = np.frombuffer(bytes(a), dtype=np.uint8).reshape(n, 32)
    total = sum(int.from_bytes(bytes(row), "little") for row in ints[:: max(1, n // 4096)]) if n <= 1 << 12 else None
    if total is not None:
        assert int.from_bytes(bytes(fa[:32]), "little") == total % RC.FR_MODULUS
    # linearity: fft(a) + fft(b) == fft(a + b) checked on a strided sample
    fb = FFT_Fr.fft(b, log_n)
    P = RC.FR_MODULUS
    idx = random.Random(log_n).sample(range(n), 64)
    s = bytearray()
    A = np.frombuffer(bytes(a), dtype=np.uint8).reshape(n, 32)
    B = np.frombuffer(bytes(b), dtype=np.uint8).reshape(n, 32)
    # a + b computed with numpy object ints only on the host for the whole vector is slow; do it in C-order chunks
    ai = [int.from_bytes(bytes(r), "little") for r in A]
    bi = [int.from_bytes(bytes(r), "little") for r in B]
    ab = RC.fr_bytes([(x + y) % P for x, y in zip(ai, bi)])
    fab = FFT_Fr.fft(ab, log_n)
    for k in idx:
        x = int.from_bytes(bytes(fa[32 * k:32 * k + 32]), "little")
        y = int.from_bytes(bytes(fb[32 * k:32 * k + 32]), "little")
        z = int.from_bytes(bytes(fab[32 * k:32 * k + 32]), "little")
        assert (x + y) % P == z


def test_non_canonical_input_is_rejected():
    from zukelang_amd._lib import ZkError
    bad = (RC.FR_MODULUS).to_bytes(32, "little") + bytes(32)
    with pytest.raises(ZkError) as e:
        FFT_Fr.fft(bad, 1)
    assert e.value.code == -3


@pytest.mark.parametrize("log_n", [18, 20])
def test_fft_large_matches_oracle_on_a_strided_sample(log_n):
    """beyond 2^16 the transform was property-tested only.  The oracle's O(n log n) restatement of
    FFT.ml:29-67 handles 2^20 in seconds: compare 256 strided outputs (plus both ends) of fft and ifft bit for bit."""
    n = 1 << log_n
    data = _rand_fr(n, 900 + log_n)
    for inverse, got in ((False, bytes(FFT_Fr.fft(data, log_n))), (True, bytes(FFT_Fr.ifft(data)))):
        ref = O.fr_ntt(data, log_n, inverse)
        idx = list(range(0, n, n // 256)) + [1, n - 1, n // 2 + 1]
        for k in idx:
            assert got[32 * k:32 * k + 32] == ref[32 * k:32 * k + 32], (log_n, inverse, k)
        if log_n == 18:
            assert got == ref


@pytest.mark.parametrize("na,nb", [(1, 1), (1, 7), (2, 2), (3, 5), (17, 1), (64, 64), (100, 29), (255, 257), (1000, 1), (1024, 1024), (4096, 3000), (4096, 4096)])
def test_polynomial_mul_matches_oracle(na, nb):
    """zk_fr_poly_mul == FFT.polynomial_mul (FFT.ml:98-105) == Polynomial.mul (polynomial.ml:124-131, the oracle's
    schoolbook restatement): unequal lengths, sizes that are not powers of two, normalized output."""
    a, b = _rand_fr(na, 5000 + na), _rand_fr(nb, 6000 + nb)
    assert bytes(FFT_Fr.polynomial_mul(a, b)) == O.poly_mul(a, b)
    assert bytes(FFT_Fr.polynomial_mul(b, a)) == O.poly_mul(a, b)


def test_polynomial_mul_normalizes_like_the_reference():
    """Polynomial.normalize (polynomial.ml:100-107): trailing zero coefficients are stripped, the zero polynomial is []."""
    one = (1).to_bytes(32, "little")
    zero = bytes(32)
    top = (RC.FR_MODULUS - 1).to_bytes(32, "little")
    a = _rand_fr(9, 1) + zero * 7                      # trailing zeros in an operand
    b = _rand_fr(5, 2) + zero * 3
    ref = O.poly_mul(a, b)
    assert bytes(FFT_Fr.polynomial_mul(a, b)) == ref and len(ref) == 32 * (9 + 5 - 1)
    assert bytes(FFT_Fr.polynomial_mul(zero * 4, a)) == b"" == O.poly_mul(zero * 4, a)      # 0 * f = []
    assert bytes(FFT_Fr.polynomial_mul(b"", a)) == b""
    assert bytes(FFT_Fr.polynomial_mul(one, a)) == O.poly_mul(one, a) == a[:32 * 9]
    assert bytes(FFT_Fr.polynomial_mul(top * 33, top * 31)) == O.poly_mul(top * 33, top * 31)   # extreme values: (r-1)^2 sums
    # polynomial.ml:135-139 KAT over Fr: (1 + x)(1 - x) = 1 - x^2
    p1, p2 = one + one, one + top
    assert bytes(FFT_Fr.polynomial_mul(p1, p2)) == one + zero + top


def test_rns_convolution_engine_gives_the_same_field_elements():
    """csrc/rns_ntt.hip (round 4, an OPTION: ZK_FR_RNS=1 when a key is uploaded): the products of the Fr stage as integer convolutions modulo 18 NTT-friendly
    31-bit primes, carried back to Fr by the Chinese remainder theorem -- exact arithmetic, so Polynomial.mul (polynomial.ml:124-131), QAP.eval
    (QAP.ml:120-135) and the proofs (groth16.ml:123-161) must come out byte for byte as through the Fr transforms: polynomial products incl. the extreme
    values (r-1)^2 sums, tau-power keys below and above the fused tree levels (n = 4096: levels 11, 12 go through the residue system), a Lagrange-form
    key, 2^16 constraints, Pinocchio."""
    import os
    from oracle import pyref as P
    from zukelang_amd import pinocchio as PIN
    from zukelang_amd.groth16 import Groth16
    old = os.environ.get("ZK_FR_RNS")
    os.environ["ZK_FR_RNS"] = "1"
    try:
        top = (RC.FR_MODULUS - 1).to_bytes(32, "little")
        for na, nb in ((1, 7), (3, 5), (100, 29), (255, 257), (1024, 1024), (4096, 3000), (20000, 12000)):
            a, b = _rand_fr(na, 7000 + na), _rand_fr(nb, 8000 + nb)
            if na * nb <= 4096 * 4096:
                assert bytes(FFT_Fr.polynomial_mul(a, b)) == O.poly_mul(a, b), (na, nb)
            else:                                   # too long for the oracle's schoolbook product: against the Fr transforms
                os.environ["ZK_FR_RNS"] = "0"
                ref = bytes(FFT_Fr.polynomial_mul(a, b))
                os.environ["ZK_FR_RNS"] = "1"
                assert bytes(FFT_Fr.polynomial_mul(a, b)) == ref, (na, nb)
        assert bytes(FFT_Fr.polynomial_mul(top * 1500, top * 1400)) == O.poly_mul(top * 1500, top * 1400)
        frs = lambda xs: b"".join(P.fr_to_bytes(x) for x in xs)
        for n in (2, 6, 300, 1000, 4096, 1 << 16):
            cs, w = RC.iterated_cubic(n, 0x515 + n)
            st = P.fr_stream(0x5EED0A00 + n)
            toxic = [next(st) for _ in range(5)]
            it = iter(toxic)
            pk, _ = Groth16.keygen(lambda: next(it), cs, lagrange=True)
            csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
            r, s = next(st), next(st)
            exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(r), P.fr_to_bytes(s))
            for lag in (False, True):
                prover = Groth16(cs, pk, lagrange=lag)          # the switch is read when the key is uploaded (tables) and per proof
                got = prover.prove_rs(w, r, s)
                assert (got.a, got.b, got.c) == exp, (n, lag)
                if not lag and n <= 4096:
                    os.environ["ZK_FR_RNS"] = "0"
                    ref = [bytes(x) for x in prover.qap_eval(w)]          # the SAME key through the Fr transforms
                    os.environ["ZK_FR_RNS"] = "1"
                    assert [bytes(x) for x in prover.qap_eval(w)] == ref, n
                w_bad = list(w)
                w_bad[n // 2] = (w_bad[n // 2] + 1) % RC.FR_MODULUS
                with pytest.raises(AssertionError):
                    prover.prove_rs(w_bad, r, s)
                prover.close()
        cs, w = RC.iterated_cubic(1000, 11)
        st = P.fr_stream(0x5EED0003)
        tox = [next(st) for _ in range(11)]
        it = iter(tox)
        pk, _ = PIN.ZK.keygen(lambda: next(it), cs)
        prover = PIN.ZK(cs, pk)
        proof = prover.prove(lambda: next(it), w)
        csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
        assert proof.to_bytes() == O.pinocchio_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox[:8]), *(P.fr_to_bytes(x) for x in tox[8:]))
        prover.close()
    finally:
        if old is None:
            os.environ.pop("ZK_FR_RNS", None)
        else:
            os.environ["ZK_FR_RNS"] = old
