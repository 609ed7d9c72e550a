"""Error behaviour of the pipelined / split entry points (the C-ABI returns negative codes, the Python host raises)."""
import ctypes as C

import numpy as np
import pytest

from oracle import pyref as P
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd import pinocchio as PIN
from zukelang_amd.groth16 import Groth16, fr_bytes, _p

pytestmark = pytest.mark.gpu


def seeded(seed):
    st = P.fr_stream(seed)
    return lambda: next(st)


def test_groth16_slot_misuse_and_null_arguments():
    cs, w = RC.iterated_cubic(64, 3)
    rng = seeded(0xE44)
    pk, _ = Groth16.keygen(rng, cs, lagrange=True)
    pr = Groth16(cs, pk)
    L = _lib.lib()
    rb, sb = fr_bytes([5]), fr_bytes([7])
    # no witness resident yet
    assert L.zk_groth16_prove_async(pr.handle, None, _p(rb), _p(sb), C.c_uint32(0)) != 0
    pr.set_witness(w)
    pr.prove_async(None, 5, 7, 0)
    with pytest.raises(_lib.ZkError):
        pr.prove_async(None, 5, 7, 0)                      # slot 0 still has a proof in flight
    pr.prove_wait(0)
    with pytest.raises(_lib.ZkError):
        pr.prove_wait(0)                                   # nothing in flight any more
    with pytest.raises(_lib.ZkError):
        pr.prove_async(None, 5, 7, 99)                     # slot index out of range
    assert L.zk_groth16_scalars_async(pr.handle, None, _p(rb), _p(sb), C.c_uint32(1), None, None, None) != 0
    assert L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(1), None, None, None) != 0
    part = np.zeros(768, dtype=np.uint8)
    assert L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(7), _p(part)) != 0        # slot never used
    pr.close()
    lag = Groth16(cs, pk, lagrange=True)
    v0, w0, h0 = Groth16(cs, pk).qap_eval(w)
    v1, w1, h1 = lag.qap_eval(w)                           # QAP.eval stays available on a Lagrange-form key (it runs the basis conversion)
    assert bytes(v0) == bytes(v1) and bytes(w0) == bytes(w1) and bytes(h0) == bytes(h1)
    lag.close()
    with pytest.raises(ValueError):
        pk2, _ = Groth16.keygen(seeded(1), cs)
        Groth16(cs, pk2, lagrange=True)                    # key without the extension


def test_pinocchio_slot_misuse():
    cs, w = RC.iterated_cubic(16, 5)
    rng = seeded(0xE45)
    pk, _ = PIN.ZK.keygen(rng, cs)
    pr = PIN.ZK(cs, pk)
    with pytest.raises(_lib.ZkError):
        pr.prove_async(1, 2, 3, 0)                         # no resident witness
    pr.set_witness(w)
    pr.prove_async(1, 2, 3, 0)
    with pytest.raises(_lib.ZkError):
        pr.prove_async(1, 2, 3, 0)
    pr.prove_wait(0)
    with pytest.raises(_lib.ZkError):
        pr.prove_wait(3)                                   # slot never used
    pr.close()


def _point_outside_the_subgroup():
    """a point of E(Fp): y^2 = x^3 + 4 whose order does not divide r (the cofactor part of the curve group)"""
    x = 0
    while True:
        x += 1
        y2 = (x ** 3 + 4) % P.P
        y = pow(y2, (P.P + 1) // 4, P.P)
        if y * y % P.P == y2:
            pt = (P.Fp1(x), P.Fp1(y))
            if P.pt_mul(pt, P.R) is not None:
                return pt


def test_key_points_outside_the_prime_order_subgroup_are_rejected_at_upload():
    """The reference's keys hold Bls12_381.G1/G2 values, which of_bytes_exn / of_compressed_bytes_exn (curve.ml:199-212) refuse to build from a point
    of the curve outside the r-torsion; a key uploaded as raw bytes gets the same check on the device ([r] P = O), for both protocols.  Identity
    points stay legal and a point off the curve is still its own error."""
    cs, w = RC.iterated_cubic(8, 3)
    rng = seeded(0xE46)
    pk, _ = Groth16.keygen(rng, cs)
    Groth16(cs, pk).close()                                              # the honest key passes
    bad = P.g1_to_bytes(_point_outside_the_subgroup())
    for idx in (0, 5, len(pk.g1) // 96 - 1):
        g1 = np.array(pk.g1, dtype=np.uint8, copy=True)
        g1[96 * idx:96 * idx + 96] = np.frombuffer(bad, dtype=np.uint8)
        with pytest.raises(_lib.ZkError) as e:
            Groth16(cs, type(pk)(g1, pk.g2))
        assert e.value.code == -2 and "subgroup" in str(e.value)
    inf = np.zeros(96, dtype=np.uint8); inf[0] = 0x40
    g1 = np.array(pk.g1, dtype=np.uint8, copy=True)
    g1[96 * 4:96 * 5] = inf                                              # the identity is a member of every subgroup
    Groth16(cs, type(pk)(g1, pk.g2)).close()
    off = np.array(pk.g1, dtype=np.uint8, copy=True)
    off[96 * 3 + 95] ^= 1
    with pytest.raises(_lib.ZkError) as e:
        Groth16(cs, type(pk)(off, pk.g2))
    assert e.value.code == -2 and "subgroup" not in str(e.value)
    pkp, _ = PIN.ZK.keygen(seeded(0xE47), cs)
    g1p = np.array(pkp.g1, dtype=np.uint8, copy=True)
    g1p[96 * 2:96 * 3] = np.frombuffer(bad, dtype=np.uint8)
    with pytest.raises(_lib.ZkError):
        PIN.ZK(cs, type(pkp)(g1p, pkp.g2))


def _set_option(name, value):
    _lib.check(_lib.lib().zk_set_option(name.encode(), None if value is None else str(value).encode()))


def test_a_key_uploaded_without_the_subgroup_check_never_gets_folded_windows():
    """Folded digits take min(s, r - s) -- one window fewer at the widths that divide 255 (c = 3, 5, 15, 17; 17 is the default from 2^20 pool points) --
    and (r - s)(-P) = s P only for points of order r.  The width and the check used to be two unrelated environment knobs (ADVICE r4
    medium): a key uploaded with the check skipped and one point outside the subgroup gave a wrong sum and ZK_OK.  Now a base set folds only
    when the [r] P = O test ran on it.  Here: the check switched off through zk_set_option (the C-ABI's own configuration call), a key with two
    points of the curve OUTSIDE the prime-order subgroup, every folding width.  For such points the only well-defined answer is the plain sum
    sum_i s_i P_i over the canonical scalars s_i in [0, r) (the reference's nested folds, groth16.ml:116-121, agree with it only on the subgroup), so
    that is the expectation: the three scalar vectors read back from the device, the pools read back from the key, the oracle's double-and-add.
    With the check back on, the same key is refused."""
    import oracle_lib as O
    cs, w = RC.iterated_cubic(8, 3)
    csr = [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]
    rng = seeded(0xF01D)
    tox = [rng() for _ in range(5)]
    it = iter(tox)
    pk, _ = Groth16.keygen(lambda: next(it), cs)
    bad = np.frombuffer(P.g1_to_bytes(_point_outside_the_subgroup()), dtype=np.uint8)
    g1 = np.array(pk.g1, dtype=np.uint8, copy=True)
    for idx in (4, len(g1) // 96 - 2):                                    # tau^1 (serves A and C) and one of ltd_mid
        g1[96 * idx:96 * idx + 96] = bad
    key = type(pk)(g1, pk.g2)
    frs = lambda xs: bytes(RC.fr_bytes(xs))
    L = _lib.lib()
    r, s = rng(), rng()

    def plain_sums(pr):
        p1, p2 = bytes(pr.pool_points(1)), bytes(pr.pool_points(2))
        n1, n2 = len(p1) // 96, len(p2) // 192
        d = [C.c_void_p() for _ in range(3)]
        host = [np.zeros(32 * k, dtype=np.uint8) for k in (n1, n1, n2)]
        for ptr, h in zip(d, host):
            _lib.check(L.zk_device_malloc(C.c_size_t(len(h)), C.byref(ptr)))
        _lib.check(L.zk_groth16_scalars_async(pr.handle, _p(fr_bytes(w)), _p(fr_bytes([r])), _p(fr_bytes([s])), C.c_uint32(1), *d))
        _lib.check(L.zk_groth16_scalars_wait(pr.handle, C.c_uint32(1)))
        for ptr, h in zip(d, host):
            _lib.check(L.zk_device_memcpy(h.ctypes.data_as(C.c_void_p), ptr, C.c_size_t(len(h))))
            _lib.check(L.zk_device_free(ptr))
        (rc1, a), (rc2, c), (rc3, b) = O.g1_msm_naive(p1, bytes(host[0])), O.g1_msm_naive(p1, bytes(host[1])), O.g2_msm_naive(p2, bytes(host[2]))
        assert rc1 == rc2 == rc3 == 0
        return a, b, c

    try:
        _set_option("key_subgroup_check", 0)
        for width in (17, 15, 5, 3, 16):
            _set_option("ZK_MSM_WINDOW", width)
            pr = Groth16(cs, key)
            assert bytes(pr.pool_points(1)) == bytes(g1)
            got = pr.prove_rs(w, r, s)
            assert (got.a, got.b, got.c) == plain_sums(pr), "window %d" % width
            pr.derive_lagrange()                                          # the derived pools inherit "not checked"
            got = pr.prove_rs(w, r, s)
            assert (got.a, got.b, got.c) == plain_sums(pr), "window %d, derived key" % width
            pr.close()
    finally:
        _set_option("msm_window", None)
        _set_option("key_subgroup_check", None)
    with pytest.raises(_lib.ZkError) as e:
        Groth16(cs, key)
    assert e.value.code == -2 and "subgroup" in str(e.value)
    # ... and the honest key still folds (same bytes as the oracle at a folding width, check on)
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, *csr, cs.mid, frs(w), frs(tox), frs([r]), frs([s]))
    try:
        _set_option("msm_window", 17)
        pr = Groth16(cs, pk)
        got = pr.prove_rs(w, r, s)
        assert (got.a, got.b, got.c) == exp and (got.a, got.b, got.c) == plain_sums(pr)
        pr.close()
    finally:
        _set_option("msm_window", None)


def test_set_option_names():
    L = _lib.lib()
    assert L.zk_set_option(b"no_such_knob", b"1") == -1 and L.zk_set_option(None, b"1") == -1 and L.zk_set_option(b"", b"1") == -1
    for name in (b"msm_window", b"ZK_MSM_WINDOW", b"Msm_Window", b"key_subgroup_check", b"slot_streams", b"graph", b"derive_side_by_side"):
        assert L.zk_set_option(name, b"16" if b"indow" in name.lower() else b"1") == 0
        assert L.zk_set_option(name, None) == 0


def test_caller_owned_scalar_vectors_are_range_checked():
    """zk_groth16_msm_partial_async takes device vectors nobody has looked at: a value >= r must come back as ZK_ERR_SCALAR_RANGE from the
    matching wait, not as a wrong sum."""
    cs, w = RC.iterated_cubic(64, 3)
    rng = seeded(0xE48)
    pk, _ = Groth16.keygen(rng, cs)
    pr = Groth16(cs, pk)
    L = _lib.lib()
    p1, p2 = len(pk.g1) // 96, len(pk.g2) // 192
    d = [C.c_void_p() for _ in range(3)]
    for ptr, size in zip(d, (32 * p1, 32 * p1, 32 * p2)):
        _lib.check(L.zk_device_malloc(C.c_size_t(size), C.byref(ptr)))
    rb, sb, wb = fr_bytes([5]), fr_bytes([7]), fr_bytes(w)
    _lib.check(L.zk_groth16_scalars_async(pr.handle, _p(wb), _p(rb), _p(sb), C.c_uint32(0), *d))
    _lib.check(L.zk_groth16_scalars_wait(pr.handle, C.c_uint32(0)))
    part = np.zeros(768, dtype=np.uint8)
    _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(0), *d))
    _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(0), _p(part)))

    def combined():          # the partial sums are raw projective limbs (not canonical: bucket order may differ run to run); the proof they combine to is
        out = np.zeros(384, dtype=np.uint8)
        _lib.check(L.zk_groth16_combine(_p(part), C.c_uint32(1), _p(out)))
        return bytes(out)
    good = combined()
    ref = pr.prove_rs(w, 5, 7)
    assert good == ref.a + ref.b + ref.c
    for which, count in ((0, p1), (1, p1), (2, p2)):
        big = np.frombuffer((P.R + 3).to_bytes(32, "little"), dtype=np.uint8).copy()          # r + 3 < 2^255: a non-canonical encoding of 3
        saved = np.zeros(32, dtype=np.uint8)
        at = C.c_void_p(d[which].value + 32 * (count - 1))
        _lib.check(L.zk_device_memcpy(saved.ctypes.data_as(C.c_void_p), at, C.c_size_t(32)))
        _lib.check(L.zk_device_memcpy(at, big.ctypes.data_as(C.c_void_p), C.c_size_t(32)))
        _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(0), *d))
        assert L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(0), _p(part)) == -3, which
        _lib.check(L.zk_device_memcpy(at, saved.ctypes.data_as(C.c_void_p), C.c_size_t(32)))
    _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(0), *d))
    _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(0), _p(part)))
    assert combined() == good
    for ptr in d:
        _lib.check(L.zk_device_free(ptr))
    pr.close()
