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
