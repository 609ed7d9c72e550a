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
