"""GPU parity of zk_g1/g2_decompress_batch (of_compressed_bytes_exn over a list, src/lib/zk/curve.ml:199-212) with the one-point host functions
zk_g1/g2_decompress -- which tests/test_wire.py and tests/test_pairing_host.py hold to the oracle -- on valid points of every shape (both signs,
the identity, y with a zero imaginary part, multiples of the generator and hashed-looking points), on every kind of invalid encoding, and through
zukelang_amd/wire.py on a key large enough to take the batched path."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import _lib, r1cs as RC, wire
from zukelang_amd.curve import G1, G2

pytestmark = pytest.mark.gpu


def host_one(grp, comp):
    out = C.create_string_buffer(grp.POINT_BYTES)
    rc = getattr(_lib.lib(), "zk_g1_decompress" if grp is G1 else "zk_g2_decompress")(bytes(comp), out)
    return rc, out.raw


def batch(grp, comps):
    n = len(comps)
    out = np.zeros(n * grp.POINT_BYTES, dtype=np.uint8)
    buf = np.frombuffer(b"".join(comps), dtype=np.uint8)
    rc = getattr(_lib.lib(), "zk_g1_decompress_batch" if grp is G1 else "zk_g2_decompress_batch")(buf.ctypes.data_as(C.c_void_p), C.c_size_t(n), out.ctypes.data_as(C.c_void_p))
    return rc, bytes(out)


@pytest.mark.parametrize("grp,gen,mul", [(G1, O.g1_generator, O.g1_mul), (G2, O.g2_generator, O.g2_mul)])
def test_batch_decompression_equals_the_host_function(grp, gen, mul):
    _lib.check(_lib.lib().zk_init(0))
    st = P.fr_stream(0x5EEDDEC0 + grp.POINT_BYTES)
    ks = [1, 2, 3, P.R - 1, P.R - 2] + [next(st) for _ in range(600)]
    pts = grp.of_Fr(RC.fr_bytes(ks))                                  # s_i G on the fixed-base kernel: both signs of y occur
    B, Cb = grp.POINT_BYTES, grp.COMPRESSED_BYTES
    comps = [grp.to_compressed_bytes(pts[B * i:B * (i + 1)]) for i in range(len(ks))]
    inf = bytes([0xC0]) + bytes(Cb - 1)
    comps = comps[:300] + [inf] + comps[300:] + [inf]
    rc, out = batch(grp, comps)
    assert rc == 0
    for i, c in enumerate(comps):
        rc1, ref = host_one(grp, c)
        assert rc1 == 0 and out[B * i:B * (i + 1)] == ref, i
    want = bytes(pts[:B * 300]) + bytes([0x40]) + bytes(B - 1) + bytes(pts[B * 300:]) + bytes([0x40]) + bytes(B - 1)
    assert out == want                                                 # ... and both are the points that were compressed
    assert grp.of_compressed_bytes_many(b"".join(comps)) == want
    assert grp.of_compressed_bytes_many(b"") == b""


def _bad_cases(grp):
    Cb = grp.COMPRESSED_BYTES
    good = grp.to_compressed_bytes(grp.of_Fr(RC.fr_bytes([5])))
    no_flag = bytes([good[0] & 0x7F]) + good[1:]                       # compression bit missing
    big = bytes([0x80 | 0x1F]) + bytes([0xFF]) * (Cb - 1)              # coordinate >= p
    # an abscissa whose x^3 + b is not a square: search small x
    x = 0
    while True:
        x += 1
        cand = bytes([0x80]) + x.to_bytes(Cb - 1, "big")
        rc, _ = host_one(grp, cand)
        if rc != 0:
            off_curve = cand
            break
    # on the curve, outside the prime-order subgroup: the first small x with a square right-hand side that the host rejects for the subgroup
    x = 0
    outside = None
    while outside is None and x < 4000:
        x += 1
        cand = bytes([0x80]) + x.to_bytes(Cb - 1, "big")
        rc, _ = host_one(grp, cand)
        if rc != 0 and b"subgroup" in _lib.lib().zk_last_error():
            outside = cand
    return good, {"no_flag": no_flag, "coordinate >= p": big, "off the curve": off_curve, "outside the subgroup": outside}


@pytest.mark.parametrize("grp", [G1, G2])
def test_batch_decompression_rejects_what_the_host_function_rejects(grp):
    _lib.check(_lib.lib().zk_init(0))
    _lib.lib().zk_last_error.restype = C.c_char_p
    good, bad = _bad_cases(grp)
    assert bad["outside the subgroup"] is not None
    for what, c in bad.items():
        rc1, _ = host_one(grp, c)
        assert rc1 != 0, what
        for pos in (0, 7, 299):                                       # one bad point anywhere in a list fails the list with the host's code
            comps = [good] * 300
            comps[pos] = c
            rc, _ = batch(grp, comps)
            assert rc == rc1, (what, pos, rc, rc1)
    rc, _ = batch(grp, [good] * 300)
    assert rc == 0


def test_a_key_in_the_reference_json_is_read_through_the_batched_path():
    """wire.groth16_pkey_of_json / pinocchio_pkey_of_json on keys of a few thousand points (above wire.BATCH_MIN): the points come back exactly."""
    from zukelang_amd.groth16 import Groth16
    from zukelang_amd import pinocchio as PIN
    cs, w = RC.iterated_cubic(600, 0xBEEF)
    st = P.fr_stream(0x5EEDDEC1)
    pk, _ = Groth16.keygen(lambda: next(st), cs)
    mids = [k for k in range(cs.m) if cs.mid[k]]
    mid_vars = [("v%d" % k, k) for k in mids]
    data = wire.groth16_pkey_to_json(pk, cs.n, mid_vars)
    back, mv = wire.groth16_pkey_of_json(data)
    assert bytes(back.g1) == bytes(pk.g1) and bytes(back.g2) == bytes(pk.g2) and mv == mid_vars
    assert len(pk.g1) // 96 > wire.BATCH_MIN and len(pk.g2) // 192 > wire.BATCH_MIN
    ppk, _ = PIN.ZK.keygen(lambda: next(st), cs)
    all_vars = [("v%d" % k, k) for k in range(cs.m)]
    pdata = wire.pinocchio_pkey_to_json(ppk, cs.n, mid_vars, all_vars)
    pback, n, mv, av = wire.pinocchio_pkey_of_json(pdata)
    assert bytes(pback.g1) == bytes(ppk.g1) and bytes(pback.g2) == bytes(ppk.g2) and n == cs.n and mv == mid_vars and av == all_vars
