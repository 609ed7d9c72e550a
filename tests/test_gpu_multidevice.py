"""N GPUs behind ONE handle of ONE process (round-3 verdict, missing 3; SURVEY 8b `zk_set_devices`): what an OCaml host reaches through the ctypes shim --
`Groth16.Make(C).prove` (src/groth16/groth16.ml:235-237, src/lib/zk/protocol.mli:3-28) is one call on one key and knows nothing of ranks.
Device lists come from the box (`physical`): k distinct cards when it has them, otherwise -- the one-card test box -- the list names card 0 several times ("virtual devices": each entry its own context, streams, tables, slots and
shard); every path but the physical peer copy is the one N real GPUs take.  Proof bytes = the single-device prover's = the trapdoor oracle's
(groth16.ml:123-161: the sums do not depend on how they are cut)."""
import ctypes as C

import numpy as np
import pytest

import oracle_lib as O
from oracle import pyref as P
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16, shard_bounds

pytestmark = pytest.mark.gpu


def frs(xs):
    return b"".join(P.fr_to_bytes(x) for x in xs)


def csrs(cs):
    return [O.CSR(M.ptr, M.col, M.val) for M in (cs.L, cs.R, cs.O)]


def seeded_rng(seed):
    st = P.fr_stream(seed)
    return lambda: next(st)


def physical(devs):
    """The device list a test asks for, on the cards the box HAS: k distinct HIP devices when at least k are visible (hipMemcpyPeerAsync over xGMI and
    per-device contexts run for real -- an 8-GPU node exercises them with no edit), the one card listed k times otherwise (virtual devices)."""
    k = len(devs)
    return list(range(k)) if _lib.lib().zk_device_count() >= k else list(devs)


@pytest.fixture
def devices():
    """sets the library's device list for the test and puts the one-entry list back afterwards (no key may be alive at either switch)"""
    _lib.check(_lib.lib().zk_init(0))
    yield _lib.set_device_list
    _lib.set_device_list([0])
    assert _lib.device_list() == [0]


@pytest.mark.parametrize("devs,n", [([0, 0], 300), ([0, 0], 4096), ([0, 0, 0, 0], 4096), ([0, 0, 0], 1000), ([0, 0], 1 << 17), ([0, 0, 0, 0], 1 << 17),
                                    ([0] * 8, 1 << 14)])
def test_multi_device_key_gives_the_single_device_bytes(devices, devs, n):
    cs, w = RC.iterated_cubic(n, 0xABCD + n)
    rng = seeded_rng(0x5EED0F00 + n + len(devs))
    toxic = [rng() for _ in range(5)]
    it = iter(toxic)
    pk, vk = Groth16.keygen(lambda: next(it), cs, lagrange=True)
    L, R_, Oo = csrs(cs)
    # proofs in flight: five on two or three list entries, fewer on longer lists -- every entry of the list is another set of streams on the ONE card of
    # this box, and the card's scratch aperture is shared by all of its queues (on N real devices each has its own)
    rs = [(rng(), rng()) for _ in range(5 if len(devs) <= 3 else (4 if len(devs) == 4 else 2))]
    exp = [O.groth16_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(r), P.fr_to_bytes(s)) for r, s in rs]
    single = Groth16(cs, pk)
    ref0 = single.prove_rs(w, *rs[0])
    v_ref = single.qap_eval(w)
    single.close()
    assert (ref0.a, ref0.b, ref0.c) == exp[0]

    devs = physical(devs)
    devices(devs)
    assert _lib.device_list() == devs
    prover = Groth16(cs, pk)                                   # the SAME call: zk_groth16_pk_upload shards over the list behind one handle
    assert prover.handle.value >= 0x6000000000
    # a live key pins the device list
    with pytest.raises(_lib.ZkError):
        _lib.set_device_list([0])
    # the handle presents the whole pools (the reference's pkey, groth16.ml:24-34) and refuses the per-rank shard API
    v = [C.c_uint64() for _ in range(6)]
    _lib.check(_lib.lib().zk_groth16_pool_layout(prover.handle, *[C.byref(x) for x in v]))
    assert [x.value for x in v] == [len(pk.g1) // 96, len(pk.g2) // 192, 0, len(pk.g1) // 96, 0, len(pk.g2) // 192]
    part = np.zeros(768, dtype=np.uint8)
    assert _lib.lib().zk_groth16_prove_partial_wait(prover.handle, C.c_uint32(0), part.ctypes.data_as(C.POINTER(C.c_uint8))) == -1
    assert bool((prover.pool_points(1) == pk.g1).all()) and bool((prover.pool_points(2) == pk.g2).all())
    got = prover.prove_rs(w, *rs[0])                            # synchronous: slot 0, Fr stage on the first device
    assert (got.a, got.b, got.c) == exp[0]
    vv = prover.qap_eval(w)
    assert all(bytes(a) == bytes(b) for a, b in zip(vv, v_ref))
    # pipelined over slots: the owner of the Fr stage rotates over the devices (slot mod N), the witness is resident on all of them
    prover.set_witness(w)
    prover.reserve_slots(len(rs))
    for slot, (r, s) in enumerate(rs):
        prover.prove_async(None, r, s, slot)
    for slot in range(len(rs)):
        g = prover.prove_wait(slot)
        assert (g.a, g.b, g.c) == exp[slot], "slot %d (owner device entry %d)" % (slot, slot % len(devs))
    # an unsatisfied witness is reported from whichever device owned the proof (QAP.ml:134), and the slot stays usable
    w_bad = list(w)
    w_bad[n // 2] = (w_bad[n // 2] + 1) % RC.FR_MODULUS
    prover.prove_async(w_bad, *rs[1], 1)
    with pytest.raises(AssertionError):
        prover.prove_wait(1)
    prover.prove_async(w, *rs[1], 1)
    g = prover.prove_wait(1)
    assert (g.a, g.b, g.c) == exp[1]
    # the key's Lagrange form, derived across the devices (one set per device, copied to all, every shard installed): all bytes of both pools equal
    # what a keygen that knows tau emits, and the proofs do not change
    prover.derive_lagrange()
    g1, g2 = prover.pool_points(1), prover.pool_points(2)
    assert g1.shape == pk.lag_g1.shape and bool((g1 == pk.lag_g1).all())
    assert g2.shape == pk.lag_g2.shape and bool((g2 == pk.lag_g2).all())
    for slot, (r, s) in enumerate(rs[:3]):
        prover.prove_async(w, r, s, slot)
    for slot in range(len(rs[:3])):
        g = prover.prove_wait(slot)
        assert (g.a, g.b, g.c) == exp[slot], "derived key, slot %d" % slot
    io_vals = [w[k] for k in range(cs.m) if not cs.mid[k]]
    assert Groth16.verify(io_vals, vk, g)
    prover.close()


def test_multi_device_lagrange_extension_upload_and_mask_form(devices):
    """zk_groth16_pk_upload_lagrange on a device list, and zk_set_devices(mask) -- the SURVEY 8b spelling -- for the one card this box has"""
    n = 2048
    cs, w = RC.iterated_cubic(n, 0x77)
    rng = seeded_rng(0x5EED0F77)
    toxic = [rng() for _ in range(5)]
    it = iter(toxic)
    pk, _ = Groth16.keygen(lambda: next(it), cs, lagrange=True)
    L, R_, Oo = csrs(cs)
    r, s = rng(), rng()
    exp = O.groth16_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), frs(toxic), P.fr_to_bytes(r), P.fr_to_bytes(s))
    devices(physical([0, 0, 0]))
    prover = Groth16(cs, pk, lagrange=True)
    g = prover.prove_rs(w, r, s)
    assert (g.a, g.b, g.c) == exp
    prover.close()
    _lib.check(_lib.lib().zk_set_devices(C.c_uint64(1)))       # mask 0b1 = device 0
    assert _lib.device_list() == [0]
    assert _lib.lib().zk_set_devices(C.c_uint64(0)) == -1       # empty mask
    assert _lib.lib().zk_set_devices(C.c_uint64(1 << 40)) == -1  # no such device
    assert _lib.device_list() == [0]                            # a refused list leaves the old one in place
    prover = Groth16(cs, pk)
    assert prover.handle.value < 0x6000000000                   # one entry: a plain single-device key
    g = prover.prove_rs(w, r, s)
    assert (g.a, g.b, g.c) == exp
    prover.close()


def test_shards_of_a_multi_device_key_follow_the_equal_work_cuts(devices):
    """the slices the library cuts are zk_groth16_shard_range's (mirrored by groth16.shard_bounds): equal WORK, the A prefix counting twice"""
    for size, heavy, world in ((196612, 65539, 8), (13, 8, 3)):
        for g in range(world):
            lo, hi = C.c_uint64(), C.c_uint64()
            _lib.check(_lib.lib().zk_groth16_shard_range(C.c_uint64(size), C.c_uint64(heavy), C.c_uint32(g), C.c_uint32(world), C.byref(lo), C.byref(hi)))
            assert (lo.value, hi.value) == shard_bounds(size, g, world, heavy)


def test_multi_device_key_error_paths(devices):
    """More list entries than key points (the README circuit's G2 pool has 7), misuse of slots, and an unsatisfied witness on the synchronous call:
    errors come back as codes / the reference's exceptions, nothing stays alive or busy."""
    cs, w = RC.readme_circuit(3)
    rng = seeded_rng(0x5EED0F99)
    pk, _ = Groth16.keygen(rng, cs)
    devices([0] * 8)
    with pytest.raises(_lib.ZkError):
        Groth16(cs, pk)                                         # a shard would be empty: ZK_ERR_ARG from every failing shard, the others are released
    _lib.set_device_list(physical([0, 0]))                      # possible only because no handle stayed behind
    prover = Groth16(cs, pk)
    out = np.zeros(384, dtype=np.uint8)
    p8 = out.ctypes.data_as(C.POINTER(C.c_uint8))
    assert _lib.lib().zk_groth16_prove_wait(prover.handle, C.c_uint32(3), p8) == -1            # slot never used
    r, s = rng(), rng()
    prover.prove_async(w, r, s, 1)
    with pytest.raises(_lib.ZkError):
        prover.prove_async(w, r, s, 1)                          # still in flight
    good = prover.prove_wait(1)
    assert _lib.lib().zk_groth16_prove_wait(prover.handle, C.c_uint32(1), p8) == -1            # nothing in flight any more
    w_bad = list(w)
    w_bad[4] = (w_bad[4] + 1) % RC.FR_MODULUS
    with pytest.raises(AssertionError):
        prover.prove_rs(w_bad, r, s)                            # QAP.ml:134 through the synchronous call (owner = the first device)
    again = prover.prove_rs(w, r, s)
    assert (again.a, again.b, again.c) == (good.a, good.b, good.c)
    prover.close()


@pytest.mark.parametrize("devs,maker", [([0, 0], lambda: RC.readme_circuit(3)), ([0, 0, 0], lambda: RC.random_r1cs(1, 6, 76, nnz=(1, 2))), ([0, 0], lambda: RC.iterated_cubic(300, 0xA1)),
                                        ([0, 0, 0], lambda: RC.random_r1cs(200, 150, 0xA2)), ([0, 0, 0, 0], lambda: RC.iterated_cubic(4096, 0xA3)),
                                        ([0, 0], lambda: RC.iterated_cubic(1 << 16, 0xA4))])
def test_multi_device_pinocchio_key_gives_the_single_device_bytes(devices, devs, maker):
    """Round 5: Pinocchio behind a device list too (csrc/pinocchio.hip, PinGroup).  zk_pinocchio_pk_upload cuts every one of the eight pools in N slices,
    a proof's Fr stage and scalar vectors run once on the slot's owner device, every device multiplies its slices and the first one adds the 1 920-byte
    blocks of partial sums -- ZKCompute.f's products (pinocchio.ml:438-505) are sums over key points, so the bytes cannot depend on the cut: they are
    compared with the single-device prover's and the trapdoor oracle's, as uploaded and with the h bases derived (on the first device, then installed
    slice by slice), blocking and pipelined over slots with rotating owners; the handle presents the WHOLE pools.  Tiny circuits leave some devices with
    EMPTY slices of some pools (the readme circuit has three mids; n = 1 has a two-point h pool)."""
    from zukelang_amd import pinocchio as PIN
    cs, w = maker()
    L, R_, Oo = csrs(cs)
    rng = seeded_rng(0x5EED0A00 + cs.n + len(devs))
    tox = [rng() for _ in range(8)]
    it = iter(tox)
    pk, _vk = PIN.ZK.keygen(lambda: next(it), cs)
    ds = [[rng() for _ in range(3)] for _ in range(4)] + [[0, 0, 0]]
    exp = [O.pinocchio_prove_trapdoor(cs.n, cs.m, L, R_, Oo, cs.mid, frs(w), frs(tox), *(P.fr_to_bytes(x) for x in d)) for d in ds]
    single = PIN.ZK(cs, pk)
    assert single.prove_with(w, *ds[0]).to_bytes() == exp[0]
    pools = [bytes(single.pool_points(i)) for i in range(8)]
    single.derive_lagrange()
    pool5_derived = bytes(single.pool_points(5))
    single.close()

    devs = physical(devs)
    devices(devs)
    prover = PIN.ZK(cs, pk)                                    # the SAME call: zk_pinocchio_pk_upload shards over the list behind one handle
    assert prover.handle.value >= 0x7000000000
    with pytest.raises(_lib.ZkError):                          # a live key pins the device list
        _lib.set_device_list([0])
    assert [bytes(prover.pool_points(i)) for i in range(8)] == pools
    for stage in ("as uploaded", "derived"):
        if stage == "derived":
            prover.derive_lagrange()
            prover.derive_lagrange()                           # idempotent
            assert bytes(prover.pool_points(5)) == pool5_derived
            assert [bytes(prover.pool_points(i)) for i in range(8) if i != 5] == [p for i, p in enumerate(pools) if i != 5]
        for d, e in zip(ds[:2] + ds[4:], exp[:2] + exp[4:]):
            assert prover.prove_with(w, *d).to_bytes() == e, (stage, "blocking")
        prover.set_witness(w)
        for slot, d in enumerate(ds):                          # five proofs in flight: owners 0, 1, ..., wrapping round the list
            prover.prove_async(*d, slot)
        for slot in range(len(ds)):
            assert prover.prove_wait(slot).to_bytes() == exp[slot], (stage, "slot %d" % slot)
    if cs.n > 1 and cs.m == cs.n + 2:                          # the iterated-cubic family (and the README circuit): variable 2 sits in a gate for sure
        w_bad = list(w)
        w_bad[2] = (w_bad[2] + 1) % RC.FR_MODULUS
        with pytest.raises(AssertionError):                    # QAP.ml:134, from the owner device's Fr stage
            prover.prove_with(w_bad, *ds[0])
        assert prover.prove_with(w, *ds[1]).to_bytes() == exp[1]      # the slot is usable afterwards
    prover.close()


def test_multi_device_pinocchio_handle_refuses_what_a_single_device_handle_refuses(devices):
    """slot discipline and error paths of the multi-device Pinocchio handle (the checks of tests/test_gpu_api_errors.py::test_pinocchio_slot_misuse, plus a
    key with a point outside the curve on a shard that is not the first: the upload fails as a whole and leaves no handle behind)."""
    from zukelang_amd import pinocchio as PIN
    cs, w = RC.iterated_cubic(16, 5)
    pk, _ = PIN.ZK.keygen(seeded_rng(0xE45), cs)
    devices(physical([0, 0, 0]))
    pr = PIN.ZK(cs, pk)
    with pytest.raises(_lib.ZkError):
        pr.prove_async(1, 2, 3, 0)                         # no resident witness
    pr.set_witness(w)
    pr.prove_async(1, 2, 3, 1)
    with pytest.raises(_lib.ZkError):
        pr.prove_async(1, 2, 3, 1)                         # the slot is busy
    with pytest.raises(_lib.ZkError):
        pr.derive_lagrange()                               # a proof is in flight
    first = pr.prove_wait(1)
    with pytest.raises(_lib.ZkError):
        pr.prove_wait(3)                                   # slot never used
    with pytest.raises(_lib.ZkError):
        pr.prove_wait(1)                                   # nothing in flight any more
    pr.derive_lagrange()
    pr.prove_async(1, 2, 3, 2)
    assert pr.prove_wait(2).to_bytes() == first.to_bytes()
    pr.close()
    # a bad point in the LAST third of the vv pool (shard 2 decodes it): (1, 1) is not on y^2 = x^3 + 4
    bad = np.array(pk.g1, copy=True)
    nm = sum(1 for k in range(cs.m) if cs.mid[k])
    off = 96 * (nm - 1)
    bad[off:off + 96] = 0
    bad[off + 47] = 1
    bad[off + 95] = 1
    before = _lib.lib().zk_device_count()
    with pytest.raises(_lib.ZkError):
        PIN.ZK(cs, PIN.PKey(bad, pk.g2))
    assert _lib.lib().zk_device_count() == before
    _lib.set_device_list([0])                              # succeeds only if the failed upload left no key handle alive
    _lib.set_device_list(physical([0, 0, 0]))
