"""One rank's share of the N-GPU job, timed on one GPU: what does a group of N proofs cost rank 0?
 distributed Fr stage: 1 Fr stage (n = 2^16 N) + N sharded MSM triples (+ a device-to-device copy standing in for the all-to-all)
 replicated Fr stage : N x (Fr stage + sharded MSM triple)
Projected job throughput = N proofs * n / group time (every rank does the same amount of work).
usage: group_emulation.py [N] [rank] [derived|tau]   (round 2: the key derived into its Lagrange form then sharded with the equal-work cuts --
bench.py's default at N > 1 -- or sharded at upload in tau-power form; the rank whose share is timed)"""
import ctypes as C, os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16, shard_bounds, fr_bytes, _p
L = _lib.lib(); _lib.check(L.zk_init(0))
W = int(sys.argv[1]) if len(sys.argv) > 1 else 8
RANK = int(sys.argv[2]) if len(sys.argv) > 2 else 0
DERIVED = (sys.argv[3] if len(sys.argv) > 3 else "derived") == "derived"
logn = 16
n = (1 << logn) * W
cs, w = RC.iterated_cubic(n, next(RC.fr_stream(1)))
st = RC.fr_stream(2); rng = lambda: next(st)
pk, _ = Groth16.keygen(rng, cs)
B = 7                               # proofs per round, as GroupProver: TWO sets of MSM slots + Fr slots: 2 batch + ceil(batch / W) <= 14
while 2 * B + (B + W - 1) // W > 14: B -= 1
G = (B + W - 1) // W                # Fr stages rank 0 runs per round at most
if DERIVED:
    pr = Groth16(cs, pk)
    t_d = time.perf_counter(); pr.derive_lagrange(); print("derive_lagrange at 2^%d: %.1f s" % (logn + (W.bit_length() - 1), time.perf_counter() - t_d), flush=True)
    pr.shard(RANK, W)
else:
    pr = Groth16(cs, pk, RANK, W)
pr.set_witness(w); pr.reserve_slots(2 * B + G)
v = [C.c_uint64() for _ in range(6)]
_lib.check(L.zk_groth16_pool_layout(pr.handle, *[C.byref(x) for x in v]))
p1, p2, lo1, hi1, lo2, hi2 = (int(x.value) for x in v)
def dmalloc(b):
    p = C.c_void_p(); _lib.check(L.zk_device_malloc(C.c_size_t(b), C.byref(p))); return p.value
full = [[dmalloc(32 * p1), dmalloc(32 * p1), dmalloc(32 * p2)] for _ in range(G)]
l1, l2 = 32 * (hi1 - lo1), 32 * (hi2 - lo2)
recv = [[[dmalloc(W * l1), dmalloc(W * l1), dmalloc(W * l2)] for _ in range(G)] for _ in range(2)]
part = np.zeros(768, dtype=np.uint8)
fr_pending = False
count = 0
def owned(base):
    """how many proofs of a round starting at job index `base` belong to this rank"""
    return len([t for t in range(B) if (base + t) % W == RANK])
def launch_fr():
    global fr_pending, n_launched
    rb, sb = fr_bytes([rng()]), fr_bytes([rng()])
    n_launched = owned(count)
    for k in range(n_launched):
        _lib.check(L.zk_groth16_scalars_async(pr.handle, None, _p(rb), _p(sb), C.c_uint32(2 * B + k), C.c_void_p(full[k][0]), C.c_void_p(full[k][1]), C.c_void_p(full[k][2])))
    fr_pending = True
pending = None        # first slot of the round whose products are in flight
rounds_done = 0
def drain():
    global pending
    if pending is not None:
        for t in range(B):
            _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(pending + t), _p(part)))
        pending = None
def group(distributed):
    """One round as GroupProver.prove_many runs it: the NEXT round's Fr stages are enqueued before this round's products, and this round's
    products before the PREVIOUS round's are waited for (two slot sets)."""
    global fr_pending, count, pending, rounds_done
    rb, sb = fr_bytes([rng()]), fr_bytes([rng()])
    half = rounds_done & 1
    slot0 = half * B
    if distributed:
        if not fr_pending:
            launch_fr()
        for k in range(n_launched):
            _lib.check(L.zk_groth16_scalars_wait(pr.handle, C.c_uint32(2 * B + k)))
        fr_pending = False
        count += B
        rv = recv[half]
        for k in range(G):
            for j in range(W):      # stand-in for the all-to-all: W slices land in recv
                _lib.check(L.zk_device_memcpy(C.c_void_p(rv[k][0] + j * l1), C.c_void_p(full[k][0] + 32 * lo1), C.c_size_t(l1)))
                _lib.check(L.zk_device_memcpy(C.c_void_p(rv[k][1] + j * l1), C.c_void_p(full[k][1] + 32 * lo1), C.c_size_t(l1)))
                _lib.check(L.zk_device_memcpy(C.c_void_p(rv[k][2] + j * l2), C.c_void_p(full[k][2] + 32 * lo2), C.c_size_t(l2)))
        launch_fr()                 # next round
        for t in range(B):
            k, j = divmod(t, W)
            _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(slot0 + t), C.c_void_p(rv[k][0] + j * l1), C.c_void_p(rv[k][1] + j * l1), C.c_void_p(rv[k][2] + j * l2)))
    else:
        for t in range(B):
            _lib.check(L.zk_groth16_prove_partial_async(pr.handle, None, _p(rb), _p(sb), C.c_uint32(slot0 + t)))
    rounds_done += 1
    prev = pending
    pending = None
    if prev is not None:
        for t in range(B):
            _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(prev + t), _p(part)))
    pending = slot0
for mode in (True, False):
    for _ in range(3): group(mode)
    drain()
    _lib.check(L.zk_sync())
    t0 = time.perf_counter(); REPS = 5
    for _ in range(REPS): group(mode)
    drain()
    if fr_pending:
        for k in range(n_launched):
            _lib.check(L.zk_groth16_scalars_wait(pr.handle, C.c_uint32(2 * B + k)))
        fr_pending = False
    _lib.check(L.zk_sync())
    dt = (time.perf_counter() - t0) / REPS
    print("N=%d rank %d (%s key, slice %d G1 + %d G2 points) n=2^%d %s Fr stage: %.2f ms per round of %d proofs = %.2f ms/proof -> projected %.1f M constraints/s on %d GPUs"
          % (W, RANK, "derived Lagrange-form" if DERIVED else "tau-power", hi1 - lo1, hi2 - lo2, logn + (W.bit_length() - 1), "distributed" if mode else "replicated ", dt * 1e3, B, dt * 1e3 / B, B * n / dt / 1e6, W), flush=True)
