"""How long does the host take to ENQUEUE one proof (prove_async returns before the GPU finishes)?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16
L = _lib.lib(); _lib.check(L.zk_init(0))
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << logn
cs, w = RC.iterated_cubic(n, next(RC.fr_stream(1)))
st = RC.fr_stream(2); rng = lambda: next(st)
pk, _ = Groth16.keygen(rng, cs)
pr = Groth16(cs, pk); pr.set_witness(w); pr.reserve_slots(8)
for depth in (1, 2, 4, 6, 8):
    for rep in range(2):
        _lib.check(L.zk_sync())
        N = 24
        enq = 0.0
        t0 = time.perf_counter()
        for i in range(N):
            if i >= depth: pr.prove_wait(i % depth)
            a = time.perf_counter()
            pr.prove_async(None, rng(), rng(), i % depth)
            enq += time.perf_counter() - a
        for i in range(max(0, N - depth), N): pr.prove_wait(i % depth)
        _lib.check(L.zk_sync())
        dt = time.perf_counter() - t0
    print("depth %d: %.3f ms/proof, host enqueue %.3f ms/proof" % (depth, dt / N * 1e3, enq / N * 1e3), flush=True)
