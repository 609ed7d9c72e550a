import os, sys, time
sys.path.insert(0, "/root/repo")
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16
L = _lib.lib(); _lib.check(L.zk_init(0))
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 16
depths = [int(x) for x in sys.argv[2].split(",")]
n = 1 << logn
cs, w = RC.iterated_cubic(n, next(RC.fr_stream(1)))
st = RC.fr_stream(2); rng = lambda: next(st)
pk, _ = Groth16.keygen(rng, cs)
pr = Groth16(cs, pk); pr.set_witness(w); pr.reserve_slots(max(depths))
def run(depth, N):
    _lib.check(L.zk_sync())
    t0 = time.perf_counter()
    for i in range(N):
        if i >= depth: pr.prove_wait(i % depth)
        pr.prove_async(None, rng(), rng(), i % depth)
    for i in range(max(0, N - depth), N): pr.prove_wait(i % depth)
    _lib.check(L.zk_sync())
    return (time.perf_counter() - t0) / N * 1e3
for d in depths:
    run(d, 2 * d)
    print("streams/slot=%s depth %2d: %.3f ms/proof" % ("1" if os.environ.get("ZK_SERIAL_STREAMS") else "3", d, run(d, max(48, 4 * d))), flush=True)
