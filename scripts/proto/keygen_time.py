import os, sys, time
sys.path.insert(0, "/root/repo")
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16
L = _lib.lib(); _lib.check(L.zk_init(0))
for logn in (16, 19):
    n = 1 << logn
    t0 = time.perf_counter()
    cs, w = RC.iterated_cubic(n, next(RC.fr_stream(1)))
    t1 = time.perf_counter()
    st = RC.fr_stream(2); rng = lambda: next(st)
    pk, vk = Groth16.keygen(rng, cs)
    t2 = time.perf_counter()
    pr = Groth16(cs, pk, 0, 8 if logn == 19 else 1)
    t3 = time.perf_counter()
    print("2^%d: circuit %.1f s, keygen %.1f s, upload(+tables) %.1f s" % (logn, t1 - t0, t2 - t1, t3 - t2), flush=True)
    pr.close()
