"""Does the bucket accumulation cost more per addition on a larger pool?  One resident-path MSM (window tables, one bucket set of 2^15) per
pool size, the accumulate family timed by the library's HIP events (profiling level 2).  usage: python scripts/proto/acc_size_probe.py"""
import ctypes as C, os, sys
sys.path.insert(0, "/root/repo")
os.environ["ZK_MSM_API_PRECOMP"] = "1"
import numpy as np
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.curve import G1
L = _lib.lib(); _lib.check(L.zk_init(0))
for n in (1 << 16, 3 << 16, 1 << 18, 3 << 18, 1 << 20):
    ks = RC.random_fr_bytes(n, 7)
    pts = G1.of_Fr(RC.random_fr_bytes(n, 11))          # n distinct points
    G1.apply_powers(ks, pts, 16)
    _lib.check(L.zk_profile_reset()); _lib.check(L.zk_profile_enable(2))
    for _ in range(3): G1.apply_powers(ks, pts, 16)
    _lib.check(L.zk_profile_enable(0))
    ms, cnt = C.c_double(), C.c_uint64()
    _lib.check(L.zk_profile_get(b"msm_accumulate_g1", C.byref(ms), C.byref(cnt)))
    per = ms.value / 3
    print("n = %8d  accumulate %.3f ms  %.4f ns per (point, window) addition  (%.2f G additions/s)" % (n, per, per * 1e6 / (n * 16), n * 16 / per / 1e6), flush=True)
