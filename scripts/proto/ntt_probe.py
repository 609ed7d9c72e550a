import os, sys, time
sys.path.insert(0, "/root/repo")
import numpy as np
from zukelang_amd import _lib
import ctypes as C
L = _lib.lib(); _lib.check(L.zk_init(0))
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 21
n = 1 << logn
rng = np.random.default_rng(1)
a = rng.integers(0, 256, size=(n, 32), dtype=np.uint8); a[:, 31] &= 0x3f
buf = a.copy()
p = buf.ctypes.data_as(C.POINTER(C.c_uint8))
for rep in range(3):
    t0 = time.perf_counter()
    _lib.check(L.zk_fr_ntt(p, logn, 0))
    _lib.check(L.zk_fr_ntt(p, logn, 1))
    print("fwd+inv incl. PCIe: %.1f ms" % ((time.perf_counter() - t0) * 1e3))
print("roundtrip ok:", bool((buf == a).all()))
