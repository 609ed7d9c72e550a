"""Scratch GPU probe: field-multiplier throughput (ALU ceiling for every kernel on the path)."""
import ctypes as C
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from zukelang_amd import _lib
L = _lib.lib()
_lib.check(L.zk_init(0))
for kind, name in ((0, "Fr"), (1, "Fp")):
    g = C.c_double()
    _lib.check(L.zk_bench_field_mul(kind, 2000, C.byref(g)))
    print("%s Montgomery mul: %.1f G mul/s" % (name, g.value))
    _lib.check(L.zk_bench_field_mul(kind | 4, 2000, C.byref(g)))
    print("%s single-wave dependent chain: %.2f us per mul" % (name, 64 / (g.value * 1e9) * 1e6))
