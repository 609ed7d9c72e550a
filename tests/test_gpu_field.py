"""GPU parity of the base-field layer (14 x 29-bit limbs, lazy reduction, value bounds in the type):
zk_selftest_fp evaluates 23 expressions per operand pair through the same code paths the group law uses;
the expected values are Python big-integer arithmetic mod p.  Edge operands: 0, 1, p-1, values whose limbs are
all-ones at the 29-bit boundaries, a = b, a + b = p."""
import ctypes as C
import random

import numpy as np
import pytest

from oracle import pyref as P
from zukelang_amd import _lib

pytestmark = pytest.mark.gpu
p = P.P
NOUT = 23


def le48(x):
    return int(x).to_bytes(48, "little")


def operands():
    rnd = random.Random(0xF1E1D)
    edge = [0, 1, 2, p - 1, p - 2, (p - 1) // 2, (p + 1) // 2, 1 << 380, (1 << 381) - 1 - ((1 << 381) - 1 - (p - 1)),
            sum(0x1FFFFFFF << (29 * i) for i in range(13)) % p, sum(0x10000000 << (29 * i) for i in range(14)) % p,
            (1 << 29) - 1, 1 << 29, (1 << 58) - 1, p - (1 << 29), 3 * ((p - 1) // 7)]
    a, b = [], []
    for x in edge:
        for y in (edge[0], edge[1], edge[3], x, (p - x) % p, rnd.randrange(p)):
            a.append(x); b.append(y)
    for _ in range(4096 - len(a)):
        a.append(rnd.randrange(p)); b.append(rnd.randrange(p))
    if len(a) & 1:
        a.append(5); b.append(7)
    return a, b


def test_field_battery_matches_big_integers():
    a, b = operands()
    n = len(a)
    out = np.zeros(NOUT * 48 * n, dtype=np.uint8)
    _lib.check(_lib.lib().zk_selftest_fp(b"".join(le48(x) for x in a), b"".join(le48(x) for x in b), C.c_size_t(n),
                                         out.ctypes.data_as(C.POINTER(C.c_uint8))))
    raw = out.tobytes()
    get = lambda i, k: int.from_bytes(raw[48 * (NOUT * i + k):48 * (NOUT * i + k + 1)], "little")
    inv = lambda x: pow(x, p - 2, p)
    for i in range(n):
        x, y = a[i], b[i]
        d1 = (-x - 2 * y) % p
        exp = [(x + y) % p, (x - y) % p, x * y % p, x * x % p, (-x) % p, inv(x), 8 * x % p, d1, d1 * (y - x) % p,
               (x * y - (x + y) * (x - y)) % p]
        for k, e in enumerate(exp):
            assert get(i, k) == e, (i, k, hex(x), hex(y))
        flags = 1 | 2 | (4 if x == 0 else 0) | 8 | (16 if x == y else 0) | 32
        assert get(i, 10) == flags, (i, hex(x), hex(y), get(i, 10))
        assert get(i, 11) == 0
        assert get(i, 15) == x * y % p
        # lane pairs: (a_even + a_odd u) and (b_even + b_odd u) in Fp2 = Fp[u]/(u^2 + 1)
        e = i & ~1
        x0, x1, y0, y1 = a[e], a[e + 1], b[e], b[e + 1]
        mul = ((x0 * y0 - x1 * y1) % p, (x0 * y1 + x1 * y0) % p)
        sqr = ((x0 * x0 - x1 * x1) % p, 2 * x0 * x1 % p)
        s0, s1, t0, t1 = (x0 - y0) % p, (x1 - y1) % p, (x0 + y0) % p, (x1 + y1) % p
        dif = ((s0 * t0 - s1 * t1) % p, (s0 * t1 + s1 * t0) % p)
        c = i & 1
        assert (get(i, 12), get(i, 13), get(i, 14)) == (mul[c], sqr[c], dif[c]), (i, "fp2 lane pair")
        # lockstep inversion (csrc/fp_inv.cuh): base field, and Fp2 on the lane pair: 1 / (x0 + x1 u) = (x0 - x1 u) / (x0^2 + x1^2)
        assert get(i, 16) == inv(x), (i, "fe_inv_fast", hex(x))
        nrm = inv((x0 * x0 + x1 * x1) % p)
        assert get(i, 17) == ((x0 * nrm) % p, (-x1 * nrm) % p)[c], (i, "fp2 fe_inv_fast")
        # round 3: the fused a - b - 2 c (one carry pass), the lazily negated factor of the fused double product, the lazy negation inside the lane-pair product
        big = (16 * x - y) % p
        assert get(i, 18) == (x * x - x * y - 2 * y * (x + y)) % p, (i, "fe_sub_sub_dbl on products", hex(x), hex(y))
        assert get(i, 19) == (x - 3 * big) % p, (i, "fe_sub_sub_dbl at the at-rest bound", hex(x), hex(y))
        assert get(i, 20) == (x * y - big * y) % p, (i, "fe_mul_sub with a lazily negated factor", hex(x), hex(y))
        b0, b1 = (16 * x0 - y0) % p, (16 * x1 - y1) % p
        assert get(i, 21) == ((b0 * y0 - b1 * y1) % p, (b0 * y1 + b1 * y0) % p)[c], (i, "fp2 lane-pair product, lazily reduced left operand")
        assert get(i, 22) == ((b0 * b0 - b1 * b1) % p, 2 * b0 * b1 % p)[c], (i, "fp2 lane-pair square with the lazy difference")
