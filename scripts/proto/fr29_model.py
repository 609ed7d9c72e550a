#!/usr/bin/env python3
"""Integer model of zukelang_amd/csrc/fr29.cuh (9 x 29-bit limbs, R' = 2^261): checks the generated constants, the
quotient estimate of fr9_reduce_weak and the bound discipline of the NTT stages on random and extreme inputs."""
import random, re, os
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
W, L, M = 29, 9, (1 << 29) - 1
src = open(os.path.join(os.path.dirname(__file__), "..", "..", "zukelang_amd", "csrc", "fr29_consts.cuh")).read()
def arr(name):
    body = re.search(name + r"(?:\[\d+\])+ = \{(.*?)\};", src, re.S).group(1)
    return [int(x, 16) for x in re.findall(r"0x([0-9a-f]+)u", body)]
MOD, KR, C32 = arr("FR29_MOD"), arr("FR29_KR"), arr("FR29_C32")
QMUL = int(re.search(r"FR29_QMUL = (\d+)u", src).group(1))
KR = [KR[9 * k:9 * k + 9] for k in range(7)]
val = lambda l: sum(x << (W * i) for i, x in enumerate(l))
assert val(MOD) == R and all(val(KR[k]) == R << k for k in range(7))
assert sum(x << (32 * i) for i, x in enumerate(C32)) == 32 * (1 << 256) % R
U32 = 0xFFFFFFFF
def unpack(v):
    return [(v >> (W * i)) & M for i in range(L - 1)] + [v >> (W * (L - 1))]
def carry(t):
    c = [t[i] >> W for i in range(L - 1)]
    r = [t[0] & M] + [(t[i] & M) + c[i - 1] for i in range(1, L - 1)] + [t[L - 1] + c[L - 2]]
    assert all(x <= U32 for x in t) and all(x <= U32 for x in r)
    return r
def add(a, b): return carry([x + y for x, y in zip(a, b)])
def sub(a, b, ki):
    assert all(k >= y for k, y in zip(KR[ki], b)), "limb borrow"
    return carry([x + (k - y) for x, y, k in zip(a, b, KR[ki])])
def sub_raw(a, b, ki):
    r = [x + (k - y) for x, y, k in zip(a, b, KR[ki])]
    assert all(0 <= x <= U32 for x in r)
    return r
def mul(a, b):
    acc, m, r = 0, [], [0] * L
    for k in range(L):
        acc += sum(a[i] * b[k - i] for i in range(k + 1)) + sum(m[i] * MOD[k - i] for i in range(k))
        assert acc < 1 << 64
        m.append((-acc) & M)
        acc += m[k]
        assert acc & M == 0
        acc >>= W
    for k in range(L, 2 * L - 1):
        acc += sum(a[i] * b[k - i] for i in range(k - L + 1, L)) + sum(m[i] * MOD[k - i] for i in range(k - L + 1, L))
        assert acc < 1 << 64
        r[k - L] = acc & M
        acc >>= W
    assert acc <= U32
    r[L - 1] = acc
    return r
def reduce_weak(a):
    t = list(a)
    for i in range(L - 1):
        t[i + 1] += t[i] >> W
        t[i] &= M
    assert t[L - 1] <= U32
    q = (t[L - 1] * QMUL) >> 50
    assert t[L - 1] * QMUL < 1 << 64
    r, cy = [], 0
    for i in range(L):
        cur = t[i] - q * MOD[i] + cy
        r.append(cur & M if i < L - 1 else cur)
        cy = cur >> W
    assert r[L - 1] >= 0 and val(r) == val(a) - q * R
    return r
def canon(a):
    t = reduce_weak(a)
    assert val(t) < 2 * R, val(t) / R
    if val(t) >= R: t = unpack(val(t) - R)
    return val(t)
rnd = random.Random(1)
def lazy(bound):   # a lazily reduced representative below bound * r
    v = rnd.randrange(bound * R) if rnd.random() < 0.8 else bound * R - 1 - rnd.randrange(1 << rnd.randrange(1, 200))
    return unpack(v)
Rp = 1 << 261
for it in range(20000):
    A = rnd.choice([1, 2, 3, 8, 16, 48, 56, 64])
    a, b = lazy(A), lazy(64 // A)
    p = mul(a, b)
    assert val(p) < 2 * R and (val(p) * Rp - val(a) * val(b)) % R == 0
    x = lazy(rnd.choice([1, 2, 24, 42, 63, 64]))
    assert canon(x) == val(x) % R
    # extreme tops for the quotient estimate
    k = rnd.randrange(1, 64)
    for d in (-1, 0, 1):
        v = k * R + d * rnd.randrange(1, 1 << rnd.randrange(1, 240))
        if 0 <= v < 64 * R: assert canon(unpack(v)) == v % R
# DIF stage chain: bounds 3 -> 6 -> 12 -> 24 -> (reduce) 2
for it in range(3000):
    u, v, w = lazy(24), lazy(24), lazy(1)
    x = add(u, v); y = mul(sub(u, v, 5), w)
    assert mul(sub_raw(u, v, 5), w) == y          # no carry pass before a product by a factor with exact limbs
    assert val(x) < 48 * R and val(reduce_weak(x)) < 2 * R and val(y) < 2 * R
    assert (val(y) * Rp - (val(u) - val(v)) * val(w)) % R == 0
    # DIT: u below 38 r, v below 42 r
    u, v = lazy(38), lazy(42)
    t = mul(v, w); x = add(u, t); y = sub(u, t, 2)
    assert val(x) < 40 * R and val(y) < 42 * R and (val(y) - val(u) + val(t)) % R == 0
print("fr29 model ok")
