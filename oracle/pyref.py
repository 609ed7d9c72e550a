"""TEST INFRASTRUCTURE ONLY -- first-principles BLS12-381 arithmetic in Python big ints.

This file is part of the parity oracle (see oracle/README.md).  Nothing in the
product path (zukelang_amd/) may import it; only tests/, __graft_entry__.smoke()
and bench.py's cpu_baseline leg do, and only as a checker.

Purpose: an implementation-independent cross-check of oracle/zk_oracle.c (the
C restatement of the reference algorithms) and of the HIP kernels on small
cases.  Pure-Python loops: small sizes only.

Parity status: **parity unpinned** by the reference -- zukelang holds no golden
vector for any field element, point, key or proof.  What
pins this file: the public BLS12-381 constants, curve
membership / group-order checks below, the standard ZCash serialization KATs
(compressed G1 generator starts 0x97f1d3a7, G2 generator 0x93e02b60), and the
pairing equation `verify = true` which is the reference's own (only) acceptance
test (src/lib/test/test.ml:96,178).

Reference arithmetic lives in opam bls12-381 = 6.1.0 (a blst binding; not
vendored under /root/reference, zukelang.opam:15); call sites
src/lib/zk/curve.ml:77,123-140,159-220.
"""

P = 0x1A0111EA397FE69A4B1BA7B6434BACD764774B84F38512BF6730D2A0F6B0F6241EABFFFEB153FFFFB9FEFFFFFFFFAAAB
R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001
BLS_X = 0xD201000000010000  # |x|, x is negative

G1_X = 0x17F1D3A73197D7942695638C4FA9AC0FC3688C4F9774B905A14E3A3F171BAC586C55E83FF97A1AEFFB3AF00ADB22C6BB
G1_Y = 0x08B3F481E3AAA0F1A09E30ED741D8AE4FCF5E095D5D00AF600DB18CB2C04B3EDD03CC744A2888AE40CAA232946C5E7E1
G2_X0 = 0x024AA2B2F08F0A91260805272DC51051C6E47AD4FA403B02B4510B647AE3D1770BAC0326A805BBEFD48056C8C121BDB8
G2_X1 = 0x13E02B6052719F607DACD3A088274F65596BD0D09920B61AB5DA61BBDC7F5049334CF11213945D57E5AC7D055D042B7E
G2_Y0 = 0x0CE5D527727D6E118CC9CDC6DA2E351AADFD9BAA8CBDD3A76D429A695160D12C923AC9CC3BACA289E193548608B82801
G2_Y1 = 0x0606C4A02EA734CC32ACD2B02BC28B99CB3E287E85A763AF267492AB572E99AB3F370D275CEC1DA1AAA9075FF05F79BE

# FFT.ml:179-220: omega = 5^((r-1)/2^32), a primitive 2^32-th root of unity.
TWO_ADICITY = 32
OMEGA = pow(5, (R - 1) >> TWO_ADICITY, R)


def fr_inv(a):
    return pow(a % R, R - 2, R)


def fp_inv(a):
    return pow(a % P, P - 2, P)


# ---------------------------------------------------------------- Fp2 = Fp[u]/(u^2+1)
class Fp2:
    __slots__ = ("a", "b")

    def __init__(self, a, b=0):
        self.a = a % P
        self.b = b % P

    def __add__(self, o):
        return Fp2(self.a + o.a, self.b + o.b)

    def __sub__(self, o):
        return Fp2(self.a - o.a, self.b - o.b)

    def __neg__(self):
        return Fp2(-self.a, -self.b)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fp2(self.a * o, self.b * o)
        return Fp2(self.a * o.a - self.b * o.b, self.a * o.b + self.b * o.a)

    __rmul__ = __mul__

    def __eq__(self, o):
        if isinstance(o, int):
            o = Fp2(o)
        return self.a == o.a and self.b == o.b

    def __hash__(self):
        return hash((self.a, self.b))

    def inv(self):
        d = fp_inv(self.a * self.a + self.b * self.b)
        return Fp2(self.a * d, -self.b * d)

    def conj(self):
        return Fp2(self.a, -self.b)

    def is_zero(self):
        return self.a == 0 and self.b == 0

    def __repr__(self):
        return "Fp2(%#x, %#x)" % (self.a, self.b)


class Fp1:
    """Fp wrapped so the generic curve code can treat Fp and Fp2 alike."""
    __slots__ = ("a",)

    def __init__(self, a):
        self.a = a % P

    def __add__(self, o):
        return Fp1(self.a + o.a)

    def __sub__(self, o):
        return Fp1(self.a - o.a)

    def __neg__(self):
        return Fp1(-self.a)

    def __mul__(self, o):
        if isinstance(o, int):
            return Fp1(self.a * o)
        return Fp1(self.a * o.a)

    __rmul__ = __mul__

    def __eq__(self, o):
        if isinstance(o, int):
            return self.a == o % P
        return self.a == o.a

    def __hash__(self):
        return hash(self.a)

    def inv(self):
        return Fp1(fp_inv(self.a))

    def is_zero(self):
        return self.a == 0

    def __repr__(self):
        return "Fp(%#x)" % self.a


# ---------------------------------------------------------------- curves (affine, None = infinity)
B1 = Fp1(4)
B2 = Fp2(4, 4)
G1 = (Fp1(G1_X), Fp1(G1_Y))
G2 = (Fp2(G2_X0, G2_X1), Fp2(G2_Y0, G2_Y1))


def on_curve(pt, b):
    if pt is None:
        return True
    x, y = pt
    return y * y == x * x * x + b


def pt_neg(pt):
    if pt is None:
        return None
    return (pt[0], -pt[1])


def pt_add(p, q):
    if p is None:
        return q
    if q is None:
        return p
    x1, y1 = p
    x2, y2 = q
    if x1 == x2:
        if y1 == y2:
            if y1.is_zero():
                return None
            lam = (3 * (x1 * x1)) * (2 * y1).inv()
        else:
            return None
    else:
        lam = (y2 - y1) * (x2 - x1).inv()
    x3 = lam * lam - x1 - x2
    y3 = lam * (x1 - x3) - y1
    return (x3, y3)


def pt_mul(p, k):
    """k*p, plain double-and-add (k taken as a non-negative integer)."""
    acc = None
    add = p
    while k:
        if k & 1:
            acc = pt_add(acc, add)
        add = pt_add(add, add)
        k >>= 1
    return acc


def msm(points, scalars):
    acc = None
    for p, s in zip(points, scalars):
        acc = pt_add(acc, pt_mul(p, s % R))
    return acc


# ---------------------------------------------------------------- ZCash-style serialization
def _fp_be(a):
    return int(a).to_bytes(48, "big")


def g1_to_bytes(pt):
    """Uncompressed 96 B: x || y big-endian; infinity = 0x40 then zeros."""
    if pt is None:
        return bytes([0x40]) + bytes(95)
    return _fp_be(pt[0].a) + _fp_be(pt[1].a)


def g1_from_bytes(b):
    assert len(b) == 96
    if b[0] & 0x40:
        return None
    x = int.from_bytes(b[:48], "big")
    y = int.from_bytes(b[48:], "big")
    return (Fp1(x), Fp1(y))


def g1_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(47)
    x, y = pt
    out = bytearray(_fp_be(x.a))
    out[0] |= 0x80
    if y.a > (P - 1) // 2:
        out[0] |= 0x20
    return bytes(out)


def g2_to_bytes(pt):
    """Uncompressed 192 B: x1 || x0 || y1 || y0."""
    if pt is None:
        return bytes([0x40]) + bytes(191)
    x, y = pt
    return _fp_be(x.b) + _fp_be(x.a) + _fp_be(y.b) + _fp_be(y.a)


def g2_from_bytes(b):
    assert len(b) == 192
    if b[0] & 0x40:
        return None
    v = [int.from_bytes(b[i * 48:(i + 1) * 48], "big") for i in range(4)]
    return (Fp2(v[1], v[0]), Fp2(v[3], v[2]))


def g2_compress(pt):
    if pt is None:
        return bytes([0xC0]) + bytes(95)
    x, y = pt
    out = bytearray(_fp_be(x.b) + _fp_be(x.a))
    out[0] |= 0x80
    ny = -y
    # lexicographically larger: compare c1 first, then c0
    if (y.b, y.a) > (ny.b, ny.a):
        out[0] |= 0x20
    return bytes(out)


def fr_to_bytes(a):
    return int(a % R).to_bytes(32, "little")


def fr_from_bytes(b):
    return int.from_bytes(b, "little")


# ---------------------------------------------------------------- NTT (FFT.ml:29-67 convention)
def ntt(a, inverse=False):
    """out[k] = sum_j a[j] * w_N^(jk); inverse uses w^-1 and divides by N.
    Natural order in and out.  O(N^2) -- small N only."""
    n = len(a)
    assert n & (n - 1) == 0
    w = pow(OMEGA, (1 << TWO_ADICITY) // n, R)
    if inverse:
        w = fr_inv(w)
    out = []
    for k in range(n):
        wk = pow(w, k, R)
        acc = 0
        x = 1
        for j in range(n):
            acc = (acc + a[j] * x) % R
            x = x * wk % R
        out.append(acc)
    if inverse:
        ninv = fr_inv(n)
        out = [x * ninv % R for x in out]
    return out


# ---------------------------------------------------------------- polynomials over Fr (polynomial.ml semantics)
def poly_normalize(p):
    p = list(p)
    while p and p[-1] % R == 0:
        p.pop()
    return p


def poly_add(a, b):
    n = max(len(a), len(b))
    return poly_normalize([((a[i] if i < len(a) else 0) + (b[i] if i < len(b) else 0)) % R for i in range(n)])


def poly_mul(a, b):
    if not a or not b:
        return []
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R
    return poly_normalize(out)


def poly_divrem(a, b):
    a = poly_normalize(a)
    b = poly_normalize(b)
    assert b
    a = list(a)
    q = [0] * max(0, len(a) - len(b) + 1)
    binv = fr_inv(b[-1])
    for i in range(len(a) - len(b), -1, -1):
        d = a[i + len(b) - 1] * binv % R
        q[i] = d
        for j, y in enumerate(b):
            a[i + j] = (a[i + j] - d * y) % R
    return poly_normalize(q), poly_normalize(a)


def poly_eval(p, x):
    acc = 0
    for c in reversed(p):
        acc = (acc * x + c) % R
    return acc


def interpolate_int_domain(ys):
    """Coefficients of the unique poly of degree < n with f(i) = ys[i], i = 0..n-1
    (QAP.ml:81-86 evaluates over F.of_int rg)."""
    n = len(ys)
    total = []
    for j in range(n):
        basis = [1]
        den = 1
        for i in range(n):
            if i != j:
                basis = poly_mul(basis, [(-i) % R, 1])
                den = den * (j - i) % R
        c = ys[j] * fr_inv(den) % R
        total = poly_add(total, [x * c % R for x in basis])
    return total + [0] * (n - len(total))


def lagrange_basis(xs):
    """polynomial.ml:212-226 for ARBITRARY distinct points: l_j = prod_{i != j} (x - x_i) / (x_j - x_i), the factors multiplied in the reference's
    order (the points before x_j in reverse, then the ones after it: `List.rev_append sx xs`) -- the product is the same polynomial in any order."""
    out = []
    for j, xj in enumerate(xs):
        others = list(reversed(xs[:j])) + list(xs[j + 1:])
        acc = [1]
        for xi in others:
            d = (xj - xi) % R
            assert d != 0                                   # polynomial.ml:220
            dinv = fr_inv(d)
            acc = poly_mul(acc, [(-xi) * dinv % R, dinv])
        out.append(acc)
    return out


def interpolate(xys):
    """polynomial.ml:228-230: sum_j y_j l_j."""
    ls = lagrange_basis([x % R for x, _ in xys])
    total = []
    for (_, y), l in zip(xys, ls):
        total = poly_add(total, [c * (y % R) % R for c in l])
    return total


def z_poly(n):
    z = [1]
    for i in range(n):
        z = poly_mul(z, [(-i) % R, 1])
    return z


# ---------------------------------------------------------------- pairing (optimal ate), for `verify`
# Tower: Fp6 = Fp2[v]/(v^3 - (1+u)), Fp12 = Fp6[w]/(w^2 - v).
XI = Fp2(1, 1)


class Fp6:
    __slots__ = ("c0", "c1", "c2")

    def __init__(self, c0, c1, c2):
        self.c0, self.c1, self.c2 = c0, c1, c2

    def __add__(self, o):
        return Fp6(self.c0 + o.c0, self.c1 + o.c1, self.c2 + o.c2)

    def __sub__(self, o):
        return Fp6(self.c0 - o.c0, self.c1 - o.c1, self.c2 - o.c2)

    def __neg__(self):
        return Fp6(-self.c0, -self.c1, -self.c2)

    def __mul__(self, o):
        a0, a1, a2 = self.c0, self.c1, self.c2
        b0, b1, b2 = o.c0, o.c1, o.c2
        t0 = a0 * b0 + XI * (a1 * b2 + a2 * b1)
        t1 = a0 * b1 + a1 * b0 + XI * (a2 * b2)
        t2 = a0 * b2 + a1 * b1 + a2 * b0
        return Fp6(t0, t1, t2)

    def mul_by_v(self):
        return Fp6(XI * self.c2, self.c0, self.c1)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1 and self.c2 == o.c2

    def inv(self):
        a0, a1, a2 = self.c0, self.c1, self.c2
        t0 = a0 * a0 - XI * (a1 * a2)
        t1 = XI * (a2 * a2) - a0 * a1
        t2 = a1 * a1 - a0 * a2
        d = (a0 * t0 + XI * (a2 * t1) + XI * (a1 * t2)).inv()
        return Fp6(t0 * d, t1 * d, t2 * d)


FP6_ZERO = Fp6(Fp2(0), Fp2(0), Fp2(0))
FP6_ONE = Fp6(Fp2(1), Fp2(0), Fp2(0))


class Fp12:
    __slots__ = ("c0", "c1")

    def __init__(self, c0, c1):
        self.c0, self.c1 = c0, c1

    def __mul__(self, o):
        a0, a1, b0, b1 = self.c0, self.c1, o.c0, o.c1
        return Fp12(a0 * b0 + (a1 * b1).mul_by_v(), a0 * b1 + a1 * b0)

    def __eq__(self, o):
        return self.c0 == o.c0 and self.c1 == o.c1

    def inv(self):
        d = (self.c0 * self.c0 - (self.c1 * self.c1).mul_by_v()).inv()
        return Fp12(self.c0 * d, -(self.c1 * d))

    def conj(self):
        return Fp12(self.c0, -self.c1)

    def pow(self, e):
        acc = FP12_ONE
        base = self
        while e:
            if e & 1:
                acc = acc * base
            base = base * base
            e >>= 1
        return acc


FP12_ONE = Fp12(FP6_ONE, FP6_ZERO)


def _untwist(q):
    """Map a point of E'(Fp2) into E(Fp12): (x, y) -> (x / w^2, y / w^3)."""
    x, y = q
    # w^2 = v, w^3 = v*w.  1/w^2 = v^2 / xi ; 1/w^3 = w * v / xi ... build via generic inverse.
    w = Fp12(FP6_ZERO, FP6_ONE)
    w2 = w * w
    w3 = w2 * w
    X = Fp12(Fp6(x, Fp2(0), Fp2(0)), FP6_ZERO) * w2.inv()
    Y = Fp12(Fp6(y, Fp2(0), Fp2(0)), FP6_ZERO) * w3.inv()
    return X, Y


def _fp12_from_fp(a):
    return Fp12(Fp6(Fp2(a), Fp2(0), Fp2(0)), FP6_ZERO)


def _f12_add(a, b):
    return Fp12(a.c0 + b.c0, a.c1 + b.c1)


def _f12_sub(a, b):
    return Fp12(a.c0 - b.c0, a.c1 - b.c1)


def miller_loop(p, q):
    """Textbook Miller loop over E(Fp12) with the untwisted Q; slow but transparent."""
    if p is None or q is None:
        return FP12_ONE
    px, py = _fp12_from_fp(p[0].a), _fp12_from_fp(p[1].a)
    qx, qy = _untwist(q)
    tx, ty = qx, qy
    f = FP12_ONE
    three = _fp12_from_fp(3)
    two = _fp12_from_fp(2)
    bits = bin(BLS_X)[3:]
    for bit in bits:
        # doubling step
        lam = (three * tx * tx) * (two * ty).inv()
        line = _f12_sub(_f12_sub(py, ty), lam * _f12_sub(px, tx))
        f = f * f * line
        nx = _f12_sub(_f12_sub(lam * lam, tx), tx)
        ny = _f12_sub(lam * _f12_sub(tx, nx), ty)
        tx, ty = nx, ny
        if bit == "1":
            lam = _f12_sub(qy, ty) * _f12_sub(qx, tx).inv()
            line = _f12_sub(_f12_sub(py, ty), lam * _f12_sub(px, tx))
            f = f * line
            nx = _f12_sub(_f12_sub(lam * lam, tx), qx)
            ny = _f12_sub(lam * _f12_sub(tx, nx), ty)
            tx, ty = nx, ny
    # x is negative: f -> conj(f) (valid after final exponentiation)
    return f.conj()


def final_exp(f):
    return f.pow((P ** 12 - 1) // R)


def pairing(p, q):
    return final_exp(miller_loop(p, q))


def pairing_product_is_one(pairs):
    """prod e(P_i, Q_i) == 1 with a single final exponentiation."""
    f = FP12_ONE
    for p, q in pairs:
        f = f * miller_loop(p, q)
    return final_exp(f) == FP12_ONE


def splitmix64(state):
    """deterministic synthetic scalars."""
    state = (state + 0x9E3779B97F4A7C15) & 0xFFFFFFFFFFFFFFFF
    z = state
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & 0xFFFFFFFFFFFFFFFF
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & 0xFFFFFFFFFFFFFFFF
    return state, z ^ (z >> 31)


def fr_stream(seed):
    """Yield Fr elements: 8 successive splitmix64 outputs -> 512 bits -> mod r."""
    st = seed
    while True:
        v = 0
        for _ in range(8):
            st, o = splitmix64(st)
            v = (v << 64) | o
        yield v % R
