"""Integer model of the lockstep modular inversion used by the batch-affine kernels (csrc/fp_inv.cuh).

Pornin, "Optimized Binary GCD for Modular Inversion" (2020), re-cut for 29-bit limbs: 27 outer iterations of 29 binary-GCD
steps on 64-bit approximations (low 29 bits + top 35 bits of a and b), the update factors applied to the 14-limb values a, b
(exact division by 2^29) and to u, v (signed, Montgomery-style division by 2^29 modulo p).  Every lane executes the same
instruction sequence: no data-dependent loop, no branch.  This model mirrors the kernel's arithmetic limb for limb and checks
the invariants and the bounds the kernel relies on; run: python scripts/proto/fp_inv_model.py"""
import random

P = 0x1a0111ea397fe69a4b1ba7b6434bacd764774b84f38512bf6730d2a0f6b0f6241eabfffeb153ffffb9feffffffffaaab
W, L = 29, 14
MASK = (1 << W) - 1
T = 27                      # outer iterations: 27 * 29 = 783 >= 2 * 381 - 1 = 761 steps
PINV = pow(P, -1, 1 << W)    # p^-1 mod 2^29


def limbs(x):
    return [(x >> (W * i)) & MASK for i in range(L)]


def approx(a, b):
    """64-bit approximations: exact when both fit 64 bits, else low 29 bits + the 35 bits below the top bit of max(a, b)."""
    al, bl = limbs(a), limbs(b)
    j = max(i for i in range(L) if (al[i] | bl[i]) or i == 0)
    top = al[j] | bl[j]
    ell = top.bit_length()
    if j < 2 or (j == 2 and ell <= 6):
        return a, b                                      # n <= 64: exact
    def ap(x):
        t = (x[j] << 35) | (x[j - 1] << 6) | (x[j - 2] >> 23)          # 64-bit window
        assert t < 1 << 64
        return ((t >> ell) << 29) | x[0]
    return ap(al), ap(bl)


def inner(xa, xb):
    f0, g0, f1, g1 = 1, 0, 0, 1
    for _ in range(W):
        odd = xa & 1
        swap = odd and xa < xb
        if swap:
            xa, xb, f0, g0, f1, g1 = xb, xa, f1, g1, f0, g0
        if odd:
            xa -= xb; f0 -= f1; g0 -= g1
        xa >>= 1
        f1 <<= 1; g1 <<= 1
        assert abs(f0) + abs(g0) <= 1 << W and abs(f1) + abs(g1) <= 1 << W
    return f0, g0, f1, g1


def inv_mont(X, check=None):
    """X = x * 2^406 mod p (any representative < 64 p is reduced first by the caller) -> x^-1 * 2^406 mod p."""
    a, b, u, v = X % P, P, 1, 0
    x = a
    maxuv = 0
    for t in range(T):
        xa, xb = approx(a, b)
        f0, g0, f1, g1 = inner(xa, xb)
        na, nb = f0 * a + g0 * b, f1 * a + g1 * b
        assert na % (1 << W) == 0 and nb % (1 << W) == 0
        na >>= W; nb >>= W
        if na < 0: na, f0, g0 = -na, -f0, -g0
        if nb < 0: nb, f1, g1 = -nb, -f1, -g1
        def upd(f, g):
            z = f * u + g * v
            q = (-(z & MASK) * PINV) & MASK
            z += q * P
            assert z & MASK == 0
            return z >> W
        u, v = upd(f0, g0), upd(f1, g1)
        a, b = na, nb
        maxuv = max(maxuv, abs(u), abs(v))
        assert (a << (W * 0)) >= 0 and b >= 0
        # invariant: a * 2^(29 (t+1)) ... with the division folded in: a = u x, b = v x (mod p)
        assert (u * x - a) % P == 0 and (v * x - b) % P == 0
    assert a == 0 and b == 1, (a, b)
    assert maxuv < 30 * P
    # v x = 1 (mod p): v = X^-1 = x^-1 R^-1; result x^-1 R = v R^2
    r = v * pow(2, 2 * 406, P) % P
    return r, maxuv


def main():
    rnd = random.Random(1)
    R = pow(2, 406, P)
    worst = 0
    cases = [1, 2, P - 1, P - 2, (P - 1) // 2, 3, 1 << 380, (1 << 381) - 1 - (1 << 381) % 1] + [rnd.randrange(1, P) for _ in range(3000)]
    for x in cases:
        x %= P
        if x == 0:
            continue
        X = x * R % P
        r, m = inv_mont(X)
        assert r == pow(x, -1, P) * R % P
        worst = max(worst, m)
    print("ok: %d cases; max |u|,|v| = %.2f p" % (len(cases), worst / P))
    print("final constant: v * C with C = R^2 mod p as a Montgomery operand -> fe_mul(v, R^3) ...")


if __name__ == "__main__":
    main()
