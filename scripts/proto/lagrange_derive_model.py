"""Integer model of the derivation of Lagrange-form bases from a tau-power key IN THE EXPONENT (csrc/lagrange_derive.hip).

[l_i(tau)] for the QAP's points 0..n-1 (QAP.ml:84,92) are a linear image of the key's powers P_k = [tau^k]: with V_ik = i^k
(coefficients -> values), sum_k a_k P_k = sum_i y_i L_i for every polynomial forces  P = V^T L,  L = V^-T P.  The prover's own
Fr stage factors V^-1 = T . E  (values -> Newton coefficients E = Conv_alt . D(1/i!); Newton -> monomial T over the subproduct
tree), so  L = E^T T^T P = D(1/i!) . Conv_alt^T . T^T P:  the TRANSPOSED tree applied top-down (every node: upper half <- middle
product of the node with its left subproduct, lower half unchanged), one correlation with alt[j] = (-1)^j / j!, one scaling --
the same polynomial products as the prover's basis conversion, on group elements instead of field elements.
Here group elements are modelled by their discrete logs (integers mod r), so scalar * point = product mod r.
Run: python scripts/proto/lagrange_derive_model.py"""
import random

R = 0x73EDA753299D7D483339D80809A1D80553BDA402FFFE5BFEFFFFFFFF00000001


def poly_mul(a, b):
    out = [0] * (len(a) + len(b) - 1)
    for i, x in enumerate(a):
        for j, y in enumerate(b):
            out[i + j] = (out[i + j] + x * y) % R
    return out


def subproduct(points):
    p = [1]
    for s in points:
        p = poly_mul(p, [(-s) % R, 1])
    return p


def transposed_tree(G, offset):
    """T^T over the points offset .. offset + len(G) - 1, top level first; in place on a list of 'group elements'."""
    n2 = len(G)
    size = n2
    while size >= 2:
        h = size // 2
        for node in range(0, n2, size):
            P = subproduct(range(offset + node, offset + node + h))          # left subproduct, degree h, monic
            blk = G[node:node + size]
            hi = [sum(P[i - j] * blk[i] for i in range(j, j + h + 1) if i < size) % R for j in range(h)]
            G[node + h:node + size] = hi
        size //= 2
    return G


def derive(P, n, offset=0):
    """P_k = [x^k] for the points offset .. offset + n - 1 -> the Lagrange-basis 'points' L_i, i < n."""
    n2 = 1
    while n2 < n:
        n2 *= 2
    G = list(P[:n]) + [0] * (n2 - n)
    Y = transposed_tree(G, offset)
    Y = Y[:n] + [0] * (n2 - n)                                                  # truncate: the n-point problem is the leading block
    fact = [1] * (n2 + 1)
    for i in range(1, n2 + 1):
        fact[i] = fact[i - 1] * i % R
    alt = [pow(fact[j], -1, R) * (1 if j % 2 == 0 else R - 1) % R for j in range(n2)]
    corr = [sum(alt[k - i] * Y[k] for k in range(i, n2)) % R for i in range(n)]
    return [pow(fact[i], -1, R) * corr[i] % R for i in range(n)]


def lagrange_at(points, x):
    out = []
    for i, xi in enumerate(points):
        num = den = 1
        for j, xj in enumerate(points):
            if i != j:
                num = num * (x - xj) % R
                den = den * (xi - xj) % R
        out.append(num * pow(den, -1, R) % R)
    return out


def main():
    rnd = random.Random(5)
    for n in (1, 2, 3, 4, 5, 7, 8, 13, 16, 31, 32):
        tau = rnd.randrange(R)
        P = [pow(tau, k, R) for k in range(n)]
        assert derive(P, n) == lagrange_at(list(range(n)), tau), n
        # the h bases: points n .. 2n-2 (n - 1 of them) through the tree built on the SHIFTED leaves, scaled by a common factor (Z(tau)/delta)
        if n >= 2:
            zd = rnd.randrange(R)
            Q = [pow(tau, k, R) * zd % R for k in range(n - 1)]
            exp = [x * zd % R for x in lagrange_at(list(range(n, 2 * n - 1)), tau)]
            assert derive(Q, n - 1, offset=n) == exp, ("h", n)
    print("ok: transposed tree + correlation + scaling reproduce l_i(tau) and the shifted-domain h bases")


if __name__ == "__main__":
    main()
