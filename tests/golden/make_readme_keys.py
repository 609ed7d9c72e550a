"""Generates tests/golden/readme_groth16_key.json: for the README circuit `x*x*x + x + 3` (README.md:49; the hand-derived compilation of
SURVEY.md 8c) and a FIXED toxic waste, the Groth16 proving key of groth16.ml:45-108 in the reference's declaration order, its Lagrange form
(scope row f4: what zk_groth16_pk_derive_lagrange must arrive at on the device, without tau), and the proof of groth16.ml:123-161 for x = 3 and
fixed (r, s) -- every point computed from FIRST PRINCIPLES: exponents as Python integers, points as affine double-and-add on them
(oracle/pyref.py), nothing from the C oracle, nothing from the GPU, nothing from the reference (it is OCaml and holds no vectors).
Run: python tests/golden/make_readme_keys.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402

R = P.R
inv = lambda a: pow(a, R - 2, R)


def lagrange_at(points, t):
    """l_i(t) over the given interpolation points"""
    out = []
    for i, xi in enumerate(points):
        num = den = 1
        for j, xj in enumerate(points):
            if j != i:
                num = num * (t - xj) % R
                den = den * (xi - xj) % R
        out.append(num * inv(den) % R)
    return out


def main():
    st = P.fr_stream(0x5EED0002)
    alpha, beta, gamma, delta, tau = (next(st) for _ in range(5))      # the order Groth16.setup draws them (groth16.ml:50-54)
    r, s = next(st), next(st)
    n, m = 3, 5
    mid = [0, 1, 1, 1, 0]                                               # ONE, c4, c5, input, v6
    # gates: c4 = input*input ; c5 = c4*input ; v6 = (c5 + input + 3 ONE) * (1 ONE): rows of L, R, O as {variable: coefficient}
    Lm = [{3: 1}, {1: 1}, {2: 1, 3: 1, 0: 3}]
    Rm = [{3: 1}, {3: 1}, {0: 1}]
    Om = [{1: 1}, {2: 1}, {4: 1}]
    x = 3
    w = [1, x * x, x ** 3, x, x ** 3 + x + 3]
    lag = lagrange_at(list(range(n)), tau)
    zt = 1
    for i in range(n):
        zt = zt * (tau - i) % R
    col = lambda M, k: sum(M[g].get(k, 0) * lag[g] for g in range(n)) % R      # u_k(tau) through the Lagrange basis of the gates' points
    vk, wk, yk = [col(Lm, k) for k in range(m)], [col(Rm, k) for k in range(m)], [col(Om, k) for k in range(m)]
    Lk = [(beta * vk[k] + alpha * wk[k] + yk[k]) % R for k in range(m)]
    dinv = inv(delta)
    ex1 = [alpha, delta, beta] + [pow(tau, k, R) for k in range(n + 2)] + [pow(tau, k, R) * zt % R * dinv % R for k in range(n - 1)] \
        + [Lk[k] * dinv % R for k in range(m) if mid[k]]
    ex2 = [beta, delta] + [pow(tau, k, R) for k in range(n + 2)]
    lam = lagrange_at(list(range(n, 2 * n - 1)), tau)                            # basis of the points n .. 2n-2
    lx1 = [alpha, delta, beta] + lag + [lam[t] * zt % R * dinv % R for t in range(n - 1)] + [Lk[k] * dinv % R for k in range(m) if mid[k]]
    lx2 = [beta, delta] + lag
    g1 = lambda e: P.g1_to_bytes(P.pt_mul(P.G1, e % R)).hex()
    g2 = lambda e: P.g2_to_bytes(P.pt_mul(P.G2, e % R)).hex()
    # the proof (groth16.ml:123-161) as exponents: A = alpha + v(tau) + r delta, B = beta + w(tau) + s delta,
    # C = (sum_mid w_k L_k(tau) + h(tau) Z(tau)) / delta + s A + r B - r s delta
    vt = sum(w[k] * vk[k] for k in range(m)) % R
    wt = sum(w[k] * wk[k] for k in range(m)) % R
    yt = sum(w[k] * yk[k] for k in range(m)) % R
    hz = (vt * wt - yt) % R                                                     # h(tau) Z(tau) = p(tau)
    ea = (alpha + vt + r * delta) % R
    eb = (beta + wt + s * delta) % R
    ec = ((sum(w[k] * Lk[k] for k in range(m) if mid[k]) + hz) * dinv + s * ea + r * eb - r * s * delta) % R
    out = {"how": "python tests/golden/make_readme_keys.py (first-principles Python big integers, oracle/pyref.py)",
           "toxic": {"alpha": hex(alpha), "beta": hex(beta), "gamma": hex(gamma), "delta": hex(delta), "tau": hex(tau)},
           "r": hex(r), "s": hex(s), "witness": [hex(v) for v in w], "mid": mid,
           "pk_g1": [g1(e) for e in ex1], "pk_g2": [g2(e) for e in ex2],
           "lagrange_g1": [g1(e) for e in lx1], "lagrange_g2": [g2(e) for e in lx2],
           "proof": {"a": g1(ea), "b": g2(eb), "c": g1(ec)}}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_groth16_key.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote readme_groth16_key.json:", len(ex1), "+", len(ex2), "key points,", len(lx1), "+", len(lx2), "Lagrange-form points")


if __name__ == "__main__":
    main()
