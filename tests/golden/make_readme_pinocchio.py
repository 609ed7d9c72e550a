"""Generates tests/golden/readme_pinocchio_key.json: for the README circuit `x*x*x + x + 3` and a FIXED toxic waste, the Pinocchio Protocol-2
proving key (KeyGen.generate, pinocchio.ml:77-189, fields flattened in the layout of include/zkmi355x.h) and the zero-knowledge proof of
ZKCompute.f (pinocchio.ml:427-514) for x = 3 with fixed dv, dw, dy -- every point from FIRST PRINCIPLES: exponents as Python integers,
points by affine double-and-add on them (oracle/pyref.py); nothing from the C oracle, the GPU or the reference.
Run: python tests/golden/make_readme_pinocchio.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402

R = P.R
inv = lambda a: pow(a, R - 2, R)


def main():
    st = P.fr_stream(0x5EED0003)
    rv, rw, s, av, aw, ay, b, gm = (next(st) for _ in range(8))          # the order KeyGen.generate draws them (pinocchio.ml:83-91)
    dv, dw, dy = (next(st) for _ in range(3))
    ry = rv * rw % R
    n, m = 3, 5
    mid = [0, 1, 1, 1, 0]
    Lm = [{3: 1}, {1: 1}, {2: 1, 3: 1, 0: 3}]
    Rm = [{3: 1}, {3: 1}, {0: 1}]
    Om = [{1: 1}, {2: 1}, {4: 1}]
    x = 3
    c = [1, x * x, x ** 3, x, x ** 3 + x + 3]
    lag = []
    for i in range(n):
        num = den = 1
        for j in range(n):
            if j != i:
                num = num * (s - j) % R
                den = den * (i - j) % R
        lag.append(num * inv(den) % R)
    t = 1
    for i in range(n):
        t = t * (s - i) % R
    col = lambda M, k: sum(M[g].get(k, 0) * lag[g] for g in range(n)) % R
    vk, wk, yk = [col(Lm, k) for k in range(m)], [col(Rm, k) for k in range(m)], [col(Om, k) for k in range(m)]
    mids = [k for k in range(m) if mid[k]]
    ex1 = [rv * vk[k] % R for k in mids] + [ry * yk[k] % R for k in mids] + [rv * av * vk[k] % R for k in mids] + [ry * ay * yk[k] % R for k in mids] \
        + [b * (rv * vk[k] + rw * wk[k] + ry * yk[k]) % R for k in mids] + [pow(s, i, R) for i in range(n + 1)] + vk + wk \
        + [rv * t % R, ry * t % R, rv * t * av % R, ry * t * ay % R, rv * t * b % R, rw * t * b % R, ry * t * b % R]
    ex2 = [rw * wk[k] % R for k in mids] + [rw * aw * wk[k] % R for k in mids] + [pow(s, i, R) for i in range(n + 1)] + [rw * t % R, rw * t * aw % R]
    # the proof as exponents (vv | ww | yy | h | vavv | waww | yayy | bvwy)
    vm = sum(c[k] * vk[k] for k in mids) % R
    wm = sum(c[k] * wk[k] for k in mids) % R
    ym = sum(c[k] * yk[k] for k in mids) % R
    va = sum(c[k] * vk[k] for k in range(m)) % R
    wa = sum(c[k] * wk[k] for k in range(m)) % R
    ya = sum(c[k] * yk[k] for k in range(m)) % R
    h = (va * wa - ya) * inv(t) % R                                       # exact: the witness satisfies the circuit
    e_vv = rv * (vm + dv * t) % R
    e_ww = rw * (wm + dw * t) % R
    e_yy = ry * (ym + dy * t) % R
    e_h = (h + dw * va + dv * wa + dv * dw * t - dy) % R
    e_bv = (b * (rv * vm + rw * wm + ry * ym) + b * t * (rv * dv + rw * dw + ry * dy)) % R
    g1 = lambda e: P.g1_to_bytes(P.pt_mul(P.G1, e % R)).hex()
    g2 = lambda e: P.g2_to_bytes(P.pt_mul(P.G2, e % R)).hex()
    proof = g1(e_vv) + g2(e_ww) + g1(e_yy) + g1(e_h) + g1(av * e_vv) + g2(aw * e_ww) + g1(ay * e_yy) + g1(e_bv)
    # the h pool after zk_pinocchio_pk_derive_lagrange: [lambda_t(s)] (n-1) | [Z(s)] | [1] | ..., lambda_t over the points n .. 2n-2
    pts = list(range(n, 2 * n - 1))
    lam = []
    for i, xi in enumerate(pts):
        num = den = 1
        for j, xj in enumerate(pts):
            if j != i:
                num = num * (s - xj) % R
                den = den * (xi - xj) % R
        lam.append(num * inv(den) % R)
    derived_h_pool_full = lam + [t, 1] + vk + wk                       # a key whose v_all / w_all fail the upload's consistency check, or ZK_PIN_COMPACT_H=0
    # the COMPACT h pool (csrc/pinocchio.hip, the default): v_all | w_all leave the product -- dw [v(s)] + dv [w(s)] ride on the bases of h plus [s^(n-1)]
    derived_h_pool = lam + [t, 1, pow(s, n - 1, R)]
    # the compact pool's scalar vector gives the same exponent as e_h: h(n+t) + dw (v - kv X^(n-1))(n+t) + dv (w - kw X^(n-1))(n+t) | dv dw | -dy | dw kv + dv kw
    def interp_eval(vals, x):                                            # the interpolant of (i, vals[i]), i < n, at x
        acc = 0
        for i in range(n):
            num = den = 1
            for j in range(n):
                if j != i:
                    num = num * (x - j) % R
                    den = den * (i - j) % R
            acc = (acc + vals[i] * num * inv(den)) % R
        return acc
    row = lambda M: [sum(M[g].get(k, 0) * c[k] for k in range(m)) % R for g in range(n)]
    a_vals, b_vals, c_vals = row(Lm), row(Rm), row(Om)
    lead = lambda vals: sum(vals[i] * inv(__import__("math").prod((i - j) for j in range(n) if j != i) % R) for i in range(n)) % R
    kv, kw = lead(a_vals), lead(b_vals)
    zat = lambda x: __import__("math").prod((x - i) for i in range(n)) % R
    hv = [((interp_eval(a_vals, x) * interp_eval(b_vals, x) - interp_eval(c_vals, x)) * inv(zat(x))) % R for x in pts]
    sc = [(hv[i] + dw * (interp_eval(a_vals, x) - kv * pow(x, n - 1, R)) + dv * (interp_eval(b_vals, x) - kw * pow(x, n - 1, R))) % R for i, x in enumerate(pts)]
    sc += [dv * dw % R, -dy % R, (dw * kv + dv * kw) % R]
    assert sum(a * b for a, b in zip(sc, derived_h_pool)) % R == e_h, "compact h pool: scalar vector does not reproduce h'"
    # verification key (pinocchio.ml:62-75) flattened as include/zkmi355x.h lays it out, and the NonZK proof (Compute.f, :210-248: no blinding)
    ios = [k for k in range(m) if not mid[k]]
    vx1 = [1, aw, gm * b % R] + [rv * vk[k] % R for k in ios] + [ry * yk[k] % R for k in ios]
    vx2 = [1, av, ay, gm, gm * b % R, ry * t % R] + [rw * wk[k] % R for k in ios]
    n_vv, n_ww, n_yy = rv * vm % R, rw * wm % R, ry * ym % R
    proof0 = g1(n_vv) + g2(n_ww) + g1(n_yy) + g1(h) + g1(av * n_vv) + g2(aw * n_ww) + g1(ay * n_yy) + g1(b * (n_vv + n_ww + n_yy))
    out = {"how": "python tests/golden/make_readme_pinocchio.py (first-principles Python big integers, oracle/pyref.py)",
           "toxic": [hex(v) for v in (rv, rw, s, av, aw, ay, b, gm)], "deltas": [hex(v) for v in (dv, dw, dy)],
           "witness": [hex(v) for v in c], "mid": mid, "pk_g1": [g1(e) for e in ex1], "pk_g2": [g2(e) for e in ex2],
           "derived_h_pool_g1": [g1(e) for e in derived_h_pool], "derived_h_pool_g1_full": [g1(e) for e in derived_h_pool_full], "proof": proof,
           "pk_exponents_g1": [hex(e % R) for e in ex1], "pk_exponents_g2": [hex(e % R) for e in ex2],
           "vk_exponents_g1": [hex(e % R) for e in vx1], "vk_exponents_g2": [hex(e % R) for e in vx2],
           "vk_g1": [g1(e) for e in vx1], "vk_g2": [g2(e) for e in vx2], "io": [hex(c[k]) for k in ios], "proof_nonzk": proof0}
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_pinocchio_key.json"), "w") as f:
        json.dump(out, f, indent=1)
    print("wrote readme_pinocchio_key.json:", len(ex1), "+", len(ex2), "key points")


if __name__ == "__main__":
    main()
