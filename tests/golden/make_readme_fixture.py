"""Generates tests/golden/readme_circuit.json from first principles (oracle/pyref.py, Python
big ints / exact rational-free field arithmetic) -- NOT from the reference (no OCaml toolchain
exists here).  The circuit is the hand-derived compilation of `x*x*x + x + 3`
(README.md:49; src/lib/zk/comp.ml:233-244,448-473 read as text).  Run: python tests/golden/make_readme_fixture.py"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import pyref as P  # noqa: E402


def main():
    out = {"modulus_r": hex(P.R), "cases": []}
    for x in (3, 2):
        w = [1, x * x, x ** 3, x, x ** 3 + x + 3]          # ONE, c4, c5, input, v6
        v_ev = [w[3], w[1], (w[2] + w[3] + 3) % P.R]       # left operands at X = 0,1,2
        w_ev = [w[3], w[3], 1]
        y_ev = [w[1], w[2], w[4]]
        v = P.interpolate_int_domain(v_ev)
        ww = P.interpolate_int_domain(w_ev)
        y = P.interpolate_int_domain(y_ev)
        p = P.poly_add(P.poly_mul(v, ww), [(-c) % P.R for c in y])
        h, rem = P.poly_divrem(p, P.z_poly(3))
        assert rem == []
        out["cases"].append({"x": x, "witness": [hex(i) for i in w], "v": [hex(i) for i in v],
                             "w": [hex(i) for i in ww], "y": [hex(i) for i in y],
                             "p": [hex(i) for i in p], "h": [hex(i) for i in h]})
    out["z"] = [hex(i) for i in P.z_poly(3)]
    out["g1_generator_compressed"] = P.g1_compress(P.G1).hex()
    out["g2_generator_compressed"] = P.g2_compress(P.G2).hex()
    out["omega_2_32"] = hex(P.OMEGA)
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "readme_circuit.json"), "w") as f:
        json.dump(out, f, indent=1)


if __name__ == "__main__":
    main()
