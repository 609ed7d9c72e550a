"""Generates tests/golden/readme_groth16_wire.hex: the README circuit's Groth16 proving key and proof of tests/golden/readme_groth16_key.json as
the JSON text the reference's yojson derivers write (groth16.ml:24-34 pkey, :110-114 proof; points as JSON strings of the RAW compressed bytes,
curve.ml:199,208 + ppx_deriving_yojson on `string`), produced here by an INDEPENDENT writer -- the record layout and Yojson's string escaping
(\\b \\t \\n \\f \\r \\" \\\\ named, other bytes below 0x20 and 0x7f as \\u00XX, everything else raw) spelled out below, points compressed by
oracle/pyref.py -- so that tests/test_wire.py can hold zukelang_amd/wire.py to exact bytes.  Two lines of hex: pkey JSON, proof JSON.
Run: python tests/golden/make_readme_wire.py"""
import json
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(os.path.dirname(HERE)))
from oracle import pyref as P  # noqa: E402

NAMED = {0x08: b"\\b", 0x09: b"\\t", 0x0A: b"\\n", 0x0C: b"\\f", 0x0D: b"\\r", 0x22: b'\\"', 0x5C: b"\\\\"}


def jstr(raw):
    out = bytearray(b'"')
    for c in raw:
        if c in NAMED:
            out += NAMED[c]
        elif c < 0x20 or c == 0x7F:
            out += b"\\u%04x" % c
        else:
            out.append(c)
    return bytes(out + b'"')


def main():
    fix = json.load(open(os.path.join(HERE, "readme_groth16_key.json")))
    c1 = lambda h: jstr(P.g1_compress(P.g1_from_bytes(bytes.fromhex(h))))
    c2 = lambda h: jstr(P.g2_compress(P.g2_from_bytes(bytes.fromhex(h))))
    g1, g2 = fix["pk_g1"], fix["pk_g2"]
    n = 3
    mids = [("c", 4), ("c", 5), ("input", 3)]                # Var.compare order of the intermediate variables (var.ml:42)
    lst = lambda xs: b"[" + b",".join(xs) + b"]"
    var = lambda name, vid: b"[" + jstr(name.encode()) + b"," + str(vid).encode() + b"]"
    o_ti, o_tz = 3, 3 + (n + 2)
    o_lt = o_tz + (n - 1)
    pkey = b"{" + b",".join([
        b'"a":' + c1(g1[0]), b'"d1":' + c1(g1[1]),
        b'"ti1":' + lst([c1(g1[o_ti + i]) for i in range(n + 2)]),
        b'"ltd_mid":' + lst([b"[" + var(nm, vid) + b"," + c1(g1[o_lt + i]) + b"]" for i, (nm, vid) in enumerate(mids)]),
        b'"tiztd":' + lst([c1(g1[o_tz + i]) for i in range(n - 1)]),
        b'"b1":' + c1(g1[2]), b'"b2":' + c2(g2[0]), b'"d2":' + c2(g2[1]),
        b'"ti2":' + lst([c2(g2[2 + i]) for i in range(n + 2)])]) + b"}"
    pr = fix["proof"]
    proof = b'{"a":' + c1(pr["a"]) + b',"b":' + c2(pr["b"]) + b',"c":' + c1(pr["c"]) + b"}"
    with open(os.path.join(HERE, "readme_groth16_wire.hex"), "w") as f:
        f.write(pkey.hex() + "\n" + proof.hex() + "\n")
    print("wrote readme_groth16_wire.hex:", len(pkey), "+", len(proof), "bytes of JSON")


if __name__ == "__main__":
    main()
