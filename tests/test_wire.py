"""Wire format (scope row f3): byte-level JSON strings as Yojson writes them, compressed points in and out
(the decompression is host code: no GPU), and the record layouts of groth16.ml:24-43,110-114."""
import numpy as np

from oracle import pyref as P
from zukelang_amd import wire
from zukelang_amd.groth16 import Proof, VKey


def test_json_strings_are_bytes_with_yojson_escapes():
    raw = bytes(range(256))
    s = wire.json_bytes_string(raw)
    assert s[:1] == b'"' and s[-1:] == b'"'
    assert b"\\u0000" in s and b"\\b" in s and b"\\t" in s and b"\\n" in s and b"\\f" in s and b"\\r" in s
    assert b'\\"' in s and b"\\\\" in s and b"\\u007f" in s
    assert bytes([0x80, 0x81]) in s and bytes([0xFF]) in s            # high bytes travel raw, not as UTF-8
    assert wire.loads(s) == raw
    v = {"a": raw, "l": [1, -2, [b"x", 3]], "e": {}, "z": []}
    assert wire.loads(wire.dumps(v)) == {"a": raw, "l": [1, -2, [b"x", 3]], "e": {}, "z": []}
    assert wire.fr_of_json(wire.fr_to_json(P.R - 1)) == P.R - 1 and wire.fr_to_json(5) == b'"5"'


def test_decompression_inverts_compression_for_both_groups():
    for k in (1, 2, 0xDEADBEEF, P.R - 1):
        p1, p2 = P.pt_mul(P.G1, k), P.pt_mul(P.G2, k + 7)
        assert wire.g1_of_json(P.g1_compress(p1)) == P.g1_to_bytes(p1)
        assert wire.g2_of_json(P.g2_compress(p2)) == P.g2_to_bytes(p2)
        assert wire.g1_of_json(P.g1_compress(P.pt_neg(p1))) == P.g1_to_bytes(P.pt_neg(p1))
        assert wire.g2_of_json(P.g2_compress(P.pt_neg(p2))) == P.g2_to_bytes(P.pt_neg(p2))
    assert wire.g1_of_json(P.g1_compress(None)) == P.g1_to_bytes(None)
    assert wire.g2_of_json(P.g2_compress(None)) == P.g2_to_bytes(None)


def test_groth16_records_round_trip():
    pt1 = lambda k: P.g1_to_bytes(P.pt_mul(P.G1, k))
    pt2 = lambda k: P.g2_to_bytes(P.pt_mul(P.G2, k))
    proof = Proof(pt1(3), pt2(4), pt1(5))
    js = wire.groth16_proof_to_json(proof)
    assert js.startswith(b'{"a":"') and b'","b":"' in js and b'","c":"' in js
    back = wire.groth16_proof_of_json(js)
    assert (back.a, back.b, back.c) == (proof.a, proof.b, proof.c)
    io_vars = [("ONE", 1), ("v", 6)]
    vk = VKey(pt1(1), np.frombuffer(pt1(8) + pt1(9), dtype=np.uint8), pt2(1), pt2(10), pt2(11), bytes(range(64)) * 9)
    vk2, vars2 = wire.groth16_vkey_of_json(wire.groth16_vkey_to_json(vk, io_vars))
    assert vars2 == io_vars and bytes(vk2.ltgm_io) == bytes(vk.ltgm_io) and (vk2.one1, vk2.one2, vk2.gm, vk2.d, vk2.ab) == (vk.one1, vk.one2, vk.gm, vk.d, vk.ab)
    n, mids = 3, [("input", 3), ("c", 4), ("c", 5)]
    from zukelang_amd.groth16 import PKey
    g1 = b"".join(pt1(20 + i) for i in range(3 + (n + 2) + (n - 1) + len(mids)))
    g2 = b"".join(pt2(50 + i) for i in range(2 + n + 2))
    pk = PKey(np.frombuffer(g1, dtype=np.uint8), np.frombuffer(g2, dtype=np.uint8))
    js = wire.groth16_pkey_to_json(pk, n, mids)
    keys = [k for k in wire.loads(js)]
    assert keys == ["a", "d1", "ti1", "ltd_mid", "tiztd", "b1", "b2", "d2", "ti2"]          # groth16.ml:24-34 declaration order
    pk2, mids2 = wire.groth16_pkey_of_json(js)
    assert mids2 == mids and bytes(pk2.g1) == g1 and bytes(pk2.g2) == g2


def test_names_are_byte_strings_and_malformed_json_raises_valueerror():
    """A variable name is an OCaml string = raw bytes: a byte >= 0x80 travels as ONE raw byte through every reader and writer (latin-1 on the
    Python side), never as a UTF-8 pair; malformed input raises ValueError, not an assert / IndexError."""
    import pytest
    pt1 = lambda k: P.g1_to_bytes(P.pt_mul(P.G1, k))
    pt2 = lambda k: P.g2_to_bytes(P.pt_mul(P.G2, k))
    io_vars = [("ONE", 1), ("caf\xe9", 6)]                       # 0xE9 as a single byte
    vk = VKey(pt1(1), np.frombuffer(pt1(8) + pt1(9), dtype=np.uint8), pt2(1), pt2(10), pt2(11), bytes(range(64)) * 9)
    js = wire.groth16_vkey_to_json(vk, io_vars)
    assert b'"caf\xe9"' in js and b"\xc3\xa9" not in js          # the raw byte, not its UTF-8 encoding
    vk2, vars2 = wire.groth16_vkey_of_json(js)
    assert vars2 == io_vars and wire.groth16_vkey_to_json(vk2, vars2) == js
    assert wire.loads(wire.dumps({"k\xff": 1})) == {"k\xff": 1}
    for bad in (b"", b"{", b'{"a" 1}', b'{"a":1', b"[1 2]", b'"abc', b'{"a":1} x', b"[,]", b'{"a":"\\q"}'):
        with pytest.raises(ValueError):
            wire.loads(bad)
    for reader in (wire.groth16_vkey_of_json, wire.groth16_pkey_of_json, wire.groth16_proof_of_json, wire.pinocchio_pkey_of_json, wire.pinocchio_vkey_of_json):
        with pytest.raises(ValueError):
            reader(b'{"one1":1}')                               # valid JSON, not the record


def test_pinocchio_key_records_round_trip():
    """pinocchio.ml:37-60 (pkey) and :62-75 (vkey): field order of the derivers, the I_mid / [m] / io maps as Var.Map
    bindings, flat pools of include/zkmi355x.h on the other side."""
    from zukelang_amd.pinocchio import PKey, VKey as PVKey
    pt1 = lambda k: P.g1_to_bytes(P.pt_mul(P.G1, k))
    pt2 = lambda k: P.g2_to_bytes(P.pt_mul(P.G2, k))
    n = 3
    all_vars = [("ONE", 1), ("c", 4), ("c", 5), ("input", 3), ("v", 6)]
    mids, ios = all_vars[1:4], [all_vars[0], all_vars[4]]
    k, m = len(mids), len(all_vars)
    g1 = b"".join(pt1(100 + i) for i in range(5 * k + (n + 1) + 2 * m + 7))
    g2 = b"".join(pt2(300 + i) for i in range(2 * k + (n + 1) + 2))
    pk = PKey(np.frombuffer(g1, dtype=np.uint8), np.frombuffer(g2, dtype=np.uint8))
    js = wire.pinocchio_pkey_to_json(pk, n, mids, all_vars)
    assert list(wire.loads(js)) == ["vv", "ww", "yy", "vav", "waw", "yay", "si", "bvwy", "si2", "vt", "wt", "yt", "vavt", "wawt", "yayt",
                                    "vbt", "wbt", "ybt", "v_all", "w_all"]
    pk2, n2, mids2, all2 = wire.pinocchio_pkey_of_json(js)
    assert (n2, mids2, all2) == (n, mids, all_vars) and bytes(pk2.g1) == g1 and bytes(pk2.g2) == g2
    d = wire.loads(js)
    assert d["yy"][1][0] == [b"c", 5] and wire.g1_of_json(d["yy"][1][1]) == g1[96 * (k + 1):96 * (k + 2)]      # yy is the 2nd G1 pool
    assert wire.g2_of_json(d["wawt"]) == g2[-192:] and wire.g1_of_json(d["ybt"]) == g1[-96:]
    v1 = b"".join(pt1(500 + i) for i in range(3 + 2 * len(ios)))
    v2 = b"".join(pt2(600 + i) for i in range(6 + len(ios)))
    vk = PVKey(np.frombuffer(v1, dtype=np.uint8), np.frombuffer(v2, dtype=np.uint8))
    js = wire.pinocchio_vkey_to_json(vk, ios)
    assert list(wire.loads(js)) == ["one", "one2", "av", "aw", "ay", "gm2", "bgm", "bgm2", "yt", "vv_io", "ww_io", "yy_io"]
    vk2, ios2 = wire.pinocchio_vkey_of_json(js)
    assert ios2 == ios and bytes(vk2.g1) == v1 and bytes(vk2.g2) == v2


def test_writer_reproduces_the_golden_json_bytes():
    """tests/golden/readme_groth16_wire.hex: the README circuit's pkey and proof as JSON text from an independent writer
    (tests/golden/make_readme_wire.py) over the first-principles key of readme_groth16_key.json -- exact bytes, both directions."""
    import json
    import os
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    fix = json.load(open(os.path.join(here, "readme_groth16_key.json")))
    pkey_js, proof_js = (bytes.fromhex(line) for line in open(os.path.join(here, "readme_groth16_wire.hex")).read().split())
    from zukelang_amd.groth16 import PKey
    cat = lambda xs: b"".join(bytes.fromhex(x) for x in xs)
    pk = PKey(np.frombuffer(cat(fix["pk_g1"]), dtype=np.uint8), np.frombuffer(cat(fix["pk_g2"]), dtype=np.uint8))
    mids = [("c", 4), ("c", 5), ("input", 3)]
    assert wire.groth16_pkey_to_json(pk, 3, mids) == pkey_js
    proof = Proof(*(bytes.fromhex(fix["proof"][k]) for k in "abc"))
    assert wire.groth16_proof_to_json(proof) == proof_js
    pk2, mids2 = wire.groth16_pkey_of_json(pkey_js)
    assert mids2 == mids and bytes(pk2.g1) == bytes(pk.g1) and bytes(pk2.g2) == bytes(pk.g2)
    back = wire.groth16_proof_of_json(proof_js)
    assert (back.a, back.b, back.c) == (proof.a, proof.b, proof.c)


def test_vectorised_compression_is_the_per_point_function():
    """curve.G1/G2.to_compressed_bytes_many (numpy byte logic over a whole key, used by the *_pkey_to_json writers) == zk_g1/g2_compress per point:
    both signs of y, the identity, and for G2 a y with zero imaginary part (the sign then comes from the real part)."""
    import oracle_lib as O
    from oracle import pyref as P
    from zukelang_amd.curve import G1, G2
    st = P.fr_stream(0x5EEDC0DE)
    ks = [1, 2, 3, P.R - 1, P.R - 2] + [next(st) for _ in range(40)]
    for grp, gen, mul in ((G1, O.g1_generator, O.g1_mul), (G2, O.g2_generator, O.g2_mul)):
        B = grp.POINT_BYTES
        pts = [mul(gen(), P.fr_to_bytes(k)) for k in ks] + [bytes([0x40]) + bytes(B - 1)]
        if grp is G2:          # a synthetic encoding with y1 = 0: only the byte logic is under test here
            y0_large = bytes(96) + bytes(48) + bytes([0x19]) + bytes(47)
            y0_small = bytes(96) + bytes(48) + bytes([0x01]) + bytes(47)
            pts += [y0_large, y0_small]
        many = grp.to_compressed_bytes_many(b"".join(pts))
        Cb = grp.COMPRESSED_BYTES
        assert len(many) == Cb * len(pts)
        signs = set()
        for i, pt in enumerate(pts):
            one = grp.to_compressed_bytes(pt)
            assert many[Cb * i:Cb * (i + 1)] == one, (grp.__name__, i)
            signs.add(one[0] & 0x20)
        assert signs == {0, 0x20}
