"""Wire / on-disk format of proofs and keys (scope row f3): the JSON the reference derives with
ppx_yojson_conv (`[@@deriving yojson]` on groth16.ml:24-43,110-114 and pinocchio.ml:37-75,195-208).

What those derivers produce, restated (none of it can be run here -- parity unpinned):
  record            {"field": value, ...} in declaration order
  'a list           [v, ...]
  'a Var.Map.t      [[[name, id], v], ...]  bindings in key order (var.ml:33-40,66-68; var = string * int)
  Fr.t              decimal string of the integer (curve.ml:139-140 via Z.to_string, misc.ml:33-37)
  G1.t / G2.t       JSON string holding the RAW 48 / 96 compressed bytes (curve.ml:199-212)
  GT.t              JSON string of the external library's bytes (curve.ml:217-219) -- not reproducible
                    here; this module writes the 576-byte coefficient encoding of include/zkmi355x.h
A JSON string of raw bytes is not UTF-8.  Yojson.Safe.to_string writes bytes >= 0x80 as they are and
escapes only '"', '\\', the control characters (\\b \\f \\n \\r \\t, the rest as \\u00XX) and 0x7f: the
writer and parser below work on BYTES and reproduce exactly that.
"""
import ctypes as C

from . import _lib
from .curve import G1, G2

_ESC = {0x22: b'\\"', 0x5C: b"\\\\", 0x08: b"\\b", 0x0C: b"\\f", 0x0A: b"\\n", 0x0D: b"\\r", 0x09: b"\\t"}
_UNESC = {ord("b"): 8, ord("f"): 12, ord("n"): 10, ord("r"): 13, ord("t"): 9, 0x22: 0x22, 0x5C: 0x5C, ord("/"): ord("/")}


def json_bytes_string(b):
    """`String s of Yojson with s = raw bytes."""
    out = bytearray(b'"')
    for c in bytes(b):
        if c in _ESC:
            out += _ESC[c]
        elif c < 0x20 or c == 0x7F:
            out += b"\\u00%02x" % c
        else:
            out.append(c)
    out += b'"'
    return bytes(out)


def dumps(v):
    """Python value -> JSON bytes: dict (insertion order) / list / tuple / int / bytes (raw string) / str."""
    if isinstance(v, dict):
        return b"{" + b",".join(json_bytes_string(k.encode("latin-1")) + b":" + dumps(x) for k, x in v.items()) + b"}"
    if isinstance(v, (list, tuple)):
        return b"[" + b",".join(dumps(x) for x in v) + b"]"
    if isinstance(v, bool):
        return b"true" if v else b"false"
    if isinstance(v, int):
        return str(v).encode()
    if isinstance(v, str):
        return json_bytes_string(v.encode("latin-1"))       # names are BYTE strings (OCaml string): one code point per byte, as Yojson writes them
    return json_bytes_string(bytes(v))


def loads(data):
    """JSON bytes -> dict / list / int / bytes (every string comes back as bytes; record field names as latin-1 str).
    Malformed input raises ValueError (never an assert: `python -O` must not turn the parser permissive)."""
    try:
        return _loads_body(bytes(data))
    except (IndexError, KeyError) as e:
        raise ValueError("JSON: truncated or malformed input (%s)" % type(e).__name__) from None


def _reader(fn):
    """A record reader fed something that is not the record (missing field, wrong nesting) raises ValueError like the parser itself."""
    import functools

    @functools.wraps(fn)
    def wrapped(data):
        try:
            return fn(data)
        except (KeyError, IndexError, TypeError, AttributeError) as e:
            raise ValueError("wire: not a %s record (%s: %s)" % (fn.__name__.replace("_of_json", ""), type(e).__name__, e)) from None
    return wrapped


def _need(cond, what):
    if not cond:
        raise ValueError("wire: " + what)


def _loads_body(data):
    pos = 0

    def need(cond, what):
        if not cond:
            raise ValueError("JSON: %s at byte %d" % (what, pos))

    def ws():
        nonlocal pos
        while pos < len(data) and data[pos] in b" \t\r\n":
            pos += 1

    def value():
        nonlocal pos
        ws()
        c = data[pos]
        if c == 0x7B:       # {
            pos += 1
            out = {}
            ws()
            if data[pos] == 0x7D:
                pos += 1
                return out
            while True:
                ws()
                k = string().decode("latin-1")
                ws()
                need(data[pos] == 0x3A, "':' expected")
                pos += 1
                out[k] = value()
                ws()
                pos += 1
                if data[pos - 1] == 0x7D:
                    return out
                need(data[pos - 1] == 0x2C, "',' expected")
        if c == 0x5B:       # [
            pos += 1
            out = []
            ws()
            if data[pos] == 0x5D:
                pos += 1
                return out
            while True:
                out.append(value())
                ws()
                pos += 1
                if data[pos - 1] == 0x5D:
                    return out
                need(data[pos - 1] == 0x2C, "',' expected")
        if c == 0x22:
            return string()
        start = pos
        while pos < len(data) and data[pos] in b"+-0123456789":
            pos += 1
        need(pos > start, "value expected")
        return int(data[start:pos])

    def string():
        nonlocal pos
        need(data[pos] == 0x22, "string expected")
        pos += 1
        out = bytearray()
        while data[pos] != 0x22:
            c = data[pos]
            if c == 0x5C:
                e = data[pos + 1]
                if e == ord("u"):
                    out.append(int(data[pos + 2:pos + 6], 16) & 0xFF)
                    pos += 6
                else:
                    out.append(_UNESC[e])
                    pos += 2
            else:
                out.append(c)
                pos += 1
        pos += 1
        return bytes(out)

    v = value()
    ws()
    need(pos == len(data), "trailing bytes after the JSON value")
    return v


def _decompress(fn, comp, n_out):
    out = C.create_string_buffer(n_out)
    _lib.check(fn(bytes(comp), out))
    return out.raw


def g1_of_json(b):
    return _decompress(_lib.lib().zk_g1_decompress, b, 96)


def g2_of_json(b):
    return _decompress(_lib.lib().zk_g2_decompress, b, 192)


# A key is lists of thousands to millions of compressed points: from BATCH_MIN points up a list goes through the GPU in one call
# (zk_g1/g2_decompress_batch: ~1 s for the five million points of a 2^20 key against half an hour of one host core), shorter ones -- proofs,
# verification keys, the fixtures of the CPU tests -- through the one-point host calls.  Same checks, same bytes either way.
BATCH_MIN = 256


def g1s_of_json(bs):
    if len(bs) < BATCH_MIN:
        return b"".join(g1_of_json(b) for b in bs)
    _need(all(len(b) == 48 for b in bs), "a compressed G1 point is 48 bytes")
    return G1.of_compressed_bytes_many(b"".join(bs))


def g2s_of_json(bs):
    if len(bs) < BATCH_MIN:
        return b"".join(g2_of_json(b) for b in bs)
    _need(all(len(b) == 96 for b in bs), "a compressed G2 point is 96 bytes")
    return G2.of_compressed_bytes_many(b"".join(bs))


# ---- Groth16 (groth16.ml:110-114, 36-43, 24-34)
def groth16_proof_to_json(proof):
    return dumps({"a": G1.to_compressed_bytes(proof.a), "b": G2.to_compressed_bytes(proof.b), "c": G1.to_compressed_bytes(proof.c)})


@_reader
def groth16_proof_of_json(data):
    from .groth16 import Proof
    d = loads(data)
    return Proof(g1_of_json(d["a"]), g2_of_json(d["b"]), g1_of_json(d["c"]))


def groth16_vkey_to_json(vk, io_vars):
    """io_vars: the (name, id) of the public variables in key order (Var.Map order = the order of vk.ltgm_io)."""
    lt = bytes(vk.ltgm_io)
    _need(len(lt) == 96 * len(io_vars), "key / record lengths or variable domains do not match")
    return dumps({"one1": G1.to_compressed_bytes(vk.one1),
                  "ltgm_io": [[[name, vid], G1.to_compressed_bytes(lt[96 * i:96 * i + 96])] for i, (name, vid) in enumerate(io_vars)],
                  "one2": G2.to_compressed_bytes(vk.one2), "gm": G2.to_compressed_bytes(vk.gm), "d": G2.to_compressed_bytes(vk.d),
                  "ab": bytes(vk.ab)})


@_reader
def groth16_vkey_of_json(data):
    from .groth16 import VKey
    import numpy as np
    d = loads(data)
    io_vars = [(b[0][0].decode("latin-1"), b[0][1]) for b in d["ltgm_io"]]
    lt = b"".join(g1_of_json(b[1]) for b in d["ltgm_io"])
    vk = VKey(g1_of_json(d["one1"]), np.frombuffer(lt, dtype=np.uint8), g2_of_json(d["one2"]), g2_of_json(d["gm"]), g2_of_json(d["d"]), d["ab"])
    return vk, io_vars


def groth16_pkey_to_json(pk, n, mid_vars):
    """Record fields in the reference's declaration order (groth16.ml:24-34): a, d1, ti1, ltd_mid, tiztd, b1, b2, d2, ti2."""
    g1, g2 = bytes(pk.g1), bytes(pk.g2)
    c1, c2 = G1.to_compressed_bytes_many(g1), G2.to_compressed_bytes_many(g2)          # every point of the key in two vectorised passes
    p1 = lambda i: c1[48 * i:48 * i + 48]
    p2 = lambda i: c2[96 * i:96 * i + 96]
    o_ti, o_tz = 3, 3 + (n + 2)
    o_lt = o_tz + (n - 1)
    _need(len(g1) == 96 * (o_lt + len(mid_vars)) and len(g2) == 192 * (2 + n + 2), "key / record lengths or variable domains do not match")
    return dumps({"a": p1(0), "d1": p1(1), "ti1": [p1(o_ti + i) for i in range(n + 2)],
                  "ltd_mid": [[[name, vid], p1(o_lt + i)] for i, (name, vid) in enumerate(mid_vars)],
                  "tiztd": [p1(o_tz + i) for i in range(n - 1)], "b1": p1(2), "b2": p2(0), "d2": p2(1),
                  "ti2": [p2(2 + i) for i in range(n + 2)]})


@_reader
def groth16_pkey_of_json(data):
    from .groth16 import PKey
    import numpy as np
    d = loads(data)
    g1 = g1s_of_json([d["a"], d["d1"], d["b1"]] + list(d["ti1"]) + list(d["tiztd"]) + [b[1] for b in d["ltd_mid"]])
    g2 = g2s_of_json([d["b2"], d["d2"]] + list(d["ti2"]))
    mid_vars = [(b[0][0].decode("latin-1"), b[0][1]) for b in d["ltd_mid"]]
    return PKey(np.frombuffer(g1, dtype=np.uint8), np.frombuffer(g2, dtype=np.uint8)), mid_vars


# ---- Pinocchio proof (pinocchio.ml:195-208)
_PIN_FIELDS = (("vv", 1), ("ww", 2), ("yy", 1), ("h", 1), ("vavv", 1), ("waww", 2), ("yayy", 1), ("bvwy", 1))


def pinocchio_proof_to_json(proof):
    return dumps({f: (G1 if g == 1 else G2).to_compressed_bytes(getattr(proof, f)) for f, g in _PIN_FIELDS})


@_reader
def pinocchio_proof_of_json(data):
    from .pinocchio import Proof
    d = loads(data)
    return Proof(*[(g1_of_json if g == 1 else g2_of_json)(d[f]) for f, g in _PIN_FIELDS])


def fr_to_json(x):
    return dumps(str(int(x)))        # Z.yojson_of_t: the decimal string


@_reader
def fr_of_json(data):
    return int(loads(data).decode())


# ---- Pinocchio keys (pinocchio.ml:37-60 pkey, :62-75 vkey), record fields in declaration order.
# Flat pools of include/zkmi355x.h:  pk g1 = vv | yy | vav | yay | bvwy (n_mid each) | si (n+1) | v_all (m) | w_all (m) | vt | yt | vavt | yayt | vbt | wbt | ybt
#                                    pk g2 = ww | waw (n_mid each) | si2 (n+1) | wt | wawt
#                                    vk g1 = one | aw | bgm | vv_io | yy_io        vk g2 = one2 | av | ay | gm2 | bgm2 | yt | ww_io
def _vmap(vars_, pts, conv):
    return [[[name, vid], conv(p)] for (name, vid), p in zip(vars_, pts)]


def pinocchio_pkey_to_json(pk, n, mid_vars, all_vars):
    """mid_vars: (name, id) of circuit.mids in Var.Map order; all_vars: every variable (the domain of v_all / w_all)."""
    g1, g2 = bytes(pk.g1), bytes(pk.g2)
    k, m = len(mid_vars), len(all_vars)
    _need(len(g1) == 96 * (5 * k + (n + 1) + 2 * m + 7) and len(g2) == 192 * (2 * k + (n + 1) + 2), "key / record lengths or variable domains do not match")
    cc1, cc2 = G1.to_compressed_bytes_many(g1), G2.to_compressed_bytes_many(g2)          # every point of the key in two vectorised passes
    c1 = lambda i: cc1[48 * i:48 * i + 48]
    c2 = lambda i: cc2[96 * i:96 * i + 96]
    r1 = lambda o, cnt: [c1(o + i) for i in range(cnt)]
    r2 = lambda o, cnt: [c2(o + i) for i in range(cnt)]
    o_si, o_va = 5 * k, 5 * k + n + 1
    o_wa = o_va + m
    o_t = o_wa + m
    o_si2 = 2 * k
    o_t2 = o_si2 + n + 1
    ident = lambda x: x
    return dumps({"vv": _vmap(mid_vars, r1(0, k), ident), "ww": _vmap(mid_vars, r2(0, k), ident), "yy": _vmap(mid_vars, r1(k, k), ident),
                  "vav": _vmap(mid_vars, r1(2 * k, k), ident), "waw": _vmap(mid_vars, r2(k, k), ident), "yay": _vmap(mid_vars, r1(3 * k, k), ident),
                  "si": r1(o_si, n + 1), "bvwy": _vmap(mid_vars, r1(4 * k, k), ident), "si2": r2(o_si2, n + 1),
                  "vt": c1(o_t), "wt": c2(o_t2), "yt": c1(o_t + 1), "vavt": c1(o_t + 2), "wawt": c2(o_t2 + 1), "yayt": c1(o_t + 3),
                  "vbt": c1(o_t + 4), "wbt": c1(o_t + 5), "ybt": c1(o_t + 6),
                  "v_all": _vmap(all_vars, r1(o_va, m), ident), "w_all": _vmap(all_vars, r1(o_wa, m), ident)})


@_reader
def pinocchio_pkey_of_json(data):
    """-> (pinocchio.PKey, n, mid_vars, all_vars)"""
    from .pinocchio import PKey
    import numpy as np
    d = loads(data)
    vars_of = lambda f: [(b[0][0].decode("latin-1"), b[0][1]) for b in d[f]]
    m1 = lambda f: [b[1] for b in d[f]]
    m2 = lambda f: [b[1] for b in d[f]]
    mid_vars, all_vars = vars_of("vv"), vars_of("v_all")
    for f in ("ww", "yy", "vav", "waw", "yay", "bvwy"):
        _need(vars_of(f) == mid_vars, "Pinocchio pkey: the I_mid maps must share one domain")
    _need(vars_of("w_all") == all_vars, "key / record lengths or variable domains do not match")
    g1 = g1s_of_json(m1("vv") + m1("yy") + m1("vav") + m1("yay") + m1("bvwy") + list(d["si"]) + m1("v_all") + m1("w_all")
                     + [d[f] for f in ("vt", "yt", "vavt", "yayt", "vbt", "wbt", "ybt")])
    g2 = g2s_of_json(m2("ww") + m2("waw") + list(d["si2"]) + [d["wt"], d["wawt"]])
    return (PKey(np.frombuffer(g1, dtype=np.uint8), np.frombuffer(g2, dtype=np.uint8)), len(d["si"]) - 1, mid_vars, all_vars)


def pinocchio_vkey_to_json(vk, io_vars):
    g1, g2 = bytes(vk.g1), bytes(vk.g2)
    k = len(io_vars)
    _need(len(g1) == 96 * (3 + 2 * k) and len(g2) == 192 * (6 + k), "key / record lengths or variable domains do not match")
    c1 = lambda i: G1.to_compressed_bytes(g1[96 * i:96 * i + 96])
    c2 = lambda i: G2.to_compressed_bytes(g2[192 * i:192 * i + 192])
    ident = lambda x: x
    return dumps({"one": c1(0), "one2": c2(0), "av": c2(1), "aw": c1(1), "ay": c2(2), "gm2": c2(3), "bgm": c1(2), "bgm2": c2(4), "yt": c2(5),
                  "vv_io": _vmap(io_vars, [c1(3 + i) for i in range(k)], ident), "ww_io": _vmap(io_vars, [c2(6 + i) for i in range(k)], ident),
                  "yy_io": _vmap(io_vars, [c1(3 + k + i) for i in range(k)], ident)})


@_reader
def pinocchio_vkey_of_json(data):
    from .pinocchio import VKey
    import numpy as np
    d = loads(data)
    io_vars = [(b[0][0].decode("latin-1"), b[0][1]) for b in d["vv_io"]]
    _need([(b[0][0].decode("latin-1"), b[0][1]) for b in d["ww_io"]] == io_vars and [(b[0][0].decode("latin-1"), b[0][1]) for b in d["yy_io"]] == io_vars, "key / record lengths or variable domains do not match")
    g1 = [g1_of_json(d["one"]), g1_of_json(d["aw"]), g1_of_json(d["bgm"])] + [g1_of_json(b[1]) for b in d["vv_io"]] + [g1_of_json(b[1]) for b in d["yy_io"]]
    g2 = [g2_of_json(d[f]) for f in ("one2", "av", "ay", "gm2", "bgm2", "yt")] + [g2_of_json(b[1]) for b in d["ww_io"]]
    return VKey(np.frombuffer(b"".join(g1), dtype=np.uint8), np.frombuffer(b"".join(g2), dtype=np.uint8)), io_vars
