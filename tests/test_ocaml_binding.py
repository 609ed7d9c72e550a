"""The OCaml binding under ocaml/ cannot be compiled in this image (no OCaml toolchain: SURVEY.md 8c).  What CAN be held to account
without one: every C symbol the .ml files name exists in the built library and in include/zkmi355x.h, every `foreign` declaration has the
header's arity and the header's argument kinds in the header's order, the records that carry the wire format declare their fields in the order
the reference does (their yojson is the reference's JSON, groth16.ml:24-43,110-114 / pinocchio.ml:37-75,195-208), and the install script
names files that exist.  The C99 hosts examples/c_prove.c and examples/c_pinocchio.c run the same call sequences on the GPU box."""
import os
import re

from zukelang_amd import _lib, wire

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ODIR = os.path.join(ROOT, "ocaml")
ML = {f: open(os.path.join(ODIR, f)).read() for f in os.listdir(ODIR) if f.endswith(".ml")}
HEADER = open(os.path.join(ROOT, "include", "zkmi355x.h")).read()


def _strip_comments(src):
    out, depth, i = [], 0, 0
    while i < len(src):
        if src.startswith("(*", i):
            depth += 1; i += 2
        elif src.startswith("*)", i) and depth:
            depth -= 1; i += 2
        else:
            if not depth:
                out.append(src[i])
            i += 1
    return "".join(out)


def _header_prototypes():
    """name -> list of C parameter declarations."""
    body = re.sub(r"/\*.*?\*/", " ", HEADER, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(?:int|const char\*)\s+(zk_[a-z0-9_]+)\s*\(([^)]*)\)\s*;", body):
        params = [p.strip() for p in m.group(2).split(",")]
        protos[m.group(1)] = [] if params == ["void"] else params
    return protos


def _kind_of_c(param):
    p = re.sub(r"\[[^\]]*\]", "*", param)           # uint8_t out[96] is a pointer
    if "*" in p:
        return "ptr"
    for t, k in (("uint64_t", "uint64_t"), ("uint32_t", "uint32_t"), ("size_t", "size_t"), ("int32_t", "int32_t"), ("int", "int"), ("double", "double")):
        if re.search(r"\b%s\b" % t, p):
            return k
    raise AssertionError("unparsed C parameter: %r" % param)


def _kind_of_ml(arg):
    a = arg.strip().strip("()").strip()
    if a in ("ocaml_bytes", "string") or a.startswith("ptr "):
        return "ptr"
    return a


def _foreign_decls():
    """(ocaml name, C symbol, [argument types], return type) of every `fn "sym" (...)` in mi355x.ml."""
    src = _strip_comments(ML["mi355x.ml"])
    out = []
    for m in re.finditer(r"let\s+(\w+)\s*=\s*fn\s+\"(zk_[a-z0-9_]+)\"\s*\(", src):
        i, depth = m.end(), 1
        while depth:
            depth += {"(": 1, ")": -1}.get(src[i], 0)
            i += 1
        sig = " ".join(src[m.end():i - 1].split())
        parts = [p.strip() for p in sig.split("@->")]
        assert parts[-1].startswith("returning "), sig
        out.append((m.group(1), m.group(2), parts[:-1], parts[-1][len("returning "):]))
    return out


def test_every_symbol_the_ocaml_files_name_is_exported_and_declared():
    protos = _header_prototypes()
    lib = _lib.lib()
    named = set()
    for f, src in ML.items():
        named |= set(re.findall(r"\bzk_[a-z0-9_]+", _strip_comments(src))) - {"zk_csr"}
    ml_lets = {d[0] for d in _foreign_decls()}
    c_syms = {d[1] for d in _foreign_decls()}
    assert len(c_syms) >= 40
    for name in sorted(named):
        sym = name if name in protos else next((d[1] for d in _foreign_decls() if d[0] == name), None)
        assert sym is not None and sym in protos, "%s is not in include/zkmi355x.h" % name
        assert sym in _lib.EXPORTS and hasattr(lib, sym), "%s is not exported by libzkmi355x.so" % name
    # ... and what the protocol files call through `Mi355x.` is bound there
    for f in ("groth16_mi355x.ml", "pinocchio_mi355x.ml", "bls12_381_mi355x.ml"):
        for used in set(re.findall(r"\b(zk_[a-z0-9_]+)", _strip_comments(ML[f]))):
            assert used in ml_lets, "%s uses %s, which mi355x.ml does not bind" % (f, used)
    # the surface an OCaml host needs
    need = {"zk_pinocchio_pk_upload", "zk_pinocchio_prove", "zk_pinocchio_prove_async", "zk_pinocchio_prove_wait", "zk_pinocchio_pk_derive_lagrange",
            "zk_pinocchio_pk_free", "zk_groth16_verify", "zk_pinocchio_verify", "zk_g1_of_fr", "zk_g2_of_fr", "zk_groth16_qap_eval",
            "zk_groth16_reserve_slots", "zk_groth16_set_witness", "zk_groth16_prove_async", "zk_groth16_prove_wait", "zk_set_devices", "zk_set_option",
            "zk_groth16_pk_upload", "zk_groth16_prove", "zk_groth16_pk_derive_lagrange", "zk_msm_g1", "zk_msm_g2", "zk_g1_powers", "zk_g2_powers", "zk_fr_ntt"}
    assert need <= c_syms, sorted(need - c_syms)


def test_foreign_declarations_match_the_header():
    protos = _header_prototypes()
    for ml_name, sym, args, ret in _foreign_decls():
        want = protos[sym]
        got = [] if args == ["void"] else args
        assert len(got) == len(want), "%s: %d arguments in mi355x.ml, %d in the header" % (ml_name, len(got), len(want))
        for a, p in zip(got, want):
            assert _kind_of_ml(a) == _kind_of_c(p), "%s: `%s` against `%s`" % (ml_name, a, p)
        assert ret in ("int", "string")
        assert (ret == "string") == (sym in ("zk_strerror", "zk_last_error"))


def _record_fields(src, type_name):
    m = re.search(r"type %s\s*=\s*\{(.*?)\}" % type_name, _strip_comments(src), flags=re.S)
    return [f.split(":")[0].strip() for f in m.group(1).split(";") if f.strip()]


def test_record_field_order_is_the_wire_format():
    g = ML["groth16_mi355x.ml"]
    assert _record_fields(g, "pkey") == ["a", "d1", "ti1", "ltd_mid", "tiztd", "b1", "b2", "d2", "ti2"]          # groth16.ml:24-34
    assert _record_fields(g, "vkey") == ["one1", "ltgm_io", "one2", "gm", "d", "ab"]                             # groth16.ml:36-43
    assert _record_fields(g, "proof") == ["a", "b", "c"]                                                         # groth16.ml:110-114
    p = ML["pinocchio_mi355x.ml"]
    assert _record_fields(p, "pkey") == ["vv", "ww", "yy", "vav", "waw", "yay", "si", "bvwy", "si2", "vt", "wt", "yt", "vavt", "wawt", "yayt", "vbt", "wbt",
                                         "ybt", "v_all", "w_all"]                                                 # pinocchio.ml:37-60
    assert _record_fields(p, "vkey") == ["one", "one2", "av", "aw", "ay", "gm2", "bgm", "bgm2", "yt", "vv_io", "ww_io", "yy_io"]      # pinocchio.ml:62-75
    assert _record_fields(p, "proof") == [f for f, _ in wire._PIN_FIELDS]                                        # pinocchio.ml:195-208
    # the same orders are what zukelang_amd/wire.py writes (golden JSON bytes: tests/test_wire.py)
    import numpy as np
    from oracle import pyref as P
    from zukelang_amd.groth16 import PKey
    pt1 = lambda k: P.g1_to_bytes(P.pt_mul(P.G1, k))
    pt2 = lambda k: P.g2_to_bytes(P.pt_mul(P.G2, k))
    pk = PKey(np.frombuffer(b"".join(pt1(2 + i) for i in range(3 + 3 + 0 + 1)), dtype=np.uint8), np.frombuffer(b"".join(pt2(2 + i) for i in range(2 + 3)), dtype=np.uint8))
    assert list(wire.loads(wire.groth16_pkey_to_json(pk, 1, [("input", 3)]))) == _record_fields(g, "pkey")


def test_error_map_and_install_script():
    src = _strip_comments(ML["mi355x.ml"])
    codes = dict(re.findall(r"#define (ZK_ERR_[A-Z_]+) \((-\d+)\)", HEADER))
    for code, exc in ((codes["ZK_ERR_APPLY_POWERS"], 'invalid_arg "apply_powers"'), (codes["ZK_ERR_REMAINDER"], "Assert_failure"), (codes["ZK_ERR_DOMAIN"], "Assert_failure"),
                      (codes["ZK_ERR_NOT_ON_CURVE"], "Not_on_curve"), (codes["ZK_ERR_SCALAR_RANGE"], "Not_in_field")):
        assert re.search(r"\|\s*%s\s*->\s*[^|]*%s" % (re.escape(code), re.escape(exc)), src), (code, exc)
    sh = open(os.path.join(ODIR, "install.sh")).read()
    for f in re.findall(r"\$HERE/([\w/.]+)", sh):
        assert os.path.exists(os.path.join(ODIR, f)), f
    # delta_v, delta_w, delta_y are drawn in the order of pinocchio.ml:428-430; r before s (groth16.ml:124-125)
    pz = _strip_comments(ML["pinocchio_mi355x.ml"])
    assert re.search(r"let dv = Fr\.gen rng in\s*let dw = Fr\.gen rng in\s*let dy = Fr\.gen rng in", pz)
    assert re.search(r"let rv = Fr\.gen rng in\s*let rw = Fr\.gen rng in\s*let s = Fr\.gen rng in\s*let av = Fr\.gen rng in\s*let aw = Fr\.gen rng in\s*let ay = Fr\.gen rng in\s*"
                     r"let b = Fr\.gen rng in\s*let gm = Fr\.gen rng in", pz)
    gz = _strip_comments(ML["groth16_mi355x.ml"])
    assert re.search(r"let r = Fr\.gen rng in\s*let s = Fr\.gen rng in", gz)
    assert re.search(r"let alpha = Fr\.gen rng in\s*let beta = Fr\.gen rng in\s*let gamma = Fr\.gen rng in\s*let delta = Fr\.gen rng in\s*let tau = Fr\.gen rng in", gz)
    # balanced comments, parentheses and brackets in every file (the cheapest syntax check there is)
    for f, s in ML.items():
        body = re.sub(r'"(?:\\.|[^"\\])*"', '""', _strip_comments(s))
        body = re.sub(r"'(?:\\.|[^'\\])'", "' '", body)
        for o, c in ("()", "[]", "{}"):
            assert body.count(o) == body.count(c), (f, o, body.count(o), body.count(c))
        assert len(re.findall(r"\bstruct\b|\bsig\b|\bbegin\b", body)) == len(re.findall(r"\bend\b", body)), f
