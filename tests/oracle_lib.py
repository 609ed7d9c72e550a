"""ctypes binding of the CPU oracle (oracle/libzkoracle.so) -- TEST INFRASTRUCTURE ONLY.

Imported by tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg; never by the
product package zukelang_amd/.
"""
import ctypes as C
import os
import subprocess
import sys

import numpy as np

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_ODIR = os.path.join(_ROOT, "oracle")
sys.path.insert(0, _ROOT)


def build():
    subprocess.check_call(["make", "-s", "-C", _ODIR])


def _load():
    so = os.environ.get("ZK_ORACLE_SO") or os.path.join(_ODIR, "libzkoracle.so")      # ZK_ORACLE_SO: the sanitizer build (make -C oracle asan)
    if not os.path.exists(so):
        build()
    return C.CDLL(so)


lib = _load()
u8p = C.POINTER(C.c_uint8)
u32p = C.POINTER(C.c_uint32)
lib.orc_qap_build.restype = C.c_void_p
lib.orc_poly_mul.restype = C.c_size_t


def _buf(n):
    return (C.c_uint8 * n)()


def _b(x):
    """bytes / numpy -> ctypes pointer (keeps a reference alive via the returned tuple)."""
    if isinstance(x, (bytes, bytearray)):
        arr = (C.c_uint8 * len(x)).from_buffer_copy(bytes(x))
        return arr
    if isinstance(x, np.ndarray):
        return x.ctypes.data_as(u8p)
    return x


def fr_mul(a, b):
    o = _buf(32); lib.orc_fr_mul(o, _b(a), _b(b)); return bytes(o)


def fr_inv(a):
    o = _buf(32); lib.orc_fr_inv(o, _b(a)); return bytes(o)


def fr_omega():
    o = _buf(32); lib.orc_fr_omega(o); return bytes(o)


def g1_generator():
    o = _buf(96); lib.orc_g1_generator(o); return bytes(o)


def g2_generator():
    o = _buf(192); lib.orc_g2_generator(o); return bytes(o)


def g1_add(a, b):
    o = _buf(96); assert lib.orc_g1_add(o, _b(a), _b(b)) == 0; return bytes(o)


def g1_mul(a, k):
    o = _buf(96); assert lib.orc_g1_mul(o, _b(a), _b(k)) == 0; return bytes(o)


def g2_add(a, b):
    o = _buf(192); assert lib.orc_g2_add(o, _b(a), _b(b)) == 0; return bytes(o)


def g2_mul(a, k):
    o = _buf(192); assert lib.orc_g2_mul(o, _b(a), _b(k)) == 0; return bytes(o)


def g1_compress(a):
    o = _buf(48); assert lib.orc_g1_compress(o, _b(a)) == 0; return bytes(o)


def g2_compress(a):
    o = _buf(96); assert lib.orc_g2_compress(o, _b(a)) == 0; return bytes(o)


def g1_powers(d, s):
    o = _buf(96 * (d + 1)); lib.orc_g1_powers(o, d, _b(s)); return bytes(o)


def g2_powers(d, s):
    o = _buf(192 * (d + 1)); lib.orc_g2_powers(o, d, _b(s)); return bytes(o)


def g1_msm_naive(bases, scalars):
    o = _buf(96)
    rc = lib.orc_g1_msm_naive(o, _b(bases), C.c_size_t(len(bases) // 96), _b(scalars), C.c_size_t(len(scalars) // 32))
    return rc, bytes(o)


def g2_msm_naive(bases, scalars):
    o = _buf(192)
    rc = lib.orc_g2_msm_naive(o, _b(bases), C.c_size_t(len(bases) // 192), _b(scalars), C.c_size_t(len(scalars) // 32))
    return rc, bytes(o)


def fr_ntt(data, log_n, inverse):
    buf = (C.c_uint8 * len(data)).from_buffer_copy(bytes(data))
    assert lib.orc_fr_ntt(buf, log_n, int(inverse)) == 0
    return bytes(buf)


def poly_mul(a, b):
    na, nb = len(a) // 32, len(b) // 32
    o = _buf(32 * (na + nb + 1))
    n = lib.orc_poly_mul(o, _b(a), C.c_size_t(na), _b(b), C.c_size_t(nb))
    return bytes(o)[:32 * n]


def poly_divrem(a, b):
    na, nb = len(a) // 32, len(b) // 32
    q = _buf(32 * (na + 1)); r = _buf(32 * (na + 1))
    nq = C.c_size_t(); nr = C.c_size_t()
    lib.orc_poly_divrem(q, C.byref(nq), r, C.byref(nr), _b(a), C.c_size_t(na), _b(b), C.c_size_t(nb))
    return bytes(q)[:32 * nq.value], bytes(r)[:32 * nr.value]


def fr_dot(a, b):
    o = _buf(32); lib.orc_fr_dot(o, _b(a), _b(b), C.c_size_t(len(a) // 32)); return bytes(o)


class CSR:
    """One R1CS matrix: rows = gates, cols = variables, vals = Fr 32 B LE."""

    def __init__(self, ptr, col, val):
        self.ptr = np.ascontiguousarray(ptr, dtype=np.uint32)
        self.col = np.ascontiguousarray(col, dtype=np.uint32)
        self.val = np.ascontiguousarray(np.frombuffer(bytes(val), dtype=np.uint8)) if not isinstance(val, np.ndarray) else np.ascontiguousarray(val, dtype=np.uint8)

    def args(self):
        return (self.ptr.ctypes.data_as(u32p), self.col.ctypes.data_as(u32p), self.val.ctypes.data_as(u8p))


def r1cs_spmv(n, M, sol):
    o = _buf(32 * n)
    lib.orc_r1cs_spmv(n, *M.args(), _b(sol), o)
    return bytes(o)


class QAP:
    def __init__(self, n, m, L, R, O):
        self.n, self.m = n, m
        self.h = C.c_void_p(lib.orc_qap_build(n, m, *L.args(), *R.args(), *O.args()))

    def __del__(self):
        if getattr(self, "h", None):
            lib.orc_qap_free(self.h); self.h = None

    def poly(self, which, k=0):
        n = self.n + 1 if which == 3 else self.n
        o = _buf(32 * n); lib.orc_qap_get(self.h, which, k, o); return bytes(o)

    def eval(self, sol):
        n = self.n
        p = _buf(32 * (2 * n + 2)); h = _buf(32 * (2 * n + 2))
        np_ = C.c_size_t(); nh = C.c_size_t()
        rc = lib.orc_qap_eval(self.h, _b(sol), p, C.byref(np_), h, C.byref(nh))
        return rc, bytes(p)[:32 * np_.value], bytes(h)[:32 * nh.value]

    def eval_vwy(self, sol):
        n = self.n
        v = _buf(32 * n); w = _buf(32 * n); y = _buf(32 * n)
        lib.orc_qap_eval_vw(self.h, _b(sol), v, w, y)
        return bytes(v), bytes(w), bytes(y)

    def groth16_setup(self, toxic, mid):
        n, m = self.n, self.m
        nmid = int(np.asarray(mid, dtype=np.int64).sum()); nio = m - nmid
        pk1 = _buf(96 * (3 + (n + 2) + max(n - 1, 0) + nmid)); pk2 = _buf(192 * (2 + n + 2))
        vk1 = _buf(96 * (1 + nio)); vk2 = _buf(192 * 3)
        lib.orc_groth16_setup(self.h, _b(toxic), _b(bytes(mid)), pk1, pk2, vk1, vk2)
        return bytes(pk1), bytes(pk2), bytes(vk1), bytes(vk2)

    def groth16_prove(self, pk1, pk2, mid, sol, r, s, literal):
        a = _buf(96); b = _buf(192); c = _buf(96)
        rc = lib.orc_groth16_prove(self.h, _b(pk1), _b(pk2), _b(bytes(mid)), _b(sol), _b(r), _b(s), int(literal), a, b, c)
        return rc, bytes(a), bytes(b), bytes(c)


def groth16_prove_trapdoor(n, m, L, R, O, mid, sol, toxic, r, s):
    a = _buf(96); b = _buf(192); c = _buf(96)
    lib.orc_groth16_prove_trapdoor(n, m, *L.args(), *R.args(), *O.args(), _b(bytes(mid)), _b(sol), _b(toxic), _b(r), _b(s), a, b, c)
    return bytes(a), bytes(b), bytes(c)


def groth16_setup_exponents(n, m, L, R, O, mid, toxic, want_io=True):
    nmid = int(np.asarray(mid, dtype=np.int64).sum()); nio = m - nmid
    e1 = _buf(32 * (3 + (n + 2) + max(n - 1, 0) + nmid)); e2 = _buf(32 * (2 + n + 2))
    eio = _buf(32 * max(nio, 1))
    lib.orc_groth16_setup_exponents(n, m, *L.args(), *R.args(), *O.args(), _b(bytes(mid)), _b(toxic), e1, e2, eio if want_io else None)
    return bytes(e1), bytes(e2), bytes(eio)[:32 * nio]


# ---------------------------------------------------------------- Pinocchio Protocol 2
def pinocchio_sizes(n, m, mid):
    nmid = int(np.asarray(mid, dtype=np.int64).sum()); nio = m - nmid
    return {"n_mid": nmid, "n_io": nio, "pk_g1": 5 * nmid + (n + 1) + 2 * m + 7, "pk_g2": 2 * nmid + (n + 1) + 2,
            "vk_g1": 3 + 2 * nio, "vk_g2": 6 + nio}


def pinocchio_keygen_exponents(qap_or_none, n, m, L, R, O, mid, toxic, literal):
    sz = pinocchio_sizes(n, m, mid)
    e1 = _buf(32 * sz["pk_g1"]); e2 = _buf(32 * sz["pk_g2"]); v1 = _buf(32 * sz["vk_g1"]); v2 = _buf(32 * sz["vk_g2"])
    lib.orc_pinocchio_keygen_exponents(qap_or_none.h if qap_or_none is not None else None, n, m, *L.args(), *R.args(), *O.args(),
                                       _b(bytes(mid)), _b(toxic), int(literal), e1, e2, v1, v2)
    return bytes(e1), bytes(e2), bytes(v1), bytes(v2)


def pinocchio_prove(qap, pk_g1, pk_g2, mid, sol, dv, dw, dy):
    out = _buf(960)
    rc = lib.orc_pinocchio_prove(qap.h, _b(pk_g1), _b(pk_g2), _b(bytes(mid)), _b(sol), _b(dv), _b(dw), _b(dy), out)
    return rc, bytes(out)


def pinocchio_prove_trapdoor(n, m, L, R, O, mid, sol, toxic, dv, dw, dy):
    out = _buf(960)
    lib.orc_pinocchio_prove_trapdoor(n, m, *L.args(), *R.args(), *O.args(), _b(bytes(mid)), _b(sol), _b(toxic), _b(dv), _b(dw), _b(dy), out)
    return bytes(out)


def points_of_exponents_g1(ex):
    g = g1_generator()
    return b"".join(g1_mul(g, ex[32 * i:32 * i + 32]) for i in range(len(ex) // 32))


def points_of_exponents_g2(ex):
    g = g2_generator()
    return b"".join(g2_mul(g, ex[32 * i:32 * i + 32]) for i in range(len(ex) // 32))


def pinocchio_verify(vk_g1, vk_g2, io_values, proof):
    """Verify.f (src/pinocchio/pinocchio.ml:254-420) with the Python big-int pairing: the four
    knowledge-of-coefficient checks (:285,298,311,361-366) and the divisibility check (:418-420).
    vk layouts as in orc_pinocchio_keygen_exponents; io_values = public c_k in variable order."""
    from oracle import pyref as P
    nio = len(io_values)
    G1p = lambda b: P.g1_from_bytes(b)
    G2p = lambda b: P.g2_from_bytes(b)
    one, aw, bgm = (G1p(vk_g1[96 * i:96 * i + 96]) for i in range(3))
    vv_io = [G1p(vk_g1[96 * (3 + i):96 * (4 + i)]) for i in range(nio)]
    yy_io = [G1p(vk_g1[96 * (3 + nio + i):96 * (4 + nio + i)]) for i in range(nio)]
    one2, av, ay, gm2, bgm2, yt = (G2p(vk_g2[192 * i:192 * i + 192]) for i in range(6))
    ww_io = [G2p(vk_g2[192 * (6 + i):192 * (7 + i)]) for i in range(nio)]
    vv, ww, yy, h = G1p(proof[:96]), G2p(proof[96:288]), G1p(proof[288:384]), G1p(proof[384:480])
    vavv, waww, yayy, bvwy = G1p(proof[480:576]), G2p(proof[576:768]), G1p(proof[768:864]), G1p(proof[864:960])
    neg = P.pt_neg
    ok = True
    ok &= P.pairing_product_is_one([(vv, av), (neg(vavv), one2)])                       # :285
    ok &= P.pairing_product_is_one([(aw, ww), (neg(one), waww)])                        # :298
    ok &= P.pairing_product_is_one([(yy, ay), (neg(yayy), one2)])                       # :311
    ok &= P.pairing_product_is_one([(bvwy, gm2), (neg(vv), bgm2), (neg(bgm), ww), (neg(yy), bgm2)])   # :361-366
    vio = wio = yio = None
    for c, a, b_, d in zip(io_values, vv_io, ww_io, yy_io):
        vio = P.pt_add(vio, P.pt_mul(a, c)); wio = P.pt_add(wio, P.pt_mul(b_, c)); yio = P.pt_add(yio, P.pt_mul(d, c))
    ok &= P.pairing_product_is_one([(P.pt_add(vio, vv), P.pt_add(wio, ww)), (neg(P.pt_add(yio, yy)), one2), (neg(h), yt)])   # :418-420
    return bool(ok)


# ---------------------------------------------------------------- fast multi-threaded CPU prover (oracle/fast_cpu.c: context baseline)
lib.orc_fast_groth16_new.restype = C.c_void_p


class FastGroth16:
    """Pippenger + NTT CPU prover over the Lagrange-form key (same proof bytes as the literal oracle); `threads` host threads."""

    def __init__(self, n, m, L, R, O, mid, lag_g1, lag_g2, threads=1):
        g1 = np.ascontiguousarray(np.frombuffer(bytes(lag_g1), dtype=np.uint8) if not isinstance(lag_g1, np.ndarray) else lag_g1, dtype=np.uint8).reshape(-1)
        g2 = np.ascontiguousarray(np.frombuffer(bytes(lag_g2), dtype=np.uint8) if not isinstance(lag_g2, np.ndarray) else lag_g2, dtype=np.uint8).reshape(-1)
        self.h = lib.orc_fast_groth16_new(n, m, *L.args(), *R.args(), *O.args(), _b(bytes(mid)), _b(g1), C.c_size_t(len(g1) // 96),
                                          _b(g2), C.c_size_t(len(g2) // 192), int(threads))
        if not self.h:
            raise ValueError("orc_fast_groth16_new: key does not have the Lagrange-form layout for this circuit")
        self._keep = (L, R, O)

    def prove(self, sol, r, s):
        out = _buf(384)
        rc = lib.orc_fast_groth16_prove(C.c_void_p(self.h), _b(sol), _b(r), _b(s), out)
        b = bytes(out)
        return rc, b[:96], b[96:288], b[288:]

    def close(self):
        if self.h:
            lib.orc_fast_groth16_free(C.c_void_p(self.h))
            self.h = None


def fast_g1_msm(bases, scalars, threads=1):
    o = _buf(96)
    rc = lib.orc_fast_g1_msm(o, _b(bases), _b(scalars), C.c_size_t(len(scalars) // 32), int(threads))
    return rc, bytes(o)


def fast_g2_msm(bases, scalars, threads=1):
    o = _buf(192)
    rc = lib.orc_fast_g2_msm(o, _b(bases), _b(scalars), C.c_size_t(len(scalars) // 32), int(threads))
    return rc, bytes(o)
