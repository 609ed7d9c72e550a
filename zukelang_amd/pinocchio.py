"""Host-side mirror of `Pinocchio.Make(C).{NonZK, ZK} : Protocol.S` (src/pinocchio/pinocchio.mli:3-15)
for the prove path, backed by the HIP library.

  keygen rng circuit  -> (pkey, vkey)   pinocchio.ml:530-534 -> KeyGen.generate :77-189
                                         (rng draws rv, rw, s, av, aw, ay, b, gm in that order, :83-91)
  ZK.prove rng sol    -> proof           :559-561 -> ZKCompute.f :427-514 (draws dv, dw, dy, :428-430)
  NonZK.prove _ sol   -> proof           :536-538 -> Compute.f :210-248 (no randomness)

Key and proof layouts are those of include/zkmi355x.h.  An unsatisfied witness raises AssertionError
(QAP.ml:134).  `verify` (13 pairings, :254-420) is outside the accelerated path (scope row f1).
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .curve import G1, G2
from .groth16 import _csr, _lagrange_at, _p, columns_at, ZK_ERR_REMAINDER
from .r1cs import FR_MODULUS, R1CS, fr_bytes


@dataclass
class PKey:
    g1: np.ndarray   # vv | yy | vav | yay | bvwy | si | v_all | w_all | vt | yt | vavt | yayt | vbt | wbt | ybt
    g2: np.ndarray   # ww | waw | si2 | wt | wawt


@dataclass
class VKey:
    g1: np.ndarray   # one | aw | bgm | vv_io | yy_io
    g2: np.ndarray   # one2 | av | ay | gm2 | bgm2 | yt | ww_io


@dataclass
class Proof:
    """Compute.proof (pinocchio.ml:195-208), fields in declaration order."""
    vv: bytes
    ww: bytes
    yy: bytes
    h: bytes
    vavv: bytes
    waww: bytes
    yayy: bytes
    bvwy: bytes

    def to_bytes(self):
        return self.vv + self.ww + self.yy + self.h + self.vavv + self.waww + self.yayy + self.bvwy

    def to_compressed(self):
        c1, c2 = G1.to_compressed_bytes, G2.to_compressed_bytes
        return c1(self.vv) + c2(self.ww) + c1(self.yy) + c1(self.h) + c1(self.vavv) + c2(self.waww) + c1(self.yayy) + c1(self.bvwy)


def _uks(circuit, s):
    """v_k(s), w_k(s), y_k(s) for every variable and t(s), through the Lagrange basis of 0..n-1."""
    lag, t = _lagrange_at(circuit.n, s)
    vk, wk, yk = columns_at(circuit, lag)          # three sparse products M^T x on the GPU (zk_fr_spmv)
    return vk, wk, yk, t


def keygen(rng, circuit: R1CS):
    """KeyGen.generate (pinocchio.ml:77-189): exponents on the host, points from the fixed-base kernel."""
    P = FR_MODULUS
    rv, rw, s, av, aw, ay, b, gm = (rng() % P for _ in range(8))
    ry = rv * rw % P
    vk_, wk_, yk_, t = _uks(circuit, s)
    mids = [k for k in range(circuit.m) if circuit.mid[k]]
    ios = [k for k in range(circuit.m) if not circuit.mid[k]]
    n = circuit.n
    si = [pow(s, i, P) for i in range(n + 1)]
    vt, wt, yt = rv * t % P, rw * t % P, ry * t % P
    e1 = ([rv * vk_[k] % P for k in mids] + [ry * yk_[k] % P for k in mids] + [rv * vk_[k] * av % P for k in mids]
          + [ry * yk_[k] * ay % P for k in mids] + [(rv * vk_[k] + rw * wk_[k] + ry * yk_[k]) * b % P for k in mids]
          + si + vk_ + wk_ + [vt, yt, vt * av % P, yt * ay % P, vt * b % P, wt * b % P, yt * b % P])
    e2 = [rw * wk_[k] % P for k in mids] + [rw * wk_[k] * aw % P for k in mids] + si + [wt, wt * aw % P]
    v1 = [1, aw, gm * b % P] + [rv * vk_[k] % P for k in ios] + [ry * yk_[k] % P for k in ios]
    v2 = [1, av, ay, gm, gm * b % P, yt] + [rw * wk_[k] % P for k in ios]
    return (PKey(G1.of_Fr(fr_bytes(e1)), G2.of_Fr(fr_bytes(e2))), VKey(G1.of_Fr(fr_bytes(v1)), G2.of_Fr(fr_bytes(v2))))


class _Prover:
    def __init__(self, circuit: R1CS, pkey: PKey):
        self.circuit = circuit
        self._keep = (circuit, pkey)
        L, R, O = _csr(circuit.L), _csr(circuit.R), _csr(circuit.O)
        h = C.c_uint64()
        g1 = np.ascontiguousarray(pkey.g1, dtype=np.uint8)
        g2 = np.ascontiguousarray(pkey.g2, dtype=np.uint8)
        mid = np.ascontiguousarray(circuit.mid, dtype=np.uint8)
        _lib.check(_lib.lib().zk_pinocchio_pk_upload(C.c_uint32(circuit.n), C.c_uint32(circuit.m), C.byref(L), C.byref(R), C.byref(O),
                                                    _p(mid), _p(g1), C.c_size_t(len(g1) // 96), _p(g2), C.c_size_t(len(g2) // 192), C.byref(h)))
        self.handle = h

    def derive_lagrange(self):
        """zk_pinocchio_pk_derive_lagrange: the h pool rewritten for the values of h (bases derived from the key's own powers si on the
        device, once); afterwards every proof skips the basis conversion.  Proof bytes do not change.  Pool 5 becomes
        [lambda_t(s)] (n-1) | [Z(s)] | [1] | [s^(n-1)] (compact, the default) or [lambda_t(s)] | [Z(s)] | [1] | v_all | w_all."""
        _lib.check(_lib.lib().zk_pinocchio_pk_derive_lagrange(self.handle))

    def pool_points(self, pool):
        """A resident base pool as uncompressed bytes, in pool order: 0..5 the G1 products (5 = the h pool), 6..7 the G2 products."""
        cnt = C.c_size_t()
        _lib.check(_lib.lib().zk_pinocchio_pool_points(self.handle, C.c_int(pool), None, C.c_size_t(0), C.byref(cnt)))
        out = np.zeros(cnt.value * (96 if pool < 6 else 192), dtype=np.uint8)
        _lib.check(_lib.lib().zk_pinocchio_pool_points(self.handle, C.c_int(pool), _p(out), C.c_size_t(cnt.value), C.byref(cnt)))
        return out

    def pool_size(self, pool):
        """Points in a resident pool.  Pool 5 tells the key's form: n + 1 + 2m = si | v_all | w_all as pinocchio.ml:481-486 reads them, n + 1 = the
        compact h pool (v_all | w_all passed the upload's consistency check and ride on si: csrc/pinocchio.hip), n + 2 = its derived form."""
        cnt = C.c_size_t()
        _lib.check(_lib.lib().zk_pinocchio_pool_points(self.handle, C.c_int(pool), None, C.c_size_t(0), C.byref(cnt)))
        return cnt.value

    def close(self):
        if getattr(self, "handle", None) is not None:
            _lib.lib().zk_pinocchio_pk_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def prove_with(self, sol, dv, dw, dy):
        w = np.ascontiguousarray(sol, dtype=np.uint8).reshape(-1) if isinstance(sol, np.ndarray) else (
            np.frombuffer(bytes(sol), dtype=np.uint8) if isinstance(sol, (bytes, bytearray)) else fr_bytes(sol))
        if len(w) != 32 * self.circuit.m:
            raise AssertionError("Variable not found")          # var.ml:75-77
        out = np.zeros(960, dtype=np.uint8)
        d = [fr_bytes([x]) for x in (dv, dw, dy)]
        rc = _lib.lib().zk_pinocchio_prove(self.handle, _p(w), _p(d[0]), _p(d[1]), _p(d[2]), _p(out))
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")      # QAP.ml:134
        _lib.check(rc)
        b = bytes(out)
        return Proof(b[:96], b[96:288], b[288:384], b[384:480], b[480:576], b[576:768], b[768:864], b[864:960])


    # ---- pipelined form (several proofs in flight on one key, witness resident in HBM)
    @staticmethod
    def _proof_of(out):
        b = bytes(out)
        return Proof(b[:96], b[96:288], b[288:384], b[384:480], b[480:576], b[576:768], b[768:864], b[864:960])

    def set_witness(self, sol):
        w = np.ascontiguousarray(sol, dtype=np.uint8).reshape(-1) if isinstance(sol, np.ndarray) else (
            np.frombuffer(bytes(sol), dtype=np.uint8) if isinstance(sol, (bytes, bytearray)) else fr_bytes(sol))
        if len(w) != 32 * self.circuit.m:
            raise AssertionError("Variable not found")          # var.ml:75-77
        _lib.check(_lib.lib().zk_pinocchio_set_witness(self.handle, _p(w)))

    def reserve_slots(self, count):
        _lib.check(_lib.lib().zk_pinocchio_reserve_slots(self.handle, C.c_uint32(count)))

    def prove_async(self, dv, dw, dy, slot):
        d = [fr_bytes([x]) for x in (dv, dw, dy)]
        _lib.check(_lib.lib().zk_pinocchio_prove_async(self.handle, None, _p(d[0]), _p(d[1]), _p(d[2]), C.c_uint32(slot)))

    def prove_wait(self, slot):
        out = np.zeros(960, dtype=np.uint8)
        rc = _lib.lib().zk_pinocchio_prove_wait(self.handle, C.c_uint32(slot), _p(out))
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")      # QAP.ml:134
        _lib.check(rc)
        return self._proof_of(out)


def verify(input_output, vkey: VKey, proof: Proof):
    """Verify.f (pinocchio.ml:254-420): the four knowledge-of-coefficient checks and the divisibility check,
    13 pairings on the host.  input_output: the public coefficients c_k in the key's variable order."""
    io = input_output if isinstance(input_output, (bytes, bytearray, np.ndarray)) else fr_bytes(list(input_output))
    io = np.ascontiguousarray(np.frombuffer(bytes(io), dtype=np.uint8))
    n_io = len(io) // 32
    g1 = np.ascontiguousarray(vkey.g1, dtype=np.uint8).reshape(-1)
    g2 = np.ascontiguousarray(vkey.g2, dtype=np.uint8).reshape(-1)
    if len(g1) != 96 * (3 + 2 * n_io) or len(g2) != 192 * (6 + n_io):
        raise AssertionError("Variable not found")          # domains of the key maps and of the public inputs must agree
    ok = C.c_int(0)
    _lib.check(_lib.lib().zk_pinocchio_verify(_p(g1), _p(g2), _p(io) if n_io else None, C.c_size_t(n_io), proof.to_bytes(), C.byref(ok)))
    return bool(ok.value)


class ZK(_Prover):
    keygen = staticmethod(keygen)
    verify = staticmethod(verify)

    def prove(self, rng, sol):
        dv = rng() % FR_MODULUS     # pinocchio.ml:428-430: dv, dw, dy in this order
        dw = rng() % FR_MODULUS
        dy = rng() % FR_MODULUS
        return self.prove_with(sol, dv, dw, dy)


class NonZK(_Prover):
    keygen = staticmethod(keygen)
    verify = staticmethod(verify)

    def prove(self, _rng, sol):
        return self.prove_with(sol, 0, 0, 0)
