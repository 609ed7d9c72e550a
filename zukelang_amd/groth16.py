"""Host-side mirror of `Groth16.Make(C) : Protocol.S` (src/groth16/groth16.mli:3-6,
src/lib/zk/protocol.mli:3-28) for the prove path, backed by the HIP library.

Same names, argument meaning and error behaviour as the OCaml module:

  keygen rng circuit        -> (pkey, vkey)     groth16.ml:227-233 -> setup :45-108
  prove  rng pkey sol       -> proof            groth16.ml:235-237 -> :123-161
                                                (the qap argument of the reference is the circuit
                                                 held by the uploaded key: dense QAP.t is 3*m*n
                                                 field elements and cannot exist at 2^16+)
`rng` is any callable returning the next Fr element as a Python int; keygen draws
alpha, beta, gamma, delta, tau in that order (groth16.ml:51-55) and prove draws r then s
(groth16.ml:124-125), exactly like `Fr.gen rng` in the reference.

An unsatisfied witness raises AssertionError like `assert (Polynomial.is_zero rem)` (QAP.ml:134).
`verify` (3 pairings, groth16.ml:163-173; scope row f1) runs on the host in the library's own pairing
(csrc/pairing_host.hip), as the reference's does in its external library.
"""
import ctypes as C
from dataclasses import dataclass

import numpy as np

from . import _lib
from .curve import G1, G2, Pairing
from .r1cs import FR_MODULUS, R1CS, fr_bytes

ZK_ERR_REMAINDER = -4


def _p(arr):
    return arr.ctypes.data_as(C.POINTER(C.c_uint8))


def _csr(M):
    c = _lib.CSR()
    c.row_ptr = M.ptr.ctypes.data_as(C.POINTER(C.c_uint32))
    c.col = M.col.ctypes.data_as(C.POINTER(C.c_uint32))
    c.val = M.val.ctypes.data_as(C.POINTER(C.c_uint8))
    return c


@dataclass
class PKey:
    """groth16.ml:24-34, fields in declaration order, flattened per group:
    g1 = a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid[n_mid] (96 B each)
    g2 = b2 | d2 | ti2[n+2] (192 B each)"""
    g1: np.ndarray
    g2: np.ndarray
    # Lagrange-form extension (scope row f4; keygen(..., lagrange=True)): the same key with the tau-power lists
    # replaced by  g1 = a | d1 | b1 | [l_i(tau)]_1 (n) | [lambda_t(tau) Z(tau)/delta]_1 (n-1) | ltd_mid,
    # g2 = b2 | d2 | [l_i(tau)]_2 (n).  None for keys in the reference's format.
    lag_g1: np.ndarray = None
    lag_g2: np.ndarray = None


@dataclass
class VKey:
    """groth16.ml:36-43.  `ab` = e(alpha, beta) as a GT element (576 B, see curve.GT)."""
    one1: bytes
    ltgm_io: np.ndarray     # [L_k(tau)/gamma]_1 for k in io, Var order
    one2: bytes
    gm: bytes
    d: bytes
    ab: bytes = b""


@dataclass
class Proof:
    a: bytes   # G1 96 B
    b: bytes   # G2 192 B
    c: bytes   # G1 96 B

    def to_compressed(self):
        """The JSON form of groth16.ml:110-114 (curve.ml:199,208): 48 | 96 | 48 B."""
        return G1.to_compressed_bytes(self.a) + G2.to_compressed_bytes(self.b) + G1.to_compressed_bytes(self.c)


def _lagrange_at(n, tau):
    """l_i(tau) for the integer domain 0..n-1 and Z(tau) (QAP.ml:84,92), O(n) with one inversion."""
    P = FR_MODULUS
    fact = [1] * (n + 1)
    for i in range(1, n + 1):
        fact[i] = fact[i - 1] * i % P
    zt = 1
    den = [0] * n
    for i in range(n):
        d = (tau - i) % P
        zt = zt * d % P
        x = d * fact[i] % P * fact[n - 1 - i] % P
        den[i] = (P - x) % P if (n - 1 - i) & 1 else x
    pre = [1] * (n + 1)
    for i in range(n):
        pre[i + 1] = pre[i] * den[i] % P
    inv = pow(pre[n], P - 2, P)
    lag = [0] * n
    for i in range(n - 1, -1, -1):
        lag[i] = inv * pre[i] % P * zt % P
        inv = inv * den[i] % P
    return lag, zt


def columns_at(circuit, lag):
    """u_k = sum_g M[g][k] * lag[g] for M = L, R, O and every variable k (Python ints): v_k(tau), w_k(tau), y_k(tau) through the Lagrange basis of
    the gates' points -- the reference evaluates `Poly.apply u_k tau` on dense polynomials per variable (groth16.ml:59-68, pinocchio.ml:104-109), the
    same field elements.  Three sparse products M^T x on the GPU (zk_fr_spmv)."""
    from .r1cs import fr_ints
    x = fr_bytes(lag)
    out = []
    for M in (circuit.L, circuit.R, circuit.O):
        T = M.transposed(circuit.m)
        y = np.zeros(32 * circuit.m, dtype=np.uint8)
        _lib.check(_lib.lib().zk_fr_spmv(C.c_uint32(circuit.m), C.c_uint32(circuit.n), C.byref(_csr(T)), _p(x), _p(y)))
        out.append(fr_ints(y))
    return out


def all_gather_bytes(part, world):
    """The exchange step of the point-sharded prover: every rank contributes one
    fixed-size uint8 block, every rank receives all of them in rank order.  RCCL on the GPU box
    (backend nccl), gloo in the CPU tests.  EC addition is not an RCCL reduction operator, so this
    is an all-gather of raw bytes followed by a local EC reduction."""
    import torch
    import torch.distributed as dist
    t = torch.from_numpy(np.ascontiguousarray(part, dtype=np.uint8).copy())
    if dist.get_backend() == "nccl":
        t = t.to(torch.device("cuda", torch.cuda.current_device()))
    out = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return np.ascontiguousarray(torch.cat(out).cpu().numpy())


def clip_bounds(bounds, nz):
    """Slices [lo, hi) cut to the non-zero prefix [0, nz) of a scalar vector (the A vector of a Groth16 proof is zero beyond
    a | d1 | b1 | ti1: groth16.ml:128-134 touches no other key point): a rank whose slice lies beyond it exchanges nothing."""
    return [(min(lo, nz), min(hi, nz)) for lo, hi in bounds]


def exchange_slices(full, bounds, rank, world, out=None):
    """The exchange step of the distributed Fr stage: every rank holds the scalar vector of ONE proof
    over the whole base pool (`full`: uint8, 32 B per element) and needs slice [lo_rank, hi_rank) of EVERY
    rank's vector.  Returns `world` blocks of this rank's slice length, in rank order.  `bounds` may be clipped
    (clip_bounds): elements outside every slice are simply not shipped.
    RCCL (backend nccl): one all-to-all with per-destination splits, device to device over xGMI.
    gloo (CPU tests, single-GPU rehearsal): every rank gathers every vector and keeps its slice."""
    import torch
    import torch.distributed as dist
    lo, hi = bounds[rank]
    mine = 32 * (hi - lo)
    if dist.get_backend() == "nccl":
        if out is None:
            out = torch.empty(world * mine, dtype=torch.uint8, device=full.device)
        # input: the slices in rank order are the contiguous prefix [bounds[0].lo, bounds[-1].hi) of the vector
        first, last = 32 * bounds[0][0], 32 * bounds[-1][1]
        dist.all_to_all_single(out, full[first:last], [mine] * world, [32 * (b - a) for a, b in bounds])
        return out
    t = full.detach().cpu() if isinstance(full, torch.Tensor) else torch.from_numpy(np.ascontiguousarray(full, dtype=np.uint8).copy())
    outs = [torch.empty_like(t) for _ in range(world)]
    dist.all_gather(outs, t)
    cat = torch.cat([o[32 * lo:32 * hi] for o in outs])
    if out is None:
        return cat
    out.copy_(cat)
    return out


def agree_on_status(rc):
    """Collective error agreement of the N > 1 paths.  A failure that only ONE rank can see (the owner of a proof finds
    `p mod Z != 0`, QAP.ml:134, or a HIP error) must not let the other ranks walk into the next all-to-all / all-gather
    alone: every rank contributes its local return code (0 or a negative ZK_ERR_*), all of them receive the most severe
    one (the minimum) and raise the same exception -- or none does.  One 4-byte all-reduce; a no-op without a process group."""
    import torch
    import torch.distributed as dist
    if not (dist.is_available() and dist.is_initialized()) or dist.get_world_size() == 1:
        return rc
    dev = torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")
    t = torch.tensor([int(rc)], dtype=torch.int32, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return int(t.item())


def msm_scalar_vectors(n, v, w, h, wit, mid, r, s):
    """Host mirror (Python ints) of the library's k_groth16_scalars: the three scalar vectors laid
    over the key pools g1 = a | d1 | b1 | ti1[n+2] | tiztd[n-1] | ltd_mid and g2 = b2 | d2 | ti2[n+2]:
      A = <g1, scalA>,  C = <g1, scalC>,  B = <g2, scalB>
    with s*A + r*B1 - r*s*delta folded into C (see groth16.hip)."""
    P = FR_MODULUS
    pad = lambda x, k: list(x) + [0] * (k - len(x))
    v, w, h = pad(v, n), pad(w, n), pad(h, n - 1)
    wm = [wit[k] for k in range(len(mid)) if mid[k]]
    scalA = [1, r, 0] + v + [0, 0] + [0] * (n - 1) + [0] * len(wm)
    scalC = [s, r * s % P, r] + [(s * v[i] + r * w[i]) % P for i in range(n)] + [0, 0] + h + wm
    scalB = [1, s] + w + [0, 0]
    return scalA, scalC, scalB


def shard_bounds(size, rank, world, heavy=0):
    """Contiguous slice [lo, hi) of a base pool owned by `rank`: the library's rule (zk_groth16_shard_range), restated.  The first `heavy`
    points count twice -- in the G1 pool a | d1 | b1 | the tau basis serve the products A and C of a proof, every other point only C -- and
    the cuts sit at equal shares of size + heavy, so the ranks get equal work; heavy = 0 is the uniform cut (the G2 pool)."""
    heavy = min(heavy, size)
    total = size + heavy

    def cut(g):
        t = total * g // world
        return t // 2 if t <= 2 * heavy else t - heavy
    return cut(rank), cut(rank + 1)


class Groth16:
    @staticmethod
    def keygen(rng, circuit: R1CS, lagrange=False):
        """Groth16 setup (groth16.ml:45-108): the exponents are host integers, the points come from the
        fixed-base kernel.  L_k(tau) is evaluated through the Lagrange basis of the integer domain
        instead of Poly.apply on dense polynomials (same field element).  lagrange=True additionally emits the
        Lagrange-form pools (PKey.lag_g1 / lag_g2): only a keygen can, it needs tau."""
        P = FR_MODULUS
        a, b, gm, d, t = (rng() % P for _ in range(5))
        n, m = circuit.n, circuit.m
        lag, zt = _lagrange_at(n, t)
        dinv, ginv = pow(d, P - 2, P), pow(gm, P - 2, P)
        vk, wk, yk = columns_at(circuit, lag)
        Lk = [(b * vk[k] + a * wk[k] + yk[k]) % P for k in range(m)]          # L_k(tau) = beta v_k(tau) + alpha w_k(tau) + y_k(tau), groth16.ml:59-68
        ex1 = [a, d, b]
        ti = 1
        for _ in range(n + 2):
            ex1.append(ti)
            ti = ti * t % P
        ztd = zt * dinv % P
        ti = 1
        for _ in range(n - 1):
            ex1.append(ti * ztd % P)
            ti = ti * t % P
        ex1 += [Lk[k] * dinv % P for k in range(m) if circuit.mid[k]]
        ex2 = [b, d]
        ti = 1
        for _ in range(n + 2):
            ex2.append(ti)
            ti = ti * t % P
        exio = [Lk[k] * ginv % P for k in range(m) if not circuit.mid[k]]
        pk = PKey(G1.of_Fr(fr_bytes(ex1)), G2.of_Fr(fr_bytes(ex2)))
        if lagrange:
            lam, _ = _lagrange_at(n - 1, (t - n) % P)           # basis of the points n..2n-2, evaluated at tau
            lx1 = [a, d, b] + lag + [lam[i] * ztd % P for i in range(n - 1)] + [Lk[k] * dinv % P for k in range(m) if circuit.mid[k]]
            lx2 = [b, d] + lag
            pk.lag_g1, pk.lag_g2 = G1.of_Fr(fr_bytes(lx1)), G2.of_Fr(fr_bytes(lx2))
        vk = VKey(bytes(G1.of_Fr(fr_bytes([1]))), G1.of_Fr(fr_bytes(exio)), bytes(G2.of_Fr(fr_bytes([1]))),
                  bytes(G2.of_Fr(fr_bytes([gm]))), bytes(G2.of_Fr(fr_bytes([d]))),
                  Pairing.pairing(bytes(pk.g1[:96]), bytes(pk.g2[:192])))        # ab = e(alpha, beta), groth16.ml:103
        return pk, vk

    def __init__(self, circuit: R1CS, pkey: PKey, rank=0, world=1, lagrange=False):
        """Uploads the proving key and the circuit once (device-resident until `close`).  lagrange=True uploads the
        Lagrange-form pools of an extended key instead (single GPU): same proofs, no basis conversion per proof."""
        self.circuit = circuit
        self.rank, self.world = rank, world
        self._keep = (circuit, pkey)
        L, R, O = _csr(circuit.L), _csr(circuit.R), _csr(circuit.O)
        h = C.c_uint64()
        if lagrange and (pkey.lag_g1 is None or world != 1):
            raise ValueError("lagrange=True needs a key made by keygen(..., lagrange=True) and a single GPU")
        g1 = np.ascontiguousarray(pkey.lag_g1 if lagrange else pkey.g1, dtype=np.uint8)
        g2 = np.ascontiguousarray(pkey.lag_g2 if lagrange else pkey.g2, dtype=np.uint8)
        mid = np.ascontiguousarray(circuit.mid, dtype=np.uint8)
        lib = _lib.lib()
        if lagrange:
            rc = lib.zk_groth16_pk_upload_lagrange(C.c_uint32(circuit.n), C.c_uint32(circuit.m), C.byref(L), C.byref(R), C.byref(O), _p(mid),
                                                   _p(g1), C.c_size_t(len(g1) // 96), _p(g2), C.c_size_t(len(g2) // 192), C.byref(h))
        elif world == 1:
            rc = lib.zk_groth16_pk_upload(C.c_uint32(circuit.n), C.c_uint32(circuit.m), C.byref(L), C.byref(R), C.byref(O), _p(mid),
                                          _p(g1), C.c_size_t(len(g1) // 96), _p(g2), C.c_size_t(len(g2) // 192), C.byref(h))
        else:
            rc = lib.zk_groth16_pk_upload_sharded(C.c_uint32(circuit.n), C.c_uint32(circuit.m), C.byref(L), C.byref(R), C.byref(O),
                                                  _p(mid), _p(g1), C.c_size_t(len(g1) // 96), _p(g2), C.c_size_t(len(g2) // 192),
                                                  C.c_uint32(rank), C.c_uint32(world), C.byref(h))
        _lib.check(rc)
        self.handle = h

    def derive_lagrange(self):
        """Turns the uploaded reference-format key into its Lagrange form on the device (zk_groth16_pk_derive_lagrange: once per key,
        O(n log^2 n) scalar multiplications, no tau needed); the proofs stay byte-identical, the per-proof basis conversion disappears."""
        _lib.check(_lib.lib().zk_groth16_pk_derive_lagrange(self.handle))

    def derive_lagrange_shared(self, rank, world):
        """The derivation shared by the `world` ranks of a node (every rank holds the key whole, as uploaded): the three derived sets are
        independent, so each is derived ONCE -- set 0 ([l_i]_1) and set 2 (the h bases) on ranks 0 and 2 (both on rank 0 of a two-rank
        world), set 1 ([l_i]_2) on rank 1 -- broadcast from its owner (RCCL on device memory; gloo through the host in the CPU rehearsals)
        and installed as this rank's shard (zk_groth16_pk_install_lagrange).  Same pools, same proofs as derive_lagrange() + shard(); the
        wall time is the longest single set instead of the sum of all three on every rank."""
        import torch
        import torch.distributed as dist
        L = _lib.lib()
        if world == 1:
            return self.derive_lagrange()
        n1, n2 = C.c_uint64(), C.c_uint64()
        _lib.check(L.zk_groth16_lagrange_pool_sizes(self.handle, C.byref(n1), C.byref(n2)))
        dev = torch.device("cuda", torch.cuda.current_device())
        t1 = torch.zeros(96 * n1.value, dtype=torch.uint8, device=dev)
        t2 = torch.zeros(192 * n2.value, dtype=torch.uint8, device=dev)
        owner = [0, 1, 0] if world == 2 else [0, 1, 2]
        mask = sum(1 << s for s in range(3) if owner[s] == rank)
        rc = L.zk_groth16_pk_derive_lagrange_sets(self.handle, C.c_uint32(mask), C.c_void_p(t1.data_ptr()), C.c_void_p(t2.data_ptr()))
        worst = agree_on_status(rc)
        if worst != 0:
            if rc == worst:
                _lib.check(rc)
            raise _lib.ZkError(worst, "a peer rank failed while deriving its set of the Lagrange-form pools")
        n = self.circuit.n
        regions = [(t1, 96 * 3, 96 * (3 + n)), (t2, 192 * 2, 192 * (2 + n)), (t1, 96 * (3 + n), 96 * (3 + n + n - 1))]
        on_device = dist.get_backend() == "nccl"
        for s, (t, lo, hi) in enumerate(regions):
            if hi <= lo:
                continue
            part = t[lo:hi]
            if on_device:
                dist.broadcast(part, src=owner[s])
            else:
                host = part.cpu()
                dist.broadcast(host, src=owner[s])
                part.copy_(host)
        torch.cuda.current_stream().synchronize()
        rc = L.zk_groth16_pk_install_lagrange(self.handle, C.c_void_p(t1.data_ptr()), C.c_void_p(t2.data_ptr()), C.c_uint32(rank), C.c_uint32(world))
        worst = agree_on_status(rc)
        if worst != 0:
            if rc == worst:
                _lib.check(rc)
            raise _lib.ZkError(worst, "a peer rank failed while installing the derived pools")
        self.rank, self.world = rank, world

    def shard(self, rank, world):
        """Turns a key uploaded whole (and possibly derived into its Lagrange form) into `rank`'s shard of a point-sharded prover
        (zk_groth16_pk_shard): afterwards prove_rs / prove_async / prove_wait run the all-gather + combine of the sharded path."""
        _lib.check(_lib.lib().zk_groth16_pk_shard(self.handle, C.c_uint32(rank), C.c_uint32(world)))
        self.rank, self.world = rank, world

    def pool_points(self, group):
        """The resident base pool (1 = G1, 2 = G2) as uncompressed bytes, in pool order."""
        cnt = C.c_size_t()
        _lib.check(_lib.lib().zk_groth16_pool_points(self.handle, C.c_int(group), None, C.c_size_t(0), C.byref(cnt)))
        out = np.zeros(cnt.value * (96 if group == 1 else 192), dtype=np.uint8)
        _lib.check(_lib.lib().zk_groth16_pool_points(self.handle, C.c_int(group), _p(out), C.c_size_t(cnt.value), C.byref(cnt)))
        return out

    def close(self):
        if getattr(self, "handle", None) is not None:
            _lib.lib().zk_groth16_pk_free(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    @staticmethod
    def _sol_bytes(sol):
        if isinstance(sol, np.ndarray):
            return np.ascontiguousarray(sol, dtype=np.uint8).reshape(-1)
        if isinstance(sol, (bytes, bytearray)):
            return np.frombuffer(bytes(sol), dtype=np.uint8)
        return fr_bytes(sol)

    def prove(self, rng, sol):
        """Groth16.prove rng qap pkey sol (groth16.ml:235-237): draws r then s from rng."""
        r = rng() % FR_MODULUS
        s = rng() % FR_MODULUS
        return self.prove_rs(sol, r, s)

    def set_witness(self, sol):
        """Uploads a witness once; later prove_rs(None, r, s) calls use the HBM-resident copy."""
        w = self._sol_bytes(sol)
        if len(w) != 32 * self.circuit.m:
            raise AssertionError("Variable not found")          # var.ml:75-77
        _lib.check(_lib.lib().zk_groth16_set_witness(self.handle, _p(w)))

    def prove_rs(self, sol, r, s):
        if sol is None:
            w = None
        else:
            w = self._sol_bytes(sol)
            if len(w) != 32 * self.circuit.m:
                raise AssertionError("Variable not found")          # var.ml:75-77
        rb, sb = fr_bytes([r]), fr_bytes([s])
        out = np.zeros(384, dtype=np.uint8)
        if self.world == 1:
            rc = _lib.lib().zk_groth16_prove(self.handle, _p(w) if w is not None else None, _p(rb), _p(sb), _p(out))
        else:
            part = np.zeros(768, dtype=np.uint8)
            rc = _lib.lib().zk_groth16_prove_partial(self.handle, _p(w) if w is not None else None, _p(rb), _p(sb), _p(part))
            rc = self._exchange_and_combine(rc, part, out)
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")      # QAP.ml:134
        _lib.check(rc)
        b = bytes(out)
        return Proof(b[:96], b[96:288], b[288:])

    def reserve_slots(self, count):
        """Allocate the scratch and streams of `count` proof slots now (otherwise at first use)."""
        _lib.check(_lib.lib().zk_groth16_reserve_slots(self.handle, C.c_uint32(count)))

    def prove_async(self, sol, r, s, slot):
        """Enqueue one proof on `slot` (0..14) and return; `prove_wait(slot)` collects it.  Several
        slots keep several proofs in flight on one key.  On a sharded key the slot produces this
        rank's partial sums; prove_wait then runs the exchange (all-gather) and the combine.
        The witness buffer handed to the library stays referenced here until prove_wait(slot): the C side
        copies it with hipMemcpyAsync and must not outlive a pageable host buffer."""
        w = None
        if sol is not None:
            w = self._sol_bytes(sol)
            if len(w) != 32 * self.circuit.m:
                raise AssertionError("Variable not found")          # var.ml:75-77
        rb, sb = fr_bytes([r]), fr_bytes([s])
        fn = _lib.lib().zk_groth16_prove_async if self.world == 1 else _lib.lib().zk_groth16_prove_partial_async
        _lib.check(fn(self.handle, _p(w) if w is not None else None, _p(rb), _p(sb), C.c_uint32(slot)))
        if not hasattr(self, "_inflight"):
            self._inflight = {}
        self._inflight[slot] = (w, rb, sb)

    def prove_wait(self, slot):
        out = np.zeros(384, dtype=np.uint8)
        if self.world == 1:
            rc = _lib.lib().zk_groth16_prove_wait(self.handle, C.c_uint32(slot), _p(out))
            getattr(self, "_inflight", {}).pop(slot, None)
        else:
            part = np.zeros(768, dtype=np.uint8)
            rc = _lib.lib().zk_groth16_prove_partial_wait(self.handle, C.c_uint32(slot), _p(part))
            getattr(self, "_inflight", {}).pop(slot, None)
            rc = self._exchange_and_combine(rc, part, out)
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")      # QAP.ml:134
        _lib.check(rc)
        b = bytes(out)
        return Proof(b[:96], b[96:288], b[288:])

    def _exchange_and_combine(self, rc, part, out):
        """The collective half of a sharded proof with the Fr stage replicated: all ranks agree on the local return codes BEFORE the all-gather
        (a failure only one rank sees -- a HIP error in its MSM slice -- must not leave its peers waiting in the collective) and again after
        the combine; every rank gets the same code back and raises the same exception, or none does."""
        local = rc
        rc = agree_on_status(rc)
        if rc == 0:
            gathered = all_gather_bytes(part, self.world)
            local = _lib.lib().zk_groth16_combine(_p(gathered), C.c_uint32(self.world), _p(out))
            rc = agree_on_status(local)
        if rc != 0 and rc != ZK_ERR_REMAINDER and local != rc:
            raise _lib.ZkError(rc, "a peer rank failed in this proof's products or combine")
        return rc

    def prove_partial(self, w, rb, sb):
        part = np.zeros(768, dtype=np.uint8)
        rc = _lib.lib().zk_groth16_prove_partial(self.handle, _p(w) if w is not None else None, _p(rb), _p(sb), _p(part))
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")
        _lib.check(rc)
        return part

    def qap_eval(self, sol):
        """QAP.eval (QAP.ml:120-135): coefficient vectors (v, w, h) padded with zeros to n, n, n-1."""
        w = self._sol_bytes(sol)
        n = self.circuit.n
        v = np.zeros(32 * n, dtype=np.uint8)
        ww = np.zeros(32 * n, dtype=np.uint8)
        h = np.zeros(32 * (n - 1), dtype=np.uint8)
        rc = _lib.lib().zk_groth16_qap_eval(self.handle, _p(w), _p(v), _p(ww), _p(h))
        if rc == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")
        _lib.check(rc)
        return v, ww, h

    @staticmethod
    def verify(input_output, vkey, proof):
        """Groth16.verify (groth16.ml:163-173): e(A, B) = ab * e(sum_k w_k [L_k(tau)/gamma]_1, gamma) * e(C, delta).
        input_output: the public coefficients in the key's variable order (ints or 32-byte LE blocks).
        Host work (three pairings), as in the reference."""
        io = input_output if isinstance(input_output, (bytes, bytearray, np.ndarray)) else fr_bytes(list(input_output))
        io = np.ascontiguousarray(np.frombuffer(bytes(io), dtype=np.uint8))
        lt = np.ascontiguousarray(vkey.ltgm_io, dtype=np.uint8).reshape(-1)
        n_io = len(lt) // 96
        if len(io) != 32 * n_io:
            raise AssertionError("Variable not found")          # var.ml:75-77 / curve.ml:96-100: domains must agree
        ok = C.c_int(0)
        _lib.check(_lib.lib().zk_groth16_verify(bytes(vkey.ab), _p(lt) if n_io else None, _p(io) if n_io else None, C.c_size_t(n_io),
                                                bytes(vkey.gm), bytes(vkey.d), bytes(proof.a) + bytes(proof.b) + bytes(proof.c), C.byref(ok)))
        return bool(ok.value)



class GroupProver:
    """N > 1 with a DISTRIBUTED Fr stage.  Proofs are handled in rounds of `batch`; proof number i of the job belongs
    to rank i % world, which alone runs QAP.eval (QAP.ml:120-135) for it and builds its three scalar vectors; one
    all-to-all per vector hands rank g its slice [lo_g, hi_g) of the scalars of the k-th proof of every owner
    (k = 0, 1, ...: ceil(batch / world) exchanges per round); every rank then runs the point-sharded MSMs
    (groth16.ml:116-161) of all proofs of the round over its slice (that many proofs in flight, one slot each), the
    768-byte partial sums of the round travel in one all-gather, and zk_groth16_combine finishes each proof.
    Per proof and rank that is 1/world of an Fr stage plus one sharded MSM triple -- with the replicated Fr stage of
    prove_async / prove_wait on a sharded key it is a whole Fr stage.
    The rounds are software-pipelined over TWO sets of MSM slots: the products of round i are enqueued before those of round
    i - 1 are waited for (and gathered, combined), and the Fr stages of round i + 1 are enqueued before either, so the GPU never
    drains between rounds (one rank's share timed on one GPU, scripts/proto/group_emulation.py: 2.05-2.27 ms per proof at N = 2
    with drained rounds).  Ownership rotates with the running proof count, so the ranks stay balanced when `batch` is not a
    multiple of `world`."""

    MAX_SLOTS = 15

    def __init__(self, prover, batch=None):
        import torch
        self.torch = torch
        self.p = prover
        self.world, self.rank = prover.world, prover.rank
        W = self.world
        if batch is None:
            # two sets of MSM slots + Fr slots = 2 batch + ceil(batch / W) streams <= 14: the chip runs 16 hardware queues side by
            # side and RCCL / the framework need some
            batch = 7
            while 2 * batch + (batch + W - 1) // W > 14:
                batch -= 1
        self.batch = batch                          # proofs per round; MSM slots [0, batch) and [batch, 2 batch) alternate between rounds
        self.K = (batch + W - 1) // W               # Fr stages a rank runs per round at most; Fr slots 2 batch .. 2 batch + K - 1
        self.fr0 = 2 * batch
        if self.batch < 1 or self.fr0 + self.K > self.MAX_SLOTS:
            raise ValueError("GroupProver: 2 batch + ceil(batch / world) must not exceed %d slots" % self.MAX_SLOTS)
        L = _lib.lib()
        v = [C.c_uint64() for _ in range(6)]
        _lib.check(L.zk_groth16_pool_layout(prover.handle, *[C.byref(x) for x in v]))
        self.p1, self.p2, self.lo1, self.hi1, self.lo2, self.hi2 = (int(x.value) for x in v)
        self.bounds1 = [shard_bounds(self.p1, g, W, self.p2 + 1) for g in range(W)]       # p2 + 1 = a | d1 | b1 | the tau basis: the A prefix
        self.bounds2 = [shard_bounds(self.p2, g, W) for g in range(W)]
        assert self.bounds1[self.rank] == (self.lo1, self.hi1) and self.bounds2[self.rank] == (self.lo2, self.hi2)
        dev = torch.device("cuda", torch.cuda.current_device())
        u8 = dict(dtype=torch.uint8, device=dev)
        self.len1, self.len2 = 32 * (self.hi1 - self.lo1), 32 * (self.hi2 - self.lo2)
        # the A vector is zero beyond a | d1 | b1 | ti1[n+2] (or the n Lagrange points): only that prefix needs to travel (-28 % volume) ...
        self.nzA = self.p2 + 1                       # 3 + (n + 2) tau powers, or 3 + n Lagrange points
        self.boundsA = clip_bounds(self.bounds1, self.nzA)
        # ... unless that leaves a rank with an EMPTY slice (with the equal-work cuts it does from two ranks up: the prefix ends inside
        # the first ranks' slices).  An all-to-all with zero-length blocks and an empty landing tensor is a corner of RCCL / the framework
        # this code has never met on hardware, and a rank that fails there alone would leave its peers hanging in the collective: the A
        # vector then travels whole, like C (+50 % exchange volume, ~3 % of a round).  The same decision on every rank.
        self.clipA = all(hi > lo for lo, hi in self.boundsA)
        if not self.clipA:
            self.boundsA = self.bounds1
        self.lenA = 32 * (self.boundsA[self.rank][1] - self.boundsA[self.rank][0])
        # per owned proof of the round: the owner's full vectors A, C, B; the received slices [world][slice], one set per slot set
        # (the products of round i - 1 still read theirs while the exchange of round i lands)
        self.full = [[torch.zeros(32 * self.p1, **u8), torch.zeros(32 * self.p1, **u8), torch.zeros(32 * self.p2, **u8)] for _ in range(self.K)]
        self.recv = [[[torch.zeros(W * self.len1, **u8), torch.zeros(W * self.len1, **u8), torch.zeros(W * self.len2, **u8)] for _ in range(self.K)] for _ in range(2)]
        self.recvA = [torch.zeros(W * self.lenA if self.clipA else 0, **u8) for _ in range(self.K)]      # compact landing buffer of the clipped A slices
        # the partial sums of a round: [proof][768] on this rank, [rank][proof][768] after the all-gather (RCCL path: device resident)
        self.parts_dev = torch.zeros(self.batch * 768, **u8)
        self.gathered_dev = torch.zeros(W * self.batch * 768, **u8)
        self.count = 0                              # proofs handled so far: proof i of the job belongs to rank i % world
        self.rounds_done = 0                        # parity of the slot set / landing buffers the next round uses
        prover.reserve_slots(self.fr0 + self.K)
        torch.cuda.current_stream().synchronize()

    def _raise_together(self, local, what):
        """All ranks exchange their worst local return code; every rank raises (its own error, or a note that a peer failed) or none does."""
        worst = agree_on_status(local)
        if worst == 0:
            return
        if local == worst:
            _lib.check(worst)
        raise _lib.ZkError(worst, "a peer rank failed in %s" % what)

    def _launch_fr(self, rnd, base):
        """rnd: list of (r, s), at most `batch`; proof t of the round belongs to rank (base + t) % world.
        Returns the indices k of the Fr slots launched."""
        launched = []
        t = (self.rank - base) % self.world
        k = 0
        while t < len(rnd):
            rb, sb = fr_bytes([rnd[t][0]]), fr_bytes([rnd[t][1]])
            f = self.full[k]
            _lib.check(_lib.lib().zk_groth16_scalars_async(self.p.handle, None, _p(rb), _p(sb), C.c_uint32(self.fr0 + k),
                                                           C.c_void_p(f[0].data_ptr()), C.c_void_p(f[1].data_ptr()), C.c_void_p(f[2].data_ptr())))
            launched.append(k)
            k += 1
            t += self.world
        return launched

    def _finish_fr_and_exchange(self, launched, count, recv):
        worst = 0
        for k in launched:                       # wait for EVERY slot launched (none stays busy), keep the most severe code
            worst = min(worst, _lib.lib().zk_groth16_scalars_wait(self.p.handle, C.c_uint32(self.fr0 + k)))
        local = worst
        worst = agree_on_status(worst)           # before the first collective of the round: all ranks raise, or none
        if worst == ZK_ERR_REMAINDER:
            raise AssertionError("Polynomial.is_zero rem")      # QAP.ml:134 (on every rank, whichever rank owned the proof)
        if worst != 0:
            if local == worst:
                _lib.check(worst)
            raise _lib.ZkError(worst, "a peer rank failed in the Fr stage of this round")
        for k in range((count + self.world - 1) // self.world):       # every rank takes part in every exchange of the round
            if self.clipA:
                exchange_slices(self.full[k][0], self.boundsA, self.rank, self.world, out=self.recvA[k])
                # [world][lenA] -> the head of each owner's [len1] block; the tail stays zero (never written)
                recv[k][0].view(self.world, self.len1)[:, :self.lenA].copy_(self.recvA[k].view(self.world, self.lenA))
            else:
                exchange_slices(self.full[k][0], self.bounds1, self.rank, self.world, out=recv[k][0])
            for i, bounds in ((1, self.bounds1), (2, self.bounds2)):
                exchange_slices(self.full[k][i], bounds, self.rank, self.world, out=recv[k][i])
        self.torch.cuda.current_stream().synchronize()        # the slices have landed before the library's streams read them

    def prove_many(self, rs_list, combine_all=True):
        """All ranks call this with the same list of (r, s); the witness is the resident one (set_witness).
        Returns the proofs in order -- all of them on every rank, or with combine_all=False only the ones
        this rank owns (None elsewhere): each proof then costs ONE combine in the job instead of one per rank."""
        L = _lib.lib()
        W = self.world
        rounds = [rs_list[i:i + self.batch] for i in range(0, len(rs_list), self.batch)]
        bases, c = [], self.count
        for rnd in rounds:
            bases.append(c % W)
            c += len(rnd)
        proofs = []

        def collect(cnt, base, slot0):
            """wait for the products of one round, all-gather its partial sums, combine.  Over RCCL the 768-byte blocks never leave the GPUs:
            slot buffer -> device tensor -> all_gather_into_tensor -> zk_groth16_combine_device; over gloo (CPU tests, rehearsal on a shared
            GPU) they travel as host bytes.  The ranks agree on the worst return code before the all-gather and after the combines."""
            import torch.distributed as dist
            torch = self.torch
            on_device = dist.get_backend() == "nccl"
            worst = 0
            if on_device:
                parts = self.parts_dev
                for t in range(cnt):
                    worst = min(worst, L.zk_groth16_prove_partial_wait_device(self.p.handle, C.c_uint32(slot0 + t), C.c_void_p(parts.data_ptr() + 768 * t)))
            else:
                parts = np.zeros((self.batch, 768), dtype=np.uint8)
                for t in range(cnt):
                    worst = min(worst, L.zk_groth16_prove_partial_wait(self.p.handle, C.c_uint32(slot0 + t), _p(parts[t])))
            self._raise_together(worst, "the products of this round")
            if on_device:
                dist.all_gather_into_tensor(self.gathered_dev, parts)            # [rank][proof][768], device to device over xGMI
                torch.cuda.current_stream().synchronize()
            else:
                gathered = all_gather_bytes(parts.reshape(-1), W).reshape(W, self.batch, 768)        # [rank][proof]
            worst = 0
            for t in range(cnt):
                if not combine_all and (base + t) % W != self.rank:
                    proofs.append(None)
                    continue
                out = np.zeros(384, dtype=np.uint8)
                if on_device:
                    rc = L.zk_groth16_combine_device(C.c_void_p(self.gathered_dev.data_ptr() + 768 * t), C.c_size_t(768 * self.batch), C.c_uint32(W), _p(out))
                else:
                    blk = np.ascontiguousarray(gathered[:, t, :]).reshape(-1)
                    rc = L.zk_groth16_combine(_p(blk), C.c_uint32(W), _p(out))
                worst = min(worst, rc)
                b = bytes(out)
                proofs.append(Proof(b[:96], b[96:288], b[288:]))
            self._raise_together(worst, "the combine of this round")

        # what is enqueued and not yet waited for -- on ANY exception below (all of them are raised on every rank together, _raise_together /
        # agree_on_status) these slots are drained so that none stays busy and the prover remains usable; their results are dropped everywhere
        state = {"fr": [], "msm": []}           # Fr slots launched (indices k); MSM rounds in flight: (count, first slot)

        def drain():
            scratch = np.zeros(768, dtype=np.uint8)
            for k in state["fr"]:
                L.zk_groth16_scalars_wait(self.p.handle, C.c_uint32(self.fr0 + k))
            for cnt_, slot0_ in state["msm"]:
                for t in range(cnt_):
                    L.zk_groth16_prove_partial_wait(self.p.handle, C.c_uint32(slot0_ + t), _p(scratch))
            state["fr"], state["msm"] = [], []

        try:
            state["fr"] = self._launch_fr(rounds[0], bases[0]) if rounds else []
            pending = None                          # the round whose products are in flight: (count, base, first slot)
            for ri, rnd in enumerate(rounds):
                cnt, base = len(rnd), bases[ri]
                half = self.rounds_done & 1
                rv_set = self.recv[half]
                launched, state["fr"] = state["fr"], []          # _finish_fr_and_exchange waits for every slot it is given, whatever the codes
                self._finish_fr_and_exchange(launched, cnt, rv_set)
                if ri + 1 < len(rounds):
                    state["fr"] = self._launch_fr(rounds[ri + 1], bases[ri + 1])      # overlaps with this round's and the previous round's products
                slot0 = half * self.batch
                state["msm"].append((0, slot0))
                worst = 0
                for t in range(cnt):
                    k, owner = t // W, (base + t) % W
                    rv = rv_set[k]
                    # a launch failure only THIS rank sees (OOM, a HIP error in its slice) must not raise here alone: its peers would wait in the next
                    # collective until it times out -- keep the code, stop launching, and let all ranks raise together below
                    rc = L.zk_groth16_msm_partial_async(self.p.handle, C.c_uint32(slot0 + t), C.c_void_p(rv[0].data_ptr() + owner * self.len1),
                                                        C.c_void_p(rv[1].data_ptr() + owner * self.len1), C.c_void_p(rv[2].data_ptr() + owner * self.len2))
                    if rc != 0:
                        worst = min(worst, rc)
                        break
                    state["msm"][-1] = (t + 1, slot0)
                self._raise_together(worst, "launching the products of this round")
                self.rounds_done += 1
                if pending is not None:
                    state["msm"].pop(0)             # collect waits for every slot of that round itself
                    collect(*pending)               # the previous round: its products ran while this round's Fr stages and exchange did
                pending = (cnt, base, slot0)
            if pending is not None:
                state["msm"].pop(0)
                collect(*pending)
        except BaseException:
            drain()
            raise
        self.count = c
        return proofs
