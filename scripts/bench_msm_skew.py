#!/usr/bin/env python3
"""the "boolean-heavy" scalar distribution (90 % of the scalars in {0, 1}), reported separately from
the headline.  The three resident-key MSMs of a Groth16 proof are run on caller-supplied scalar vectors
(zk_groth16_msm_partial_async) with 12 proofs in flight: uniform 255-bit scalars against boolean-heavy ones.
PARITY: every timed distribution is checked -- the three sums of the resident-key path (window tables, one bucket set, LDS sort) must equal
G.apply_powers over the same pools and scalars (zk_msm_g1 / zk_msm_g2: classic per-window buckets, another code path; that path is held to
the oracle's naive fold of curve.ml:112-118 on boolean-heavy inputs by tests/test_gpu_msm.py).  A mismatch aborts: no figure is printed for wrong sums.
Usage: python scripts/bench_msm_skew.py [log_n]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "32")
import numpy as np
from zukelang_amd import _lib, r1cs as RC
from zukelang_amd.groth16 import Groth16, _p
from zukelang_amd.curve import G1, G2
L = _lib.lib(); _lib.check(L.zk_init(0))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
n = 1 << log_n
cs, w = RC.iterated_cubic(n, next(RC.fr_stream(1)))
st = RC.fr_stream(2); rng = lambda: next(st)
pk, _ = Groth16.keygen(rng, cs)
pr = Groth16(cs, pk)
depth = 12
pr.reserve_slots(depth)
v = [C.c_uint64() for _ in range(6)]
_lib.check(L.zk_groth16_pool_layout(pr.handle, *[C.byref(x) for x in v]))
p1, p2 = int(v[0].value), int(v[1].value)
def dbuf(host):
    p = C.c_void_p(); _lib.check(L.zk_device_malloc(C.c_size_t(len(host)), C.byref(p)))
    _lib.check(L.zk_device_memcpy(p, host.ctypes.data_as(C.c_void_p), C.c_size_t(len(host)))); return p
def scalars(count, boolean_share, seed):
    r = np.random.default_rng(seed)
    s = r.integers(0, 256, size=(count, 32), dtype=np.uint8); s[:, 31] &= 0x3F      # < 2^254 < r
    if boolean_share:
        pick = r.random(count) < boolean_share
        s[pick] = 0
        s[pick, 0] = r.integers(0, 2, size=int(pick.sum()), dtype=np.uint8)
    return np.ascontiguousarray(s.reshape(-1))
part = np.zeros(768, dtype=np.uint8)
out = {}
pool1, pool2 = pr.pool_points(1), pr.pool_points(2)


def check(name, sA, sC, sB):
    """the last partial sums in `part` (one rank = the whole sums) against the classic-window MSM entry points, and the oracle for small n"""
    got = np.zeros(384, dtype=np.uint8)
    _lib.check(L.zk_groth16_combine(_p(part), C.c_uint32(1), _p(got)))
    a, b, c = bytes(got[:96]), bytes(got[96:288]), bytes(got[288:])
    ea, ec, eb = bytes(G1.apply_powers(sA, pool1)), bytes(G1.apply_powers(sC, pool1)), bytes(G2.apply_powers(sB, pool2))
    if (a, b, c) != (ea, eb, ec):
        raise SystemExit("PARITY FAILURE (%s): resident-key sums differ from G.apply_powers over the same pools and scalars" % name)
    how = "== zk_msm_g1 / zk_msm_g2 (classic windows) on the same pools and scalars"
    return how
for name, share in (("uniform", 0.0), ("boolean_heavy_90pct", 0.9)):
    hA, hC, hB = scalars(p1, share, 1), scalars(p1, share, 2), scalars(p2, share, 3)
    dA, dC, dB = dbuf(hA), dbuf(hC), dbuf(hB)
    def run(count):
        for i in range(count):
            if i >= depth: _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(i % depth), _p(part)))
            _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(i % depth), dA, dC, dB))
        for i in range(max(0, count - depth), count): _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(i % depth), _p(part)))
    run(2 * depth); _lib.check(L.zk_sync())
    K = 4 * depth
    t0 = time.perf_counter(); run(K); _lib.check(L.zk_sync())
    dt = (time.perf_counter() - t0) / K
    # per-family kernel time of ONE un-overlapped proof (HIP events around every family)
    _lib.check(L.zk_profile_reset()); _lib.check(L.zk_profile_enable(2))
    for i in range(3):
        _lib.check(L.zk_groth16_msm_partial_async(pr.handle, C.c_uint32(0), dA, dC, dB))
        _lib.check(L.zk_groth16_prove_partial_wait(pr.handle, C.c_uint32(0), _p(part)))
    buf = C.create_string_buffer(4096); _lib.check(L.zk_profile_names(buf, 4096))
    fam = {}
    for nm in buf.value.decode().split(","):
        if nm:
            ms, cnt = C.c_double(), C.c_uint64(); _lib.check(L.zk_profile_get(nm.encode(), C.byref(ms), C.byref(cnt))); fam[nm] = round(ms.value / 3, 3)
    _lib.check(L.zk_profile_enable(0))
    out[name] = {"ms_per_proof_msm_stage": dt * 1e3, "constraints_per_s_msm_stage": n / dt, "kernel_ms_one_proof": fam, "parity": check(name, hA, hC, hB)}
    for d in (dA, dC, dB): _lib.check(L.zk_device_free(d))
print(json.dumps({"workload": "the three resident-key MSMs of a Groth16 proof (A and C over 3n+4 G1 points with ALL scalars non-trivial, B over n+4 G2 points), %d in flight; no Fr stage" % depth,
                  "constraints": n, **out}))
