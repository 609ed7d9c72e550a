#!/usr/bin/env python3
"""Round 5 A/B on one box: Pinocchio ZK prove (BASELINE config 5) with the compact h pool (default) and with the reference's full pool
(ZK_PIN_COMPACT_H=0), alternating, through bench.py's own bench_pinocchio (parity gate on).  usage: r05_pinocchio_ab.py [log_n] [rounds] [inflight]"""
import argparse, importlib.util, json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
spec = importlib.util.spec_from_file_location("bench", os.path.join(ROOT, "bench.py"))
bench = importlib.util.module_from_spec(spec)
spec.loader.exec_module(bench)
from zukelang_amd import _lib
L = _lib.lib()
_lib.check(L.zk_init(0))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 18
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 2
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 8
args = argparse.Namespace(no_live_events=False, no_parity_gate=False, derive_lagrange_upto=22, no_cpu_baseline=True)
for r in range(rounds):
    for compact in (1, 0):
        _lib.check(L.zk_set_option(b"ZK_PIN_COMPACT_H", str(compact).encode()))
        res = bench.bench_pinocchio(args, L, _lib, log_n, 48, inflight)
        k = res["kernel_ms_per_proof"]
        print(json.dumps({"compact": compact, "log_n": log_n, "h_pool_points": res["h_pool_points"], "value_derived": round(res["value"] / 1e6, 2), "ms_per_proof": round(res["ms_per_proof"], 3),
                          "latency_ms": round(res["single_proof_latency_ms"], 2), "as_uploaded": round(res["as_uploaded"]["value"] / 1e6, 2), "as_uploaded_ms": round(res["as_uploaded"]["ms_per_proof"], 3),
                          "derive_s": res["derive_lagrange_s"], "parity": res["parity"] is not None,
                          "alone_ms": {q: k[q] for q in ("msm_accumulate_g1", "msm_accumulate_g2", "msm_sort", "msm_reduce") if q in k}}), flush=True)
