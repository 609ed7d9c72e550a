#!/usr/bin/env python3
"""BASELINE config 3: Groth16 at 2^20 constraints, Pippenger window-size sweep on one MI355X.
Runs bench.py once per window size (ZK_MSM_WINDOW); every row passes bench.py's PARITY GATE (the last timed proof equals the
oracle's trapdoor evaluation) or the row says FAILED.  Windows above 16 bits (more than 2^15 bucket counters) run the two-level
counting sort and the grouped digit weights.  Prints a table and writes gpurun_out/window_sweep.json.
Usage: python scripts/window_sweep.py [log_n] [c ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cs = [int(x) for x in sys.argv[2:]] or [12, 13, 14, 15, 16, 17, 18, 19, 20]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
rows = []
print("%4s %12s %10s %8s  %s" % ("c", "constr/s", "ms/proof", "parity", "kernel ms per proof (one proof in flight)"))
for c in cs:
    env = dict(os.environ, ZK_MSM_WINDOW=str(c))      # (the per-kernel column comes from bench.py's un-overlapped one-proof pass)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", str(log_n), "--steps", "1", "--proofs-per-step", "12", "--warmup", "0",
                          "--inflight", "6", "--settle", "2", "--headline-only", "--no-cpu-baseline", "--derive-lagrange-upto", "-1"], env=env, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(c, "FAILED", out.stderr[-300:])
        rows.append({"window_bits": c, "failed": out.stderr[-300:]})
        continue
    d = json.loads(line[-1])
    k = d["kernel_ms_per_proof"]
    rows.append({"window_bits": c, "windows": 255 // c + 1, "buckets": 1 << (c - 1), "value": d["value"], "ms_per_proof": d["ms_per_proof"],
                 "single_proof_latency_ms": d["single_proof_latency_ms"], "parity": d["parity"], "kernel_ms_per_proof": k})
    print("%4d %12.0f %10.2f %8s  acc_g1 %.2f acc_g2 %.2f reduce %.2f sort %.2f fr %.2f" % (
        c, d["value"], d["ms_per_proof"], "ok" if d["parity"] else "-", k.get("msm_accumulate_g1", 0), k.get("msm_accumulate_g2", 0), k.get("msm_reduce", 0),
        k.get("msm_sort", 0), k.get("fr_tree", 0) + k.get("fr_newton", 0) + k.get("fr_quotient", 0)))
    sys.stdout.flush()
json.dump({"workload": "groth16_prove 2^%d, iterated-cubic R1CS, 6 proofs in flight, 12 timed proofs per row" % log_n, "rows": rows},
          open(os.path.join(ROOT, "gpurun_out", "window_sweep.json"), "w"), indent=1)
