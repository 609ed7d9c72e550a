#!/usr/bin/env python3
"""BASELINE config 3: Groth16 at 2^20 constraints, Pippenger window-size sweep on one MI355X.
Runs bench.py once per window size (ZK_MSM_WINDOW) and prints a table; the JSON lines go to
gpurun_out/sweep_c<bits>.json.  Usage: python scripts/window_sweep.py [log_n] [c ...]"""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
log_n = int(sys.argv[1]) if len(sys.argv) > 1 else 20
cs = [int(x) for x in sys.argv[2:]] or [10, 12, 13, 14, 15, 16, 17, 18]
os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
print("%4s %12s %10s  %s" % ("c", "constr/s", "ms/proof", "kernel ms per proof (serial pass)"))
for c in cs:
    env = dict(os.environ, ZK_MSM_WINDOW=str(c))
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--log-n", str(log_n), "--steps", "6", "--warmup", "2",
                          "--inflight", "6", "--no-cpu-baseline"], env=env, capture_output=True, text=True, cwd=ROOT)
    line = [l for l in out.stdout.splitlines() if l.startswith("{")]
    if not line:
        print(c, "FAILED", out.stderr[-300:])
        continue
    d = json.loads(line[-1])
    open(os.path.join(ROOT, "gpurun_out", "sweep_c%d.json" % c), "w").write(line[-1])
    k = d["kernel_ms_per_proof"]
    print("%4d %12.0f %10.2f  acc_g1 %.2f acc_g2 %.2f red_g1 %.2f red_g2 %.2f sort %.2f fr %.2f" % (
        c, d["value"], d["ms_per_step"], k.get("msm_accumulate_g1", 0), k.get("msm_accumulate_g2", 0), k.get("msm_reduce_g1", 0),
        k.get("msm_reduce_g2", 0), k.get("msm_sort", 0), k.get("fr_tree", 0) + k.get("fr_newton", 0) + k.get("fr_quotient", 0)))
    sys.stdout.flush()
