#!/bin/bash
# un-overlapped accumulate kernel times (one proof in flight) and pipelined throughput
for n in "$@"; do
  timeout -k 10 300 python bench.py --log-n $n --no-cpu-baseline --steps 8 --warmup 2 --inflight 1 > gpurun_out/acc1_$n.json 2> gpurun_out/acc1_$n.err || exit 1
  timeout -k 10 300 python bench.py --log-n $n --no-cpu-baseline --steps 24 --warmup 4 > gpurun_out/accp_$n.json 2> gpurun_out/accp_$n.err || exit 1
  python - <<PY
import json
a=json.load(open("gpurun_out/acc1_$n.json")); p=json.load(open("gpurun_out/accp_$n.json"))
k=a["kernel_ms_per_proof"]
print("2^$n serial: %.3f ms/proof  acc_g1/proof=%.3f acc_g2/proof=%.3f | pipelined: %.3f ms/proof = %.2f M/s" % (a["ms_per_step"], k["msm_accumulate_g1"], k["msm_accumulate_g2"], p["ms_per_step"], p["value"]/1e6))
PY
done
