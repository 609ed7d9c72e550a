#!/bin/bash
# A/B of environment settings on the Pinocchio 2^18 workload (config 5): scripts/pin_ab.sh <tag> "ENV1=a" "ENV2=b" ...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 400 python bench.py --log-n 16 --sizes "" --no-cpu-baseline --derive-lagrange-upto 18 --cpu-fast-upto -1 --steps 4 > $O/p_${i}_$rep.json 2> $O/p_${i}_$rep.err || { tail -5 $O/p_${i}_$rep.err; exit 1; }
    python -c "
import json
d=json.load(open('$O/p_${i}_$rep.json')); w=[x for x in d['other_workloads'] if 'pinocchio' in x['workload']][0]
print('[%s] rep $rep: Pinocchio 2^18 %.2f M/s derived  %.2f M/s as uploaded   (Groth16 2^16 beside it: %.2f)' % ('$e', w['value']/1e6, (w['tau_power_value'] or 0)/1e6, d['value']/1e6))"
  done
done
