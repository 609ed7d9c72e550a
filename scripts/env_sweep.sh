#!/bin/bash
# Sweep of one environment knob on the same box, round robin: scripts/env_sweep.sh NAME "v1 v2 ..." [rounds] [bench.py args...]
N=$1; VALS=$2; R=${3:-2}; shift 3
for i in $(seq $R); do
  for v in $VALS; do
    env $N=$v timeout -k 10 300 python bench.py --no-cpu-baseline "$@" > gpurun_out/sweep_${N}_$v.$i.json 2> gpurun_out/sweep_${N}_$v.$i.err || exit 1
    python -c "import json; d=json.load(open('gpurun_out/sweep_${N}_$v.$i.json')); print('$N=$v #$i %.2f M/s  %.3f ms' % (d['value']/1e6, d['ms_per_step']))"
  done
done
