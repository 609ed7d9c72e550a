#!/bin/bash
# What does the 16x window table cost?  A/B of the real tables against ZK_EXPERIMENT_TABLE_ALIAS=1 (every window gathers from
# window 0's entries: same additions and gathers, 1/16 of the footprint; proofs are WRONG, the parity gate is off).
# The experiment is NOT in the shipped library: this script rebuilds msm.o with -DZK_EXPERIMENTS first and restores the product build afterwards.
O=gpurun_out/$1; mkdir -p $O
( cd zukelang_amd/csrc && rm -f msm.o && make -s CXXFLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wall -Wno-unused-function -fvisibility=hidden -fvisibility-inlines-hidden -DZK_EXPERIMENTS" ) || exit 1
trap '( cd zukelang_amd/csrc && rm -f msm.o && make -s )' EXIT
for ln in 16 20; do
  for al in 0 1; do
    for infl in 1 12; do
      steps=$([ $ln = 16 ] && echo 6 || echo 2)
      ZK_SERIAL_STREAMS=1 ZK_EXPERIMENT_TABLE_ALIAS=$al timeout -k 10 400 python bench.py --headline-only --no-cpu-baseline --no-parity-gate --derive-lagrange-upto -1 --log-n $ln --steps $steps --warmup 0 --inflight $infl > $O/alias_${ln}_${al}_$infl.json 2> $O/alias.err || { tail -5 $O/alias.err; exit 1; }
      python -c "
import json; d=json.load(open('$O/alias_${ln}_${al}_$infl.json')); k=d['kernel_ms_per_proof']
print('2^$ln alias=$al inflight=$infl: %.2f M/s  %.3f ms/proof | one proof in flight: acc_g1 %.3f ms acc_g2 %.3f ms' % (d['value']/1e6, d['ms_per_proof'], k['msm_accumulate_g1'], k['msm_accumulate_g2']))"
    done
  done
done
