#!/bin/bash
# A/B of two builds on the same box: zukelang_amd/libzk_prev.so (A) against zukelang_amd/libzkmi355x.so (B), alternating.
# usage: [ENV=...] scripts/ab_bench.sh [rounds] [bench.py args...]
R=${1:-2}; shift
L=zukelang_amd/libzkmi355x.so
cp $L /tmp/zk_B.so && cp zukelang_amd/libzk_prev.so /tmp/zk_A.so || exit 1
for i in $(seq $R); do
  for v in A B; do
    cp /tmp/zk_$v.so $L
    timeout -k 10 400 python bench.py --headline-only --no-cpu-baseline --derive-lagrange-upto ${DERIVE:--1} "$@" > gpurun_out/ab_$v$i.json 2> gpurun_out/ab_$v$i.err || { cp /tmp/zk_B.so $L; tail -5 gpurun_out/ab_$v$i.err; exit 1; }
    python -c "import json; d=json.load(open('gpurun_out/ab_$v$i.json')); print('$v$i %.2f M/s  %.3f ms/proof  latency %.2f ms' % (d['value']/1e6, d['ms_per_proof'], d['single_proof_latency_ms']))"
  done
done
cp /tmp/zk_B.so $L
