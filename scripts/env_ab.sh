#!/bin/bash
# A/B of environment settings on one box, alternating: scripts/env_ab.sh <tag> "<bench args>" "ENV1=a ENV2=b" "ENV1=c" ...
# (ZK_LIBZKMI355X_PATH=zukelang_amd/libzkmi355x_A.so in a setting selects another BUILD of the library: A/B of code changes on one box.)
# Prints the headline, the latency and the un-overlapped reduction / sort / Fr families of each run (bench_detail.json).
TAG=$1; ARGS=$2; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
for rep in 1 2; do
  i=0
  for e in "$@"; do
    i=$((i+1))
    env $e timeout -k 10 400 python bench.py --headline-only --no-cpu-baseline --derive-lagrange-upto ${DERIVE:--1} $ARGS > $O/ab_${i}_$rep.json 2> $O/ab_${i}_$rep.err || { tail -5 $O/ab_${i}_$rep.err; exit 1; }
    cp bench_detail.json $O/ab_${i}_${rep}_detail.json
    python -c "
import json
d=json.load(open('$O/ab_${i}_$rep.json')); k=json.load(open('$O/ab_${i}_${rep}_detail.json')).get('kernel_ms_per_proof') or {}
fam=' '.join('%s %.3f' % (n.replace('msm_reduce:','red:').replace('msm_',''), k[n]) for n in sorted(k) if n.startswith(('msm_reduce', 'msm_sort', 'fr_', 'ntt_', 'msm_acc')))
print('[%s] rep $rep: %.2f M/s  %.3f ms/proof  latency %.2f ms | %s' % ('$e', d['value']/1e6, d['ms_per_proof'], d['single_proof_latency_ms'], fam))"
  done
done
