#!/bin/bash
# Round-2 evidence collection on the GPU box.  usage: scripts/r02_collect.sh <tag> <stage>...
# stages: tests | bench | stats | pmc | valu      (each leaves its files under gpurun_out/<tag>/)
set -o pipefail
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BARGS="--headline-only --no-cpu-baseline --no-parity-gate --derive-lagrange-upto ${DERIVE:--1} ${LOGN:+--log-n $LOGN}"      # LOGN=20 scripts/r02_collect.sh ... for the other sizes
for st in "$@"; do
  case $st in
    tests) echo "== gpu tests"; timeout -k 10 900 python -m pytest tests -m gpu -x -q > $O/pytest_gpu.log 2>&1 || { tail -30 $O/pytest_gpu.log; exit 1; }; tail -3 $O/pytest_gpu.log;;
    bench) echo "== bench"; timeout -k 10 600 python bench.py > $O/bench.json 2> $O/bench.err || { tail -20 $O/bench.err; exit 1; }; cut -c1-600 $O/bench.json;;
    stats) echo "== rocprofv3 stats"
      timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py $BARGS > $O/bench_under_rocprofv3.json 2> $O/stats.err || { tail -20 $O/stats.err; exit 1; }
      timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o run -- python bench.py $BARGS --inflight 1 --steps 1 --proofs-per-step 10 --warmup 0 > $O/bench1_under_rocprofv3.json 2> $O/stats1.err || { tail -20 $O/stats1.err; exit 1; };;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do echo "== pmc $c"
        timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py $BARGS --inflight 1 --steps 1 --proofs-per-step 6 --warmup 0 --settle 0 > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail -20 $O/pmc_$c.err; exit 1; }
        python scripts/pmc_summary.py $O/pmc_$c/run_counter_collection.csv --steady k_fr_to_mont_flag2 --json $O/pmc_$c.summary.json > $O/pmc_$c.txt; done;;
    valu) echo "== pmc SQ_INSTS_VALU"
        timeout -k 10 500 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc_valu -o run -- python bench.py $BARGS --inflight 1 --steps 1 --proofs-per-step 6 --warmup 0 --settle 0 > $O/pmc_valu.json 2> $O/pmc_valu.err || { tail -20 $O/pmc_valu.err; exit 1; }
        python scripts/pmc_summary.py $O/pmc_valu/run_counter_collection.csv --steady k_fr_to_mont_flag2 > $O/pmc_valu.txt; head -16 $O/pmc_valu.txt;;
    *) echo "unknown stage $st"; exit 2;;
  esac
done
echo done
