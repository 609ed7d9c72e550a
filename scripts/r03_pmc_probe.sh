#!/bin/bash
# Which stall keeps the accumulate kernels at ~63 % of the multiplier rate?  SQ counters of one proof in flight at 2^LN (tau-power key: no derivation),
# one --pmc pass per counter group (separate runs, kernel-trace only).  usage: scripts/r03_pmc_probe.sh <tag> <log_n> "<counters>" ["<counters>" ...]
set -o pipefail
TAG=$1; LN=$2; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--headline-only --no-cpu-baseline --no-parity-gate --cpu-fast-upto -1 --derive-lagrange-upto -1 --log-n $LN --inflight 1 --steps 1 --proofs-per-step 3 --warmup 0 --settle 0"
i=0
for grp in "$@"; do
  i=$((i+1))
  timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_$i -o run -- python bench.py $ARGS > $O/pmc_$i.json 2> $O/pmc_$i.err || { tail -5 $O/pmc_$i.err; continue; }
  python scripts/pmc_summary.py $O/pmc_$i/run_counter_collection.csv --steady k_fr_to_mont_flag2 > $O/pmc_$i.txt
  grep -E "^#|k_msm_accumulate" $O/pmc_$i.txt | cut -c1-150
done
