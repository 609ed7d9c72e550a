#!/bin/bash
# quick profile of one workload: kernel stats (one proof in flight) + FETCH/WRITE PMC.  usage: scripts/r02_prof.sh <tag> <log_n> [stats|pmc]...
set -o pipefail
TAG=$1; LN=$2; shift; shift
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
ARGS="--headline-only --no-cpu-baseline --no-parity-gate --derive-lagrange-upto -1 --log-n $LN --inflight 1 --steps 1 --proofs-per-step 4 --warmup 0 --settle 0"
for st in "$@"; do
  case $st in
    stats) timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py $ARGS > $O/stats.json 2> $O/stats.err || { tail -20 $O/stats.err; exit 1; }
           python - <<PY
import csv
rows=list(csv.DictReader(open("$O/stats/run_kernel_stats.csv")))
for r in rows[:22]:
    print("%-60s calls %6s total_ms %10.3f avg_us %10.1f  %5s%%" % (r["Name"][:60], r["Calls"], float(r["TotalDurationNs"])/1e6, float(r["AverageNs"])/1e3, r["Percentage"]))
PY
           ;;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do
           timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py $ARGS > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail -20 $O/pmc_$c.err; exit 1; }
           python scripts/pmc_summary.py $O/pmc_$c/run_counter_collection.csv --proofs 9 > $O/pmc_$c.txt; head -12 $O/pmc_$c.txt | cut -c1-140; done;;
  esac
done
