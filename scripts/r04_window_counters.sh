#!/bin/bash
# BASELINE config 3 with counters (round 4: the same passes on the round-4 build): Groth16 at 2^20 constraints (reference-format key as uploaded), Pippenger window widths c = 14 16 18 20:
# per width one timed, parity-gated bench row and three rocprofv3 --pmc passes of their own (FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE;
# kernel-trace only, program directly after `--`, one proof in flight).  scripts/r04_window_counters.py turns gpurun_out/<tag>/ into
# profiles/r04_window_sweep_counters.json.   usage: scripts/r04_window_counters.sh <tag> [c ...]
set -o pipefail
TAG=$1; shift
WIDTHS=${@:-14 16 18 20}
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
BARGS="--headline-only --no-cpu-baseline --cpu-fast-upto -1 --derive-lagrange-upto -1 --log-n 20"
ONE="--no-parity-gate --inflight 1 --steps 1 --proofs-per-step 4 --warmup 0 --settle 0"
for c in $WIDTHS; do
  export ZK_MSM_WINDOW=$c
  echo "== c = $c: timed row (6 proofs in flight, parity gate on)"
  timeout -k 10 300 python bench.py $BARGS --steps 1 --proofs-per-step 12 --warmup 0 --inflight 6 --settle 2 > $O/row_$c.json 2> $O/row_$c.err || { tail -5 $O/row_$c.err; exit 1; }
  cp bench_detail.json $O/row_$c.detail.json
  i=0
  for grp in "FETCH_SIZE" "WRITE_SIZE" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"; do
    i=$((i+1))
    timeout -k 10 300 rocprofv3 --kernel-trace --pmc $grp --output-format csv -d $O/pmc_${c}_$i -o run -- python bench.py $BARGS $ONE > $O/pmc_${c}_$i.json 2> $O/pmc_${c}_$i.err || { tail -5 $O/pmc_${c}_$i.err; exit 1; }
    python scripts/pmc_summary.py $O/pmc_${c}_$i/run_counter_collection.csv --steady k_fr_to_mont_flag2 --json $O/pmc_${c}_$i.summary.json > $O/pmc_${c}_$i.txt
    rm -rf $O/pmc_${c}_$i          # the raw per-dispatch CSVs are tens of MB
  done
done
python scripts/r04_window_counters.py $O $WIDTHS > $O/summary.txt && cat $O/summary.txt
