#!/bin/bash
# Round-5 evidence (same passes as rounds 3-4, plus the Pinocchio passes `pin` and `pinpmc`) collection on the GPU box, on the path the headline runs: a reference-format key whose Lagrange form is DERIVED on the device
# (bench.py --derived-only: no tau-power pass in the profile; the one-time derivation kernels k_gntt_* / k_lag_derive_* appear once per run).
# usage: LOGN=20 scripts/r05_collect.sh <tag> <stage>...      stages: stats | stats1 | pmc | valu | lds | pin | pin1 | pinpmc
# Each rocprofv3 pass is its own run (counters never share a run with --stats), the program comes directly after `--`.
set -o pipefail
TAG=$1; shift
LN=${LOGN:-20}
O=gpurun_out/$TAG; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
KEYFORM="--derived-only"; [ "${TAU_POWER:-0}" = 1 ] && KEYFORM="--derive-lagrange-upto -1"      # TAU_POWER=1: the key as uploaded (the basis-conversion Fr stage) instead of the derived form
BARGS="--headline-only $KEYFORM --no-cpu-baseline --no-parity-gate --cpu-fast-upto -1 --log-n $LN"
ONE="--inflight 1 --steps 1 --proofs-per-step 6 --warmup 0 --settle 0"
for st in "$@"; do
  case $st in
    stats) echo "== rocprofv3 --kernel-trace --stats, pipelined (the timed region's condition, serialised by the profiler: see profiles/README.md)"
      timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats -o run -- python bench.py $BARGS --steps 4 > $O/bench_under_rocprofv3.json 2> $O/stats.err || { tail -20 $O/stats.err; exit 1; };;
    stats1) echo "== rocprofv3 --kernel-trace --stats, one proof in flight on ONE stream (ZK_SERIAL_STREAMS=1: slot 0 does not fork its three products, so every kernel runs un-overlapped)"
      ZK_SERIAL_STREAMS=1 timeout -k 10 700 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats1 -o run -- python bench.py $BARGS $ONE > $O/bench1_under_rocprofv3.json 2> $O/stats1.err || { tail -20 $O/stats1.err; exit 1; };;
    pmc) for c in FETCH_SIZE WRITE_SIZE; do echo "== pmc $c"
        timeout -k 10 700 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py $BARGS $ONE > $O/pmc_$c.json 2> $O/pmc_$c.err || { tail -20 $O/pmc_$c.err; exit 1; }
        python scripts/pmc_summary.py $O/pmc_$c/run_counter_collection.csv --steady k_fr_to_mont_flag2 --json $O/pmc_$c.summary.json > $O/pmc_$c.txt; done;;
    valu) echo "== pmc SQ_INSTS_VALU"
        timeout -k 10 700 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU --output-format csv -d $O/pmc_valu -o run -- python bench.py $BARGS $ONE > $O/pmc_valu.json 2> $O/pmc_valu.err || { tail -20 $O/pmc_valu.err; exit 1; }
        python scripts/pmc_summary.py $O/pmc_valu/run_counter_collection.csv --steady k_fr_to_mont_flag2 > $O/pmc_valu.txt; head -16 $O/pmc_valu.txt;;
    lds) echo "== pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE"
        timeout -k 10 700 rocprofv3 --kernel-trace --pmc SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --output-format csv -d $O/pmc_lds -o run -- python bench.py $BARGS $ONE > $O/pmc_lds.json 2> $O/pmc_lds.err || { tail -20 $O/pmc_lds.err; exit 1; }
        python scripts/pmc_summary.py $O/pmc_lds/run_counter_collection.csv --steady k_fr_to_mont_flag2 > $O/pmc_lds.txt; head -30 $O/pmc_lds.txt;;
    pin) echo "== rocprofv3 --kernel-trace --stats, Pinocchio ZK prove 2^18 (BASELINE config 5), compact h pool + shared sorts, derived h bases, 8 proofs in flight"
      PIN_DERIVE=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pin -o run -- python scripts/bench_pinocchio.py 18 48 8 > $O/pin_under_rocprofv3.json 2> $O/pin.err || { tail -20 $O/pin.err; exit 1; };;
    pin1) echo "== the same, ONE proof in flight on one stream"
      PIN_DERIVE=1 ZK_SERIAL_STREAMS=1 timeout -k 10 500 rocprofv3 --kernel-trace --stats --output-format csv -d $O/pin1 -o run -- python scripts/bench_pinocchio.py 18 6 1 > $O/pin1_under_rocprofv3.json 2> $O/pin1.err || { tail -20 $O/pin1.err; exit 1; };;
    pinpmc) for c in FETCH_SIZE WRITE_SIZE; do echo "== pmc $c (Pinocchio, one proof in flight)"
        PIN_DERIVE=1 ZK_SERIAL_STREAMS=1 timeout -k 10 500 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pinpmc_$c -o run -- python scripts/bench_pinocchio.py 18 6 1 > $O/pinpmc_$c.json 2> $O/pinpmc_$c.err || { tail -20 $O/pinpmc_$c.err; exit 1; }
        python scripts/pmc_summary.py $O/pinpmc_$c/run_counter_collection.csv --steady k_fr_to_mont_flag2 --steady-skip 1 --json $O/pinpmc_$c.summary.json > $O/pinpmc_$c.txt; done;;
    calib) echo "== FETCH_SIZE calibration on the gather pattern (scripts/proto/gather_calib.hip)"
        timeout -k 10 300 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $O/calib -o run -- ./scripts/proto/gather_calib > $O/calib.txt 2> $O/calib.err || { tail -20 $O/calib.err; exit 1; }
        python - <<PY
import csv
rows = list(csv.DictReader(open("$O/calib/run_counter_collection.csv")))
known = {"k_gather_regs": (1 << 24) * 128, "k_gather_lds": (1 << 24) * 128, "k_stream": (1 << 24) * 128}
with open("$O/calib.txt", "a") as f:
    for r in rows:
        name = r["Kernel_Name"].split("(")[0]
        if name in known and r["Counter_Name"] == "FETCH_SIZE":
            b = float(r["Counter_Value"]) * 1024
            line = "%-14s FETCH_SIZE %.1f MB   known %.1f MB   known / reported = %.3f" % (name, b / 1e6, known[name] / 1e6, known[name] / b)
            print(line); f.write(line + "\n")
PY
        ;;
    *) echo "unknown stage $st"; exit 2;;
  esac
done
echo done
