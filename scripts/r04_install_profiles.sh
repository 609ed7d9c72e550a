#!/bin/bash
# gpurun_out/<tag20>, gpurun_out/<tag16> (scripts/r04_collect.sh stats1 pmc valu stats at LOGN=20 / 16) -> the round-4 files under profiles/.
# usage: scripts/r04_install_profiles.sh <tag20> <tag16>
set -e
T20=gpurun_out/$1; T16=gpurun_out/$2; P=profiles
for s in 20:$T20 16:$T16; do
  ln=${s%%:*}; d=${s#*:}
  cp $d/stats/run_kernel_stats.csv "$P/r04_kernel_stats_2^${ln}_derived_pipelined_rocprofv3.csv"
  cp $d/stats1/run_kernel_stats.csv "$P/r04_kernel_stats_2^${ln}_derived_one_proof_in_flight_rocprofv3.csv"
  cp $d/bench_under_rocprofv3.json "$P/r04_bench_line_2^${ln}_pipelined_under_rocprofv3.json"
  cp $d/bench1_under_rocprofv3.json "$P/r04_bench_line_2^${ln}_one_proof_in_flight_under_rocprofv3.json"
done
python scripts/make_pmc_traffic.py $T20/pmc_FETCH_SIZE.summary.json $T20/pmc_WRITE_SIZE.summary.json "groth16_2^20_derived" $((1<<20)) $P/r04_pmc_traffic.json
python scripts/make_pmc_traffic.py $T16/pmc_FETCH_SIZE.summary.json $T16/pmc_WRITE_SIZE.summary.json "groth16_2^16_derived" $((1<<16)) $P/r04_pmc_traffic.json
{
  echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU, one proof in flight, derived key (scripts/r04_collect.sh valu), final build: vector instructions per proof by kernel"
  echo "## 2^20 constraints"; cat $T20/pmc_valu.txt
  echo; echo "## 2^16 constraints"; cat $T16/pmc_valu.txt
} > $P/r04_valu_instructions_per_proof.txt
bash scripts/kernel_resources.sh > $P/r04_kernel_resources.txt
echo installed
