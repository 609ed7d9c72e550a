#!/bin/bash
# gpurun_out/<tag20> (scripts/r05_collect.sh stats1 stats pmc valu at LOGN=20) and gpurun_out/<tagpin> (pin pin1 pinpmc) -> the round-5 files under profiles/.
# usage: scripts/r05_install_profiles.sh <tag20> <tagpin>
set -e
T20=gpurun_out/$1; TP=gpurun_out/$2; P=profiles
cp $T20/stats/run_kernel_stats.csv "$P/r05_kernel_stats_2^20_derived_pipelined_rocprofv3.csv"
cp $T20/stats1/run_kernel_stats.csv "$P/r05_kernel_stats_2^20_derived_one_proof_in_flight_rocprofv3.csv"
cp $T20/bench_under_rocprofv3.json "$P/r05_bench_line_2^20_pipelined_under_rocprofv3.json"
cp $T20/bench1_under_rocprofv3.json "$P/r05_bench_line_2^20_one_proof_in_flight_under_rocprofv3.json"
cp $TP/pin/run_kernel_stats.csv "$P/r05_kernel_stats_pinocchio_2^18_derived_pipelined_rocprofv3.csv"
cp $TP/pin1/run_kernel_stats.csv "$P/r05_kernel_stats_pinocchio_2^18_derived_one_proof_in_flight_rocprofv3.csv"
cp $TP/pin_under_rocprofv3.json "$P/r05_bench_line_pinocchio_2^18_pipelined_under_rocprofv3.json"
cp $TP/pin1_under_rocprofv3.json "$P/r05_bench_line_pinocchio_2^18_one_proof_in_flight_under_rocprofv3.json"
rm -f $P/r05_pmc_traffic.json
python scripts/make_pmc_traffic.py $T20/pmc_FETCH_SIZE.summary.json $T20/pmc_WRITE_SIZE.summary.json "groth16_2^20_derived" $((1<<20)) $P/r05_pmc_traffic.json
python scripts/make_pmc_traffic.py $TP/pinpmc_FETCH_SIZE.summary.json $TP/pinpmc_WRITE_SIZE.summary.json "pinocchio_2^18_derived" $((1<<18)) $P/r05_pmc_traffic.json
{
  echo "# rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU, one proof in flight, derived key (scripts/r05_collect.sh valu), round-5 build: vector instructions per proof by kernel"
  echo "## Groth16, 2^20 constraints"; cat $T20/pmc_valu.txt
} > $P/r05_valu_instructions_per_proof.txt
bash scripts/kernel_resources.sh > $P/r05_kernel_resources.txt
echo installed
