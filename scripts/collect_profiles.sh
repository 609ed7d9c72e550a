#!/bin/bash
# Collects the round's committed evidence on the GPU box: bench lines, rocprofv3 kernel stats, PMC traffic,
# Pinocchio, window sweep.  Everything lands under gpurun_out/r01c/; copy what is judged into profiles/.
set -o pipefail
O=gpurun_out/r01f; mkdir -p $O
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
echo "== default bench"; timeout -k 10 400 python bench.py > $O/bench_2^16.json 2> $O/bench_2^16.err || exit 1
for n in 18 20; do echo "== 2^$n"; timeout -k 10 400 python bench.py --log-n $n --no-cpu-baseline > $O/bench_2^$n.json 2> $O/bench_2^$n.err || exit 1; done
echo "== rocprofv3 stats (default command)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_default -o run -- python bench.py --no-cpu-baseline > $O/bench_under_rocprofv3.json 2> $O/stats_default.err || exit 1
echo "== rocprofv3 stats (one proof in flight)"
timeout -k 10 400 rocprofv3 --kernel-trace --stats --output-format csv -d $O/stats_serial -o run -- python bench.py --no-cpu-baseline --inflight 1 --steps 10 > $O/bench_serial_under_rocprofv3.json 2> $O/stats_serial.err || exit 1
for c in FETCH_SIZE WRITE_SIZE; do
  echo "== pmc $c"
  timeout -k 10 400 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $O/pmc_$c -o run -- python bench.py --no-cpu-baseline --inflight 1 --steps 6 --warmup 1 > $O/pmc_$c.json 2> $O/pmc_$c.err || exit 1
done
echo "== 2^22"; timeout -k 10 900 python bench.py --log-n 22 --no-cpu-baseline --steps 6 --warmup 2 --inflight 4 > $O/bench_2^22.json 2> $O/bench_2^22.err || exit 1
for n in 16 18 20; do echo "== lagrange-form key 2^$n"; timeout -k 10 400 python bench.py --log-n $n --no-cpu-baseline --lagrange-key > $O/bench_lagrange_2^$n.json 2> $O/bench_lagrange_2^$n.err || exit 1; done
echo "== pinocchio 2^18"; timeout -k 10 600 python scripts/bench_pinocchio.py 18 8 8 > $O/pinocchio_2^18.json 2> $O/pinocchio.err || exit 1

echo done
