#!/bin/bash
# Derivation time of a key's Lagrange form, sets one after another vs side by side (ZK_DERIVE_SIDE_BY_SIDE): scripts/derive_ab.sh <tag> <log_n>...
TAG=$1; shift
O=gpurun_out/$TAG; mkdir -p $O
for ln in "$@"; do
  for sbs in 0 1; do
    ZK_DERIVE_SIDE_BY_SIDE=$sbs timeout -k 10 500 python bench.py --headline-only --no-cpu-baseline --derived-only --no-parity-gate --cpu-fast-upto -1 --log-n $ln --steps 2 > $O/d_${ln}_$sbs.json 2> $O/d_${ln}_$sbs.err || { tail -5 $O/d_${ln}_$sbs.err; exit 1; }
    python -c "
import json; d=json.load(open('$O/d_${ln}_$sbs.json')); print('2^$ln side by side $sbs: derivation %.2f s   (%.2f M constraints/s afterwards)' % (d['config']['derive_lagrange_s'], d['value']/1e6))"
  done
done
