#!/bin/bash
# Round 5: Pinocchio 2^18 (compact h pool + shared sorts) over the proofs-in-flight depth and the window width, through bench.py's bench_pinocchio (scripts/r05_pinocchio_ab.py).
# Measured: 8 / 10 / 12 / 14 in flight = 39.5 / 39.7 / 39.8 / 39.3 M constraints/s (the default stays 8); ZK_MSM_WINDOW=15 / 17 = 30.4 / 38.9 M against 39.5 M at the default 16.
for inf in 8 10 12 14; do echo "inflight $inf"; timeout -k 10 200 python scripts/r05_pinocchio_ab.py 18 1 $inf 2>/dev/null | grep 'compact": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   derived %.2f M  as uploaded %.2f M  latency %.2f' % (d['value_derived'], d['as_uploaded'], d['latency_ms']))"; done
for w in 15 17; do echo "window $w (inflight 8)"; ZK_MSM_WINDOW=$w timeout -k 10 200 python scripts/r05_pinocchio_ab.py 18 1 8 2>/dev/null | grep 'compact": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   derived %.2f M  as uploaded %.2f M  latency %.2f' % (d['value_derived'], d['as_uploaded'], d['latency_ms']))"; done
