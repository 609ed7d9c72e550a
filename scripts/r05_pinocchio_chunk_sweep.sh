#!/bin/bash
# Round 5: Pinocchio 2^18 over the chunk length of the bucket accumulation (ZK_MSM_CHUNK_MIN; the fix-up of the chunk partial sums is 8.8 % of a Pinocchio proof's vector instructions at 16 entries per chunk).
# Measured (derived key, 8 in flight): 16 / 20 / 24 / 32 / 48 entries = 38.8 / 38.8 / 38.7 / 39.0 / 38.4 M constraints/s -- the fix-up shrinks (reduction 1.44 -> 1.11 ms at 32) and the accumulate's waves balance worse by as much; 16 stays.
for ch in 16 20 24 32 48; do echo "chunk_min $ch"; ZK_MSM_CHUNK_MIN=$ch timeout -k 10 200 python scripts/r05_pinocchio_ab.py 18 1 8 2>/dev/null | grep 'compact": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   derived %.2f M  as uploaded %.2f M  latency %.2f  alone: %s' % (d['value_derived'], d['as_uploaded'], d['latency_ms'], d['alone_ms']))"; done
