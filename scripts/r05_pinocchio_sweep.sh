for inf in 8 10 12 14; do echo "inflight $inf"; timeout -k 10 200 python scripts/r05_pinocchio_ab.py 18 1 $inf 2>/dev/null | grep 'compact": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   derived %.2f M  as uploaded %.2f M  latency %.2f' % (d['value_derived'], d['as_uploaded'], d['latency_ms']))"; done
for w in 15 17; do echo "window $w (inflight 8)"; ZK_MSM_WINDOW=$w timeout -k 10 200 python scripts/r05_pinocchio_ab.py 18 1 8 2>/dev/null | grep 'compact": 1' | python -c "
import sys,json
for l in sys.stdin:
    d=json.loads(l); print('   derived %.2f M  as uploaded %.2f M  latency %.2f' % (d['value_derived'], d['as_uploaded'], d['latency_ms']))"; done
