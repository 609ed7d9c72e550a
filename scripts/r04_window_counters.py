#!/usr/bin/env python3
"""gpurun_out/<tag>/ of scripts/r03_window_counters.sh -> window_sweep_counters.json (written next to the inputs; copy to profiles/).
Per window width: the timed row (constraints/s, ms per proof, parity), HBM-side traffic per kernel family (FETCH_SIZE / WRITE_SIZE in bytes per proof, and
2 * FETCH + WRITE, the guide's wide-read correction -- see profiles/README.md on what that factor means for random gathers) and the LDS conflict share
(SQ_LDS_BANK_CONFLICT / SQ_LDS_IDX_ACTIVE, cycles) of the kernels that use LDS: the counting sort and the NTT tiles."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from make_pmc_traffic import FAMILY          # noqa: E402

out_dir = sys.argv[1]
widths = [int(x) for x in sys.argv[2:]]
rows = []
for c in widths:
    row = json.load(open(os.path.join(out_dir, "row_%d.json" % c)))
    detail = json.load(open(os.path.join(out_dir, "row_%d.detail.json" % c)))          # the long form of the same run (bench_detail.json)
    f = json.load(open(os.path.join(out_dir, "pmc_%d_1.summary.json" % c)))
    w = json.load(open(os.path.join(out_dir, "pmc_%d_2.summary.json" % c)))
    l = json.load(open(os.path.join(out_dir, "pmc_%d_3.summary.json" % c)))
    fams = {}
    for k in sorted(set(f) | set(w)):
        if k not in FAMILY:
            continue
        fb = f.get(k, {}).get("FETCH_SIZE", {}).get("per_proof", 0.0) * 1024
        wb = w.get(k, {}).get("WRITE_SIZE", {}).get("per_proof", 0.0) * 1024
        e = fams.setdefault(FAMILY[k], {"FETCH_SIZE_bytes_per_proof": 0.0, "WRITE_SIZE_bytes_per_proof": 0.0})
        e["FETCH_SIZE_bytes_per_proof"] += fb
        e["WRITE_SIZE_bytes_per_proof"] += wb
    for e in fams.values():
        e["hbm_bytes_per_proof_2F_plus_W"] = 2 * e["FETCH_SIZE_bytes_per_proof"] + e["WRITE_SIZE_bytes_per_proof"]
    lds = {}
    for k, v in l.items():
        a = v.get("SQ_LDS_IDX_ACTIVE", {}).get("per_proof", 0.0)
        b = v.get("SQ_LDS_BANK_CONFLICT", {}).get("per_proof", 0.0)
        if a > 0:
            lds[k] = {"SQ_LDS_IDX_ACTIVE_per_proof": a, "SQ_LDS_BANK_CONFLICT_per_proof": b, "conflict_share": b / a}
    nw = 255 // c + 1
    rows.append({"window_bits": c, "windows": nw, "buckets": 1 << (c - 1), "value": row["value"], "ms_per_proof": row["ms_per_proof"],
                 "single_proof_latency_ms": row["single_proof_latency_ms"], "parity": row["parity"], "kernel_ms_per_proof": detail["kernel_ms_per_proof"],
                 "hbm_traffic_per_family": fams, "lds": lds})
    tot = sum(e["hbm_bytes_per_proof_2F_plus_W"] for e in fams.values())
    print("c = %2d: %6.2f M constraints/s, %6.2f ms per proof | 2F+W per proof %7.2f GB: %s" % (
        c, row["value"] / 1e6, row["ms_per_proof"], tot / 1e9, {k: round(e["hbm_bytes_per_proof_2F_plus_W"] / 1e9, 2) for k, e in sorted(fams.items())}))
    print("         LDS conflict share: %s" % {k.split("<")[0]: round(v["conflict_share"], 3) for k, v in sorted(lds.items())})
json.dump({"workload": "groth16_prove 2^20, iterated-cubic R1CS, reference-format key as uploaded; timed rows: 6 proofs in flight, 12 timed proofs, parity gate on; "
                       "counters: one proof in flight, rocprofv3 --pmc in separate passes (FETCH_SIZE | WRITE_SIZE | SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE)",
           "rows": rows}, open(os.path.join(out_dir, "window_sweep_counters.json"), "w"), indent=1)
