#!/usr/bin/env python3
"""profiles/rNN_pmc_traffic.json from the two PMC passes of scripts/r02_collect.sh (FETCH_SIZE and WRITE_SIZE, separate runs of
`bench.py --inflight 1`, per-kernel sums from scripts/pmc_summary.py --json).

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) * 1024: MI355X_MICROARCH.md -- FETCH_SIZE / WRITE_SIZE are in KiB, and on
gfx950 FETCH_SIZE reports half of the bytes of wide reads (128-byte requests tallied at 64 B); calibrated here on a kernel of
known traffic in the same run (k_fr_to_mont_flag2 reads and writes 32 B per element once).
usage: make_pmc_traffic.py <fetch.summary.json> <write.summary.json> <workload key> <n> <out.json>"""
import json
import sys

PREFIX = [   # kernel name prefix -> bench.py family, per-proof kernels only (first match wins)
    ("k_msm_accumulate<FpB", "msm_accumulate_g1"), ("k_msm_accumulate<Fp2HB", "msm_accumulate_g2"),
    ("k_ba_round<FpB", "msm_accumulate_g1"), ("k_ba_round<Fp2HB", "msm_accumulate_g2"), ("k_ba_plan", "msm_accumulate_g1"),
    ("k_msm_count", "msm_sort"), ("k_msm_scatter", "msm_sort"), ("k_scan", "msm_sort"), ("k_sort_", "msm_sort"),
    ("k_msm_fixup", "msm_reduce"), ("k_msm_digit_", "msm_reduce"), ("k_msm_final", "msm_reduce"), ("k_tail_", "msm_reduce"),
    ("k_proof_to_bytes", "proof_to_bytes"), ("k_groth16_scalars", "groth16_scalars"),
    ("k_ntt_", "ntt"), ("k_tree_levels_fused", "ntt"),
    ("k_spmv", "fr_pointwise"), ("k_check_r1cs", "fr_pointwise"), ("k_fr_to_mont_flag2", "fr_pointwise"), ("k_scale_pad", "fr_pointwise"), ("k_reverse_pad", "fr_pointwise"), ("k_lag_", "fr_pointwise"),
]


class _Family(dict):
    def __contains__(self, k):
        return any(k.startswith(p) for p, _ in PREFIX)

    def __getitem__(self, k):
        return next(f for p, f in PREFIX if k.startswith(p))


FAMILY = _Family()


def main():
    fetch, write = json.load(open(sys.argv[1])), json.load(open(sys.argv[2]))
    key, n, out = sys.argv[3], int(sys.argv[4]), sys.argv[5]
    kernels, fams = {}, {}
    for k in sorted(set(fetch) | set(write)):
        f = fetch.get(k, {}).get("FETCH_SIZE", {})
        w = write.get(k, {}).get("WRITE_SIZE", {})
        lp = f.get("launches_per_proof", w.get("launches_per_proof", 0))
        if k not in FAMILY or lp == 0:
            continue            # not a per-proof kernel (the summaries are taken with --steady: key set-up is already excluded; a launch count
                                # that is not an integer only means that slot 0 ran some proofs forked over three streams and some not)
        fb, wb = f.get("per_proof", 0.0) * 1024, w.get("per_proof", 0.0) * 1024
        kernels[k] = {"launches_per_proof": lp, "FETCH_SIZE_bytes_per_proof": fb, "WRITE_SIZE_bytes_per_proof": wb, "hbm_bytes_per_proof": 2 * fb + wb,
                      "hbm_bytes_per_launch": (2 * fb + wb) / lp, "family": FAMILY[k]}
        e = fams.setdefault(FAMILY[k], {"hbm_bytes_per_proof": 0.0, "kernels": []})
        e["hbm_bytes_per_proof"] += 2 * fb + wb
        e["kernels"].append(k)
    for fam, e in fams.items():
        # bench.py's "launch" of an accumulate family = the family's launches of one proof (one per proof for the default XYZZ path)
        e["hbm_bytes_per_launch"] = e["hbm_bytes_per_proof"]
    cal = None
    if "k_fr_to_mont_flag2" in kernels:
        k = kernels["k_fr_to_mont_flag2"]
        m = n + 2
        cal = {"kernel": "k_fr_to_mont_flag2", "known_bytes_read": 32 * m, "known_bytes_written": 32 * m,
               "FETCH_SIZE_bytes": k["FETCH_SIZE_bytes_per_proof"], "WRITE_SIZE_bytes": k["WRITE_SIZE_bytes_per_proof"],
               "fetch_ratio_known_over_reported": 32 * m / k["FETCH_SIZE_bytes_per_proof"] if k["FETCH_SIZE_bytes_per_proof"] else None}
    try:
        doc = json.load(open(out))
    except Exception:
        doc = {"how": "rocprofv3 --kernel-trace --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --headline-only --inflight 1` (one proof in "
                      "flight), summed per kernel over the run and divided by the proofs of the run; hbm bytes = (2 * FETCH_SIZE + WRITE_SIZE) * 1024 "
                      "(MI355X_MICROARCH.md: FETCH_SIZE on gfx950 reports half of wide reads); `calibration` checks that factor on a kernel of known traffic",
               "workloads": {}}
    doc["workloads"][key] = dict(fams)
    doc["workloads"][key]["_kernels"] = kernels
    doc["workloads"][key]["_calibration"] = cal
    json.dump(doc, open(out, "w"), indent=1)
    tot = sum(e["hbm_bytes_per_proof"] for e in fams.values())
    print("%s: %.1f MB per proof over the per-proof kernels; families: %s" % (key, tot / 1e6, {f: round(e["hbm_bytes_per_proof"] / 1e6, 1) for f, e in fams.items()}))


if __name__ == "__main__":
    main()
