#!/usr/bin/env python3
"""Per-kernel summary of a rocprofv3 --pmc collection (the *_counter_collection.csv of one pass).

usage: pmc_summary.py <counter_collection.csv> [--proofs N | --steady KERNEL [--steady-skip K]] [--json out.json]

--steady KERNEL: count only dispatches from the first dispatch of KERNEL on (the first kernel of the first proof, e.g.
k_fr_to_mont_flag2: key set-up launches the same NTT kernels and would blur the per-proof figures) and take the number of
KERNEL dispatches as the number of proofs.

Kernel names are shortened to the function name; values are summed per kernel over the run and divided by the
number of proofs (launch counts that are not a multiple of N belong to set-up kernels: they are listed as they are).
FETCH_SIZE / WRITE_SIZE are in KiB (MI355X_MICROARCH.md: double FETCH_SIZE for wide streaming reads on gfx950 --
the doubling is applied by the caller, see scripts/r02_collect.sh, not here).
"""
import csv
import json
import re
import sys
from collections import OrderedDict


def short(name):
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0]
    return name.replace("zk::", "")


def main():
    args = sys.argv[1:]
    path = args[0]
    proofs = int(args[args.index("--proofs") + 1]) if "--proofs" in args else 1
    out_json = args[args.index("--json") + 1] if "--json" in args else None
    acc = OrderedDict()
    steady = args[args.index("--steady") + 1] if "--steady" in args else None
    with open(path, newline="") as f:
        rows = sorted(csv.DictReader(f), key=lambda r: int(r["Dispatch_Id"]))
    if steady:
        # --steady-skip K: the first K dispatches of KERNEL belong to key set-up (Pinocchio's upload runs the Fr stage once for its consistency check)
        skip = int(args[args.index("--steady-skip") + 1]) if "--steady-skip" in args else 0
        counters0 = {r["Counter_Name"] for r in rows}
        hits = [i for i, r in enumerate(rows) if short(r["Kernel_Name"]) == steady]
        first = hits[skip * max(1, len(counters0))]
        rows = rows[first:]
        counters = {r["Counter_Name"] for r in rows}
        proofs = sum(1 for r in rows if short(r["Kernel_Name"]) == steady) // max(1, len(counters))
    if True:
        for row in rows:
            k = (short(row["Kernel_Name"]), row["Counter_Name"])
            e = acc.setdefault(k, {"sum": 0.0, "launches": 0, "grid": 0, "vgpr": int(row["VGPR_Count"]), "agpr": int(row["Accum_VGPR_Count"]),
                                   "scratch": int(row["Scratch_Size"]), "lds": int(row["LDS_Block_Size"])})
            e["sum"] += float(row["Counter_Value"])
            e["launches"] += 1
    counters = sorted({c for _, c in acc})
    res = {}
    for c in counters:
        rows = [(k, v) for (k, cc), v in acc.items() if cc == c]
        total = sum(v["sum"] for _, v in rows)
        rows.sort(key=lambda kv: -kv[1]["sum"])
        print("# %s   (per proof = run total / %d)" % (c, proofs))
        for k, v in rows:
            print("%-44s launches/proof %6.2f   %s/proof %14.2f   %5.1f %%   per launch %14.2f   vgpr %3d agpr %3d scratch %5d lds %6d" % (
                k, v["launches"] / proofs, c, v["sum"] / proofs, 100.0 * v["sum"] / total if total else 0.0, v["sum"] / v["launches"],
                v["vgpr"], v["agpr"], v["scratch"], v["lds"]))
            res.setdefault(k, {})[c] = {"per_proof": v["sum"] / proofs, "per_launch": v["sum"] / v["launches"], "launches_per_proof": v["launches"] / proofs}
        print("total per proof: %.2f" % (total / proofs))
    if out_json:
        json.dump(res, open(out_json, "w"), indent=1)


if __name__ == "__main__":
    main()
