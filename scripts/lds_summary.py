#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_LDS SQ_ACTIVE_INST_LDS of
bench.py: per kernel, the share of LDS-active cycles lost to bank conflicts.

usage: lds_summary.py <counter_collection.csv> <out.json>"""
import collections
import csv
import json
import re
import sys


def main():
    path, out = sys.argv[1], sys.argv[2]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    for r in csv.DictReader(open(path)):
        m = re.search(r"(sgm[23]_\w+<[^>]*>|sgm[23]_\w+|census3x3_kernel|pad_reflect_lr_kernel)", r["Kernel_Name"])
        if m:
            agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
    rows = []
    for k, c in agg.items():
        act = c["SQ_LDS_IDX_ACTIVE"]
        rows.append({"kernel": k, "lds_instructions": c["SQ_INSTS_LDS"], "lds_active_cycles": act,
                     "bank_conflict_cycles": c["SQ_LDS_BANK_CONFLICT"],
                     "bank_conflict_share": c["SQ_LDS_BANK_CONFLICT"] / act if act else 0.0})
    rows.sort(key=lambda r: -r["lds_active_cycles"])
    json.dump({"kernels": rows}, open(out, "w"), indent=1)
    for r in rows:
        print(f"{r['kernel']:55s} LDS active {r['lds_active_cycles']:.3g} cycles, bank conflicts {100 * r['bank_conflict_share']:.1f} %")


if __name__ == "__main__":
    main()
