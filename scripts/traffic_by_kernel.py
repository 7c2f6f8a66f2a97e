#!/usr/bin/env python3
"""Per-kernel time and HBM traffic of any rocprofv3 target, from three runs of the same command:
a --kernel-trace --stats run and two --pmc passes (FETCH_SIZE, WRITE_SIZE).

usage: traffic_by_kernel.py <kernel_stats.csv> <fetch counter_collection.csv> <write counter_collection.csv> <out.json> [min share %]

Counter values are KiB per launch; FETCH_SIZE is doubled (on gfx950 it tallies the 128-byte requests of wide coalesced
reads as 64 bytes), WRITE_SIZE is taken as it is."""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        agg[r["Kernel_Name"]][0] += 1
        agg[r["Kernel_Name"]][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] * 1024 for k, v in agg.items()}


def main():
    stats, fetch, write, out = sys.argv[1:5]
    floor = float(sys.argv[5]) if len(sys.argv) > 5 else 0.5
    f, w = per_kernel(fetch), per_kernel(write)
    rows = []
    for r in csv.DictReader(open(stats)):
        if float(r["Percentage"]) < floor:
            continue
        k = r["Name"]
        avg = float(r["AverageNs"])
        fb, wb = 2.0 * f.get(k, 0.0), w.get(k, 0.0)
        rows.append({"kernel": k[:140], "calls": int(r["Calls"]), "average_us": avg / 1e3, "share_percent": float(r["Percentage"]),
                     "fetch_bytes_per_launch": fb, "write_bytes_per_launch": wb,
                     "hbm_gb_per_s": (fb + wb) / avg if avg else 0.0})
    json.dump({"fetch_correction": 2.0, "kernels": rows}, open(out, "w"), indent=1)
    for r in rows:
        print(f"{r['kernel'][:70]:70s} {r['calls']:5d} x {r['average_us']:9.1f} us  {r['share_percent']:5.1f} %  "
              f"{(r['fetch_bytes_per_launch'] + r['write_bytes_per_launch']) / 1e6:9.1f} MB  {r['hbm_gb_per_s']:7.0f} GB/s")


if __name__ == "__main__":
    main()
