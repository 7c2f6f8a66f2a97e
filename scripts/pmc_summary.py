#!/usr/bin/env python3
"""Summarise rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of bench.py into HBM bytes per stereo pair.

usage: pmc_summary.py <fetch counter_collection.csv> <write counter_collection.csv> <pairs per launch> <out.json>

Counter values are KiB.  On gfx950 FETCH_SIZE tallies the 128-byte requests of wide coalesced reads
as 64 bytes, so fetched bytes are doubled; WRITE_SIZE is exact for 16-byte-per-lane stores."""
import collections
import csv
import json
import sys


def per_kernel(path):
    agg = collections.defaultdict(lambda: [0, 0.0])
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"]
        if "md_stream_copy" in k:        # bench.py's bandwidth probe, not part of the pipeline
            continue
        agg[k][0] += 1
        agg[k][1] += float(r["Counter_Value"])
    return {k: v[1] / v[0] for k, v in agg.items()}, {k: v[0] for k, v in agg.items()}


def main():
    fetch_csv, write_csv, pairs, out = sys.argv[1], sys.argv[2], int(sys.argv[3]), sys.argv[4]
    f, fn = per_kernel(fetch_csv)
    w, _ = per_kernel(write_csv)
    calls_per_batch = {k: (1 if ("sgm2" in k or "sgm3" in k) else None) for k in f}
    rows, tot_f, tot_w = [], 0.0, 0.0
    batches = max(n for k, n in fn.items() if ("sgm2" in k or "sgm3" in k))
    for k in sorted(f, key=lambda k: -(f[k] * fn[k])):
        launches_per_batch = fn[k] / batches
        fb = f[k] * 1024 * 2 * launches_per_batch / pairs
        wb = w.get(k, 0.0) * 1024 * launches_per_batch / pairs
        tot_f += fb
        tot_w += wb
        rows.append({"kernel": k[:100], "launches_per_batch": launches_per_batch,
                     "fetch_bytes_per_pair": fb, "write_bytes_per_pair": wb})
    json.dump({"pairs_per_launch": pairs, "fetch_correction": 2.0, "fetch_bytes_per_pair": tot_f,
               "write_bytes_per_pair": tot_w, "traffic_bytes_per_pair": tot_f + tot_w, "kernels": rows},
              open(out, "w"), indent=1)
    print(f"fetch {tot_f/1e9:.3f} GB + write {tot_w/1e9:.3f} GB = {(tot_f+tot_w)/1e9:.3f} GB per pair")


if __name__ == "__main__":
    main()
