#!/usr/bin/env python3
"""Summarise a rocprofv3 --pmc pass with SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_WAIT_ANY SQ_WAIT_INST_ANY (+ any
others) of bench.py into VALU instructions per stereo pair and the issue-bound time they imply.

usage: sq_summary.py <counter_collection.csv> <pairs per launch> <out.json>

Counts are per launch (SQ_WAVES equals the number of waves launched).  One wave64 VALU instruction is priced at
4 cycles of its SIMD (measured: about 4.2 for packed, three-operand, min/max, shift, permute and bit-count
instructions, about 2.5 for plain 32-bit add / sub / and / or / xor, scripts/valu_ubench2.hip; the counter does not
tell the two kinds apart, so the figure is an estimate of the issue time, not a hard floor); the chip has 256 CUs x
4 SIMDs at 2.4 GHz."""
import collections
import csv
import json
import re
import sys

SIMDS, CLOCK, CYCLES_PER_INSTR = 1024, 2.4e9, 4.0


def main():
    path, pairs, out = sys.argv[1], int(sys.argv[2]), sys.argv[3]
    agg = collections.defaultdict(lambda: collections.defaultdict(float))
    launches = collections.Counter()
    for r in csv.DictReader(open(path)):
        m = re.search(r"(sgm[23]_\w+<[^>]*>|sgm[23]_\w+|census3x3_kernel|pad_reflect_lr_kernel)", r["Kernel_Name"])
        if not m:
            continue
        agg[m.group(1)][r["Counter_Name"]] += float(r["Counter_Value"])
        if r["Counter_Name"] == "SQ_WAVES":
            launches[m.group(1)] += 1
    batches = max(v for k, v in launches.items() if "shear" in k or "sweep" in k or "stagger" in k or "rows" in k or "meet" in k)
    rows, total = [], 0.0
    for k, c in agg.items():
        valu = c["SQ_INSTS_VALU"] / batches / pairs
        total += valu
        wc = c["SQ_WAVE_CYCLES"] or 1.0
        rows.append({"kernel": k, "valu_instructions_per_pair": valu, "valu_share_of_wave_cycles": c["SQ_INSTS_VALU"] / wc,
                     "wait_any_share": c["SQ_WAIT_ANY"] / wc, "wait_issue_share": c["SQ_WAIT_INST_ANY"] / wc})
    rows.sort(key=lambda r: -r["valu_instructions_per_pair"])
    issue_ms = total * CYCLES_PER_INSTR / (SIMDS * CLOCK) * 1e3
    json.dump({"pairs_per_launch": pairs, "valu_instructions_per_pair": total, "issue_bound_ms_per_pair": issue_ms,
               "cycles_per_wave_instruction": CYCLES_PER_INSTR, "simds": SIMDS, "clock_hz": CLOCK, "kernels": rows}, open(out, "w"), indent=1)
    print(f"{total/1e6:.1f} M VALU wave-instructions per pair -> {issue_ms:.3f} ms per pair at full issue")


if __name__ == "__main__":
    main()
