#!/usr/bin/env python3
"""Issue bound of the headline SGM kernels as (instructions per step, by opcode) x (measured issue cost of that opcode).

    python scripts/issue_bound.py <ledger.json> <ubench.json> <out.json> <clock GHz> [sq_issue.json]

ledger.json: scripts/isa_ledger.py (opcode counts of the compute waves' main loop, two steps per iteration, for the first
and the second sweep); ubench.json: scripts/valu_ubench4.hip run on the GPU (shader cycles per wave-instruction per SIMD with
all SIMDs busy, 5 waves per SIMD as the row-band kernel keeps); clock: the shader clock the kernel itself runs at
(GRBM_GUI_ACTIVE of its launches / 8 XCDs / their duration -- the microbenchmark's own clock is lower, it draws more power);
sq_issue.json (scripts/sq_summary.py): the counted vector instructions per pair, to which the static count is scaled (the
image-border branches inside the loop are rarely taken).  A wave-step handles 4 pixels x 256 disparities; a band of
16 rows takes W + 30 steps for W pixels per row and a pair has ceil(H / 16) bands per sweep, each swept once as the first
and once as the second sweep (the two directions split the image).  Every wave-instruction occupies the issue port of
its SIMD for the measured time whatever else is resident, so the sum over the pair divided by the 1024 SIMDs is a floor
for this instruction stream.  Scalar instructions issue from another port and are left out, as are LDS and memory
instructions (their issue slots are a few per cent and overlap); s_nop is counted: it holds the wave's issue slot."""
import json
import math
import re
import sys

ALIASES = [(r"^v_add_u32", "v_add_u32"), (r"^v_sub_u32|^v_subrev_u32", "v_sub_u32"), (r"^v_and_b32", "v_and_b32"),
           (r"^v_cndmask_b32", "v_cndmask_b32"), (r"^v_mov_b32_dpp", "v_mov_b32_dpp"), (r"^v_mov_b32", "v_mov_b32"),
           (r"^v_mul_u32_u24", "v_mul_u32_u24"), (r"^v_mul_i32_i24", "v_mul_i32_i24"), (r"^v_lshlrev_b32|^v_lshrrev_b32", "v_lshlrev_b32"),
           (r"^v_min_u32_dpp", "v_min_u32_dpp"), (r"^v_min_u32_sdwa|^v_sub_u32_sdwa|^v_or_b32_sdwa", "v_min_u32_sdwa"),
           (r"^v_min_u32|^v_max_u32|^v_min_i32|^v_max_i32", "v_max_i32"), (r"^v_cmp", "v_cmp"), (r"^v_or_b32", "v_or_b32"),
           (r"^v_lshl_add_u64", "v_lshl_add_u32"), (r"^s_nop", "s_nop")]


def cost_name(op, table):
    base = re.sub(r"_e32$|_e64$", "", op)
    if base in table:
        return base
    for pat, name in ALIASES:
        if re.search(pat, op):
            return name
    return None


def main():
    ledger, ub, out = json.load(open(sys.argv[1])), json.load(open(sys.argv[2])), sys.argv[3]
    H, W = 1080, 1920
    ghz = float(sys.argv[4])
    sq = json.load(open(sys.argv[5])) if len(sys.argv) > 5 else None
    table = ub["waves_per_simd_5"]
    bands = math.ceil(H / 16)
    steps_per_sweep = bands * 4 * (W + 30)          # four compute waves per band, W + 2 * 15 steps each
    report = {"image": [H, W], "wave_steps_per_pair_per_sweep": steps_per_sweep, "simds": 1024, "sweeps": {}}
    total_ns, total_valu = 0.0, 0.0
    for title, k in ledger.items():
        rows, ns_step, n_vec, unknown = [], 0.0, 0, {}
        for op, c2 in k["by_opcode_per_two_steps"].items():
            if not (op.startswith("v_") or op == "s_nop"):
                continue
            name = cost_name(op, table)
            if name is None:
                unknown[op] = c2 / 2
                name = "v_pk_min_u16"          # priced as the slow class
            ns = table[name]["cycles"] / ghz
            rows.append({"opcode": op, "per_step": c2 / 2, "priced_as": name, "ns_each": ns, "cycles_each": table[name]["cycles"], "ns_per_step": c2 / 2 * ns})
            ns_step += c2 / 2 * ns
            n_vec += c2 / 2
        rows.sort(key=lambda r: -r["ns_per_step"])
        ms = ns_step * steps_per_sweep / 1024 * 1e-6
        report["sweeps"][title] = {"vector_instructions_per_step": n_vec, "ns_per_step_of_issue": ns_step, "mean_ns_per_instruction": ns_step / n_vec,
                                   "ms_per_pair": ms, "unpriced_opcodes_taken_as_slow": unknown, "by_opcode": rows}
        total_ns += ms
        total_valu += sum(r["per_step"] for r in rows if r["opcode"].startswith("v_")) * steps_per_sweep
    report["ms_per_pair_at_measured_issue_cost"] = total_ns
    report["kernel_clock_ghz"] = ghz
    report["clock_ghz_during_microbenchmark"] = table["v_pk_min_u16"]["ghz"]
    report["static_valu_per_pair"] = total_valu
    if sq:
        counted = next(k["valu_instructions_per_pair"] for k in sq["kernels"] if "meet" in k["kernel"])
        report["counted_valu_per_pair"] = counted
        report["ms_per_pair_scaled_to_counted_instructions"] = total_ns * counted / total_valu
        report["flat_4_cycle_convention_ms_per_pair"] = counted * 4.0 / (1024 * ghz * 1e9) * 1e3
    json.dump(report, open(out, "w"), indent=1)
    for t, s in report["sweeps"].items():
        print(f"{t}: {s['vector_instructions_per_step']:.1f} vector instructions per step, {s['ns_per_step_of_issue']:.1f} ns of issue "
              f"({s['mean_ns_per_instruction']:.3f} ns each) -> {s['ms_per_pair']:.4f} ms per pair")
        for r in s["by_opcode"][:12]:
            print(f"   {r['per_step']:6.1f} x {r['opcode']:22s} {r['cycles_each']:.2f} cycles = {r['ns_per_step']:.1f} ns")
    print(f"issue time of this instruction stream at {ghz} GHz: {total_ns:.4f} ms per pair (static count {total_valu / 1e6:.1f} M vector instructions)")
    if sq:
        print(f"scaled to the counted {report['counted_valu_per_pair'] / 1e6:.1f} M: {report['ms_per_pair_scaled_to_counted_instructions']:.4f} ms per pair; "
              f"flat 4-cycle convention: {report['flat_4_cycle_convention_ms_per_pair']:.4f} ms")


if __name__ == "__main__":
    main()
