#!/usr/bin/env python3
"""Instruction ledger of the headline SGM kernel from the compiler's listing (hipcc -S of sgm3.hip, gfx950).

For the D = 256 one-launch row-band kernel (first sweep; second sweep with the winner-take-all) it takes the main loops of
the compute waves (two steps, i.e. two pixels per slot, per iteration), counts their instructions by opcode and groups the
opcodes by what they are used for in these kernels.  One wave-step handles 4 pixels x 256 disparities.

    python scripts/isa_ledger.py [out_prefix]      # writes <out_prefix>.json and <out_prefix>.txt (default profiles/sgm_isa_ledger)
"""
import collections
import json
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# the one-launch kernel holds four copies of the sweep body (down / up x first / second sweep): its compute-wave loops are
# the loops of more than 500 instructions; the second sweep's are the ones with the winner-take-all's packed multiply-add
MEET = "sgm3_meet_kernelILb0ELi16ELi8ELi256EEEvNS_7RowArgsE"
KERNELS = {
    "first sweep (stores the four-path sum)": (MEET, False),
    "second sweep + winner-take-all": (MEET, True),
}
GROUPS = [
    ("packed minimum (recursion: neighbours, three-input minimum; minimum trees of the paths and of the winner-take-all)",
     r"^v_pk_min_u16$|^v_pk_minimum3_f16$"),
    ("32-bit add / subtract on packed pairs (P1, matching cost, path sums, normalisation; a few address updates)",
     r"^v_add_u32|^v_sub_u32|^v_subrev_u32|^v_add3_u32$|^v_sad_u32$"),
    ("byte pack / unpack of the states that leave the registers and of the four-path sums", r"^v_perm_b32$|^v_and_b32"),
    ("matching cost (xor + mask, population count, pair assembly)", r"^v_bcnt_u32_b32$|^v_bitop3_b32$|^v_lshl_or_b32$|^v_xor_b32"),
    ("cross-lane (neighbouring lanes' edge disparities, minimum over the pixel's lanes)", r"_dpp$|_sdwa$|^v_alignbit_b32$|^v_readlane|^v_readfirstlane"),
    ("winner-take-all keys and tests (packed multiply-add, 24-bit multiplies, and-or masks)", r"^v_pk_mad_u16$|^v_mul_|^v_and_or_b32$|^v_mad_"),
    ("selects, moves, compares (image-border cases, bookkeeping)", r"^v_cndmask|^v_mov_b32|^v_cmp|^v_min_|^v_max_|^v_bfe|^v_lshl|^v_lshr|^v_ashr|^v_or_b32|^v_lshl_add"),
    ("LDS", r"^ds_"),
    ("global / buffer memory", r"^global_|^buffer_|^flat_"),
    ("scalar, waits, barrier, branches", r"^s_"),
]


def main():
    out = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "profiles", "sgm_isa_ledger")
    with tempfile.TemporaryDirectory() as tmp:
        asm = os.path.join(tmp, "sgm3.s")
        subprocess.check_call([os.environ.get("HIPCC", "/opt/rocm/bin/hipcc"), "--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off",
                               "-S", "--cuda-device-only", "-I" + os.path.join(ROOT, "include"),
                               os.path.join(ROOT, "cameracalibration_amd", "csrc", "sgm3.hip"), "-o", asm], stderr=subprocess.DEVNULL)
        lines = open(asm).read().split("\n")
    report, text = {}, []
    for title, (tag, second) in KERNELS.items():
        start = next(i for i, l in enumerate(lines) if l.startswith("_ZN") and tag in l and l.split(":")[0].endswith("E"))
        end = next(i for i in range(start + 1, len(lines)) if lines[i].startswith(".Lfunc_end"))
        body = lines[start:end]
        labels = {m.group(1): i for i, l in enumerate(body) for m in [re.match(r"^(\.LBB\d+_\d+):", l)] if m}
        loops = []
        for i, l in enumerate(body):
            m = re.search(r"\s(s_cbranch_\w+|s_branch)\s+(\.LBB\d+_\d+)", l)
            if m and m.group(2) in labels and labels[m.group(2)] < i:
                loops.append((i - labels[m.group(2)], labels[m.group(2)], i))
        big = [(n, a, b) for n, a, b in loops if n > 500 and not any(a <= a2 and b2 <= b and (a2, b2) != (a, b) and n2 > 500 for n2, a2, b2 in loops)]
        big = [(n, a, b) for n, a, b in big if sum("v_pk_minimum3_f16" in l for l in body[a:b + 1]) >= 50]      # (not the communication wave's)
        big = [(n, a, b) for n, a, b in big if any("v_pk_mad_u16" in l for l in body[a:b + 1]) == second]
        assert big, "no main loop found for " + title
        ops = collections.Counter()
        for _, a, b in big:                     # the down and the up copy: averaged
            for l in body[a:b + 1]:
                l = l.strip()
                if not l or l[0] in ";." or l.split()[0].endswith(":"):
                    continue
                ops[l.split()[0]] += 1
        ops = collections.Counter({o: c / len(big) for o, c in ops.items()})
        groups = collections.OrderedDict((g, 0) for g, _ in GROUPS)
        groups["other"] = 0
        for op, c in ops.items():
            for g, pat in GROUPS:
                if re.search(pat, op):
                    groups[g] += c
                    break
            else:
                groups["other"] += c
        valu = sum(c for o, c in ops.items() if o.startswith("v_"))
        regs = {}
        for l in lines[end:end + 60]:
            m = re.match(r";\s*(NumVgprs|ScratchSize|LDSByteSize|Occupancy):\s*(\d+)", l)
            if m:
                regs[m.group(1)] = int(m.group(2))
        report[title] = {"kernel": tag, "loops_averaged": len(big), "steps_per_loop_iteration": 2, "instructions_per_step": sum(ops.values()) / 2, "valu_per_step": valu / 2,
                         "by_use_per_step": {g: c / 2 for g, c in groups.items()}, "by_opcode_per_two_steps": dict(ops.most_common()), **regs}
        text.append(f"{title}\n  {tag}\n  {regs}\n  per step of one wave (4 pixels x 256 disparities): {sum(ops.values()) / 2:.1f} instructions, {valu / 2:.1f} VALU")
        for g, c in groups.items():
            text.append(f"    {c / 2:7.1f}  {g}")
        text.append("  opcodes (two steps): " + ", ".join(f"{o} {c:g}" for o, c in ops.most_common()))
        text.append("")
    with open(out + ".json", "w") as f:
        json.dump(report, f, indent=1)
    with open(out + ".txt", "w") as f:
        f.write("\n".join(text))
    print("\n".join(text))


if __name__ == "__main__":
    main()
