#!/usr/bin/env python3
"""Looks through the device code of libmdisp.so for a store whose data registers are overwritten by the very next
vector instruction.

For a 12- or 16-byte buffer store with an immediate scalar offset the compiler leaves the wait state the hardware needs
between the store and a vector instruction that overwrites its data registers; for the form with the scalar offset in
a register it does not, and on gfx950 the store then picks up the new register contents under load (seen in
sgm3.hip, which now places an s_nop behind such stores).  This check disassembles every gfx950 code object of the
library and reports each buffer_store_dwordx3 / x4 with a register scalar offset that is directly followed by a vector
instruction writing one of its data registers.

    python scripts/check_store_hazard.py [path to libmdisp.so]      # exit status 1 if any is found
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = os.environ.get("LLVM_BIN", "/opt/rocm/lib/llvm/bin")
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib, tmp):
    """The gfx950 code objects of every translation unit linked into the library."""
    fat = os.path.join(tmp, "fat.bin")
    subprocess.check_call([os.path.join(LLVM, "llvm-objcopy"), "--dump-section=.hip_fatbin=" + fat, lib])
    blob = open(fat, "rb").read()
    starts = [m.start() for m in re.finditer(re.escape(MAGIC), blob)]
    out = []
    for i, a in enumerate(starts):
        b = starts[i + 1] if i + 1 < len(starts) else len(blob)
        part = os.path.join(tmp, "bundle%d.bin" % i)
        open(part, "wb").write(blob[a:b])
        co = os.path.join(tmp, "co%d.o" % i)
        r = subprocess.run([os.path.join(LLVM, "clang-offload-bundler"), "--unbundle", "--type=o", "--input=" + part,
                            "--targets=hipv4-amdgcn-amd-amdhsa--gfx950", "--output=" + co], capture_output=True)
        if r.returncode == 0 and os.path.exists(co) and os.path.getsize(co) > 0:
            out.append(co)
    return out


def regs(operand):
    m = re.match(r"v\[(\d+):(\d+)\]", operand)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", operand)
    return {int(m.group(1))} if m else set()


def scan(co):
    return scan_text(subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--no-show-raw-insn", co], capture_output=True, text=True).stdout)


def scan_text(txt):
    """(offending pairs, number of wide buffer stores with a register scalar offset) of a disassembly listing."""
    found, stores, func = [], 0, "?"
    lines = [l for l in txt.split("\n")]
    prev = None
    for l in lines:
        m = re.match(r"^[0-9a-f]+ <(.+)>:", l)
        if m:
            func, prev = m.group(1), None
            continue
        ins = l.strip().split("//")[0].strip()
        if not ins or ins.endswith(":"):
            continue
        if prev is not None:
            op = ins.split()[0]
            if op.startswith("v_") and not op.startswith("v_cmp") and not op.startswith("v_readlane") and not op.startswith("v_readfirstlane"):
                dst = ins[len(op):].split(",")[0].strip()
                if regs(dst) & prev[1]:
                    found.append((func, prev[0], ins))
            prev = None
        m = re.match(r"buffer_store_dwordx[34]\s+(v\[\d+:\d+\]),\s*[^,]+,\s*s\[\d+:\d+\],\s*(s\d+|m0|vcc_lo|vcc_hi)\b", ins)
        if m:
            stores += 1
            prev = (ins, regs(m.group(1)))
    return found, stores


def main():
    lib = sys.argv[1] if len(sys.argv) > 1 else os.path.join(ROOT, "cameracalibration_amd", "lib", "libmdisp.so")
    with tempfile.TemporaryDirectory() as tmp:
        found, stores, n = [], 0, 0
        for co in code_objects(lib, tmp):
            f, s = scan(co)
            found += f
            stores += s
            n += 1
    print(f"{n} code objects, {stores} wide buffer stores with a register scalar offset, {len(found)} followed by an overwrite of their data")
    for func, st, ins in found:
        print("  ", func[:90], "|", st, "->", ins)
    return 1 if found or n == 0 else 0


if __name__ == "__main__":
    sys.exit(main())
