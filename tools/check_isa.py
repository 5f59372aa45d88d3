#!/usr/bin/env python3
"""Static check of the shipped library (no GPU needed): no packed fp32 instruction (v_pk_*_f32) may select the HIGH register
of src1 for its LOW result (op_sel bit 1).

Why: on the MI355X such instructions - the SLP vectoriser emits them when the operands of two paired scalar operations sit
crosswise in a register pair - intermittently return a wrong low result in lanes 32-63 (round 3: rare, run-dependent wrong
rows in the tanh / outer instantiations).  amber_amd/csrc/pkfix.py rewrites them between compiler and
assembler (operands exchanged: the same selection on src0 is sound); this check fails the build test if any kernel of the
library still holds one - a translation unit built around the Makefile, a new instruction form the rewrite does not know.

usage: tools/check_isa.py [library or object files ...]    (default: amber_amd/csrc/libamber_mpn.so)
"""
import os
import re
import struct
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "amber_amd", "csrc")
OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def device_code_objects(path):
    """The gfx950 code objects embedded in a host object or shared library (clang offload bundles inside .hip_fatbin)."""
    blob = open(path, "rb").read()
    out, pos = [], 0
    while True:
        pos = blob.find(MAGIC, pos)
        if pos < 0:
            return out
        n, = struct.unpack_from("<Q", blob, pos + len(MAGIC))
        cur = pos + len(MAGIC) + 8
        for _ in range(n):
            off, size, tlen = struct.unpack_from("<QQQ", blob, cur)
            triple = blob[cur + 24:cur + 24 + tlen].decode()
            cur += 24 + tlen
            if "gfx950" in triple and size > 0:
                out.append(blob[pos + off:pos + off + size])
        pos += len(MAGIC)


def scan(disassembly):
    """{kernel: [instruction, ...]} of the packed fp32 instructions with op_sel bit 1 set"""
    findings, kernel = {}, None
    for line in disassembly.splitlines():
        m = re.match(r"^[0-9a-f]+ <(\S+)>:", line)
        if m:
            kernel = m.group(1)
            continue
        text = line.split("//")[0].strip()
        if not kernel or not text.startswith("v_pk_") or not text.split()[0].endswith("_f32"):
            continue
        m = re.search(r"\bop_sel:\[([0-9,]+)\]", text)
        if m:
            sel = m.group(1).split(",")
            if len(sel) > 1 and sel[1] == "1":
                findings.setdefault(kernel, []).append(text)
    return findings


def check(paths):
    bad = {}
    kernels = packed = 0
    for path in paths:
        cos = device_code_objects(path)
        if not cos:
            raise SystemExit(f"{path}: no gfx950 code object found")
        for co in cos:
            with tempfile.NamedTemporaryFile(suffix=".co") as f:
                f.write(co)
                f.flush()
                dis = subprocess.run([OBJDUMP, "-d", "--no-show-raw-insn", f.name], capture_output=True, text=True, check=True).stdout
            kernels += len(re.findall(r"^[0-9a-f]+ <\S+>:", dis, flags=re.M))
            packed += len(re.findall(r"^\s*v_pk_\w+_f32\b", dis, flags=re.M))
            for k, v in scan(dis).items():
                bad[(os.path.basename(path), k)] = v
    return kernels, packed, bad


def main():
    paths = sys.argv[1:] or [os.path.join(CSRC, "libamber_mpn.so")]
    kernels, packed, bad = check(paths)
    for (obj, k), v in sorted(bad.items()):
        name = subprocess.run(["c++filt", k], capture_output=True, text=True).stdout.strip()
        print(f"{obj}: {name}: {len(v)} packed fp32 instruction(s) take src1's high register for the low result, e.g. {v[0]}")
    print(f"{kernels} kernels, {packed} packed fp32 instructions checked; {len(bad)} kernels hold the unsound form")
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
