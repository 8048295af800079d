#!/usr/bin/env python3
"""Check the hand-issued LDS reads of the HIP kernels against the compiler's register copies.

The kernels issue `ds_read_*` through inline asm and wait for them later with an asm `s_waitcnt lgkmcnt(0)` that names the
destination registers as "+v" operands.  That tie stops the compiler from USING a register before the wait -- it does not stop it
from COPYING one ahead of the wait when it cannot keep the value in one physical register across the two statements (seen with
run-time branches around the reads: proj_gemm.hip, round 4).  Such a copy reads whatever the register held before the data landed
and the result depends on timing.

This script compiles a source file to gfx950 assembly and walks every kernel linearly: a register an asm `ds_read` wrote is
"pending" until an `s_waitcnt lgkmcnt(0)` (asm or compiler-emitted); any compiler instruction that mentions a pending register is
reported.  The walk ignores control flow (labels keep the pending set), so a report is a place to look at, not a proof.

A second pattern, also met in round 4 (proj_gemm.hip with 64-bit item indices): `s_cselect` reading SCC after a 64-bit compare that
was moved to the VALU (`v_cmp_*64` writing vcc) -- SCC then still holds the carry of the `s_addc_u32` in front of it and the select
takes the wrong side.  Reported as "select on a stale SCC".

    python tools/check_asm_waits.py transkun_amd/csrc/proj_gemm.hip [more.hip ...]
exit status 1 if anything was reported."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
REG = re.compile(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b")


def regs_of(text):
    out = set()
    for m in REG.finditer(text):
        if m.group(3) is not None:
            out.add(int(m.group(3)))
        else:
            out.update(range(int(m.group(1)), int(m.group(2)) + 1))
    return out


def compile_to_asm(src, extra):
    fd, path = tempfile.mkstemp(suffix=".s")
    os.close(fd)
    cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
           "-I" + os.path.join(ROOT, "include"), "-I" + os.path.join(ROOT, "transkun_amd", "csrc")] + extra + [src, "-o", path]
    subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
    with open(path) as f:
        text = f.read()
    os.unlink(path)
    return text


def check(text, name):
    kernel = None
    pending = {}                       # register -> line number of the read
    in_asm = False
    reports = []
    scc_from_addc = False              # the last SCC writer was s_addc_u32/s_subb_u32 (a 64-bit add's upper half)
    vcmp64_since = False
    for ln, raw in enumerate(text.splitlines(), 1):
        line = raw.split(";", 1)[0].strip() if not raw.lstrip().startswith(";;#") else raw.strip()
        lab = re.match(r"^(_Z\w+):", raw)
        if lab:
            kernel = lab.group(1)
            pending = {}
            continue
        if line.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if line.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not line or line.endswith(":") or line.startswith("."):
            continue
        if line.startswith("s_endpgm"):
            pending = {}
            continue
        if "s_waitcnt" in line and "lgkmcnt(0)" in line:
            pending = {}
            continue
        if in_asm:
            m = re.match(r"ds_read\w*\s+(v\[\d+:\d+\]|v\d+)", line)
            if m:
                for r in regs_of(m.group(1)):
                    pending[r] = ln
            continue
        mn = line.split()[0]
        if mn.startswith("s_") and not mn.startswith(("s_cselect", "s_cbranch", "s_mov", "s_load", "s_waitcnt", "s_nop", "s_branch", "s_barrier",
                                                      "s_sleep", "s_setprio", "s_getreg", "s_setreg", "s_memtime", "s_memrealtime")):
            scc_from_addc = mn in ("s_addc_u32", "s_subb_u32")
            vcmp64_since = False
        elif re.match(r"v_cmp\w*_[iu]64", mn) and scc_from_addc:
            vcmp64_since = True
        elif mn.startswith("s_cselect") and scc_from_addc and vcmp64_since:
            reports.append((kernel, ln, line, ["select on a stale SCC"]))
        hit = regs_of(line) & set(pending)
        if hit:
            reports.append((kernel, ln, line, sorted(hit)))
    for k, ln, line, hit in reports:
        what = hit[0] if hit and isinstance(hit[0], str) else f"touches v{hit} ahead of the wait"
        print(f"{name}: {k}: line {ln}: `{line}` {what}")
    return len(reports)


def main(argv):
    if not argv:
        print(__doc__)
        return 2
    bad = 0
    extra = [a for a in argv if a.startswith("-")]
    for src in [a for a in argv if not a.startswith("-")]:
        text = open(src).read() if src.endswith(".s") else compile_to_asm(src, extra)
        n = check(text, os.path.basename(src))
        print(f"{os.path.basename(src)}: {n} suspicious instruction(s)")
        bad += n
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
