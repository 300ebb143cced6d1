"""Audit of the hand-issued (inline asm) VMEM loads of the wide kernels.

hipcc treats an asm load's destination as written at the end of the asm statement, so nothing stops it from
copying / re-using that register while the load is still in flight (cdna_hip_programming.md §5.7).  The kernels
wait with their own `s_waitcnt vmcnt(N)` asm statement that names every destination register; this script
checks, on the generated ISA, that between an asm load and the next asm vmcnt wait naming its destination no
compiler-generated instruction touches that destination register.  Linear scan per kernel in layout order,
restarted at every asm wait; returns the list of violations.

usage: python scripts/audit_asm_loads.py brainmagick_amd/csrc/conv_nn_h2w.hip [...]
"""
import re
import subprocess
import sys
import tempfile
from pathlib import Path


def regs_of(text):
    out = set()
    for lo, hi in re.findall(r"\bv\[(\d+):(\d+)\]", text):
        out.update(range(int(lo), int(hi) + 1))
    out.update(int(r) for r in re.findall(r"\bv(\d+)\b", text))
    return out


def audit_text(asm: str):
    violations = []
    for m in re.finditer(r"^(\w+):\s*; @\1\n(.*?)^\.Lfunc_end", asm, flags=re.M | re.S):
        name, body = m.group(1), m.group(2).splitlines()
        inflight = {}
        in_asm = False
        is_barrier = False
        for ln, line in enumerate(body):
            t = line.strip()
            if t.startswith(";;#ASMSTART"):
                in_asm = True
                # the barrier statement's vmcnt(N) only covers the weight slab: it is not a wait for the staged set
                block = []
                for nxt in body[ln + 1:]:
                    if nxt.strip().startswith(";;#ASMEND"):
                        break
                    block.append(nxt)
                is_barrier = any("s_barrier" in b for b in block)
                continue
            if t.startswith(";;#ASMEND"):
                in_asm = False
                continue
            if not t or t.startswith(";") or t.startswith("."):
                continue
            if in_asm:
                if t.startswith("buffer_load") and " lds" not in t:
                    dst = t.split()[1].rstrip(",")
                    for r in regs_of(dst):
                        inflight[r] = ln
                elif "s_waitcnt" in t and "vmcnt" in t and (not is_barrier or "vmcnt(0)" in t):
                    inflight.clear()          # (a barrier statement only counts when it drains the queue)          # the kernels' waits name every staged register of the set
                continue
            if inflight:
                hit = regs_of(t) & set(inflight)
                if hit:
                    violations.append((name, ln, t, sorted(hit)))
    return violations


def audit_file(src: Path):
    hipcc = "/opt/rocm/bin/hipcc"
    with tempfile.TemporaryDirectory() as tmp:
        out = Path(tmp) / "k.s"
        subprocess.run([hipcc, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only",
                        f"-I{src.parent}", "-o", str(out), str(src)], check=True, capture_output=True)
        return audit_text(out.read_text())


if __name__ == "__main__":
    bad = 0
    for f in sys.argv[1:]:
        v = audit_file(Path(f))
        print(f, "violations:", len(v))
        for item in v[:10]:
            print("   ", item)
        bad += len(v)
    sys.exit(1 if bad else 0)
