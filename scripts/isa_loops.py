#!/usr/bin/env python3
"""Where the scratch (spill) accesses of a kernel sit: disassembles the gfx950 code object inside libadvntr_hip.so, finds the loops
of one kernel (a backward branch closes a loop) and prints, per loop, its instruction census -- fp64 arithmetic, LDS reads, scalar
stores, and scratch_ / buffer_ accesses -- plus the kernel's resource note (VGPRs, spills).  Writes the excerpt DESIGN.md cites.
  python3 scripts/isa_loops.py 'viterbi_rows_kernelILi5ELi2E' > profiles/r06_rows_5_2_isa_loop.txt"""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
LLVM = "/opt/rocm/lib/llvm/bin"


def main():
    want = sys.argv[1] if len(sys.argv) > 1 else "viterbi_rows_kernelILi5ELi2E"
    lib = sys.argv[2] if len(sys.argv) > 2 else os.path.join(ROOT, "advntr_amd", "libadvntr_hip.so")
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        so = os.path.join(tmp, "lib.so")
        shutil.copy(lib, so)
        subprocess.run([LLVM + "/llvm-objdump", "--offloading", so], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        co = [f for f in os.listdir(tmp) if "amdgcn" in f]
        if not co:
            sys.exit("no gfx950 code object found in " + lib)
        co = os.path.join(tmp, co[0])
        asm = subprocess.run([LLVM + "/llvm-objdump", "-d", "--no-show-raw-insn", co], stdout=subprocess.PIPE,
                             stderr=subprocess.DEVNULL, universal_newlines=True).stdout
        notes = subprocess.run([LLVM + "/llvm-readelf", "--notes", co], stdout=subprocess.PIPE, stderr=subprocess.DEVNULL,
                               universal_newlines=True).stdout
    finally:
        shutil.rmtree(tmp, ignore_errors=True)
    # the kernel's body
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^[0-9a-f]+ <.*%s.*>:" % re.escape(want), l))
    name = re.search(r"<(.*)>", lines[start]).group(1)
    body = []
    for l in lines[start + 1:]:
        if re.match(r"^[0-9a-f]+ <", l):
            break
        m = re.match(r"^\s*(\S.*?)\s*//\s*([0-9A-Fa-f]+):", l)
        if m:
            body.append((int(m.group(2), 16), m.group(1)))
    addr_index = {a: i for i, (a, _) in enumerate(body)}
    # loops: a branch whose target lies at or before it
    loops = []
    for i, (a, ins) in enumerate(body):
        m = re.match(r"s_cbranch_\w+\s+(\S+)|s_branch\s+(\S+)", ins)
        if not m:
            continue
        tgt = m.group(1) or m.group(2)
        # llvm-objdump prints the target as a signed word offset from the next instruction
        try:
            off = int(tgt, 0)
        except ValueError:
            continue
        if off >= 0x8000:
            off -= 0x10000
        t = a + 4 + 4 * off
        if t <= a and t in addr_index:
            loops.append((addr_index[t], i))

    def census(lo, hi):
        c = collections.Counter()
        for _, ins in body[lo:hi + 1]:
            op = ins.split()[0]
            if op.startswith(("v_add_f64", "v_max_f64", "v_cmp_gt_f64", "v_fma_f64", "v_mul_f64")):
                c["fp64 " + op.split("_e")[0]] += 1
            elif op.startswith("scratch_"):
                c["SCRATCH " + op] += 1
            elif op.startswith("buffer_"):
                c["BUFFER " + op] += 1
            elif op.startswith("ds_"):
                c["lds " + op] += 1
            elif op.startswith("s_store"):
                c["scalar store " + op] += 1
            elif op.startswith("global_"):
                c["global " + op] += 1
            elif op.startswith("v_"):
                c["other VALU"] += 1
            elif op.startswith("s_"):
                c["scalar"] += 1
        return c

    kd = re.search(r"\.name:\s+%s.*?(?=\n\s+- \.|\Z)" % re.escape(name), notes, re.S)
    print("kernel  %s" % name)
    print("library %s" % os.path.relpath(lib, ROOT))
    if kd:
        for key in (".vgpr_count", ".vgpr_spill_count", ".sgpr_count", ".sgpr_spill_count", ".private_segment_fixed_size",
                    ".group_segment_fixed_size"):
            m = re.search(r"%s:\s+(\d+)" % re.escape(key), kd.group(0))
            if m:
                print("  %-30s %s" % (key, m.group(1)))
    whole = census(0, len(body) - 1)
    print("whole kernel: %d instructions; scratch accesses %d, buffer accesses %d" %
          (len(body), sum(v for k, v in whole.items() if k.startswith("SCRATCH")),
           sum(v for k, v in whole.items() if k.startswith("BUFFER"))))
    print()
    print("loops (closed by a backward branch), innermost bodies with fp64 arithmetic first:")
    seen = set()
    hot = sorted(loops, key=lambda p: -sum(v for k, v in census(*p).items() if k.startswith("fp64")) / max(1, p[1] - p[0]))
    for lo, hi in hot:
        c = census(lo, hi)
        f64 = sum(v for k, v in c.items() if k.startswith("fp64"))
        if f64 < 20 or (lo, hi) in seen:
            continue
        seen.add((lo, hi))
        scr = sum(v for k, v in c.items() if k.startswith(("SCRATCH", "BUFFER")))
        print("  0x%x .. 0x%x  %5d instructions  fp64 %4d  lds %3d  scalar stores %3d  global %2d  scratch_/buffer_ %d"
              % (body[lo][0], body[hi][0], hi - lo + 1, f64, sum(v for k, v in c.items() if k.startswith("lds")),
                 sum(v for k, v in c.items() if k.startswith("scalar store")), sum(v for k, v in c.items() if k.startswith("global")),
                 scr))
        for k in sorted(c):
            if k.startswith(("fp64", "SCRATCH", "BUFFER", "lds", "scalar store", "global")):
                print("        %-40s %d" % (k, c[k]))
    print()
    # (a step loop is mostly fp64 arithmetic; the tile / round loops around the sweep also hold the set-up and the finish phase)
    def f64_of(lo, hi):
        return sum(v for k, v in census(lo, hi).items() if k.startswith("fp64"))
    step_loops = [(lo, hi) for lo, hi in seen if f64_of(lo, hi) >= 100 and f64_of(lo, hi) >= 0.35 * (hi - lo + 1)]
    span = (min(body[lo][0] for lo, _ in step_loops), max(body[hi][0] for _, hi in step_loops)) if step_loops else (0, 0)
    print("the sweep's step loops (>= 100 fp64 instructions, >= 35 %% of the body) span 0x%x .. 0x%x" % span)
    print("scratch accesses of the kernel, by address, and whether one lies inside a step loop:")
    n_in = 0
    for a, ins in body:
        if ins.startswith(("scratch_", "buffer_")):
            inside = any(body[lo][0] <= a <= body[hi][0] for lo, hi in step_loops)
            n_in += inside
            print("  0x%x  %-56s %s" % (a, ins, "INSIDE A STEP LOOP" if inside else "outside (tile / round loop, set-up, finish phase)"))
    print("scratch_/buffer_ accesses inside the step loops: %d" % n_in)


if __name__ == "__main__":
    main()
