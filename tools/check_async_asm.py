#!/usr/bin/env python3
"""Build-time guard for the hand-pipelined kernels of csrc/sparse_conv.hip.

Those kernels issue their global loads as inline asm (so that the compiler neither moves them nor waits for them early) and
wait for them with a hand-placed `s_waitcnt vmcnt(0)`.  The compiler does not know that the destination registers of such a
load are in flight: if its register allocator copies one of them (a phi at a control-flow merge, a loop rotation) or reuses it
between the load and the wait, the copy holds stale data and the kernel computes garbage — silently, and any unrelated edit
can provoke it (it happened when conv_rows_ksplit got an outer loop).  This script compiles the file to gfx950 assembly and
checks, for every kernel, that no instruction reads or writes the destination registers of an inline-asm load before the next
`s_waitcnt vmcnt(0)` on every straight-line path the assembly lists (conservative: branches are followed in listing order,
a pending set is carried across labels).

    python tools/check_async_asm.py [file.hip ...]      exit code 1 on a violation
"""
import os
import re
import subprocess
import sys
import tempfile

REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
PKG = os.path.join(REPO, "from-voxel-to-point_amd")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-ffp-contract=off", "-mllvm", "-amdgpu-mfma-vgpr-form=1",
         "-S", "--cuda-device-only", "-I" + os.path.join(REPO, "include"), "-Wno-unused-command-line-argument"]


def regs(tok):
    m = re.match(r"v\[(\d+):(\d+)\]", tok)
    if m:
        return set(range(int(m.group(1)), int(m.group(2)) + 1))
    m = re.match(r"v(\d+)$", tok)
    return {int(m.group(1))} if m else set()


def check_asm(text):
    """-> list of (kernel, line number, instruction, line of the load).  Per kernel: basic blocks from the labels and branches of
    the listing, forward may-analysis of "registers with an un-waited inline-asm load" to a fixed point, then one pass that reports
    every instruction touching such a register.  `s_waitcnt vmcnt(0)` clears the set; a HAND-PLACED (inline asm) `s_waitcnt vmcnt(N)`
    retires every pending load that has at least N inline-asm loads issued after it (loads return in issue order; other vector-memory
    instructions in between only make the real age larger, so the rule is conservative; the kernels issue no stores inside their
    pipelined loops).  The compiler's own vmcnt(N > 0) waits count loads it does not know about and promise nothing for these."""
    lines = text.split("\n")
    out = []
    i = 0
    while i < len(lines):
        m = re.match(r"^([A-Za-z_][\w.$]*):", lines[i])
        if not m or lines[i].startswith(".L"):
            i += 1
            continue
        kernel, j = m.group(1), i + 1
        while j < len(lines) and lines[j].strip().split(";")[0].strip() != "s_endpgm" and not lines[j].startswith(".Lfunc_end"):
            j += 1
        out += _check_kernel(kernel, lines, i + 1, j)
        i = j + 1
    return out


def _check_kernel(kernel, lines, lo, hi):
    # instructions: (line no, op, operands, in_asm); blocks start at labels and after branches
    blocks, labels, cur, in_asm = [], {}, None, False

    def new_block():
        nonlocal cur
        cur = {"ins": [], "succ": [], "fall": True}
        blocks.append(cur)
    new_block()
    for no in range(lo, hi):
        line = lines[no]
        s = line.strip()
        lm = re.match(r"^(\.LBB\w+):", line)
        if lm:
            if cur["ins"] or cur is blocks[0]:
                new_block()
            labels[lm.group(1)] = len(blocks) - 1
            continue
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        if not s or s[0] in ";.":
            continue
        ins = s.split(";")[0].strip()
        parts = [p for p in re.split(r"[ ,]+", ins) if p]
        op, ops = parts[0], parts[1:]
        cur["ins"].append((no + 1, op, ops, in_asm, ins))
        if op.startswith("s_cbranch") or op == "s_branch":
            cur["succ"].append(ops[-1])
            cur["fall"] = op != "s_branch"
            new_block()
    n = len(blocks)
    succ = []
    for b, blk in enumerate(blocks):
        t = [labels[x] for x in blk["succ"] if x in labels]
        if blk["fall"] and b + 1 < n:
            t.append(b + 1)
        succ.append(t)

    AGE_CAP = 255

    def transfer(pending, blk, report=None):
        # pending: register -> (line of its load, age = inline-asm loads issued after that load on the youngest path)
        pending = dict(pending)
        for no, op, ops, in_asm, ins in blk["ins"]:
            if op == "s_waitcnt":
                m = re.search(r"vmcnt\((\d+)\)", ins)
                if m and int(m.group(1)) == 0:
                    pending = {}
                elif m and in_asm:
                    keep = int(m.group(1))
                    pending = {r: v for r, v in pending.items() if v[1] < keep}
                continue
            if in_asm and (op.startswith("global_load") or op.startswith("buffer_load")) and "lds" not in ins:
                if report is not None:
                    for t in ops[1:]:
                        for r in regs(t):
                            if r in pending:
                                report.append((kernel, no, ins, pending[r][0]))
                pending = {r: (at, min(age + 1, AGE_CAP)) for r, (at, age) in pending.items()}
                for r in regs(ops[0]):
                    pending[r] = (no, 0)
                continue
            if report is not None:
                for t in ops:
                    hit = [r for r in regs(t) if r in pending]
                    if hit:
                        report.append((kernel, no, ins, pending[hit[0]][0]))
                        break
        return pending

    inn = [dict() for _ in range(n)]
    work = list(range(n))
    while work:
        b = work.pop()
        o = transfer(inn[b], blocks[b])
        for t in succ[b]:
            grew = False
            for r, (at, age) in o.items():
                if r not in inn[t] or age < inn[t][r][1]:     # the youngest load of a register over all paths: it stays pending longest
                    inn[t][r] = (at, age)
                    grew = True
            if grew and t not in work:
                work.append(t)
    report = []
    for b in range(n):
        transfer(inn[b], blocks[b], report)
    return report


def main(files):
    bad = 0
    for f in files:
        with tempfile.TemporaryDirectory() as d:
            asm = os.path.join(d, "k.s")
            subprocess.check_call(["/opt/rocm/bin/hipcc"] + FLAGS + [f, "-o", asm], stderr=subprocess.DEVNULL)
            text = open(asm).read()
        v = check_asm(text)
        for k, no, ins, at in v[:40]:
            print(f"{os.path.basename(f)}: {k}: line {no}: `{ins}` touches a register whose inline-asm load (line {at}) has not been waited for")
        print(f"{os.path.basename(f)}: {len(v)} violation(s)")
        bad += len(v)
    return 1 if bad else 0


if __name__ == "__main__":
    args = sys.argv[1:]
    if args and args[0].endswith(".s"):
        v = check_asm(open(args[0]).read())
        for k, no, ins, at in v[:40]:
            print(f"{k}: line {no}: `{ins}` (load at line {at})")
        print(len(v), "violation(s)")
        sys.exit(1 if v else 0)
    sys.exit(main(args or [os.path.join(PKG, "csrc", "sparse_conv.hip")]))
