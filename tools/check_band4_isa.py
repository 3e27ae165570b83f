#!/usr/bin/env python3
"""Safety check of the hand-managed stream loads in csrc/band4.hip and csrc/band4f.hip (see STREAM LOADS there).

The g-row / coarse-row loads of k_band4 and the ring / neighbour loads of k_band4f are issued from inline assembly and waited for with explicit
`s_waitcnt vmcnt(N)`; the compiler does not know that their destination registers are "in flight" in between.
This script reads the generated assembly and checks, for every loop of every k_band4 instantiation, that no
instruction reads or writes a destination register of a stream load between the load and the wait that covers
it (walking each loop body twice in layout order, so that the back edge is covered, and then the whole kernel once in
layout order for the straight-line code between the loops; loads return in order, so `vmcnt(N)` retires all but the N
youngest; spill reloads and stores are vector-memory operations too and sit in the same queue).  `make` runs it on the assembly
of the very build it links (csrc/Makefile: band4.isa.ok).  The walk is in LAYOUT order: it relies on the compiler keeping a loop's
blocks together, which it does for these kernels as written; a control-flow-graph version (tools/experimental/) was tried and
drowns in paths the structuriser creates but no execution takes (profiles/r03_dev_notes.txt 12).

    tools/isa_band4.sh && python tools/check_band4_isa.py [/tmp/isa/band4_new.s]
"""
import re
import sys


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


def _walk(name, body, passes, what):
    """Walk `body` in layout order `passes` times with the queue of vector-memory operations in flight (oldest first);
    report every instruction that reads or writes the destination of a load that no wait has covered yet."""
    bad = n_loads = 0
    queue = []                 # (set of regs, text)
    for it in range(passes):
        for l in body:
            code = l.split(";")[0].strip()
            if not code or code.endswith(":") or code.startswith("."):
                continue
            op = code.split()[0]
            m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", code)
            if m:
                n = int(m.group(1))
                while len(queue) > n:
                    queue.pop(0)
                continue
            touched = set(regs(code.split(None, 1)[1])) if len(code.split(None, 1)) > 1 else set()
            for rs, txt in queue:
                if touched & rs:
                    print(f"{name} [{what}]: '{code}' touches v{sorted(touched & rs)} while '{txt}' is in flight")
                    bad += 1
            if op.startswith("global_load"):
                dst = set(regs(code.split(None, 1)[1].split(",")[0]))
                queue.append((dst, code))
                n_loads += it == 0
            elif op.startswith(("scratch_load", "buffer_load", "flat_load")):
                # a spill reload (or any other vector-memory load) is one more operation in the vmcnt queue: younger than the
                # hand-issued loads before it, so the exact counts of the hand-written waits no longer cover those
                queue.append((set(regs(code.split(None, 1)[1].split(",")[0])), code))
            elif op.startswith(("global_store", "buffer_store", "scratch_store", "flat_store")):
                queue.append((set(), code))      # stores count in vmcnt on gfx9
    return bad, n_loads


def check_kernel(name, lines):
    # loops = [header label line, last line that branches back to it]
    labels = {l.split(":")[0]: i for i, l in enumerate(lines) if re.match(r"^\.LBB\d+_\d+:", l)}
    loops = []
    for i, l in enumerate(lines):
        m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
        if m and m.group(1) in labels and labels[m.group(1)] < i:
            loops.append((labels[m.group(1)], i))
    bad = 0
    n_loads = 0
    for lo, hi in loops:
        body = lines[lo:hi + 1]
        # streaming loops = loops that contain a hand-issued load (inline asm shows up between ;;#ASMSTART / ;;#ASMEND)
        if not any("global_load_dword" in l and "ASMSTART" in body[i - 1] for i, l in enumerate(body) if i > 0):
            continue
        b, n = _walk(name, body, 2, f"loop at line {lo}")      # twice around: covers the back edge
        bad += b
        n_loads += n
    # the straight-line code around the loops (prologue -> first loop, loop exit -> drain, epilogue): the whole kernel once in
    # layout order.  Every loop is followed by a drain (vmcnt(0)) and the taken path of a forward branch is a subsequence of
    # the layout order, so a register touched on any path between a load and its covering wait is seen here (conservatively:
    # an untaken arm's accesses count too).
    b, _ = _walk(name, lines, 1, "layout order")
    bad += b
    return bad, n_loads, len(loops)


def main(paths):
    total_bad = 0
    for path in paths:
        total_bad += check_file(path)
    return 1 if total_bad else 0


def check_file(path):
    text = open(path).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN5cvvdp\d+k_band4[fs]?(_edge)?(_heat|_feat)?[IE].*:\s*(;.*)?$", l)]
    total_bad = 0
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith("\t.section"))
        if "k_band4f_heat" in text[s] or "k_band4f_feat" in text[s] or re.match(r"^_ZN5cvvdp8k_band4fILi\dELi1EE", text[s]):
            # the HEAT / FEAT instantiations of k_band4f and (round 6) its plain EDGE == 1 instantiation, which only the one-wave A/B layout
            # runs, use ordinary, compiler-tracked loads (band4f.hip, F_SAFE): nothing hand-issued may
            # be left in them; what the compiler schedules around its own loads (spill reloads included) is its business
            body = text[s:e]
            hand = sum(1 for i, l in enumerate(body) if i > 0 and "global_load_dword" in l and "ASMSTART" in body[i - 1])
            assert hand == 0, f"{text[s].split(':')[0]}: {hand} hand-issued loads in a compiler-managed instantiation"
            print(f"{text[s].split(':')[0]}: compiler-managed loads, 0 hand-issued")
            continue
        bad, n_loads, n_loops = check_kernel(text[s].split(":")[0], text[s:e])
        print(f"{text[s].split(':')[0]}: {n_loops} loops, {n_loads} loads in streaming loops, {bad} violations")
        assert n_loads > 0, f"{text[s].split(':')[0]}: no hand-issued loads found (checker out of date?)"
        total_bad += bad
    assert starts, f"{path}: no k_band4 / k_band4f kernels found"
    return total_bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] if len(sys.argv) > 1 else ["/tmp/isa/band4_new.s"]))
