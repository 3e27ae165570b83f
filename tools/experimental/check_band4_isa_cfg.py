#!/usr/bin/env python3
"""Safety check of the hand-managed stream loads in csrc/band4.hip and csrc/band4f.hip (see STREAM LOADS there).

The row loads of k_band4 / k_band4f are issued from inline assembly and waited for with explicit `s_waitcnt vmcnt(N)`;
the compiler does not know that their destination registers are "in flight" in between.  This script reads the generated
assembly and proves, for every kernel, that no instruction reads or writes a destination register of a load between the
load and the wait that covers it -- on EVERY path through the kernel, whatever the block layout the compiler chose:

  * the kernel is cut into basic blocks (labels, branches, s_endpgm) and their successor edges;
  * the state at a program point is the set of vector-memory loads that MAY still be in flight there, each with the
    smallest number of vector-memory operations issued after it on any path (loads and stores share the vmcnt counter on
    gfx9 and return in order, so `s_waitcnt vmcnt(N)` retires exactly the operations with at least N younger ones);
  * a forward data-flow iteration to the fixed point (merge = union, age = minimum) gives that state at every
    instruction; an instruction that names a register of a load in the state is a violation.

`make` runs it on the assembly of the very build it links (csrc/Makefile: band4.isa.ok) and deletes the library when it
finds anything.

    python tools/check_band4_isa.py colorvideovdp_amd/csrc/build/band4.s colorvideovdp_amd/csrc/build/band4f.s
"""
import re
import sys

AGE_CAP = 64                      # vmcnt is a 6-bit counter


def regs(tok):
    out = []
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        if m.group(1):
            out += list(range(int(m.group(1)), int(m.group(2)) + 1))
        else:
            out.append(int(m.group(3)))
    return out


class Ins:
    __slots__ = ("code", "op", "hand", "touched", "dst", "vm", "wait", "target", "ends", "falls")

    def __init__(self, code, hand):
        self.code, self.hand = code, hand
        parts = code.split(None, 1)
        self.op = parts[0]
        rest = parts[1] if len(parts) > 1 else ""
        self.touched = frozenset(regs(rest))
        is_vmem = self.op.startswith(("global_", "buffer_", "scratch_", "flat_"))
        self.vm = is_vmem                                                # counts in vmcnt (loads, stores, atomics)
        self.dst = frozenset(regs(rest.split(",")[0])) if is_vmem and "_load" in self.op else frozenset()
        m = re.match(r"s_waitcnt.*vmcnt\((\d+)\)", code)
        self.wait = int(m.group(1)) if m else None
        m = re.match(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", code)
        self.target = m.group(1) if m else None
        self.ends = self.op in ("s_endpgm",) or self.op.startswith(("s_branch", "s_cbranch", "s_setpc"))
        self.falls = not (self.op in ("s_endpgm", "s_branch") or self.op.startswith("s_setpc"))


def parse(lines):
    ins, label_at, in_asm = [], {}, False
    for l in lines:
        s = l.strip()
        if s.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if s.startswith(";;#ASMEND"):
            in_asm = False
            continue
        code = l.split(";")[0].strip()
        if not code:
            continue
        if re.match(r"^\.LBB\d+_\d+:$", code):
            label_at[code[:-1]] = len(ins)
            continue
        if code.endswith(":") or code.startswith("."):
            continue
        ins.append(Ins(code, in_asm))
    return ins, label_at


def blocks_of(ins, label_at):
    leaders = {0} | set(label_at.values()) | {i + 1 for i, x in enumerate(ins) if x.ends}
    leaders = sorted(x for x in leaders if x < len(ins))
    start_to_block = {s: b for b, s in enumerate(leaders)}
    blocks = []
    for b, s in enumerate(leaders):
        e = leaders[b + 1] if b + 1 < len(leaders) else len(ins)
        last = ins[e - 1]
        succ = []
        if last.target is not None and last.target in label_at and label_at[last.target] < len(ins):
            succ.append(start_to_block[label_at[last.target]])
        if last.falls and e < len(ins):
            succ.append(start_to_block[e])
        blocks.append((s, e, succ))
    return blocks


def step(state, i, x, report=None):
    """state: {index of a load in flight: fewest vector-memory operations issued after it}; advanced over instruction x in place."""
    if x.wait is not None:
        for k in [k for k, age in state.items() if age >= x.wait]:
            del state[k]
        return
    if report is not None and x.touched:
        for k in state:
            if x.touched & report[0][k].dst:
                report[1].add((i, k))
    if x.vm:
        for k in state:
            if state[k] < AGE_CAP:
                state[k] += 1
        if x.dst:
            state[i] = 0


def in_cycles(blocks):
    """Blocks that lie on a cycle of the control-flow graph (Tarjan's strongly connected components, iteratively)."""
    n = len(blocks)
    index, low, on, stack, out, counter = [None] * n, [0] * n, [False] * n, [], set(), [0]
    for root in range(n):
        if index[root] is not None:
            continue
        work = [(root, 0)]
        while work:
            v, pi = work.pop()
            if pi == 0:
                index[v] = low[v] = counter[0]
                counter[0] += 1
                stack.append(v)
                on[v] = True
            succ = blocks[v][2]
            if pi < len(succ):
                work.append((v, pi + 1))
                w = succ[pi]
                if index[w] is None:
                    work.append((w, 0))
                elif on[w]:
                    low[v] = min(low[v], index[w])
                continue
            if low[v] == index[v]:
                comp = []
                while True:
                    w = stack.pop()
                    on[w] = False
                    comp.append(w)
                    if w == v:
                        break
                if len(comp) > 1 or v in blocks[v][2]:
                    out.update(comp)
            if work:
                u = work[-1][0]
                low[u] = min(low[u], low[v])
    return out


def check_kernel(name, lines):
    """-> (violations, hand-issued loads that lie on a cycle of the control-flow graph, back edges in layout order)"""
    ins, label_at = parse(lines)
    blocks = blocks_of(ins, label_at)
    preds = [[] for _ in blocks]
    for b, (_, _, succ) in enumerate(blocks):
        for s in succ:
            preds[s].append(b)
    state_in = [None] * len(blocks)                  # None = not reached yet
    state_out = [None] * len(blocks)
    state_in[0] = {}
    work = [0]
    while work:
        b = work.pop()
        st = dict(state_in[b])
        s, e, succ = blocks[b]
        for i in range(s, e):
            step(st, i, ins[i])
        if state_out[b] == st:
            continue
        state_out[b] = st
        for t in succ:
            cur = state_in[t]
            new = dict(cur) if cur is not None else {}
            for k, age in st.items():
                new[k] = min(new.get(k, AGE_CAP), age)
            if new != cur:
                state_in[t] = new
                work.append(t)
    found = set()
    for b, (s, e, _) in enumerate(blocks):
        if state_in[b] is None:
            continue
        st = dict(state_in[b])
        for i in range(s, e):
            step(st, i, ins[i], (ins, found))
    for i, k in sorted(found):
        print(f"{name}: '{ins[i].code}' touches v{sorted(ins[i].touched & ins[k].dst)} while '{ins[k].code}' may be in flight")
    cyc = in_cycles(blocks)
    n_hand = sum(1 for b in cyc for i in range(blocks[b][0], blocks[b][1]) if ins[i].hand and ins[i].dst)
    n_back = sum(1 for b, (s, e, succ) in enumerate(blocks) for t in succ if t <= b)
    return len(found), n_hand, n_back


def main(paths):
    total_bad = 0
    for path in paths:
        total_bad += check_file(path)
    return 1 if total_bad else 0


def check_file(path):
    text = open(path).read().split("\n")
    starts = [i for i, l in enumerate(text) if re.match(r"^_ZN5cvvdp\d+k_band4f?I.*:\s*(;.*)?$", l)]
    total_bad = 0
    for s in starts:
        e = next(i for i in range(s, len(text)) if ".end_amdhsa_kernel" in text[i] or text[i].startswith("\t.section"))
        bad, n_loads, n_loops = check_kernel(text[s].split(":")[0], text[s:e])
        print(f"{text[s].split(':')[0]}: {n_loops} back edges, {n_loads} hand-issued loads in loops, {bad} violations")
        assert n_loads > 0, f"{text[s].split(':')[0]}: no hand-issued loads found (checker out of date?)"
        total_bad += bad
    assert starts, f"{path}: no k_band4 / k_band4f kernels found"
    return total_bad


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:] if len(sys.argv) > 1 else ["/tmp/isa/band4_new.s"]))
