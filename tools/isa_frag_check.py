"""Register discipline of the hand-issued fragment reads of gemm2.hip (G2_PIN = 3: `ds_read_b128` from inline asm, which hipcc neither
counts nor waits for): between such a read and the `s_waitcnt lgkmcnt(N)` that retires it, no instruction may touch its destination
registers - on any path.  Forward dataflow over the basic blocks of every gemm2_kernel instantiation in an assembly file; the state is
the list of pending reads, oldest first (LDS reads retire in order: a wait for lgkmcnt(N) leaves the N youngest pending; waits the
compiler emits for its own operations count the same way); states meeting at a join are merged youngest-aligned (a superset).

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -mllvm -amdgpu-mfma-vgpr-form=1 -DG2_PIN=3 -S --cuda-device-only -o /tmp/gemm2.s comat_amd/csrc/gemm2.hip
    python tools/isa_frag_check.py /tmp/gemm2.s [name-pattern]
"""
import re
import sys


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return frozenset(out)


def blocks_of(body):
    """-> (blocks: list of instruction lists, succ: list of successor index lists)"""
    insts, label_at = [], {}
    for l in body:
        l = l.split(";")[0].strip()
        if not l or l.startswith("."):
            m = re.match(r"^(\.LBB\d+_\d+):", l)
            if m:
                label_at[m.group(1)] = len(insts)
            continue
        insts.append(l)
    starts = {0} | set(label_at.values())
    for k, l in enumerate(insts):
        if l.startswith(("s_branch", "s_cbranch", "s_endpgm")):
            starts.add(k + 1)
    starts = sorted(s for s in starts if s < len(insts))
    index = {s: i for i, s in enumerate(starts)}
    blocks, succ = [], []
    for i, s in enumerate(starts):
        e = starts[i + 1] if i + 1 < len(starts) else len(insts)
        blocks.append(insts[s:e])
        last = insts[e - 1]
        out = []
        m = re.search(r"(\.LBB\d+_\d+)", last)
        if last.startswith("s_branch"):
            out = [index[label_at[m.group(1)]]]
        elif last.startswith("s_cbranch"):
            out = [index[label_at[m.group(1)]]] + ([i + 1] if i + 1 < len(starts) else [])
        elif not last.startswith("s_endpgm") and i + 1 < len(starts):
            out = [i + 1]
        succ.append(out)
    return blocks, succ


def merge(a, b):
    if a is None:
        return b
    n = max(len(a), len(b))
    pa, pb = (frozenset(),) * (n - len(a)) + a, (frozenset(),) * (n - len(b)) + b
    return tuple(x | y for x, y in zip(pa, pb))


def transfer(state, block, report):
    pending = list(state)
    bad = 0
    for l in block:
        op = l.split()[0]
        if op == "s_waitcnt":
            m = re.search(r"lgkmcnt\((\d+)\)", l)
            if m:
                n = int(m.group(1))
                pending = pending[len(pending) - n:] if 0 < n < len(pending) else ([] if n == 0 else pending)
            continue
        touched = regs_of(l[len(op):])
        live = frozenset().union(*pending) if pending else frozenset()
        if op.startswith("ds_read"):
            dst = regs_of(l[len(op):].split(",")[0])
            if touched & live:
                bad += 1
                if report:
                    print("   BAD", l)
            pending.append(dst)
        elif touched & live:
            bad += 1
            if report:
                print("   BAD", l)
    return tuple(pending), bad


def check(body):
    blocks, succ = blocks_of(body)
    state_in = [None] * len(blocks)
    state_in[0] = ()
    work = [0]
    while work:
        i = work.pop()
        out, _ = transfer(state_in[i], blocks[i], False)
        for j in succ[i]:
            m = merge(state_in[j], out)
            if m != state_in[j]:
                state_in[j] = m
                work.append(j)
    bad = 0
    for i, b in enumerate(blocks):
        if state_in[i] is not None:
            bad += transfer(state_in[i], b, True)[1]
    reads = sum(1 for b in blocks for l in b if l.startswith("ds_read"))
    return reads, len(blocks), bad


def main():
    src = open(sys.argv[1]).read().split("\n")
    pat = sys.argv[2] if len(sys.argv) > 2 else "gemm2_kernel"
    heads = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    total_bad = 0
    for i, n in heads:
        if pat not in n:
            continue
        end = next(j for j in range(i, len(src)) if src[j].startswith(".Lfunc_end"))
        print(f"{n[:100]}:")
        r, nb, b = check(src[i + 1:end])
        total_bad += b
        print(f"   {r} LDS reads in {nb} blocks, {b} violations")
    sys.exit(1 if total_bad else 0)


if __name__ == "__main__":
    main()
