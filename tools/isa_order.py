"""Order of memory instructions, waits and MFMAs inside a kernel's main loop, from the gfx950 ISA hipcc emits (no GPU needed).

    python tools/isa_order.py attention [pattern ...] [--lds] [--trcheck] [-DFOO=1 ...]

Compiles comat_amd/csrc/<file>.hip to assembly with the Makefile's flags and prints, for every kernel whose demangled name
contains one of the patterns (all kernels without one), the loop with the most MFMAs as one line:
    L<n> global / buffer loads   S<n> global stores   [wN] s_waitcnt vmcnt(N)   M<n> MFMAs   d<n> LDS writes   | s_barrier
--lds prints LDS reads (r<n>), `s_waitcnt lgkmcnt(N)` ([kN]), exp2 (e<n>) and MFMAs instead.  The scan is linear over the
loop's text: blocks a branch skips at run time (edge tiles, slow paths) appear in it.

Round 4 found with it that every forward and dK/dV attention loop read `[w0]L4[w0]M14 d4|`: the next tile's loads, then a wait
for ALL of them in front of the first MFMA (the compiler's guard for prologue loads whose own waits sit in divergent branches).
After `loads_landed()`: `[w0]L4 M14[w0]d4|`.

--trcheck verifies the register discipline of the hand-issued `ds_read_b64_tr_b16` (asm the compiler does not count): no
instruction touches a destination register between the read and an `s_waitcnt lgkmcnt(0)`."""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comat_amd", "csrc")


def makefile_flags(stem):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^FLAGS_%s\s*:=\s*(.*)$" % re.escape(stem), mk, re.M)
    return m.group(1).split() if m else []


def kernels(stem, extra):
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
               os.path.join(CSRC, stem + ".hip"), "-o", asm] + [f for f in makefile_flags(stem) if "COMAT_SRC_HASH" not in f] + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        src = open(asm).read().split("\n")
    text = "\n".join(src)
    heads = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"] + [n for _, n in heads], capture_output=True, text=True).stdout.split("\n")
    for (i, n), d in zip(heads, names):
        end = next(j for j in range(i, len(src)) if src[j].strip().startswith("s_endpgm"))
        body = [l for l in src[i + 1:end] if l.strip() and not l.strip().startswith(";")]
        m = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(n), text)
        yield d.replace("(anonymous namespace)::", ""), body, (m.group(1) if m else "?")


def main_loop(body):
    labels = {m.group(1): k for k, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for k, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), 1 << 30) < k:
            loops.append((labels[m.group(1)], k))
    n_mfma = lambda ab: sum(1 for l in body[ab[0]:ab[1] + 1] if "v_mfma" in l)
    loops = [ab for ab in loops if n_mfma(ab)]
    return max(loops, key=n_mfma) if loops else None


def order(lines, lds):
    seq = []
    for l in lines:
        t = l.strip()
        op = t.split()[0]
        if op.startswith("v_mfma"):
            seq.append("M")
        elif op == "s_barrier":
            seq.append("|")
        elif lds:
            if op.startswith("ds_read"):
                seq.append("r")
            elif op == "s_waitcnt" and "lgkmcnt" in t:
                seq.append("[k%s]" % re.search(r"lgkmcnt\((\d+)\)", t).group(1))
            elif op.startswith("v_exp"):
                seq.append("e")
        else:
            if op.startswith(("global_load", "buffer_load")):
                seq.append("L")
            elif op.startswith(("global_store", "buffer_store")):
                seq.append("S")
            elif op == "s_waitcnt" and "vmcnt" in t:
                seq.append("[w%s]" % re.search(r"vmcnt\((\d+)\)", t).group(1))
            elif op.startswith("ds_write"):
                seq.append("d")
    s = "".join(seq)
    for ch in "MLSdre":
        s = re.sub(ch + "+", lambda m: "%s%d" % (ch, len(m.group(0))), s)
    return s


def regs_of(tok):
    out = set()
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]|\bv(\d+)\b", tok):
        out |= set(range(int(m.group(1)), int(m.group(2)) + 1)) if m.group(1) else {int(m.group(3))}
    return out


def trcheck(name, body):
    pending, bad, n = {}, 0, 0
    for k, l in enumerate(body):
        l = l.split(";")[0].strip()
        if not l or l.endswith(":"):
            continue
        op = l.split()[0]
        if op == "s_waitcnt" and "lgkmcnt(0)" in l:
            pending.clear()
            continue
        touched = regs_of(l[len(op):])
        if op == "ds_read_b64_tr_b16":
            n += 1
            dst = regs_of(l[len(op):].split(",")[0])
            if (touched - dst) & set(pending) or dst & set(pending):
                print("BAD", name[:70], k, l)
                bad += 1
            pending.update({r: k for r in dst})
        elif touched & set(pending):
            print("BAD", name[:70], k, l)
            bad += 1
    return n, bad


def main():
    args = sys.argv[1:]
    stem = args[0]
    extra = [a for a in args[1:] if a.startswith("-D") or a.startswith("-mllvm") or a.startswith("-amdgpu")]
    pats = [a for a in args[1:] if not a.startswith("-")]
    lds, chk = "--lds" in args, "--trcheck" in args
    n_tr = n_bad = 0
    for name, body, vgpr in kernels(stem, extra):
        if pats and not any(p in name for p in pats):
            continue
        if chk:
            a, b = trcheck(name, body)
            n_tr += a
            n_bad += b
            continue
        ab = main_loop(body)
        if ab:
            print(f"{name[:72]:72s} vgpr {vgpr:>3s}   {order(body[ab[0]:ab[1] + 1], lds)[:300]}")
    if chk:
        print(f"{n_tr} transposed reads checked, {n_bad} violations")


if __name__ == "__main__":
    main()
