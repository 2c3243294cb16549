"""Instruction mix of a kernel's loops, from the gfx950 ISA hipcc emits (no GPU needed).

    python tools/isa_loop.py attention 'flash_dq2_kernel<64, 3>' [extra hipcc flags ...]

Compiles comat_amd/csrc/<file>.hip to assembly with the Makefile's flags (plus the per-file FLAGS_<file> and whatever is given
on the command line), finds the kernel whose demangled name contains the pattern, and prints, for the three innermost loops with the most MFMAs,
the instruction count per class (MFMA, VALU, packed VALU, transcendental, AGPR moves, LDS, global, waits) and the VALU
opcodes by frequency.  A loop's count covers every line between its head label and its backward branch, i.e. including
blocks the hardware skips (masked tail tiles) - read it as an upper bound per iteration.

Round 3: flash_dq2_kernel<64, 3> (head dim 40), per iteration of 64 keys: 20 MFMA (640 cycles of the matrix pipe) against
400 VALU instructions (~2000 cycles: 4 per instruction, 16 per transcendental) of which 128 were v_accvgpr_read / _write
- the kernel is VALU-bound, and `-mllvm -amdgpu-mfma-vgpr-form=1` (MFMA results in VGPRs) removes those 128."""
import collections
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "comat_amd", "csrc")


def classify(op):
    if op.startswith("v_mfma"):
        return "mfma"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "buffer_", "flat_")):
        return "vmem"
    if op.startswith(("s_waitcnt", "s_barrier", "s_nop")):
        return op
    if op.startswith("s_"):
        return "salu"
    if op.startswith(("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")):
        return "trans"
    if op.startswith("v_accvgpr"):
        return "agpr_mov"
    if op.startswith("v_pk_"):
        return "valu_pk"
    if op.startswith("v_"):
        return "valu"
    return "other"


def makefile_flags(stem):
    mk = open(os.path.join(CSRC, "Makefile")).read()
    m = re.search(r"^FLAGS_%s\s*:=\s*(.*)$" % re.escape(stem), mk, re.M)
    return m.group(1).split() if m else []


def main():
    stem, pattern, extra = sys.argv[1], sys.argv[2], sys.argv[3:]
    with tempfile.TemporaryDirectory() as td:
        asm = os.path.join(td, "k.s")
        cmd = ["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "--cuda-device-only", "-S",
               os.path.join(CSRC, stem + ".hip"), "-o", asm] + makefile_flags(stem) + extra
        subprocess.run(cmd, check=True, stderr=subprocess.DEVNULL)
        src = open(asm).read().split("\n")
    heads = [(i, l[:-1].split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"] + [n for _, n in heads], capture_output=True, text=True).stdout.split("\n")
    hit = [(i, n, d) for (i, n), d in zip(heads, names) if pattern in d.replace("(anonymous namespace)::", "")]
    if not hit:
        raise SystemExit(f"no kernel matches {pattern!r}")
    start, mangled, dem = hit[0]
    end = next(i for i in range(start, len(src)) if src[i].strip().startswith("s_endpgm"))
    body = src[start:end]
    regs = {k: re.search(r"\.set %s\.%s, (\d+)" % (re.escape(mangled), k), "\n".join(src)) for k in ("num_vgpr", "num_agpr")}
    print(f"# {dem.replace('(anonymous namespace)::', '')}: " + ", ".join(f"{k} {m.group(1)}" for k, m in regs.items() if m)
          + (f"   flags: {' '.join(makefile_flags(stem) + extra)}" if makefile_flags(stem) + extra else ""))
    labels = {m.group(1): i for i, l in enumerate(body) if (m := re.match(r"^(\.LBB\d+_\d+):", l))}
    loops = []
    for i, l in enumerate(body):
        m = re.search(r"s_c?branch\w* (\.LBB\d+_\d+)", l)
        if m and labels.get(m.group(1), len(body)) < i:
            loops.append((labels[m.group(1)], i))
    n_mfma = lambda ab: sum(1 for l in body[ab[0]:ab[1] + 1] if l.strip().startswith("v_mfma"))
    inner = [ab for ab in loops if not any(o != ab and ab[0] <= o[0] and o[1] <= ab[1] and n_mfma(o) for o in loops)]
    for a, b in sorted(inner, key=lambda ab: (-n_mfma(ab), ab[0] - ab[1]))[:3]:  # innermost loops, most MFMAs first
        c, ops = collections.Counter(), collections.Counter()
        for l in body[a:b + 1]:
            l = l.strip()
            if not l or l[0] in ";." or l.endswith(":"):
                continue
            op = l.split()[0]
            k = classify(op)
            c[k] += 1
            if k in ("valu", "valu_pk", "trans", "agpr_mov"):
                ops[op] += 1
        valu = c["valu"] + c["valu_pk"] + c["agpr_mov"]
        print(f"loop of {b - a} lines: " + ", ".join(f"{k} {v}" for k, v in c.most_common()))
        print(f"   ~cycles per wave and iteration: matrix pipe {32 * c['mfma']} (32x32x16 bf16 = 32 each), vector pipe "
              f"{4 * valu + 16 * c['trans']} (4 per instruction, 16 per transcendental)")
        print("   VALU opcodes: " + ", ".join(f"{k} {v}" for k, v in ops.most_common(16)))


if __name__ == "__main__":
    main()
