"""Per kernel of one or more assembly files (hipcc -S --cuda-device-only): instruction count, scratch instructions, private segment
bytes, VGPRs - the quick check for the two things that made the round-5 GEMM epilogue slow (round 6: 19 000 unrolled instructions and
arrays in scratch).    python tools/isa_sizes.py /tmp/k_*.s [/tmp/gemm2.s]"""
import re
import subprocess
import sys

for f in sys.argv[1:]:
    src = open(f).read().split("\n")
    text = "\n".join(src)
    heads = [(i, l.split(":")[0]) for i, l in enumerate(src) if re.match(r"^_Z\w+:", l)]
    names = subprocess.run(["c++filt"] + [n for _, n in heads], capture_output=True, text=True).stdout.split("\n")
    for (i, n), d in zip(heads, names):
        try:
            end = next(j for j in range(i, len(src)) if src[j].startswith(".Lfunc_end"))
        except StopIteration:
            continue
        body = [l.split(";")[0].strip() for l in src[i + 1:end]]
        body = [l for l in body if l and not l.endswith(":") and not l.startswith(".")]
        sc = sum(1 for l in body if l.startswith("scratch"))
        m = re.search(r"\.amdhsa_kernel %s\n(.*?)\.end_amdhsa_kernel" % re.escape(n), text, re.S)
        priv = re.search(r"\.amdhsa_private_segment_fixed_size (\d+)", m.group(1)).group(1) if m else "?"
        vg = re.search(r"\.set %s\.num_vgpr, (\d+)" % re.escape(n), text)
        d = d.replace("(anonymous namespace)::", "").replace("void ", "")
        print(f"{len(body):7d} instr  {sc:4d} scratch  {priv:>6s} B private  {vg.group(1) if vg else '?':>4s} vgpr  {d[:110]}")
