"""Compare the device instruction streams of the kernels in comat_amd/csrc/*.hip between a git revision and the
working tree (hipcc -S, device only; labels, comments and symbol names normalised).  Used to show that adding
experimental variants left the GPU-validated default kernels bit-for-bit unchanged:

    python tools/check_default_isa.py <validated-rev> [file.hip ...]
"""
import os
import re
import subprocess
import sys
import tempfile

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HIPCC = "/opt/rocm/bin/hipcc"


def kernels(asm):
    out = {}
    for m in re.finditer(r"^(_Z[^\s:]+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", asm, re.S | re.M):
        name, body = m.group(1), m.group(2)
        body = "\n".join(line.split(";")[0].rstrip() for line in body.split("\n"))
        body = re.sub(r"\.L[A-Za-z0-9_]+", "L", body).replace(name, "SELF")
        out[name] = body
    return out


def compile_asm(src_text, workdir, name):
    csrc = os.path.join(workdir, "comat_amd", "csrc")
    os.makedirs(csrc, exist_ok=True)
    os.makedirs(os.path.join(workdir, "include"), exist_ok=True)
    path = os.path.join(csrc, name)
    open(path, "w").write(src_text)
    out = path + ".s"
    subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-o", out, path])
    return kernels(open(out).read())


def git_show(rev, rel):
    return subprocess.check_output(["git", "-C", ROOT, "show", f"{rev}:{rel}"]).decode()


def main():
    rev = sys.argv[1]
    files = sys.argv[2:] or sorted(f for f in os.listdir(os.path.join(ROOT, "comat_amd", "csrc"))
                                   if f.endswith(".hip") and f != "gemm_exp.hip")
    bad = 0
    for f in files:
        rel = f"comat_amd/csrc/{f}"
        with tempfile.TemporaryDirectory() as old_d, tempfile.TemporaryDirectory() as new_d:
            for d, getter in ((old_d, lambda r: git_show(rev, r)), (new_d, lambda r: open(os.path.join(ROOT, r)).read())):
                for dep in ("comat_amd/csrc/common.h", "include/comat_hip.h"):
                    p = os.path.join(d, dep)
                    os.makedirs(os.path.dirname(p), exist_ok=True)
                    open(p, "w").write(getter(dep))
            old = compile_asm(git_show(rev, rel), old_d, f)
            new = compile_asm(open(os.path.join(ROOT, rel)).read(), new_d, f)
        # a kernel of the old build may have gained trailing default template arguments: match by name prefix
        same = diff = gone = 0
        for name, body in old.items():
            stem = re.sub(r"E+v.*$", "", name)
            cands = [n for n in new if n.startswith(stem)]
            if not cands:
                gone += 1  # no longer instantiated in this translation unit (e.g. moved to gemm_exp.hip)
            elif any(new[n] == body for n in cands):
                same += 1
            else:
                diff += 1
                print(f"  DIFFERENT: {name[:100]}")
        bad += diff
        print(f"{f}: {len(old)} kernels at {rev}, {len(new)} now; identical {same}, different {diff}, "
              f"not in this unit any more {gone}")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
