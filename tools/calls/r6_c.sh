#!/bin/bash
# round 6, call 3: cache-policy bits of the LDS-DMA loads (COMAT_G2_AUX builds), same box, one library per process
cd "$GRAFT_REPO_ROOT" || exit 1
export TMPDIR=/tmp HIP_FORCE_DEV_KERNARG=1
O=gpurun_out
mkdir -p $O
for lib in hip hip_aux2 hip_aux16 hip_aux17 hip; do
  echo "== lib $lib"
  MB_BASE_ONLY=1 COMAT_LIB_PATH=$PWD/comat_amd/lib/libcomat_$lib.so timeout 600 python tools/mb_pp.py > $O/r6c_mb_$lib.txt 2>&1
  grep -c . $O/r6c_mb_$lib.txt
done
python - <<'PY'
import re,glob
rows={}
libs=['hip','hip_aux2','hip_aux16','hip_aux17']
for l in libs:
    for line in open(f'gpurun_out/r6c_mb_{l}.txt'):
        m=re.match(r'(\S.*?)\s{2,}(\S.*?)\s+([\d.]+) us',line)
        if m and 'BEST' not in line:
            rows.setdefault((m.group(1),m.group(2)),{})[l]=float(m.group(3))
print('%-40s %-12s'%('problem','variant')+''.join('%10s'%l for l in libs))
for k,v in rows.items():
    print('%-40s %-12s'%k+''.join('%10.1f'%v.get(l,0) for l in libs))
PY
echo done
