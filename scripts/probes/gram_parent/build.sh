#!/bin/bash
# libbm_gar.so with the Gram kernel as it was BEFORE commit 369a08a (per-lane wave index, clamped row index, pointer
# table padded with row 0) and everything else as in the tree: the B side of profiles/r05_f_gram_scalar_wave_ab.txt.
#   scripts/probes/gram_parent/build.sh            -> scratch/gram_parent/libbm_gar_gram_parent.so  (scratch/ is git-ignored)
#   for i in 1 2 3; do python scripts/gram_variant_probe.py; \
#     BM_GAR_LIB=scratch/gram_parent/libbm_gar_gram_parent.so python scripts/gram_variant_probe.py; done     (ONE gpurun call)
set -e
root=$(cd $(dirname $0)/../../.. && pwd)
work=$root/scratch/gram_parent
mkdir -p $work/obj $root/scratch/include
cp -r $root/byzantinemomentum_amd/csrc $work/
cp $root/include/bm_gar.h $root/scratch/include/   # (csrc includes "../../include/bm_gar.h")
python3 - "$work/csrc/gram_bf16.hip" <<'PY'
import sys
p = sys.argv[1]
s = open(p).read()
# the three edits of commit 369a08a, undone
s = s.replace("const int wave = __builtin_amdgcn_readfirstlane(tid >> 6), lane = tid & 63;", "const int wave = tid >> 6, lane = tid & 63;")
s = s.replace("row_ptr[tid] = (const float*)karg[tid < n ? tid : n - 1];", "row_ptr[tid] = (const float*)karg[tid < n ? tid : 0];")
old = "        const int r = 4 * k + rho;  // (rows >= n: the table repeats row n - 1)"
assert s.count(old) == 3
s = s.replace(old, "        int r = 4 * k + rho;\n        if (k == K - 1) r = r < n ? r : n - 1;")
open(p, "w").write(s)
PY
for f in $work/csrc/*.hip $work/csrc/*.cpp; do
  x=""; case $f in *.cpp) x="-x hip";; esac
  /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -ffp-contract=off $x -c $f -o $work/obj/$(basename $f).o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $work/libbm_gar_gram_parent.so $work/obj/*.o
ls -la $work/libbm_gar_gram_parent.so
