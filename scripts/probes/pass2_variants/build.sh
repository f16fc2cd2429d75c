#!/bin/bash
# Variants of libbm_gar.so that differ ONLY in how bulyan.hip is compiled — which property of bulyan_pass2_kernel<25,5,4>
# makes it lose a quarter of a vector register under GPU sharing (scripts/stale_read_hunt.py --lib ...)?
#   packed    bulyan.hip as the compiler builds it by default (packed fp32 allowed): the failing baseline of rounds 4-5
#   noslp     -fno-slp-vectorize: no v_pk_add_f32 / v_pk_fma_f32 / v_pk_mul_f32 (the compiler pairs suffix sums)
#   fminmax   comparators through __builtin_fminf / fmaxf instead of inline asm (visible to the hazard recogniser)
#   plainload row loads without the non-temporal hint
#   vec2      8-byte columns per lane (other register allocation: 70 VGPRs / 95 SGPRs)
#   ldsptr    ranked pointers through LDS + readfirstlane instead of run-time-indexed scalar loads from the kernarg segment
#   vgprdiv   the division constants (count, 1 / count) made opaque in VECTOR registers: packed ops may stay, but none takes a scalar register PAIR as operand
#   scalaradd the suffix sums through inline-asm v_add_f32 (the vectoriser cannot pair them): the packed additions go, the packed divisions with scalar pairs stay
# Output: scratch/pass2_variants/libbm_gar_<name>.so (scratch/ is git-ignored, travels with gpurun).
set -e
root=$(cd $(dirname $0)/../../.. && pwd)
work=$root/scratch/pass2_variants
mkdir -p $work/include
cp $root/include/bm_gar.h $work/include/   # (csrc includes "../../include/bm_gar.h")
python3 $root/byzantinemomentum_amd/build.py > /dev/null
FLAGS="-O3 -std=c++17 -fPIC --offload-arch=gfx950 -Wno-unused-function -fno-gpu-rdc -ffp-contract=off"
others=$(ls $root/byzantinemomentum_amd/_build/*.o | grep -v bulyan.hip.o)
for name in ${@:-noslp fminmax plainload vec2 ldsptr vgprdiv scalaradd}; do
  (
  mkdir -p $work/$name/csrc; cp $root/byzantinemomentum_amd/csrc/*.h $root/byzantinemomentum_amd/csrc/bulyan.hip $work/$name/csrc/
  extra=""
  case $name in
    packed) ;;
    noslp) extra="-fno-slp-vectorize";;
    fminmax) python3 - $work/$name/csrc/bm_common.h <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = s[s.index('  float lo, hi;\n  asm("v_min_f32'):s.index('  a = lo;\n  b = hi;')]
s = s.replace(old, '  const float lo = __builtin_fminf(a, b), hi = __builtin_fmaxf(a, b);\n')
open(p, 'w').write(s)
PY
    ;;
    plainload) sed -i 's/const T v = __builtin_nontemporal_load(reinterpret_cast<const T\*>(p));/const T v = *reinterpret_cast<const T*>(p);/' $work/$name/csrc/bm_common.h;;
    vec2) sed -i 's/constexpr int kMaxVec = (MMAX <= 20) ? 4 : (MMAX <= 44 ? 2 : 1);/constexpr int kMaxVec = (MMAX <= 44 ? 2 : 1);/' $work/$name/csrc/bulyan.hip;;
    ldsptr) sed -i 's/if constexpr (MMAX <= 25) {/if constexpr (MMAX <= 0) {/' $work/$name/csrc/bulyan.hip;;
    vgprdiv) python3 - $work/$name/csrc/bulyan.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "        const float cnt = (float)(MMAX - i);\n        sel[i] = div_small_int(s, cnt, 1.0f / cnt);"
assert old in s
s = s.replace(old, "        float cnt = (float)(MMAX - i), rcnt = 1.0f / (float)(MMAX - i);\n        asm(\"\" : \"+v\"(cnt), \"+v\"(rcnt));\n        sel[i] = div_small_int(s, cnt, rcnt);")
open(p, 'w').write(s)
PY
    ;;
    scalaradd) python3 - $work/$name/csrc/bulyan.hip <<'PY'
import sys
p = sys.argv[1]; s = open(p).read()
old = "        for (int t = i; t < MMAX; ++t) s += x[c][t];"
assert old in s
s = s.replace(old, "        for (int t = i; t < MMAX; ++t) asm(\"v_add_f32 %0, %1, %2\" : \"=v\"(s) : \"v\"(s), \"v\"(x[c][t]));")
open(p, 'w').write(s)
PY
    ;;
  esac
  /opt/rocm/bin/hipcc $FLAGS $extra -I$root -c $work/$name/csrc/bulyan.hip -o $work/$name/bulyan.hip.o
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o $work/libbm_gar_$name.so $others $work/$name/bulyan.hip.o
  ) &
done
wait
ls -la $work/*.so
