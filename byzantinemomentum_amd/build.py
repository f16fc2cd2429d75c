"""In-tree build of libbm_gar.so (HIP kernels + C ABI) for gfx950.

`hipcc --offload-arch=gfx950` cross-compiles without a GPU, so this runs in the build
container; the resulting .so is git-ignored but travels to the GPU box with the snapshot.
No JIT, no cache outside the tree.
"""

import concurrent.futures
import os
import pathlib
import shutil
import subprocess
import sys

PKG_DIR = pathlib.Path(__file__).resolve().parent
SRC_DIR = PKG_DIR / "csrc"
OBJ_DIR = PKG_DIR / "_build"
LIB_PATH = PKG_DIR / "libbm_gar.so"
INCLUDE_DIR = PKG_DIR.parent / "include"

ARCH = "gfx950"
# NO packed-fp32 VALU instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in this library.  The SLP vectoriser
# pairs independent fp32 chains into them — in Bulyan's second pass the suffix sums of neighbouring ranks: 497 packed
# instructions, 398 of them with op_sel / op_sel_hi — and that kernel returned wrong coordinates in ~1.5 % of its
# launches whenever several processes time-shared the GPU (always lanes 48-63 of one register of a wave, on inputs
# nobody had written; several experiments per GPU is the reference's own deployment mode, reproduce.py:62-73,118).
# The same source without the packed forms: 0 of 3 200 launch sets against 44 of 3 200 (profiles/r06_pass2_variants.txt,
# DESIGN 8).  -fno-slp-vectorize stops the pairing; turning the `packed-fp32-ops` target feature off makes the code
# generator unable to select the instructions at all, whatever the source or a later compiler does (the flag reaches
# the host compilation too, which says it does not know the feature: filtered from the build's messages).
# tests/test_kernel_meta.py disassembles the library and fails if one comes back.
NO_PACKED_FP32 = ["-fno-slp-vectorize", "-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
CXXFLAGS = ["-O3", "-std=c++17", "-fPIC", f"--offload-arch={ARCH}", "-Wall", "-Wno-unused-function",
            "-fno-gpu-rdc", "-ffp-contract=off"]
if os.environ.get("BM_BUILD_PACKED_FP32", "0") in ("", "0"):  # (1: the A side of an A/B build, never shipped)
  CXXFLAGS += NO_PACKED_FP32


def _hipcc():
  exe = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not os.path.exists(exe):
    raise RuntimeError("hipcc not found: libbm_gar.so cannot be built (ROCm toolchain required)")
  return exe


def sources():
  return sorted(list(SRC_DIR.glob("*.hip")) + list(SRC_DIR.glob("*.cpp")))


def _deps_mtime():
  headers = list(SRC_DIR.glob("*.h")) + list(INCLUDE_DIR.glob("*.h"))
  return max(p.stat().st_mtime for p in headers)


def _compile(src, obj):
  cmd = [_hipcc(), *CXXFLAGS, "-c", str(src), "-o", str(obj)]
  if src.suffix == ".cpp":
    cmd.insert(1, "-x")
    cmd.insert(2, "hip")
  proc = subprocess.run(cmd, capture_output=True, text=True)
  if proc.returncode != 0:
    raise RuntimeError(f"hipcc failed on {src.name}:\n{proc.stderr}")
  return "\n".join(line for line in proc.stderr.splitlines() if "packed-fp32-ops" not in line)


def build(force=False, verbose=False):
  """Compile every csrc/*.hip|*.cpp that is out of date and link libbm_gar.so. Returns its path."""
  OBJ_DIR.mkdir(exist_ok=True)
  hdr_time = _deps_mtime()
  todo, objs = [], []
  for src in sources():
    obj = OBJ_DIR / (src.name + ".o")
    objs.append(obj)
    if force or not obj.exists() or obj.stat().st_mtime < max(src.stat().st_mtime, hdr_time):
      todo.append((src, obj))
  if todo:
    with concurrent.futures.ThreadPoolExecutor(max_workers=min(len(todo), os.cpu_count() or 4)) as pool:
      for warn in pool.map(lambda so: _compile(*so), todo):
        if verbose and warn:
          sys.stderr.write(warn)
  if todo or not LIB_PATH.exists() or any(o.stat().st_mtime > LIB_PATH.stat().st_mtime for o in objs):
    cmd = [_hipcc(), f"--offload-arch={ARCH}", "-shared", "-fPIC", "-o", str(LIB_PATH), *map(str, objs)]
    proc = subprocess.run(cmd, capture_output=True, text=True)
    if proc.returncode != 0:
      raise RuntimeError(f"link of libbm_gar.so failed:\n{proc.stderr}")
  return LIB_PATH


if __name__ == "__main__":
  path = build(force="--force" in sys.argv, verbose=True)
  print(path)
