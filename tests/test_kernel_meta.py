"""Code-object metadata of the kernels behind the bench line (no GPU needed: hipcc cross-compiles, the notes of the
gfx950 code objects inside libbm_gar.so say what the register allocator did).  A scratch spill inside one of these
kernels is a regression the GPU tests would not notice (same bits, slower)."""

import importlib.util
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

# (regex on the demangled name, VGPR ceiling): the ceiling is what the launch bounds of the kernel aim at
# (168 = three workgroups of 256 lanes per CU, 128 = one workgroup of 1024 lanes per CU ... DESIGN 4)
HOT = [
  (r"colwise_burst_kernel<25, 0, 4>", 128),          # C2 median
  (r"colwise_burst_kernel<25, 1, 4>", 128),          # C2 trimmed mean
  (r"gram3_partial_kernel<7, 2, true, (true|false)>", 168),        # C4 distance pass (n = 25, two planes)
  (r"gram3_partial_kernel<13, 2, true, (true|false)>", 256),       # C3 distance pass (n = 51)
  (r"bulyan_pass2_kernel<25, 5, 4>", 128),           # C4 pass 2
  (r"selected_mean_burst_kernel", 128),              # C3 average of the selected rows
  # C5 first pass with the distance pass riding along: built without packed fp32 (build.py) it parks 4 scalar registers
  # in the lanes of a vector register (v_writelane / v_readlane outside the streaming loop; measured: no change of the
  # step's time, profiles/r06_fixcheck_bench_ab.txt) — a handful is tolerated, a scratch spill is not
  (r"momentum_gram_kernel<20, false, false>", 256, 8),
  (r"study_stats_burst_kernel<true, 3, false, false>", 128),  # C5 study block
  (r"study_stats_burst_kernel<true, 3, false, true>", 128),   # ... carrying the momentum of the update (--momentum-at update)
]


@pytest.fixture(scope="module")
def kernels():
  spec = importlib.util.spec_from_file_location("kernel_meta", ROOT / "scripts" / "kernel_meta.py")
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  if not mod.LIB.exists() or not (mod.LLVM / "llvm-objdump").exists():
    pytest.skip("libbm_gar.so or the LLVM tools are not here")
  return mod.kernels()


@pytest.mark.parametrize("pattern,ceiling,sgpr_spills", [(h + (0,))[:3] for h in HOT])
def test_hot_kernels_keep_their_registers(kernels, pattern, ceiling, sgpr_spills):
  hits = [k for k in kernels if re.search(pattern, k["demangled"])]
  assert hits, f"no kernel matches {pattern}"
  for k in hits:
    assert k["vgpr_spill"] == 0 and k["scratch"] == 0, (k["demangled"], k["vgpr_spill"], k["scratch"])
    assert k["sgpr_spill"] <= sgpr_spills, (k["demangled"], k["sgpr_spill"])
    assert k["vgpr"] <= ceiling, (k["demangled"], k["vgpr"], ceiling)


def test_no_packed_fp32_instruction_in_the_library():
  """v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32 (what the SLP vectoriser makes of paired fp32 chains) gave wrong results
  in Bulyan's second pass when several processes shared the GPU (profiles/r06_pass2_variants.txt): the library
  is built without them (-fno-slp-vectorize) and no kernel may bring them back."""
  spec = importlib.util.spec_from_file_location("kernel_meta", ROOT / "scripts" / "kernel_meta.py")
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  if not mod.LIB.exists() or not (mod.LLVM / "llvm-objdump").exists():
    pytest.skip("libbm_gar.so or the LLVM tools are not here")
  found = mod.packed_fp32()
  assert not found, sorted(found.items(), key=lambda kv: -kv[1])[:10]
