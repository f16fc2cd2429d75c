"""The C-ABI library loads on a machine WITHOUT a GPU and exports every symbol include/bm_gar.h
declares (no compute is launched here)."""

import ctypes
import pathlib
import re

import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent


def declared_functions():
  text = (ROOT / "include" / "bm_gar.h").read_text()
  text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
  return sorted(set(re.findall(r"\b(bm_[a-z0-9_]+)\s*\(", text)))


@pytest.fixture(scope="module")
def lib():
  from byzantinemomentum_amd import build, _lib
  build.build()
  return _lib.load()


def test_header_declares_the_path():
  names = declared_functions()
  for must in ("bm_colwise", "bm_pairwise_sqdist", "bm_krum_rank", "bm_selected_mean", "bm_bulyan_pass2",
               "bm_aksel_pass1", "bm_stack_stats", "bm_multi_dot", "bm_multi_axpby", "bm_brute_select"):
    assert must in names


def test_every_declared_symbol_is_exported(lib):
  for name in declared_functions():
    assert hasattr(lib, name), f"{name} declared in include/bm_gar.h but not exported"


def test_ctypes_signatures_cover_the_header(lib):
  from byzantinemomentum_amd import _lib
  assert sorted(_lib.SIGNATURES) == declared_functions()
  assert lib.bm_abi_version() == _lib.ABI_VERSION


def test_error_strings(lib):
  assert lib.bm_error_string(0) == b"success"
  assert b"invalid" in lib.bm_error_string(-100000)


def test_argument_validation_without_gpu(lib):
  """Bad arguments are refused before any HIP call, so this is safe on a CPU-only box."""
  from byzantinemomentum_amd import _lib
  rows = (ctypes.c_void_p * 1)()
  assert lib.bm_colwise(0, rows, 0, 10, 0, None, None) == _lib.EINVAL      # n < 1
  assert lib.bm_colwise(0, rows, 65, 10, 0, None, None) == _lib.EINVAL     # n > BM_MAX_ROWS
  assert lib.bm_workspace_bytes(0, 0, 10) == _lib.EINVAL
  assert lib.bm_workspace_bytes(_lib.WS_PAIRWISE, 25, 1000) > 0


def test_new_entry_points_validate_arguments_without_gpu(lib):
  """Round-2 entry points: bad arguments are refused before any HIP or RCCL call."""
  from byzantinemomentum_amd import _lib
  rows = (ctypes.c_void_p * 4)()
  assert lib.bm_momentum_stats(rows, 3, rows, 4, 10, 0.9, 0.1, None, None, None, None, 1.0, 0, None, None, None) == _lib.EINVAL
  assert lib.bm_multi_fma3(rows, rows, rows, 0, 10, 1.0, 1.0, None, None) == _lib.EINVAL
  assert lib.bm_clip_factors(None, 3, 1.0, None, None) == _lib.EINVAL
  assert lib.bm_sharded_workspace_bytes(0, 10) == _lib.EINVAL
  assert lib.bm_sharded_workspace_bytes(25, 1000) > 0
  assert lib.bm_step_workspace_bytes(25, 1000) > lib.bm_sharded_workspace_bytes(25, 1000)
  assert lib.bm_step_stats_count() == 32
  par = _lib.StepParams(n=25, f_decl=5, f_real=5, ks=10, rule=0)  # ks < honest count
  assert lib.bm_step_worker(None, ctypes.byref(par), rows, rows, 10, 10, *([None] * 13)) == _lib.EINVAL
  ok = _lib.StepParams(n=25, f_decl=5, f_real=5, ks=20, rule=0)
  assert lib.bm_step_worker(None, ctypes.byref(ok), rows, rows, 10, 9, *([None] * 13)) == _lib.EINVAL  # d_total < d
  # round 4: the sharded rules take the length of the whole vectors (every rank states the same number)
  assert lib.bm_sharded_krum(None, rows, 4, 10, 9, 1, 1, rows, None, rows, None) == _lib.EINVAL      # d_total < d_local
  assert lib.bm_sharded_bulyan(None, rows, 7, 10, 9, 1, 4, rows, None, rows, None) == _lib.EINVAL
  assert lib.bm_comm_size(None) == 1
  assert lib.bm_allreduce_sum_f64(None, None, 4, None) == _lib.EINVAL
  assert lib.bm_comm_init(None, 1, 0, None) == _lib.EINVAL
  # round 3
  assert lib.bm_study_stats(None, None, None, None, 0, None, None, None, None, 0, 0.9, 0.0, None, None, 10, None, None,
                            None) == _lib.EINVAL
  assert lib.bm_study_stats(None, None, None, None, 0, None, None, None, None, 5, 0.9, 0.0, None, None, 0, rows, rows,
                            None) == _lib.EINVAL  # curv_mode out of range
  assert lib.bm_row_sqnorms(rows, 0, 10, None, None, None) == _lib.EINVAL
  out6 = (ctypes.c_double * 6)()
  assert lib.bm_momentum_stats_colwise(rows, 4, rows, 4, 10, 0.9, 0.1, None, None, None, rows, 1.0, 0, 0, 0, 0, rows,
                                       out6, rows, None) == _lib.EINVAL   # no Byzantine copy
  assert lib.bm_momentum_stats_colwise(rows, 4, rows, 4, 10, 0.9, 0.1, None, None, None, rows, 1.0, 16, 0, 0, 1, rows,
                                       out6, rows, None) == _lib.EINVAL   # direction-only attack vector
  assert lib.bm_momentum_stats_colwise(rows, 4, rows, 4, 10, 0.9, 0.1, None, None, None, rows, 1.0, 0, 1, 3, 1, rows,
                                       out6, rows, None) == _lib.EINVAL   # trmean with n = 5 < 2 f + 1
  assert lib.bm_pairwise_sqdist_shard(rows, 3, 10, 5, None, None, None) == _lib.EINVAL  # d_total < d
  assert lib.bm_workspace_bytes(_lib.WS_STUDY, 1, 1000) > 0
  assert b"RCCL" in lib.bm_error_string(_lib.ENOCOMM) and b"RCCL" in lib.bm_error_string(_lib.ECOMM)


def test_round4_entry_points_validate_arguments_without_gpu(lib):
  """bm_brute_select_device, bm_colwise_eval, bm_pairwise_rank: bad arguments are refused before any HIP call."""
  from byzantinemomentum_amd import _lib
  rows = (ctypes.c_void_p * 64)()
  buf = (ctypes.c_double * 8)()
  # the device Brute search: n within 1..64, 0 <= f < n, non-NULL buffers
  assert lib.bm_brute_select_device(None, 25, 5, rows, rows, None) == _lib.EINVAL
  assert lib.bm_brute_select_device(buf, 0, 0, rows, rows, None) == _lib.EINVAL
  assert lib.bm_brute_select_device(buf, 65, 5, rows, rows, None) == _lib.EINVAL
  assert lib.bm_brute_select_device(buf, 25, 25, rows, rows, None) == _lib.EINVAL
  assert lib.bm_brute_select_device(buf, 25, 5, rows, None, None) == _lib.EINVAL
  # the evaluate-only search: instances for the trimmed mean / phocas / meamed at n = 11, 25, 51 only
  assert lib.bm_colwise_eval_supported(_lib.OP_TRMEAN, 25) == 1 and lib.bm_colwise_eval_supported(_lib.OP_MEAMED, 51) == 1
  assert lib.bm_colwise_eval_supported(_lib.OP_PHOCAS, 11) == 1
  assert lib.bm_colwise_eval_supported(_lib.OP_MEDIAN, 25) == 0 and lib.bm_colwise_eval_supported(_lib.OP_TRMEAN, 24) == 0
  assert lib.bm_colwise_eval_workspace_bytes() >= 2 * 8
  args = (rows, 20, 5, 1000, 5, rows, rows, ctypes.c_float(1.0), rows, rows, None)
  assert lib.bm_colwise_eval(_lib.OP_MEDIAN, *args) == _lib.EINVAL            # no such instance
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 19, 5, *args[3:]) == _lib.EINVAL   # n = 24
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 20, 0, *args[3:]) == _lib.EINVAL   # no Byzantine copy
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 20, 5, 1000, 13, *args[5:]) == _lib.EINVAL  # n < 2 f + 1
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 20, 5, 1000, 5, None, rows, ctypes.c_float(1.0), rows, rows, None) == _lib.EINVAL
  # distances + ranking in one call
  assert lib.bm_pairwise_rank(rows, 25, 1000, 999, 5, 18, _lib.RANK_KRUM, rows, rows, None, rows, None) == _lib.EINVAL  # d_total < d
  assert lib.bm_pairwise_rank(rows, 25, 1000, 1000, 5, 18, 7, rows, rows, None, rows, None) == _lib.EINVAL            # mode
  assert lib.bm_pairwise_rank(rows, 25, 1000, 1000, 5, 18, _lib.RANK_KRUM, rows, None, None, rows, None) == _lib.EINVAL  # no order


def test_header_is_plain_c_and_a_c_host_links_and_runs(tmp_path):
  """The boundary is a C ABI: include/bm_gar.h compiles as C99 (gcc, -pedantic, no C++ in sight), a C host links
  against libbm_gar.so, reads the version and an error string, and every argument check answers BM_EINVAL — without a
  GPU, because validation comes before any HIP call."""
  import shutil
  import subprocess
  gcc = shutil.which("gcc")
  if gcc is None:
    pytest.skip("gcc not here")
  from byzantinemomentum_amd import build
  lib_path = build.LIB_PATH
  src = tmp_path / "host.c"
  src.write_text(r'''
#include <stdio.h>
#include <string.h>
#include "bm_gar.h"
int main(void) {
  const float* rows[2] = {0, 0};
  float out[1];
  int rc;
  printf("abi %d\n", bm_abi_version());
  if (strlen(bm_error_string(BM_EINVAL)) == 0) return 2;
  rc = bm_colwise(BM_OP_MEDIAN, rows, 0, 1, 0, out, 0);           /* n < 1 */
  if (rc != BM_EINVAL) return 3;
  rc = bm_colwise(BM_OP_MEDIAN, rows, BM_MAX_ROWS + 1, 1, 0, out, 0); /* too many rows */
  if (rc != BM_EINVAL) return 4;
  rc = bm_colwise(99, rows, 2, 1, 0, out, 0);                      /* unknown rule */
  if (rc != BM_EINVAL) return 5;
  if (bm_workspace_bytes(BM_WS_PAIRWISE, 25, 1000) <= 0) return 6;
  printf("ok\n");
  return 0;
}
''')
  exe = tmp_path / "host"
  include = pathlib.Path(build.INCLUDE_DIR)
  subprocess.run([gcc, "-std=c99", "-pedantic", "-Wall", "-Werror", "-fsyntax-only", "-I", str(include), str(src)],
                 check=True, capture_output=True)
  done = subprocess.run([gcc, "-std=c99", "-I", str(include), str(src), "-o", str(exe), str(lib_path),
                         f"-Wl,-rpath,{lib_path.parent}", "-Wl,-rpath,/opt/rocm/lib"], capture_output=True, text=True)
  assert done.returncode == 0, done.stderr
  run = subprocess.run([str(exe)], capture_output=True, text=True, timeout=120)
  assert run.returncode == 0, (run.returncode, run.stdout, run.stderr)
  from byzantinemomentum_amd import _lib
  assert run.stdout.split() == ["abi", str(_lib.ABI_VERSION), "ok"]


def test_second_pass_entry_points_validate_arguments_without_gpu(lib):
  """bm_bulyan_pass2, bm_colwise_eval: their argument checks (no launch).  ABI 19 dropped the `_walk` twins of both (a
  measured no-win, profiles/r05_a_second_pass_walk_ab.txt): the library must not export them any more."""
  from byzantinemomentum_amd import _lib
  rows = (ctypes.c_void_p * 64)()
  assert lib.bm_bulyan_pass2(rows, 25, None, 5, 18, 1000, rows, None) == _lib.EINVAL      # no ranking
  assert lib.bm_bulyan_pass2(rows, 22, rows, 5, 15, 1000, rows, None) == _lib.EINVAL      # n < 4 f + 3
  assert lib.bm_bulyan_pass2(rows, 25, rows, 5, 19, 1000, rows, None) == _lib.EINVAL      # m > n - f - 2
  assert lib.bm_bulyan_pass2(rows, 25, rows, 5, 18, 0, None, None) == 0                   # an empty shard
  tail = (rows, rows, ctypes.c_float(1.0))
  assert lib.bm_colwise_eval(_lib.OP_MEDIAN, rows, 20, 5, 1000, 5, *tail, rows, rows, None) == _lib.EINVAL
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 19, 5, 1000, 5, *tail, rows, rows, None) == _lib.EINVAL
  assert lib.bm_colwise_eval(_lib.OP_TRMEAN, rows, 20, 5, 1000, 5, None, rows, ctypes.c_float(1.0), rows, rows, None) == _lib.EINVAL
  for gone in ("bm_bulyan_pass2_walk", "bm_colwise_eval_walk"):
    assert not hasattr(lib, gone), gone
  assert lib.bm_abi_version() >= 19


def test_device_search_entry_points_validate_arguments_without_gpu(lib):
  """ABI 20: bm_attack_line_search_device (krum / average only: Brute's search is the host form's) and
  bm_multi_fma3_bdev — their argument checks (no launch)."""
  from byzantinemomentum_amd import _lib
  buf = (ctypes.c_double * 64)()
  krum, brute, average = _lib.RULE_IDS["krum"], _lib.RULE_IDS["brute"], _lib.RULE_IDS["average"]
  search = lib.bm_attack_line_search_device
  assert search(None, 20, 5, 5, krum, 0, 16, 0, buf, None) == _lib.EINVAL       # no matrix
  assert search(buf, 20, 5, 5, krum, 0, 16, 0, None, None) == _lib.EINVAL       # nowhere to write
  assert search(buf, 20, 5, 5, brute, 0, 16, 0, buf, None) == _lib.EINVAL       # host form only
  assert search(buf, 60, 5, 5, average, 0, 16, 0, buf, None) == _lib.EINVAL     # n > BM_MAX_ROWS
  assert search(buf, 20, 5, 5, krum, 26, 16, 0, buf, None) == _lib.EINVAL       # m > n
  assert search(buf, 20, 5, 5, krum, 0, 0, 0, buf, None) == _lib.EINVAL         # no evaluation
  rows = (ctypes.c_void_p * 64)()
  assert lib.bm_multi_fma3_bdev(rows, rows, rows, 1, 1000, ctypes.c_float(1.0), None, None, None) == _lib.EINVAL
  assert lib.bm_multi_fma3_bdev(rows, rows, rows, 1, 0, ctypes.c_float(1.0), buf, None, None) == 0   # empty vectors
  # ABI 21: the cursor in device memory and the evaluation that reads its factor there
  nxt = lib.bm_search_device_next
  shape = (ctypes.c_double(0.0), ctypes.c_double(1.0), ctypes.c_double(0.8))
  assert nxt(None, None, 0, 0, *shape, buf, buf, None) == _lib.EINVAL            # no state
  assert nxt(buf, None, 0, 0, *shape, None, buf, None) == _lib.EINVAL            # nowhere to put the candidate
  assert nxt(buf, None, 0, 1, *shape, None, buf, None) == _lib.EINVAL            # the last call needs the last objective
  assert nxt(buf, None, 0, 0, ctypes.c_double(0.0), ctypes.c_double(1.0), ctypes.c_double(0.4), buf, buf, None) == _lib.EINVAL
  tail = (rows, rows)
  assert lib.bm_colwise_eval_tdev(_lib.OP_TRMEAN, rows, 20, 5, 1000, 5, *tail, None, rows, rows, None) == _lib.EINVAL
  assert lib.bm_colwise_eval_tdev(_lib.OP_MEDIAN, rows, 20, 5, 1000, 5, *tail, buf, rows, rows, None) == _lib.EINVAL
  # ABI 22: the objective of a candidate in one pass over two vectors; the median's own search as the middle of three
  assert lib.bm_sqdist2(None, rows, 1000, buf, buf, None) == _lib.EINVAL          # no first vector
  assert lib.bm_sqdist2(rows, rows, 1000, None, buf, None) == _lib.EINVAL         # nowhere to write
  assert lib.bm_sqdist2(rows, rows, 1000, buf, None, None) == _lib.EINVAL         # no workspace
  assert lib.bm_sqdist2(rows, rows, -1, buf, buf, None) == _lib.EINVAL            # negative length
  assert lib.bm_colwise_eval_supported(_lib.OP_MEDIAN, 3) == 1 and lib.bm_colwise_eval_supported(_lib.OP_MEDIAN, 5) == 0
  assert lib.bm_colwise_eval_supported(_lib.OP_TRMEAN, 3) == 0
  assert lib.bm_colwise_eval_tdev(_lib.OP_MEDIAN, rows, 2, 2, 1000, 0, *tail, buf, rows, rows, None) == _lib.EINVAL  # n = 4
  assert lib.bm_order_pair_supported(1) == 1 and lib.bm_order_pair_supported(51) == 1
  assert lib.bm_order_pair_supported(0) == 0 and lib.bm_order_pair_supported(52) == 0
  assert lib.bm_order_pair(None, 20, 1000, 7, 12, buf, rows, None) == _lib.EINVAL      # no rows
  assert lib.bm_order_pair(rows, 52, 1000, 7, 12, buf, rows, None) == _lib.EINVAL      # more rows than the largest network
  assert lib.bm_order_pair(rows, 20, -1, 7, 12, buf, rows, None) == _lib.EINVAL        # negative length
  assert lib.bm_order_pair(rows, 20, 1000, 7, 12, buf, buf, None) == _lib.EINVAL       # lo and hi are one buffer
  assert lib.bm_order_pair(rows, 20, 1000, 7, 12, None, buf, None) == _lib.EINVAL      # nowhere to write
  assert lib.bm_order_pair(rows, 20, 1000, 7, 12, buf, rows, None) == _lib.EINVAL      # a null row (the table is empty)
  assert lib.bm_order_pair(rows, 20, 0, 7, 12, None, None, None) == 0                  # empty vectors: nothing to do
  # ABI 23: Bulyan's second pass, evaluate only
  sup = lib.bm_bulyan_pass2_eval_supported
  assert sup(25, 5, 18) == 1 and sup(11, 2, 7) == 1 and sup(51, 12, 37) == 1
  assert sup(25, 5, 17) == 0 and sup(15, 3, 10) == 0 and sup(51, 10, 39) == 0
  ev = lib.bm_bulyan_pass2_eval
  one = ctypes.c_float(1.0)
  assert ev(None, 20, 5, rows, 5, 18, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # no rows
  assert ev(rows, 20, 5, None, 5, 18, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # no ranking
  assert ev(rows, 20, 5, rows, 5, 18, 1000, None, rows, one, None, buf, buf, None) == _lib.EINVAL    # no average
  assert ev(rows, 20, 5, rows, 5, 18, 1000, rows, rows, one, None, None, buf, None) == _lib.EINVAL   # nowhere to write
  assert ev(rows, 20, 5, rows, 5, 18, 1000, rows, rows, one, None, buf, None, None) == _lib.EINVAL   # no workspace
  assert ev(rows, 20, 5, rows, 5, 17, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # m != n - f - 2
  assert ev(rows, 19, 5, rows, 5, 17, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # n = 24: no instance
  assert ev(rows, 20, 0, rows, 5, 18, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # no copy of the candidate
  assert ev(rows, 20, 5, rows, 5, 18, (1 << 29) + 1, rows, rows, one, None, buf, buf, None) == _lib.EINVAL   # too long for one launch
  assert ev(rows, 20, 5, rows, 5, 18, 1000, rows, rows, one, None, buf, buf, None) == _lib.EINVAL    # a null row (the table is empty)
  rk = lib.bm_attack_ranking_device
  assert rk(None, 20, 5, 5, _lib.RANK_BULYAN, 0, buf, rows, None) == _lib.EINVAL     # no matrix
  assert rk(buf, 20, 5, 5, _lib.RANK_BULYAN, 0, None, rows, None) == _lib.EINVAL     # no factor
  assert rk(buf, 20, 5, 5, _lib.RANK_BULYAN, 0, buf, None, None) == _lib.EINVAL      # nowhere to write
  assert rk(buf, 20, 0, 5, _lib.RANK_BULYAN, 0, buf, rows, None) == _lib.EINVAL      # no copy of the candidate
  assert rk(buf, 60, 5, 5, _lib.RANK_KRUM, 0, buf, rows, None) == _lib.EINVAL        # n > BM_MAX_ROWS
  assert rk(buf, 20, 5, 5, 7, 0, buf, rows, None) == _lib.EINVAL                     # no such mode
  assert rk(buf, 20, 5, 5, _lib.RANK_KRUM, 26, buf, rows, None) == _lib.EINVAL       # m > n
  assert lib.bm_abi_version() == 23
