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
