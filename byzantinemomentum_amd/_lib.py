"""ctypes binding of libbm_gar.so (declared in include/bm_gar.h).

`import torch` must precede the CDLL so that the loader resolves our DT_NEEDED
libamdhip64.so.7 to the copy torch has already mapped (one HIP runtime per process,
streams and device pointers are then shared with torch).

There is NO fallback: if the library is missing or does not load, every product entry
point raises.  The CPU oracle under oracle/ is test infrastructure and is never used here.
"""

import ctypes
import pathlib

import torch  # noqa: F401  (must be imported before the CDLL, see above)

import os

_PKG_DIR = pathlib.Path(__file__).resolve().parent
# BM_GAR_LIB: another build of the SAME library (A/B runs of a kernel against an older version of itself,
# scripts/gram_variant_probe.py); the default, and the only thing the tests and the bench load, is the in-tree build.
LIB_PATH = pathlib.Path(os.environ["BM_GAR_LIB"]).resolve() if os.environ.get("BM_GAR_LIB") else _PKG_DIR / "libbm_gar.so"

ABI_VERSION = 23
MAX_ROWS = 64
EINVAL = -100000
ENOCOMM, ECOMM = -100001, -100002

OP_MEDIAN, OP_TRMEAN, OP_PHOCAS, OP_MEAMED = 0, 1, 2, 3
WS_PAIRWISE, WS_AKSEL, WS_STATS, WS_DOT, WS_STEP, WS_STUDY = 0, 1, 2, 3, 4, 5
STUDY_SLOTS = 32
RANK_KRUM, RANK_BULYAN = 0, 1
ATTACK_EMPIRE, ATTACK_LITTLE, ATTACK_DIRECTION = 0, 1, 16

_c_float_pp = ctypes.POINTER(ctypes.c_void_p)

# name -> (restype, argtypes); mirrors include/bm_gar.h one to one
SIGNATURES = {
  "bm_abi_version": (ctypes.c_int, []),
  "bm_tuning_set": (ctypes.c_int, [ctypes.c_char_p, ctypes.c_int]),
  "bm_error_string": (ctypes.c_char_p, [ctypes.c_int]),
  "bm_colwise": (ctypes.c_int, [ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                ctypes.c_void_p, ctypes.c_void_p]),
  "bm_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int, ctypes.c_int64]),
  "bm_pairwise_sqdist": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                        ctypes.c_void_p, ctypes.c_void_p]),
  "bm_pairwise_sqdist_shard": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_krum_rank": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_selected_mean": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_void_p, ctypes.c_int,
                                      ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_bulyan_pass2": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                     ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_aksel_pass1": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_stack_stats": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_float, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]),
  "bm_multi_dot": (ctypes.c_int, [_c_float_pp, ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                  ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_momentum_stats_colwise": (ctypes.c_int, [_c_float_pp, ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                               ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int,
                                               ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                               ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_momentum_stats_sqdist": (ctypes.c_int, [_c_float_pp, ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                              ctypes.c_int64, ctypes.c_float, ctypes.c_float, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float,
                                              ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_sharded_sq_slot": (ctypes.c_void_p, [ctypes.c_void_p]),
  "bm_sharded_pair_workspace": (ctypes.c_void_p, [ctypes.c_void_p]),
  "bm_sharded_rule_from_sq": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                             ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                             ctypes.c_void_p, ctypes.c_void_p]),
  "bm_stack_stats_colwise": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                            ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_stack_stats_sqdist": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_float, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_study_stats": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                    ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_study_stats_update": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int,
                                           ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p, ctypes.c_float, ctypes.c_float, ctypes.c_int64,
                                           ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_row_sqnorms": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                    ctypes.c_void_p]),
  "bm_stable_argsort": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_multi_axpby": (ctypes.c_int, [_c_float_pp, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                    ctypes.c_float, ctypes.c_float, ctypes.c_void_p]),
  "bm_brute_select": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
  "bm_momentum_stats": (ctypes.c_int, [_c_float_pp, ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                       ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_int,
                                       ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_multi_fma3": (ctypes.c_int, [_c_float_pp, _c_float_pp, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                   ctypes.c_float, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_multi_fma3_bdev": (ctypes.c_int, [_c_float_pp, _c_float_pp, _c_float_pp, ctypes.c_int, ctypes.c_int64,
                                        ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_clip_factors": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_float, ctypes.c_void_p,
                                     ctypes.c_void_p]),
  "bm_multi_scale": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_comm_available": (ctypes.c_int, []),
  "bm_comm_unique_id": (ctypes.c_int, [ctypes.c_void_p]),
  "bm_comm_init": (ctypes.c_int, [ctypes.POINTER(ctypes.c_void_p), ctypes.c_int, ctypes.c_int, ctypes.c_void_p]),
  "bm_comm_destroy": (ctypes.c_int, [ctypes.c_void_p]),
  "bm_comm_size": (ctypes.c_int, [ctypes.c_void_p]),
  "bm_allreduce_sum_f64": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p]),
  "bm_allgather_f32": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64,
                                      ctypes.c_void_p]),
  "bm_brute_select_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p,
                                            ctypes.c_void_p]),
  "bm_colwise_eval_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
  "bm_colwise_eval_workspace_bytes": (ctypes.c_int64, []),
  "bm_colwise_eval": (ctypes.c_int, [ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int64, ctypes.c_int,
                                     ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p, ctypes.c_void_p,
                                     ctypes.c_void_p]),
  "bm_colwise_eval_tdev": (ctypes.c_int, [ctypes.c_int, _c_float_pp, ctypes.c_int, ctypes.c_int, ctypes.c_int64,
                                          ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p]),
  "bm_sqdist2": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p,
                                ctypes.c_void_p]),
  "bm_attack_ranking_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                              ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_bulyan_pass2_eval_supported": (ctypes.c_int, [ctypes.c_int, ctypes.c_int, ctypes.c_int]),
  "bm_bulyan_pass2_eval": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_int, ctypes.c_int,
                                          ctypes.c_int64, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_float, ctypes.c_void_p,
                                          ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_order_pair_supported": (ctypes.c_int, [ctypes.c_int]),
  "bm_order_pair": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int, ctypes.c_int, ctypes.c_void_p,
                                   ctypes.c_void_p, ctypes.c_void_p]),
  "bm_search_device_next": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_double,
                                           ctypes.c_double, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                           ctypes.c_void_p]),
  "bm_pairwise_rank": (ctypes.c_int, [_c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64, ctypes.c_int, ctypes.c_int,
                                      ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                      ctypes.c_void_p]),
  "bm_sharded_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int64]),
  "bm_sharded_krum": (ctypes.c_int, [ctypes.c_void_p, _c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                     ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_sharded_bulyan": (ctypes.c_int, [ctypes.c_void_p, _c_float_pp, ctypes.c_int, ctypes.c_int64, ctypes.c_int64,
                                       ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_void_p,
                                       ctypes.c_void_p]),
  "bm_step_stats_count": (ctypes.c_int, []),
  "bm_step_workspace_bytes": (ctypes.c_int64, [ctypes.c_int, ctypes.c_int64]),
  "bm_step_worker": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p, _c_float_pp, _c_float_pp, ctypes.c_int64,
                                    ctypes.c_int64] + [ctypes.c_void_p] * 13),
  "bm_search_begin": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double, ctypes.c_double, ctypes.c_double]),
  "bm_search_propose": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_void_p]),
  "bm_search_report": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_double]),
  "bm_attack_objective": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                         ctypes.c_int, ctypes.c_double, ctypes.c_void_p, ctypes.c_void_p,
                                         ctypes.c_void_p]),
  "bm_attack_line_search": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p]),
  "bm_attack_line_search_device": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                                  ctypes.c_void_p, ctypes.c_void_p]),
  "bm_attack_ranking": (ctypes.c_int, [ctypes.c_void_p, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       ctypes.c_int, ctypes.c_double, ctypes.c_void_p]),
}

class Search(ctypes.Structure):
  """bm_search of include/bm_gar.h: the caller-owned cursor of the factor search."""
  _fields_ = [("best_x", ctypes.c_double), ("best_y", ctypes.c_double), ("probe", ctypes.c_double),
              ("step", ctypes.c_double), ("ratio", ctypes.c_double), ("phase", ctypes.c_int32),
              ("evaluations", ctypes.c_int32), ("awaiting", ctypes.c_int32), ("reserved", ctypes.c_int32)]


RULE_IDS = {"krum": 0, "bulyan": 1, "median": 2, "trmean": 3, "phocas": 4, "meamed": 5, "brute": 6, "average": 7}


class StepParams(ctypes.Structure):
  """bm_step_params of include/bm_gar.h."""
  _fields_ = [("n", ctypes.c_int32), ("f_decl", ctypes.c_int32), ("f_real", ctypes.c_int32), ("ks", ctypes.c_int32),
              ("rule", ctypes.c_int32), ("m", ctypes.c_int32), ("attack_kind", ctypes.c_int32),
              ("nb_past", ctypes.c_int32), ("past_count", ctypes.c_int32), ("attack_scale", ctypes.c_float),
              ("mu", ctypes.c_float), ("one_minus_damp", ctypes.c_float), ("clip", ctypes.c_float),
              ("oldest_weight", ctypes.c_float)]


class NativeLibraryError(RuntimeError):
  """libbm_gar.so is missing, stale or failed to load: the HIP path cannot run."""


_lib = None


def load():
  """Load (once) and return the ctypes handle; raise NativeLibraryError on any problem."""
  global _lib
  if _lib is not None:
    return _lib
  if not LIB_PATH.exists():
    raise NativeLibraryError(
      f"{LIB_PATH} not found: build it with `python -m byzantinemomentum_amd.build` "
      "(hipcc --offload-arch=gfx950). There is no CPU fallback.")
  try:
    lib = ctypes.CDLL(str(LIB_PATH))
  except OSError as err:
    raise NativeLibraryError(f"cannot load {LIB_PATH}: {err}") from err
  for name, (restype, argtypes) in SIGNATURES.items():
    try:
      fn = getattr(lib, name)
    except AttributeError as err:
      raise NativeLibraryError(f"{LIB_PATH} does not export {name!r} (stale build?)") from err
    fn.restype = restype
    fn.argtypes = argtypes
  if lib.bm_abi_version() != ABI_VERSION:
    raise NativeLibraryError(
      f"{LIB_PATH} has ABI {lib.bm_abi_version()}, host code expects {ABI_VERSION}: rebuild")
  _lib = lib
  return lib


def check(code, what):
  """Turn a negative status of an entry point into a RuntimeError."""
  if code != 0:
    msg = load().bm_error_string(code).decode("utf-8", "replace")
    raise RuntimeError(f"libbm_gar {what} failed: {msg} (code {code})")


def pointer_table(tensors):
  """Host array of device pointers for a list of tensors (aliased entries allowed)."""
  arr = (ctypes.c_void_p * len(tensors))()
  for i, t in enumerate(tensors):
    arr[i] = t.data_ptr()
  return arr
