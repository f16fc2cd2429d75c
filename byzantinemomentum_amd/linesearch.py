"""The factor search of the "identical" attacks (attacks/identical.py:67-77, tools/misc.py:468-514).

Host logic of libbm_gar.so (csrc/linesearch.cpp), no device work here:
  line_maximize          the exploration routine around any Python callable, driven through the bm_search_* cursor;
  attack_objective       one evaluation of |GAR(honests + [avg + t*att]*k) - avg|^2 from scalars only,
  attack_line_search     the whole search from scalars only
for the rules whose output is the mean of a selected subset (krum, brute, average), and
  attack_ranking         the ranking of a candidate stack from scalars only (Bulyan ranks its candidates with it and
                         runs only its second pass on the vectors).
The scalars are the (h+2) x (h+2) squared distances among the h honest rows, their average and average + att: ONE
distance pass instead of `evals` evaluations of the rule on d-sized vectors.
"""

import ctypes

import torch

from . import _lib

ANALYTIC_RULES = ("krum", "brute", "average")


def line_maximize(scape, evals=16, start=0., delta=1., ratio=0.8):
  """The exploration of tools.line_maximize (tools/misc.py:468-514) around any Python callable: returns
  (best x, [(x, y) in evaluation order]).  The candidates come from the library's cursor (bm_search_*), the
  evaluations are made here, between a proposal and its report — nothing calls back through the C frame, so
  an exception of `scape` propagates as it is."""
  lib = _lib.load()
  if not isinstance(evals, int) or evals < 1:
    _lib.check(_lib.EINVAL, "line_maximize (evals must be a positive integer)")
  cursor = _lib.Search()
  _lib.check(lib.bm_search_begin(ctypes.byref(cursor), start, delta, ratio), "bm_search_begin")
  x = ctypes.c_double()
  trace = []
  for _ in range(evals):
    _lib.check(lib.bm_search_propose(ctypes.byref(cursor), ctypes.byref(x)), "bm_search_propose")
    y = float(scape(x.value))
    trace.append((x.value, y))
    _lib.check(lib.bm_search_report(ctypes.byref(cursor), y), "bm_search_report")
  return cursor.best_x, trace


def _ext_pointer(ext, h):
  if not (isinstance(ext, torch.Tensor) and ext.dtype == torch.float64 and ext.device.type == "cpu"
          and ext.is_contiguous() and tuple(ext.shape) == (h + 2, h + 2)):
    raise ValueError(f"ext must be a contiguous float64 CPU tensor of shape ({h + 2}, {h + 2})")
  return ctypes.c_void_p(ext.data_ptr())


def attack_objective(ext, h, k, f, rule, t, m=None):
  """(objective, indices the rule averages; >= h: Byzantine copies) at attack factor t."""
  lib = _lib.load()
  y = ctypes.c_double()
  sel = (ctypes.c_int32 * _lib.MAX_ROWS)()
  count = ctypes.c_int32()
  _lib.check(lib.bm_attack_objective(_ext_pointer(ext, h), h, k, f, _lib.RULE_IDS[rule], m or 0, float(t),
                                     ctypes.cast(ctypes.pointer(y), ctypes.c_void_p),
                                     ctypes.cast(sel, ctypes.c_void_p),
                                     ctypes.cast(ctypes.pointer(count), ctypes.c_void_p)), "bm_attack_objective")
  return y.value, list(sel[:count.value])


def attack_ranking(ext, h, k, f, mode, t, m=None):
  """The ranking bm_krum_rank would give for honests + [avg + t*att] * k (mode "krum" / "bulyan"), from the scalars:
  n indices, those >= h being Byzantine copies.  Bulyan's factor search ranks with it and runs only the second pass
  of the rule on the vectors (step.py)."""
  lib = _lib.load()
  order = (ctypes.c_int32 * _lib.MAX_ROWS)()
  mode_id = {"krum": _lib.RANK_KRUM, "bulyan": _lib.RANK_BULYAN}[mode]
  _lib.check(lib.bm_attack_ranking(_ext_pointer(ext, h), h, k, f, mode_id, m or 0, float(t),
                                   ctypes.cast(order, ctypes.c_void_p)), "bm_attack_ranking")
  return list(order[:h + k])


def attack_line_search(ext, h, k, f, rule, evals=16, negative=False, m=None):
  """(factor, [(x, y) in evaluation order]) of the search at identical.py:67-77 for `rule`."""
  if rule not in ANALYTIC_RULES:
    raise ValueError(f"no scalar form of the search for rule {rule!r}")
  lib = _lib.load()
  factor = ctypes.c_double()
  trace = (ctypes.c_double * (2 * evals))()
  _lib.check(lib.bm_attack_line_search(_ext_pointer(ext, h), h, k, f, _lib.RULE_IDS[rule], m or 0, evals,
                                       1 if negative else 0, ctypes.cast(ctypes.pointer(factor), ctypes.c_void_p),
                                       ctypes.cast(trace, ctypes.c_void_p)), "bm_attack_line_search")
  return factor.value, [(trace[2 * i], trace[2 * i + 1]) for i in range(evals)]
