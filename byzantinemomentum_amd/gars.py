"""Host side of the gradient aggregation rules: argument plumbing around libbm_gar.so.

Mirrors the reference's plugin surface (aggregators/__init__.py:15-31): every rule takes a
Python `list` of n flat fp32 tensors living on one GPU (entries may alias) plus `f`, returns a
NEW tensor, never touches its inputs, and runs asynchronously on the caller's current stream.
All arithmetic happens in the HIP kernels; nothing here falls back to torch ops or to the CPU
oracle — on a machine without the library or without a GPU these functions raise.

Reference semantics, per rule:
  median   aggregators/median.py:31-39      trmean/phocas/meamed  aggregators/trmean.py:24-109
  krum     aggregators/krum.py:31-80        bulyan                aggregators/bulyan.py:31-84
  brute    aggregators/brute.py:32-80       aksel                 aggregators/aksel.py:24-64
  average  aggregators/average.py:21-29     cge                   aggregators/cge.py:28-57
"""

import ctypes
import weakref

import torch

from . import _lib

__all__ = ["median", "trmean", "phocas", "meamed", "krum", "bulyan", "brute", "aksel", "average", "cge",
           "krum_selection", "bulyan_ranking", "brute_selection", "aksel_selection", "cge_selection",
           "pairwise_sqdist", "rank_from_sqdist", "selected_mean", "bulyan_pass2", "aksel_sqdist",
           "stable_argsort", "GarInputError"]


class GarInputError(ValueError):
  """The gradients cannot be served by the HIP path (wrong device, dtype, layout or count)."""


# ---------------------------------------------------------------------------- #
# Input validation and scratch memory

def _validate(gradients):
  if not isinstance(gradients, (list, tuple)) or len(gradients) < 1:
    raise GarInputError(f"expected a non-empty list of gradients, got {type(gradients).__name__}")
  n = len(gradients)
  if n > _lib.MAX_ROWS:
    raise GarInputError(f"at most {_lib.MAX_ROWS} gradients are supported, got {n}")
  g0 = gradients[0]
  if not isinstance(g0, torch.Tensor):
    raise GarInputError("gradients must be torch tensors")
  if not g0.is_cuda:
    raise GarInputError(
      "the MI355X aggregation path needs gradients on a GPU (device 'cuda:N'); there is no CPU fallback")
  for g in gradients:
    if (g.device != g0.device or g.dtype != torch.float32 or g.dim() != 1 or g.shape != g0.shape
        or not g.is_contiguous()):
      raise GarInputError("gradients must be contiguous 1-D float32 tensors of equal length on one device")
  return n, g0.shape[0], g0.device


def _stream(device):
  return ctypes.c_void_p(torch.cuda.current_stream(device).cuda_stream)


class _Scratch:
  """Per-(device, stream) scratch tensors. The C library owns nothing; this is the caller side."""
  _cache = {}

  @classmethod
  def get(cls, device, name, nbytes=None, shape=None, dtype=torch.uint8):
    key = (device.index, torch.cuda.current_stream(device).cuda_stream, name)
    buf = cls._cache.get(key)
    want = (nbytes,) if shape is None else tuple(shape)
    if buf is None or buf.dtype != dtype or tuple(buf.shape) != want:
      buf = torch.empty(want, dtype=dtype, device=device)
      cls._cache[key] = buf
    return buf


def _workspace(device, kind, n, d, name):
  nbytes = _lib.load().bm_workspace_bytes(kind, n, d)
  if nbytes < 0:
    raise RuntimeError(f"bm_workspace_bytes({kind}, {n}, {d}) failed")
  return _Scratch.get(device, name, nbytes=int(nbytes))


def _ptr(t):
  return ctypes.c_void_p(t.data_ptr())


# ---------------------------------------------------------------------------- #
# Coordinate-wise rules

def _colwise(op, gradients, f):
  n, d, device = _validate(gradients)
  lib = _lib.load()
  out = torch.empty(d, dtype=torch.float32, device=device)
  if d == 0:
    return out
  with torch.cuda.device(device):
    _lib.check(lib.bm_colwise(op, _lib.pointer_table(gradients), n, d, f, _ptr(out), _stream(device)),
               "bm_colwise")
  return out


def median(gradients, **kwargs):
  """Coordinate-wise lower median (aggregators/median.py:31-39)."""
  return _colwise(_lib.OP_MEDIAN, gradients, 0)


def trmean(gradients, f, **kwargs):
  """Coordinate-wise trimmed mean of sorted ranks f..n-f-1 (aggregators/trmean.py:69-79)."""
  return _colwise(_lib.OP_TRMEAN, gradients, f)


def phocas(gradients, f, **kwargs):
  """Mean of the n-f values closest to the trimmed mean (aggregators/trmean.py:81-94)."""
  return _colwise(_lib.OP_PHOCAS, gradients, f)


def meamed(gradients, f, **kwargs):
  """Mean of the n-f values closest to the median (aggregators/trmean.py:96-109)."""
  return _colwise(_lib.OP_MEAMED, gradients, f)


# ---------------------------------------------------------------------------- #
# Distance-based rules

def pairwise_sqdist(gradients, d_total=None):
  """n x n fp64 matrix of squared L2 distances, on the device, no host sync.  d_total: length of the whole
  vectors when `gradients` are one shard of a dimension-partitioned stack (the precision plan follows it)."""
  n, d, device = _validate(gradients)
  lib = _lib.load()
  sq = torch.empty((n, n), dtype=torch.float64, device=device)  # result: a fresh tensor per call
  ws = _workspace(device, _lib.WS_PAIRWISE, n, d, "ws_pair")
  with torch.cuda.device(device):
    _lib.check(lib.bm_pairwise_sqdist_shard(_lib.pointer_table(gradients), n, d, d if d_total is None else int(d_total),
                                            _ptr(sq), _ptr(ws), _stream(device)), "bm_pairwise_sqdist_shard")
  return sq


def rank_from_sqdist(sq, n, f, m, mode):
  """Scores + stable order from an n x n squared-distance matrix that is already on the device
  (possibly the all-reduced sum of per-shard partial matrices). Returns (order, scores)."""
  device = sq.device
  lib = _lib.load()
  order = torch.empty(_lib.MAX_ROWS, dtype=torch.int32, device=device)
  scores = torch.empty(_lib.MAX_ROWS, dtype=torch.float64, device=device)
  with torch.cuda.device(device):
    _lib.check(lib.bm_krum_rank(_ptr(sq), n, f, m, mode, _ptr(order), _ptr(scores), _stream(device)),
               "bm_krum_rank")
  return order, scores


def _rank(gradients, f, m, mode):
  """Distances -> scores -> stable order, all on the device, in the distance pass's own launches (bm_pairwise_rank:
  its last workgroups rank the rows).  Returns (order int32[MAX_ROWS], scores f64[MAX_ROWS])."""
  n, d, device = _validate(gradients)
  lib = _lib.load()
  sq = torch.empty((n, n), dtype=torch.float64, device=device)
  order = torch.empty(_lib.MAX_ROWS, dtype=torch.int32, device=device)
  scores = torch.empty(_lib.MAX_ROWS, dtype=torch.float64, device=device)
  ws = _workspace(device, _lib.WS_PAIRWISE, n, d, "ws_pair")
  with torch.cuda.device(device):
    _lib.check(lib.bm_pairwise_rank(_lib.pointer_table(gradients), n, d, d, f, m, mode, _ptr(sq), _ptr(order),
                                    _ptr(scores), _ptr(ws), _stream(device)), "bm_pairwise_rank")
  return order, scores


def selected_mean(gradients, idx, m):
  """Sequential fp32 mean of gradients[idx[0..m)], idx being a DEVICE int32 tensor."""
  n, d, device = _validate(gradients)
  lib = _lib.load()
  out = torch.empty(d, dtype=torch.float32, device=device)
  if d == 0:
    return out
  with torch.cuda.device(device):
    _lib.check(lib.bm_selected_mean(_lib.pointer_table(gradients), n, _ptr(idx), m, d, _ptr(out),
                                    _stream(device)), "bm_selected_mean")
  return out


# Last ranking per rule, so that `influence` right after `aggregate` (attack.py:821-822) does not
# recompute the distance matrix.  An entry is only reused for the SAME tensor objects (weak
# references: a freed gradient can never alias a new one) at the same in-place version.
_last_rank = {}


def invalidate_rank_cache():
  """Forget cached rankings; called by every routine of this package that writes user tensors."""
  _last_rank.clear()


def _rank_cache_get(tag, gradients, params):
  hit = _last_rank.get(tag)
  if hit is None or hit[0] != params or len(hit[1]) != len(gradients):
    return None
  for (ref, version), g in zip(hit[1], gradients):
    if ref() is not g or g._version != version:
      return None
  return hit[2]


def _rank_cache_put(tag, gradients, params, order):
  _last_rank[tag] = (params, [(weakref.ref(g), g._version) for g in gradients], order)


def _cached_rank(tag, gradients, f, m, mode):
  order = _rank_cache_get(tag, gradients, (f, m))
  if order is None:
    order, _ = _rank(gradients, f, m, mode)
    _rank_cache_put(tag, gradients, (f, m), order)
  return order


def krum(gradients, f, m=None, **kwargs):
  """Multi-Krum (aggregators/krum.py:65-80): mean, in score order, of the m best-scored rows."""
  n = len(gradients)
  if m is None:
    m = n - f - 2
  order = _cached_rank("krum", gradients, f, m, _lib.RANK_KRUM)
  return selected_mean(gradients, order, m)


def krum_selection(gradients, f, m=None, **kwargs):
  """Indices (score order) of the m rows Multi-Krum averages — host list, synchronises."""
  n = len(gradients)
  if m is None:
    m = n - f - 2
  order = _cached_rank("krum", gradients, f, m, _lib.RANK_KRUM)
  return order[:m].tolist()


def bulyan_ranking(gradients, f, m=None, **kwargs):
  """Initial stable score order used by every Bulyan iteration — host list, synchronises."""
  n = len(gradients)
  if m is None:
    m = n - f - 2
  order = _cached_rank("bulyan", gradients, f, m, _lib.RANK_BULYAN)
  return order[:n].tolist()


def bulyan_pass2(gradients, order, f, m):
  """Second pass of Bulyan given the device-resident ranking (aggregators/bulyan.py:64-84)."""
  n, d, device = _validate(gradients)
  lib = _lib.load()
  out = torch.empty(d, dtype=torch.float32, device=device)
  if d == 0:
    return out
  with torch.cuda.device(device):
    _lib.check(lib.bm_bulyan_pass2(_lib.pointer_table(gradients), n, _ptr(order), f, m, d, _ptr(out), _stream(device)),
               "bm_bulyan_pass2")
  return out


def bulyan(gradients, f, m=None, **kwargs):
  """Bulyan over Multi-Krum (aggregators/bulyan.py:31-84)."""
  n = len(gradients)
  if m is None:
    m = n - f - 2
  order = _cached_rank("bulyan", gradients, f, m, _lib.RANK_BULYAN)
  return bulyan_pass2(gradients, order, f, m)


def brute_select_host(dist_host, n, f):
  """Subset search of the Brute rule on a HOST fp64 n x n distance matrix (aggregators/brute.py:47-68)."""
  lib = _lib.load()
  sel = (ctypes.c_int32 * (n - f))()
  rc = lib.bm_brute_select(ctypes.c_void_p(dist_host.data_ptr()), n, f, ctypes.cast(sel, ctypes.c_void_p))
  if rc != 0:
    raise RuntimeError("brute: too many non-finite gradients, no subset of n-f rows has a finite diameter")
  return list(sel)


def brute_select_device(sq, n, f):
  """Subset search of the Brute rule on a DEVICE fp64 n x n matrix of SQUARED distances (bm_brute_select_device: one
  workgroup, no host round trip).  Returns (sel int32[MAX_ROWS]: the n - f rows ascending, status int32[1]: 0, -1 when
  every subset touches a non-finite distance, -2 when the search gave up on its node budget), both on the device."""
  lib = _lib.load()
  device = sq.device
  sel = torch.empty(_lib.MAX_ROWS, dtype=torch.int32, device=device)
  status = torch.empty(1, dtype=torch.int32, device=device)
  with torch.cuda.device(device):
    _lib.check(lib.bm_brute_select_device(_ptr(sq), n, f, _ptr(sel), _ptr(status), _stream(device)),
               "bm_brute_select_device")
  return sel, status


def _brute_sel(gradients, f):
  """(sel, status) of the stack, from the cache when `influence` follows `aggregate` on the same tensors
  (attack.py:821-822): the distance pass is then not repeated."""
  hit = _rank_cache_get("brute", gradients, (f,))
  if hit is None:
    n, d, device = _validate(gradients)
    # brute keeps non-finite distances as they are and skips the subsets that contain one (brute.py:45,56-57)
    hit = brute_select_device(pairwise_sqdist(gradients), n, f)
    _rank_cache_put("brute", gradients, (f,), hit)
  return hit


def brute_selection(gradients, f, **kwargs):
  """Index set (ascending) of the n-f rows of smallest diameter (aggregators/brute.py:32-68) — host list,
  synchronises; raises when no subset of n-f rows has a finite diameter (the reference then has no selection)."""
  n = len(gradients)
  sel, status = _brute_sel(gradients, f)
  if int(status.item()) == -2:  # the device search gave up: the host search has no budget
    sel = brute_host_selection(gradients, f)
    _rank_cache_put("brute", gradients, (f,), (sel, torch.zeros(1, dtype=torch.int32, device=sel.device)))
  else:
    brute_check(status)
  return sel[:n - f].tolist()


BRUTE_NO_SUBSET = "brute: too many non-finite gradients, no subset of n-f rows has a finite diameter"
BRUTE_BUDGET = ("brute: the device search gave up after its budget of search-tree nodes (csrc/brute.hip; a distance matrix "
                "built against the search?) — gars.brute_select_host on the host has no such limit")
# (convenience only: the status of the latest brute() of this PROCESS, whatever its device or stream; the status of a
#  given call travels with its result, `result.brute_status`, and that is what callers with several devices, streams
#  or aggregators must read — ShardedAggregator keeps its own)
last_brute_status = None


def brute_check(status=None):
  """Raise when the given Brute search (default: the latest one of this process) found no admissible subset — the
  reference's assertion (brute.py:68) — or gave up on its node budget.  Synchronises (one 4-byte read); not to be
  called while a stream is being captured."""
  status = last_brute_status if status is None else status
  code = 0 if status is None else int(status.item())
  if code == -2:
    raise RuntimeError(BRUTE_BUDGET)
  if code != 0:
    raise RuntimeError(BRUTE_NO_SUBSET)


def brute_host_selection(gradients, f, sq=None):
  """The selection of the Brute rule by the HOST search (bm_brute_select: no node budget), as a device index tensor;
  one synchronisation.  What the checked paths fall back to when the device search reports status -2 — the reference
  would keep computing (brute.py:47-68), so does this."""
  n, d, device = _validate(gradients)
  if sq is None:
    sq = pairwise_sqdist(gradients)
  sel = brute_select_host(sq.sqrt().cpu().contiguous(), n, f)
  table = torch.zeros(_lib.MAX_ROWS, dtype=torch.int32)
  table[:n - f] = torch.tensor(sel, dtype=torch.int32)
  return table.to(device)


def brute(gradients, f, check=False, **kwargs):
  """Brute rule (aggregators/brute.py:70-80): mean of the minimum-diameter subset, index order.  Distances, subset
  search and average all run on the device, on the caller's stream: no host synchronisation (graph-capturable).
  The search's status (device int32[1]) travels with the result as `result.brute_status`:
    -1  no subset of n-f rows has a finite diameter — more than f gradients with non-finite coordinates, where the
        reference fails its assertion (brute.py:56-57,68): the result is the average of n-f copies of one bad
        gradient, i.e. non-finite where that gradient is;
    -2  the device search gave up on its budget of search-tree nodes: the result is NaN EVERYWHERE (the index table
        then holds -1, which the averaging kernel answers with NaN) — never an average of some rows.
  check=True (what the `native-brute` plugin passes) reads the status, at the price of one host synchronisation,
  unless the stream is being captured into a graph: -1 raises like the reference, -2 falls back to the host search,
  which has no budget, and returns ITS average — the reference would have kept computing."""
  global last_brute_status
  n, d, device = _validate(gradients)
  sel, status = _brute_sel(gradients, f)
  last_brute_status = status
  if check and not torch.cuda.is_current_stream_capturing():
    code = int(status.item())
    if code == -2:
      sel = brute_host_selection(gradients, f)
      status = torch.zeros(1, dtype=torch.int32, device=device)
      _rank_cache_put("brute", gradients, (f,), (sel, status))
      last_brute_status = status
    elif code != 0:
      raise RuntimeError(BRUTE_NO_SUBSET)
  out = selected_mean(gradients, sel, n - f)
  out.brute_status = status
  return out


def aksel_sqdist(gradients):
  """fp64[n] squared distances of every row to the coordinate-wise median (device, no sync)."""
  n, d, device = _validate(gradients)
  lib = _lib.load()
  sq = torch.empty(_lib.MAX_ROWS, dtype=torch.float64, device=device)
  ws = _workspace(device, _lib.WS_AKSEL, n, d, "ws_aksel")
  with torch.cuda.device(device):
    _lib.check(lib.bm_aksel_pass1(_lib.pointer_table(gradients), n, d, None, _ptr(sq), _ptr(ws),
                                  _stream(device)), "bm_aksel_pass1")
  return sq


def stable_argsort(keys, n):
  """Device-side stable argsort of the first n fp64 keys (NaN last). Returns int32[MAX_ROWS]."""
  device = keys.device
  lib = _lib.load()
  order = torch.empty(_lib.MAX_ROWS, dtype=torch.int32, device=device)
  with torch.cuda.device(device):
    _lib.check(lib.bm_stable_argsort(_ptr(keys), n, _ptr(order), _stream(device)), "bm_stable_argsort")
  return order


def _aksel_order(gradients):
  order = _rank_cache_get("aksel", gradients, ())
  if order is None:
    order = stable_argsort(aksel_sqdist(gradients), len(gradients))
    _rank_cache_put("aksel", gradients, (), order)
  return order


def _aksel_count(n, f, mode):
  if mode == "mid":
    return (n + 1) // 2
  if mode == "n-f":
    return n - f
  raise NotImplementedError(f"aksel mode {mode!r}")


def aksel_selection(gradients, f, mode="mid", **kwargs):
  c = _aksel_count(len(gradients), f, mode)
  return _aksel_order(gradients)[:c].tolist()


def aksel(gradients, f, mode="mid", **kwargs):
  """Aksel (aggregators/aksel.py:52-64): mean of the c rows closest to the coordinate-wise median."""
  c = _aksel_count(len(gradients), f, mode)
  return selected_mean(gradients, _aksel_order(gradients), c)


def average(gradients, **kwargs):
  """Arithmetic mean (aggregators/average.py:21-29), same sequential summation order."""
  n, d, device = _validate(gradients)
  idx = torch.arange(n, dtype=torch.int32, device=device)
  return selected_mean(gradients, idx, n)


def cge_selection(gradients, f, **kwargs):
  """Device order of the gradients by increasing norm, non-finite last (aggregators/cge.py:28-38)."""
  from . import stats
  return stable_argsort(stats.row_sqnorms(gradients), len(gradients))


def cge(gradients, f, **kwargs):
  """Comparative gradient elimination (aggregators/cge.py:40-57)."""
  n = len(gradients)
  return selected_mean(gradients, cge_selection(gradients, f), n - f)
