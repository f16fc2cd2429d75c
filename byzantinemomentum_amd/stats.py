"""Per-step gradient statistics and momentum of the simulation loop, on the HIP path.

Mirrors tools/pytorch.py:97-125 (`compute_avg_dev_max`), the study block of attack.py:842-868
(norms, max-abs, six cosines, past-gradient cosine and curvature) and the worker-side momentum
update of attack.py:800-804.  One fused pass per stack and one fused pass for all the dot
products; results stay on the device until the caller asks for Python floats (one sync).
"""

import ctypes
import os
import math

import torch

from . import _lib
from . import gars

__all__ = ["compute_avg_dev_max", "stack_stats_async", "study_dots", "study_stats", "multi_axpby", "row_sqnorms", "momentum_stats", "momentum_stats_colwise", "momentum_stats_sqdist", "stack_stats_colwise", "stack_stats_sqdist",
           "multi_fma3", "clip_factors", "clip_factors_from_sq", "multi_scale", "clip_gradients", "l2_distance",
           "step_worker"]

_ptr = gars._ptr


def _attack_id(attack, direction):
  kind = _lib.ATTACK_LITTLE if attack == "little" else _lib.ATTACK_EMPIRE
  return kind | (_lib.ATTACK_DIRECTION if direction else 0)


def stack_stats_async(samples, scale=None, attack="empire", want_avg=True, direction=False):
  """avg vector + device tensor [sum avg^2, sum_i ||s_i-avg||^2, max|avg|] (fp64), no sync.

  With `scale`, also returns the Byzantine vector of an "identical" attack computed in the same pass
  (third element of the tuple): attack="empire" -> avg + scale*(-avg); attack="little" ->
  avg + scale*sqrt(unbiased column variance) (attacks/identical.py:63-86,129-141).
  direction=True: the third element is scale * (attack direction) alone, without the average
  (`grad_att` of identical.py:65, what the factor search combines with the average).
  """
  k, d, device = gars._validate(samples)
  lib = _lib.load()
  avg = torch.empty(d, dtype=torch.float32, device=device) if want_avg else None  # None: statistics only
  scaled = torch.empty(d, dtype=torch.float32, device=device) if scale is not None else None
  out3 = torch.empty(3, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STATS, k, d, "ws_stats")
  with torch.cuda.device(device):
    _lib.check(lib.bm_stack_stats(_lib.pointer_table(samples), k, d, _ptr(avg) if avg is not None else None,
                                  _ptr(scaled) if scaled is not None else None,
                                  ctypes.c_float(scale if scale is not None else 0.0),
                                  _attack_id(attack, direction), _ptr(out3),
                                  _ptr(ws), gars._stream(device)), "bm_stack_stats")
  if scale is not None:
    return avg, out3, scaled
  return avg, out3


def compute_avg_dev_max(samples):
  """Drop-in for tools.compute_avg_dev_max (tools/pytorch.py:97-125).

  Returns (average gradient or None, norm of the average, norm standard deviation, max |coord|).
  """
  if len(samples) == 0:
    return None, math.nan, math.nan, math.nan
  avg, out3 = stack_stats_async(samples)
  norm2, dev2, amax = out3.tolist()  # the only synchronisation
  k = len(samples)
  norm_dev = math.sqrt(dev2 / (k - 1)) if k >= 2 else math.nan
  return avg, math.sqrt(norm2), norm_dev, amax


def study_dots(core, extra=()):
  """Gram matrix (fp64, nc x nc) of up to 4 vectors and dot(core[0], e) for up to 32 more.

  Covers every `torch.dot`/`norm` of attack.py:851-868 in one pass over the vectors.
  Returns (gram tensor nc x nc, extras tensor ne), both on the device.
  """
  core = list(core)
  extra = list(extra)
  nc, d, device = gars._validate(core)
  if extra:
    gars._validate(extra + [core[0]])
  lib = _lib.load()
  out = torch.empty(nc * nc + len(extra), dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_DOT, 1, d, "ws_dot")
  with torch.cuda.device(device):
    _lib.check(lib.bm_multi_dot(_lib.pointer_table(core), nc,
                                _lib.pointer_table(extra) if extra else None, len(extra), d, _ptr(out),
                                _ptr(ws), gars._stream(device)), "bm_multi_dot")
  return out[:nc * nc].view(nc, nc), out[nc * nc:]


_EVAL_OPS = {"trmean": _lib.OP_TRMEAN, "phocas": _lib.OP_PHOCAS, "meamed": _lib.OP_MEAMED, "median": _lib.OP_MEDIAN}


def colwise_eval_supported(rule, n):
  """Does the evaluate-only form of the factor search exist for this rule and worker count (bm_colwise_eval)?"""
  return rule in _EVAL_OPS and bool(_lib.load().bm_colwise_eval_supported(_EVAL_OPS[rule], n))


def colwise_eval(rule, honests, copies, f, h_avg, direction, t):
  """| RULE(honests + [h_avg + t * direction] * copies) - h_avg |^2 as a device fp64[1] tensor, the candidate and the
  rule's output formed in registers only (bm_colwise_eval): one candidate of attacks/identical.py:67-77. No sync.
  t: a number, or the DEVICE float64 tensor a DeviceSearch left its candidate in (bm_colwise_eval_tdev)."""
  honests = list(honests)
  h, d, device = gars._validate(honests)
  gars._validate([h_avg, direction] + honests[:1])
  lib = _lib.load()
  out = torch.empty(1, dtype=torch.float64, device=device)
  ws = gars._Scratch.get(device, "ws_eval", nbytes=int(lib.bm_colwise_eval_workspace_bytes()))
  with torch.cuda.device(device):
    if isinstance(t, torch.Tensor):
      if not (t.is_cuda and t.device == device and t.dtype == torch.float64 and t.numel() >= 1 and t.is_contiguous()):
        raise gars.GarInputError("colwise_eval: a tensor t must be a contiguous float64 tensor on the vectors' device")
      _lib.check(lib.bm_colwise_eval_tdev(_EVAL_OPS[rule], _lib.pointer_table(honests), h, copies, d, f, _ptr(h_avg),
                                          _ptr(direction), _ptr(t), _ptr(out), _ptr(ws), gars._stream(device)),
                 "bm_colwise_eval_tdev")
    else:
      _lib.check(lib.bm_colwise_eval(_EVAL_OPS[rule], _lib.pointer_table(honests), h, copies, d, f, _ptr(h_avg),
                                     _ptr(direction), float(t), _ptr(out), _ptr(ws), gars._stream(device)),
                 "bm_colwise_eval")
  return out


def sqdist2(a, b):
  """|a - b|^2 as a device fp64[1] tensor in one pass over the two vectors (bm_sqdist2): the objective of a candidate of
  the factor search (identical.py:75-76).  No sync."""
  _, d, device = gars._validate([a, b])
  lib = _lib.load()
  out = torch.empty(1, dtype=torch.float64, device=device)
  ws = gars._Scratch.get(device, "ws_eval", nbytes=int(lib.bm_colwise_eval_workspace_bytes()))
  with torch.cuda.device(device):
    _lib.check(lib.bm_sqdist2(_ptr(a), _ptr(b), d, _ptr(out), _ptr(ws), gars._stream(device)), "bm_sqdist2")
  return out


def bulyan_pass2_eval_supported(n, f, m, d=0):
  """Is there an evaluate-only instance of Bulyan's second pass for this shape (bm_bulyan_pass2_eval)?"""
  return d <= (1 << 29) and bool(_lib.load().bm_bulyan_pass2_eval_supported(int(n), int(f), int(m)))


def bulyan_pass2_eval(honests, copies, order, f, m, h_avg, direction, t):
  """| pass2(honests + [h_avg + t * direction] * copies, order) - h_avg |^2 as a device fp64[1] tensor, the candidate in
  registers only (bm_bulyan_pass2_eval): one candidate of attacks/identical.py:67-77 against Bulyan, given the ranking of
  the stack (device int32 tensor; indices >= len(honests) name a copy of the candidate).  t: a number, or a DEVICE
  float64 tensor.  No sync."""
  honests = list(honests)
  h, d, device = gars._validate(honests)
  gars._validate([h_avg, direction] + honests[:1])
  lib = _lib.load()
  out = torch.empty(1, dtype=torch.float64, device=device)
  ws = gars._Scratch.get(device, "ws_eval", nbytes=int(lib.bm_colwise_eval_workspace_bytes()))
  t_dev = None
  if isinstance(t, torch.Tensor):
    if not (t.is_cuda and t.device == device and t.dtype == torch.float64 and t.numel() >= 1 and t.is_contiguous()):
      raise gars.GarInputError("bulyan_pass2_eval: a tensor t must be a contiguous float64 tensor on the vectors' device")
    t_dev, t = t, 0.0
  with torch.cuda.device(device):
    _lib.check(lib.bm_bulyan_pass2_eval(_lib.pointer_table(honests), h, int(copies), _ptr(order), int(f), int(m), d,
                                        _ptr(h_avg), _ptr(direction), ctypes.c_float(float(t)),
                                        _ptr(t_dev) if t_dev is not None else None, _ptr(out), _ptr(ws),
                                        gars._stream(device)), "bm_bulyan_pass2_eval")
  return out


def order_pair_supported(h):
  """Is there a one-pass form of two order statistics of h rows (bm_order_pair)?"""
  return bool(_lib.load().bm_order_pair_supported(int(h)))


def order_pair(rows, il, ih):
  """(lo, hi): per coordinate the values of rank il and ih (0-based, ascending) among the rows; -inf below rank 0, +inf
  beyond the last rank, NaN where the column holds one (bm_order_pair).  One pass over the rows, no sync: the two
  vectors of the median's factor search (median.py:31-39 on the rows with k copies of -inf / +inf)."""
  rows = list(rows)
  h, d, device = gars._validate(rows)
  lib = _lib.load()
  lo, hi = torch.empty_like(rows[0]), torch.empty_like(rows[0])
  with torch.cuda.device(device):
    _lib.check(lib.bm_order_pair(_lib.pointer_table(rows), h, d, int(il), int(ih), _ptr(lo), _ptr(hi),
                                 gars._stream(device)), "bm_order_pair")
  return lo, hi


class DeviceSearch:
  """The cursor of tools.line_maximize (tools/misc.py:468-514) kept in device memory (bm_search_device_next): `next(y)`
  takes the objective the last evaluation left on the device (None the first time) and returns the DEVICE float64[1]
  tensor holding the next candidate's signed factor — hand it to multi_fma3 / colwise_eval as b / t —, `finish(y)`
  returns the float64 device tensor [factor, x0, y0, x1, y1, ...] of attack_search_device.  Nothing is awaited: the
  host queues the whole search."""

  def __init__(self, device, evals, negative=False, start=0., delta=1., ratio=0.8):
    if not isinstance(evals, int) or evals < 1:
      _lib.check(_lib.EINVAL, "DeviceSearch (evals must be a positive integer)")
    self.device, self.evals, self.negative, self.shape = device, evals, bool(negative), (start, delta, ratio)
    self.state = torch.zeros(8, dtype=torch.float64, device=device)   # sizeof(bm_search) = 56 bytes
    self.t = torch.zeros(1, dtype=torch.float64, device=device)
    self.out = torch.zeros(1 + 2 * evals, dtype=torch.float64, device=device)
    self.proposed = 0
    self._keep = None

  def _call(self, y, last):
    if y is not None and not (isinstance(y, torch.Tensor) and y.is_cuda and y.device == self.device
                              and y.dtype == torch.float64 and y.numel() >= 1):
      raise gars.GarInputError("DeviceSearch: the objective must be a float64 tensor on the search's device")
    lib = _lib.load()
    self._keep = y  # (alive until the kernel that reads it has been queued behind its producer)
    with torch.cuda.device(self.device):
      _lib.check(lib.bm_search_device_next(_ptr(self.state), _ptr(y) if y is not None else None, 1 if self.negative else 0,
                                           1 if last else 0, *self.shape, _ptr(self.t) if not last else None,
                                           _ptr(self.out), gars._stream(self.device)), "bm_search_device_next")

  def next(self, y=None):
    if self.proposed >= self.evals or (y is None) != (self.proposed == 0):
      raise RuntimeError("DeviceSearch.next: one objective per candidate, `evals` candidates")
    self._call(y, False)
    self.proposed += 1
    return self.t

  def finish(self, y):
    if self.proposed != self.evals:
      raise RuntimeError("DeviceSearch.finish: the search has candidates left")
    self._call(y, True)
    return self.out


def study_stats(s_avg, h_avg, defense, byz, f_real, past_newest=None, curv=None, past_oldest=None, curv_mode=0, mu=0.0,
                oldest_weight=0.0, params=None, origin=None, attack_avg_out=None, update_momentum=None, update_mu=0.0,
                update_omd=0.0):
  """The study block of a step in ONE pass (bm_study_stats, include/bm_gar.h): statistics of the attack stack
  (f_real copies of `byz`) and of the defense vector, the Gram matrix of (sampled avg, honest avg, defense, attack
  avg), <s, past_newest>, <s, C>, |params - origin|^2, and the in-place update of the curvature combination C =
  `curv` for the next step (curv_mode: 0 none, 1 C <- s, 2 C <- s + mu C, 3 the same after taking
  oldest_weight * past_oldest out of C).  update_momentum: the momentum of the update (attack.py:836-838), updated in
  place in the same pass: M <- update_omd * defense + update_mu * M (bm_study_stats_update).
  Returns the device fp64 vector of STUDY_SLOTS statistics; no sync."""
  vecs = [t for t in (s_avg, h_avg, defense, byz if f_real > 0 else None, past_newest if curv_mode >= 2 else None,
                      curv if curv_mode >= 1 else None, past_oldest if curv_mode == 3 else None, params, origin,
                      attack_avg_out, update_momentum) if t is not None]
  _, d, device = gars._validate(vecs)
  if (params is None) != (origin is None):
    raise gars.GarInputError("study_stats needs both params and origin, or neither")
  lib = _lib.load()
  out = torch.empty(_lib.STUDY_SLOTS, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STUDY, 1, d, "ws_study")
  opt = lambda t: _ptr(t) if t is not None else None  # noqa: E731
  with torch.cuda.device(device):
    _lib.check(lib.bm_study_stats_update(_ptr(s_avg), _ptr(h_avg), _ptr(defense), opt(byz if f_real > 0 else None),
                                         int(f_real), opt(attack_avg_out), opt(past_newest), opt(curv), opt(past_oldest),
                                         int(curv_mode), ctypes.c_float(mu), ctypes.c_float(oldest_weight), opt(params),
                                         opt(origin), opt(update_momentum), ctypes.c_float(update_mu),
                                         ctypes.c_float(update_omd), d, _ptr(out), _ptr(ws), gars._stream(device)),
               "bm_study_stats_update")
  if update_momentum is not None:
    gars.invalidate_rank_cache()  # (a user tensor was written)
  return out


def row_sqnorms(gradients):
  """Squared L2 norm of every gradient (fp64, on the device): one read of the stack, one call (bm_row_sqnorms)."""
  n, d, device = gars._validate(gradients)
  lib = _lib.load()
  res = torch.empty(_lib.MAX_ROWS, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_DOT, 1, d, "ws_dot")
  with torch.cuda.device(device):
    _lib.check(lib.bm_row_sqnorms(_lib.pointer_table(gradients), n, d, _ptr(res), _ptr(ws), gars._stream(device)),
               "bm_row_sqnorms")
  return res[:n]


def multi_axpby(ys, xs, a, b):
  """In place y_i <- a*y_i + b*x_i for every pair: `gmtm.mul_(mu).add_(grad, alpha=1-damp)`."""
  k, d, device = gars._validate(list(ys))
  gars._validate(list(xs) + [ys[0]])
  if len(xs) != k:
    raise gars.GarInputError("multi_axpby needs as many x as y vectors")
  lib = _lib.load()
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_multi_axpby(_lib.pointer_table(ys), _lib.pointer_table(xs), k, d,
                                  ctypes.c_float(a), ctypes.c_float(b), gars._stream(device)),
               "bm_multi_axpby")


def momentum_stats(sampled, buffers, mu, one_minus_damp, clip_factors_dev=None, attack_scale=None, attack="empire",
                   direction=False):
  """First pass of a step in one kernel (attack.py:791-804,846-847 and attacks/identical.py:63-86):
  worker momentum in place on `buffers` (which then ARE the honest gradients), statistics of the
  sampled stack and of the honest stack, and the Byzantine vector of an "identical" attack.

  Returns (sampled_avg, honest_avg, byz or None, out6) with out6 a device fp64 tensor
  [sum avg_s^2, sum_i |s_i-avg_s|^2, max|avg_s|, sum avg_h^2, sum_i |b_i-avg_h|^2, max|avg_h|]. No sync.
  """
  ks, d, device = gars._validate(list(sampled))
  h, _, _ = gars._validate(list(buffers) + [sampled[0]])
  h -= 1
  if h < 1 or ks < h:
    raise gars.GarInputError("momentum_stats needs 1 <= len(buffers) <= len(sampled)")
  lib = _lib.load()
  s_avg = torch.empty(d, dtype=torch.float32, device=device)
  h_avg = torch.empty(d, dtype=torch.float32, device=device)
  byz = torch.empty(d, dtype=torch.float32, device=device) if attack_scale is not None else None
  out6 = torch.empty(6, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STEP, 1, d, "ws_step")
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_momentum_stats(
      _lib.pointer_table(sampled), ks, _lib.pointer_table(buffers), h, d, ctypes.c_float(mu),
      ctypes.c_float(one_minus_damp), _ptr(clip_factors_dev) if clip_factors_dev is not None else None,
      _ptr(s_avg), _ptr(h_avg), _ptr(byz) if byz is not None else None,
      ctypes.c_float(attack_scale if attack_scale is not None else 0.0),
      _attack_id(attack, direction), _ptr(out6), _ptr(ws),
      gars._stream(device)), "bm_momentum_stats")
  return s_avg, h_avg, byz, out6


_COLWISE_OPS = {"median": _lib.OP_MEDIAN, "trmean": _lib.OP_TRMEAN, "phocas": _lib.OP_PHOCAS, "meamed": _lib.OP_MEAMED}


def momentum_stats_colwise(sampled, buffers, mu, one_minus_damp, clip_factors_dev, attack_scale, attack, rule, f, n_byz):
  """momentum_stats followed by a coordinate-wise rule over the updated buffers and `n_byz` copies of the Byzantine
  vector — for the median / trimmed mean over 20 buffers and 1..6 copies INSIDE the same kernel
  (bm_momentum_stats_colwise).  Returns (sampled_avg, honest_avg, byz, defense, out6). No sync."""
  ks, d, device = gars._validate(list(sampled))
  h = gars._validate(list(buffers) + [sampled[0]])[0] - 1
  if h < 1 or ks < h or n_byz < 1:
    raise gars.GarInputError("momentum_stats_colwise needs 1 <= len(buffers) <= len(sampled) and n_byz >= 1")
  lib = _lib.load()
  s_avg, h_avg, byz, defense = (torch.empty(d, dtype=torch.float32, device=device) for _ in range(4))
  out6 = torch.empty(6, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STEP, 1, d, "ws_step")
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_momentum_stats_colwise(
      _lib.pointer_table(sampled), ks, _lib.pointer_table(buffers), h, d, ctypes.c_float(mu),
      ctypes.c_float(one_minus_damp), _ptr(clip_factors_dev) if clip_factors_dev is not None else None,
      _ptr(s_avg), _ptr(h_avg), _ptr(byz), ctypes.c_float(attack_scale), _attack_id(attack, False),
      _COLWISE_OPS[rule], int(f), int(n_byz), _ptr(defense), _ptr(out6), _ptr(ws), gars._stream(device)),
      "bm_momentum_stats_colwise")
  return s_avg, h_avg, byz, defense, out6


def momentum_stats_sqdist(sampled, buffers, mu, one_minus_damp, clip_factors_dev, attack_scale, attack, n_byz,
                          d_total=None):
  """momentum_stats together with the n x n squared distances (n = len(buffers) + n_byz) of the updated buffers and
  the Byzantine copies — contracted inside the same kernel at ks = h = 20 for long gradients
  (bm_momentum_stats_sqdist).  Returns (sampled_avg, honest_avg, byz, sq, out6). No sync."""
  ks, d, device = gars._validate(list(sampled))
  h = gars._validate(list(buffers) + [sampled[0]])[0] - 1
  if h < 1 or ks < h or n_byz < 1 or h + n_byz > _lib.MAX_ROWS:
    raise gars.GarInputError("momentum_stats_sqdist needs 1 <= len(buffers) <= len(sampled), n_byz >= 1, at most 64 rows")
  n = h + n_byz
  lib = _lib.load()
  s_avg, h_avg, byz = (torch.empty(d, dtype=torch.float32, device=device) for _ in range(3))
  sq = torch.empty((n, n), dtype=torch.float64, device=device)
  out6 = torch.empty(6, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STEP, 1, d, "ws_step")
  ws_pair = gars._workspace(device, _lib.WS_PAIRWISE, n, d, "ws_pair")
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_momentum_stats_sqdist(
      _lib.pointer_table(sampled), ks, _lib.pointer_table(buffers), h, d, d if d_total is None else int(d_total),
      ctypes.c_float(mu), ctypes.c_float(one_minus_damp),
      _ptr(clip_factors_dev) if clip_factors_dev is not None else None, _ptr(s_avg), _ptr(h_avg), _ptr(byz),
      ctypes.c_float(attack_scale), _attack_id(attack, False), int(n_byz), _ptr(sq), _ptr(out6), _ptr(ws), _ptr(ws_pair),
      gars._stream(device)), "bm_momentum_stats_sqdist")
  return s_avg, h_avg, byz, sq, out6


def stack_stats_colwise(rows, attack_scale, attack, rule, f, n_byz):
  """stack_stats (average, Byzantine vector, statistics) followed by a coordinate-wise rule over the rows and `n_byz`
  copies of the Byzantine vector — ONE pass over the rows at k = 20 (1..6 copies) or 14 (11 copies)
  (bm_stack_stats_colwise: the honest rows of `--momentum-at update`).  Returns (avg, byz, defense, out6). No sync."""
  k, d, device = gars._validate(list(rows))
  if n_byz < 1:
    raise gars.GarInputError("stack_stats_colwise needs n_byz >= 1")
  lib = _lib.load()
  avg, byz, defense = (torch.empty(d, dtype=torch.float32, device=device) for _ in range(3))
  out6 = torch.empty(6, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STEP, 1, d, "ws_step")
  with torch.cuda.device(device):
    _lib.check(lib.bm_stack_stats_colwise(_lib.pointer_table(rows), k, d, _ptr(avg), _ptr(byz),
                                          ctypes.c_float(attack_scale), _attack_id(attack, False), _COLWISE_OPS[rule],
                                          int(f), int(n_byz), _ptr(defense), _ptr(out6), _ptr(ws),
                                          gars._stream(device)), "bm_stack_stats_colwise")
  return avg, byz, defense, out6


def stack_stats_sqdist(rows, attack_scale, attack, n_byz, d_total=None):
  """stack_stats together with the n x n squared distances (n = len(rows) + n_byz) of the rows and the Byzantine
  copies, in one pass at k = 20 / 14 for long gradients (bm_stack_stats_sqdist).  Returns (avg, byz, sq, out6)."""
  k, d, device = gars._validate(list(rows))
  if n_byz < 1 or k + n_byz > _lib.MAX_ROWS:
    raise gars.GarInputError("stack_stats_sqdist needs n_byz >= 1 and at most 64 rows")
  n = k + n_byz
  lib = _lib.load()
  avg, byz = (torch.empty(d, dtype=torch.float32, device=device) for _ in range(2))
  sq = torch.empty((n, n), dtype=torch.float64, device=device)
  out6 = torch.empty(6, dtype=torch.float64, device=device)
  ws = gars._workspace(device, _lib.WS_STEP, 1, d, "ws_step")
  ws_pair = gars._workspace(device, _lib.WS_PAIRWISE, n, d, "ws_pair")
  with torch.cuda.device(device):
    _lib.check(lib.bm_stack_stats_sqdist(_lib.pointer_table(rows), k, d, d if d_total is None else int(d_total),
                                         _ptr(avg), _ptr(byz), ctypes.c_float(attack_scale), _attack_id(attack, False),
                                         int(n_byz), _ptr(sq), _ptr(out6), _ptr(ws), _ptr(ws_pair),
                                         gars._stream(device)), "bm_stack_stats_sqdist")
  return avg, byz, sq, out6


def multi_fma3(outs, ps, qs, a, b, p_scale_dev=None):
  """out_i = b*q_i + a*(p_scale_i*p_i) for every triple (outs may alias ps; qs may repeat one tensor):
  worker / server / update momentum and the Nesterov look-ahead of attack.py:757-810,832-839.
  b: a number, or a DEVICE float64 tensor whose first element is read by the kernel (bm_multi_fma3_bdev: the factor
  attack_search_device left on the device — no host round trip)."""
  outs, ps, qs = list(outs), list(ps), list(qs)
  k, d, device = gars._validate(outs)
  gars._validate(ps + [outs[0]])
  gars._validate(qs + [outs[0]])
  if len(ps) != k or len(qs) != k:
    raise gars.GarInputError("multi_fma3 needs as many p and q as out vectors")
  lib = _lib.load()
  gars.invalidate_rank_cache()
  scale = _ptr(p_scale_dev) if p_scale_dev is not None else None
  with torch.cuda.device(device):
    if isinstance(b, torch.Tensor):
      if not (b.is_cuda and b.device == device and b.dtype == torch.float64 and b.numel() >= 1 and b.is_contiguous()):
        raise gars.GarInputError("multi_fma3: a tensor b must be a contiguous float64 tensor on the vectors' device")
      _lib.check(lib.bm_multi_fma3_bdev(_lib.pointer_table(outs), _lib.pointer_table(ps), _lib.pointer_table(qs), k, d,
                                        ctypes.c_float(a), _ptr(b), scale, gars._stream(device)), "bm_multi_fma3_bdev")
    else:
      _lib.check(lib.bm_multi_fma3(_lib.pointer_table(outs), _lib.pointer_table(ps), _lib.pointer_table(qs), k, d,
                                   ctypes.c_float(a), ctypes.c_float(b), scale, gars._stream(device)), "bm_multi_fma3")


DEVICE_SEARCH_RULES = ("krum", "average")


def attack_search_device(ext, h, k, f, rule, evals=16, negative=False, m=None):
  """The factor search of attacks/identical.py:67-77 on the device (bm_attack_line_search_device) from the DEVICE
  (h+2) x (h+2) squared distances among honests + [avg, avg + att]: a float64 device tensor of 1 + 2 * evals values —
  [0] the factor, then (abscissa, objective) per evaluation — with the candidates and the bits of
  linesearch.attack_line_search.  Nothing is copied or awaited: hand `out[:1]` to multi_fma3 as b."""
  if rule not in DEVICE_SEARCH_RULES:
    raise ValueError(f"no device form of the search for rule {rule!r}")
  if not (isinstance(ext, torch.Tensor) and ext.is_cuda and ext.dtype == torch.float64 and ext.is_contiguous()
          and tuple(ext.shape) == (h + 2, h + 2)):
    raise gars.GarInputError(f"ext must be a contiguous float64 device tensor of shape ({h + 2}, {h + 2})")
  if not isinstance(evals, int) or evals < 1:
    _lib.check(_lib.EINVAL, "attack_search_device (evals must be a positive integer)")
  lib = _lib.load()
  # (BM_SEARCH_TRACE=1, measurement only: the kernel appends 24 phase timestamps per candidate behind the results)
  extra = 24 * evals if os.environ.get("BM_SEARCH_TRACE", "") == "1" else 0
  out = torch.zeros(1 + 2 * evals + extra, dtype=torch.float64, device=ext.device) if extra else \
      torch.empty(1 + 2 * evals, dtype=torch.float64, device=ext.device)
  with torch.cuda.device(ext.device):
    _lib.check(lib.bm_attack_line_search_device(_ptr(ext), h, k, f, _lib.RULE_IDS[rule], m or 0, evals,
                                                1 if negative else 0, _ptr(out), gars._stream(ext.device)),
               "bm_attack_line_search_device")
  return out


def attack_ranking_device(ext, h, k, f, mode, t_dev, m=None):
  """linesearch.attack_ranking on the device (bm_attack_ranking_device): the ranking of honests + [avg + t*att] * k as a
  device int32[64] tensor (the n rows by rank, then zeros) from the DEVICE (h+2) x (h+2) matrix and a factor in DEVICE
  memory (float64[1]) — what bm_bulyan_pass2 / bm_bulyan_pass2_eval take as `order`.  Nothing is copied or awaited."""
  if not (isinstance(ext, torch.Tensor) and ext.is_cuda and ext.dtype == torch.float64 and ext.is_contiguous()
          and tuple(ext.shape) == (h + 2, h + 2)):
    raise gars.GarInputError(f"ext must be a contiguous float64 device tensor of shape ({h + 2}, {h + 2})")
  if not (isinstance(t_dev, torch.Tensor) and t_dev.is_cuda and t_dev.device == ext.device and t_dev.dtype == torch.float64
          and t_dev.numel() >= 1 and t_dev.is_contiguous()):
    raise gars.GarInputError("attack_ranking_device: the factor must be a contiguous float64 tensor on the matrix's device")
  lib = _lib.load()
  order = torch.empty(_lib.MAX_ROWS, dtype=torch.int32, device=ext.device)
  mode_id = {"krum": _lib.RANK_KRUM, "bulyan": _lib.RANK_BULYAN}[mode]
  with torch.cuda.device(ext.device):
    _lib.check(lib.bm_attack_ranking_device(_ptr(ext), h, k, f, mode_id, m or 0, _ptr(t_dev), _ptr(order),
                                            gars._stream(ext.device)), "bm_attack_ranking_device")
  return order


def clip_factors_from_sq(sq, k, clip):
  """Device float32 factors from a device fp64 tensor of k squared norms (possibly all-reduced)."""
  lib = _lib.load()
  device = sq.device
  if not sq.is_contiguous():
    sq = sq.contiguous()
  factors = torch.empty(_lib.MAX_ROWS, dtype=torch.float32, device=device)
  with torch.cuda.device(device):
    _lib.check(lib.bm_clip_factors(_ptr(sq), k, ctypes.c_float(clip), _ptr(factors), gars._stream(device)),
               "bm_clip_factors")
  return factors


def clip_factors(gradients, clip):
  """Device float32[k]: clip/||g_i|| where ||g_i|| > clip, else 1 (attack.py:791-794). No sync."""
  k, d, device = gars._validate(gradients)
  return clip_factors_from_sq(row_sqnorms(gradients), k, clip)


def multi_scale(ys, factors_dev):
  """y_i *= factors[i] in place; rows whose factor is exactly 1 are not touched."""
  ys = list(ys)
  k, d, device = gars._validate(ys)
  lib = _lib.load()
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_multi_scale(_lib.pointer_table(ys), k, d, _ptr(factors_dev), gars._stream(device)),
               "bm_multi_scale")


def clip_gradients(gradients, clip):
  """In-place gradient clipping of attack.py:776-779,791-794 for a whole list, without a host sync."""
  factors = clip_factors(gradients, clip)
  multi_scale(gradients, factors)
  return factors


def l2_distance(a, b):
  """||a - b||_2 as a device fp64 scalar tensor (attack.py:830 `l2_origin`), no sync.  Two rows through
  the centred pairwise kernel: the centre is row 0, so the contraction sees a - b itself (no cancellation)."""
  return gars.pairwise_sqdist([a, b])[0, 1].sqrt()


def step_worker(comm, sampled, buffers, n, f_decl, f_real, rule, m, mu, one_minus_damp, clip, attack, attack_scale,
                nb_past, past_count, past_newest, curv, past_oldest, params=None, origin=None, d_total=None):
  """One simulation step with worker-side momentum as ONE C call (bm_step_worker, include/bm_gar.h).

  comm: sharded.NativeComm or None; d_total: length of the whole vectors over all shards (default: this shard's).
  Returns (defense, sampled_avg, honest_avg, byz, stats) with `stats` the
  device fp64 vector documented in the header (already reduced over the ranks). No sync."""
  ks, d, device = gars._validate(list(sampled))
  lib = _lib.load()
  new = lambda: torch.empty(d, dtype=torch.float32, device=device)  # noqa: E731
  defense, s_avg, h_avg = new(), new(), new()
  byz = new() if f_real > 0 else None  # (the attack average is not materialised: attack_avg_out = NULL)
  stats = torch.empty(lib.bm_step_stats_count(), dtype=torch.float64, device=device)
  ws = gars._Scratch.get(device, "ws_stepcall", nbytes=int(lib.bm_step_workspace_bytes(n, d)))
  par = _lib.StepParams(n=n, f_decl=f_decl, f_real=f_real, ks=ks, rule=_lib.RULE_IDS[rule], m=m or 0,
                        attack_kind=_lib.ATTACK_LITTLE if attack == "little" else _lib.ATTACK_EMPIRE, nb_past=nb_past,
                        past_count=past_count, attack_scale=attack_scale, mu=mu, one_minus_damp=one_minus_damp,
                        clip=clip if clip is not None else 0.0,
                        oldest_weight=-(mu ** (nb_past - 1)) if nb_past > 0 else 0.0)
  opt = lambda t: _ptr(t) if t is not None else None  # noqa: E731
  gars.invalidate_rank_cache()
  with torch.cuda.device(device):
    _lib.check(lib.bm_step_worker(comm.handle if comm is not None else None, ctypes.byref(par),
                                  _lib.pointer_table(sampled), _lib.pointer_table(buffers), d,
                                  int(d_total) if d_total is not None else d, _ptr(defense),
                                  _ptr(s_avg), _ptr(h_avg), opt(byz), None, opt(past_newest), opt(curv),
                                  opt(past_oldest), opt(params), opt(origin), _ptr(stats), _ptr(ws),
                                  gars._stream(device)), "bm_step_worker")
  return defense, s_avg, h_avg, byz, stats
