"""CPU oracle for the gradient-aggregation hot path.  TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may import this module,
and only as the checker / the timed CPU baseline — never as part of the product path
(byzantinemomentum_amd/ and native/ do not import it and raise when the HIP library is absent).

It restates, in independent code, the algorithms of LPD-EPFL/ByzantineMomentum's hot path:

  pairwise distances + Krum scores   aggregators/krum.py:41-63
  Multi-Krum average                 aggregators/krum.py:76-80
  Bulyan (static scores, see below)  aggregators/bulyan.py:41-84
  median / trmean / closest          aggregators/median.py:39, trmean.py:33,45-50
  brute subset search                aggregators/brute.py:41-68,80
  aksel                              aggregators/aksel.py:35-50,64
  average / cge                      aggregators/average.py:29, cge.py:28-57
  compute_avg_dev_max                tools/pytorch.py:105-125
  study block / momentum             attack.py:800-810,830,851-868

Two arithmetic modes:
  "f32"  every tensor operation is the same PyTorch-CPU fp32 operation the reference issues, in the
         same order, so results are BIT-IDENTICAL to the reference (pinned by tests/test_oracle_vs_
         reference.py against the real reference in the build container, and by the committed
         fixtures under tests/golden/ everywhere else).  This is also the "port" timed as the CPU
         baseline in bench.py.
  "f64"  same selection logic, all reductions in float64: the ground truth that norm-type
         quantities are compared with at large d, where the reference's own fp32 `norm()` is off by
         up to 2e-3.

Pinning status: the reference ships NO golden vectors or tests for this path, so the
pin is "outputs of the reference itself, run here": see tests/golden/README.md.

Known reference quirk kept on purpose: Bulyan's score-update branch (bulyan.py:74-76) is dead code
(its guard compares against the entry that was just overwritten), so the scores never change and
iteration i averages ranks [i, i+min(m, m_max-i)) of the INITIAL stable order.
"""

import itertools
import math

import numpy as np
import torch

__all__ = ["pairwise_distances", "krum_scores", "krum_order", "krum", "bulyan_order", "bulyan", "median",
           "trmean", "phocas", "meamed", "closest_window", "brute_selection", "brute_selection_from_distances",
           "brute_selection_is_the_references", "brute", "aksel_order", "aksel",
           "average", "cge", "compute_avg_dev_max", "study_block", "worker_momentum", "make_stack"]


# ---------------------------------------------------------------------------- #
# Helpers

def _as64(t):
  return t.detach().to("cpu", torch.float64)


def _seq_sum_div(tensors, count):
  """`sum(iterable).div_(count)`: Python's sum starts from int 0, then adds left to right."""
  acc = 0
  for t in tensors:
    acc = acc + t
  return acc.div_(count)


# ---------------------------------------------------------------------------- #
# Distances and scores

def pairwise_distances(gradients, precision="f32", clamp_nonfinite=True):
  """Symmetric n x n numpy float64 matrix of L2 distances (diagonal 0).

  f32: `gradients[x].sub(gradients[y]).norm().item()` for x < y (krum.py:44-45).
  f64: sqrt of the float64 sum of squared differences.
  Non-finite distances become +inf unless clamp_nonfinite is False (krum.py:46-47 vs brute.py:45).
  """
  n = len(gradients)
  dist = np.zeros((n, n), dtype=np.float64)
  g64 = [_as64(g) for g in gradients] if precision == "f64" else None
  for x in range(n - 1):
    for y in range(x + 1, n):
      if gradients[x] is gradients[y]:
        val = 0.0 if torch.isfinite(gradients[x]).all() else math.nan
      elif precision == "f64":
        diff = g64[x] - g64[y]
        val = math.sqrt(torch.dot(diff, diff).item())
      else:
        val = gradients[x].sub(gradients[y]).norm().item()
      if clamp_nonfinite and not math.isfinite(val):
        val = math.inf
      dist[x, y] = dist[y, x] = val
  return dist


def _sum_smallest(row_values, count):
  """float64 sum, in ascending order, of the `count` smallest entries (Python `sum` of a sorted list)."""
  total = 0
  for v in sorted(row_values)[:count]:
    total = total + v
  return total


def krum_scores(dist, f):
  """score_i = sum of the n-f-1 smallest distances of row i to the others (krum.py:50-60)."""
  n = dist.shape[0]
  return [_sum_smallest([dist[i, j] for j in range(n) if j != i], n - f - 1) for i in range(n)]


def _stable_order(scores):
  return sorted(range(len(scores)), key=lambda i: scores[i])  # Python's sort is stable (krum.py:62)


def krum_order(gradients, f, precision="f32"):
  dist = pairwise_distances(gradients, precision)
  scores = krum_scores(dist, f)
  return _stable_order(scores), scores


def krum(gradients, f, m=None, precision="f32"):
  n = len(gradients)
  if m is None:
    m = n - f - 2
  order, _ = krum_order(gradients, f, precision)
  rows = [gradients[i] if precision == "f32" else _as64(gradients[i]) for i in order[:m]]
  return _seq_sum_div(rows, m)


def bulyan_order(gradients, f, m=None, precision="f32"):
  """Initial Bulyan ranking: score_i = sum of the m smallest of row i's distances, the +inf diagonal
  being part of the candidate pool (bulyan.py:48-62)."""
  n = len(gradients)
  if m is None:
    m = n - f - 2
  dist = pairwise_distances(gradients, precision)
  scores = []
  for i in range(n):
    pool = [dist[i, j] if j != i else math.inf for j in range(n)]
    scores.append(_sum_smallest(pool, m))
  return _stable_order(scores), scores


def bulyan(gradients, f, m=None, precision="f32"):
  n = len(gradients)
  m_max = n - f - 2
  if m is None:
    m = m_max
  theta = n - 2 * f - 2
  beta = theta - 2 * f
  order, _ = bulyan_order(gradients, f, m, precision)
  cast = (lambda t: t) if precision == "f32" else _as64
  picked = []
  for i in range(theta):
    count = min(m, m_max - i)  # `m = min(m, m_max - i)` is monotone, so this equals the running value
    picked.append(_seq_sum_div([cast(gradients[g]) for g in order[i:i + count]], count))
  selected = torch.stack(picked)
  centre = selected.median(dim=0).values
  return _closest_like_reference(selected, beta, centre)


# ---------------------------------------------------------------------------- #
# Coordinate-wise rules

def _stack(gradients, precision):
  g = torch.stack([x.detach().cpu() for x in gradients])
  return g if precision == "f32" else g.to(torch.float64)


def median(gradients, precision="f32"):
  return _stack(gradients, precision).median(dim=0).values


def _trimmed(g, f):
  return g.sort(dim=0).values[f:g.shape[0] - f].mean(dim=0)


def trmean(gradients, f, precision="f32"):
  return _trimmed(_stack(gradients, precision), f)


def _closest_like_reference(g, keep, centre):
  """Mean of the `keep` values of each column of g nearest `centre`, through the same torch calls
  as trmean.py:45-50 / bulyan.py:80-82 (topk of |g-c|, unsorted), gathered instead of take()n."""
  idx = g.clone().sub_(centre).abs_().topk(keep, dim=0, largest=False, sorted=False).indices
  return g.gather(0, idx).mean(dim=0)


def phocas(gradients, f, precision="f32"):
  g = _stack(gradients, precision)
  return _closest_like_reference(g, g.shape[0] - f, _trimmed(g, f))


def meamed(gradients, f, precision="f32"):
  g = _stack(gradients, precision)
  return _closest_like_reference(g, g.shape[0] - f, g.median(dim=0).values)


def closest_window(g, keep, centre):
  """Independent float64 formulation of `closest`: in sorted order the `keep` nearest values form a
  window; returns (mean, ambiguous) where ambiguous[j] is True when an excluded value is exactly as
  far from the centre as an included one (the reference's topk may then keep either)."""
  g = g.to(torch.float64)
  centre = centre.to(torch.float64)
  n, d = g.shape
  srt = g.sort(dim=0).values
  dev = (srt - centre).abs()
  lo = torch.zeros(d, dtype=torch.long)
  hi = torch.full((d,), n - 1, dtype=torch.long)
  cols = torch.arange(d)
  for _ in range(n - keep):
    drop_lo = dev[lo, cols] > dev[hi, cols]
    lo = torch.where(drop_lo, lo + 1, lo)
    hi = torch.where(drop_lo, hi, hi - 1)
  csum = torch.cat([torch.zeros(1, d, dtype=torch.float64), srt.cumsum(dim=0)])
  # exact window sum by direct accumulation (cumsum differences would cancel)
  total = torch.zeros(d, dtype=torch.float64)
  for k in range(keep):
    total += srt[lo + k, cols]
  del csum
  amb = torch.zeros(d, dtype=torch.bool)
  for outside, valid in (((lo - 1).clamp(min=0), lo > 0), ((hi + 1).clamp(max=n - 1), hi < n - 1)):
    for inside in (lo, hi):
      # an excluded value exactly as far as an included one, but a different number
      amb |= valid & (dev[outside, cols] == dev[inside, cols]) & (srt[outside, cols] != srt[inside, cols])
  return total / keep, amb


# ---------------------------------------------------------------------------- #
# Brute, Aksel, average, CGE

def brute_selection(gradients, f, precision="f32"):
  """First subset (lexicographic order) of n-f rows with the strictly smallest diameter; subsets
  touching a non-finite distance are skipped (brute.py:47-68)."""
  dist = pairwise_distances(gradients, precision, clamp_nonfinite=False)
  best = brute_selection_from_distances(dist, f)
  if best is None:
    raise AssertionError("too many non-finite gradients")
  return best


def brute_selection_from_distances(dist, f):
  """The loop of brute.py:47-68 on an n x n distance matrix (entries [x, y], x < y): running maximum from 0.,
  a subset is dropped at its first non-finite distance, first subset of strictly smallest diameter."""
  n = dist.shape[0]
  best, best_diam = None, None
  for subset in itertools.combinations(range(n), n - f):
    diam, finite = 0., True
    for x, y in itertools.combinations(subset, 2):
      v = dist[x, y]
      if not math.isfinite(v):
        finite = False
        break
      if v > diam:
        diam = v
    if finite and (best is None or diam < best_diam):
      best, best_diam = list(subset), diam
  return best


def brute_selection_is_the_references(dist, f, selection):
  """Would brute.py:47-68 return `selection` on this distance matrix?  Decided WITHOUT enumerating the C(n, f)
  subsets (1.6e11 at n = 51, f = 12), so that answers at such shapes can be checked at all: with D the diameter of
  `selection`,
    (1) every distance inside it is finite;
    (2) no subset of n-f rows has all its distances finite and < D   (strictly smallest);
    (3) for every position i and every row c below selection[i] (above selection[i-1]) no subset of finite diameter
        <= D starts with selection[:i] + [c]                          (first in lexicographic order).
  (2) and (3) ask whether a graph on the rows — pairs at finite distance < D, resp. <= D — holds a set of mutually
  adjacent rows of a given size among given candidates, i.e. whether its complement has a vertex cover within a
  budget <= f: decided by branching on an uncovered pair (one of its two ends must go), 2^f leaves at most.
  Test infrastructure like the rest of this module; deliberately a different search from the product's
  (csrc/api.cpp branches on the row with most non-neighbours and bisects over the distances)."""
  n = dist.shape[0]
  k = n - f
  sel = list(selection)
  if len(sel) != k or sorted(set(sel)) != sel or (sel and (sel[0] < 0 or sel[-1] >= n)):
    return False
  diam = 0.
  for x, y in itertools.combinations(sel, 2):
    v = dist[x, y]
    if not math.isfinite(v):
      return False
    diam = max(diam, v)

  def adjacent(x, y, strict):
    v = dist[min(x, y), max(x, y)]
    return math.isfinite(v) and (v < diam if strict else v <= diam)

  def holds(cand, need, strict):
    """`need` mutually adjacent rows among the list `cand`?"""
    if len(cand) < need:
      return False
    if need <= 1:
      return True
    for i, x in enumerate(cand):
      for y in cand[i + 1:]:
        if not adjacent(x, y, strict):
          if len(cand) == need:
            return False
          return holds([c for c in cand if c != x], need, strict) or holds([c for c in cand if c != y], need, strict)
    return True

  if diam > 0. and holds(list(range(n)), k, True):
    return False
  prefix, floor = [], 0
  for i, s_i in enumerate(sel):
    for c in range(floor, s_i):
      cand = [r for r in range(c + 1, n) if all(adjacent(r, q, False) for q in prefix + [c])]
      if all(adjacent(c, q, False) for q in prefix) and holds(cand, k - i - 1, False):
        return False
    prefix.append(s_i)
    floor = s_i + 1
  return True


def brute(gradients, f, precision="f32"):
  sel = brute_selection(gradients, f, precision)
  cast = (lambda t: t) if precision == "f32" else _as64
  return _seq_sum_div([cast(gradients[i]) for i in sel], len(gradients) - f)


def aksel_order(gradients, precision="f32"):
  """Rows by increasing squared distance to the coordinate-wise median (aksel.py:35-48)."""
  g = _stack(gradients, precision)
  med = g.median(dim=0).values
  cast = (lambda t: t.detach().cpu()) if precision == "f32" else _as64
  sq = [(cast(x) - med).pow_(2).sum().item() for x in gradients]
  return _stable_order(sq), sq


def aksel(gradients, f, mode="mid", precision="f32"):
  n = len(gradients)
  count = (n + 1) // 2 if mode == "mid" else n - f
  order, _ = aksel_order(gradients, precision)
  cast = (lambda t: t) if precision == "f32" else _as64
  return _seq_sum_div([cast(gradients[i]) for i in order[:count]], count)


def average(gradients, precision="f32"):
  cast = (lambda t: t) if precision == "f32" else _as64
  acc = 0
  for g in gradients:
    acc = acc + cast(g)
  return acc / len(gradients)


def cge_order(gradients, precision="f32"):
  norms = []
  for g in gradients:
    v = (g if precision == "f32" else _as64(g)).norm().item()
    norms.append(v if math.isfinite(v) else math.inf)
  return _stable_order(norms), norms


def cge(gradients, f, precision="f32"):
  keep = len(gradients) - f
  order, _ = cge_order(gradients, precision)
  cast = (lambda t: t) if precision == "f32" else _as64
  acc = cast(gradients[order[0]]).clone()
  for i in order[1:keep]:
    acc.add_(cast(gradients[i]))
  return acc.div_(keep)


# ---------------------------------------------------------------------------- #
# Statistics and momentum

def compute_avg_dev_max(samples, precision="f32"):
  """(avg, ||avg||, sqrt(sum_i ||s_i-avg||^2/(k-1)), max|avg|)  (tools/pytorch.py:97-125)."""
  k = len(samples)
  if k == 0:
    return None, math.nan, math.nan, math.nan
  cast = (lambda t: t) if precision == "f32" else _as64
  avg = cast(samples[0]).clone()
  for s in samples[1:]:
    avg.add_(cast(s))
  avg.div_(k)
  norm_avg = avg.norm().item()
  norm_max = avg.abs().max().item()
  if k >= 2:
    var = 0.
    for s in samples:
      diff = cast(s).sub(avg)
      var += diff.dot(diff).item()
    norm_dev = math.sqrt(var / (k - 1))
  else:
    norm_dev = math.nan
  return avg, norm_avg, norm_dev, norm_max


def study_block(sampled, honests, attacks, defense, pasts=(), momentum=0.9, precision="f32"):
  """The floats of attack.py:845-868 as a dict (pasts = [(grad, norm), ...], newest first)."""
  cast = (lambda t: t) if precision == "f32" else _as64
  s_avg, s_norm, s_dev, s_max = compute_avg_dev_max(sampled, precision)
  h_avg, h_norm, h_dev, h_max = compute_avg_dev_max(honests, precision)
  a_avg, a_norm, a_dev, a_max = compute_avg_dev_max(attacks, precision)
  dfs = cast(defense)
  d_norm = dfs.norm().item()
  d_max = dfs.abs().max().item()

  def cos(u, nu, v, nv):
    if u is None or v is None:
      return math.nan
    return torch.dot(u, v).div_(nu).div_(nv).item()

  res = {
    "sampled_norm_dev": s_dev, "honest_norm_dev": h_dev, "attack_norm_dev": a_dev,
    "sampled_norm_avg": s_norm, "honest_norm_avg": h_norm, "attack_norm_avg": a_norm, "defense_norm_avg": d_norm,
    "sampled_norm_max": s_max, "honest_norm_max": h_max, "attack_norm_max": a_max, "defense_norm_max": d_max,
    "cosin_splhon": cos(s_avg, s_norm, h_avg, h_norm), "cosin_splatt": cos(s_avg, s_norm, a_avg, a_norm),
    "cosin_spldef": cos(s_avg, s_norm, dfs, d_norm), "cosin_honatt": cos(h_avg, h_norm, a_avg, a_norm),
    "cosin_hondef": cos(h_avg, h_norm, dfs, d_norm), "cosin_attdef": cos(a_avg, a_norm, dfs, d_norm),
  }
  if len(pasts) > 0:
    res["cosin_sampled"] = torch.dot(s_avg, cast(pasts[0][0])).div_(s_norm).div_(pasts[0][1]).item()
    res["curv_sampled"] = momentum * sum(momentum ** i * torch.dot(s_avg, cast(p)).item()
                                         for i, (p, _) in enumerate(pasts))
  else:
    res["cosin_sampled"] = math.nan
    res["curv_sampled"] = math.nan
  res["sampled_grad_avg"] = s_avg
  return res


def worker_momentum(buffers, grads, mu, dampening):
  """In place `buf.mul_(mu).add_(grad, alpha=1-damp)` per worker (attack.py:800-804)."""
  for buf, grad in zip(buffers, grads):
    buf.mul_(mu).add_(grad, alpha=(1. - dampening))
  return buffers


# ---------------------------------------------------------------------------- #
# The "identical" attacks with their factor search (attacks/identical.py:45-86, tools/misc.py:468-514)

def line_maximize(scape, evals=16, start=0., delta=1., ratio=0.8):
  """Restatement of tools.line_maximize (tools/misc.py:468-514) as an explicit two-state machine.
  Returns (best x, [(x, y) in evaluation order]).  Expanding: candidates best+delta, delta doubling while
  the value strictly improves; the first failure shrinks delta by `ratio` and switches to contracting:
  the candidate moves towards (and past) the best point by a delta that shrinks after every evaluation,
  bounced back into x >= 0 by halving the overshoot."""
  trace = []

  def value(x):
    y = scape(x)
    trace.append((x, y))
    return y

  best = (start, value(start))
  cand = None
  expanding = True
  for _ in range(evals - 1):
    if expanding:
      cand = best[0] + delta
    elif cand < best[0]:
      cand = cand + delta
    else:
      x = cand - delta
      while x < 0:
        x = (x + cand) / 2
      cand = x
    y = value(cand)
    improved = y > best[1]
    if improved:
      best = (cand, y)
    if expanding:
      if improved:
        delta *= 2
      else:
        delta *= ratio
        expanding = False
    else:
      delta *= ratio
  return best[0], trace


def identical_attack(honests, f_real, f_decl, defense, kind="empire", factor=-16, negative=False, precision="f32"):
  """attacks/identical.py:45-86 for kind "empire" (:129-134) / "little" (:136-141).  defense(gradients, f) ->
  aggregated vector.  Returns (list of f_real aliases of the Byzantine vector, factor used, search trace or None).
  precision "f64": the search's objective is taken in float64 (the vectors stay the reference's fp32 ones)."""
  if f_real == 0:
    return [], None, None
  stck = torch.stack(honests)
  avg = stck.mean(dim=0)
  att = avg.neg() if kind == "empire" else stck.var(dim=0).sqrt_()
  trace = None
  if factor < 0:
    def scape(x):
      if negative:
        x = -x
      cand = avg + x * att
      out = defense(list(honests) + [cand] * f_real, f_decl)
      if precision == "f32":
        out = out.sub(avg)
        return out.dot(out).item()
      diff = out.double() - avg.double()
      return diff.dot(diff).item()
    factor, trace = line_maximize(scape, evals=math.ceil(-factor))
  elif negative:
    factor = -factor
  att.mul_(factor)
  byz = avg.add(att)
  return [byz] * f_real, factor, trace


# ---------------------------------------------------------------------------- #
# Synthetic gradient stacks

def make_stack(kind, n, f, d, seed, device="cpu", dtype=torch.float32):
  """Return (gradients list, nb_honests).  Byzantine rows alias ONE tensor (attacks/identical.py:86).

  iid     n rows randn (no Byzantine structure; selection is ill-conditioned — throughput only)
  hetero  mu = 0.1 randn; honest g_i = mu + sigma_i randn, sigma = linspace(0.5, 1.5, h);
          f Byzantine rows = -0.1 mean(honest)  ("empire", factor 1.1: (1 - 1.1) * mean)
  little  honest as hetero; Byzantine = mean - 1.5 std (coordinate-wise, unbiased std)
  nan     honest as hetero; Byzantine = all-NaN (attacks/nan.py)
  tight   mu = 10 randn; honest g_i = mu + sigma_i randn, sigma = linspace(0.01, 0.1, h): rows whose
          distances are 1e-3..1e-2 of their norms (what a Gram formulation G_ii+G_jj-2G_ij cancels
          on); Byzantine = -0.1 mean(honest), i.e. far outliers next to the tight cluster
  momentum  the honest rows are worker momentum buffers after 50 steps of
          buf <- 0.99 buf + 0.01 g_t (attack.py:800-804) with g_t = mu + sigma_i randn (hetero's
          distribution, a fresh draw per step); Byzantine = -0.1 mean(honest)
  """
  gen = torch.Generator(device="cpu").manual_seed(seed)
  if kind == "iid":
    return [torch.randn(d, generator=gen, dtype=dtype).to(device) for _ in range(n)], n
  h = n - f
  if kind == "tight":
    mu = 10.0 * torch.randn(d, generator=gen, dtype=dtype)
    sigmas = torch.linspace(0.01, 0.1, h)
  else:
    mu = 0.1 * torch.randn(d, generator=gen, dtype=dtype)
    sigmas = torch.linspace(0.5, 1.5, h)
  if kind == "momentum":
    honests = [torch.zeros(d, dtype=dtype) for _ in range(h)]
    for _ in range(50):
      for i in range(h):
        honests[i].mul_(0.99).add_(mu + sigmas[i] * torch.randn(d, generator=gen, dtype=dtype), alpha=0.01)
  else:
    honests = [(mu + sigmas[i] * torch.randn(d, generator=gen, dtype=dtype)) for i in range(h)]
  stack = torch.stack(honests)
  if kind in ("hetero", "tight", "momentum"):
    byz = stack.mean(dim=0).mul_(-0.1)
  elif kind == "little":
    byz = stack.mean(dim=0) - 1.5 * stack.std(dim=0)
  elif kind == "nan":
    byz = torch.full((d,), math.nan, dtype=dtype)
  else:
    raise ValueError(kind)
  byz = byz.to(device)
  return [g.to(device) for g in honests] + [byz] * f, h
