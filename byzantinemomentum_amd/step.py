"""One aggregation step of the simulation loop on the HIP path (mirror of attack.py:800-878).

Given the sampled honest gradients of a step it performs, without leaving the GPU:
  1. worker-side momentum        buf_i <- mu*buf_i + (1-damp)*g_i            attack.py:800-804
  2. the "empire" / "little" attack  byz = avg_h + factor*dir, repeated f      attacks/identical.py:63-86,129-141
     fused with the honest-stack statistics (one pass over the honest stack)  attack.py:847
  3. the aggregation rule         defense = GAR(honests + [byz]*f, f)          attack.py:821
  4. the study statistics         sampled / attack stacks, defense norm and max, six cosines,
                                  previous-step cosine and curvature           attack.py:842-868
The model update itself (attack.py:832-839) belongs to the caller.  Everything is asynchronous
on the current stream until `AggregationStep.floats()` fetches the scalars (one sync).
"""

import collections
import math

import torch

from . import gars
from . import stats

__all__ = ["AggregationStep"]

_RULES = {"krum": gars.krum, "bulyan": gars.bulyan, "median": gars.median, "trmean": gars.trmean,
          "phocas": gars.phocas, "meamed": gars.meamed, "aksel": gars.aksel, "brute": gars.brute,
          "average": gars.average, "cge": gars.cge}


class AggregationStep:
  def __init__(self, nb_workers, nb_decl_byz, nb_real_byz, gar="krum", gar_args=None, momentum=0.99,
               dampening=0.99, attack="empire", attack_factor=1.1, nb_past=25):
    if gar not in _RULES:
      raise ValueError(f"unknown aggregation rule {gar!r}")
    self.n = nb_workers
    self.f_decl = nb_decl_byz
    self.f_real = nb_real_byz
    self.h = nb_workers - nb_real_byz
    self.rule = _RULES[gar]
    self.gar_args = dict(gar_args or {})
    self.mu = momentum
    self.damp = dampening
    if attack not in ("empire", "little"):
      raise ValueError(f"unknown attack {attack!r} (empire: factor, little: factor, use a negative one for negative:True)")
    self.attack = attack
    self.factor = attack_factor
    self.buffers = None                      # storage["momentum"]: one per honest worker (attack.py:676)
    self.pasts = collections.deque(maxlen=nb_past)  # (sampled average, its squared norm tensor)
    self._pending = None

  def run(self, grad_sampleds):
    """grad_sampleds: list of >= h flat fp32 GPU tensors (the step's sampled gradients).
    Returns the aggregated gradient; statistics stay on the device until floats()."""
    h = self.h
    sampled = list(grad_sampleds)
    self._nb_sampled = len(sampled)
    if self.buffers is None:
      self.buffers = [torch.zeros_like(g) for g in sampled[:h]]
    # 1. worker momentum in place; the buffers ARE the honest gradients the rule sees
    stats.multi_axpby(self.buffers, sampled[:h], self.mu, 1.0 - self.damp)
    honests = self.buffers
    # 2. honest-stack statistics + empire vector in one pass
    h_avg, h_out3, byz = stats.stack_stats_async(honests, scale=self.factor, attack=self.attack)
    attacks = [byz] * self.f_real
    # 3. aggregation
    if self.rule in (gars.median, gars.average):
      defense = self.rule(honests + attacks)
    else:
      defense = self.rule(honests + attacks, self.f_decl, **self.gar_args)
    # 4. remaining statistics
    s_avg, s_out3 = stats.stack_stats_async(sampled)
    a_avg, a_out3 = stats.stack_stats_async(attacks) if self.f_real > 0 else (None, None)
    _, d_out3 = stats.stack_stats_async([defense])
    core = [s_avg, h_avg, defense] + ([a_avg] if a_avg is not None else [])
    gram, extra = stats.study_dots(core, [p for p, _ in self.pasts])
    self._pending = (s_out3, h_out3, a_out3, d_out3, gram, extra, len(self.pasts), s_avg)
    return defense

  def floats(self):
    """Python floats of the study row (attack.py:845-868) for the last run(); synchronises once."""
    s_out3, h_out3, a_out3, d_out3, gram, extra, npast, s_avg = self._pending
    parts = [s_out3, h_out3, d_out3, gram.reshape(-1), extra] + ([a_out3] if a_out3 is not None else [])
    flat = torch.cat(parts).tolist()
    s3, h3, d3 = flat[0:3], flat[3:6], flat[6:9]
    nc = gram.shape[0]
    g = [flat[9 + i * nc: 9 + (i + 1) * nc] for i in range(nc)]
    ex = flat[9 + nc * nc: 9 + nc * nc + npast]
    a3 = flat[9 + nc * nc + npast:] if a_out3 is not None else None
    k_h, k_a = self.h, self.f_real

    def dev(out3, k):
      return math.sqrt(out3[1] / (k - 1)) if k >= 2 else math.nan

    def cos(i, j):
      if i >= nc or j >= nc:
        return math.nan
      return g[i][j] / math.sqrt(g[i][i]) / math.sqrt(g[j][j])

    res = {
      "sampled_norm_avg": math.sqrt(s3[0]), "sampled_norm_dev": dev(s3, self._nb_sampled), "sampled_norm_max": s3[2],
      "honest_norm_avg": math.sqrt(h3[0]), "honest_norm_dev": dev(h3, k_h), "honest_norm_max": h3[2],
      "attack_norm_avg": math.sqrt(a3[0]) if a3 else math.nan, "attack_norm_dev": dev(a3, k_a) if a3 else math.nan,
      "attack_norm_max": a3[2] if a3 else math.nan,
      "defense_norm_avg": math.sqrt(d3[0]), "defense_norm_max": d3[2],
      "cosin_splhon": cos(0, 1), "cosin_spldef": cos(0, 2), "cosin_hondef": cos(1, 2),
      "cosin_splatt": cos(0, 3), "cosin_honatt": cos(1, 3), "cosin_attdef": cos(3, 2),
    }
    if npast > 0:
      past_norm = self.pasts[0][1]
      res["cosin_sampled"] = ex[0] / math.sqrt(s3[0]) / past_norm
      res["curv_sampled"] = self.mu * sum(self.mu ** i * ex[i] for i in range(npast))
    else:
      res["cosin_sampled"] = math.nan
      res["curv_sampled"] = math.nan
    # grad_pasts.appendleft(PastGrad(sampled_grad_avg, sampled_norm_avg))  (attack.py:868)
    self.pasts.appendleft((s_avg, math.sqrt(s3[0])))
    return res

