"""The simulation step of attack.py:757-878 written with the CPU oracle — TEST INFRASTRUCTURE
(imported by tests/ and by the `cpu_baseline` leg of bench.py only; the product never imports it).

An independent restatement of the loop body (clipping, the three momentum placements, the
"identical" attacks, the rule, the momentum of the update, the study block) used to check
byzantinemomentum_amd.step.AggregationStep on the GPU and, with the oracle-backed compute legs,
on CPU under gloo.  Tensor arithmetic is the reference's own fp32 torch-CPU operations; the
norm-type floats are taken in float64 (see oracle/gar_oracle.py on why).
"""

import collections
import math

import torch

from oracle import gar_oracle as O

RULES = {"krum": O.krum, "bulyan": O.bulyan, "trmean": O.trmean, "phocas": O.phocas, "meamed": O.meamed,
         "aksel": O.aksel, "brute": O.brute, "cge": O.cge}


class ReferenceLoop:
  def __init__(self, n, f_decl, f_real, gar, momentum_at="worker", mu=0.9, damp=0.9, attack="empire", factor=1.1,
               clip=None, nb_past=3, gar_args=None, precision="f64", evals=None, negative=False):
    """precision: arithmetic of the norm-type floats ("f64" ground truth; "f32" = the reference's own fp32
    operations, bit-faithful: what tests/test_step_reference_vs_reference.py pins against the real loop body)."""
    self.precision = precision
    self.evals, self.negative = evals, negative  # evals = E: the attack's `factor:-E` (search), identical.py:67-77
    self.last_factor, self.last_search = None, None
    self.n, self.f_decl, self.f_real, self.gar = n, f_decl, f_real, gar
    self.h = n - f_real
    self.momentum_at, self.mu, self.damp = momentum_at, mu, damp
    self.attack, self.factor, self.clip = attack, factor, clip
    self.gar_args = dict(gar_args or {})
    self.workers = None
    self.server = None
    self.pasts = collections.deque(maxlen=max(nb_past, 1))
    self.nb_past = nb_past

  def step(self, sampled, params=None, origin=None):
    """sampled: list of fp32 CPU tensors (cloned here, clipping is in place in the reference)."""
    h = self.h
    sampled = [g.clone() for g in sampled]
    if self.clip is not None:  # attack.py:791-794 (norm in float64: the reference's fp32 norm() is itself off)
      for g in sampled:
        norm = g.norm().item() if self.precision == "f32" else math.sqrt(g.double().pow(2).sum().item())
        if norm > self.clip:
          g.mul_(self.clip / norm)
    if self.workers is None:
      self.workers = [torch.zeros_like(g) for g in sampled[:h]]
    if self.server is None:
      self.server = torch.zeros_like(sampled[0])
    if self.momentum_at == "worker":  # attack.py:800-804
      honests = O.worker_momentum(self.workers, sampled[:h], self.mu, self.damp)
    elif self.momentum_at == "server":  # attack.py:805-808
      honests = [g.mul(1. - self.damp).add_(self.server, alpha=self.mu) for g in sampled[:h]]
    else:
      honests = sampled[:h]
    # attacks/identical.py:63-86,129-141
    def rule(grads, f):
      if self.gar == "median":
        return O.median(grads)
      if self.gar == "average":
        return O.average(grads)
      return RULES[self.gar](grads, f, **self.gar_args)
    if self.evals is None:
      stck = torch.stack(honests)
      avg = stck.mean(dim=0)
      att = avg.neg() if self.attack == "empire" else stck.var(dim=0).sqrt_()
      att.mul_(self.factor)
      byz = avg.add(att)
      attacks = [byz] * self.f_real
    else:
      attacks, self.last_factor, self.last_search = O.identical_attack(
        honests, self.f_real, self.f_decl, rule, self.attack, -self.evals, self.negative, self.precision)
    grads = list(honests) + attacks
    self.last_gradients = grads  # what the rule saw (tests look at single columns of it: exact-tie identification)
    defense = rule(grads, self.f_decl)
    if params is None:
      l2 = math.nan
    elif self.precision == "f32":
      l2 = params.sub(origin).norm().item()
    else:
      l2 = math.sqrt((params.double() - origin.double()).pow(2).sum().item())
    if self.momentum_at == "server":  # attack.py:832-839
      self.server = defense
      update = defense
    elif self.momentum_at == "update":
      self.server.mul_(self.mu).add_(defense, alpha=(1. - self.damp))
      update = self.server
    else:
      update = defense
    pasts = list(self.pasts) if self.nb_past > 0 else []
    res = O.study_block(sampled, honests, attacks, defense, pasts, self.mu, self.precision)
    res["l2_origin"] = l2
    if self.nb_past > 0:
      self.pasts.appendleft((res["sampled_grad_avg"], res["sampled_norm_avg"]))
    return defense, update, res


FLOAT_KEYS = ("sampled_norm_avg", "honest_norm_avg", "attack_norm_avg", "defense_norm_avg", "sampled_norm_dev",
              "honest_norm_dev", "sampled_norm_max", "honest_norm_max", "attack_norm_max", "defense_norm_max")
COS_KEYS = ("cosin_splhon", "cosin_splatt", "cosin_spldef", "cosin_honatt", "cosin_hondef", "cosin_attdef",
            "cosin_sampled")


def assert_floats_close(got, want, tag="", tol=1e-5):
  for key in FLOAT_KEYS:
    assert abs(got[key] - want[key]) <= tol * max(abs(want[key]), 1e-3), (tag, key, got[key], want[key])
  assert abs(got["attack_norm_dev"] - want["attack_norm_dev"]) <= tol * want["attack_norm_avg"], (tag, "attack_norm_dev")
  for key in COS_KEYS:
    assert (math.isnan(got[key]) and math.isnan(want[key])) or abs(got[key] - want[key]) <= tol, (tag, key, got[key], want[key])
  if not math.isnan(want["curv_sampled"]):
    assert abs(got["curv_sampled"] - want["curv_sampled"]) <= tol * max(abs(want["curv_sampled"]), 1.0), (tag, "curv")
  if not math.isnan(want.get("l2_origin", math.nan)):
    assert abs(got["l2_origin"] - want["l2_origin"]) <= tol * want["l2_origin"], (tag, "l2_origin")
