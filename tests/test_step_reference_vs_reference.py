"""Pin tests/step_reference.ReferenceLoop (the restatement of the loop body, attack.py:786-868) against the
REAL reference: the same sequence of calls made with the reference's own `aggregators.gars`,
`attacks.attacks` and `tools.compute_avg_dev_max`, imported unmodified from /root/reference.
Only possible in the build container (`reference` marker)."""

import collections
import math
import sys

import pytest
import torch

from oracle import reference_loader
from tests.step_reference import ReferenceLoop

pytestmark = pytest.mark.reference

N, F, D = 11, 2, 1200


@pytest.fixture(scope="module")
def ref():
  aggregators, tools = reference_loader.load(with_native=False)
  saved = (sys.stdout, sys.stderr, sys.excepthook)
  sys.path.insert(0, reference_loader.REFERENCE_DIR)
  try:
    import attacks  # the reference's attack registry (attacks/__init__.py)
  finally:
    sys.path.remove(reference_loader.REFERENCE_DIR)
    sys.stdout, sys.stderr, sys.excepthook = saved
  return aggregators, attacks, tools


def real_loop_body(ref, state, sampled, cfg, params, origin):
  """attack.py:786-868 with the reference's own functions; `state` carries what the script keeps across steps."""
  aggregators, attacks, tools = ref
  h = N - F
  mu, damp = 0.9, 0.9
  grad_sampleds = [g.clone() for g in sampled]
  if cfg["clip"] is not None:                                             # attack.py:791-794
    for grad in grad_sampleds:
      grad_norm = grad.norm().item()
      if grad_norm > cfg["clip"]:
        grad.mul_(cfg["clip"] / grad_norm)
  if cfg["momentum_at"] == "worker":                                      # attack.py:800-810
    grad_honests = []
    for gmtm, grad in zip(state["workers"], grad_sampleds[:h]):
      gmtm.mul_(mu).add_(grad, alpha=(1. - damp))
      grad_honests.append(gmtm)
  elif cfg["momentum_at"] == "server":
    grad_honests = [grad.mul(1. - damp).add_(state["server"], alpha=mu) for grad in grad_sampleds[:h]]
  else:
    grad_honests = grad_sampleds[:h]
  defense = aggregators.gars[cfg["gar"]]
  attack = attacks.attacks[cfg["attack"]]
  if "evals" in cfg:  # the attack's own default form: factor = -E searches the factor within E evaluations
    factor, extra = -cfg["evals"], {"negative": cfg.get("negative", False)}
  else:
    factor, extra = abs(cfg["factor"]), ({"negative": True} if cfg["factor"] < 0 else {})
  grad_attacks = attack.unchecked(grad_honests=grad_honests, f_decl=F, f_real=F, model=None, defense=defense,
                                  factor=factor, **extra)                 # attack.py:819
  grad_defense = defense.unchecked(gradients=(grad_honests + grad_attacks), f=F, model=None)   # attack.py:821
  l2_origin = params.sub(origin).norm().item()                           # attack.py:830
  if cfg["momentum_at"] == "server":                                      # attack.py:832-839
    state["server"] = grad_defense
    update = grad_defense
  elif cfg["momentum_at"] == "update":
    state["server"].mul_(mu).add_(grad_defense, alpha=(1. - damp))
    update = state["server"]
  else:
    update = grad_defense
  sampled_grad_avg, sampled_norm_avg, sampled_norm_dev, sampled_norm_max = tools.compute_avg_dev_max(grad_sampleds)
  honest_grad_avg, honest_norm_avg, honest_norm_dev, honest_norm_max = tools.compute_avg_dev_max(grad_honests)
  attack_grad_avg, attack_norm_avg, attack_norm_dev, attack_norm_max = tools.compute_avg_dev_max(grad_attacks)
  defense_norm_avg = grad_defense.norm().item()
  res = {
    "l2_origin": l2_origin, "sampled_norm_avg": sampled_norm_avg, "sampled_norm_dev": sampled_norm_dev,
    "sampled_norm_max": sampled_norm_max, "honest_norm_avg": honest_norm_avg, "honest_norm_dev": honest_norm_dev,
    "honest_norm_max": honest_norm_max, "attack_norm_avg": attack_norm_avg, "attack_norm_dev": attack_norm_dev,
    "attack_norm_max": attack_norm_max, "defense_norm_avg": defense_norm_avg,
    "defense_norm_max": grad_defense.abs().max().item(),
    "cosin_splhon": torch.dot(sampled_grad_avg, honest_grad_avg).div_(sampled_norm_avg).div_(honest_norm_avg).item(),
    "cosin_splatt": torch.dot(sampled_grad_avg, attack_grad_avg).div_(sampled_norm_avg).div_(attack_norm_avg).item(),
    "cosin_spldef": torch.dot(sampled_grad_avg, grad_defense).div_(sampled_norm_avg).div_(defense_norm_avg).item(),
    "cosin_honatt": torch.dot(honest_grad_avg, attack_grad_avg).div_(honest_norm_avg).div_(attack_norm_avg).item(),
    "cosin_hondef": torch.dot(honest_grad_avg, grad_defense).div_(honest_norm_avg).div_(defense_norm_avg).item(),
    "cosin_attdef": torch.dot(attack_grad_avg, grad_defense).div_(attack_norm_avg).div_(defense_norm_avg).item(),
  }
  pasts = state["pasts"]
  if len(pasts) > 0:                                                      # attack.py:861-866
    res["cosin_sampled"] = torch.dot(sampled_grad_avg, pasts[0][0]).div_(sampled_norm_avg).div_(pasts[0][1]).item()
    res["curv_sampled"] = mu * sum((mu ** i * torch.dot(sampled_grad_avg, p).item()) for i, (p, _) in enumerate(pasts))
  else:
    res["cosin_sampled"] = math.nan
    res["curv_sampled"] = math.nan
  pasts.appendleft((sampled_grad_avg, sampled_norm_avg))                  # attack.py:868
  return grad_defense, update, res


CONFIGS = [
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", factor=1.1),
  dict(gar="bulyan", momentum_at="server", clip=24.0, attack="little", factor=1.5),
  dict(gar="median", momentum_at="update", clip=26.0, attack="empire", factor=1.1),
  dict(gar="trmean", momentum_at="worker", clip=None, attack="little", factor=-1.5),
  dict(gar="aksel", momentum_at="server", clip=None, attack="empire", factor=1.1),
  dict(gar="krum", momentum_at="worker", clip=None, attack="empire", factor=1.1, evals=16),
  dict(gar="krum", momentum_at="server", clip=None, attack="little", factor=1.1, evals=9, negative=True),
  dict(gar="median", momentum_at="update", clip=None, attack="little", factor=1.1, evals=16),
  dict(gar="brute", momentum_at="worker", clip=24.0, attack="empire", factor=1.1, evals=5),
]


@pytest.mark.parametrize("cfg", CONFIGS, ids=lambda c: f"{c['gar']}-{c['momentum_at']}-clip{c['clip']}-{c['attack']}"
                                                      + (f"-search{c['evals']}" if "evals" in c else ""))
def test_reference_loop_restatement_is_bit_faithful(ref, cfg):
  h = N - F
  gen = torch.Generator().manual_seed(11)
  origin = torch.randn(D, generator=gen)
  params = origin + 0.01 * torch.randn(D, generator=gen)
  state = {"workers": [torch.zeros(D) for _ in range(h)], "server": torch.zeros(D),
           "pasts": collections.deque(maxlen=3)}
  mine = ReferenceLoop(N, F, F, cfg["gar"], cfg["momentum_at"], 0.9, 0.9, cfg["attack"], cfg["factor"], cfg["clip"], 3,
                       precision="f32", evals=cfg.get("evals"), negative=cfg.get("negative", False))
  for it in range(4):
    base = 0.2 * torch.randn(D, generator=gen)
    sampled = [base + (0.5 + 0.1 * i) * torch.randn(D, generator=gen) for i in range(h)]
    want_def, want_upd, want = real_loop_body(ref, state, sampled, cfg, params, origin)
    got_def, got_upd, got = mine.step(sampled, params, origin)
    assert torch.equal(got_def, want_def), (cfg, it)
    assert torch.equal(got_upd, want_upd), (cfg, it)
    for key, val in want.items():
      g = got[key]
      assert (math.isnan(g) and math.isnan(val)) or g == val, (cfg, it, key, g, val)
    params = params - 0.05 * want_upd
