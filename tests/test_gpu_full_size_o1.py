"""Full-size selection parity against the reference's OWN fp32 arithmetic (SURVEY.md §8c "O1").

BASELINE.json: "Krum/Bulyan selected indices bit-identical to reference".  The other full-size tests rank
fp64 distances; here the very functions of the reference (`aggregators/krum.py:31-80`,
`aggregators/bulyan.py:31-84`) run on the GPU box's host cores on CPU copies of the same tensors — the staged
checkout of `scripts/stage_reference.sh` when it is there, else the oracle's f32 port (pinned bit-identical to
the reference by tests/test_oracle_vs_reference.py) — at C3 (n = 51, f = 12) and C4 (n = 25, f = 5),
d = 11 173 962, on the `hetero` and `little` stacks of SURVEY.md §8d.  Each case prints the decisive score gap.
"""

import math
import os
import time

import pytest
import torch

from oracle import gar_oracle as O
from oracle import reference_loader

pytestmark = pytest.mark.gpu

DEV = "cuda:0"
D_RESNET18 = 11173962


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()
  return byzantinemomentum_amd


@pytest.fixture(scope="module")
def ref():
  torch.set_num_threads(min(32, os.cpu_count() or 1))  # the per-pair torch ops do not scale past it (256 threads: 174 s at C3, 32: 16 s)
  if reference_loader.available():
    return reference_loader.load(with_native=False)[0]
  return None


def gpu_stack(kind, n, f, d, seed):
  gen = torch.Generator(device=DEV).manual_seed(seed)
  h = n - f
  mu = 0.1 * torch.randn(d, device=DEV, generator=gen)
  honest = [mu + s * torch.randn(d, device=DEV, generator=gen) for s in torch.linspace(0.5, 1.5, h).tolist()]
  stack = torch.stack(honest)
  if kind == "hetero":
    byz = stack.mean(dim=0).mul_(-0.1)          # empire, factor 1.1 (attacks/identical.py:72,134)
  else:
    byz = stack.mean(dim=0) - 1.5 * stack.var(dim=0).sqrt_()   # little, factor 1.5, negative (identical.py:141)
  del stack
  return honest + [byz] * f, h


def host_copy(rows):
  seen = {}
  return [seen.setdefault(id(g), g.cpu()) for g in rows]


def classes(indices, h):
  """A selection as the reference can express it: honest rows by index, the aliased Byzantine rows (ONE tensor
  object, `attacks/identical.py:86`) by how many of them were taken."""
  return sorted(i for i in indices if i < h), sum(1 for i in indices if i >= h)


@pytest.mark.parametrize("kind", ["hetero", "little"])
def test_c3_krum_selection_equals_the_references_fp32(bm, ref, kind):
  n, f, d = 51, 12, D_RESNET18
  m = n - f - 2
  rows, h = gpu_stack(kind, n, f, d, seed=31)
  cpu = host_copy(rows)
  t0 = time.perf_counter()
  if ref is not None:
    ranked = ref.gars["krum"].unchecked.__globals__["_compute_scores"](cpu, f, m)   # krum.py:31-63
    scores = [s for s, _ in ranked]
    index_of = {id(g): i for i, g in reversed(list(enumerate(cpu)))}
    want = [index_of[id(g)] for _, g in ranked[:m]]   # an aliased row reads as the first Byzantine index
    want_classes = (sorted(i for i in want if i < h), sum(1 for i in want if i >= h))
    want_out = sum(g for _, g in ranked[:m]).div_(m)                                # krum.py:80
    source = "reference (staged checkout)"
  else:
    order, sc = O.krum_order(cpu, f)
    scores = sorted(sc)
    want_classes = classes(order[:m], h)
    want_out = O._seq_sum_div([cpu[i] for i in order[:m]], m)
    source = "oracle f32 port"
  elapsed = time.perf_counter() - t0
  got = bm.gars.krum_selection(rows, f)
  gap = (scores[m] - scores[m - 1]) / scores[m]
  print(f"C3 {kind}: {source} {elapsed:.2f} s on {torch.get_num_threads()} threads; decisive gap between ranks "
        f"{m - 1} and {m}: {gap:.3e} relative ({scores[m - 1]:.6f} < {scores[m]:.6f})")
  assert classes(got, h) == want_classes
  out = bm.krum(rows, f).cpu()
  scale = float(want_out.abs().max())
  assert float((out - want_out).abs().max()) <= 1e-6 * scale


@pytest.mark.parametrize("kind", ["hetero", "little"])
def test_c4_bulyan_equals_the_references_fp32(bm, ref, kind):
  n, f, d = 25, 5, D_RESNET18
  m = n - f - 2
  theta, beta = n - 2 * f - 2, n - 4 * f - 2
  rows, h = gpu_stack(kind, n, f, d, seed=32)
  cpu = host_copy(rows)
  t0 = time.perf_counter()
  if ref is not None:
    want_out = ref.gars["bulyan"].unchecked(gradients=cpu, f=f)                      # bulyan.py:31-84
    source = "reference (staged checkout)"
  else:
    want_out = O.bulyan(cpu, f)
    source = "oracle f32 port"
  elapsed = time.perf_counter() - t0
  # the ranking is not observable through the reference's API; the port's (same arithmetic, pinned) is
  order, scores = O.bulyan_order(cpu, f)
  srt = sorted(scores)
  gaps = [(b - a) / b for a, b in zip(srt, srt[1:]) if b > a]
  print(f"C4 {kind}: {source} {elapsed:.2f} s on {torch.get_num_threads()} threads; smallest gap between "
        f"distinct neighbouring scores: {min(gaps):.3e} relative")
  got = bm.gars.bulyan_ranking(rows, f)
  assert classes(got[:m], h) == classes(order[:m], h)
  assert [r if r < h else h for r in got[:m]] == [r if r < h else h for r in order[:m]], "rank order of pass 2's rows"
  out = bm.bulyan(rows, f)
  scale = float(want_out.abs().max())
  bad = ((out.cpu() - want_out).abs() > 2e-6 * scale).to(DEV)
  nbad = int(bad.sum())
  if nbad:
    # the only legitimate disagreement with `topk(..., sorted=False)` (bulyan.py:81): the beta-th and (beta+1)-th
    # deviations from the median are EQUAL, either row may be kept — identify every such column
    sel = []
    for i in range(theta):
      acc = torch.zeros(d, dtype=torch.float32, device=DEV)
      for r in got[i:m]:
        acc = acc + rows[r]
      sel.append(torch.div(acc, torch.full_like(acc, float(m - i))))
    sel = torch.stack(sel)
    dev = (sel - sel.median(dim=0).values).abs().sort(dim=0).values
    tie = dev[beta - 1] == dev[beta]
    assert int((bad & ~tie).sum()) == 0, (nbad, int((bad & ~tie).sum()))
    assert nbad <= d // 1000
  print(f"C4 {kind}: {nbad} of {d} columns differ by more than 2e-6 of the largest coordinate, all exact ties")
