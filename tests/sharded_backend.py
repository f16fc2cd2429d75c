"""Oracle-backed compute legs for byzantinemomentum_amd.sharded.ShardedAggregator — lets the CPU
tests exercise the partitioning and the collectives (gloo, world_size 2) without a GPU.
TEST INFRASTRUCTURE: the product's default backend is HipBackend and has no CPU path."""

import math

import torch

from oracle import gar_oracle as O


class OracleBackend:
  def pairwise_sqdist(self, gradients):
    n = len(gradients)
    sq = torch.zeros(n, n, dtype=torch.float64)
    g64 = [g.double() for g in gradients]
    for i in range(n):
      for j in range(i + 1, n):
        diff = g64[i] - g64[j]
        sq[i, j] = sq[j, i] = torch.dot(diff, diff)
    return sq

  def rank(self, sq, n, f, m, mode):
    dist = sq.sqrt().numpy().copy()
    dist[~torch.isfinite(sq).numpy()] = math.inf
    scores = []
    for i in range(n):
      others = sorted(dist[i, j] for j in range(n) if j != i)
      take = (n - f - 1) if mode == 0 else m
      total = 0
      for v in others[:take]:
        total = total + v
      scores.append(total)
    order = sorted(range(n), key=lambda i: scores[i])
    return torch.tensor(order + [0] * (64 - n), dtype=torch.int32)

  def selected_mean(self, gradients, order, m):
    acc = 0
    for i in order[:m].tolist():
      acc = acc + gradients[i]
    return acc.div_(m)

  def bulyan_pass2(self, gradients, order, f, m):
    n = len(gradients)
    m_max, theta = n - f - 2, n - 2 * f - 2
    beta = theta - 2 * f
    idx = order[:n].tolist()
    picked = []
    for i in range(theta):
      count = min(m, m_max - i)
      acc = 0
      for gi in idx[i:i + count]:
        acc = acc + gradients[gi]
      picked.append(acc.div_(count))
    sel = torch.stack(picked)
    return O._closest_like_reference(sel, beta, sel.median(dim=0).values)

  def colwise(self, rule, gradients, f):
    return O.median(gradients) if rule == "median" else getattr(O, rule)(gradients, f)

  def aksel_sqdist(self, gradients):
    med = torch.stack(gradients).median(dim=0).values
    return torch.tensor([(g - med).double().pow(2).sum().item() for g in gradients], dtype=torch.float64)

  def argsort(self, keys, n):
    vals = keys[:n].tolist()
    order = sorted(range(n), key=lambda i: vals[i])
    return torch.tensor(order + [0] * (64 - n), dtype=torch.int32)

  def stack_stats(self, samples):
    avg, _, _, _ = O.compute_avg_dev_max(samples)
    a64 = avg.double()
    dev2 = sum((s.double() - a64).pow(2).sum().item() for s in samples)
    out3 = torch.tensor([a64.pow(2).sum().item(), dev2, avg.abs().max().item()], dtype=torch.float64)
    return avg, out3
