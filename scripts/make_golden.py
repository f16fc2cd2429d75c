"""Generate tests/golden/*.npz by running the REAL reference (imported unmodified from
/root/reference) on seeded inputs.  Run in the build container only:

    python scripts/make_golden.py

Every fixture stores the inputs themselves (so nothing depends on RNG reproducibility across
machines) and the reference's outputs: aggregated vectors of all eleven rules, Krum/Bulyan/Aksel
orders and scores, the Brute selection, and compute_avg_dev_max of the honest and attack stacks.
Byzantine rows are stored once plus a count, and re-aliased on load (attacks/identical.py:86).
"""

import os
import pathlib
import sys

import numpy as np
import torch

ROOT = pathlib.Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))

from oracle import gar_oracle as O  # noqa: E402
from oracle import reference_loader  # noqa: E402

CASES = [
  # name, kind, n, f, d, seed, with_brute
  ("hetero_n11_f2", "hetero", 11, 2, 512, 101, True),
  ("hetero_n25_f5", "hetero", 25, 5, 777, 102, False),
  ("little_n25_f5", "little", 25, 5, 500, 103, False),
  ("iid_n13_f2", "iid", 13, 2, 300, 104, True),
  ("hetero_n51_f12", "hetero", 51, 12, 200, 105, False),
  ("nan_n11_f2", "nan", 11, 2, 100, 106, True),
  ("hetero_n7_f1", "hetero", 7, 1, 64, 107, True),
]


def hand_cases():
  """Small known-answer inputs."""
  out = {}
  # n = 4: the LOWER median is rank 1
  out["hand_lower_median"] = [torch.tensor([4., 0., -1.]), torch.tensor([1., 0., 5.]),
                              torch.tensor([3., 0., 2.]), torch.tensor([2., 0., 9.])]
  # trmean f=1 on arange(10).reshape(5, 2) -> [4, 5]
  out["hand_trmean"] = [row.clone() for row in torch.arange(10, dtype=torch.float32).reshape(5, 2)]
  # duplicated values (aliased Byzantine rows) around the centre
  base = [torch.tensor([1., 1., 1., 3., 4., 5., 6.][i:i + 1] * 3) for i in range(7)]
  out["hand_duplicates"] = base
  return out


def run_case(ref, tools, grads, f, with_brute):
  n = len(grads)
  res = {}
  res["median"] = ref.gars["median"].unchecked(gradients=grads).numpy()
  if n >= 2 * f + 1 and f >= 1:
    for name in ("trmean", "phocas", "meamed"):
      res[name] = ref.gars[name].unchecked(gradients=grads, f=f).numpy()
    for mode in ("mid", "n-f"):
      res[f"aksel_{mode}"] = ref.gars["aksel"].unchecked(gradients=grads, f=f, mode=mode).numpy()
    ak = sys.modules["aggregators.aksel"]._compute_distances(grads, f, "mid")[0]
    res["aksel_order"] = np.array([i for i, _ in ak], dtype=np.int32)
    res["aksel_sqdist"] = np.array([v for _, v in ak], dtype=np.float64)
    res["cge"] = ref.gars["cge"].unchecked(gradients=grads, f=f).numpy()
  res["average"] = ref.gars["average"].unchecked(gradients=grads).numpy()
  if n >= 2 * f + 3 and f >= 1:
    res["krum"] = ref.gars["krum"].unchecked(gradients=grads, f=f).numpy()
    res["krum_m1"] = ref.gars["krum"].unchecked(gradients=grads, f=f, m=1).numpy()
    scores = sys.modules["aggregators.krum"]._compute_scores(grads, f, None)
    order = []
    for _, gr in scores:
      # first index not yet used that IS this object (aliased rows keep index order: stable sort)
      for i, g in enumerate(grads):
        if g is gr and i not in order:
          order.append(i)
          break
    res["krum_order"] = np.array(order, dtype=np.int32)
    res["krum_scores"] = np.array([s for s, _ in scores], dtype=np.float64)
  if n >= 4 * f + 3 and f >= 1:
    res["bulyan"] = ref.gars["bulyan"].unchecked(gradients=grads, f=f).numpy()
    res["bulyan_m3"] = ref.gars["bulyan"].unchecked(gradients=grads, f=f, m=3).numpy()
  if with_brute and n >= 2 * f + 1 and f >= 1:
    res["brute_selection"] = np.array(sys.modules["aggregators.brute"]._compute_selection(grads, f), dtype=np.int32)
    res["brute"] = ref.gars["brute"].unchecked(gradients=grads, f=f).numpy()
  return res


def stats_of(tools, samples, prefix, res):
  avg, norm, dev, mx = tools.compute_avg_dev_max(samples)
  if avg is not None:
    res[prefix + "_avg"] = avg.numpy()
  res[prefix + "_stats"] = np.array([norm, dev, mx], dtype=np.float64)


def main():
  ref, tools = reference_loader.load(with_native=False)
  out_dir = ROOT / "tests" / "golden"
  out_dir.mkdir(parents=True, exist_ok=True)
  for name, kind, n, f, d, seed, with_brute in CASES:
    grads, h = O.make_stack(kind, n, f, d, seed)
    res = run_case(ref, tools, grads, f, with_brute)
    stats_of(tools, grads[:h], "honest", res)
    if h < n:
      stats_of(tools, grads[h:], "attack", res)
    res["in_honest"] = torch.stack(grads[:h]).numpy()
    if h < n:
      res["in_byz"] = grads[h].numpy()
    res["meta"] = np.array([n, f, d, h], dtype=np.int64)
    np.savez_compressed(out_dir / f"{name}.npz", **res)
    print(name, sorted(res))
  for name, grads in hand_cases().items():
    n = len(grads)
    f = 1 if n >= 5 else 0
    res = {"median": ref.gars["median"].unchecked(gradients=grads).numpy(),
           "in_honest": torch.stack(grads).numpy(), "meta": np.array([n, f, grads[0].shape[0], n], dtype=np.int64)}
    if f:
      for rule in ("trmean", "phocas", "meamed"):
        res[rule] = ref.gars[rule].unchecked(gradients=grads, f=f).numpy()
    np.savez_compressed(out_dir / f"{name}.npz", **res)
    print(name, {k: v.tolist() for k, v in res.items() if k not in ("in_honest", "meta")})


if __name__ == "__main__":
  main()
