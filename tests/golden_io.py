"""Loader of tests/golden/*.npz (see tests/golden/README.md)."""

import pathlib

import numpy as np
import torch

GOLDEN_DIR = pathlib.Path(__file__).resolve().parent / "golden"

CASES = sorted(p.stem for p in GOLDEN_DIR.glob("*.npz") if not p.stem.startswith("hand_"))
HAND_CASES = sorted(p.stem for p in GOLDEN_DIR.glob("hand_*.npz"))


class Golden:
  def __init__(self, name, device="cpu"):
    data = np.load(GOLDEN_DIR / f"{name}.npz")
    self.name = name
    self.data = {k: data[k] for k in data.files}
    self.n, self.f, self.d, self.h = (int(v) for v in self.data["meta"])
    honest = [torch.from_numpy(row.copy()).to(device) for row in self.data["in_honest"]]
    if "in_byz" in self.data:
      byz = torch.from_numpy(self.data["in_byz"].copy()).to(device)
      self.gradients = honest + [byz] * (self.n - self.h)  # ONE aliased tensor, as the attacks do
    else:
      self.gradients = honest
    self.honests = self.gradients[:self.h]
    self.attacks = self.gradients[self.h:]

  def has(self, key):
    return key in self.data

  def tensor(self, key):
    return torch.from_numpy(self.data[key])

  def array(self, key):
    return self.data[key]


def same_bits(a, b):
  """torch.equal that treats NaN == NaN (and -0 == +0 like torch.equal)."""
  a = torch.as_tensor(a).detach().cpu()
  b = torch.as_tensor(b).detach().cpu()
  if a.shape != b.shape:
    return False
  nan_a, nan_b = torch.isnan(a), torch.isnan(b)
  return bool((nan_a == nan_b).all()) and torch.equal(a[~nan_a], b[~nan_b])
