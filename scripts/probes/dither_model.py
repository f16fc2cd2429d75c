"""Numpy model of the two-plane split of gram_bf16.hip (centre = median of three rows, h = rne_bf16, m = remainder
rounded to bf16 to nearest or with the coordinate dither): worst relative error of a squared distance on structured
stacks.  Run on the CPU; the GPU tests make the same comparison on the kernel itself."""
import sys
import numpy as np


def bf16_rne(x):
  b = x.astype(np.float32).view(np.uint32).astype(np.uint64)
  b = (b + 0x7FFF + ((b >> 16) & 1)) & 0xFFFF0000
  return b.astype(np.uint32).view(np.float32)


def dither16(coord, seed=0):
  def mix(c):
    z = (c.astype(np.uint64) * 0x9E3779B1 + 0x7F4A7C15) & 0xFFFFFFFF
    z ^= z >> 15
    z = (z * 0x85EBCA77) & 0xFFFFFFFF
    z ^= z >> 13
    z = (z * 0xC2B2AE3D) & 0xFFFFFFFF
    z ^= z >> 16
    return z
  even = coord & ~np.uint64(1)
  z = mix(even + np.uint64(seed))
  return np.where(coord & 1, z >> 16, z & 0xFFFF).astype(np.uint64)


def two_plane(x, mode, seed=0):
  h = bf16_rne(x)
  r = (x - h).astype(np.float32)
  bits = r.view(np.uint32).astype(np.uint64)
  if mode == "rne":
    m = bf16_rne(r)
  else:
    add = dither16(np.arange(x.shape[-1], dtype=np.uint64), seed)
    m = ((bits + add) & 0xFFFF0000).astype(np.uint32).view(np.float32)
  return h.astype(np.float64) + m.astype(np.float64)


def worst(rows, mode, tau=2e-3):
  n = rows.shape[0]
  K = (n + 3) // 4
  pa, pb, pc = 0, 4 * (K // 3), 4 * ((2 * K) // 3)
  c = np.median(rows[[pa, pb, pc]], axis=0).astype(np.float32)
  xc = (rows - c).astype(np.float32)
  xt = two_plane(xc, mode)
  x64 = rows.astype(np.float64)
  g = (xc.astype(np.float64) ** 2).sum(1)
  w, gated = 0.0, 0
  for i in range(n):
    for j in range(i + 1, n):
      true = ((x64[i] - x64[j]) ** 2).sum()
      got = ((xt[i] - xt[j]) ** 2).sum()
      if true == 0:
        assert got == 0
        continue
      if true < tau * (g[i] + g[j]):
        gated += 1  # the product recomputes these pairs with the direct kernel
        continue
      w = max(w, abs(got - true) / true)
  return w, gated


def stacks(n, d, rng):
  out = {}
  out["gauss"] = rng.standard_normal((n, d)).astype(np.float32)
  wts = rng.uniform(0.5, 2.0, n).astype(np.float32)
  out["const+1e-3noise"] = (wts[:, None] + 1e-3 * rng.standard_normal((n, d))).astype(np.float32)
  out["const"] = np.repeat(wts[:, None], d, 1).astype(np.float32)
  out["sign"] = (np.float32(0.0123) * np.sign(rng.standard_normal((n, d)))).astype(np.float32)
  g = rng.standard_normal((n, d)).astype(np.float32)
  scale = np.abs(g).max(1, keepdims=True) / 127
  out["int8"] = (np.round(g / scale) * scale).astype(np.float32)
  thr = np.quantile(np.abs(g), 0.99, axis=1, keepdims=True)
  out["top1%"] = np.where(np.abs(g) >= thr, g, 0).astype(np.float32)
  close = wts.copy()
  close[1] = close[0] * (1 + 0.07)  # a pair just above the gate
  out["const_close_pair"] = (close[:, None] + 1e-4 * rng.standard_normal((n, d))).astype(np.float32)
  return out


if __name__ == "__main__":
  d = int(sys.argv[1]) if len(sys.argv) > 1 else 1 << 20
  rng = np.random.default_rng(5)
  for name, rows in stacks(13, d, rng).items():
    a, ga = worst(rows, "rne")
    b, gb = worst(rows, "dither")
    print(f"{name:18s} d={d}: worst rel err  rne {a:.2e}   dither {b:.2e}   (gated pairs {ga})")
