"""The arithmetic of the bf16 split behind the distance pass (csrc/gram_split.h), on the numpy model of
scripts/probes/dither_model.py — the claims DESIGN 4.2 makes about it, checked without a GPU:

  * three planes: x = h + m + l EXACTLY, every part a bf16 number (split3);
  * two planes: h = rne_bf16(x), m = the remainder rounded to bf16 by adding 16 coordinate-dependent pseudo-random bits
    and truncating — what is dropped has (nearly) zero mean over the coordinates and is bounded by one bf16 ulp of the
    remainder; every row sees the same dither at a coordinate, so bitwise-equal rows give bitwise-equal planes;
  * on structured stacks (rows of few distinct values) the dithered split keeps every squared distance above the
    accuracy gate within 1e-5, where rounding the remainder to nearest does not.

(The kernel itself is compared with fp64 direct differences on the GPU: tests/test_gpu_parity_r3.py.)"""

import importlib.util
import pathlib

import numpy as np

ROOT = pathlib.Path(__file__).resolve().parent.parent


def _model():
  spec = importlib.util.spec_from_file_location("dither_model", ROOT / "scripts" / "probes" / "dither_model.py")
  mod = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(mod)
  return mod


def _is_bf16(x):
  return bool(((x.astype(np.float32).view(np.uint32) & 0xFFFF) == 0).all())


def _values(rng, count):
  """fp32 values over many binades, both signs, with exact zeros and values whose low bits are all set."""
  x = (rng.standard_normal(count) * np.exp2(rng.integers(-20, 20, count))).astype(np.float32)
  x[:7] = np.array([0.0, -0.0, 1.0, -1.0, 1.0 + 2.0 ** -23, 3.0 - 2.0 ** -22, 1.9999999], dtype=np.float32)
  return x


def test_three_plane_split_is_exact():
  M = _model()
  x = _values(np.random.default_rng(1), 200000)
  h = M.bf16_rne(x)
  r = (x - h).astype(np.float32)
  m = M.bf16_rne(r)
  low = (r - m).astype(np.float32)
  assert _is_bf16(h) and _is_bf16(m) and _is_bf16(low)        # l has at most 8 significant bits: a bf16 number as it is
  assert np.array_equal(h.astype(np.float64) + m.astype(np.float64) + low.astype(np.float64), x.astype(np.float64))
  assert np.array_equal((x - h).astype(np.float64), x.astype(np.float64) - h.astype(np.float64))  # the subtractions are exact


def test_two_plane_dithered_split_is_unbiased_and_bounded():
  M = _model()
  rng = np.random.default_rng(2)
  d = 1 << 18
  x = (1.0 + 0.37 * rng.random(d)).astype(np.float32)          # one binade: the dropped part is comparable across coordinates
  approx = M.two_plane(x, "dither")
  err = approx - x.astype(np.float64)
  h = M.bf16_rne(x)
  r = np.abs((x - h).astype(np.float32))
  ulp_r = np.exp2(np.floor(np.log2(np.maximum(r, 1e-30))) - 7)  # one bf16 ulp of the remainder
  assert (np.abs(err) <= ulp_r + 1e-30).all()                   # truncation after the add: less than one ulp either way
  # zero mean: the sum over d coordinates is a random walk (~ sqrt(d) * rms), not d * bias
  rms = float(np.sqrt((err ** 2).mean()))
  assert abs(err.sum()) <= 5.0 * np.sqrt(d) * rms
  # round-to-nearest of the same remainder on a CONSTANT row is as biased as a split can be: every coordinate drops the same
  const = np.full(d, 1.2345678, dtype=np.float32)
  e_rne = M.two_plane(const, "rne") - const.astype(np.float64)
  e_dit = M.two_plane(const, "dither") - const.astype(np.float64)
  assert abs(e_rne.sum()) == d * abs(e_rne[0]) and abs(e_rne[0]) > 0
  assert abs(e_dit.sum()) <= 0.02 * abs(e_rne.sum())


def test_equal_rows_give_equal_planes_and_the_dither_depends_on_the_coordinate_only():
  M = _model()
  rng = np.random.default_rng(3)
  row = rng.standard_normal(4096).astype(np.float32)
  stack = np.stack([row, row.copy(), (row * np.float32(1.5)).astype(np.float32)])
  planes = M.two_plane(stack, "dither")
  assert np.array_equal(planes[0], planes[1])                   # exact ties between aliased rows survive the split
  a = M.dither16(np.arange(64, dtype=np.uint64))
  assert np.array_equal(a, M.dither16(np.arange(64, dtype=np.uint64))) and len(set(a.tolist())) > 48
  assert int(a.max()) < (1 << 16)


def test_structured_stacks_stay_within_1e5_with_the_dither():
  M = _model()
  rng = np.random.default_rng(5)
  for name, rows in M.stacks(13, 1 << 16, rng).items():
    worst, _ = M.worst(rows, "dither")
    assert worst <= 1e-5, (name, worst)
