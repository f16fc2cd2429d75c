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


def test_markstein_division_by_a_small_integer_is_the_ieee_quotient():
  """div_small_int (csrc/bm_common.h): q = x * (1/m), r = fma(-q, m, x), q1 = fma(r, 1/m, q) — three VALU ops in
  place of the IEEE division sequence — gives the correctly rounded x / m, the bits of torch's `.div_(m)` (but for the sign of a zero quotient)
  (aggregators/krum.py:80, bulyan.py:70, trmean.py:33), for every count m = 1..64 the rules divide by; non-finite
  inputs keep the plain product.  Model: fp32 operations with the fused multiply-adds evaluated exactly (64-bit
  mantissa: the 48-bit product and the cancelling sum are exact) and rounded once."""
  f32 = np.float32
  assert np.finfo(np.longdouble).nmant >= 63

  def fma32(a, b, c):
    return (a.astype(np.longdouble) * b.astype(np.longdouble) + c.astype(np.longdouble)).astype(np.float32)

  def div_small_int(x, m):
    mm = np.full_like(x, m, dtype=np.float32)
    rm = (f32(1.0) / mm).astype(np.float32)
    q = (x * rm).astype(np.float32)
    with np.errstate(invalid="ignore"):
      r = fma32(-q, mm, x)
      q1 = fma32(r, rm, q)
    return np.where(q1 == q1, q1, q)

  rng = np.random.default_rng(0)
  for m in range(1, 65):
    x = (rng.standard_normal(60000) * np.exp2(rng.integers(-30, 30, 60000))).astype(np.float32)
    x[:6] = np.array([0.0, -0.0, 1.0, float(m), 3.0 * m, 16777215.0], dtype=np.float32)
    got, want = div_small_int(x, m), (x / f32(m)).astype(np.float32)
    nonzero = want != 0
    assert np.array_equal(got.view(np.uint32)[nonzero], want.view(np.uint32)[nonzero]), m
    # a zero quotient comes out as +0 whatever its sign (fma(+0, m, -0) = +0): equal as a value, the one bit that differs
    assert np.array_equal(got[~nonzero], np.zeros(int((~nonzero).sum()), dtype=np.float32)) and not np.signbit(got[~nonzero]).any()
    edge = np.array([np.inf, -np.inf, np.nan], dtype=np.float32)
    ge = div_small_int(edge, m)
    assert ge[0] == np.inf and ge[1] == -np.inf and np.isnan(ge[2])
