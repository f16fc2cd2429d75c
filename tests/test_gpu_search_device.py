"""The attacks' factor search evaluated on the device (csrc/search_device.hip, bm_attack_line_search_device) against the
host form (csrc/linesearch.cpp, bm_attack_line_search) — attacks/identical.py:67-77 with tools/misc.py:468-514.

Both forms work from the (h+2) x (h+2) squared distances of one distance pass and share their closed forms and their
cursor (csrc/search_core.h), so the bar is bit identity: the same sixteen abscissae, the same sixteen objectives, the
same factor — on matrices that come from real stacks and on adversarial ones (ties everywhere, zero and non-finite
entries, geometry no set of vectors has).  The host form itself is pinned to the reference's loop elsewhere
(tests/test_step_cpu.py, tests/test_gpu_parity_r2.py / _r4.py: same candidates, objectives within 2e-5).
Needs an MI355X: `pytest -m gpu`.
"""

import math

import pytest
import torch

from tests.test_gpu_parity_r2 import DEV

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def bm():
  import byzantinemomentum_amd
  byzantinemomentum_amd._lib.load()
  return byzantinemomentum_amd


def _both(bm, ext_dev, h, k, f, rule, evals, negative, m):
  from byzantinemomentum_amd import linesearch
  out = bm.stats.attack_search_device(ext_dev, h, k, f, rule, evals=evals, negative=negative, m=m).cpu().tolist()
  got = (out[0], [(out[1 + 2 * i], out[2 + 2 * i]) for i in range(evals)])
  want = linesearch.attack_line_search(ext_dev.cpu().contiguous(), h, k, f, rule, evals=evals, negative=negative, m=m)
  return got, want


def _same_bits(got, want, tag):
  (fg, tg), (fw, tw) = got, want
  assert len(tg) == len(tw), tag
  for e, ((x, y), (xo, yo)) in enumerate(zip(tg, tw)):
    same_y = (y == yo) or (math.isnan(y) and math.isnan(yo))
    assert x == xo and same_y, (tag, e, (x, y), (xo, yo))
  assert fg == fw, (tag, fg, fw)


def _ext_of_stack(bm, h, d, seed, kind):
  """(h+2) x (h+2) device matrix of honests + [avg, avg + att] as AggregationStep forms it."""
  gen = torch.Generator().manual_seed(seed)
  base = torch.randn(d, generator=gen)
  if kind == "hetero":
    rows = [base + (0.3 + 0.05 * i) * torch.randn(d, generator=gen) for i in range(h)]
  elif kind == "tight":
    rows = [base + 1e-3 * torch.randn(d, generator=gen) for _ in range(h)]
  elif kind == "duplicates":  # exact ties among the honest distances
    few = [base + torch.randn(d, generator=gen) for _ in range(max(1, h // 3))]
    rows = [few[i % len(few)].clone() for i in range(h)]
  else:
    raise ValueError(kind)
  honests = [r.to(DEV) for r in rows]
  avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
  unit = torch.empty_like(avg)
  bm.stats.multi_fma3([unit], [avg], [direction], 1.0, 1.0)
  return bm.gars.pairwise_sqdist(honests + [avg, unit])


SHAPES = [
  # h, k, f, m, evals
  (9, 2, 2, None, 16),
  (20, 5, 5, None, 16),      # C2 / C4 worker counts
  (20, 5, 5, 1, 12),         # plain Krum
  (39, 12, 12, None, 16),    # C3
  (39, 12, 12, 51, 16),      # every row selected
  (62, 2, 2, None, 16),      # the most honest rows a distance pass serves (h + 2 = 64)
  (33, 31, 15, None, 16),    # as many copies as the row count allows
  (14, 0, 3, None, 8),       # no Byzantine row at all
  (1, 0, 0, 1, 4),           # one row
  (2, 1, 0, 1, 16),
]


@pytest.mark.parametrize("kind", ["hetero", "tight", "duplicates"])
@pytest.mark.parametrize("shape", SHAPES, ids=lambda s: "h%d-k%d-f%d-m%s-e%d" % s)
def test_device_search_equals_host_search_on_stacks(bm, shape, kind):
  h, k, f, m, evals = shape
  ext = _ext_of_stack(bm, h, 20011, 7 * h + k, kind)
  for rule in ("krum", "average"):
    for negative in (False, True):
      got, want = _both(bm, ext, h, k, f, rule, evals, negative, m if rule == "krum" else None)
      _same_bits(got, want, (shape, kind, rule, negative))


def test_device_search_equals_host_search_on_adversarial_matrices(bm):
  """Matrices no stack produces: small integers (ties in every row and among the scores), zeros off the diagonal,
  +inf / NaN entries (krum.py:46-47: a non-finite distance counts as +inf), a huge |att|^2 and a zero one."""
  gen = torch.Generator().manual_seed(99)
  cases = 0
  for h, k, f in ((7, 2, 2), (20, 5, 5), (39, 12, 12), (50, 14, 14), (63, 1, 1), (64, 0, 3)):  # (the last two: more honest rows than a distance pass can serve, legal for the search)
    e = h + 2
    for trial in range(12):
      a = torch.randint(0, 6, (e, e), generator=gen).double()
      ext = a + a.t()
      ext.fill_diagonal_(0.0)
      if trial % 4 == 1:
        ext[h, h + 1] = ext[h + 1, h] = 0.0          # att = 0: every candidate is the honest average
      if trial % 4 == 2:
        ext[h, h + 1] = ext[h + 1, h] = 1e12
      if trial % 4 == 3:
        i, j = int(torch.randint(0, h, (1,), generator=gen)), int(torch.randint(0, h, (1,), generator=gen))
        ext[i, j] = ext[j, i] = math.inf if trial % 8 == 3 else math.nan
        ext[i, h + 1] = ext[h + 1, i] = math.inf
      if trial >= 8:
        ext = ext * (0.5 + torch.rand(e, e, generator=gen).double())  # no ties, not symmetric in its low bits
      dev = ext.contiguous().to(DEV)
      for rule in ("krum", "average"):
        got, want = _both(bm, dev, h, k, f, rule, 16, trial % 2 == 1, None)
        _same_bits(got, want, (h, k, f, trial, rule))
        cases += 1
  assert cases == 144


def test_factor_applied_from_device_memory_has_the_bits_of_the_host_factor(bm):
  """bm_multi_fma3_bdev: avg + factor * att with the factor read from device memory = the same call with the number."""
  gen = torch.Generator().manual_seed(5)
  for d in (1, 7, 4096, 100003):
    avg, att = torch.randn(d, generator=gen).to(DEV), torch.randn(d, generator=gen).to(DEV)
    for factor in (0.0, 1.0, 0.8 ** 7, 1234.56789, 1e-30):
      want, got = torch.empty_like(avg), torch.empty_like(avg)
      bm.stats.multi_fma3([want], [avg], [att], 1.0, factor)
      where = torch.tensor([factor, 777.0, -1.0], dtype=torch.float64, device=DEV)
      bm.stats.multi_fma3([got], [avg], [att], 1.0, where)
      assert torch.equal(got, want), (d, factor)
  with pytest.raises(bm.gars.GarInputError):
    bm.stats.multi_fma3([got], [avg], [att], 1.0, torch.tensor([1.0], dtype=torch.float32, device=DEV))


def test_step_with_device_search_has_no_host_round_trip_and_matches_the_host_form(bm):
  """AggregationStep(line_search="auto") against Multi-Krum: the factor never leaves the device during run() (the
  search and the Byzantine vector are captured in a HIP graph together with the distance pass, which a copy to the host
  or a stream synchronisation would refuse), and the step equals the one that searches on the host, bit for bit."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d = 25, 5, 60013
  h = n - f
  gen = torch.Generator().manual_seed(11)
  steps = {mode: AggregationStep(n, f, f, gar="krum", momentum=0.9, dampening=0.9, momentum_at="update",
                                 attack="empire", nb_past=2, attack_evals=16, line_search=mode) for mode in ("auto", "host")}
  origin = torch.randn(d, generator=gen).to(DEV)
  for it in range(3):
    sampled = [(0.2 * torch.randn(d, generator=gen) + (0.5 + 0.1 * i) * torch.randn(d, generator=gen)).to(DEV) for i in range(h)]
    outs = {}
    for mode, step in steps.items():
      outs[mode] = step.run([g.clone() for g in sampled], origin, origin).clone()
      if mode == "auto":
        assert isinstance(step._factor_now, torch.Tensor), "the device search must leave its factor on the device"
    assert torch.equal(outs["auto"], outs["host"]), it
    assert steps["auto"].last_search == steps["host"].last_search and steps["auto"].last_factor == steps["host"].last_factor
    assert steps["auto"].floats() == steps["host"].floats()
  # the search + the Byzantine vector under stream capture
  honests = sampled
  avg, _, direction = bm.stats.stack_stats_async(honests, scale=1.0, attack="empire", direction=True)
  unit, byz = torch.empty_like(avg), torch.empty_like(avg)
  bm.stats.multi_fma3([unit], [avg], [direction], 1.0, 1.0)
  sq = bm.gars.pairwise_sqdist(honests + [avg, unit])
  eager = bm.stats.attack_search_device(sq, h, f, f, "krum", evals=16)
  bm.stats.multi_fma3([byz], [avg], [direction], 1.0, eager)
  torch.cuda.synchronize()
  want_byz, want_out = byz.clone(), eager.clone()
  stream = torch.cuda.Stream()
  stream.wait_stream(torch.cuda.current_stream())
  graph = torch.cuda.CUDAGraph()
  with torch.cuda.stream(stream):
    out_graph = torch.empty_like(eager)
    byz_graph = torch.empty_like(byz)
    with torch.cuda.graph(graph, stream=stream):
      found = bm.stats.attack_search_device(sq, h, f, f, "krum", evals=16)
      out_graph.copy_(found)
      bm.stats.multi_fma3([byz_graph], [avg], [direction], 1.0, found)
  byz_graph.zero_()
  graph.replay()
  torch.cuda.synchronize()
  assert torch.equal(out_graph, want_out) and torch.equal(byz_graph, want_byz)


# ---------------------------------------------------------------------------- #
# The exploration's cursor in device memory (bm_search_device_next)

@pytest.mark.parametrize("negative", [False, True])
def test_device_cursor_proposes_the_candidates_of_the_host_cursor(bm, negative):
  """stats.DeviceSearch against linesearch.line_maximize (itself pinned to tools.line_maximize on 1 500 random scapes,
  tests/test_linesearch_cpu.py) on scapes evaluated with the same IEEE operations on both sides: peaked, monotone,
  flat (nothing ever strictly better), NaN beyond a point."""
  from byzantinemomentum_amd import linesearch
  scapes = [
    (lambda t: 10.0 - (t - 3.7) * (t - 3.7), lambda t: 10.0 - (t - 3.7) * (t - 3.7)),
    (lambda t: t * 0.5 + 1.0, lambda t: t * 0.5 + 1.0),
    (lambda t: t * 0.0 + 2.0, lambda t: t * 0.0 + 2.0),
    (lambda t: 1.0 / (1.0 + (t + 0.25) * (t + 0.25)), lambda t: 1.0 / (1.0 + (t + 0.25) * (t + 0.25))),
    (lambda t: math.nan if abs(t) > 5.0 else abs(t), lambda t: torch.where(t.abs() > 5.0, torch.full_like(t, math.nan), t.abs())),
  ]
  for evals in (1, 2, 7, 16, 40):
    for idx, (on_host, on_device) in enumerate(scapes):
      want_factor, want_trace = linesearch.line_maximize(lambda x: on_host(-x if negative else x), evals=evals)
      cursor = bm.stats.DeviceSearch(torch.device(DEV), evals, negative)
      y = None
      for _ in range(evals):
        y = on_device(cursor.next(y))
      out = cursor.finish(y).cpu().tolist()
      got = (out[0], [(out[1 + 2 * i], out[2 + 2 * i]) for i in range(evals)])
      _same_bits(got, (want_factor, want_trace), (idx, evals, negative))
      with pytest.raises(RuntimeError):
        cursor.next(y)


STEP_RULES = [("median", {}), ("trmean", {}), ("phocas", {}), ("meamed", {}), ("aksel", {}), ("cge", {})]


@pytest.mark.parametrize("gar,opts", STEP_RULES, ids=[r for r, _ in STEP_RULES])
def test_step_with_the_cursor_on_the_device_equals_the_step_with_the_cursor_on_the_host(bm, gar, opts):
  """AggregationStep(line_search="auto") keeps the cursor on the device for every rule but Bulyan / Brute: the same
  kernels are fed the same factors as with the host's cursor ("host"), so candidates, objectives, factor, aggregated
  gradient and study floats are the same bits — and the step queues its search without waiting (the factor is a tensor
  when run() returns)."""
  from byzantinemomentum_amd.step import AggregationStep
  n, f, d = 25, 5, 60013
  h = n - f
  gen = torch.Generator().manual_seed(13)
  modes = ("auto", opts.get("line", "host"))
  steps = {mode: AggregationStep(n, f, f, gar=gar, momentum=0.9, dampening=0.9, momentum_at="worker", attack="little",
                                 nb_past=2, attack_evals=12, attack_negative=(gar in ("phocas", "cge")), line_search=mode)
           for mode in modes}
  origin = torch.randn(d, generator=gen).to(DEV)
  for it in range(3):
    sampled = [(0.2 * torch.randn(d, generator=gen) + (0.5 + 0.1 * i) * torch.randn(d, generator=gen)).to(DEV) for i in range(h)]
    outs = {}
    for mode, step in steps.items():
      outs[mode] = step.run([g.clone() for g in sampled], origin, origin).clone()
      assert isinstance(step._factor_now, torch.Tensor) == (mode == "auto"), (gar, mode)
    a, b = (steps[m] for m in modes)
    assert a.last_search == b.last_search and a.last_factor == b.last_factor, (gar, it, a.last_search, b.last_search)
    assert torch.equal(outs[modes[0]], outs[modes[1]]), (gar, it)
    assert a.floats() == b.floats(), (gar, it)


# ---------------------------------------------------------------------------- #
# ABI 22: the objective of a candidate in one pass (bm_sqdist2); the median's search as the middle of three rows

@pytest.mark.parametrize("d", [0, 1, 3, 63, 1024, 65537, 1000003, 11173962])
def test_sqdist2_against_fp64_and_the_evaluate_only_accumulation(bm, d):
  """bm_sqdist2 = |a - b|^2 (aggregated.sub_(grad_avg); aggregated.dot(aggregated), identical.py:75-76) against fp64
  (1e-6 relative: fp32 squares, 16 per lane, fp64 beyond), repeated launches bit-equal, 4- / 8- / 16-byte aligned views
  within 1e-6 of each other, a NaN answered NaN, and — what the search tests below rest on — the SAME bits as the
  evaluate-only kernel leaves for the rule's output on the same vectors (median of (lo, hi, candidate))."""
  gen = torch.Generator(device=DEV).manual_seed(5 + d % 97)
  a = torch.randn(d + 4, device=DEV, generator=gen)
  b = 0.5 * torch.randn(d + 4, device=DEV, generator=gen)
  want = (a[:d].double() - b[:d].double()).pow(2).sum().item()
  got = bm.stats.sqdist2(a[:d], b[:d])
  assert got.dtype == torch.float64 and got.shape == (1,)
  assert abs(got.item() - want) <= 1e-6 * want + 0.0
  assert torch.equal(got, bm.stats.sqdist2(a[:d], b[:d]))
  if d == 0:
    return
  for off in (1, 2):
    ao, bo = a.clone()[off:off + d], b[:d]
    ao.copy_(a[:d])
    assert abs(bm.stats.sqdist2(ao, bo).item() - want) <= 1e-6 * want
  # the evaluate-only kernel on (lo, hi) = (-inf, +inf): the middle of (lo, hi, candidate) is the candidate
  # avg + t * dir, so its objective is |t * dir|^2 up to the rounding of the fused multiply-add — and with avg = 0 exactly
  # sqdist2(t * dir, 0)
  lo, hi = torch.full((d,), -math.inf, device=DEV), torch.full((d,), math.inf, device=DEV)
  zero, direction = torch.zeros(d, device=DEV), a[:d].clone()
  cand = torch.empty(d, device=DEV)
  bm.stats.multi_fma3([cand], [zero], [direction], 1.0, 0.75)
  fused = bm.stats.colwise_eval("median", [lo, hi], 1, 0, zero, direction, 0.75)
  assert torch.equal(fused, bm.stats.sqdist2(cand, zero)), d
  a[d // 2] = math.nan
  assert math.isnan(bm.stats.sqdist2(a[:d], b[:d]).item())


@pytest.mark.parametrize("n,f,d", [(25, 5, 200003), (11, 4, 65536), (51, 24, 30001)])
def test_median_of_three_evaluate_only_form_against_the_median_kernel(bm, n, f, d):
  """bm_colwise_eval(BM_OP_MEDIAN, [lo, hi], copies = 1) against the median of the n rows with the candidate written out
  (median.py:31-39 on honests + [avg + t * dir] * f, then identical.py:75-76): the same objective bit for bit at several
  factors, from the host's number and from a factor in device memory; other shapes are refused."""
  h = n - f
  gen = torch.Generator(device=DEV).manual_seed(77)
  honests = [0.3 * torch.randn(d, device=DEV, generator=gen) + 0.1 * i for i in range(h)]
  for g in honests:
    g[::7] = g[::7].round()   # ties
  avg = torch.stack(honests).mean(dim=0)
  direction = -avg
  lo = bm.median(honests + [torch.full_like(avg, -math.inf)] * f)
  hi = bm.median(honests + [torch.full_like(avg, math.inf)] * f)
  for t in (0.0, 0.3, 1.0, 2.5, -4.0, 1e6):
    cand = torch.empty_like(avg)
    bm.stats.multi_fma3([cand], [avg], [direction], 1.0, t)
    want = bm.stats.sqdist2(bm.median(honests + [cand] * f), avg)
    got = bm.stats.colwise_eval("median", [lo, hi], 1, 0, avg, direction, t)
    assert torch.equal(got, want), (n, t, got.item(), want.item())
    t_dev = torch.tensor([t], dtype=torch.float64, device=DEV)
    assert torch.equal(bm.stats.colwise_eval("median", [lo, hi], 1, 0, avg, direction, t_dev), want)
  assert not bm.stats.colwise_eval_supported("median", n) and bm.stats.colwise_eval_supported("median", 3)
  with pytest.raises(Exception):
    bm.stats.colwise_eval("median", [lo, hi, lo], 1, 0, avg, direction, 1.0)


@pytest.mark.parametrize("h,k,d", [(1, 1, 1000), (2, 1, 4099), (6, 5, 65537), (11, 4, 300001), (14, 11, 262144), (20, 5, 1000003),
                                   (25, 1, 70001), (26, 25, 50002), (39, 12, 100003), (51, 13, 30001)])
def test_order_pair_equals_the_two_medians_it_replaces(bm, h, k, d):
  """bm_order_pair(honests, (n-1)/2 - k, (n-1)/2) against median(honests + [-inf] * k) and median(honests + [+inf] * k)
  (median.py:31-39): bit-equal — ties, +-inf among the honest values, NaN columns, ranks that fall off either end
  (k > h: the +-inf copies themselves are the median), views that are only 4- and 8-byte aligned, every tail length."""
  n = h + k
  gen = torch.Generator(device=DEV).manual_seed(1000 * h + k)
  honests = [torch.randn(d + 3, device=DEV, generator=gen) + 0.05 * i for i in range(h)]
  for i, g in enumerate(honests):
    g[::5] = g[::5].round()
    g[(7 + i)::97] = math.inf if i % 2 else -math.inf
  honests[0][11] = math.nan
  honests[h - 1][d - 1] = math.nan
  for off in (0, 1, 2):
    rows = [g[off:off + d] for g in honests]
    if n <= 64:
      want_lo = bm.median(rows + [torch.full((d,), -math.inf, device=DEV)] * k)
      want_hi = bm.median(rows + [torch.full((d,), math.inf, device=DEV)] * k)
    else:
      continue
    lo, hi = bm.stats.order_pair(rows, (n - 1) // 2 - k, (n - 1) // 2)
    assert torch.equal(lo.isnan(), want_lo.isnan()) and torch.equal(hi.isnan(), want_hi.isnan()), (h, k, off)
    assert lo.isnan().sum().item() >= 2
    assert torch.equal(lo.nan_to_num(nan=7.0), want_lo.nan_to_num(nan=7.0)), (h, k, off)
    assert torch.equal(hi.nan_to_num(nan=7.0), want_hi.nan_to_num(nan=7.0)), (h, k, off)
  # ranks off both ends
  rows = [g[:d] for g in honests]
  lo, hi = bm.stats.order_pair(rows, -1, h)
  clean = ~torch.stack(rows).isnan().any(dim=0)
  assert bool((lo[clean] == -math.inf).all()) and bool((hi[clean] == math.inf).all())
  lo, hi = bm.stats.order_pair(rows, 0, h - 1)
  stack = torch.stack(rows)
  assert torch.equal(lo[clean], stack.min(dim=0).values[clean]) and torch.equal(hi[clean], stack.max(dim=0).values[clean])
  assert bm.stats.order_pair_supported(51) and not bm.stats.order_pair_supported(52)


# ---------------------------------------------------------------------------- #
# ABI 23: Bulyan's second pass, evaluate only (bm_bulyan_pass2_eval)

@pytest.mark.parametrize("n,f,d", [(11, 2, 65537), (25, 5, 1000003), (25, 5, 4099), (25, 5, 11173962), (51, 12, 200002)])
def test_bulyan_pass2_evaluate_only_form_against_the_written_form(bm, n, f, d):
  """bm_bulyan_pass2_eval against the three launches it replaces — the candidate vector written (bm_multi_fma3), Bulyan's
  second pass on honests + [candidate] * f (bulyan.py:64-84) and |out - avg|^2 (bm_sqdist2, identical.py:75-76) — under
  the ranking of that very stack: the same bits where both sums walk 16-byte groups (n = 11, 25), 1e-6 where the wide
  instance holds 8-byte groups (n = 51) or the views are only 4-byte aligned; the factor from the host and from device
  memory; a NaN column answered NaN by both; shapes without an instance refused."""
  from byzantinemomentum_amd import _lib, gars
  h, m = n - f, n - f - 2
  gen = torch.Generator(device=DEV).manual_seed(97 + n)
  base = 0.2 * torch.randn(d + 1, device=DEV, generator=gen)
  store = [base + (0.5 + 0.05 * i) * torch.randn(d + 1, device=DEV, generator=gen) for i in range(h)]
  assert bm.stats.bulyan_pass2_eval_supported(n, f, m, d) and not bm.stats.bulyan_pass2_eval_supported(n, f, m - 1, d)
  assert not bm.stats.bulyan_pass2_eval_supported(n + 4, f, m + 4, d) and not bm.stats.bulyan_pass2_eval_supported(n, f, m, (1 << 29) + 1)
  for off in (0, 1):
    honests = [g[off:off + d] for g in store]
    avg = torch.stack(honests).mean(dim=0)
    direction = -avg if n != 25 else torch.stack(honests).var(dim=0).sqrt_()
    exact = off == 0 and n in (11, 25)
    for t in (0.0, 0.6, 1.1, 7.5, -3.0):
      cand = torch.empty_like(avg)
      bm.stats.multi_fma3([cand], [avg], [direction], 1.0, t)
      rows = honests + [cand] * f
      gars.invalidate_rank_cache()
      order, _ = gars._rank(rows, f, m, _lib.RANK_BULYAN)
      want = bm.stats.sqdist2(gars.bulyan_pass2(rows, order, f, m), avg)
      got = bm.stats.bulyan_pass2_eval(honests, f, order, f, m, avg, direction, t)
      got_dev = bm.stats.bulyan_pass2_eval(honests, f, order, f, m, avg, direction,
                                           torch.tensor([t], dtype=torch.float64, device=DEV))
      assert torch.equal(got, got_dev), (n, t)
      if exact:
        assert torch.equal(got, want), (n, off, t, got.item(), want.item())
      else:
        assert abs(got.item() - want.item()) <= 1e-6 * want.item() + 1e-30, (n, off, t, got.item(), want.item())
      assert want.item() > 0 or t == 0.0
  honests[1][5] = math.nan
  avg2 = torch.stack([g.nan_to_num() for g in honests]).mean(dim=0)
  cand = torch.empty_like(avg2)
  bm.stats.multi_fma3([cand], [avg2], [direction], 1.0, 0.9)
  rows = honests + [cand] * f
  gars.invalidate_rank_cache()
  honests[1][5] = 0.0
  order, _ = gars._rank(honests + [cand] * f, f, m, _lib.RANK_BULYAN)   # (a ranking from finite rows: the NaN is the second pass's business)
  honests[1][5] = math.nan
  want = bm.stats.sqdist2(gars.bulyan_pass2(rows, order, f, m), avg2)
  got = bm.stats.bulyan_pass2_eval(honests, f, order, f, m, avg2, direction, 0.9)
  assert math.isnan(want.item()) == math.isnan(got.item())
  with pytest.raises(Exception):
    bm.stats.bulyan_pass2_eval(honests[:-1], f, order, f, m, avg2, direction, 0.9)   # n - 1 rows: no instance


# ---------------------------------------------------------------------------- #
# ABI 23: the ranking of a candidate stack on the device (bm_attack_ranking_device)

def _rankings(bm, ext_dev, h, k, f, mode, m, ts, tag):
  from byzantinemomentum_amd import linesearch
  host_ext = ext_dev.cpu().contiguous()
  n = h + k
  for t in ts:
    t_dev = torch.tensor([t], dtype=torch.float64, device=DEV)
    got = bm.stats.attack_ranking_device(ext_dev, h, k, f, mode, t_dev, m).cpu().tolist()
    want = linesearch.attack_ranking(host_ext, h, k, f, mode, t, m)
    assert got[:n] == want and got[n:] == [0] * (64 - n), (tag, mode, t, got, want)


@pytest.mark.parametrize("kind", ["hetero", "tight", "duplicates"])
@pytest.mark.parametrize("shape", [s for s in SHAPES if s[1] >= 1], ids=lambda s: "h%d-k%d-f%d-m%s-e%d" % s)
def test_device_ranking_equals_host_ranking_on_stacks(bm, shape, kind):
  """bm_attack_ranking_device against bm_attack_ranking (linesearch.cpp; itself the ranking of the rank kernel on the
  materialised stack, tests/test_gpu_parity_r3.py): the whole permutation, Krum's scores and Bulyan's (the m smallest
  distances, bulyan.py:48-62), factors of both signs, zero, tiny and huge — stacks with exact ties included."""
  h, k, f, m, _ = shape
  ext = _ext_of_stack(bm, h, 20011, 7 * h + k, kind)
  for mode in ("krum", "bulyan"):
    _rankings(bm, ext, h, k, f, mode, m, (0.0, 0.3, 1.0, 1.1, -2.5, 17.0, 1e-9, 32767.0), (shape, kind))


def test_device_ranking_equals_host_ranking_on_adversarial_matrices(bm):
  """Integer matrices (ties in every row and among the scores), att = 0, a huge |att|^2, asymmetric low bits: the device
  ranking is the host's, rank by rank."""
  gen = torch.Generator().manual_seed(199)
  cases = 0
  for h, k, f in ((7, 2, 2), (20, 5, 5), (39, 12, 12), (50, 14, 14), (63, 1, 1)):
    e = h + 2
    for trial in range(9):
      a = torch.randint(0, 6, (e, e), generator=gen).double()
      ext = a + a.t()
      ext.fill_diagonal_(0.0)
      if trial % 3 == 1:
        ext[h, h + 1] = ext[h + 1, h] = 0.0
      if trial % 3 == 2:
        ext[h, h + 1] = ext[h + 1, h] = 1e12
      if trial >= 6:
        ext = ext * (0.5 + torch.rand(e, e, generator=gen).double())
      dev = ext.contiguous().to(DEV)
      for mode in ("krum", "bulyan"):
        _rankings(bm, dev, h, k, f, mode, None, (0.0, 0.5, -1.0, 3.0), (h, k, f, trial))
        cases += 1
  assert cases == 90


def test_bulyan_search_with_the_cursor_on_the_device_matches_the_host_cursor(bm):
  """AggregationStep against Bulyan: line_search="auto" ranks every candidate on the device from the factor the device
  cursor left there and evaluates pass 2 in place (no synchronisation in the search), "host" ranks on the host from the
  host cursor's number — same candidates, same rankings, hence the same objectives bit for bit, the same factor and the
  same aggregated gradient; n = 25 (evaluate-only pass 2) and n = 15 (no instance: the written form)."""
  from byzantinemomentum_amd.step import AggregationStep
  for n, f, d, attack, negative in ((25, 5, 200003, "empire", False), (15, 3, 65537, "little", True), (51, 12, 40001, "empire", False)):
    h = n - f
    gen = torch.Generator(device=DEV).manual_seed(41 + n)
    base = 0.2 * torch.randn(d, device=DEV, generator=gen)
    honests = [base + (0.5 + 0.05 * i) * torch.randn(d, device=DEV, generator=gen) for i in range(h)]
    traces = {}
    for mode in ("auto", "host"):
      step = AggregationStep(n, f, f, gar="bulyan", momentum=0.9, dampening=0.9, momentum_at="update", attack=attack,
                             attack_factor=1.1, nb_past=0, attack_evals=12, attack_negative=negative, line_search=mode)
      out = step.run([g.clone() for g in honests])
      if mode == "auto":
        assert isinstance(step._factor_now, torch.Tensor), "the device cursor must leave its factor on the device"
      traces[mode] = (step.last_factor, list(step.last_search), out)
    (fa, sa, oa), (fh, sh, oh) = traces["auto"], traces["host"]
    assert sa == sh and fa == fh and len(sa) == 12, (n, sa, sh)
    assert torch.equal(oa, oh), n
