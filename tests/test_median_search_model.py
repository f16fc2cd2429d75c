"""The median's own factor search (byzantinemomentum_amd/step.py, csrc/search_eval.hip) rests on one identity, pinned here
without a GPU against the reference's own operation (aggregators/median.py:31-39: torch.stack(...).median(dim=0)):

    median(honests + [b] * k)  ==  middle of (lo, hi, b)        per coordinate, bit for bit,
    lo = the honest value of rank (n-1)/2 - k   (-inf below rank 0)       = median(honests + [-inf] * k)
    hi = the honest value of rank (n-1)/2       (+inf beyond rank h - 1)  = median(honests + [+inf] * k)

with n = h + k and torch's LOWER median — for every split of n = 1 .. 64 the search can meet, candidates below, between,
above and ON honest values, ties among the honest values, infinite honest values, and NaN columns (torch.median answers
NaN whenever the column holds one: both sides must).  `order_pair` below is the model of bm_order_pair: sort, pick two
ranks, NaN if the column holds one."""

import math

import pytest
import torch


def order_pair(stack, il, ih):
  """Model of bm_order_pair on an (h, d) tensor: the values of rank il / ih per column, -inf / +inf off the ends, NaN
  where the column holds one."""
  h, d = stack.shape
  srt = stack.sort(dim=0).values  # (NaN sorts last; those columns are overwritten below)
  lo = srt[il] if 0 <= il < h else torch.full((d,), -math.inf if il < 0 else math.inf)
  hi = srt[ih] if 0 <= ih < h else torch.full((d,), math.inf if ih >= h else -math.inf)
  bad = stack.isnan().any(dim=0)
  nan = torch.full((d,), math.nan)
  return torch.where(bad, nan, lo), torch.where(bad, nan, hi)


def same_bits(a, b):
  return torch.equal(a.isnan(), b.isnan()) and torch.equal(a.nan_to_num(nan=3.0), b.nan_to_num(nan=3.0))


@pytest.mark.parametrize("h,k", [(1, 1), (2, 1), (1, 2), (3, 2), (6, 5), (9, 2), (14, 11), (20, 5), (26, 25), (39, 12), (51, 13),
                                 (62, 2), (33, 31), (5, 40)])
def test_median_of_honests_and_copies_is_the_middle_of_two_order_statistics_and_the_candidate(h, k):
  n = h + k
  d = 4096
  gen = torch.Generator().manual_seed(100 * h + k)
  stack = torch.randn(h, d, generator=gen)
  stack[:, ::5] = stack[:, ::5].round()                      # ties among the honest values
  stack[torch.randint(0, h, (64,), generator=gen), torch.randint(0, d, (64,), generator=gen)] = math.inf
  stack[torch.randint(0, h, (64,), generator=gen), torch.randint(0, d, (64,), generator=gen)] = -math.inf
  stack[0, 7] = math.nan
  stack[h - 1, 11] = math.nan
  m = (n - 1) // 2
  lo, hi = order_pair(stack, m - k, m)
  # the two vectors ARE the medians with the copies at -inf / +inf (what step.py computed with two median calls before ABI 22)
  for fill, got in ((-math.inf, lo), (math.inf, hi)):
    want = torch.cat([stack, torch.full((k, d), fill)]).median(dim=0).values
    assert same_bits(got, want), (h, k, fill)
  # and the rule at any candidate is the middle of the three
  srt = stack.nan_to_num(nan=0.0).sort(dim=0).values
  candidates = [torch.randn(d, generator=gen) * s for s in (0.1, 1.0, 30.0)]
  candidates += [srt[min(m, h - 1)].clone(), srt[max(m - k, 0)].clone(), torch.full((d,), math.inf), torch.full((d,), -math.inf),
                 torch.zeros(d)]
  for b in candidates:
    b = b.clone()
    b[b.isnan()] = 0.0
    want = torch.cat([stack, b.expand(k, d)]).median(dim=0).values
    got = torch.stack([lo, hi, b]).median(dim=0).values
    assert same_bits(got, want), (h, k)
    # the candidate itself NaN in a column: NaN there on both sides
    b2 = b.clone()
    b2[3] = math.nan
    assert same_bits(torch.stack([lo, hi, b2]).median(dim=0).values, torch.cat([stack, b2.expand(k, d)]).median(dim=0).values)
