"""Placement of gradient rows in HBM.

The kernels of this package read n rows (n = 25 ... 64) at the same column offset at the same time: n concurrent
streams.  How fast the memory system serves them depends on where the rows lie relative to each other: the same
kernel, same data, ran the first pass of a C5 step in 1 527 ... 1 788 us and the 51-row distance pass in 395 ... 447 us
depending only on the placement of its row buffers (profiles/r03_f_momentum_layout_probe.txt,
r03_g_layout_probe.txt).  Separately allocated tensors land wherever the caching allocator's history puts them;
rows cut out of ONE allocation at a stride of (a multiple of 2 MB) + 4 352 bytes were at or near the best time in
every run measured: consecutive rows are shifted by 17 x 256 bytes in the channel-interleave pattern, so the n
streams of a column window fall on different HBM channels instead of piling onto a few.  Successive calls continue
the sequence of shifts (the rows of two stacks that one kernel reads together do not line up pairwise either).

`alloc_rows` is what `bench.py` uses for the stacks of its rule benchmarks (C2: +3 % in A/B on one box; C3, C4
unchanged); a training loop that flattens each worker's gradient into a vector (the reference's
`model.get_gradient()`, experiments/model.py:333-366) can flatten into these rows instead.  The write-heavy first pass
of the step did NOT gain from it (0.3-5 % slower than separate allocations in every A/B), so `AggregationStep` keeps
one allocation per momentum buffer.  Rows are ordinary
contiguous 1-D float32 views: every rule accepts them, and tensors allocated any other way work as before.
"""

import torch

__all__ = ["alloc_rows", "ROW_SKEW_BYTES"]

ROW_SKEW_BYTES = 4352          # 17 x 256 B: coprime with any power-of-two interleave of 256-byte granules
_ALIGN = 2 << 20


_rows_handed_out = 0  # successive allocations continue the skew sequence, so that the rows of two stacks a kernel reads
                       # together (sampled gradients and momentum buffers, say) do not line up pairwise either


def alloc_rows(count, d, device, dtype=torch.float32, skew=ROW_SKEW_BYTES, zero=False):
  """`count` vectors of `d` elements as views of one allocation, row i starting at (i0 + i) * skew bytes past a
  multiple of the row pitch (d * itemsize rounded up to 2 MB — 256 B for rows under 1 MB), from a 256-byte aligned
  base; i0 counts the rows handed out before (modulo the pitch alignment).  The views keep the allocation alive."""
  global _rows_handed_out
  if count < 1 or d < 0:
    raise ValueError("alloc_rows needs count >= 1 and d >= 0")
  item = torch.empty((), dtype=dtype).element_size()
  if skew % 256 != 0:
    raise ValueError("skew must be a multiple of 256 bytes (rows stay aligned for 16-byte loads)")
  big = d * item >= (1 << 20)
  align = _ALIGN if big else 256  # short rows: no point in spending 2 MB on each
  window = _ALIGN if big else 0    # where the first row may start inside the allocation
  phase = (_rows_handed_out * skew) % window if window else 0
  _rows_handed_out += count
  stride = ((d * item + align - 1) // align * align + skew) // item
  make = torch.zeros if zero else torch.empty
  slab = make(stride * (count - 1) + d + (256 + window) // item, dtype=dtype, device=device)
  base = ((-slab.data_ptr() % 256) + phase) // item
  return [slab[base + i * stride: base + i * stride + d] for i in range(count)]
