"""HIP graphs of host-mirror calls, for the launch-bound regime.

On one GPU a rule is two or three kernels of 150-400 us each and the launches hide behind them.  A rank of an
8-GPU job runs the same sequence on 1/8 of the coordinates: at C4 (Bulyan, n = 25, d = 11.2 M / 8) six launches
of 5-25 us each plus the n x n all-reduce, 73-78 us of which about 30 are gaps between dependent launches
(profiles/r03_e_per_rank_p8.txt).  When the row buffers of the stack stay where they are from one
aggregation to the next — momentum buffers do (attack.py:676), and a loop that writes its gradients into fixed
buffers has the same property — the whole sequence, collective included, can be recorded once into a HIP graph and
replayed: one submission per aggregation, the kernel arguments (the by-value row table of every kernel, include/
bm_gar.h) frozen at the recorded addresses.

    g = GraphedCall(lambda: aggregator.bulyan(rows, f))   # records: rows are the buffers every replay will read
    out = g()                                             # replays on the current stream; `out` is the SAME tensor
                                                          # each time (consume or copy it before the next replay)

What a replay does NOT do is run the Python of the call again: the shapes, the rule's arguments and the addresses
are those of the recording.  Contents may change freely; a stack at other addresses needs its own GraphedCall.

`fn` must not synchronise with the host while it is recorded.  Calls that do, and therefore cannot be graphed: the
attacks' factor search against Brute and any search with `line_search="host"` / `"generic"` (a read per evaluation; with
`"auto"` the searches against every other rule keep cursor, ranking and factor on the device), `floats()`, sharded
rules over torch.distributed collectives that stage through the host, and `ShardedAggregator.brute` with a backend
that has no device search (the HIP backend has one, bm_brute_select_device: its brute IS capturable; the status of
the search stays on the device until `check_brute()` / `floats()`).  A failed recording raises GraphCaptureError and
leaves nothing behind.
"""

import torch

from . import gars

__all__ = ["GraphedCall", "GraphCaptureError"]


class GraphCaptureError(RuntimeError):
  """`fn` could not be recorded into a HIP graph (it synchronised with the host, or a launch failed under capture)."""


class GraphedCall:
  """`fn()` (any call of this package whose tensors live on the current GPU) recorded into a HIP graph."""

  def __init__(self, fn, warmup=2):
    if not torch.cuda.is_available():
      raise RuntimeError("GraphedCall records HIP graphs: it needs the GPU the rows live on")
    self._graph = torch.cuda.CUDAGraph()
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
      # everything the call creates lazily (scratch buffers, the LDS size attribute of a kernel, RCCL channels)
      # must exist before the recording; the ranking cache of gars.py must not turn the recorded call into
      # "reuse the last ranking" (the graph would then hold the averaging kernel alone)
      for _ in range(max(int(warmup), 1)):
        gars.invalidate_rank_cache()
        fn()
    torch.cuda.current_stream().wait_stream(side)
    torch.cuda.synchronize()
    gars.invalidate_rank_cache()
    # thread_local: RCCL's helper threads may call the runtime while this thread records
    try:
      with torch.cuda.graph(self._graph, stream=side, capture_error_mode="thread_local"):
        self.output = fn()
    except Exception as err:  # noqa: BLE001
      # nothing of the failed recording may survive: the scratch entries the warm-up created for the side stream (a
      # later stream that maps to the same handle would find them), the ranking cache, the half-built graph
      for key in [k for k in gars._Scratch._cache if k[1] == side.cuda_stream]:
        del gars._Scratch._cache[key]
      gars.invalidate_rank_cache()
      self._graph = None
      torch.cuda.synchronize()
      raise GraphCaptureError(
        "the call could not be recorded into a HIP graph: it must not synchronise with the host while recording "
        "(the attacks' factor search, floats(), host-staged collectives and the sharded brute rule do; see the module "
        f"docstring) — {type(err).__name__}: {err}") from err
    gars.invalidate_rank_cache()  # (the entry the recording left points into the graph's private pool)
    # the scratch buffers the recorded kernels use belong to gars._Scratch, keyed by (device, stream): a later call on
    # a stream that maps to the same handle with another size would replace — and free — them under the graph
    self._scratch = [buf for key, buf in gars._Scratch._cache.items() if key[1] == side.cuda_stream]
    self._stream = side

  def __call__(self):
    self._graph.replay()
    return self.output
