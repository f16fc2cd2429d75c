"""The RCCL code path of the multi-GPU runs on ONE GPU: a one-rank NCCL(=RCCL) process group with forced collectives
(libbm_gar's own communicator inside the single-call rules, the torch.distributed form of the same rules, the
all-to-all of the worker-parallel layout, a whole step).  Collected AFTER the parity files on purpose: it depends on
sockets and a process group, and a hiccup there must not hide the parity tests from a `-x` run.  SURVEY.md 8e / f4.
"""

import math

import pytest
import torch

from oracle import gar_oracle as O
from tests.test_gpu_parity import DEV, bm, to_dev  # noqa: F401  (bm: the module-scoped fixture of the parity file)

pytestmark = pytest.mark.gpu


def test_rccl_path_on_one_gpu(bm):
  """One-rank NCCL(=RCCL) process group with forced collectives: the exact code path of the
  multi-GPU runs (all-reduce of the fp64 distance matrix, all-gather of the output)."""
  import torch.distributed as dist
  from byzantinemomentum_amd.sharded import ShardedAggregator
  import os
  import socket
  with socket.socket() as s:
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
  os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
  dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device(DEV))
  try:
    rows, h = O.make_stack("hetero", 25, 5, 40007, seed=18)
    dev = to_dev(rows)
    agg = ShardedAggregator(force_collectives=True)
    assert agg.collective
    assert agg.native is not None and agg.single_call   # libbm_gar's own RCCL communicator, one C call per rule
    out = agg.bulyan(dev, 5)
    assert torch.equal(out, bm.bulyan(dev, 5))
    assert torch.equal(agg.all_gather_output(out, 40007), out)
    assert torch.equal(agg.krum(dev, 5), bm.krum(dev, 5))
    want = bm.compute_avg_dev_max(dev[:h])
    got = agg.compute_avg_dev_max(dev[:h])
    assert torch.equal(got[0], want[0]) and got[1:] == want[1:]
    # the torch.distributed form of the same rules (native_comm=False) must give the same bits
    plain = ShardedAggregator(force_collectives=True, native_comm=False)
    assert plain.native is None and not plain.single_call
    assert torch.equal(plain.bulyan(dev, 5), out) and torch.equal(plain.krum(dev, 5), bm.krum(dev, 5))
    # worker-major -> dimension-major: the all-to-all really goes through RCCL here (one rank, forced
    # collectives); with one rank the layout it returns is every gradient restricted to [0, d): the inputs themselves,
    # as contiguous 256-byte aligned views of ONE receive buffer, accepted as they are by the rules
    # ONE aggregator serves every length: the shards carry the length of the whole vectors (sharded.Shards)
    for d_odd in (40007, 64, 1):
      grads = [g[:d_odd].contiguous() for g in dev[:7]]
      local = agg.to_dim_sharded(grads, 7, d_odd)
      assert local.d_total == d_odd
      assert len(local) == 7 and all(torch.equal(a, b) for a, b in zip(local, grads))
      assert all(t.is_contiguous() and t.data_ptr() % 256 == 0 for t in local)
      assert local[0].untyped_storage().data_ptr() == local[6].untyped_storage().data_ptr()
      assert torch.equal(agg.median(local), bm.median(grads)) and torch.equal(agg.krum(local, 1), bm.krum(grads, 1))
      assert torch.equal(agg.bulyan(local, 1), bm.bulyan(grads, 1)) and torch.equal(agg.brute(local, 1), bm.brute(grads, 1))
      assert torch.equal(plain.krum(local, 1), bm.krum(grads, 1))
    # plain lists of another length: stated total, or an error that says what to do — never a guess
    short = [g[:4096].contiguous() for g in dev[:7]]
    assert torch.equal(agg.krum(short, 1, d_total=4096), bm.krum(short, 1))
    with pytest.raises(ValueError, match="d_total"):
      agg.krum(short, 1)
    assert torch.equal(agg.krum(agg.shard_rows(short), 1), bm.krum(short, 1))
    # a whole step through forced collectives equals the step without any
    from byzantinemomentum_amd.step import AggregationStep
    a = AggregationStep(25, 5, 5, gar="bulyan", nb_past=2, aggregator=agg)
    b = AggregationStep(25, 5, 5, gar="bulyan", nb_past=2)
    for it in range(3):
      sampled = [g * (1.0 + 0.1 * it) for g in dev[:h]]
      assert torch.equal(a.run(sampled), b.run([g.clone() for g in sampled]))
      fa, fb = a.floats(), b.floats()
      assert all(fa[k] == fb[k] or (math.isnan(fa[k]) and math.isnan(fb[k])) for k in fa)
  finally:
    dist.destroy_process_group()
