import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
  config.addinivalue_line("markers", "reference: needs the read-only reference checkout at /root/reference")


def pytest_sessionstart(session):
  """Make sure the in-tree HIP library exists and is current before any test imports it
  (incremental: a no-op when libbm_gar.so is newer than its sources; needs only hipcc)."""
  from byzantinemomentum_amd import build
  try:
    build.build()
  except Exception as err:  # the ABI tests will then fail loudly with the loader's message
    print(f"[conftest] could not build libbm_gar.so: {err}", file=sys.stderr)


def pytest_collection_modifyitems(config, items):
  import torch
  from oracle import reference_loader
  skip_gpu = pytest.mark.skip(reason="no GPU visible")
  skip_ref = pytest.mark.skip(reason="reference checkout not present (GPU box)")
  has_gpu = torch.cuda.is_available()
  has_ref = reference_loader.available()
  for item in items:
    if "gpu" in item.keywords and not has_gpu:
      item.add_marker(skip_gpu)
    if "reference" in item.keywords and not has_ref:
      item.add_marker(skip_ref)
