import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
  sys.path.insert(0, ROOT)


def pytest_configure(config):
  config.addinivalue_line("markers", "gpu: needs a real MI355X (run with `-m gpu` on the GPU box)")
  config.addinivalue_line("markers", "reference: needs the read-only reference checkout at /root/reference")


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
