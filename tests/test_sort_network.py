"""The compile-time merge-exchange network of the column kernels (csrc/bm_common.h, MergeExchange<N>): the very
comparator tables the kernels unroll — dumped by a host-only compilation of the header — sort every input, for every
row count 1..64.  Zero-one principle: a comparator network sorts all inputs iff it sorts all 0/1 inputs; exhaustive up
to 16 rows, beyond that every 0/1 vector with at most three ones or at most three zeros plus 2^17 random ones per
density sweep, and random floats with ties, infinities and signed zeros.  (aggregators/median.py:39, trmean.py:33: the
reference sorts with torch; the kernels' results are compared with it on the GPU.)"""

import itertools
import pathlib
import shutil
import subprocess

import numpy as np
import pytest

ROOT = pathlib.Path(__file__).resolve().parent.parent

DUMP = r'''
#include <cstdio>
#include "bm_common.h"
template <int N>
void dump() {
  const auto& t = bm::MergeExchange<N>::table;
  printf("%d %d", N, t.count);
  for (int i = 0; i < t.count; ++i) printf(" %d %d", (int)t.a[i], (int)t.b[i]);
  printf("\n");
  if constexpr (N < 64) dump<N + 1>();
}
int main() { dump<1>(); return 0; }
'''


@pytest.fixture(scope="module")
def tables(tmp_path_factory):
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not pathlib.Path(hipcc).exists():
    pytest.skip("hipcc not here")
  tmp = tmp_path_factory.mktemp("network")
  (tmp / "dump.cpp").write_text(DUMP)
  subprocess.run([hipcc, "-std=c++17", "-O1", "-x", "hip", "--offload-host-only", "-I",
                  str(ROOT / "byzantinemomentum_amd" / "csrc"), str(tmp / "dump.cpp"), "-o", str(tmp / "dump")],
                 check=True, capture_output=True)
  out = subprocess.run([str(tmp / "dump")], check=True, capture_output=True, text=True).stdout
  nets = {}
  for line in out.strip().splitlines():
    nums = list(map(int, line.split()))
    n, count = nums[0], nums[1]
    pairs = list(zip(nums[2::2], nums[3::2]))
    assert len(pairs) == count
    nets[n] = pairs
  assert sorted(nets) == list(range(1, 65))
  return nets


def _apply(pairs, x):
  x = x.copy()
  for a, b in pairs:
    lo = np.minimum(x[:, a], x[:, b])
    hi = np.maximum(x[:, a], x[:, b])
    x[:, a], x[:, b] = lo, hi
  return x


def _sorted_rows(x):
  return bool((x[:, :-1] <= x[:, 1:]).all()) if x.shape[1] > 1 else True


def test_comparators_are_well_formed(tables):
  for n, pairs in tables.items():
    assert all(0 <= a < b < n for a, b in pairs), n
    assert len(pairs) <= max(1, 10 * n)  # the kMax bound of the constexpr table
  assert len(tables[25]) < 25 * 24 // 2 and len(tables[51]) < 51 * 50 // 2  # (far fewer than a bubble network)


def test_networks_sort_every_zero_one_input_up_to_16_rows(tables):
  for n in range(1, 17):
    x = ((np.arange(1 << n, dtype=np.uint32)[:, None] >> np.arange(n, dtype=np.uint32)[None, :]) & 1).astype(np.int8)
    assert _sorted_rows(_apply(tables[n], x)), n


def test_networks_sort_sparse_dense_and_random_zero_one_inputs_up_to_64_rows(tables):
  rng = np.random.default_rng(9)
  for n in range(17, 65):
    rows = []
    for k in range(0, 4):  # every vector with at most three ones, and with at most three zeros
      for idx in itertools.combinations(range(n), k):
        v = np.zeros(n, dtype=np.int8)
        v[list(idx)] = 1
        rows.append(v)
        rows.append(1 - v)
    x = np.stack(rows)
    assert _sorted_rows(_apply(tables[n], x)), n
    dens = rng.random((1 << 17, 1))
    x = (rng.random((1 << 17, n)) < dens).astype(np.int8)
    assert _sorted_rows(_apply(tables[n], x)), n


def test_networks_sort_floats_with_ties_infinities_and_signed_zeros(tables):
  rng = np.random.default_rng(10)
  for n in (1, 2, 3, 11, 25, 37, 51, 64):
    x = rng.standard_normal((4096, n)).astype(np.float32)
    x[:1024] = np.round(x[:1024] * 2) / 2                      # many ties
    x[1024:1100, : max(1, n // 3)] = np.inf
    x[1100:1200, -max(1, n // 4):] = -np.inf
    x[1200:1300, ::2] = 0.0
    x[1200:1300, 1::2] = -0.0
    got = _apply(tables[n], x)
    assert np.array_equal(got, np.sort(x, axis=1)), n           # (-0.0 == 0.0 compare equal: any order of the two passes)
    # the rules read ranks of the sorted column: lower median (n-1)//2 (median.py:39), ranks f..n-f-1 (trmean.py:33)
    assert np.array_equal(got[:, (n - 1) // 2], np.sort(x, axis=1)[:, (n - 1) // 2])


TRI = r'''
#include <cstdio>
#include "gram_split.h"
int main() {
  for (int n = 1; n <= 64; ++n) {
    printf("%d", n);
    for (int i = 0; i < n; ++i)
      for (int j = i; j < n; ++j) printf(" %d", bm::b3_tri_index(i, j, n));
    printf("\n");
  }
  // launch-shape helpers of bm_common.h
  printf("grid %d %d %d %d\n", bm::stream_grid(0, 256, 16384), bm::stream_grid(1, 256, 16384),
         bm::stream_grid(257, 256, 16384), bm::stream_grid((int64_t)1 << 40, 256, 16384));
  const void* p[3] = {(void*)0x1000, (void*)0x2010, (void*)0x3020};
  const void* q[2] = {(void*)0x1008, (void*)0x2010};
  const void* r[2] = {(void*)0x1004, (void*)0x2010};
  printf("vec %d %d %d %d\n", bm::common_vec_width(p, 3, nullptr), bm::common_vec_width(q, 2, nullptr),
         bm::common_vec_width(r, 2, nullptr), bm::common_vec_width(p, 3, (void*)0x4004));
  return 0;
}
'''


def test_triangle_index_and_launch_helpers_of_the_headers(tmp_path):
  """b3_tri_index (csrc/gram_split.h) — where the Gram kernel, its reduction and the distance kernel all look up entry
  (i, j) of the compact upper triangle — is a bijection onto 0 .. n(n+1)/2 - 1 in row-major order for every n <= 64;
  stream_grid and common_vec_width (csrc/bm_common.h) answer as documented.  Host-only compilation of the headers."""
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not pathlib.Path(hipcc).exists():
    pytest.skip("hipcc not here")
  (tmp_path / "tri.cpp").write_text(TRI)
  subprocess.run([hipcc, "-std=c++17", "-O1", "-x", "hip", "--offload-host-only", "-I",
                  str(ROOT / "byzantinemomentum_amd" / "csrc"), str(tmp_path / "tri.cpp"), "-o", str(tmp_path / "tri")],
                 check=True, capture_output=True)
  lines = subprocess.run([str(tmp_path / "tri")], check=True, capture_output=True, text=True).stdout.strip().splitlines()
  for line in lines[:64]:
    nums = list(map(int, line.split()))
    n, idx = nums[0], nums[1:]
    assert idx == list(range(n * (n + 1) // 2)), n   # row-major over i <= j: consecutive, no gap, no repeat
  assert lines[64].split() == ["grid", "1", "1", "2", "16384"]
  assert lines[65].split() == ["vec", "4", "2", "1", "1"]


DITHER = r'''
#include <cstdio>
#include "gram_split.h"
int main() {
  const unsigned coords[] = {0u, 2u, 4u, 62u, 64u, 1000u, 11173960u, 36546978u, 0x7ffffffeu, 0xfffffffeu};
  for (unsigned c : coords) printf("%u %u\n", c, bm::dither_pair(c));
  return 0;
}
'''


def test_numpy_model_of_the_dither_is_the_kernels_hash(tmp_path):
  """dither_pair (csrc/gram_split.h: two 16-bit dither words for coordinates c, c + 1 from one mix of the even
  coordinate index) compiled for the host, against dither16 of scripts/probes/dither_model.py — the model on which
  tests/test_split_model.py checks the split's properties is the kernel's own hash."""
  import importlib.util
  hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
  if not pathlib.Path(hipcc).exists():
    pytest.skip("hipcc not here")
  (tmp_path / "dither.cpp").write_text(DITHER)
  subprocess.run([hipcc, "-std=c++17", "-O1", "-x", "hip", "--offload-host-only", "-I",
                  str(ROOT / "byzantinemomentum_amd" / "csrc"), str(tmp_path / "dither.cpp"), "-o", str(tmp_path / "dither")],
                 check=True, capture_output=True)
  out = subprocess.run([str(tmp_path / "dither")], check=True, capture_output=True, text=True).stdout.split()
  spec = importlib.util.spec_from_file_location("dither_model", ROOT / "scripts" / "probes" / "dither_model.py")
  model = importlib.util.module_from_spec(spec)
  spec.loader.exec_module(model)
  pairs = list(zip(map(int, out[0::2]), map(int, out[1::2])))
  assert len(pairs) == 10
  for coord, z in pairs:
    lo, hi = z & 0xFFFF, z >> 16
    got = model.dither16(np.array([coord, coord + 1], dtype=np.uint64))
    assert [int(got[0]), int(got[1])] == [lo, hi], coord
