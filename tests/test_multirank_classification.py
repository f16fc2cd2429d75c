"""How tests/test_gpu_zz_multirank.py classifies what its rank processes report (no GPU: `_run_ranks` is replaced):
the signature of a read served stale under GPU sharing is an EXPECTED failure (xfail, reported with everything the
ranks found), any other mismatch fails, a lost rank is repeated."""

import pytest

import tests.test_gpu_zz_multirank as M

the_test = M.test_multi_rank_sharded_path_on_the_hip_kernels


def _outcomes(monkeypatch, sequence):
  calls = iter(sequence)
  monkeypatch.setattr(M, "_run_ranks", lambda world, d: next(calls))


PEER = "RuntimeError: Connection closed by peer"
GOOD = ({0: {"shard": (0, 1), "x": 1.0}, 1: {"shard": (1, 2), "x": 1.0}}, [0, 0], "")


def test_transient_signature_is_an_expected_failure(monkeypatch):
  stale = "Traceback ...\nAssertionError: TRANSIENT-STALE-READ hetero bulyan: 281 coordinates of [0, 100) differ ..."
  _outcomes(monkeypatch, [({0: {"error": stale}, 1: {"error": PEER}}, [1, 1], "--- stderr of rank 0 ---\n" + stale)])
  with pytest.raises(pytest.xfail.Exception, match="served stale"):
    the_test(2, 200003)


def test_any_other_mismatch_fails(monkeypatch):
  wrong = "Traceback ...\nAssertionError: hetero bulyan: 90000 coordinates of [0, 100000) differ ..."
  _outcomes(monkeypatch, [({0: {"error": wrong}, 1: {"error": PEER}}, [1, 1], "")])
  with pytest.raises(AssertionError, match="90000 coordinates"):
    the_test(2, 200003)
  both = "AssertionError: TRANSIENT-STALE-READ step 2 ..."
  _outcomes(monkeypatch, [({0: {"error": both}, 1: {"error": wrong}}, [1, 1], "")])
  with pytest.raises(AssertionError, match="90000 coordinates"):
    the_test(2, 200003)


def test_a_lost_rank_is_repeated_and_a_clean_run_passes(monkeypatch):
  _outcomes(monkeypatch, [({1: {"error": PEER}}, [-11, 1], "--- stderr of rank 0 ---\nSegmentation fault"), GOOD])
  with pytest.warns(UserWarning, match="lost a rank"):
    the_test(2, 200003)
  _outcomes(monkeypatch, [GOOD])
  the_test(2, 200003)
