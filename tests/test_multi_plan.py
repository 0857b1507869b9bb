"""The pair split of the multi-device context (ndt2d_multi_plan): host logic, no device."""
import numpy as np
import pytest

from gtsam_ndt_amd import matcher as M
from gtsam_ndt_amd._lib import NdtError


def _offsets(sizes):
    off = np.zeros(len(sizes) + 1, dtype=np.uint64)
    off[1:] = np.cumsum(sizes)
    return off


def test_equal_pairs_split_evenly(ndt_lib):
    toff = _offsets([1000] * 64)
    soff = _offsets([1000] * 64)
    for shards in (1, 2, 3, 4, 8):
        b = M.multi_plan(shards, toff, soff)
        assert b[0] == 0 and b[-1] == 64 and np.all(np.diff(b.astype(np.int64)) >= 0)
        sizes = np.diff(b.astype(np.int64))
        assert sizes.max() - sizes.min() <= 1


def test_split_balances_work_not_pairs(ndt_lib):
    # the first 8 pairs are 10x heavier than the other 80
    sizes = [10000] * 8 + [1000] * 80
    toff = _offsets(sizes)
    soff = _offsets(sizes)
    b = M.multi_plan(2, toff, soff, iterations_hint=30).astype(np.int64)
    w = np.array(sizes, dtype=np.float64)
    left, right = w[: b[1]].sum(), w[b[1]:].sum()
    assert abs(left - right) <= w.max()
    assert b[1] < 44          # far from the pair-count midpoint


def test_more_shards_than_pairs_and_empty_pairs(ndt_lib):
    toff = _offsets([100, 0, 100])
    soff = _offsets([100, 0, 100])
    b = M.multi_plan(8, toff, soff).astype(np.int64)
    assert b[0] == 0 and b[-1] == 3 and np.all(np.diff(b) >= 0)
    assert sorted(set(np.repeat(np.arange(8), np.diff(b)))) == sorted(set(np.repeat(np.arange(8), np.diff(b))))
    assert np.diff(b).sum() == 3


def test_plan_is_deterministic_and_covers_every_pair(ndt_lib):
    rng = np.random.default_rng(5)
    sizes_t = rng.integers(0, 5000, 257)
    sizes_s = rng.integers(0, 5000, 257)
    toff, soff = _offsets(sizes_t), _offsets(sizes_s)
    a = M.multi_plan(6, toff, soff, 12)
    b = M.multi_plan(6, toff, soff, 12)
    assert np.array_equal(a, b)
    work = 3.0 * sizes_t + 12.0 * sizes_s + 1.0
    per = [work[int(a[d]): int(a[d + 1])].sum() for d in range(6)]
    assert max(per) - min(per) <= 2 * work.max()


def test_plan_rejects_bad_arguments(ndt_lib):
    toff = _offsets([10, 10])
    with pytest.raises(NdtError):
        M.multi_plan(0, toff, toff)
    bad = np.array([0, 10, 5], dtype=np.uint64)
    with pytest.raises(NdtError):
        M.multi_plan(2, bad, toff)


def test_plan_properties_hold_for_random_inputs(ndt_lib):
    """Property check (hypothesis): for any sizes and shard count the plan is a monotone cover of
    the pairs and no shard exceeds the even share by more than one pair's work."""
    from hypothesis import given, settings, strategies as st

    @settings(max_examples=150, deadline=None)
    @given(st.lists(st.tuples(st.integers(0, 20000), st.integers(0, 20000)), min_size=1, max_size=200),
           st.integers(1, 16), st.integers(0, 60))
    def check(sizes, shards, hint):
        toff = _offsets([a for a, _ in sizes])
        soff = _offsets([b for _, b in sizes])
        b = M.multi_plan(shards, toff, soff, hint).astype(np.int64)
        assert b[0] == 0 and b[-1] == len(sizes) and np.all(np.diff(b) >= 0) and len(b) == shards + 1
        k = hint if hint > 0 else 30
        work = np.array([3.0 * a + k * s + 1.0 for a, s in sizes])
        share = work.sum() / shards
        for d in range(shards):
            assert work[b[d]: b[d + 1]].sum() <= share + work.max() + 1e-6

    check()


def test_per_pair_iteration_hints_move_the_split(ndt_lib):
    """ndt2d_multi_plan_hinted: equal clouds, but the first quarter of the candidates needs 50 iterations and the rest 15 -
    the shards balance the hinted work, not the pair count; without hints the split is even."""
    n = 128
    toff = _offsets([2000] * n)
    soff = _offsets([2000] * n)
    hints = np.array([50] * 32 + [15] * 96, dtype=np.int32)
    even = M.multi_plan(4, toff, soff, 30).astype(np.int64)
    assert np.diff(even).tolist() == [32, 32, 32, 32]
    b = M.multi_plan(4, toff, soff, 30, pair_iterations=hints).astype(np.int64)
    assert b[0] == 0 and b[-1] == n and np.all(np.diff(b) > 0)
    work = 3.0 * 2000 + hints * 2000.0 + 1.0
    per = [work[b[d]: b[d + 1]].sum() for d in range(4)]
    assert max(per) - min(per) <= 2 * work.max()
    assert b[1] < 24                                   # the heavy candidates are spread over more than one shard
    zero = M.multi_plan(4, toff, soff, 30, pair_iterations=np.zeros(n, np.int32))     # hint 0 = "no hint for this pair"
    assert np.array_equal(zero.astype(np.int64), even)
    with pytest.raises(ValueError):
        M.multi_plan(4, toff, soff, 30, pair_iterations=hints[:5])
