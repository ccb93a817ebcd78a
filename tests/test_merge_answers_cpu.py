"""CPU: bbq_merge_answers (host-only C ABI) - the global answer from shard-local answers.

Each shard's block is built here, in numpy, the way the last finalize launch of a shard leaves it (include/bbq.h,
bbq_shard_scan_begin): the cut = the (k+1)-th largest score key over the rows the shard has SEEN (its own + a pilot replica of the
global prefix), its own rows above the cut in descending order.  Whatever the merge answers (status 0) must be bit-identical to the
reference heap over all rows (the oracle); and it must answer whenever the k+1 largest scores are pairwise different."""
import numpy as np
import pytest

import orclib as O
from bbqlib import capi as B


def key_of(s32):
    b = s32.view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


def shard_block(s32, keys, r0, r1, pilot, k, stride):
    blk = np.zeros(stride, np.uint64)
    seen = np.unique(np.concatenate([np.arange(0, pilot if r0 > 0 else 0), np.arange(r0, r1)])).astype(np.int64)
    ks = np.sort(keys[seen])[::-1]
    cut = int(ks[k]) if len(ks) >= k + 1 else 0
    own = np.arange(r0, r1)
    own = own[keys[own] > cut]
    own = own[np.argsort(-keys[own], kind="stable")]
    assert len(own) <= k
    blk[1], blk[2] = len(own), cut
    blk[3:3 + len(own)] = (own.astype(np.uint64) << np.uint64(32)) | s32[own].view(np.uint32).astype(np.uint64)
    return blk


def run_case(rng, n, k, shards, pilot, levels):
    """levels: number of distinct score values (small = many ties); 0 = continuous scores"""
    nq = 6
    scores = rng.standard_normal((nq, n)).astype(np.float32)
    if levels:
        scores = (np.round(scores * levels) / levels).astype(np.float32)
        scores[0, : n // 2] = 0.0
        scores[0, n // 2:] = -0.0          # +0 and -0 compare equal as floats but have different keys
    k2 = min(k, n)
    stride = k2 + 3
    per = -(-n // shards)
    bounds = [(min(s * per, n), min((s + 1) * per, n)) for s in range(shards)]
    blocks = []
    for (r0, r1) in bounds:
        blocks.append(np.stack([shard_block(scores[q], key_of(scores[q]), r0, r1, min(pilot, r0), k2, stride) for q in range(nq)]))
    order = rng.permutation(shards)            # sources in any order
    idx, sc, cnt, status = B.merge_answers([blocks[s] for s in order], nq, n, k, 1)
    answered = 0
    for q in range(nq):
        oi, osc = O.heap_topk(scores[q], k)
        top = np.sort(scores[q])[::-1][: k2 + 1].astype(np.float64)
        distinct = len(np.unique(top)) == len(top)   # np.unique treats +0 / -0 as equal, like the float comparison
        if status[q] == 0:
            answered += 1
            assert cnt[q] == len(oi)
            assert (idx[q, : cnt[q]] == oi).all(), (q, idx[q, :cnt[q]], oi)
            assert (sc[q, : cnt[q]].view(np.uint32) == osc.view(np.uint32)).all()
        else:
            assert status[q] == 1
            assert not distinct, "the merge must answer whenever the k+1 largest scores are pairwise different"
        if distinct:
            assert status[q] == 0
    return answered


@pytest.mark.parametrize("shards,pilot", [(1, 0), (2, 64), (3, 0), (8, 128), (5, 1000)])
def test_merge_answers_continuous_scores(shards, pilot):
    rng = np.random.default_rng(shards * 100 + pilot)
    for n, k in [(3000, 100), (700, 10), (90, 100), (64, 1), (1500, 1024)]:
        assert run_case(rng, n, k, shards, pilot, 0) == 6


@pytest.mark.parametrize("shards,pilot", [(2, 0), (4, 64), (7, 256)])
def test_merge_answers_with_equal_scores(shards, pilot):
    """heavily tied scores (a few distinct values): whatever is answered is the heap's answer, the rest is handed to the replay"""
    rng = np.random.default_rng(7 + shards)
    for levels in (2, 8, 40):
        for n, k in [(2000, 50), (300, 100), (50, 100)]:
            run_case(rng, n, k, shards, pilot, levels)


def test_merge_answers_flags_and_unproven():
    rng = np.random.default_rng(1)
    n, k, nq = 500, 20, 3
    scores = rng.standard_normal((nq, n)).astype(np.float32)
    stride = k + 3
    blocks = [np.stack([shard_block(scores[q], key_of(scores[q]), r0, r1, 0, k, stride) for q in range(nq)]) for (r0, r1) in ((0, 250), (250, 500))]
    blocks[1][1, 0] |= np.uint64(2) << np.uint64(32)     # NaN flag on query 1 of shard 1
    blocks[0][2, 1] |= np.uint64(1) << np.uint64(32)     # shard 0 could not select for query 2
    idx, sc, cnt, status = B.merge_answers(blocks, nq, n, k, 1)
    assert status.tolist() == [0, 2, 1]
    oi, osc = O.heap_topk(scores[0], k)
    assert (idx[0] == oi).all() and (sc[0].view(np.uint32) == osc.view(np.uint32)).all()
    # a block that claims more entries than its stride holds is refused
    blocks[0][0, 1] = np.uint64(stride)
    with pytest.raises(B.BBQError):
        B.merge_answers(blocks, nq, n, k, 1)


def test_merge_answers_threads_agree():
    rng = np.random.default_rng(5)
    n, k, nq, shards = 4000, 100, 300, 4
    scores = rng.standard_normal((nq, n)).astype(np.float32)
    per = n // shards
    blocks = [np.stack([shard_block(scores[q], key_of(scores[q]), s * per, (s + 1) * per, 128 if s else 0, k, k + 3) for q in range(nq)]) for s in range(shards)]
    a = B.merge_answers(blocks, nq, n, k, 1)
    b = B.merge_answers(blocks, nq, n, k, 4)
    for x, y in zip(a, b):
        assert (x.view(np.uint8) == y.view(np.uint8)).all()
    assert (a[3] == 0).all()


def test_key_of_score_is_monotone():
    v = np.array([-np.inf, -3.5, -1e-30, -0.0, 0.0, 1e-30, 0.25, 7.0, np.inf], np.float32)
    keys = [B.key_of_score(x) for x in v]
    assert keys == sorted(keys) and len(set(keys)) == len(keys)
    assert keys == key_of(v).tolist()
