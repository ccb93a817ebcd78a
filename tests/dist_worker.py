"""worker of tests/test_distributed_cpu.py: world_size ranks over gloo on the CPU.  The sweep of each shard is
emulated with the oracle (test infrastructure) exactly as the device does it - dense first segment on the root,
pilot-derived threshold elsewhere - and the PRODUCT's distributed host logic (bbq_amd.distributed + bbq_replay_batch)
gathers and merges.  Rank 0 writes the merged top-k next to the oracle's global answer."""
import json
import os
import sys

import numpy as np
import torch.distributed as dist

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import orclib as O  # noqa: E402
from bbqlib import bbq_amd as B  # noqa: E402
from bbq_amd.distributed import ShardedSearcher  # noqa: E402


def key_of(s32):
    b = s32.view(np.uint32).astype(np.int64)
    return np.where(b & 0x80000000, (~b) & 0xFFFFFFFF, b | 0x80000000)


def main():
    out_path, mode = sys.argv[1], sys.argv[2]
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    n, dim, k, nq, sim, P = 6000, 64, 20, 5, 1, 1024
    rng = np.random.default_rng(3)
    pool = rng.standard_normal((60, dim)).astype(np.float32)            # duplicates -> ties across shards
    base = pool[rng.integers(0, 60, n)] if mode != "plain" else rng.standard_normal((n, dim)).astype(np.float32)
    queries = rng.standard_normal((nq, dim)).astype(np.float32)
    codes, corr, cen = B.quantize_vectors(base, sim)                     # product host quantizer (no GPU needed)
    cdp = B.centroid_dp(cen)
    qs = [B.quantize_query(q, cen, sim, 4) for q in queries]
    qq, qc = np.stack([a for a, _ in qs]), np.stack([b for _, b in qs])
    shard = (n + world - 1) // world
    r0, r1 = rank * shard, min((rank + 1) * shard, n)

    def scores(q):
        return O.score_all(codes, corr, dim, qq[q], qc[q], 4, sim, cdp)[2]

    def answer_block(keys, s32, flagged):
        """what the shard's last finalize launch leaves (include/bbq.h, bbq_shard_scan_begin): the cut = the (k+1)-th largest key over
        every row the shard has SEEN (its own and the pilot replica's), and its own rows above the cut, descending"""
        blk = np.zeros(k + 3, np.uint64)
        if flagged:
            blk[0] = np.uint64(1) << np.uint64(32)
            blk[1] = np.uint64(1) << np.uint64(32)
            return blk
        seen = np.unique(np.concatenate([np.arange(0, P), np.arange(r0, r1)]))
        ks = np.sort(keys[seen])[::-1]
        cut = int(ks[k]) if len(ks) >= k + 1 else 0
        own = np.arange(r0, r1)
        own = own[keys[own] > cut]
        own = own[np.argsort(-keys[own], kind="stable")]
        assert len(own) <= k
        blk[1], blk[2] = len(own), cut
        blk[3:3 + len(own)] = (own.astype(np.uint64) << np.uint64(32)) | s32[own].view(np.uint32).astype(np.uint64)
        return blk

    def scan_fn(qq_b, qc_b):
        packed, offsets, flags, blocks = [], [0], [], []
        for qb_ in range(qq_b.shape[0]):
            q = [i for i in range(nq) if (qq[i] == qq_b[qb_]).all() and (qc[i] == qc_b[qb_]).all()][0]
            s32 = scores(q)
            keys = key_of(s32)
            th = np.sort(keys[:P])[-(k + 1)]                             # threshold from the pilot rows [0, P): rank k + 1, as the device runs it
            rows = np.arange(r0, r1)
            if rank == 0:
                keep = (rows < P) | (keys[r0:r1] > th)
            else:
                keep = keys[r0:r1] > th
            if mode == "flag" and rank == world - 1 and q == 2:
                flags.append(1)
                keep[:] = False
            else:
                flags.append(0)
            blocks.append(answer_block(keys, s32, flags[-1] != 0))
            rr = rows[keep]
            packed.append((rr.astype(np.uint64) << np.uint64(32)) | s32[rr].view(np.uint32).astype(np.uint64))
            offsets.append(offsets[-1] + len(rr))
        return (np.concatenate(packed) if packed else np.zeros(0, np.uint64), np.array(offsets, np.int64), np.array(flags, np.int32),
                np.stack(blocks))

    def dense_fn(qv, qcv):
        q = [i for i in range(nq) if (qq[i] == qv).all()][0]
        return scores(q)[r0:r1]

    S = ShardedSearcher(None, n, k, nq, device="cpu", scan_fn=scan_fn, dense_fn=dense_fn, n_local_rows=r1 - r0,
                        list_cap_per_query=n, replay_threads=2)
    res = S.search(qq, qc)
    res2 = S.search_stream([(qq[:3], qc[:3]), (qq[3:], qc[3:]), (qq, qc)])
    if rank == 0:
        idx, sc, cnt = res
        want = [O.heap_topk(scores(q), k) for q in range(nq)]
        ok = all((idx[q] == want[q][0]).all() and (sc[q].view(np.uint32) == want[q][1].view(np.uint32)).all() for q in range(nq))
        s_idx = np.concatenate([res2[0][0], res2[1][0]])
        ok2 = (s_idx == idx).all() and (res2[2][0] == idx).all()
        # which path the batches took: tie-free data must never need the lists, duplicated vectors / a flagged query must
        json.dump({"ok": bool(ok), "stream_ok": bool(ok2), "world": world, "ties": int(len(np.unique(sc[0])) < k),
                   "list_path_batches": int(S.list_batches)}, open(out_path, "w"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
