"""Row-sharded search over torch.distributed (one process per GPU; backend "nccl" = RCCL over xGMI, or "gloo" for
rehearsals).  Host plumbing only: the sweep runs in libbbq (bbq_shard_scan), the merge in libbbq (bbq_replay_batch).

Per batch of queries every rank sweeps its shard and packs its candidates; the ranks exchange
  1. all_gather  [total, any_flag]                       (2 int64 per rank)
  2. all_gather  offsets  [Q+1] int64
  3. all_gather  packed[:max total over ranks >= 1]      (the only sizeable message: ~2K entries x 8 B per query and rank)
and rank 0 replays the reference heap over (own list, rank 1's, rank 2's, ...) in global row order.  Rank 0's own list
(which carries the dense first segment) never travels.  A scanner thread keeps the GPU sweeping batch i+1 while the
main thread gathers and replays batch i.  Queries some shard could not bound (flags) are scored densely by every rank, one
by one; the rest of their batch stays on the sparse path.
"""
import queue
import threading

import numpy as np

from . import capi


class ShardedSearcher:
    def __init__(self, index, n_total, k, max_queries, query_bits=4, sim=capi.COSINE, replay_threads=8, device="cuda",
                 scan_fn=None, dense_fn=None, n_buffers=3, n_local_rows=0, list_cap_per_query=None, collective_device=None):
        import torch
        import torch.distributed as dist
        self.torch, self.dist = torch, dist
        self.rank, self.world = dist.get_rank(), dist.get_world_size()
        self.index, self.n_total, self.k, self.Q = index, int(n_total), int(k), int(max_queries)
        self.qb, self.sim, self.threads, self.device = query_bits, sim, replay_threads, device
        # nccl (RCCL) moves device tensors; gloo (rehearsals on one GPU / CPU tests) needs them staged on the host
        self.cdev = collective_device or device
        self._scan_fn, self._dense_fn = scan_fn, dense_fn
        self._dense_rows = int(n_local_rows)
        per_query = int(index.shard_list_cap(k)) if index is not None else int(list_cap_per_query or 0)
        cap = torch.tensor([per_query * self.Q], dtype=torch.int64, device=self.cdev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)   # same capacity everywhere: slices of any rank fit any buffer
        self.cap = int(cap.item())
        self.bufs = []
        for _ in range(n_buffers):
            self.bufs.append({
                "packed": torch.zeros(self.cap, dtype=torch.int64, device=device),
                "offsets": torch.zeros(self.Q + 1, dtype=torch.int64, device=device),
                "flags": torch.zeros(self.Q, dtype=torch.int32, device=device),
            })
        self.g_packed = torch.zeros(self.world * self.cap, dtype=torch.int64, device=self.cdev)
        self.g_offsets = torch.zeros(self.world * (self.Q + 1), dtype=torch.int64, device=self.cdev)
        self.g_meta = torch.zeros(self.world * 2, dtype=torch.int64, device=self.cdev)
        pin = str(self.cdev).startswith("cuda")
        # pinned landing buffers on rank 0: a pageable .cpu() of a few MB per batch would cost more than the sweep
        self.h_packed = torch.empty(self.world * self.cap if self.rank == 0 else 1, dtype=torch.int64, pin_memory=pin)
        self.h_own = torch.empty(self.cap if self.rank == 0 else 1, dtype=torch.int64, pin_memory=pin)

    # ------------------------------------------------------------------ one rank's sweep of one batch
    def _scan(self, buf, qq, qc):
        nq = qq.shape[0]
        if self._scan_fn is not None:  # tests without a GPU inject (packed, offsets, flags) numpy arrays
            packed, offsets, flags = self._scan_fn(qq, qc)
            t = self.torch
            buf["packed"][:len(packed)] = t.from_numpy(packed.view(np.int64))
            buf["offsets"][:nq + 1] = t.from_numpy(offsets.astype(np.int64))
            buf["flags"][:nq] = t.from_numpy(flags.astype(np.int32))
            return int(offsets[nq])
        return self.index.shard_scan(qq, qc, self.qb, self.sim, self.k, buf["packed"].data_ptr(), self.cap,
                                     buf["offsets"].data_ptr(), buf["flags"].data_ptr())

    # ------------------------------------------------------------------ exchange + replay of one batch
    def _to_host(self, dst, src):
        n = src.numel()
        dst[:n].copy_(src, non_blocking=True)
        if src.is_cuda:
            self.torch.cuda.current_stream().synchronize()
        return dst[:n].numpy()

    def _merge(self, buf, nq, total, qq, qc):
        t, dist = self.torch, self.dist
        any_flag = int(buf["flags"][:nq].ne(0).any().item())
        meta = t.tensor([total, any_flag], dtype=t.int64, device=self.cdev)
        dist.all_gather_into_tensor(self.g_meta, meta)
        m = self.g_meta.cpu().numpy().reshape(self.world, 2)
        flagged = []
        if m[:, 1].any():
            # some shard could not bound some query: find out WHICH queries (the rest of the batch stays on the sparse path)
            gf = t.zeros(self.world * nq, dtype=t.int32, device=self.cdev)
            dist.all_gather_into_tensor(gf, buf["flags"][:nq].to(self.cdev).contiguous())
            flagged = np.nonzero(gf.cpu().numpy().reshape(self.world, nq).any(axis=0))[0].tolist()
        go = self.g_offsets[:self.world * (nq + 1)]
        dist.all_gather_into_tensor(go, buf["offsets"][:nq + 1].to(self.cdev).contiguous())
        maxtot = int(m[1:, 0].max()) if self.world > 1 else 0
        gp = None
        if maxtot > 0:
            gp = self.g_packed[:self.world * maxtot]
            dist.all_gather_into_tensor(gp, buf["packed"][:maxtot].to(self.cdev).contiguous())
        res = None
        if self.rank == 0:
            h_off = go.cpu().numpy().reshape(self.world, nq + 1)
            packed = [self._to_host(self.h_own, buf["packed"][:int(m[0, 0])]).view(np.uint64)]
            offsets = [h_off[0]]
            if self.world > 1 and maxtot > 0:
                h_p = self._to_host(self.h_packed, gp).view(np.uint64).reshape(self.world, maxtot)
                for r in range(1, self.world):
                    packed.append(h_p[r, :int(m[r, 0])])
                    offsets.append(h_off[r])
            elif self.world > 1:
                for r in range(1, self.world):
                    packed.append(np.zeros(0, np.uint64))
                    offsets.append(h_off[r])
            res = capi.replay_batch(packed, offsets, nq, self.n_total, self.k, self.threads)
        if flagged:   # collective: every rank takes part; rank 0 overwrites those queries' rows
            self._merge_dense(flagged, qq, qc, res)
        return res

    def _merge_dense(self, which, qq, qc, res):
        """a shard could not bound these queries (NaN scores / a flood beyond every buffer): every rank scores all its rows
        for them, rank 0 replays every row into `res` - slow, exact, rare"""
        t, dist = self.torch, self.dist
        rows = t.tensor([self.index.n if self.index is not None else self._dense_rows], dtype=t.int64, device=self.cdev)
        allrows = t.zeros(self.world, dtype=t.int64, device=self.cdev)
        dist.all_gather_into_tensor(allrows, rows)
        allrows = allrows.cpu().numpy()
        mx = int(allrows.max())
        for q in which:
            if self._dense_fn is not None:
                s32 = self._dense_fn(qq[q], qc[q])
            else:
                _, _, s32 = self.index.score_rows(qq[q], qc[q], self.qb, self.sim)
            mine = t.zeros(mx, dtype=t.float32, device=self.cdev)
            mine[:len(s32)] = t.from_numpy(np.ascontiguousarray(s32))
            everyone = t.zeros(self.world * mx, dtype=t.float32, device=self.cdev)
            dist.all_gather_into_tensor(everyone, mine)
            if self.rank == 0:
                idx, sc, cnt = res
                e = everyone.cpu().numpy().reshape(self.world, mx)
                s_all = np.concatenate([e[r, :int(allrows[r])] for r in range(self.world)])
                ent = (np.arange(len(s_all), dtype=np.uint64) << np.uint64(32)) | s_all.view(np.uint32).astype(np.uint64)
                i1, s1 = capi.replay([ent], self.n_total, self.k)
                idx[q, :len(i1)], sc[q, :len(i1)], cnt[q] = i1, s1, len(i1)

    # ------------------------------------------------------------------ public
    def search(self, qq, qc):
        """one batch, no overlap.  Returns (idx [nq,k], score [nq,k], count [nq]) on rank 0, None elsewhere."""
        buf = self.bufs[0]
        total = self._scan(buf, qq, qc)
        return self._merge(buf, qq.shape[0], total, qq, qc)

    def search_stream(self, batches):
        """batches: list of (qq, qc).  The scanner thread sweeps batch i+1 while this thread merges batch i.
        Returns the list of per-batch results (rank 0) / Nones."""
        free = queue.Queue()
        for b in self.bufs:
            free.put(b)
        ready = queue.Queue()
        err = []

        def scanner():
            try:
                for qq, qc in batches:
                    b = free.get()
                    ready.put((b, self._scan(b, qq, qc), qq, qc))
            except BaseException as e:  # surface in the consumer
                err.append(e)
                ready.put(None)

        th = threading.Thread(target=scanner, daemon=True)
        th.start()
        out = []
        for _ in batches:
            item = ready.get()
            if item is None:
                raise err[0]
            b, total, qq, qc = item
            out.append(self._merge(b, qq.shape[0], total, qq, qc))
            free.put(b)
        th.join()
        return out
