"""Row-sharded search over torch.distributed (one process per GPU; backend "nccl" = RCCL over xGMI, or "gloo" for
rehearsals).  Host plumbing only: the sweep runs in libbbq (bbq_shard_scan_begin / _wait), the merge in libbbq
(bbq_merge_answers; bbq_replay_batch for the queries that need the heap's history).

Per batch of Q queries every rank sweeps its shard and leaves, in device memory, (a) its SHARD-LOCAL ANSWER per query - its own rows
above its cut, at most k entries, k + 3 words (include/bbq.h) - and (b) its candidate lists, packed and ordered by query.  The MERGE is
sharded too: rank r owns the queries of block r (Qb = ceil(Q / world) consecutive queries).

  every batch:
    1. all_to_all_single, EQUAL splits: the answer blocks of owner block d go to rank d - [Qb, k + 3] words per pair of ranks, fixed
       size, so there is no header round and no host sync before the payload
    2. each owner merges its Qb queries on the host (bbq_merge_answers: a world-way merge of sorted k-entry lists plus the proof that
       the reference heap returns exactly that order)
    3. all_gather of the owners' per-query status (W x Qb bytes): which queries could not be proven (equal scores in or at the edge of
       the answer: about 1 % of the queries at 1 M rows, so nearly every batch of a few hundred has one) or were flagged by a shard
    4. gather to rank 0: [Qb, 2k+1] int32 per rank (indices | f32 score bits | count)
  only for the queries step 3 names (not for their batch):
    their packed lists go from every shard to rank 0 (an all_gather of the counts, one uneven all_to_all), rank 0 replays the reference
    heap over them in shard order (bbq_replay); queries a shard flagged are scored densely by every rank and replayed on rank 0.
  k > 1024 (no shard-local answers): the whole batch takes the list exchange of ABI 2 (headers + uneven payload all_to_all, heap
  replay per owner).

A scanner thread enqueues batch i+1's sweep (bbq_shard_scan_begin returns at once) while batch i is still running on the device, so the
device never drains between batches; the main thread waits for batch i, exchanges and merges it.
"""
import queue
import threading
import time

import numpy as np

from . import capi


class ShardedSearcher:
    def __init__(self, index, n_total, k, max_queries, query_bits=4, sim=capi.COSINE, replay_threads=8, device="cuda",
                 scan_fn=None, dense_fn=None, n_buffers=3, n_local_rows=0, list_cap_per_query=None, collective_device=None,
                 use_answers=True):
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
        self.Qb = (self.Q + self.world - 1) // self.world          # queries per owner block
        self.hdr_len = 2 + self.Qb + self.Qb + 1
        self.k2 = min(self.k, self.n_total)
        self.answers = bool(use_answers) and 1 <= self.k2 <= 1024   # the finalize launch selects answers up to k = 1024
        self.stride = self.k2 + 3
        per_query = int(index.shard_list_cap(self.k2)) if index is not None else int(list_cap_per_query or 0)
        cap = torch.tensor([per_query * self.Q], dtype=torch.int64, device=self.cdev)
        dist.all_reduce(cap, op=dist.ReduceOp.MAX)   # same capacity everywhere
        self.cap = int(cap.item())
        self.bufs = []
        for _ in range(n_buffers):
            self.bufs.append({
                "packed": torch.zeros(self.cap, dtype=torch.int64, device=device),
                "offsets": torch.zeros(self.Q + 1, dtype=torch.int64, device=device),
                "flags": torch.zeros(self.Q, dtype=torch.int32, device=device),
                # padded to world x Qb queries: the equal-split exchange sends whole blocks (rows past the batch stay zero)
                "answers": torch.zeros(self.world * self.Qb * self.stride, dtype=torch.int64, device=device) if self.answers else None,
            })
        self._pin = str(self.cdev).startswith("cuda")
        # the exchange's small device operations (header arithmetic, copies, the collectives' kernels) share the GPU with sweeps that
        # keep every CU busy: on a stream of normal priority each of them waited for up to a whole sweep (3 ms at 10 M rows per rank).
        # They run on a high-priority stream instead (TORCH_NCCL_HIGH_PRIORITY does the same for RCCL's own stream, set by bench.py)
        self._stream = torch.cuda.Stream(priority=-1) if self._pin else None
        self.hdr_in = torch.zeros(self.world * self.hdr_len, dtype=torch.int64, device=self.cdev)
        self.ans_in = torch.zeros(self.world * self.Qb * self.stride, dtype=torch.int64, device=self.cdev) if self.answers else None
        self._h_ans = torch.zeros(self.world * self.Qb * self.stride, dtype=torch.int64, pin_memory=self._pin) if self.answers else None
        self.res_out = torch.zeros(self.Qb * (2 * self.k + 1), dtype=torch.int32, device=self.cdev)
        self.res_in = [torch.zeros(self.Qb * (2 * self.k + 1), dtype=torch.int32, device=self.cdev) for _ in range(self.world)] if self.rank == 0 else None
        self._recv = None        # grow-only landing buffer of the packed entries of my block (collective device)
        self._h_recv = None      # its pinned host twin
        self._in_flight = threading.Semaphore(2)   # libbbq keeps at most two batches of one index in flight
        self.last_exchange = {}  # sizes of the last batch's exchange (bench / diagnostics)
        self.list_batches = 0    # batches that needed the list path
        self.phase_s = {"scan": 0.0, "exchange": 0.0, "to_host": 0.0, "merge": 0.0, "answers": 0.0, "lists": 0.0, "batches": 0}  # this rank's wall time per phase

    # ------------------------------------------------------------------ one rank's sweep of one batch
    def _begin(self, buf, qq, qc):
        """enqueue the sweep of one batch; returns what _finish needs"""
        t0 = time.perf_counter()
        nq = qq.shape[0]
        if self._scan_fn is not None:  # tests without a GPU inject (packed, offsets, flags[, answers]) numpy arrays
            r = self._scan_fn(qq, qc)
            packed, offsets, flags = r[0], r[1], r[2]
            t = self.torch
            buf["packed"][:len(packed)] = t.from_numpy(packed.view(np.int64))
            buf["offsets"][:nq + 1] = t.from_numpy(offsets.astype(np.int64))
            buf["flags"][:nq] = t.from_numpy(flags.astype(np.int32))
            if self.answers:
                buf["answers"].zero_()
                buf["answers"][:nq * self.stride] = t.from_numpy(np.ascontiguousarray(r[3][:nq], np.uint64).view(np.int64).reshape(-1))
            return {"total": int(offsets[nq]), "t0": t0}
        self._in_flight.acquire()
        try:
            self.index.shard_scan_begin(qq, qc, self.qb, self.sim, self.k2, buf["packed"].data_ptr(), self.cap, buf["offsets"].data_ptr(),
                                        buf["flags"].data_ptr(), buf["answers"].data_ptr() if self.answers else None, self.stride)
        except BaseException:
            self._in_flight.release()
            raise
        return {"total": None, "t0": t0}

    def _finish(self, ticket):
        if ticket["total"] is None:
            try:
                ticket["total"] = self.index.shard_scan_wait()
            finally:
                self._in_flight.release()
        self.phase_s["scan"] += time.perf_counter() - ticket["t0"]   # begin -> device done (overlaps the previous batch's merge)
        return ticket["total"]

    # ------------------------------------------------------------------ exchange + merge of one batch
    def _grow(self, n):
        t = self.torch
        if self._recv is None or self._recv.numel() < n:
            n = max(n, 1024) * 5 // 4
            self._recv = t.empty(n, dtype=t.int64, device=self.cdev)
            self._h_recv = t.empty(n, dtype=t.int64, pin_memory=self._pin)
        return self._recv

    def _merge(self, buf, nq, total, qq, qc):
        if self._stream is None:
            return self._merge_on_stream(buf, nq, total, qq, qc)
        with self.torch.cuda.stream(self._stream):
            return self._merge_on_stream(buf, nq, total, qq, qc)

    def _gather_answers(self, res, nq):
        """[Qb, 2k+1] int32 of every owner to rank 0 -> (idx, score, count) of the batch there"""
        t, dist, W, Qb, k = self.torch, self.dist, self.world, self.Qb, self.k
        self.res_out.copy_(t.from_numpy(res.reshape(-1)))
        dist.gather(self.res_out, self.res_in, dst=0)
        if self.rank != 0:
            return None
        allr = t.stack(self.res_in).cpu().numpy().reshape(W * Qb, 2 * k + 1)[:nq]
        return (np.ascontiguousarray(allr[:, :k]), np.ascontiguousarray(allr[:, k:2 * k]).view(np.float32), allr[:, 2 * k].astype(np.int64))

    def _merge_on_stream(self, buf, nq, total, qq, qc):
        t, dist, W, Qb, k = self.torch, self.dist, self.world, self.Qb, self.k
        ph, clock = self.phase_s, time.perf_counter
        nqb = max(0, min(Qb, nq - self.rank * Qb))
        if not self.answers:
            t0 = clock()
            out = self._merge_lists(buf, nq, total, qq, qc)
            ph["lists"] += clock() - t0
            ph["batches"] += 1
            self.list_batches += 1
            return out
        t0 = clock()
        # ---- 1. the answer blocks of owner block d to rank d (equal splits: no header round, no host sync)
        send = buf["answers"].to(self.cdev)
        dist.all_to_all_single(self.ans_in, send)
        t1 = clock()
        ph["exchange"] += t1 - t0
        self._h_ans.copy_(self.ans_in, non_blocking=True)
        if self.ans_in.is_cuda:
            t.cuda.current_stream().synchronize()
        t2 = clock()
        ph["to_host"] += t2 - t1
        # ---- 2. merge of my block
        res = np.zeros((Qb, 2 * k + 1), np.int32)
        status = np.zeros(0, np.uint8)
        if nqb > 0:
            blocks = self._h_ans.numpy().view(np.uint64).reshape(W, Qb, self.stride)
            idx, sc, cnt, status = capi.merge_answers([blocks[s] for s in range(W)], nqb, self.n_total, k, self.threads)
            res[:nqb, :k] = idx
            res[:nqb, k:2 * k] = sc.view(np.int32)
            res[:nqb, 2 * k] = cnt
        t3 = clock()
        ph["merge"] += t3 - t2
        # ---- 3. which queries could their owner not prove?  Every owner's status of its block, to everybody (W x Qb bytes)
        st_local = t.zeros(Qb, dtype=t.uint8)
        if nqb > 0:
            st_local[:nqb] = t.from_numpy(status)
        st_all = t.empty(W * Qb, dtype=t.uint8, device=self.cdev)
        dist.all_gather_into_tensor(st_all, st_local.to(self.cdev))
        st = st_all.cpu().numpy()[:nq]                               # the one host sync of the fast path
        need = np.nonzero(st == 1)[0]                                  # equal scores in or at the edge of the answer: replay the lists
        dense = np.nonzero(st == 2)[0]                                 # a shard flagged the query: dense path
        self.last_exchange = {"answer_words_per_pair": int(Qb * self.stride), "block_queries": int(nqb), "replayed_queries": int(len(need)),
                              "dense_queries": int(len(dense))}
        # ---- 4. the answers of every block to rank 0
        out = self._gather_answers(res, nq)
        ph["answers"] += clock() - t3
        if len(need):   # rare: only THOSE queries' lists travel, to rank 0, which replays the reference heap over them
            t4 = clock()
            self._replay_on_root(buf, need, out)
            ph["lists"] += clock() - t4
            self.list_batches += 1
        if len(dense):
            self._merge_dense(dense.tolist(), qq, qc, out)
        ph["batches"] += 1
        return out

    def _replay_on_root(self, buf, need, res):
        """the candidate lists of the queries in `need` (global indices into the batch, the same on every rank) from every shard to
        rank 0, which replays the reference heap over them in shard order and overwrites those rows of `res`"""
        t, dist, W, k = self.torch, self.dist, self.world, self.k
        n_need = len(need)
        idx_t = t.as_tensor(need, dtype=t.int64, device=buf["offsets"].device)
        se = t.stack([buf["offsets"][idx_t], buf["offsets"][idx_t + 1]]).cpu().numpy()      # my slices of the packed buffer
        counts = (se[1] - se[0]).astype(np.int64)
        cnt_all = t.empty(W * n_need, dtype=t.int64, device=self.cdev)
        dist.all_gather_into_tensor(cnt_all, t.from_numpy(counts).to(self.cdev))
        cnt_all = cnt_all.cpu().numpy().reshape(W, n_need)
        parts = [buf["packed"][int(a):int(b)] for a, b in zip(se[0], se[1]) if b > a]
        send = (t.cat(parts) if parts else buf["packed"][:0]).to(self.cdev)
        in_splits = [int(cnt_all[r].sum()) for r in range(W)] if self.rank == 0 else [0] * W
        out_splits = [int(counts.sum())] + [0] * (W - 1)
        n_in = int(sum(in_splits))
        recv = self._grow(n_in)[:n_in]
        dist.all_to_all_single(recv, send, output_split_sizes=in_splits, input_split_sizes=out_splits)
        if self.rank != 0:
            return
        self._h_recv[:n_in].copy_(recv, non_blocking=True)
        if recv.is_cuda:
            t.cuda.current_stream().synchronize()
        hp = self._h_recv[:n_in].numpy().view(np.uint64)
        idx, sc, cnt = res
        base = np.concatenate([[0], np.cumsum(in_splits)])
        within = np.concatenate([np.zeros((W, 1), np.int64), np.cumsum(cnt_all, axis=1)], axis=1)
        for j, q in enumerate(need.tolist()):
            lists = [hp[base[r] + within[r, j]: base[r] + within[r, j + 1]] for r in range(W)]
            i1, s1 = capi.replay(lists, self.n_total, k)
            idx[q, :len(i1)], sc[q, :len(i1)], cnt[q] = i1, s1, len(i1)

    def _merge_lists(self, buf, nq, total, qq, qc):
        """headers -> packed entries of my block from every shard -> heap replay of my block -> gather; dense path for flagged queries"""
        t, dist, W, Qb, k = self.torch, self.dist, self.world, self.Qb, self.k
        # ---- headers: what I hold for every owner block
        off = buf["offsets"][:nq + 1].to(self.cdev)
        flags = buf["flags"][:nq].to(self.cdev)
        pad = W * Qb - nq
        if pad:
            off = t.cat([off, off[-1:].expand(pad)])
            flags = t.cat([flags, t.zeros(pad, dtype=flags.dtype, device=flags.device)])
        offm = off[:-1].view(W, Qb)
        starts = offm[:, 0]
        ends = off[Qb::Qb]
        counts = ends - starts
        anyf = flags.ne(0).any().to(t.int64).expand(W)
        hdr_out = t.cat([counts[:, None], anyf[:, None], flags.view(W, Qb).to(t.int64), offm - starts[:, None], counts[:, None]], 1).contiguous()
        dist.all_to_all_single(self.hdr_in, hdr_out.view(-1))
        h = t.cat([self.hdr_in, counts]).cpu().numpy()          # the one host sync of the exchange
        hin = h[:W * self.hdr_len].reshape(W, self.hdr_len)
        out_counts = h[W * self.hdr_len:].tolist()
        in_counts = hin[:, 0].tolist()
        any_flag = bool(hin[:, 1].any())
        # ---- the packed entries of my block, from every shard
        n_in = int(sum(in_counts))
        recv = self._grow(n_in)[:n_in]
        send = buf["packed"][:total].to(self.cdev)
        dist.all_to_all_single(recv, send, output_split_sizes=in_counts, input_split_sizes=out_counts)
        self._h_recv[:n_in].copy_(recv, non_blocking=True)
        if recv.is_cuda:
            t.cuda.current_stream().synchronize()
        hp = self._h_recv[:n_in].numpy().view(np.uint64)
        # ---- replay of my block (shard order = rank order)
        nqb = max(0, min(Qb, nq - self.rank * Qb))
        res = np.zeros((Qb, 2 * k + 1), np.int32)
        if nqb > 0:
            packed, offsets, cum = [], [], 0
            for s in range(W):
                packed.append(hp[cum:cum + in_counts[s]])
                offsets.append(np.ascontiguousarray(hin[s, 2 + Qb:2 + Qb + nqb + 1]))
                cum += in_counts[s]
            idx, sc, cnt = capi.replay_batch(packed, offsets, nqb, self.n_total, k, self.threads)
            res[:nqb, :k] = idx
            res[:nqb, k:2 * k] = sc.view(np.int32)
            res[:nqb, 2 * k] = cnt
        self.last_exchange.update({"entries_received": n_in, "entries_sent": int(total), "header_int64": int(W * self.hdr_len)})
        out = self._gather_answers(res, nq)
        if any_flag:   # collective: every rank takes part; rank 0 overwrites those queries' rows
            gf = t.zeros(W * nq, dtype=t.int32, device=self.cdev)
            dist.all_gather_into_tensor(gf, buf["flags"][:nq].to(self.cdev).contiguous())
            flagged = np.nonzero(gf.cpu().numpy().reshape(W, nq).any(axis=0))[0].tolist()
            self._merge_dense(flagged, qq, qc, out)
        return out

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

    def phases_ms(self, reset=True):
        """this rank's average wall time per batch and phase ("scan" = enqueue to device-done; it overlaps the previous batch's
        exchange and merge)"""
        n = max(self.phase_s["batches"], 1)
        out = {k_: round(v / n * 1e3, 3) for k_, v in self.phase_s.items() if k_ != "batches"}
        out["batches"] = self.phase_s["batches"]
        out["list_path_batches"] = self.list_batches
        out.update(self.last_exchange)
        if reset:
            for k_ in self.phase_s:
                self.phase_s[k_] = 0 if k_ == "batches" else 0.0
            self.list_batches = 0
        return out

    # ------------------------------------------------------------------ public
    def search(self, qq, qc):
        """one batch, no overlap.  Returns (idx [nq,k], score [nq,k], count [nq]) on rank 0, None elsewhere."""
        buf = self.bufs[0]
        total = self._finish(self._begin(buf, qq, qc))
        return self._merge(buf, qq.shape[0], total, qq, qc)

    def search_stream(self, batches):
        """batches: list of (qq, qc).  The scanner thread enqueues batch i+1's sweep while batch i is running / being merged.
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
                    ready.put((b, self._begin(b, qq, qc), qq, qc))
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
            b, ticket, qq, qc = item
            total = self._finish(ticket)
            out.append(self._merge(b, qq.shape[0], total, qq, qc))
            free.put(b)
        th.join()
        return out
