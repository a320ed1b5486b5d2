"""Utterance-level sharding across the GPUs of one node (one process per GPU, torch.distributed).

Every animation frame -- and so every utterance -- is independent (speech_anime/datasets/sliding_window.py:348-371
cuts self-contained windows; speech_anime/model/model.py:450-461 batches are independent; eval-mode BatchNorm is
stateless), so the path shards by contiguous blocks of clips with replicated weights and NO collective inside
the compute.  The only exchange is the one the task names: an all-gather that reassembles the per-frame output
sequence on every rank (backend "nccl" = RCCL over xGMI on ROCm; "gloo" in the CPU tests).
"""
import torch
import torch.distributed as dist


def shard_range(n_items, rank, world):
    """Contiguous block [lo, hi) of `n_items` for `rank`; sizes differ by at most one."""
    base, rem = divmod(int(n_items), int(world))
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def _comm_device(group=None):
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend(group) == "nccl" else torch.device("cpu")


def frame_counts_all(local_count, group=None):
    """Every rank's frame count (ragged shards are allowed)."""
    world = dist.get_world_size(group)
    dev = _comm_device(group)
    mine = torch.tensor([int(local_count)], dtype=torch.int64, device=dev)
    allc = [torch.empty(1, dtype=torch.int64, device=dev) for _ in range(world)]
    dist.all_gather(allc, mine, group=group)
    return [int(v.item()) for v in allc]


def run_chunks(n_local, chunk, gatherer, compute):
    """The per-step chunk loop of a rank: `compute(f0, f1)` produces this rank's output rows for its frames [f0, f1) and
    the rows are handed to the gatherer chunk by chunk.  With a gatherer the loop runs over ITS chunk count -- the longest
    shard's -- so that every rank issues the same number of collectives; a rank whose shard has ended joins the remaining
    gathers with no rows of its own (ragged shards: different ranks hold different numbers of chunks)."""
    n_local, chunk = int(n_local), int(chunk)
    n = gatherer.n_chunks if gatherer is not None else (n_local + chunk - 1) // chunk
    for ci in range(n):
        f0, f1 = min(n_local, ci * chunk), min(n_local, (ci + 1) * chunk)
        rows = compute(f0, f1) if f1 > f0 else None
        if gatherer is not None:
            gatherer.gather_chunk(rows, ci)
    if gatherer is not None:
        gatherer.finish()


class FrameGatherer:
    """All-gather of per-frame rows, issued chunk by chunk so that the transfer of chunk i overlaps the
    compute of chunk i+1 (the collective is asynchronous on the process group's own stream).

    Even shards (every rank holds the same number of frames, the bench case) use ONE contiguous
    all_gather_into_tensor per chunk straight into the final buffer, which is laid out chunk-major:
    buffer[chunk][rank][row] -- no staging copy, no extra memory.  `rows(rank)` returns that rank's frames in
    order as a list of per-chunk views; `gathered()` materialises the rank-major (= original clip order)
    matrix when a caller wants one tensor.  Ragged shards fall back to padded staging buffers.
    """

    def __init__(self, counts, row_width, dtype, device, chunk_len, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.counts = [int(c) for c in counts]
        assert len(self.counts) == self.world
        self.width, self.chunk = int(row_width), int(chunk_len)
        self.even = len(set(self.counts)) == 1
        self.n_chunks = (max(self.counts) + self.chunk - 1) // self.chunk
        self.dtype, self.device = dtype, device
        self._pending = []
        if self.even:
            n = self.counts[0]
            self._lens = [min(self.chunk, n - c * self.chunk) for c in range(self.n_chunks)]
            self._base = [0]
            for l in self._lens:
                self._base.append(self._base[-1] + l * self.world)
            self.buf = torch.empty((self._base[-1], self.width), dtype=dtype, device=device)
        else:
            self.offsets = [0]
            for c in self.counts:
                self.offsets.append(self.offsets[-1] + c)
            self.buf = torch.empty((self.offsets[-1], self.width), dtype=dtype, device=device)

    def gather_chunk(self, rows, chunk_index):
        """`rows`: this rank's output rows for its frames [chunk_index*chunk, (chunk_index+1)*chunk) (fewer, or
        None, at the end of a short shard).  Every rank calls this for chunk_index = 0 .. n_chunks-1."""
        f0 = chunk_index * self.chunk
        if self.even:
            dst = self.buf[self._base[chunk_index]: self._base[chunk_index + 1]]
            self._pending.append((dist.all_gather_into_tensor(dst, rows.contiguous(), group=self.group, async_op=True), None, f0, None))
            return
        lens = [max(0, min(c - f0, self.chunk)) for c in self.counts]
        pad = max(lens)
        if pad == 0:
            return
        send = torch.zeros((pad, self.width), dtype=self.dtype, device=self.device)
        if rows is not None and rows.shape[0]:
            send[:rows.shape[0]] = rows
        stage = torch.empty((self.world * pad, self.width), dtype=self.dtype, device=self.device)
        work = dist.all_gather_into_tensor(stage, send, group=self.group, async_op=True)
        self._pending.append((work, stage.view(self.world, pad, self.width), f0, lens))

    def finish(self):
        for work, stage, f0, lens in self._pending:
            work.wait()
            if stage is not None:
                for r in range(self.world):
                    if lens[r]:
                        self.buf[self.offsets[r] + f0: self.offsets[r] + f0 + lens[r]] = stage[r, :lens[r]]
        self._pending = []

    def rows(self, rank):
        """Rank `rank`'s frames, in order, as a list of views (one per chunk)."""
        if not self.even:
            return [self.buf[self.offsets[rank]: self.offsets[rank + 1]]]
        return [self.buf[self._base[c] + rank * l: self._base[c] + (rank + 1) * l] for c, l in enumerate(self._lens)]

    def gathered(self):
        """(sum(counts), width) matrix in rank order = original clip order (copies in the even case)."""
        if not self.even:
            return self.buf
        return torch.cat([v for r in range(self.world) for v in self.rows(r)], 0)


class ExpandGatherer:
    """Reassembles the per-frame output on every rank WITHOUT moving it: the regressor's output is a fixed linear map of
    the PCA coefficients (output_module.py:94-116), 1 KB per frame against 359 KB of dgrad, so the ranks all-gather the
    coefficients (RCCL, chunk by chunk as FrameGatherer does) and every rank expands its peers' frames locally with the
    same kernel that wrote its own (Engine.expand_coef -> sdfa_expand_coef: bit-identical rows).  Per step and rank this
    trades (world - 1) x 7.3 GB over xGMI for (world - 1) x one extra PCA expansion of a shard (4.8 ms for 20,352 frames).

    `buf` is [sum(counts)][out_dim] in rank order; a rank's own rows are written into its slot directly (`own(f0, f1)` is
    the `out=` of Engine.regress)."""

    def __init__(self, counts, engine, device, chunk_len, group=None):
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.counts = [int(c) for c in counts]
        self.engine = engine
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.coefs = FrameGatherer(self.counts, engine.coef_dim, torch.float32, device, chunk_len, group)
        self.n_chunks, self.chunk = self.coefs.n_chunks, self.coefs.chunk
        self.buf = torch.empty((self.offsets[-1], engine.out_dim), dtype=torch.float32, device=device)

    def own(self, f0, f1):
        lo = self.offsets[self.rank]
        return self.buf[lo + f0: lo + f1]

    def gather_chunk(self, coef_rows, chunk_index):
        self.coefs.gather_chunk(coef_rows, chunk_index)

    def finish(self):
        self.coefs.finish()
        for r in range(self.world):
            if r == self.rank:
                continue
            f0 = self.offsets[r]
            for view in self.coefs.rows(r):            # per-chunk views (even shards) or one view (ragged)
                for g0 in range(0, view.shape[0], self.chunk):
                    c = view[g0: g0 + self.chunk]
                    self.engine.expand_coef(c if c.is_contiguous() else c.contiguous(), out=self.buf[f0: f0 + c.shape[0]])
                    f0 += c.shape[0]

    def rows(self, rank):
        return [self.buf[self.offsets[rank]: self.offsets[rank + 1]]]

    def gathered(self):
        return self.buf


class DirectGatherer:
    """One-shot direct all-gather over the fully connected xGMI mesh, independent of RCCL's
    algorithm choice: every rank owns a gathered buffer [sum(counts)][width] in rank order; each rank maps every peer's
    buffer into its own address space (CUDA/HIP IPC handles, exchanged ONCE through the process group) and its regressor
    epilogue stores each output row to its own buffer AND to its slot in the 7 peers' buffers
    (Engine.regress_multi -> sdfa_regress_forward_multi).  One shard crosses each link once; there is no staging copy and no
    collective in the data path -- `finish()` is a device synchronise plus a barrier, after which every rank holds all rows.

    Cross-step hazard and how it is closed: `finish()` only says that THIS step's rows have arrived; a peer's kernels that
    still READ the gathered rows of step k (enqueued after finish(k)) must not be overtaken by step k+1's stores into the same
    memory.  So there are TWO gathered buffers, used alternately (`begin_step()` switches): step k+1 writes the other buffer,
    and the buffer of step k is written again only in step k+2 -- after finish(k+1), whose device synchronise on every rank
    has drained every consumer of step k that was enqueued before it, and whose barrier has told everyone so.  Contract for
    callers: enqueue the consumers of a step's rows on this device before calling the next finish().

    Works with any control backend ("nccl" or "gloo"): the data moves by peer stores, not through the process group.
    Every rank must see every peer's device (no per-rank HIP_VISIBLE_DEVICES / ROCR_VISIBLE_DEVICES mask): checked here.
    Opt-in only (`bench.py --gather direct`); never a default until a run on more than one physical GPU has passed.
    """

    N_BUFFERS = 2

    def __init__(self, counts, row_width, device, group=None):
        from torch.multiprocessing.reductions import reduce_tensor
        self.group = group
        self.world, self.rank = dist.get_world_size(group), dist.get_rank(group)
        self.counts = [int(c) for c in counts]
        assert len(self.counts) == self.world and self.world <= 8, "direct gather: one destination per rank, at most 8"
        self.width = int(row_width)
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.device = torch.device(device)
        # peer mappings are rebuilt by device INDEX: every rank's device must be visible here under the same index
        mine = (self.device.index if self.device.index is not None else torch.cuda.current_device(), torch.cuda.device_count())
        seen = [None] * self.world
        dist.all_gather_object(seen, mine, group=group)
        for r, (idx, _) in enumerate(seen):
            if idx >= torch.cuda.device_count():
                raise RuntimeError(f"direct gather: rank {r} computes on device {idx}, which rank {self.rank} cannot see "
                                   f"({torch.cuda.device_count()} visible): launch without a per-rank *_VISIBLE_DEVICES mask")
        self.bufs = [torch.empty((self.offsets[-1], self.width), dtype=torch.float32, device=self.device) for _ in range(self.N_BUFFERS)]
        handles = [None] * self.world
        dist.all_gather_object(handles, [reduce_tensor(b) for b in self.bufs], group=group)      # (rebuild function, IPC handle + geometry)
        self._peers = []                                                            # keeps the mappings alive
        lo, hi = self.offsets[self.rank], self.offsets[self.rank + 1]
        self._dests = []                                                            # per buffer: my slot in every rank's buffer; own buffer first
        for b in range(self.N_BUFFERS):
            dests = []
            for r in [self.rank] + [r for r in range(self.world) if r != self.rank]:
                if r == self.rank:
                    t = self.bufs[b]
                else:
                    fn, args = handles[r][b]
                    t = fn(*args)
                    self._peers.append(t)
                dests.append(t[lo:hi])
            self._dests.append(dests)
        self.cur = 0

    @property
    def buf(self):
        return self.bufs[self.cur]

    @property
    def dests(self):
        return self._dests[self.cur]

    def begin_step(self):
        """Call before the first store of a step: switches to the other gathered buffer (see the class docstring)."""
        self.cur = (self.cur + 1) % self.N_BUFFERS

    def dest_views(self, f0, f1):
        """Destinations of this rank's frames [f0, f1): one (f1 - f0, width) view per rank, the local one first."""
        return [d[f0:f1] for d in self.dests]

    def finish(self):
        """Every rank's stores are complete and visible: kernels are done (stream synchronise = system-scope release of
        the peer stores), then all ranks have said so."""
        torch.cuda.synchronize(self.device)
        dist.barrier(group=self.group)

    def rows(self, rank):
        return [self.buf[self.offsets[rank]: self.offsets[rank + 1]]]

    def gathered(self):
        return self.buf
