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


class FrameGatherer:
    """All-gather of per-frame rows, issued chunk by chunk so that the transfer of chunk i overlaps the
    compute of chunk i+1 (the collective is asynchronous on the process group's own stream).

    Rank r's rows land in out[offset[r] : offset[r] + counts[r]] on every rank, i.e. the gathered buffer is
    the concatenation of the shards in rank order = the original clip order.  Ragged shards are handled by
    padding a chunk to the longest shard's chunk and copying the valid rows into place afterwards.
    """

    def __init__(self, counts, row_width, dtype, device, group=None):
        self.group = group
        self.world = dist.get_world_size(group)
        self.counts = [int(c) for c in counts]
        assert len(self.counts) == self.world
        self.offsets = [0]
        for c in self.counts:
            self.offsets.append(self.offsets[-1] + c)
        self.width = int(row_width)
        self.out = torch.empty((self.offsets[-1], self.width), dtype=dtype, device=device)
        self._pending = []

    def n_chunks(self, chunk_len):
        return (max(self.counts) + chunk_len - 1) // chunk_len

    def gather_chunk(self, rows, f0, chunk_len):
        """`rows`: this rank's output rows for its frames [f0, f0 + chunk_len) (fewer, possibly zero, at the end
        of a short shard).  Every rank calls this for f0 = 0, chunk_len, 2*chunk_len, ... (n_chunks times)."""
        lens = [max(0, min(c - f0, chunk_len)) for c in self.counts]
        pad = max(lens)
        if pad == 0:
            return
        n = 0 if rows is None else int(rows.shape[0])
        if all(l == pad for l in lens):
            outs = [self.out[self.offsets[r] + f0: self.offsets[r] + f0 + pad] for r in range(self.world)]
            work = dist.all_gather(outs, rows.contiguous(), group=self.group, async_op=True)
            self._pending.append((work, None, f0, lens))
            return
        send = torch.zeros((pad, self.width), dtype=self.out.dtype, device=self.out.device)
        if n:
            send[:n] = rows
        stage = [torch.empty((pad, self.width), dtype=self.out.dtype, device=self.out.device) for _ in range(self.world)]
        work = dist.all_gather(stage, send, group=self.group, async_op=True)
        self._pending.append((work, stage, f0, lens))

    def finish(self):
        for work, stage, f0, lens in self._pending:
            work.wait()
            if stage is not None:
                for r in range(self.world):
                    if lens[r]:
                        self.out[self.offsets[r] + f0: self.offsets[r] + f0 + lens[r]] = stage[r][:lens[r]]
        self._pending = []
        return self.out
