"""world_size-2 gloo test of the N>1 path's sharding and chunked all-gather bookkeeping (CPU tensors)."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, n_clips, frames_per_clip, width, chunk, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "sdfa-2019_amd"))
    from sdfa_amd import dist as sd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lo, hi = sd.shard_range(n_clips, rank, world)
    # stand-in for the per-frame output: row value encodes (global clip, frame-in-clip, column)
    rows = []
    for c in range(lo, hi):
        for f in range(frames_per_clip[c]):
            rows.append(c * 1000.0 + f + np.arange(width) * 1e-3)
    local = torch.tensor(np.asarray(rows, np.float32).reshape(-1, width))
    counts = sd.frame_counts_all(local.shape[0])
    g = sd.FrameGatherer(counts, width, torch.float32, "cpu", chunk)
    calls = []
    for step in range(2):       # two "steps" of bench.py's loop shape: a rank that skipped a collective would mispair here
        g.buf.zero_()
        sd.run_chunks(local.shape[0], chunk, g, lambda f0, f1: (calls.append((f0, f1)), local[f0:f1])[1])
    assert all(f1 > f0 for f0, f1 in calls)
    out = g.gathered()
    per_rank = torch.cat(g.rows(1), 0)
    assert per_rank.shape[0] == counts[1]
    q.put((rank, counts, out.numpy()))
    dist.barrier()
    dist.destroy_process_group()


# [9, 3]: rank 0 holds 3 chunks of 4 frames, rank 1 one -- different chunk counts per rank (the bench loop hung there)
@pytest.mark.parametrize("frames_per_clip", [[5, 5, 5, 5], [3, 9, 4, 1, 7], [9, 3], [2, 11]])
def test_sharded_gather_reassembles_clip_order(frames_per_clip):
    world, width, chunk = 2, 6, 4
    n_clips = len(frames_per_clip)
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_clips, frames_per_clip, width, chunk, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    expect = np.asarray([c * 1000.0 + f + np.arange(width) * 1e-3 for c in range(n_clips) for f in range(frames_per_clip[c])], np.float32)
    for rank, counts, out in res:
        assert sum(counts) == len(expect)
        assert np.array_equal(out, expect), rank


class _LinearEngine:
    """Stand-in for Engine on CPU: the expansion is a fixed linear map of the coefficients, as the PCA stage is."""
    coef_dim, out_dim = 5, 12

    def __init__(self):
        rs = np.random.RandomState(3)
        self.basis = torch.tensor(rs.normal(0, 1, (self.coef_dim, self.out_dim)).astype(np.float32))
        self.mean = torch.tensor(rs.normal(0, 1, self.out_dim).astype(np.float32))

    def expand_coef(self, coef, out=None):
        r = coef @ self.basis + self.mean
        if out is None:
            return r
        out.copy_(r)
        return out


def _expand_worker(rank, world, port, counts_in, chunk, q):
    import sys
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "sdfa-2019_amd"))
    from sdfa_amd import dist as sd
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    eng = _LinearEngine()
    n = counts_in[rank]
    coef = torch.tensor(np.random.RandomState(100 + rank).normal(0, 1, (n, eng.coef_dim)).astype(np.float32))
    counts = sd.frame_counts_all(n)
    g = sd.ExpandGatherer(counts, eng, "cpu", chunk)
    for step in range(2):
        g.buf.fill_(float("nan"))

        def compute(f0, f1):
            eng.expand_coef(coef[f0:f1], out=g.own(f0, f1))         # the regressor writes this rank's rows in place
            return coef[f0:f1]
        sd.run_chunks(n, chunk, g, compute)
    q.put((rank, counts, g.gathered().numpy().copy()))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("counts", [[8, 8], [9, 3], [2, 11]])
def test_expand_gather_rebuilds_every_ranks_rows(counts):
    """ExpandGatherer: coefficients travel, rows are rebuilt on every rank, rank order, even and ragged shards."""
    world, chunk = 2, 4
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_expand_worker, args=(r, world, port, counts, chunk, q)) for r in range(world)]
    for p in procs: p.start()
    res = [q.get(timeout=120) for _ in procs]
    for p in procs: p.join(timeout=60)
    eng = _LinearEngine()
    expect = np.concatenate([(torch.tensor(np.random.RandomState(100 + r).normal(0, 1, (counts[r], 5)).astype(np.float32)) @ eng.basis + eng.mean).numpy()
                             for r in range(world)])
    for rank, got_counts, out in res:
        assert got_counts == counts
        assert np.array_equal(out, expect), rank


def test_shard_range_partitions():
    import sys
    from sdfa_amd.dist import shard_range
    for n in (0, 1, 7, 32, 256, 257):
        for w in (1, 2, 3, 8):
            spans = [shard_range(n, r, w) for r in range(w)]
            assert spans[0][0] == 0 and spans[-1][1] == n
            assert all(spans[i][1] == spans[i + 1][0] for i in range(w - 1))
            sizes = [b - a for a, b in spans]
            assert max(sizes) - min(sizes) <= 1
