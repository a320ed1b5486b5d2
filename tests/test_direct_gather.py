"""One-shot direct all-gather: the regressor epilogue stores every output row to several
destinations -- on a node, this rank's slot in each peer GPU's gathered buffer.  One GPU is enough to pin the kernels: the
destinations are simply other buffers; two ranks sharing the GPU pin the IPC mapping and the bookkeeping."""
import os
import socket

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("head", ["dgrad", "offsets"])
def test_every_destination_gets_identical_rows(synth_sd, head):
    from sdfa_amd.engine import Engine
    eng = Engine(synth_sd[head], max_frames=256)
    rs = np.random.RandomState(8)
    n = 300                                                   # two chunks of the 256-frame workspace, ragged tail
    z = torch.from_numpy(rs.normal(0, 1, (n, 512)).astype(np.float32)).cuda()
    spk = torch.from_numpy(rs.randint(0, 8, n)).cuda()
    _, ref = eng.regress(z, spk)
    for k in (1, 2, 8):
        outs = [torch.full((n, eng.out_dim), float("nan"), device="cuda") for _ in range(k)]
        coef = eng.regress_multi(z, spk, outs, want_coef=True)
        for o in outs:
            assert torch.equal(o, ref), (head, k)
        assert coef.shape == (n, eng.coef_dim)
    from sdfa_amd._lib import SdfaError
    with pytest.raises(SdfaError):
        eng.regress_multi(z, spk, [torch.empty((n, eng.out_dim), device="cuda") for _ in range(9)])


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _rank(rank, world, port, q):
    import sys
    import torch.distributed as dist
    here = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, os.path.join(here, "sdfa-2019_amd"))
    from sdfa_amd import dist as sd, synth
    from sdfa_amd.engine import Engine
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(0)
    dist.init_process_group("gloo", rank=rank, world_size=world)           # control only; the rows move by peer stores
    eng = Engine(synth.make_state_dict("offsets", 1234), max_frames=256)
    counts = [130, 77]                                                      # ragged shards
    n = counts[rank]
    rs = np.random.RandomState(50 + rank)
    z = torch.from_numpy(rs.normal(0, 1, (n, 512)).astype(np.float32)).cuda()
    spk = torch.from_numpy(rs.randint(0, 8, n)).cuda()
    g = sd.DirectGatherer(counts, eng.out_dim, "cuda:0")
    held = []
    for step in range(3):                                                   # step-varying rows: a stale or overwritten buffer would show
        zs = z * float(step + 1)
        g.begin_step()
        sd.run_chunks(n, 64, None, lambda f0, f1: eng.regress_multi(zs[f0:f1], spk[f0:f1], g.dest_views(f0, f1), check_ids=False))
        g.finish()
        held.append(g.gathered())                                           # a consumer that still holds step k's rows during step k+1
        if step >= 1:                                                       # the previous step's gathered rows are still intact
            _, prev = eng.regress(z * float(step), spk)
            lo = sum(counts[:rank])
            assert torch.equal(held[step - 1][lo: lo + n], prev)
    _, mine = eng.regress(z * 3.0, spk)
    q.put((rank, mine.cpu().numpy(), g.gathered().cpu().numpy()))
    dist.barrier()
    dist.destroy_process_group()


def test_two_ranks_one_gpu_peer_mapped_buffers():
    import torch.multiprocessing as mp
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_rank, args=(r, 2, port, q)) for r in range(2)]
    for p in procs: p.start()
    res = sorted([q.get(timeout=600) for _ in procs], key=lambda t: t[0])
    for p in procs: p.join(timeout=120)
    want = np.concatenate([res[0][1], res[1][1]])
    for rank, _, gathered in res:
        assert np.array_equal(gathered, want), rank
