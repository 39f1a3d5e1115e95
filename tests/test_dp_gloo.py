"""Data-parallel reducer (diverse_channel_vit_amd/dp.py) on CPU with the gloo backend, world_size 2.
The N>1 path of bench.py uses exactly this class with RCCL; here every rank fills a flat gradient
arena with rank-dependent values, reports layer slices in reverse order as the model's backward does,
and checks the averaged result, the bucket merging, and the hooks for parameters outside the arena."""
import os
import socket
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


class _Stub(torch.nn.Module):
    """Mimics the attributes DataParallel touches on the arena-backed model."""

    def __init__(self):
        super().__init__()
        self.inside = torch.nn.Parameter(torch.zeros(10))
        self.misc = torch.nn.Linear(4, 3)
        self.unused = torch.nn.Parameter(torch.ones(5))  # like `proxies` in CE mode: never gets a grad
        self._enc_params = [self.inside]
        self._arena = None
        self._dp = None


def _worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diverse_channel_vit_amd.dp import DataParallel
        torch.manual_seed(100 + rank)
        model = _Stub()
        dp = DataParallel(model, min_bucket_bytes=4 * 3000)
        dp.broadcast_parameters(0)
        w0 = model.misc.weight.detach().clone()
        dp.hook_misc_params()
        # arena of 6 "layers" x 1000 floats + head/tail, reported last layer first
        n_layers, per = 6, 1000
        arena = torch.arange(n_layers * per + 200, dtype=torch.float32) * (rank + 1)
        expect = torch.arange(n_layers * per + 200, dtype=torch.float32) * (sum(range(1, world + 1)) / world)
        dp.grad_ready(arena, n_layers * per + 100, n_layers * per + 200)  # final norm (small, pending)
        for l in range(n_layers - 1, -1, -1):
            dp.grad_ready(arena, 100 + l * per, 100 + (l + 1) * per)
        dp.grad_ready(arena, 0, 100)
        dp.flush()
        # misc params through autograd hooks
        x = torch.full((2, 4), float(rank + 1))
        model.misc(x).sum().backward()
        g_local = model.misc.weight.grad.clone()
        dp.finalize()
        ok_arena = torch.allclose(arena, expect, rtol=1e-6, atol=1e-3)
        # untouched gap [layers end, +100) must be untouched: ranges are exact
        exp_w = torch.full((3, 4), 2.0 * (sum(range(1, world + 1)) / world))
        ok_misc = torch.allclose(model.misc.weight.grad, exp_w)
        ok_bcast = torch.equal(w0, model.misc.weight.detach())
        gathered = [torch.zeros_like(w0) for _ in range(world)]
        dist.all_gather(gathered, w0)
        ok_same = all(torch.equal(gathered[0], g) for g in gathered)
        q.put((rank, ok_arena, ok_misc, ok_bcast and ok_same, model.unused.grad is None, dp.buckets_launched))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_dp_reducer_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_arena, ok_misc, ok_bcast, unused_none, nb in res:
        assert ok_arena, f"rank {rank}: arena average wrong"
        assert ok_misc, f"rank {rank}: misc parameter average wrong"
        assert ok_bcast, f"rank {rank}: broadcast_parameters did not equalise weights"
        assert unused_none
        # 6 layers x 4 KB merged into >= 12 KB buckets + head + tail + 2 hooked params: fewer collectives than slices
        assert 4 <= nb <= 8, nb


class _Fixed(torch.nn.Module):
    """A stand-in with the plugin's forward signature: logits that make the first `k` samples of a batch correct."""

    def forward(self, x, chunk_name, training_chunks=None, init_first_layer=None, new_channel_init=None):
        out = torch.zeros(x.shape[0], 4)
        out[torch.arange(x.shape[0]), x[:, 0].long() % 4] = 1.0
        return out


def _eval_worker(rank, world, port, q):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from diverse_channel_vit_amd.checkpoint import evaluate
        # rank 0: 6 samples, 4 right; rank 1: 10 samples, 3 right -> 7/16 over all ranks (not the mean of the two ratios)
        n, right = (6, 4) if rank == 0 else (10, 3)
        y = torch.arange(n) % 4
        x = y.clone().float()
        x[right:] += 1.0
        acc = evaluate(_Fixed(), [(x[:, None], y)], "test")
        q.put((rank, acc))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(300)
def test_evaluate_sums_counts_over_ranks_gloo_world2():
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_eval_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = [q.get(timeout=240) for _ in range(world)]
    for p in procs:
        p.join(60)
    for rank, acc in res:
        assert abs(acc - 100.0 * 7 / 16) < 1e-9
