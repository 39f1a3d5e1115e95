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


def _collect(procs, q, world, timeout=400):
    """Results of all ranks; fails as soon as a rank dies instead of waiting for the queue to time out."""
    import queue as _q
    import time as _t
    res, t0 = [], _t.time()
    while len(res) < world:
        try:
            res.append(q.get(timeout=2))
        except _q.Empty:
            dead = [p.exitcode for p in procs if p.exitcode not in (None, 0)]
            if dead or _t.time() - t0 > timeout:
                for p in procs:
                    if p.is_alive():
                        p.terminate()
                pytest.fail(f"a rank exited with {dead} (or timed out) before reporting")
    return res


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

    def _enc_param_list(self):
        return [self.inside]


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
    res = _collect(procs, q, world)
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


# ----------------------------------------------------------------------------------------------------------------------
# The REAL DiChaViT object (parameter arena, gradient arena, autograd node, _run_backward's data-parallel logic) under
# DataParallel on two gloo ranks.  Only the two kernel-sequence methods are replaced by CPU stand-ins that produce
# deterministic, rank- and pass-dependent "gradients" in the arena and report the layer slices exactly as the HIP
# backward does (final norm, blocks last to first, tokeniser).  Follows INTEGRATION.md's call order (hooks BEFORE the
# first forward), runs TWO backward passes per optimiser step (the CHAMMI step, trainer.py:846-935) and never calls
# finalize() itself: whatever comes after loss.backward() must already see reduced gradients.
def _real_model_worker(rank, world, port, q, grad_dtype, overlap):
    sys.path.insert(0, ROOT)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import diverse_channel_vit_amd as dcv
        from diverse_channel_vit_amd.dichavit import _EncoderFn

        class Cfg(dict):
            __getattr__ = dict.get

        cfg = Cfg(name="dichavit", pretrained_model_name="tiny", patch_size=8, temperature=0.07, learnable_temp=False, enable_sample=False,
                  use_channelvit_channels=True, orthogonal_channel_emb_init=True, dropout_tokens_hcs="none", freeze_channel_emb=False,
                  block_type="block", hcs_sampling="none", hcs_sampling_temp=0.1, proxy_loss_lambda=0.001, ortho_loss_v1_lambda=0.001,
                  drop_path_rate=0.0, gamma_s=1.0, gamma_d=4.0, reverse_pos_pairs=True, use_square=False,
                  in_channel_names=["a", "b", "c"], img_size=[32], num_classes=5)
        torch.manual_seed(7 + rank)  # different initial weights per rank: broadcast_parameters must equalise them
        model = dcv.dichavit(cfg, mapper={"train": [0, 1, 2]})
        dp = dcv.DataParallel(model, min_bucket_bytes=1 << 20, grad_dtype=grad_dtype, overlap=overlap)
        dp.broadcast_parameters(0)
        dp.hook_misc_params()  # BEFORE the arena exists (INTEGRATION.md order)
        n_hooked = len(dp._hooks)
        model._ensure_arena(torch.device("cpu"))
        dp.broadcast_parameters(0)
        fe = model.feature_extractor
        D, B = model.dim, 2
        state = {"pass": 0}

        def fake_forward(x, ch_idx_dev, C, E, pos_tab, want_ortho, save, keep=None, st_scale=None, st_shift=None, tok=None):
            return dict(feat=torch.ones(B, D), stats=torch.zeros(B, 2), C=C, n=16)

        def fake_backward_body(st, dfeat, dstats, ga, g, dp_, nt_kw):
            k = state["pass"]
            state["pass"] += 1
            base = torch.arange(ga.numel(), dtype=torch.float32) % 97
            ga.copy_(base * (rank + 1) * 0.25 + (k + 1))  # exactly representable in bf16 too (small integers and quarters)
            dp_.grad_ready(ga, *model._range_of([fe.norm.weight, fe.norm.bias]))
            for li in range(len(fe.blocks) - 1, -1, -1):
                blk = fe.blocks[li]
                dp_.grad_ready(ga, *model._range_of([blk.norm1.weight, blk.mlp.fc2.bias]))
            dp_.grad_ready(ga, *model._range_of([fe.cls_token, fe.patch_embed.proj.bias]))
            dp_.flush()
            grads = [model._gview(ga, p) for p in model._enc_params]
            return torch.full((st["C"], D), float(rank + 1)), torch.full((st["n"] + 1, D), float(2 * rank + 1)), grads

        model._run_forward = fake_forward
        model._run_backward_body = fake_backward_body
        x = torch.zeros(B, 3, 32, 32, requires_grad=False)
        head = model.classifer_head
        for k in range(2):  # two backward passes, gradients accumulate
            E = fe.patch_embed.channel_embed(torch.tensor([0, 1, 2]))
            pos_tab = model._pos_table(3, 16, 32, 32)
            feat, stats = _EncoderFn.apply(model, None, 3, False, None, None, x, E, pos_tab, *model._enc_params)
            loss = head(feat).sum() * (rank + 1) + stats.sum()
            loss.backward()  # no finalize() here: the backward is self-synchronising
        assert not dp._works and dp._pending is None and not dp._deferred
        # expectation: sum over the two passes of the rank AVERAGE of each pass's arena
        n = model._enc_size
        base = (torch.arange(n, dtype=torch.float32) % 97)
        avg_rank = sum(r + 1 for r in range(world)) / world
        expect = base * avg_rank * 0.25 * 2 + (1 + 2)
        got = torch.cat([p.grad.reshape(-1) for p in model._enc_params])
        sizes = [p.numel() for p in model._enc_params]
        want = torch.cat([expect[o:o + s_] for o, s_ in zip(model._enc_off, sizes)])
        # bf16 exchange: the rank sum is rounded to 8 significant bits (twice: per pass)
        ok_enc = torch.allclose(got, want, rtol=0 if grad_dtype == torch.float32 else 2.0 ** -7, atol=1e-5)
        # channel_embed: rows get dE = rank+1 per pass -> average over ranks, times 2 passes (a parameter outside the arena node)
        ce = fe.patch_embed.channel_embed.weight.grad
        ok_misc = torch.allclose(ce, torch.full_like(ce, 2 * avg_rank), atol=1e-5 if grad_dtype == torch.float32 else 2e-2)
        hw = head.weight.grad  # d/dW of sum(head(ones[B, D])) * (rank+1) = B * (rank+1) per entry per pass
        ok_head = torch.allclose(hw, torch.full_like(hw, 2 * B * avg_rank), atol=1e-5 if grad_dtype == torch.float32 else 4e-2)
        # all ranks identical afterwards
        flat = torch.cat([got, ce.reshape(-1), hw.reshape(-1)])
        gathered = [torch.zeros_like(flat) for _ in range(world)]
        dist.all_gather(gathered, flat)
        ok_same = all(torch.equal(gathered[0], t) for t in gathered)
        q.put((rank, ok_enc, ok_misc, ok_head, ok_same, n_hooked, dp.buckets_launched, model.proxies.grad is None))
    finally:
        dist.destroy_process_group()


@pytest.mark.timeout(600)
@pytest.mark.parametrize("grad_dtype,overlap", [("float32", True), ("bfloat16", True), ("float32", False)])
def test_real_model_two_backward_passes_gloo_world2(grad_dtype, overlap):
    world = 2
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_real_model_worker, args=(r, world, port, q, getattr(torch, grad_dtype), overlap)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect(procs, q, world)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, ok_enc, ok_misc, ok_head, ok_same, n_hooked, nb, proxies_none in res:
        assert ok_enc, f"rank {rank}: encoder gradients are not the sum over passes of the rank averages"
        assert ok_misc and ok_head, f"rank {rank}: parameters outside the arena wrong"
        assert ok_same, f"rank {rank}: ranks diverged"
        assert proxies_none
        # only the handful of parameters outside the encoder arena carry hooks (pos_embed, channel_embed, channel_emb_proxies,
        # classifer_head.{weight,bias}, proxies) — not the ~150 encoder parameters
        assert n_hooked <= 8, n_hooked
        assert nb >= 2


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
    res = _collect(procs, q, world)
    for p in procs:
        p.join(60)
    for rank, acc in res:
        assert abs(acc - 100.0 * 7 / 16) < 1e-9


def test_begin_forward_recovers_after_a_failed_backward():
    """the autograd engine skips its end-of-backward callbacks when a backward pass raises, which used to leave the reducer's
    `_callback_queued` flag set for good (no later pass would queue finalize()).  begin_forward() — called by the model at the start of every
    training forward — clears the flag and the state the failed pass left behind."""
    import torch.distributed as dist
    from diverse_channel_vit_amd.dp import DataParallel

    class Stub:
        _dp = None

        def parameters(self):
            return []

        def _enc_param_list(self):
            return []

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dp = DataParallel(Stub(), force_collectives=True)
        t = torch.ones(8)
        dp._callback_queued = True          # a backward queued the callback, then raised: the engine never ran it
        dp._pending = (t, 0, 4)
        dp._deferred = [t[:2]]

        class W:
            def wait(self):
                raise RuntimeError("the failed pass's collective")

        dp._works.append((W(), t, None))
        dp.begin_forward()
        assert not dp._callback_queued and dp._pending is None and not dp._deferred and not dp._works
        dp.begin_forward()                  # a clean state is left alone
        dp.grad_ready(t, 0, 8)
        dp.finalize()
        assert torch.equal(t, torch.ones(8))
    finally:
        dist.destroy_process_group()


def test_begin_forward_inside_a_running_backward_is_left_alone():
    """a training forward that legitimately runs INSIDE a backward pass (a checkpoint recompute, a forward called from a
    hook after the callback was queued) sees `_callback_queued` set; that is not a failed pass and nothing may be dropped — the in-flight
    buckets of the running pass must still be waited for by its own callback."""
    import torch.distributed as dist
    from diverse_channel_vit_amd.dp import DataParallel

    class Stub:
        _dp = None

        def parameters(self):
            return []

        def _enc_param_list(self):
            return []

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(_free_port()))
    dist.init_process_group("gloo", rank=0, world_size=1)
    try:
        dp = DataParallel(Stub(), force_collectives=True)
        arena = torch.ones(8)
        seen = {}

        class Probe(torch.autograd.Function):
            @staticmethod
            def forward(ctx, x):
                return x * 1.0

            @staticmethod
            def backward(ctx, g):
                dp.queue_finalize()                 # what the model's node does first
                dp.grad_ready(arena, 0, 8)          # a bucket of this pass is in flight / pending
                assert dp._callback_queued
                dp.begin_forward()                  # the recompute's forward
                seen["queued_after"] = dp._callback_queued
                seen["task"] = DataParallel._current_task()
                return g

        x = torch.ones(3, requires_grad=True)
        Probe.apply(x).sum().backward()
        assert seen["task"] != -1, "this torch build does not expose the running graph task: the guard cannot work"
        assert seen["queued_after"] is True           # left alone inside the pass ...
        assert dp._callback_queued is False           # ... and the pass's own callback ran at its end
        assert not dp._works and dp._pending is None
    finally:
        dist.destroy_process_group()


def _poison_worker(rank, world, port, q):
    try:
        import torch.distributed as dist
        from diverse_channel_vit_amd.dp import DataParallel
        os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        dist.init_process_group("gloo", rank=rank, world_size=world)

        class Stub:
            _dp = None

            def parameters(self):
                return []

            def _enc_param_list(self):
                return []

        dp = DataParallel(Stub())
        dp._callback_queued = True  # a backward pass queued the callback and then raised on this rank
        msgs = []
        for _ in range(2):          # this forward AND every later one
            try:
                dp.begin_forward()
                msgs.append("no error")
            except RuntimeError as e:
                msgs.append(str(e))
        q.put((rank, msgs))
        dist.destroy_process_group()
    except Exception as e:  # noqa: BLE001
        q.put((rank, ["worker failed: %r" % (e,)]))


def test_failed_backward_poisons_the_reducer_at_world_size_two():
    """at world size > 1 a rank cannot repair a failed backward alone (its peers issued every bucket of that step);
    the reducer must refuse to continue instead of resetting silently."""
    world, port = 2, _free_port()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_poison_worker, args=(r, world, port, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = _collect(procs, q, world)
    for p in procs:
        p.join(60)
    for rank, msgs in res:
        assert len(msgs) == 2 and all("did not finish" in m and "restart" in m for m in msgs), (rank, msgs)


def test_merge_views_of_one_arena():
    """DataParallel._merge_views (overlap=False): neighbouring 1-D views of one arena become ONE view over their union, whatever order they
    arrive in; views that do not touch stay separate; tensors that are not contiguous 1-D views of a base are returned as they are."""
    from diverse_channel_vit_amd.dp import DataParallel
    arena = torch.arange(100, dtype=torch.float32)
    other = torch.zeros(5)
    ts = [arena[60:80], arena[40:60], arena[10:20], other, arena[80:100], torch.ones(2, 3).t()[0]]  # the last one: a strided view
    merged, loose = DataParallel._merge_views(ts)
    spans = sorted((int(m[0].item()), m.numel()) for m in merged)
    assert spans == [(10, 10), (40, 60)]
    assert all(m.data_ptr() == arena.data_ptr() + 4 * int(m[0].item()) for m in merged)  # views, not copies
    assert len(loose) == 2 and loose[0] is other and not loose[1].is_contiguous()
    merged[1 if merged[0].numel() == 10 else 0].mul_(0)  # writing the merged view writes the arena
    assert float(arena[40:100].abs().sum()) == 0.0 and float(arena[10:20].sum()) == sum(range(10, 20))

