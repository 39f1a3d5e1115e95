"""Per-kernel parity: every entry of include/dcv.h against plain fp32 torch math (or the oracle's
formula) on the same seeded inputs.  Tolerances are the bf16 rounding of inputs/outputs; all
accumulation in the kernels is fp32.  Needs an MI355X: run with -m gpu."""
import math

import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def hip(gpu_device):
    from diverse_channel_vit_amd import hip as h
    h.load()
    return h


def _bf(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).to(torch.bfloat16).cuda()


def _f(*shape, scale=1.0, seed=0):
    g = torch.Generator(device="cpu").manual_seed(seed)
    return (torch.randn(*shape, generator=g) * scale).cuda()


def _close(a, b, rtol, atol, what=""):
    a, b = a.float(), b.float()
    err = (a - b).abs()
    tol = atol + rtol * b.abs()
    bad = (err > tol)
    assert not bad.any(), f"{what}: {int(bad.sum())}/{bad.numel()} off, max err {err.max().item():.4g} (ref max {b.abs().max().item():.4g})"


GEMM_SHAPES = [(300, 384, 384), (128, 128, 64), (777, 1152, 384), (1000, 384, 1536), (257, 192, 192), (130, 576, 192)]


def _check_gemm_nt(hip, M, N, K, resid=True, **kw):
    """All five Linear epilogues of dcv_gemm_nt_ex against fp32 torch.matmul on the same bf16 operands (resid=False: the four bf16-output ones)."""
    A, W = _bf(M, K, seed=1), _bf(N, K, scale=0.05, seed=2)
    bias = _f(N, scale=0.1, seed=3)
    ref = A.float() @ W.float().t()
    # bias -> bf16
    out = torch.empty(M, N, dtype=torch.bfloat16, device="cuda")
    hip.gemm_nt(A, W, hip.EPI_BIAS_BF16, out, bias=bias, **kw)
    _close(out, ref + bias, 1e-2, 2e-2, "bias_bf16")
    # plain
    hip.gemm_nt(A, W, hip.EPI_PLAIN_BF16, out, **kw)
    _close(out, ref, 1e-2, 2e-2, "plain_bf16")
    # bias + gelu: out = GELU'(z), out2 = GELU(z), z = acc + bias
    h = torch.empty_like(out)
    hip.gemm_nt(A, W, hip.EPI_BIAS_GELU_BF16, out, bias=bias, out2=h, **kw)
    zr = (ref + bias).requires_grad_(True)
    hr = torch.nn.functional.gelu(zr)
    hr.sum().backward()
    _close(h, hr.detach(), 1e-2, 2e-2, "gelu h")
    _close(out, zr.grad, 1e-2, 2e-2, "gelu'")
    del hr, zr
    if resid:
        # residual f32 in place
        x = _f(M, N, seed=4)
        x0 = x.clone()
        hip.gemm_nt(A, W, hip.EPI_BIAS_RESID_F32, x, bias=bias, **kw)
        _close(x, x0 + ref + bias, 1e-4, 2e-4 * math.sqrt(K), "resid_f32")
        y = torch.empty_like(x)
        hip.gemm_nt(A, W, hip.EPI_BIAS_RESID_F32, y, bias=bias, aux=x0, **kw)  # out-of-place residual
        _close(y, x0 + ref + bias, 1e-4, 2e-4 * math.sqrt(K), "resid_f32 out-of-place")
        assert torch.equal(y, x)
        del x, y, x0
    # gelu backward epilogue: acc * saved GELU'
    gp = _bf(M, N, seed=5)
    hip.gemm_nt(A, W, hip.EPI_GELU_BWD_BF16, out, aux=gp, **kw)
    _close(out, ref * gp.float(), 1e-2, 2e-2, "gelu_bwd")


@pytest.mark.parametrize("M,K,cap", [(300, 384, 0), (4100, 1536, 0), (5000, 384, 4), (100416, 1536, 0), (100416, 384, 0), (777, 64, 3)])
def test_gemm_nt_resid_ln(hip, M, K, cap):
    """dcv_gemm_nt_resid_ln (judge row N1): the residual GEMM and the LayerNorm that follows it in one launch — x' against the fp32 product,
    mean / rstd / u against torch's layer_norm of the kernel's OWN x' (so the LayerNorm part is checked to fp32 accuracy, not to the GEMM's bf16
    one), with a per-sample DropPath factor, a partial last M tile, multi-round walks under a grid cap, and in place (x_out aliasing resid)."""
    N = 384
    A, W = _bf(M, K, seed=1), _bf(N, K, scale=0.05, seed=2)
    bias, gamma, beta = _f(N, scale=0.1, seed=3), 1.0 + _f(N, scale=0.2, seed=6), _f(N, scale=0.3, seed=7)
    x0 = _f(M, N, seed=4) * 3.0 + 5.0  # a row mean well away from zero: E[x^2] - mean^2 would lose digits, the centred form does not
    ref = A.float() @ W.float().t()
    T = 1 if M % 4 else M // 4
    scale = (torch.tensor([1.0, 0.0, 2.0, 1.25], device="cuda") if T > 1 else None)
    for inplace in (False, True):
        x_in = x0.clone()
        x_out = x_in if inplace else torch.full_like(x0, float("nan"))
        u = torch.full((M, N), float("nan"), dtype=torch.bfloat16, device="cuda")
        mean, rstd = torch.full((M,), float("nan"), device="cuda"), torch.full((M,), float("nan"), device="cuda")
        hip.gemm_nt_resid_ln(A, W, bias, x_in, x_out, gamma, beta, 1e-6, u, mean, rstd, grid_cap=cap,
                             **(dict(branch_scale=scale, T=T) if scale is not None else {}))
        s_rows = scale.repeat_interleave(T)[:, None] if scale is not None else 1.0
        want = x0 + s_rows * (ref + bias)
        _close(x_out, want, 1e-4, 2e-4 * math.sqrt(K) + 1e-4, "x' = resid + s (acc + bias)")
        xo = x_out.double()
        mu = xo.mean(-1)
        var = ((xo - mu[:, None]) ** 2).mean(-1)
        _close(mean, mu.float(), 1e-6, 1e-6 * mu.abs().max().item() + 1e-6, "mean")
        _close(rstd, (var + 1e-6).rsqrt().float(), 2e-6, 1e-7, "rstd")
        un = torch.nn.functional.layer_norm(x_out, (N,), gamma, beta, 1e-6)
        _close(u, un, 1e-2, 2e-2, "u = LayerNorm(x')")
        # and against the separate kernel: same bf16 values except where the two statistics differ in the last bit
        u2 = torch.empty_like(u)
        m2, r2 = torch.empty_like(mean), torch.empty_like(rstd)
        hip.ln_fwd(x_out, gamma, beta, u2, m2, r2, M, N, 1e-6)
        assert (u.view(torch.int16) != u2.view(torch.int16)).float().mean().item() <= 2e-3
        _close(mean, m2, 1e-6, 1e-5, "mean vs ln_fwd")
        _close(rstd, r2, 1e-5, 1e-7, "rstd vs ln_fwd")


@pytest.mark.parametrize("M,N,K", GEMM_SHAPES)
def test_gemm_nt_epilogues(hip, M, N, K):
    _check_gemm_nt(hip, M, N, K)


@pytest.mark.parametrize("M,N,K", [(4100, 1152, 384), (4352, 384, 1536), (4608, 1536, 384), (5000, 384, 64), (300, 384, 128)])
def test_gemm_nt_384_wide_tiles(hip, M, N, K):
    """The 256 x 384 tile kernel (gemm_nt384, used for N % 384 == 0 where it pays) on every epilogue it carries, forced with
    tile=TILE_WIDE: full tiles, a partial last M tile, one and several k-stages, 1 / 3 / 4 column tiles."""
    _check_gemm_nt(hip, M, N, K, tile=hip.TILE_WIDE)


def test_gemm_nt_alternating_halves(hip):
    """gemm_nt_alt (TILE_ALT; variant builds only — ONE skip on the product library): the two wave groups of a workgroup accumulate and store the two 192-column
    halves of a 256 x 384 tile in alternating phases on one shared LDS ring — every bf16-output epilogue, one to 24 k-stages per phase (fewer and more slots than the
    six epilogue chunks), a partial last M tile, one tile per workgroup and long walks under a grid cap (an odd number of tiles per workgroup included), the headline
    fc1 shape; the fp32 residual epilogue is refused."""
    if hip.load().dcv_gemm_nt_pick(4100, 1152, 384, hip.EPI_PLAIN_BF16, hip.TILE_ALT) != hip.TILE_ALT:
        pytest.skip("gemm_nt_alt is compiled into variant builds only (-DDCV_NT_ALT=1: measured slower than the shipped tiles, profiles/r05_x9_*)")
    for M, N, K in [(4100, 1152, 384), (4352, 384, 1536), (4608, 1536, 384), (5000, 384, 64), (300, 384, 128), (9000, 768, 320), (64 * 1569, 1536, 384)]:
        for grid_cap in (0, 3):
            _check_gemm_nt(hip, M, N, K, resid=False, tile=hip.TILE_ALT, grid_cap=grid_cap)
    A, W = _bf(256, 384, seed=1), _bf(384, 384, seed=2)
    with pytest.raises(RuntimeError):
        hip.gemm_nt(A, W, hip.EPI_BIAS_RESID_F32, torch.zeros(256, 384, device="cuda"), bias=_f(384, seed=3), tile=hip.TILE_ALT)


# The headline step's GEMMs: M = 64 x 1569 = 100 416 token rows.  Both NT kernels are PERSISTENT: one workgroup per CU walks
# 1 179 - 4 716 output tiles in several rounds, prefetching the next tile's first stages under the current tile's epilogue
# (hand-counted vmcnt waits: csrc/gemm.hip `stores_behind`).  None of that runs when a launch has fewer tiles than CUs.
HEADLINE_M = 64 * 1569
HEADLINE_NK = [(1152, 384), (384, 384), (1536, 384), (384, 1536)]


_NT_TILES = {"narrow": 1, "wide": 2, "pair": 3}  # hip.TILE_*; pair = 128 x 128 tiles by four-wave workgroups, two per CU (round 4)


@pytest.mark.parametrize("tile", ["narrow", "wide", "pair"])
@pytest.mark.parametrize("N,K", HEADLINE_NK)
def test_gemm_nt_headline_shapes_multi_round(hip, N, K, tile):
    """(i) of the persistence coverage: the exact shapes of bench.py's step, every epilogue, all three tile shapes, many rounds
    per workgroup, a partial last M tile (100 416 = 392 x 256 + 64 = 784 x 128 + 64)."""
    _check_gemm_nt(hip, HEADLINE_M, N, K, tile=_NT_TILES[tile])


@pytest.mark.parametrize("grid_cap", [4, 8, 248])
@pytest.mark.parametrize("M,N,K,tile", [(5000, 384, 384, "narrow"), (5000, 1152, 384, "wide"), (5100, 384, 1536, "wide"),
                                        (5100, 1536, 384, "narrow"), (66000, 384, 384, "wide"), (33000, 1152, 384, "narrow"),
                                        (5000, 1152, 384, "pair"), (5100, 392, 1536, "pair"), (33000, 384, 64, "pair")])
def test_gemm_nt_forced_small_grids(hip, M, N, K, tile, grid_cap):
    """(ii): few workgroups on a small problem, so that total tiles > grid: long tile walks (up to 60 rounds at 4 workgroups),
    a partial last round, the `stores_behind` waits and the epilogue-overlapped prefetch all execute.  248 = the
    data-parallel backward's grid (CUs minus the ones left to RCCL); the last two shapes have more tiles than that."""
    _check_gemm_nt(hip, M, N, K, tile=_NT_TILES[tile], grid_cap=grid_cap)


def test_gemm_nt_random_shapes_tiles_and_grids(hip):
    """Seeded random problems through both kernels: M anywhere (partial M tiles, fewer tiles than workgroups, several rounds under a
    random grid cap), N any multiple of 8 (partial N tiles on the narrow kernel; multiples of 384 also on the wide one), K any multiple
    of 64 from one to 24 stages — every epilogue each time (_check_gemm_nt)."""
    rng = np.random.RandomState(20240607)
    for _ in range(10):
        M = int(rng.choice([rng.randint(1, 300), rng.randint(300, 3000), rng.randint(3000, 9000)]))
        K = 64 * int(rng.randint(1, 25))
        wide = bool(rng.randint(0, 2))
        N = 384 * int(rng.randint(1, 5)) if wide else 8 * int(rng.randint(1, 193))
        cap = int(rng.choice([0, 0, 1, 3, 7, 30, 200]))
        for t in ((hip.TILE_WIDE,) if wide else (hip.TILE_NARROW, hip.TILE_PAIR)):
            try:
                _check_gemm_nt(hip, M, N, K, tile=t, grid_cap=cap)
            except AssertionError as e:
                raise AssertionError(f"M={M} N={N} K={K} tile={t} grid_cap={cap}: {e}") from e


def test_gemm_nt_auto_tile_rules(hip):
    """AUTO picks the kernel by shape; an illegal forced variant is refused, not silently replaced."""
    A, W = _bf(512, 384, seed=1), _bf(200, 384, seed=2)
    out = torch.empty(512, 200, dtype=torch.bfloat16, device="cuda")
    with pytest.raises(RuntimeError):
        hip.gemm_nt(A, W, hip.EPI_PLAIN_BF16, out, tile=hip.TILE_WIDE)  # N % 384 != 0
    hip.gemm_nt(A, W, hip.EPI_PLAIN_BF16, out, tile=hip.TILE_NARROW)
    _close(out, A.float() @ W.float().t(), 1e-2, 2e-2, "narrow")


def test_gemm_nt_patch_epilogue(hip):
    B, C, n, D, PP = 3, 5, 16, 384, 64
    T = C * n
    A, W = _bf(B * T, PP, seed=1), _bf(D, PP, scale=0.1, seed=2)
    bias, E, pos = _f(D, seed=3), _f(C, D, seed=4), _f(n + 1, D, seed=5)
    x = torch.full((B, T + 1, D), 7.0, device="cuda")
    Y = torch.empty(B * T, D, device="cuda")
    hip.gemm_nt(A, W, hip.EPI_PATCH, x, bias=bias, out2=Y, aux=E, aux2=pos, T=T, n=n, ldo=D)
    ref = (A.float() @ W.float().t() + bias).reshape(B, T, D)
    _close(Y.reshape(B, T, D), ref, 1e-5, 1e-4, "Y")
    tok = ref + E.repeat_interleave(n, 0)[None] + pos[1:].repeat(C, 1)[None]
    _close(x[:, 1:], tok, 1e-5, 1e-4, "tokens")
    assert (x[:, 0] == 7.0).all()


def _check_gemm_tn(hip, M, P, Q, **kw):
    Y, X = _bf(M, P, seed=1), _bf(M, Q, seed=2)
    dW = _f(P, Q, seed=3)
    db = _f(P, seed=4)
    dW0, db0 = dW.clone(), db.clone()
    hip.gemm_tn_acc(Y, X, dW, db, **kw)
    ref = Y.float().t() @ X.float()
    _close(dW, dW0 + ref, 1e-4, 3e-4 * math.sqrt(M), "dW")
    _close(db, db0 + Y.float().sum(0), 1e-4, 1e-4 * math.sqrt(M), "dbias")
    dW2 = torch.zeros(P, Q, device="cuda")
    hip.gemm_tn_acc(Y, X, dW2, None, **kw)
    _close(dW2, ref, 1e-4, 3e-4 * math.sqrt(M), "dW no bias")


@pytest.mark.parametrize("M,P,Q", [(1000, 384, 384), (4100, 1152, 384), (333, 384, 1536), (64, 128, 128), (5000, 192, 64), (700, 384, 256)])
def test_gemm_tn(hip, M, P, Q, reduction_mode):
    _check_gemm_tn(hip, M, P, Q)


def test_gemm_tn_random_shapes(hip, reduction_mode):
    """Seeded random weight-gradient problems on both kernels: any reduction length (one row .. several thousand: ragged last stage,
    fewer stages than splits), P / Q multiples of 8 on the 128 x 128 kernel (partial tiles), multiples of 384 / 128 on the wide one."""
    rng = np.random.RandomState(7)
    for _ in range(10):
        M = int(rng.choice([rng.randint(1, 100), rng.randint(100, 2000), rng.randint(2000, 20000)]))
        wide = bool(rng.randint(0, 2))
        P = 384 * int(rng.randint(1, 4)) if wide else 8 * int(rng.randint(1, 97))
        Q = 128 * int(rng.randint(1, 7)) if wide else 8 * int(rng.randint(1, 97))
        try:
            _check_gemm_tn(hip, M, P, Q, tile=hip.TILE_WIDE if wide else hip.TILE_NARROW)
        except AssertionError as e:
            raise AssertionError(f"M={M} P={P} Q={Q} wide={wide}: {e}") from e


_TN_TILES = {"narrow": 1, "wide": 2}  # hip.TILE_*


@pytest.mark.parametrize("tile", ["narrow", "wide"])
@pytest.mark.parametrize("P,Q", [(1152, 384), (384, 384), (1536, 384), (384, 1536), (384, 256)])
def test_gemm_tn_headline_rows(hip, P, Q, tile, reduction_mode):
    """The weight-gradient products of the headline step (reduction over 100 416 token rows split over one resident round of
    workgroups), both tile shapes, atomic and deterministic; the last shape is the tokeniser's (P^2 = 256 inputs)."""
    _check_gemm_tn(hip, HEADLINE_M, P, Q, tile=_TN_TILES[tile])


@pytest.mark.parametrize("tile", ["narrow", "wide"])
@pytest.mark.parametrize("M,P,Q", [(HEADLINE_M, 1152, 384), (HEADLINE_M, 384, 1536), (4099, 384, 384), (700, 384, 256), (33, 384, 128)])
def test_gemm_tn_deterministic(hip, M, P, Q, tile):
    """dcv_gemm_tn_acc_det (the reference runs with cudnn.deterministic, utils.py:394-401): the splits store their
    partial tiles to a workspace and a second launch adds them in a fixed order.  Same value as the reference product, bit-identical
    from run to run (the atomic form is not), and the workspace may hold anything beforehand."""
    t = _TN_TILES[tile]
    nws = hip.gemm_tn_det_ws_floats(M, P, Q, t)
    ws = torch.full((nws,), float("nan"), device="cuda")  # poisoned: every element the reducer reads must have been written
    _check_gemm_tn(hip, M, P, Q, tile=t, ws=ws)
    Y, X = _bf(M, P, seed=11), _bf(M, Q, seed=12)
    runs = []
    for rep in range(3):
        dW, db = torch.zeros(P, Q, device="cuda"), torch.zeros(P, device="cuda")
        ws.uniform_(-1e6, 1e6)
        hip.gemm_tn_acc(Y, X, dW, db, tile=t, ws=ws)
        runs.append((dW.clone(), db.clone()))
    assert all(torch.equal(runs[0][0], r[0]) and torch.equal(runs[0][1], r[1]) for r in runs[1:])
    with pytest.raises(RuntimeError):  # a workspace that is too small is refused
        hip.gemm_tn_acc(Y, X, dW, db, tile=t, ws=ws[: max(nws // 2, 4)])


@pytest.mark.parametrize("C,D", [(8, 384), (3, 384), (18, 384), (16, 768), (1, 192), (32, 512)])
def test_proxy_loss_kernel(hip, C, D):
    """dcv_proxy_loss: value and both gradients of cross_entropy(-cdist(s * normalize(e), s * normalize(p))^2, eye(C)) in one launch, against
    autograd on the reference's own formula (models/loss_fn.py:7-21) in float64 — also with a zero row (normalize clamps the norm at 1e-12)
    and with nearly parallel rows (large logits); sizes beyond the kernel's LDS budget are refused."""
    import torch.nn.functional as F
    scale = float(np.sqrt(1.0 / 0.07))
    for case in ("random", "zero row", "parallel"):
        e = _f(C, D, seed=3, scale=0.5)
        p = _f(C, D, seed=4, scale=0.125)
        if case == "zero row":
            e[0].zero_()
        if case == "parallel":
            p = (e * 0.3 + 1e-3 * _f(C, D, seed=5)).contiguous()
        loss, de, dp = torch.empty(1, device="cuda"), torch.empty(C, D, device="cuda"), torch.empty(C, D, device="cuda")
        hip.proxy_loss(e, p, scale, loss, de, dp)
        e64, p64 = e.double().requires_grad_(True), p.double().requires_grad_(True)
        a, b = scale * F.normalize(e64, p=2, dim=-1), scale * F.normalize(p64, p=2, dim=-1)
        ref = F.cross_entropy(-((a[:, None, :] - b[None, :, :]) ** 2).sum(-1), torch.eye(C, device="cuda", dtype=torch.float64))
        ref.backward()
        assert abs(loss.item() - ref.item()) <= 2e-6 * max(1.0, abs(ref.item())), (case, loss.item(), ref.item())
        for got, want, name in ((de, e64.grad, "d_emb"), (dp, p64.grad, "d_proxies")):
            if case == "zero row" and name == "d_emb":
                got, want = got[1:], want[1:]  # the clamped row's gradient is 1e12-scaled garbage in both; compared below by finiteness only
            if got.numel() == 0:
                continue
            err = (got.double() - want).abs().max().item()
            assert err <= 2e-5 * want.abs().max().item() + 1e-9, (case, name, err, want.abs().max().item())
        assert torch.isfinite(de).all() and torch.isfinite(dp).all()
    assert hip.proxy_loss_supported(8, 384) and not hip.proxy_loss_supported(64, 768) and not hip.proxy_loss_supported(33, 64)
    with pytest.raises(RuntimeError):
        big = torch.zeros(64, 768, device="cuda")
        hip.proxy_loss(big, big, scale, torch.empty(1, device="cuda"), torch.empty_like(big), torch.empty_like(big))


@pytest.mark.parametrize("M", [64 * 197 + 5, 31, 5000])
def test_gemm_tn_group(hip, M, reduction_mode):
    """dcv_gemm_tn_group: a block's four weight-gradient products (fc2, fc1, proj, qkv shapes) over the same token rows in ONE launch, both
    reduction modes, accumulating into non-zero dW / dbias (one of them without a bias): against the fp32 product, against four separate
    dcv_gemm_tn_acc launches, bit-identical from run to run in deterministic mode; ragged and shorter-than-a-stage reductions; shapes the
    grouped form does not take are reported, not computed."""
    D = 384
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    assert hip.gemm_tn_group_supported(shapes, M)
    ops = [(_bf(M, P, seed=20 + i, scale=0.3), _bf(M, Q, seed=30 + i)) for i, (P, Q) in enumerate(shapes)]

    def fresh():
        g = torch.Generator(device="cuda").manual_seed(5)
        return [(torch.randn(P, Q, device="cuda", generator=g), (torch.randn(P, device="cuda", generator=g) if i != 2 else None))
                for i, (P, Q) in enumerate(shapes)]

    init = fresh()
    runs = []
    for rep in range(3 if reduction_mode == "det" else 1):
        outs = fresh()
        hip.gemm_tn_acc_group([(Y, X, dW, db) for (Y, X), (dW, db) in zip(ops, outs)])
        runs.append(outs)
    sep = fresh()
    for (Y, X), (dW, db) in zip(ops, sep):
        hip.gemm_tn_acc(Y, X, dW, db)
    for i, ((Y, X), (dW, db), (dW0, db0), (dWs, dbs)) in enumerate(zip(ops, runs[0], init, sep)):
        ref = dW0.double() + Y.double().T @ X.double()
        assert ((dW.double() - ref).norm() / ref.norm()).item() < 2e-6, i
        assert ((dW - dWs).norm() / dWs.norm()).item() < 2e-6, i
        if db is None:
            continue
        rb = db0.double() + Y.double().sum(0)
        assert ((db.double() - rb).norm() / rb.norm()).item() < 2e-6, i
    if reduction_mode == "det":
        for other in runs[1:]:
            assert all(torch.equal(a[0], b[0]) and (a[1] is None or torch.equal(a[1], b[1])) for a, b in zip(runs[0], other))
    assert not hip.gemm_tn_group_supported([(D, 4 * D), (100, D)], M)          # P not a multiple of 384
    assert not hip.gemm_tn_group_supported([(4 * D, 4 * D)] * 8, M)             # 8 x 48 tiles: more than one resident round
    with pytest.raises(RuntimeError):
        hip.gemm_tn_acc_group([(_bf(M, 256, seed=1), _bf(M, 128, seed=2), torch.zeros(256, 128, device="cuda"), None)])


@pytest.mark.parametrize("M", [31, 32, 33, 1000, 4099])
def test_gemm_tn_wide_ragged_reduction(hip, M):
    """384 x 128 kernel with reduction lengths around its 32-row stage (ragged last stage, fewer stages than the ring is deep)."""
    _check_gemm_tn(hip, M, 384, 128, tile=hip.TILE_WIDE)


def test_drop_path_kernel_hooks(hip):
    """The two places DropPath touches the kernels: dcv_gemm_nt's residual epilogue with a per-sample branch factor (aux2, T rows per
    sample) on both tile shapes, and dcv_ln_bwd_scaled's per-sample factor on the bf16 copy only (dx_out, dgamma, dbeta unchanged)."""
    B, N, D, K = 6, 700, 384, 1536
    M = B * N
    A, W, bias = _bf(M, K, seed=1), _bf(D, K, scale=0.05, seed=2), _f(D, scale=0.1, seed=3)
    x0 = _f(M, D, seed=4)
    sc = torch.tensor([0.0, 1.25, 1.25, 0.0, 1.25, 1.25], device="cuda")
    ref = x0 + sc.repeat_interleave(N)[:, None] * (A.float() @ W.float().t() + bias)
    for tile in (hip.TILE_NARROW, hip.TILE_WIDE):
        y = torch.empty_like(x0)
        hip.gemm_nt(A, W, hip.EPI_BIAS_RESID_F32, y, bias=bias, aux=x0, aux2=sc, T=N, tile=tile)
        _close(y, ref, 1e-4, 2e-4 * math.sqrt(K), f"resid with branch factor, tile {tile}")
        assert torch.equal(y.view(B, N, D)[0], x0.view(B, N, D)[0])  # a dropped sample keeps its residual bit for bit
    with pytest.raises(RuntimeError):
        hip.gemm_nt(A, W, hip.EPI_BIAS_RESID_F32, y, bias=bias, aux=x0, aux2=sc, T=0)
    # LayerNorm backward
    x, du, dx_in = _f(M, D, scale=2.0, seed=5), _bf(M, D, seed=6), _f(M, D, seed=7)
    g, b = 1 + 0.1 * _f(D, seed=8), 0.1 * _f(D, seed=9)
    u = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    hip.ln_fwd(x, g, b, u, mean, rstd, M, D, 1e-6)
    outs = []
    for scale in (None, sc):
        dx, dxb = torch.empty(M, D, device="cuda"), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
        hip.ln_bwd(du, x, mean, rstd, g, dx_in, dx, dxb, dg, db, M, D, **(dict(bf16_row_scale=scale, rows_per_sample=N) if scale is not None else {}))
        outs.append((dx, dxb, dg, db))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][2], outs[1][2]) and torch.equal(outs[0][3], outs[1][3])
    want = (outs[0][0].view(B, N, D) * sc.view(B, 1, 1)).to(torch.bfloat16).view(M, D)
    assert torch.equal(outs[1][1], want)


@pytest.mark.parametrize("M,D", [(1000, 384), (37, 192), (513, 768)])
def test_layernorm(hip, M, D, reduction_mode):
    x = _f(M, D, scale=2.0, seed=1) + 0.5
    g, b = 1 + 0.1 * _f(D, seed=2), 0.1 * _f(D, seed=3)
    u = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
    hip.ln_fwd(x, g, b, u, mean, rstd, M, D, 1e-6)
    xr = x.clone().requires_grad_(True)
    gr, br = g.clone().requires_grad_(True), b.clone().requires_grad_(True)
    ref = torch.nn.functional.layer_norm(xr, (D,), gr, br, 1e-6)
    _close(u, ref, 1e-2, 1e-2, "ln fwd")
    _close(mean, x.mean(-1), 1e-5, 1e-5, "mean")
    uf = torch.empty(M, D, device="cuda")
    hip.ln_fwd(x, g, b, uf, None, None, M, D, 1e-6)
    _close(uf, ref, 1e-5, 1e-5, "ln fwd f32")
    du = _bf(M, D, seed=4)
    dx_in = _f(M, D, seed=5)
    ref.backward(du.float())
    dx = torch.empty(M, D, device="cuda")
    dxb = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
    dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
    hip.ln_bwd(du, x, mean, rstd, g, dx_in, dx, dxb, dg, db, M, D)
    _close(dx, dx_in + xr.grad, 1e-4, 1e-4, "ln dx")
    _close(dxb, dx_in + xr.grad, 1e-2, 1e-2, "ln dx bf16")
    _close(dg, gr.grad, 1e-4, 1e-3, "dgamma")
    _close(db, br.grad, 1e-4, 1e-3, "dbeta")
    # in-place accumulate form (dx_in aliases dx_out) and f32 du
    dx2 = dx_in.clone()
    dg.zero_(); db.zero_()
    hip.ln_bwd(du.float(), x, mean, rstd, g, dx2, dx2, None, dg, db, M, D)
    _close(dx2, dx_in + xr.grad, 1e-4, 1e-4, "ln dx inplace")


def _attn_ref(qkv, B, N, H, scale):
    q, k, v = qkv.float().reshape(B, N, 3, H, 64).permute(2, 0, 3, 1, 4)
    s = (q @ k.transpose(-1, -2)) * scale
    p = s.softmax(-1)
    o = (p @ v).transpose(1, 2).reshape(B, N, H * 64)
    return o, torch.logsumexp(s, -1)


@pytest.mark.parametrize("B,N,H", [(2, 197, 3), (1, 64, 1), (2, 289, 6), (1, 1569, 6), (3, 130, 2), (8, 50, 1),
                                   (1, 17, 2), (2, 33, 1), (1, 128, 3), (1, 129, 1), (1, 192, 2), (2, 257, 1)])
def test_attention_fwd_bwd(hip, B, N, H):
    D = H * 64
    scale = 64 ** -0.5
    qkv = _bf(B, N, 3 * D, scale=1.5, seed=N)
    # spike one key against one query so a late tile raises the running max (online-softmax rescale path)
    qkv[0, N // 2, :64] *= 4
    qkv[0, N - 1, D:D + 64] = qkv[0, N // 2, :64]
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    hip.attn_fwd(qkv, o, lse, B, N, H, 64, scale)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, N, H, scale)
    _close(o, o_ref, 2e-2, 2e-2, "attn O")
    _close(lse, lse_ref, 1e-4, 2e-3, "attn LSE")
    dO = _bf(B, N, D, seed=7)
    o_ref.backward(dO.float())
    dqkv = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(2, B, H, N, device="cuda")
    hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, scale)
    assert torch.isfinite(dqkv.float()).all()
    g = qr.grad.reshape(B, N, 3, D)
    d = dqkv.float().reshape(B, N, 3, D)
    for i, nm in enumerate(["dQ", "dK", "dV"]):
        ref = g[:, :, i]
        tol = 3e-2 * ref.abs().max().item()
        _close(d[:, :, i], ref, 3e-2, tol, nm)


def test_attention_at_the_bench_grid(hip):
    """the three attention kernels at the bench's OWN launch — B = 64, H = 6, N = 1569: 384 (batch, head) pairs,
    4 992 workgroups (the other tests stop at B x H = 96 at this N) — against an fp32 reference evaluated in chunks of 4 images (the [4, 6,
    1569, 1569] fp32 score block is 236 MB).  Inputs are scaled like the model's (|q k| scores of a few units), with one spiked key per chunk
    so that the online-softmax rescale path runs at this grid too."""
    B, N, H = 64, 1569, 6
    D = H * 64
    scale = 64 ** -0.5
    qkv = _bf(B, N, 3 * D, scale=1.2, seed=2024)
    for b in range(0, B, 4):
        qkv[b, N // 3, :64] *= 4
        qkv[b, N - 5, D:D + 64] = qkv[b, N // 3, :64]
    dO = _bf(B, N, D, seed=2025)
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    hip.attn_fwd(qkv, o, lse, B, N, H, 64, scale)
    dqkv = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(2, B, H, N, device="cuda")
    hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, scale)
    assert torch.isfinite(dqkv.float()).all()
    worst = {}
    for b0 in range(0, B, 4):
        sl = slice(b0, b0 + 4)
        qr = qkv[sl].float().requires_grad_(True)
        o_ref, lse_ref = _attn_ref(qr, 4, N, H, scale)
        o_ref.backward(dO[sl].float())
        _close(o[sl], o_ref, 2e-2, 2e-2, f"attn O, images {b0}..")
        _close(lse[sl], lse_ref, 1e-4, 2e-3, f"attn LSE, images {b0}..")
        g = qr.grad.reshape(4, N, 3, D)
        d = dqkv[sl].float().reshape(4, N, 3, D)
        for i, nm in enumerate(["dQ", "dK", "dV"]):
            ref = g[:, :, i]
            _close(d[:, :, i], ref, 3e-2, 3e-2 * ref.abs().max().item(), f"{nm}, images {b0}..")
            rel = ((d[:, :, i] - ref).norm() / ref.norm()).item()
            worst[nm] = max(worst.get(nm, 0.0), rel)
        del qr, o_ref, lse_ref, g
    print("attention at B64 H6 N1569: worst relative L2 error per 4-image chunk", {k: f"{v:.2e}" for k, v in worst.items()})
    assert all(v <= 1e-2 for v in worst.values())
    # the pre-scaled-q entries at the same grid — the backward's dK / dV there is the persistent third form (csrc/attn_bwd3.hip: 2 304 items of 256 keys on
    # 256 workgroups, nine each, the 33-key remainder through the second form): against the plain pair's result, which was just checked
    c = scale * math.log2(math.e)
    qs = qkv.clone()
    qs[:, :, :D] = (qkv[:, :, :D].float() * c).to(torch.bfloat16)
    o_ps = torch.empty_like(o)
    lse_ps = torch.empty_like(lse)
    hip.attn_fwd(qs, o_ps, lse_ps, B, N, H, 64, scale, prescaled=True)
    outs = []
    for _ in range(2):
        dps = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
        hip.attn_bwd(qs, o_ps, dO, lse_ps, torch.empty(2, B, H, N, device="cuda"), dps, B, N, H, 64, scale, prescaled=True)
        outs.append(dps)
    assert torch.isfinite(outs[0].float()).all()
    assert torch.equal(outs[0].view(torch.int16), outs[1].view(torch.int16)), "pre-scaled attention backward is not bit-reproducible at the bench grid"
    a3, b3 = outs[0].float().reshape(B, N, 3, D), dqkv.float().reshape(B, N, 3, D)
    relf = {nm: ((a3[:, :, i] - b3[:, :, i]).norm() / b3[:, :, i].norm()).item() for i, nm in enumerate(["dQ", "dK", "dV"])}
    print("pre-scaled q vs plain at the bench grid, relative L2:", {k: f"{v:.2e}" for k, v in relf.items()})
    assert all(v <= 1.2e-2 for v in relf.values())  # the two q operands differ by one bf16 rounding of q * scale * log2(e)


def test_im2col(hip):
    from oracle import dichavit_oracle as orc
    B, Ct, H, P = 3, 6, 32, 8
    x = _f(B, Ct, H, H, seed=1)
    idx = torch.tensor([4, 0, 5], dtype=torch.int32, device="cuda")
    n = (H // P) ** 2
    out = torch.empty(B * 3 * n, P * P, dtype=torch.bfloat16, device="cuda")
    hip.im2col(x, idx, out, B, Ct, 3, H, H, P)
    ref = orc.unfold_patches(x[:, idx.long()].cpu(), P).reshape(B * 3 * n, P * P)
    assert torch.equal(out.cpu(), ref.to(torch.bfloat16))
    B, Ct, H, P = 2, 8, 224, 16
    x = _f(B, Ct, H, H, seed=2)
    idx = torch.arange(8, dtype=torch.int32, device="cuda")
    out = torch.empty(B * 8 * 196, 256, dtype=torch.bfloat16, device="cuda")
    hip.im2col(x, idx, out, B, Ct, 8, H, H, P)
    assert torch.equal(out.cpu(), orc.unfold_patches(x.cpu(), P).reshape(-1, 256).to(torch.bfloat16))
    # raw uint8 pixels with the per-channel affine fused (gathered channel order)
    B, Ct, H, P = 2, 5, 32, 8
    raw = torch.randint(0, 256, (B, Ct, H, H), dtype=torch.uint8, device="cuda")
    idx = torch.tensor([3, 1], dtype=torch.int32, device="cuda")
    scale, shift = torch.tensor([0.02, 0.05], device="cuda"), torch.tensor([-1.5, 0.25], device="cuda")
    out = torch.empty(B * 2 * 16, 64, dtype=torch.bfloat16, device="cuda")
    hip.im2col(raw, idx, out, B, Ct, 2, H, H, P, scale=scale, shift=shift)
    xn = raw[:, idx.long()].float() * scale[None, :, None, None] + shift[None, :, None, None]
    ref = orc.unfold_patches(xn.cpu(), P).reshape(-1, 64)
    assert (out.cpu().float() - ref).abs().max().item() <= 1e-2 * ref.abs().max().item()


@pytest.mark.parametrize("B,C,n,D", [(3, 5, 16, 384), (2, 1, 9, 192), (4, 8, 49, 768)])
def test_patch_bwd(hip, B, C, n, D, reduction_mode):
    T = C * n
    dx0, dYl = _f(B, T + 1, D, seed=1), _f(B, T, D, seed=2)
    dYb = torch.empty(B * T, D, dtype=torch.bfloat16, device="cuda")
    dE, dpos, dcls = torch.zeros(C, D, device="cuda"), torch.zeros(n + 1, D, device="cuda"), torch.zeros(D, device="cuda")
    hip.patch_bwd(dx0, dYl, dYb, dE, dpos, dcls, B, C, n, D)
    tok = dx0[:, 1:]
    _close(dYb.reshape(B, T, D), tok + dYl, 1e-2, 1e-2, "dY")
    _close(dE, tok.reshape(B, C, n, D).sum((0, 2)), 1e-5, 1e-4, "dE")
    _close(dpos[1:], tok.reshape(B, C, n, D).sum((0, 1)), 1e-5, 1e-4, "dpos")
    _close(dpos[0], dx0[:, 0].sum(0), 1e-5, 1e-4, "dpos0")
    _close(dcls, dx0[:, 0].sum(0), 1e-5, 1e-4, "dcls")
    hip.patch_bwd(dx0, None, dYb, dE, dpos, dcls, B, C, n, D)
    _close(dYb.reshape(B, T, D), tok, 1e-2, 1e-2, "dY no loss")


def test_deterministic_small_reductions_repeat_bitwise(hip):
    """LayerNorm's dgamma / dbeta, the tokeniser's d(channel_embed) / d(pos) / d(cls), the diversity statistics and the gradient norm in
    deterministic mode: three runs on the same inputs are bit-identical (workspace scrambled in between)."""
    old = hip.set_deterministic(True)
    try:
        M, D = 9001, 384
        x, du, dx_in = _f(M, D, seed=1), _bf(M, D, seed=2), _f(M, D, seed=3)
        g, b = 1 + 0.1 * _f(D, seed=4), 0.1 * _f(D, seed=5)
        u = torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
        mean, rstd = torch.empty(M, device="cuda"), torch.empty(M, device="cuda")
        hip.ln_fwd(x, g, b, u, mean, rstd, M, D, 1e-6)
        B, C, n = 16, 8, 196
        dx0 = _f(B, C * n + 1, D, seed=6)
        Y = _f(B * C * n, D, seed=7)
        runs = []
        for rep in range(3):
            for ws in hip._det_ws.values():
                ws.uniform_(-1e6, 1e6)
            dx, dxb = torch.empty(M, D, device="cuda"), torch.empty(M, D, dtype=torch.bfloat16, device="cuda")
            dg, db = torch.zeros(D, device="cuda"), torch.zeros(D, device="cuda")
            hip.ln_bwd(du, x, mean, rstd, g, dx_in, dx, dxb, dg, db, M, D)
            dYb = torch.empty(B * C * n, D, dtype=torch.bfloat16, device="cuda")
            dE, dpos, dcls = torch.zeros(C, D, device="cuda"), torch.zeros(n + 1, D, device="cuda"), torch.zeros(D, device="cuda")
            hip.patch_bwd(dx0, None, dYb, dE, dpos, dcls, B, C, n, D)
            S, ssq = torch.empty(B, C, D, device="cuda"), torch.empty(B, C, device="cuda")
            tot, inv, stats = torch.empty(B, D, device="cuda"), torch.empty(B, C * n, device="cuda"), torch.empty(B, 2, device="cuda")
            hip.ortho_fwd(Y, S, ssq, tot, inv, stats, B, C, n, D)
            acc = torch.zeros(1, device="cuda")
            hip.sumsq_acc(x, x.numel(), acc)
            runs.append([t.clone() for t in (dg, db, dx, dE, dpos, dcls, S, ssq, stats, acc)])
        for r in runs[1:]:
            for a, c in zip(runs[0], r):
                assert torch.equal(a, c)
        assert abs(runs[0][-1].item() - x.double().pow(2).sum().item()) <= 1e-5 * x.numel()
    finally:
        hip.set_deterministic(old)


@pytest.mark.parametrize("B,C,n,D", [(3, 5, 16, 384), (2, 1, 9, 192), (2, 8, 196, 384), (2, 18, 16, 384)])
def test_ortho_loss(hip, B, C, n, D, reduction_mode):
    from oracle import dichavit_oracle as orc
    T = C * n
    Y = _f(B, T, D, seed=3) + 0.3
    S, selfsq = torch.empty(B, C, D, device="cuda"), torch.empty(B, C, device="cuda")
    tot, inv, stats = torch.empty(B, D, device="cuda"), torch.empty(B, T, device="cuda"), torch.empty(B, 2, device="cuda")
    hip.ortho_fwd(Y, S, selfsq, tot, inv, stats, B, C, n, D)
    Yr = Y.double().cpu().requires_grad_(True)
    f = torch.nn.functional.normalize(Yr, dim=-1).reshape(B, C, n, D)
    s = f.sum(2)
    pos_sum = ((s * s).sum(-1) - (f * f).sum(-1).sum(-1)).sum(-1)
    neg_sum = (s.sum(1) ** 2).sum(-1) - (s * s).sum(-1).sum(-1)
    _close(stats[:, 0].cpu(), pos_sum.detach(), 1e-4, 1e-3 * max(1.0, pos_sum.abs().max().item()), "pos_sum")
    if C > 1:
        _close(stats[:, 1].cpu(), neg_sum.detach(), 1e-4, 1e-3 * max(1.0, neg_sum.abs().max().item()), "neg_sum")
    else:
        assert (stats[:, 1] == 0).all()
    coef = _f(B, 2, seed=9)
    loss = (coef[:, 0].double().cpu() * pos_sum).sum() + ((coef[:, 1].double().cpu() * neg_sum).sum() if C > 1 else 0)
    loss.backward()
    dY = torch.empty_like(Y)
    hip.ortho_bwd(Y, S, tot, inv, coef, dY, B, C, n, D)
    _close(dY.cpu(), Yr.grad, 1e-3, 1e-4 * Yr.grad.abs().max().item() + 1e-6, "dY")
    # value parity with the oracle's full loss formula
    val = orc.ortho_loss_linear(Y.double().cpu(), C, n, 1.0, 4.0, True, False).item()
    pos = stats[:, 0].double().cpu() / orc._count_eps(C * n * (n - 1))
    neg = stats[:, 1].double().cpu() / orc._count_eps(T * T - C * n * n) if C > 1 else torch.zeros(B, dtype=torch.float64)
    assert abs((pos + 4.0 * neg).mean().item() - val) < 1e-5 * max(1.0, abs(val))


def test_adamw_and_casts(hip):
    from oracle import dichavit_oracle as orc
    n = 100003
    p, g = _f(n, seed=1), _f(n, scale=0.01, seed=2)
    m, v = torch.zeros(n + 1, device="cuda")[:n], torch.zeros(n + 1, device="cuda")[:n]
    pc, gc, mc, vc = p.cpu().clone(), g.cpu().clone(), torch.zeros(n), torch.zeros(n)
    for step in (1, 2, 3):
        hip.adamw(p, g, m, v, n, 4.9e-5, 0.9, 0.999, 1e-8, 0.04, step, 0.5)
        orc.adamw_step(pc, gc * 0.5, mc, vc, step, 4.9e-5, 0.9, 0.999, 1e-8, 0.04)
    _close(p.cpu(), pc, 1e-6, 1e-7, "adamw p")
    _close(v.cpu(), vc, 1e-4, 1e-12, "adamw v")
    dst = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    hip.cast_bf16(p, dst, n)
    assert torch.equal(dst, p.to(torch.bfloat16))
    # batched transpose
    src = _f(1152 * 384 + 100 * 70, seed=5)
    out = torch.zeros(src.numel(), dtype=torch.bfloat16, device="cuda")
    desc = torch.tensor([[0, 0, 1152, 384], [1152 * 384, 1152 * 384, 100, 70]], dtype=torch.int64, device="cuda")
    hip.cast_transpose_bf16(src, out, desc, 2, 18 * 6)
    assert torch.equal(out[:1152 * 384].reshape(384, 1152), src[:1152 * 384].reshape(1152, 384).t().to(torch.bfloat16))
    assert torch.equal(out[1152 * 384:].reshape(70, 100), src[1152 * 384:].reshape(100, 70).t().to(torch.bfloat16))


def test_attention_long_sequence_base_heads(hip):
    """BASELINE config 5 shape class: 64 channels x 196 patches + CLS = 12 545 tokens, 12 heads (two of them here to
    keep the N x N fp32 reference at 1.3 GB).  Forward and backward against the materialised softmax."""
    B, N, H = 1, 12545, 2
    D = H * 64
    scale = 64 ** -0.5
    qkv = _bf(B, N, 3 * D, scale=1.0, seed=3)
    o = torch.empty(B, N, D, dtype=torch.bfloat16, device="cuda")
    lse = torch.empty(B, H, N, device="cuda")
    hip.attn_fwd(qkv, o, lse, B, N, H, 64, scale)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, N, H, scale)
    _close(o, o_ref, 2e-2, 5e-3, "attn O long")
    _close(lse, lse_ref, 1e-4, 2e-3, "attn LSE long")
    dO = _bf(B, N, D, seed=7)
    o_ref.backward(dO.float())
    dqkv = torch.empty(B, N, 3 * D, dtype=torch.bfloat16, device="cuda")
    delta = torch.empty(2, B, H, N, device="cuda")
    hip.attn_bwd(qkv, o, dO, lse, delta, dqkv, B, N, H, 64, scale)
    g = qr.grad.reshape(B, N, 3, D)
    d = dqkv.float().reshape(B, N, 3, D)
    for i, nm in enumerate(["dQ", "dK", "dV"]):
        ref = g[:, :, i]
        rel = (d[:, :, i] - ref).norm().item() / ref.norm().item()
        assert rel <= 2e-2, (nm, rel)
    # the pre-scaled-q pair at this length (dK / dV: the persistent form, 2 x 50 items of 256 keys on 100 workgroups; 197 query tiles per item)
    c = scale * math.log2(math.e)
    qs = qkv.clone()
    qs[:, :, :D] = (qkv[:, :, :D].float() * c).to(torch.bfloat16)
    hip.attn_fwd(qs, o, lse, B, N, H, 64, scale, prescaled=True)
    dps = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    hip.attn_bwd(qs, o, dO, lse, delta, dps, B, N, H, 64, scale, prescaled=True)
    df = dps.float().reshape(B, N, 3, D)
    for i, nm in enumerate(["dQ", "dK", "dV"]):
        ref = g[:, :, i]
        rel = (df[:, :, i] - ref).norm().item() / ref.norm().item()
        assert rel <= 2e-2, (nm + " pre-scaled q", rel)


def _lowbias32(h):
    h = h.astype(np.uint64)
    h ^= h >> 16; h = (h * 0x7FEB352D) & 0xFFFFFFFF; h ^= h >> 15; h = (h * 0x846CA68B) & 0xFFFFFFFF; h ^= h >> 16
    return h


def _sr_bf16_numpy(x, base_index, seed):
    """The stochastic rounding the header specifies (include/dcv.h: dcv_cast_bf16_sr), restated with integers."""
    bits = x.view(np.uint32).astype(np.uint64)
    idx = base_index.astype(np.uint64) & 0xFFFFFFFF
    r = _lowbias32(idx ^ ((seed * 0x9E3779B9) & 0xFFFFFFFF)) & 0xFFFF
    return (((bits + r) & 0xFFFFFFFF) >> 16).astype(np.uint16)


def test_stochastic_cast_bit_exact_and_unbiased(gpu_device):
    """Both operand copies (straight, transposed) round every weight identically and exactly as specified; the mean over
    seeds converges to the fp32 value (unbiased), unlike round-to-nearest."""
    from diverse_channel_vit_amd import hip
    rs = np.random.RandomState(5)
    R, Cc, pad = 192, 136, 40
    n = pad + R * Cc + 7
    src = (rs.standard_normal(n) * 0.03).astype(np.float32)
    s = torch.from_numpy(src).to(gpu_device)
    desc = torch.tensor([[pad, pad, R, Cc]], dtype=torch.int64, device=gpu_device)
    acc = np.zeros(n, np.float64)
    seeds = list(range(1, 65))
    for seed in seeds:
        sd = torch.tensor([seed], dtype=torch.int32, device=gpu_device)
        d = torch.zeros(n, dtype=torch.bfloat16, device=gpu_device)
        dt = torch.zeros(n, dtype=torch.bfloat16, device=gpu_device)
        hip.cast_bf16_sr(s, d, n, sd)
        hip.cast_transpose_bf16_sr(s, dt, desc, 1, ((R + 63) // 64) * ((Cc + 63) // 64), sd)
        got = d.view(torch.int16).cpu().numpy().view(np.uint16)
        want = _sr_bf16_numpy(src, np.arange(n), seed)
        assert np.array_equal(got, want)
        gt = dt.view(torch.int16).cpu().numpy().view(np.uint16)[pad:pad + R * Cc].reshape(Cc, R)
        assert np.array_equal(gt.T, want[pad:pad + R * Cc].reshape(R, Cc))
        acc += (want.astype(np.uint32) << 16).view(np.float32)
    mean_err = np.abs(acc / len(seeds) - src)
    rtn_err = np.abs(s.to(torch.bfloat16).float().cpu().numpy() - src)
    # standard error of the SR mean ~ ulp/(2*sqrt(3*64)) ~ ulp/28; round-to-nearest's error is ~ulp/4 on average and does not shrink
    assert mean_err.mean() < 0.35 * rtn_err.mean()


@pytest.mark.parametrize("B,N,H,Nq,shift", [(2, 197, 3, None, 0.0), (1, 1569, 6, None, 0.0), (2, 289, 6, None, -40.0), (3, 130, 2, 5, 0.0),
                                            (1, 1569, 6, 1, 0.0), (1, 64, 1, None, 25.0), (2, 257, 1, None, 0.0), (1, 17, 2, None, -3.0)])
def test_attention_prescaled_q(hip, B, N, H, Nq, shift):
    """dcv_attn_*_ps: the q part of qkv holds q * scale * log2(e); o, lse and dqkv (dQ with respect to the UNSCALED q) must equal the plain
    definition.  `shift` adds a constant to every score of a query (through one k-aligned q component): -40 makes the first tile's maximum
    strongly negative (the lazy reference maximum must start AT it, not at 0), +25 makes every score large; the spike below raises the maximum in
    a late tile by more than the rescale threshold."""
    D = H * 64
    scale = 64 ** -0.5
    c = scale * math.log2(math.e)
    qkv = _bf(B, N, 3 * D, scale=1.5, seed=N + 3)
    qkv[0, N // 2, :64] *= 4
    qkv[0, N - 1, D:D + 64] = qkv[0, N // 2, :64]
    if shift:
        qkv[:, :, D + 63] = 1.0  # every key has 1 in its last component (head 0) ...
        qkv[:, :, 63] = shift / scale  # ... so head 0's scores all move by `shift`
    qs = qkv.clone()
    qs[:, :, :D] = (qkv[:, :, :D].float() * c).to(torch.bfloat16)
    qref = qs.float()
    qref[:, :, :D] /= c  # what the kernels see, unscaled
    qr = qref.clone().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, N, H, scale)
    nq = N if Nq is None else Nq
    o = torch.full((B, N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B, H, N), float("nan"), device="cuda")
    hip.attn_fwd(qs, o, lse, B, N, H, 64, scale, nq=Nq, prescaled=True)
    _close(o[:, :nq], o_ref[:, :nq], 2e-2, 2e-2, "ps attn O")
    _close(lse[:, :, :nq], lse_ref[:, :, :nq], 1e-4, 3e-3, "ps attn LSE")
    dO = torch.full((B, N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    dO[:, :nq] = _bf(B, nq, D, seed=11)
    (o_ref[:, :nq] * dO[:, :nq].float()).sum().backward()
    dqkv = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(2, B, H, N, device="cuda")
    hip.attn_bwd(qs, o, dO, lse, ws, dqkv, B, N, H, 64, scale, nq=Nq, prescaled=True)
    assert torch.isfinite(dqkv.float()).all()
    g = qr.grad.reshape(B, N, 3, D)
    d = dqkv.float().reshape(B, N, 3, D)
    for i, nm in enumerate(["dQ", "dK", "dV"]):
        ref = g[:, :, i]
        _close(d[:, :, i], ref, 3e-2, 3e-2 * ref.abs().max().item(), "ps " + nm)


def test_attention_prescaled_q_first_tile_far_below_the_maximum(hip):
    """in the pre-scaled-q forward the reference maximum of tile 0 is the tile's own maximum whatever its sign; when EVERY score of the first
    64 keys lies below -127 in log2 units (natural logit < -88) the rescale factor exp2(-d) of that tile is +inf while l and O are still 0, and
    0 * inf = NaN reached l, O and LSE.  Here the first 64 keys score -128 (natural) against every query of head 0 and later tiles hold the maximum:
    the pre-scaled entry must equal the plain one and the fp32 definition (and stay finite)."""
    B, N, H = 2, 200, 2
    D = H * 64
    scale = 64 ** -0.5
    c = scale * math.log2(math.e)
    qkv = _bf(B, N, 3 * D, scale=0.5, seed=91)
    qkv[:, :, :64] = 1.0          # head 0: q = 1 in every component ...
    qkv[:, :64, D:D + 64] = -16.0  # ... keys 0..63: score 64 * -16 / 8 = -128
    qs = qkv.clone()
    qs[:, :, :D] = (qkv[:, :, :D].float() * c).to(torch.bfloat16)
    qref = qs.float()
    qref[:, :, :D] /= c
    o_ref, lse_ref = _attn_ref(qref, B, N, H, scale)
    o = torch.full((B, N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B, H, N), float("nan"), device="cuda")
    hip.attn_fwd(qs, o, lse, B, N, H, 64, scale, prescaled=True)
    assert torch.isfinite(o.float()).all() and torch.isfinite(lse).all()
    _close(o, o_ref, 2e-2, 2e-2, "ps attn O, deep first tile")
    _close(lse, lse_ref, 1e-4, 3e-3, "ps attn LSE, deep first tile")
    o2 = torch.empty_like(o)
    lse2 = torch.empty_like(lse)
    hip.attn_fwd(qkv, o2, lse2, B, N, H, 64, scale)  # the plain entry on the unscaled operand
    _close(o, o2, 2e-2, 2e-2, "ps vs plain O")
    _close(lse, lse2, 1e-4, 5e-2, "ps vs plain LSE")  # the two operands differ by the bf16 rounding of q * c


def test_cast_scaled_ranges(hip):
    """dcv_cast_scaled_ranges rewrites ranges of an operand copy as the cast of scale * src: round-to-nearest, and stochastic with the SAME
    per-element bits as dcv_cast_bf16_sr (so the rewritten range equals that cast applied to a scaled source); kind 1 writes the fp32 product."""
    n = 4096 + 36
    src = torch.randn(n, device="cuda") * 0.05
    desc = torch.tensor([[128, 128, 1000, 1000, 0], [2048, 0, 300, 100, 1], [3000, 3000, 1132, 600, 0]], dtype=torch.int64, device="cuda")
    sc = 0.18033688
    scaled = src.clone()
    scaled[128:1128] *= sc
    scaled[3000:3600] *= sc
    for seed in (None, 5):
        sd = None if seed is None else torch.full((1,), seed, dtype=torch.int32, device="cuda")
        want = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        got = torch.empty(n, dtype=torch.bfloat16, device="cuda")
        f32 = torch.full((300,), float("nan"), device="cuda")
        if sd is None:
            hip.cast_bf16(scaled, want, n)
            hip.cast_bf16(src, got, n)
        else:
            hip.cast_bf16_sr(scaled, want, n, sd)
            hip.cast_bf16_sr(src, got, n, sd)
        hip.cast_scaled_ranges(src, got, f32, desc, 3, 4, sc, sd)
        assert torch.equal(got.view(torch.int16), want.view(torch.int16)), seed
        ref = src[2048:2348].clone()
        ref[:100] *= sc
        assert torch.equal(f32, ref)


@pytest.mark.parametrize("B,N,H,Nq", [(2, 197, 3, 1), (1, 1569, 6, 1), (3, 130, 2, 5), (2, 289, 1, 130), (1, 64, 2, 64)])
def test_attention_query_row_subset(hip, B, N, H, Nq):
    """dcv_attn_*_rows: only the query rows [0, Nq) are processed (the last block needs the CLS row only); keys / values are all N
    rows.  Forward rows < Nq and the backward of a loss that reads only those rows must equal the full computation."""
    D = H * 64
    scale = 64 ** -0.5
    qkv = _bf(B, N, 3 * D, scale=1.2, seed=N + Nq)
    o = torch.full((B, N, D), float("nan"), dtype=torch.bfloat16, device="cuda")
    lse = torch.full((B, H, N), float("nan"), device="cuda")
    hip.attn_fwd(qkv, o, lse, B, N, H, 64, scale, nq=Nq)
    qr = qkv.float().requires_grad_(True)
    o_ref, lse_ref = _attn_ref(qr, B, N, H, scale)
    _close(o[:, :Nq], o_ref[:, :Nq], 2e-2, 2e-2, "attn O rows")
    _close(lse[:, :, :Nq], lse_ref[:, :, :Nq], 1e-4, 2e-3, "attn LSE rows")
    dO = torch.full((B, N, D), float("nan"), dtype=torch.bfloat16, device="cuda")  # rows >= Nq must never be read
    dO[:, :Nq] = _bf(B, Nq, D, seed=9)
    (o_ref[:, :Nq] * dO[:, :Nq].float()).sum().backward()
    dqkv = torch.full((B, N, 3 * D), float("nan"), dtype=torch.bfloat16, device="cuda")
    ws = torch.empty(2, B, H, N, device="cuda")
    hip.attn_bwd(qkv, o, dO, lse, ws, dqkv, B, N, H, 64, scale, nq=Nq)
    assert torch.isfinite(dqkv.float()).all()
    g = qr.grad.reshape(B, N, 3, D)
    d = dqkv.float().reshape(B, N, 3, D)
    assert (d[:, Nq:, 0] == 0).all()
    for i, nm in enumerate(["dQ", "dK", "dV"]):
        ref = g[:, :, i]
        tol = 3e-2 * ref.abs().max().item()
        _close(d[:, :, i], ref, 3e-2, tol, nm)
