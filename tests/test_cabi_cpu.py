"""CPU-side checks of the drop-in boundary: the C-ABI library loads and exports every symbol that
include/dcv.h declares (no compute call is made without a GPU), and the product path refuses to run
on the CPU instead of falling back."""
import os
import re

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _declared():
    src = open(os.path.join(ROOT, "include", "dcv.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(dcv_[a-z0-9_]+)\s*\(", src)))


def test_library_exports_every_declared_symbol():
    from diverse_channel_vit_amd import _build, hip
    if not os.path.exists(hip.LIB_PATH):
        _build.build(verbose=False)
    lib = hip.load()
    names = _declared()
    assert len(names) >= 19
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/dcv.h but not exported"
    assert sorted(hip.EXPORTS) == names, "ctypes signature table and header disagree"
    assert lib.dcv_version() >= 100
    assert lib.dcv_error_string(-3).decode().startswith("unsupported")


def test_no_cpu_fallback_and_state_dict_layout():
    import diverse_channel_vit_amd as dcv
    from conftest import load_golden
    meta, _ = load_golden("so2sat_s")

    class Cfg(dict):
        __getattr__ = dict.get

    cfg = Cfg(meta["cfg"], in_channel_names=[f"c{i}" for i in range(18)], img_size=[32], num_classes=17)
    model = dcv.dichavit(cfg, mapper={"train": list(range(18))})
    assert sorted(model.state_dict().keys()) == sorted(meta["state_keys"])  # the reference's 156 keys
    assert sum(p.numel() for p in model.parameters()) == 21353105  # SURVEY §8a a1 (So2Sat S)
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        model(torch.zeros(2, 18, 32, 32), "train", None)
    for bad in (dict(block_type="block_v2"), dict(block_type="nope"), dict(pretrained_model_name="huge")):
        with pytest.raises(ValueError):
            dcv.dichavit(Cfg(cfg, **bad), mapper={"train": list(range(18))})


def test_product_never_imports_oracle():
    pkg = os.path.join(ROOT, "diverse_channel_vit_amd")
    for f in os.listdir(pkg):
        if f.endswith(".py"):
            txt = open(os.path.join(pkg, f)).read()
            assert "import oracle" not in txt and "from oracle" not in txt, f


def test_gemm_tile_choice_rules():
    """dcv_gemm_nt_pick / dcv_gemm_tn_pick are host logic (no GPU call): the automatic tile choice for the headline step's products
    (measured per shape in profiles/r02_x4_* and r02_x10_*) and the refusal of an illegal forced tile."""
    from diverse_channel_vit_amd import hip
    lib = hip.load()
    M = 64 * 1569
    A, N_, W = hip.TILE_AUTO, hip.TILE_NARROW, hip.TILE_WIDE
    nt = lambda m, n, k, epi, t=A: lib.dcv_gemm_nt_pick(m, n, k, epi, t)
    assert nt(M, 1152, 384, hip.EPI_BIAS_BF16) == W            # qkv
    assert nt(M, 1536, 384, hip.EPI_BIAS_GELU_BF16) == W       # fc1 + GELU
    assert nt(M, 1536, 384, hip.EPI_GELU_BWD_BF16) == N_       # GELU backward: 616 MB of epilogue traffic, narrow is faster
    assert nt(M, 384, 1536, hip.EPI_BIAS_RESID_F32) == W       # fc2 + residual
    assert nt(M, 384, 1536, hip.EPI_PLAIN_BF16) == W           # input gradient of fc1
    assert nt(M, 384, 1152, hip.EPI_PLAIN_BF16) == N_          # input gradient of qkv
    assert nt(M, 384, 384, hip.EPI_BIAS_RESID_F32) == N_       # proj
    assert nt(64, 1536, 384, hip.EPI_BIAS_GELU_BF16) == N_     # the CLS-only last block
    assert nt(M, 200, 384, hip.EPI_PLAIN_BF16) == N_           # N % 384 != 0
    assert nt(M, 384, 256, hip.EPI_PATCH) == N_                # the tokeniser epilogue lives on the narrow kernel only
    assert nt(M, 200, 384, hip.EPI_PLAIN_BF16, W) < 0 and nt(M, 384, 256, hip.EPI_PATCH, W) < 0 and nt(M, 384, 384, 0, 7) < 0
    assert nt(M, 1152, 384, hip.EPI_BIAS_BF16, N_) == N_ and nt(M, 384, 384, hip.EPI_PLAIN_BF16, W) == W
    # a few rounds of tiles: rounds x time per tile decides (256 CUs assumed when no device is present, as on the MI355X).
    # Measured on MI355X (tools/gemm_mid_m.py): M = 12 369 qkv 19.5 us narrow / 20.6 wide, fc1 37.7 / 32.1; M = 16 485 qkv 26.6 / 22.0,
    # fc1 43.8 / 52.3; M = 25 104 qkv 33.8 / 40.0, fc1T 45.3 / 48.0; M = 50 208 qkv 55.5 / 62.5, fc1T 69.7 / 53.1
    if lib.dcv_gemm_nt_pick(12369, 1536, 384, hip.EPI_BIAS_GELU_BF16, A) in (N_, W):
        assert nt(12369, 1152, 384, hip.EPI_BIAS_BF16) == N_ and nt(12369, 1536, 384, hip.EPI_BIAS_GELU_BF16) == W
        assert nt(16485, 1152, 384, hip.EPI_BIAS_BF16) == W and nt(16485, 1536, 384, hip.EPI_BIAS_GELU_BF16) == N_
        assert nt(25104, 1152, 384, hip.EPI_BIAS_BF16) == N_ and nt(25104, 384, 1536, hip.EPI_PLAIN_BF16) == N_
        assert nt(50208, 1152, 384, hip.EPI_BIAS_BF16) == N_ and nt(50208, 384, 1536, hip.EPI_PLAIN_BF16) == W
    tn = lambda p, q, t=A: lib.dcv_gemm_tn_pick(M, p, q, t)
    assert tn(384, 1536) == W and tn(1536, 384) == W and tn(1152, 384) == W
    assert tn(384, 384) == N_ and tn(384, 256) == N_ and tn(200, 128) == N_
    assert tn(200, 128, W) < 0 and tn(384, 384, W) == W and tn(384, 1536, N_) == N_


def test_deterministic_workspace_sizes_and_switch():
    """The *_det_ws_floats entries are host logic (no GPU call): sizes for the headline step (256 CUs assumed without a device), and the
    package-level switch that selects the deterministic forms (default on; the reference sets cudnn.deterministic = True)."""
    import diverse_channel_vit_amd as dcv
    from diverse_channel_vit_amd import hip
    lib = hip.load()
    M = 64 * 1569
    # gemm_tn384: 256 CUs / 12 tiles = 21 splits of [P*Q] partial tiles + 21 x 3 rows of bias partials
    assert lib.dcv_gemm_tn_det_ws_floats(M, 1536, 384, hip.TILE_AUTO) == 21 * 1536 * 384 + 21 * 3 * 1536
    assert lib.dcv_gemm_tn_det_ws_floats(M, 384, 1536, hip.TILE_AUTO) == 21 * 384 * 1536 + 21 * 12 * 384
    # 128 x 128 kernel: 512 slots / 9 tiles = 56 splits asked for, 55 with rows once a split is a multiple of the 64-row stage
    # (29 stages = 1856 rows each); the bias partial sits inside each split's slab
    assert lib.dcv_gemm_tn_det_ws_floats(M, 384, 384, hip.TILE_AUTO) == 55 * (384 * 384 + 384)
    assert lib.dcv_gemm_tn_det_ws_floats(10, 384, 384, hip.TILE_NARROW) == 1 * (384 * 384 + 384)  # fewer rows than a stage: one split
    assert lib.dcv_gemm_tn_det_ws_floats(M, 200, 128, hip.TILE_WIDE) < 0                              # illegal forced tile
    assert lib.dcv_ln_bwd_det_ws_floats(M, 384) == 1024 * 2 * 384 and lib.dcv_ln_bwd_det_ws_floats(10, 384) == 3 * 2 * 384
    assert lib.dcv_patch_bwd_det_ws_floats(64, 8, 196, 384) == 98 * 2 * 8 * 384 + 8 * 2 * 196 * 384
    assert lib.dcv_ortho_fwd_det_ws_floats(64, 8, 196, 384) == 7 * (64 * 8 * 384 + 64 * 8)
    assert lib.dcv_sumsq_det_ws_floats(21_600_000) == 1024 and lib.dcv_sumsq_det_ws_floats(100) == 1
    assert lib.dcv_ln_bwd_det_ws_floats(0, 384) < 0
    # grouped weight gradients of one block: 12 + 12 + 3 + 9 = 36 tiles -> 256 / 36 = 7 splits for every product
    D = 384
    shapes = [(D, 4 * D), (4 * D, D), (D, D), (3 * D, D)]
    arr = (hip._TnItem * 4)()
    for it, (P, Q) in zip(arr, shapes):
        it.Y = it.X = it.dW = 16
        it.ldy, it.ldx, it.P, it.Q, it.lddw = P, Q, P, Q, Q
    import ctypes as C
    want = sum(7 * P * Q + 7 * (Q // 128) * P for P, Q in shapes)
    assert lib.dcv_gemm_tn_group_ws_floats(C.cast(arr, C.c_void_p), 4, M) == want
    assert hip.gemm_tn_group_supported(shapes, M) and not hip.gemm_tn_group_supported([(100, 128)], M)
    assert not hip.gemm_tn_group_supported([(4 * D, 4 * D)] * 8, M) and not hip.gemm_tn_group_supported([], M)
    old = dcv.set_deterministic(False)
    try:
        assert not dcv.is_deterministic()
        assert dcv.set_deterministic(True) is False and dcv.is_deterministic()
    finally:
        dcv.set_deterministic(old)
