"""ctypes binding of libdcv_hip.so (include/dcv.h).  Thin: tensors in, raw device pointers +
sizes + the current torch stream out.  There is NO fallback: if the library is missing or a call
returns an error code, this raises.  PyTorch is used only for device memory and streams."""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("DCV_LIB", os.path.join(_HERE, "libdcv_hip.so"))  # DCV_LIB: A/B builds of the same ABI

EPI_BIAS_BF16, EPI_BIAS_GELU_BF16, EPI_BIAS_RESID_F32, EPI_PLAIN_BF16, EPI_GELU_BWD_BF16, EPI_PATCH = range(6)
TILE_AUTO, TILE_NARROW, TILE_WIDE, TILE_PAIR, TILE_ALT = range(5)  # include/dcv.h DCV_TILE_*

_vp, _i, _l, _f = C.c_void_p, C.c_int, C.c_long, C.c_float
_SIGS = {
    "dcv_version": ([], C.c_int),
    "dcv_error_string": ([_i], C.c_char_p),
    "dcv_gemm_nt": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _vp], _i),
    "dcv_gemm_nt_ex": ([_vp, _i, _vp, _i, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _i, _i, _i, _i, _vp], _i),
    "dcv_gemm_nt_pick": ([_i, _i, _i, _i, _i], _i),
    "dcv_gemm_nt_resid_ln": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _vp, _i, _vp, _i, _vp, _i, _vp, _vp, _f, _vp, _i, _vp, _vp, _i, _vp], _i),
    "dcv_gemm_tn_pick": ([_i, _i, _i, _i], _i),
    "dcv_gemm_tn_acc": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _vp], _i),
    "dcv_gemm_tn_acc_ex": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp], _i),
    "dcv_gemm_tn_det_ws_floats": ([_i, _i, _i, _i], _l),
    "dcv_gemm_tn_acc_det": ([_vp, _i, _vp, _i, _i, _i, _i, _vp, _i, _vp, _i, _vp, _l, _vp], _i),
    "dcv_gemm_tn_group_ws_floats": ([_vp, _i, _i], _l),
    "dcv_gemm_tn_group": ([_vp, _i, _i, _vp, _l, _vp], _i),
    "dcv_ln_fwd": ([_vp, _l, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _f, _vp], _i),
    "dcv_ln_bwd": ([_vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _vp], _i),
    "dcv_ln_bwd_det_ws_floats": ([_i, _i], _l),
    "dcv_ln_bwd_det": ([_vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _vp, _l, _vp], _i),
    "dcv_ln_bwd_scaled": ([_vp, _i, _vp, _l, _vp, _vp, _vp, _vp, _vp, _l, _vp, _vp, _vp, _i, _i, _vp, _i, _vp, _l, _vp], _i),
    "dcv_patch_bwd_det_ws_floats": ([_i, _i, _i, _i], _l),
    "dcv_patch_bwd_det": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _l, _vp], _i),
    "dcv_ortho_fwd_det_ws_floats": ([_i, _i, _i, _i], _l),
    "dcv_ortho_fwd_det": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _l, _vp], _i),
    "dcv_sumsq_det_ws_floats": ([_l], _l),
    "dcv_sumsq_acc_det": ([_vp, _l, _vp, _vp, _l, _vp], _i),
    "dcv_attn_fwd": ([_vp, _vp, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_delta": ([_vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "dcv_attn_bwd_dq": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_dkdv": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_fwd_rows": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_rows": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_dq_rows": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_dkdv_rows": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_fwd_rows_ps": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "dcv_attn_bwd_rows_ps": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_dq_rows_ps": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_attn_bwd_dkdv_rows_ps": ([_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _f, _vp], _i),
    "dcv_im2col_bf16": ([_vp, _i, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _vp], _i),
    "dcv_patch_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "dcv_gather_tokens": ([_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp], _i),
    "dcv_fill_cls": ([_vp, _vp, _vp, _i, _l, _i, _vp], _i),
    "dcv_ortho_fwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "dcv_ortho_bwd": ([_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp], _i),
    "dcv_proxy_loss_supported": ([_i, _i], _i),
    "dcv_proxy_loss": ([_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp], _i),
    "dcv_adamw": ([_vp, _vp, _vp, _vp, _l, _f, _f, _f, _f, _f, _i, _f, _vp], _i),
    "dcv_adamw_dyn": ([_vp, _vp, _vp, _vp, _l, _vp, _vp], _i),
    "dcv_adamw_set_hyper": ([_vp, _f, _f, _f, _f, _f, _i, _f, _vp], _i),
    "dcv_cast_bf16": ([_vp, _vp, _l, _vp], _i),
    "dcv_cast_transpose_bf16": ([_vp, _vp, _vp, _i, _i, _vp], _i),
    "dcv_sumsq_acc": ([_vp, _l, _vp, _vp], _i),
    "dcv_clip_scale": ([_vp, _l, _vp, _f, _vp], _i),
    "dcv_cast_bf16_sr": ([_vp, _vp, _l, _vp, _vp], _i),
    "dcv_cast_transpose_bf16_sr": ([_vp, _vp, _vp, _i, _i, _vp, _vp], _i),
    "dcv_cast_scaled_ranges": ([_vp, _vp, _vp, _vp, _i, _i, _f, _vp, _vp], _i),
}
EXPORTS = tuple(_SIGS)

_lib = None


def load() -> C.CDLL:
    """Loads the in-tree library; raises (never falls back) when it is absent."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                f"{LIB_PATH} not found: the HIP extension is the only compute path of diverse_channel_vit_amd. "
                "Build it with `python -m diverse_channel_vit_amd._build` (needs hipcc, gfx950).")
        lib = C.CDLL(LIB_PATH)
        for name, (args, res) in _SIGS.items():
            fn = getattr(lib, name)
            fn.argtypes, fn.restype = args, res
        _lib = lib
    return _lib


# optional per-kernel timing with events on the launch stream (bench.py).  Records are keyed by the device SYMBOL the entry
# launches plus its problem shape, e.g. ("gemm_nt384_kernel<1>", "M100416 N1536 K384"); every record carries the algorithmic
# and the executed FLOPs and the algorithmic HBM bytes of that launch.
_prof = None          # None | {key: {"ev": [(start, end)], "flops": f, "flops_exec": f, "bytes": b}}
_prof_only = None     # None | set of symbols to time (everything when None)


def set_profiler(enable=False, only=None):
    """enable: start a fresh record set (False/None disables).  only: iterable of symbols to time.  Returns the previous records."""
    global _prof, _prof_only
    old = _prof
    _prof = {} if enable else None
    _prof_only = None if only is None else set(only)
    return old


class _timed:
    def __init__(self, symbol, shape="", flops=0.0, flops_exec=None, nbytes=0.0):
        self.rec = None
        if _prof is not None and (_prof_only is None or symbol in _prof_only):
            self.rec = _prof.setdefault((symbol, shape), {"ev": [], "flops": float(flops), "flops_exec": float(flops if flops_exec is None else flops_exec),
                                                          "bytes": float(nbytes)})

    def __enter__(self):
        if self.rec is not None:
            self.s = torch.cuda.Event(enable_timing=True)
            self.s.record()

    def __exit__(self, *a):
        if self.rec is not None:
            e = torch.cuda.Event(enable_timing=True)
            e.record()
            self.rec["ev"].append((self.s, e))


class _Null:
    def __enter__(self):
        return None

    def __exit__(self, *a):
        return False


_NULL = _Null()


def _timer(describe):
    """describe() -> (symbol, shape, flops, flops_exec, bytes); evaluated only while profiling (no host cost otherwise)."""
    if _prof is None:
        return _NULL
    return _timed(*describe())


_EPI_OUT_BYTES = {0: 2, 1: 4, 2: 4, 3: 2, 4: 2, 5: 8}   # bytes written per output element (GELU: two bf16; PATCH: tokens + pre-embedding copy)
_EPI_AUX_BYTES = {0: 0, 1: 0, 2: 4, 3: 0, 4: 2, 5: 0}   # bytes read per output element besides the operands (residual f32, saved GELU' bf16)


def _check(rc: int, what: str) -> None:
    if rc != 0:
        raise RuntimeError(f"{what} failed: {load().dcv_error_string(rc).decode()} ({rc})")


def _p(t: Optional[torch.Tensor]):
    return None if t is None else C.c_void_p(t.data_ptr())


def _stream():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req(t: torch.Tensor, dtype, name: str):
    if not t.is_cuda:
        raise RuntimeError(f"{name}: expected a GPU tensor (the HIP path has no CPU fallback)")
    if t.dtype != dtype:
        raise RuntimeError(f"{name}: expected {dtype}, got {t.dtype}")
    if not t.is_contiguous():
        raise RuntimeError(f"{name}: expected a contiguous tensor")


# ---------------------------------------------------------------------------------------------
# Deterministic mode — the DEFAULT since round 3 (the reference's trainer sets cudnn.deterministic = True, utils.py:394-401).  Every entry
# that would add partial sums of several workgroups with atomics runs in its *_det form: partials through a workspace, summed in a
# fixed order by a second launch — a training step is bit-reproducible.  Cost measured on the headline step: 35.52 -> 35.78 ms (+0.7 %; the
# weight-gradient GEMMs are 5-7 % FASTER with plain partial stores + one reduction pass than with their fp32-atomic flush, LayerNorm's
# 23 extra reduction launches cost the rest).  set_deterministic(False) / DCV_DETERMINISTIC=0 selects the atomic forms.  One workspace per
# (device, stream), grown on demand; kernels on one stream run in order, so they can share it.
_deterministic = os.environ.get("DCV_DETERMINISTIC", "1") not in ("0", "", "false", "False")
_det_ws = {}
_det_ws_retired = []  # outgrown workspaces, kept alive (see _workspace)


def set_deterministic(on: bool = True) -> bool:
    """Returns the previous setting."""
    global _deterministic
    old, _deterministic = _deterministic, bool(on)
    return old


def is_deterministic() -> bool:
    return _deterministic


def _workspace(n_floats: int, like: torch.Tensor) -> torch.Tensor:
    key = (like.device.index, torch.cuda.current_stream(like.device).cuda_stream)
    ws = _det_ws.get(key)
    if ws is None or ws.numel() < n_floats:
        if ws is not None:
            # never freed: a captured HIP graph (graph.GraphedTrainStep) replays launches that carry the OLD workspace's address; handing
            # that block back to the allocator would let replays write partial tiles into somebody else's memory.  Growth is
            # rare (the first steps of a run) and the sizes small (<= 50 MB each).
            _det_ws_retired.append(ws)
        ws = torch.empty(max(int(n_floats), 1 << 20), dtype=torch.float32, device=like.device)
        _det_ws[key] = ws
    return ws


def _ws_size(n) -> int:
    if n < 0:
        _check(int(n), "det workspace size")
    return int(n)


def gemm_nt(A, W, epilogue, out, *, bias=None, out2=None, aux=None, aux2=None, T=0, n=0, ldo=None, ldo2=None, ldaux=None,
            grid_cap=0, tile=TILE_AUTO):
    """C = A[M,K] @ W[N,K]^T with the given epilogue (see include/dcv.h).  grid_cap / tile: dcv_gemm_nt_ex's launch controls."""
    _req(A, torch.bfloat16, "A"); _req(W, torch.bfloat16, "W")
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K
    ldo = out.shape[-1] if ldo is None else ldo
    ldo2 = (out2.shape[-1] if out2 is not None else 0) if ldo2 is None else ldo2
    ldaux = (aux.shape[-1] if aux is not None else 0) if ldaux is None else ldaux
    lib = load()

    def describe():
        kind = {TILE_WIDE: "gemm_nt384", TILE_PAIR: "gemm_nt_pair", TILE_ALT: "gemm_nt_alt"}.get(lib.dcv_gemm_nt_pick(M, N, K, epilogue, tile), "gemm_nt")
        nbytes = 2.0 * M * K + 2.0 * N * K + M * N * (_EPI_OUT_BYTES[epilogue] + _EPI_AUX_BYTES[epilogue])
        return f"{kind}_kernel<{epilogue}>", f"M{M} N{N} K{K}", 2.0 * M * N * K, None, nbytes

    with _timer(describe):
        rc = lib.dcv_gemm_nt_ex(_p(A), K, _p(W), K, M, N, K, epilogue, _p(bias), _p(out), ldo, _p(out2), ldo2, _p(aux), ldaux,
                                   _p(aux2), T, n, grid_cap, tile, _stream())
    _check(rc, "dcv_gemm_nt")


def gemm_nt_resid_ln(A, W, bias, resid, x_out, gamma, beta, eps, u_out, mean, rstd, *, branch_scale=None, T=0, grid_cap=0):
    """x_out = resid + s (A W^T + bias) and u_out = LayerNorm(x_out) * gamma + beta (bf16), mean / rstd per row, in ONE launch of the
    256 x 384 kernel (include/dcv.h: dcv_gemm_nt_resid_ln; N must be 384).  x_out may alias resid."""
    _req(A, torch.bfloat16, "A"); _req(W, torch.bfloat16, "W"); _req(u_out, torch.bfloat16, "u_out")
    M, K = A.shape
    N = W.shape[0]
    assert W.shape[1] == K and x_out.shape[-1] == N and u_out.shape[-1] == N and resid.shape[-1] == N
    lib = load()
    # the residual GEMM's bytes plus the bf16 LayerNorm output and the row statistics; FLOPs of the product
    nbytes = 2.0 * M * K + 2.0 * N * K + M * N * (4 + 4 + 2) + 8.0 * M
    with _timer(lambda: ("gemm_nt384_kernel<6>", f"M{M} N{N} K{K}", 2.0 * M * N * K, None, nbytes)):
        rc = lib.dcv_gemm_nt_resid_ln(_p(A), K, _p(W), K, M, N, K, _p(bias), _p(resid), N, _p(branch_scale), T, _p(x_out), N, _p(gamma), _p(beta),
                                      float(eps), _p(u_out), N, _p(mean), _p(rstd), grid_cap, _stream())
    _check(rc, "dcv_gemm_nt_resid_ln")


def gemm_tn_det_ws_floats(M, P, Q, tile=TILE_AUTO) -> int:
    n = load().dcv_gemm_tn_det_ws_floats(M, P, Q, tile)
    if n < 0:
        _check(int(n), "dcv_gemm_tn_det_ws_floats")
    return int(n)


class _TnItem(C.Structure):  # include/dcv.h: dcv_tn_item
    _fields_ = [("Y", C.c_void_p), ("X", C.c_void_p), ("dW", C.c_void_p), ("dbias", C.c_void_p),
                ("ldy", C.c_int), ("ldx", C.c_int), ("P", C.c_int), ("Q", C.c_int), ("lddw", C.c_int), ("reserved", C.c_int)]


def gemm_tn_group_supported(shapes, M) -> bool:
    """shapes: iterable of (P, Q).  True when dcv_gemm_tn_group takes them in one launch on this device (every P % 384 == 0, Q % 128 == 0,
    all tiles in one resident round)."""
    shapes = list(shapes)
    if not 1 <= len(shapes) <= 8:
        return False
    arr = (_TnItem * len(shapes))()
    for it, (P, Q) in zip(arr, shapes):
        it.Y = it.X = it.dW = 16  # non-null, aligned placeholders: the size query reads shapes only
        it.ldy, it.ldx, it.P, it.Q, it.lddw = P, Q, P, Q, Q
    return load().dcv_gemm_tn_group_ws_floats(C.cast(arr, C.c_void_p), len(shapes), M) >= 0


def gemm_tn_acc_group(items):
    """items: list of (Y [M,P] bf16, X [M,Q] bf16, dW [P,Q] f32, dbias [P] f32 or None) over the SAME M rows: every dW += Y^T X and dbias +=
    colsum(Y) in ONE launch (dcv_gemm_tn_group); deterministic mode takes the partial tiles through the per-stream workspace."""
    lib = load()
    n = len(items)
    arr = (_TnItem * n)()
    M = items[0][0].shape[0]
    flops = 0.0
    nbytes = 0.0
    for it, (Y, X, dW, db) in zip(arr, items):
        _req(Y, torch.bfloat16, "Y"); _req(X, torch.bfloat16, "X"); _req(dW, torch.float32, "dW")
        if Y.shape[0] != M or X.shape[0] != M or dW.shape != (Y.shape[1], X.shape[1]):
            raise ValueError("gemm_tn_acc_group: every item is (Y [M,P], X [M,Q], dW [P,Q]) over the same M")
        it.Y, it.X, it.dW, it.dbias = Y.data_ptr(), X.data_ptr(), dW.data_ptr(), (db.data_ptr() if db is not None else None)
        it.ldy, it.ldx, it.P, it.Q, it.lddw = Y.stride(0), X.stride(0), Y.shape[1], X.shape[1], dW.stride(0)
        flops += 2.0 * M * Y.shape[1] * X.shape[1]
        nbytes += 2.0 * M * (Y.shape[1] + X.shape[1]) + 8.0 * Y.shape[1] * X.shape[1]
    ptr = C.cast(arr, C.c_void_p)
    ws = None
    if _deterministic:
        ws = _workspace(_ws_size(lib.dcv_gemm_tn_group_ws_floats(ptr, n, M)), items[0][2])
    shape = f"M{M} " + "+".join(f"{it.P}x{it.Q}" for it in arr)
    with _timer(lambda: ("gemm_tn384_group_kernel", shape, flops, flops, nbytes)):
        rc = lib.dcv_gemm_tn_group(ptr, n, M, _p(ws), ws.numel() if ws is not None else 0, _stream())
    _check(rc, "dcv_gemm_tn_group")


def gemm_tn_acc(Y, X, dW, dbias=None, tile=TILE_AUTO, ws=None):
    """dW[P,Q] += Y[M,P]^T @ X[M,Q]; dbias[P] += colsum(Y).  ws (fp32 scratch of >= gemm_tn_det_ws_floats elements): the deterministic
    form — partial tiles through ws, summed in a fixed order (dcv_gemm_tn_acc_det) — instead of fp32 atomics."""
    _req(Y, torch.bfloat16, "Y"); _req(X, torch.bfloat16, "X"); _req(dW, torch.float32, "dW")
    M, P = Y.shape
    Q = X.shape[1]
    assert X.shape[0] == M and dW.numel() == P * Q
    lib = load()
    if ws is None and _deterministic:
        ws = _workspace(_ws_size(lib.dcv_gemm_tn_det_ws_floats(M, P, Q, tile)), dW)
    with _timer(lambda: ("gemm_tn384_kernel" if lib.dcv_gemm_tn_pick(M, P, Q, tile) == TILE_WIDE else "gemm_tn_kernel", f"M{M} P{P} Q{Q}",
                         2.0 * M * P * Q, None, 2.0 * M * (P + Q) + 8.0 * P * Q)):
        if ws is None:
            rc = lib.dcv_gemm_tn_acc_ex(_p(Y), P, _p(X), Q, M, P, Q, _p(dW), Q, _p(dbias), tile, _stream())
        else:
            _req(ws, torch.float32, "ws")
            rc = lib.dcv_gemm_tn_acc_det(_p(Y), P, _p(X), Q, M, P, Q, _p(dW), Q, _p(dbias), tile, _p(ws), ws.numel(), _stream())
    _check(rc, "dcv_gemm_tn_acc")


def ln_fwd(x, gamma, beta, out, mean, rstd, M, D, eps, x_row_stride=None):
    with _timer(lambda: (f"ln_fwd_kernel<{'true' if out.dtype == torch.float32 else 'false'}, {2 if D <= 512 else 4}>", f"M{M} D{D}", 0.0, None,
                         M * D * (4.0 + (4.0 if out.dtype == torch.float32 else 2.0)))):
        rc = load().dcv_ln_fwd(_p(x), D if x_row_stride is None else x_row_stride, _p(gamma), _p(beta), _p(out),
                               1 if out.dtype == torch.float32 else 0, _p(mean), _p(rstd), M, D, eps, _stream())
    _check(rc, "dcv_ln_fwd")


def ln_bwd(du, x, mean, rstd, gamma, dx_in, dx_out, dx_bf16, dgamma, dbeta, M, D, x_row_stride=None, dx_row_stride=None,
           bf16_row_scale=None, rows_per_sample=1):
    """bf16_row_scale [M / rows_per_sample] (fp32): dx_bf16 = bf16(scale[row / rows_per_sample] * dx_out) (DropPath, dcv_ln_bwd_scaled)."""
    with _timer(lambda: (f"ln_bwd_kernel<{'true' if du.dtype == torch.float32 else 'false'}, {2 if D <= 512 else 4}>", f"M{M} D{D}", 0.0, None,
                         M * D * ((4.0 if du.dtype == torch.float32 else 2.0) + 4.0 + (4.0 if dx_in is not None else 0.0) + 4.0 + (2.0 if dx_bf16 is not None else 0.0)))):
        lib = load()
        args = (_p(du), 1 if du.dtype == torch.float32 else 0, _p(x), D if x_row_stride is None else x_row_stride,
                _p(mean), _p(rstd), _p(gamma), _p(dx_in), _p(dx_out), D if dx_row_stride is None else dx_row_stride,
                _p(dx_bf16), _p(dgamma), _p(dbeta), M, D)
        ws = _workspace(_ws_size(lib.dcv_ln_bwd_det_ws_floats(M, D)), dx_out) if _deterministic else None
        if bf16_row_scale is not None:
            _req(bf16_row_scale, torch.float32, "bf16_row_scale")
            rc = lib.dcv_ln_bwd_scaled(*args, _p(bf16_row_scale), rows_per_sample, _p(ws), ws.numel() if ws is not None else 0, _stream())
        elif ws is not None:
            rc = lib.dcv_ln_bwd_det(*args, _p(ws), ws.numel(), _stream())
        else:
            rc = lib.dcv_ln_bwd(*args, _stream())
    _check(rc, "dcv_ln_bwd")


def attn_fwd(qkv, o, lse, B, N, H, hd, scale, nq=None, prescaled=False):
    """nq: only the query rows [0, nq) of every (batch, head) are processed (default: all N).
    prescaled: the q part of qkv holds q * scale * log2(e) (dcv_attn_fwd_rows_ps)."""
    nq_ = N if nq is None else nq
    prod = 2.0 * B * H * nq_ * N * hd  # one (query rows) x N x head_dim product over all (batch, head) pairs
    D_ = H * hd
    with _timer(lambda: (f"attn_fwd3_kernel<{'true' if prescaled else 'false'}>", f"B{B} N{N} H{H} Nq{nq_}", 2 * prod, 2 * prod, 2.0 * B * (N * 2 * D_ + nq_ * 2 * D_) + 4.0 * B * H * nq_)):
        if prescaled:
            rc = load().dcv_attn_fwd_rows_ps(_p(qkv), _p(o), _p(lse), B, N, N if nq is None else nq, H, hd, _stream())
        else:
            rc = load().dcv_attn_fwd_rows(_p(qkv), _p(o), _p(lse), B, N, N if nq is None else nq, H, hd, scale, _stream())
    _check(rc, "dcv_attn_fwd")


def attn_bwd(qkv, o, dO, lse, delta_ws, dqkv, B, N, H, hd, scale, nq=None, prescaled=False):
    lib = load()
    dq_fn = lib.dcv_attn_bwd_dq_rows_ps if prescaled else lib.dcv_attn_bwd_dq_rows
    dkdv_fn = lib.dcv_attn_bwd_dkdv_rows_ps if prescaled else lib.dcv_attn_bwd_dkdv_rows
    nq = N if nq is None else nq
    if delta_ws.numel() < 2 * B * H * N or delta_ws.dtype != torch.float32:
        raise ValueError("attention backward workspace: 2*B*H*N float32 (-delta, then lse*log2e)")
    prod = 2.0 * B * H * nq * N * hd
    D_ = H * hd
    # algorithmic credit: the backward is 4 products (dP, dV, dK, dQ); the S recomputation is executed
    # in both kernels but not credited: dQ kernel 1 credited / 3 executed, dK/dV kernel 3 credited / 4 executed
    tp = "<true>" if prescaled else "<false>"  # the symbols as rocprofv3 prints them (the kernels are templates on the pre-scaled-q form)
    with _timer(lambda: ("attn_bwd_dq2_kernel" + tp, f"B{B} N{N} H{H} Nq{nq}", 1 * prod, 3 * prod, 2.0 * B * (N * 3 * D_ + nq * 3 * D_))):  # also writes the row statistics
        rc = dq_fn(_p(qkv), _p(o), _p(dO), _p(lse), _p(delta_ws), _p(dqkv), B, N, nq, H, hd, scale, _stream())
    _check(rc, "dcv_attn_bwd_dq")
    # pre-scaled q: the persistent third form (csrc/attn_bwd3.hip) + the second form's launch for a key remainder < 129; plain: the second form
    with _timer(lambda: (("attn_bwd_dkdv3p_kernel" if prescaled else "attn_bwd_dkdv2_kernel<false>"), f"B{B} N{N} H{H} Nq{nq}", 3 * prod, 4 * prod, 2.0 * B * (N * 3 * D_ + nq * 1 * D_ + N * 2 * D_))):
        rc = dkdv_fn(_p(qkv), _p(dO), _p(lse), _p(delta_ws), _p(dqkv), B, N, nq, H, hd, scale, _stream())
    _check(rc, "dcv_attn_bwd_dkdv")


def im2col(x, ch_idx, out, B, Ct, C, H, W, P, scale=None, shift=None):
    if x.dtype not in (torch.float32, torch.uint8):
        raise RuntimeError(f"x: expected float32 or uint8 images, got {x.dtype}")
    _req(x, x.dtype, "x"); _req(ch_idx, torch.int32, "ch_idx")
    _check(load().dcv_im2col_bf16(_p(x), 1 if x.dtype == torch.uint8 else 0, _p(ch_idx), _p(scale), _p(shift), _p(out), B, Ct, C, H, W,
                                  P, _stream()), "dcv_im2col_bf16")


def patch_bwd(dx0, dYloss, dY_bf16, dE, dpos, dcls, B, C, n, D):
    lib = load()
    if _deterministic:
        ws = _workspace(_ws_size(lib.dcv_patch_bwd_det_ws_floats(B, C, n, D)), dx0)
        rc = lib.dcv_patch_bwd_det(_p(dx0), _p(dYloss), _p(dY_bf16), _p(dE), _p(dpos), _p(dcls), B, C, n, D, _p(ws), ws.numel(), _stream())
    else:
        rc = lib.dcv_patch_bwd(_p(dx0), _p(dYloss), _p(dY_bf16), _p(dE), _p(dpos), _p(dcls), B, C, n, D, _stream())
    _check(rc, "dcv_patch_bwd")


def gather_tokens(x, idx, out, B, N, Nk, D, scatter=False):
    _check(load().dcv_gather_tokens(_p(x), _p(idx), _p(out), B, N, Nk, D, 1 if scatter else 0, _stream()), "dcv_gather_tokens")


def fill_cls(x, cls, pos0, B, batch_stride, D):
    _check(load().dcv_fill_cls(_p(x), _p(cls), _p(pos0), B, batch_stride, D, _stream()), "dcv_fill_cls")


def ortho_fwd(Y, S, selfsq, tot, inv_norm, stats, B, Cc, n, D):
    lib = load()
    if _deterministic:
        ws = _workspace(_ws_size(lib.dcv_ortho_fwd_det_ws_floats(B, Cc, n, D)), Y)
        rc = lib.dcv_ortho_fwd_det(_p(Y), _p(S), _p(selfsq), _p(tot), _p(inv_norm), _p(stats), B, Cc, n, D, _p(ws), ws.numel(), _stream())
    else:
        rc = lib.dcv_ortho_fwd(_p(Y), _p(S), _p(selfsq), _p(tot), _p(inv_norm), _p(stats), B, Cc, n, D, _stream())
    _check(rc, "dcv_ortho_fwd")


def ortho_bwd(Y, S, tot, inv_norm, coef, dY, B, Cc, n, D):
    _check(load().dcv_ortho_bwd(_p(Y), _p(S), _p(tot), _p(inv_norm), _p(coef), _p(dY), B, Cc, n, D, _stream()), "dcv_ortho_bwd")


def adamw(p, g, m, v, n, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    _check(load().dcv_adamw(_p(p), _p(g), _p(m), _p(v), n, lr, b1, b2, eps, wd, step, grad_scale, _stream()), "dcv_adamw")


def adamw_dyn(p, g, m, v, n, hyper_dev):
    _check(load().dcv_adamw_dyn(_p(p), _p(g), _p(m), _p(v), n, _p(hyper_dev), _stream()), "dcv_adamw_dyn")


def adamw_set_hyper(hyper_dev, lr, b1, b2, eps, wd, step, grad_scale=1.0):
    _check(load().dcv_adamw_set_hyper(_p(hyper_dev), lr, b1, b2, eps, wd, step, grad_scale, _stream()), "dcv_adamw_set_hyper")


def cast_bf16(src, dst, n):
    _check(load().dcv_cast_bf16(_p(src), _p(dst), n, _stream()), "dcv_cast_bf16")


def cast_transpose_bf16(src_base, dst_base, desc_dev, n_desc, max_tiles):
    _check(load().dcv_cast_transpose_bf16(_p(src_base), _p(dst_base), _p(desc_dev), n_desc, max_tiles, _stream()), "dcv_cast_transpose_bf16")


def sumsq_acc(x, n, acc):
    lib = load()
    if _deterministic:
        ws = _workspace(_ws_size(lib.dcv_sumsq_det_ws_floats(n)), x)
        rc = lib.dcv_sumsq_acc_det(_p(x), n, _p(acc), _p(ws), ws.numel(), _stream())
    else:
        rc = lib.dcv_sumsq_acc(_p(x), n, _p(acc), _stream())
    _check(rc, "dcv_sumsq_acc")


def clip_scale(x, n, sumsq_dev, max_norm):
    _check(load().dcv_clip_scale(_p(x), n, _p(sumsq_dev), float(max_norm), _stream()), "dcv_clip_scale")


def cast_bf16_sr(src, dst, n, seed_dev):
    _check(load().dcv_cast_bf16_sr(_p(src), _p(dst), n, _p(seed_dev), _stream()), "dcv_cast_bf16_sr")


def cast_transpose_bf16_sr(src_base, dst_base, desc_dev, n_desc, max_tiles, seed_dev):
    _check(load().dcv_cast_transpose_bf16_sr(_p(src_base), _p(dst_base), _p(desc_dev), n_desc, max_tiles, _p(seed_dev), _stream()),
           "dcv_cast_transpose_bf16_sr")


def cast_scaled_ranges(src_base, dst_bf16, dst_f32, desc_dev, n_desc, blocks_per_desc, scale, seed_dev=None):
    """desc_dev int64 [n_desc][5] = {src offset, dst offset, count, scaled_count, kind (0: bf16 copy, 1: fp32 copy)} (dcv_cast_scaled_ranges)."""
    _check(load().dcv_cast_scaled_ranges(_p(src_base), _p(dst_bf16), _p(dst_f32), _p(desc_dev), n_desc, blocks_per_desc, float(scale),
                                         _p(seed_dev), _stream()), "dcv_cast_scaled_ranges")


def proxy_loss_supported(C: int, D: int) -> bool:
    return bool(load().dcv_proxy_loss_supported(C, D))


def proxy_loss(emb, proxies, scale: float, loss, d_emb, d_proxies):
    """loss[0] = CE(-cdist(scale * normalize(emb), scale * normalize(proxies))^2, eye(C)); d_emb / d_proxies = its gradients (dcv_proxy_loss)."""
    _req(emb, torch.float32, "emb"); _req(proxies, torch.float32, "proxies")
    C_, D_ = emb.shape
    if proxies.shape != emb.shape or not emb.is_contiguous() or not proxies.is_contiguous():
        raise ValueError("proxy_loss: emb and proxies are contiguous [C, D] fp32 tensors of the same shape")
    _check(load().dcv_proxy_loss(_p(emb), _p(proxies), C_, D_, float(scale), _p(loss), _p(d_emb), _p(d_proxies), _stream()), "dcv_proxy_loss")

