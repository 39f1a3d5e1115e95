"""ORACLE — TEST INFRASTRUCTURE ONLY.  Never imported by the product package.

A CPU restatement, written from the maths, of the DiChaViT training hot path of
chaudatascience/diverse_channel_vit (the path BASELINE.json:north_star names):

    per-channel patch tokeniser  ->  ViT encoder (pre-LN MHSA + GELU MLP)  ->
    channel-diversity / orthogonality regularisers  ->  loss

It is plain PyTorch (fp32 or fp64, CPU), differentiable through torch autograd, and
works on a flat ``{state_dict key: tensor}`` mapping that uses exactly the reference's
state-dict keys.  Only ``tests/``, ``__graft_entry__.smoke()`` and the ``cpu_baseline``
leg of ``bench.py`` may import it, and only as the checker / the timed CPU baseline.

Parity status: PINNED.  ``tests/golden/make_golden.py`` imports the real reference from
/root/reference in the build container and stores its outputs as fixtures under
``tests/golden/``; ``tests/test_oracle_golden.py`` checks this restatement against them
(plus the RNG-free known-answer values of SURVEY.md Appendix E).

Every function cites the reference lines it restates (paths relative to /root/reference).
"""
from __future__ import annotations

import math
import random as _pyrandom
from typing import Dict, List, Optional, Sequence, Tuple

import numpy as np
import torch
import torch.nn.functional as F

Tensor = torch.Tensor

# embed dim, depth, heads — models/dichavit.py:676-745 (distill/tiny/small/base factories)
MODEL_SIZES = {
    "tiny": (192, 12, 3),
    "small": (384, 12, 6),
    "base": (768, 12, 12),
    "distill": (384, 12, 6),
}
LN_EPS = 1e-6  # models/dichavit.py:688,706,724,742  (partial(nn.LayerNorm, eps=1e-6))


# --------------------------------------------------------------------------------------
# positional embedding: bicubic resample (models/dichavit.py:518-552)
# --------------------------------------------------------------------------------------
def _cubic_w(t: float) -> List[float]:
    """The four Keys-cubic taps (A = -0.75) at fractional offset t, as torch's
    upsample_bicubic2d uses them."""
    A = -0.75

    def c1(x):  # |x| <= 1
        return ((A + 2.0) * x - (A + 3.0)) * x * x + 1.0

    def c2(x):  # 1 < |x| < 2
        return ((A * x - 5.0 * A) * x + 8.0 * A) * x - 4.0 * A

    return [c2(t + 1.0), c1(t), c1(1.0 - t), c2(2.0 - t)]


def bicubic_matrix_1d(g_in: int, g_out: int, scale: float) -> np.ndarray:
    """R[o, i]: 1-D resampling matrix of F.interpolate(mode='bicubic', align_corners=False,
    scale_factor=scale) — source coordinate uses 1/scale (no recompute_scale_factor),
    taps are index-clamped.  models/dichavit.py:541-545."""
    R = np.zeros((g_out, g_in), dtype=np.float64)
    for o in range(g_out):
        f = (o + 0.5) / scale - 0.5
        i0 = math.floor(f)
        t = f - i0
        for k, wk in enumerate(_cubic_w(t)):
            i = min(max(i0 - 1 + k, 0), g_in - 1)
            R[o, i] += wk
    return R


class _BicubicResample(torch.autograd.Function):
    """Forward: out[o,p,:] = sum_{y,x} Ry[o,y] Rx[p,x] grid[y,x,:].
    Backward: the exact adjoint, EXCEPT when the output grid has the input's size — then
    ATen's upsample_bicubic2d_backward takes its "same size: just copy" early-out and the
    gradient passes straight through (identity), although the forward did resample with
    1/scale.  Observed with the reference in this container (torch 2.10 CPU); in training the
    grid size never changes, so d(pos_embed[1:]) == sum over channels and batch of d(tokens)."""

    @staticmethod
    def forward(ctx, grid, Ry, Rx):
        ctx.save_for_backward(Ry, Rx)
        ctx.same = (Ry.shape[0] == Ry.shape[1]) and (Rx.shape[0] == Rx.shape[1])
        return torch.einsum("oy,px,yxd->opd", Ry, Rx, grid)

    @staticmethod
    def backward(ctx, g):
        Ry, Rx = ctx.saved_tensors
        if ctx.same:
            return g, None, None
        return torch.einsum("oy,px,opd->yxd", Ry, Rx, g), None, None


def pos_embed_for(pos_embed: Tensor, n_tokens_minus_cls: int, w_img: int, h_img: int, nc: int, patch: int) -> Tensor:
    """interpolate_pos_encoding (models/dichavit.py:518-552).  pos_embed [1, 1+g*g, D].
    Returns [1, 1+nc*w0*h0, D]."""
    n_pos = pos_embed.shape[1] - 1
    if n_tokens_minus_cls == n_pos and w_img == h_img:  # :529-530 (nc == 1 at the native size; also nc > 1 when nc * n == n_pos)
        return pos_embed
    D = pos_embed.shape[-1]
    g = int(math.sqrt(n_pos))
    w0, h0 = w_img // patch, h_img // patch
    sw, sh = (w0 + 0.1) / math.sqrt(n_pos), (h0 + 0.1) / math.sqrt(n_pos)  # :540-543
    ow, oh = int(math.floor(g * sw)), int(math.floor(g * sh))
    assert ow == w0 and oh == h0  # :546
    # reference reshapes to (1, g, g, D) -> NCHW; first spatial axis gets scale sw, second sh
    Ry = torch.from_numpy(bicubic_matrix_1d(g, ow, sw)).to(pos_embed)  # dtype AND device of the table (the checker may run on the device: tests)
    Rx = torch.from_numpy(bicubic_matrix_1d(g, oh, sh)).to(pos_embed)
    grid = pos_embed[0, 1:].reshape(g, g, D)
    out = _BicubicResample.apply(grid, Ry, Rx).reshape(1, ow * oh, D)
    out = out.expand(nc, ow * oh, D).reshape(1, nc * ow * oh, D)  # :550 same tile per channel
    return torch.cat([pos_embed[:, :1], out], dim=1)  # :552


# --------------------------------------------------------------------------------------
# regularisers (models/loss_fn.py)
# --------------------------------------------------------------------------------------
def ortho_loss_dense(feat: Tensor, labels: Tensor, gamma_s: float, gamma_d: float,
                     reverse_pos_pairs: bool, use_square: bool) -> Tensor:
    """ortho_proj_loss_fn_v2 exactly as the reference forms it: dense [B,T,T] cosine matrix
    and boolean masks (models/loss_fn.py:24-59)."""
    f = F.normalize(feat, p=2, dim=-1)  # :33 (eps 1e-12)
    same = labels[:, None] == labels[None, :]  # :37
    eye = torch.eye(labels.numel(), dtype=torch.bool, device=feat.device)
    m_pos = (same & ~eye).to(feat.dtype)  # :40
    m_neg = (~same).to(feat.dtype)  # :41
    dots = f @ f.transpose(-2, -1)  # :42
    pos = (m_pos * dots).sum(dim=(-2, -1)) / _count_eps(m_pos.sum().item())  # :44,47
    neg = (m_neg * dots).sum(dim=(-2, -1)) / _count_eps(m_neg.sum().item())  # :45,48
    return _combine_ortho(pos, neg, gamma_s, gamma_d, reverse_pos_pairs, use_square)


def _count_eps(count: float) -> float:
    """mask.sum() + 1e-6 as the reference evaluates it: the masks are ``.float()`` (fp32,
    models/loss_fn.py:40-41) whatever the feature dtype, so the sum and the +1e-6 are rounded
    in fp32 (the 1e-6 vanishes once the count is >= 32)."""
    return float(np.float32(np.float32(count) + np.float32(1e-6)))


def _combine_ortho(pos, neg, gamma_s, gamma_d, reverse_pos_pairs, use_square):
    if use_square:  # :50-51
        neg = neg ** 2
    if reverse_pos_pairs:  # :53-56
        if use_square:
            pos = pos ** 2
        loss = gamma_s * pos + gamma_d * neg
    else:  # :58
        loss = gamma_s * (1.0 - pos) + gamma_d * neg
    return loss.mean()  # :59


def ortho_loss_linear(feat: Tensor, C: int, n: int, gamma_s: float, gamma_d: float,
                      reverse_pos_pairs: bool, use_square: bool) -> Tensor:
    """Same value through the O(T*D) identity of SURVEY §2.3 K6 (tokens are channel-major,
    label(t) = t // n):  with fh = normalised tokens, s_c = sum_{t in c} fh_t,
        pos_sum = sum_c (|s_c|^2 - sum_{t in c} |fh_t|^2),   neg_sum = |sum_c s_c|^2 - sum_c |s_c|^2.
    This is the formulation the HIP kernel implements; the dense form above is what the
    reference computes.  tests check the two agree."""
    B, T, D = feat.shape
    assert T == C * n
    f = F.normalize(feat, p=2, dim=-1)
    fc = f.reshape(B, C, n, D)
    s = fc.sum(dim=2)  # [B,C,D]
    s_sq = (s * s).sum(-1)  # [B,C]
    self_sq = (fc * fc).sum(-1).sum(-1)  # [B,C]  (== n except for zero tokens)
    tot = s.sum(dim=1)
    pos_sum = (s_sq - self_sq).sum(-1)
    neg_sum = (tot * tot).sum(-1) - s_sq.sum(-1)
    pos = pos_sum / _count_eps(C * n * (n - 1))
    if C == 1:  # no different-channel pair exists: the reference's mask_neg is all-zero -> exactly 0
        neg = torch.zeros_like(pos)
    else:
        neg = neg_sum / _count_eps(T * T - C * n * n)
    return _combine_ortho(pos, neg, gamma_s, gamma_d, reverse_pos_pairs, use_square)


def proxy_loss(proxies: Tensor, emb: Tensor, target: Tensor, scale: float) -> Tensor:
    """proxy_loss (models/loss_fn.py:7-21) with pairwise_distance_v2(squared=True)
    (utils.py:461-465):  logits[i,j] = -|s*n(emb_i) - s*n(proxy_j)|^2,  CrossEntropy(mean)
    with either class indices or probability targets."""
    p = scale * F.normalize(proxies, p=2, dim=-1)
    e = scale * F.normalize(emb, p=2, dim=-1)
    d2 = ((e[:, None, :] - p[None, :, :]) ** 2).sum(-1)
    logp = F.log_softmax(-d2, dim=-1)
    if target.dtype in (torch.int64, torch.int32):
        return -logp.gather(1, target.long()[:, None]).mean()
    return -(target.to(logp.dtype) * logp).sum(-1).mean()


# --------------------------------------------------------------------------------------
# tokeniser (models/dichavit.py:110-126, 377-417) — train path, channel subset injected
# --------------------------------------------------------------------------------------
def unfold_patches(x: Tensor, P: int) -> Tensor:
    """[B,C,H,W] -> [B, C*h*w, P*P]; token t = c*h*w + i*w + j, k = u*P + v.  This is the
    im2col of Conv3d(1,D,(1,P,P),stride (1,P,P)) on x.unsqueeze(1) (models/dichavit.py:77-82,377)."""
    B, C, H, W = x.shape
    h, w = H // P, W // P
    x = x[:, :, : h * P, : w * P]
    return x.reshape(B, C, h, P, w, P).permute(0, 1, 2, 4, 3, 5).reshape(B, C * h * w, P * P)


def patch_tokens(sd: Dict[str, Tensor], x: Tensor, P: int) -> Tensor:
    """conv output WITH bias, before channel-emb/pos: [B,T,D]  (models/dichavit.py:377)."""
    Wc = sd["feature_extractor.patch_embed.proj.weight"]
    D = Wc.shape[0]
    return unfold_patches(x, P) @ Wc.reshape(D, P * P).t() + sd["feature_extractor.patch_embed.proj.bias"]


def tokenise(sd: Dict[str, Tensor], x: Tensor, cfg, ch_ids: Sequence[int], idx: Sequence[int],
             channel_embed_rows: Optional[Tensor] = None) -> Tuple[Tensor, Tensor]:
    """PatchEmbedPerChannel.forward + prepare_tokens (models/dichavit.py:377-417, 554-565).

    ch_ids : global channel ids used this step (rows of channel_embed / channel_emb_proxies)
    idx    : their positions inside x's channel axis
    channel_embed_rows : [C,D] override (eval-time synthesised rows, :219-374).
    Returns (tokens [B,1+T,D], extra_loss 0-d)."""
    P = cfg["patch_size"]
    B, _, Hi, Wi = x.shape
    xs = x[:, list(idx)]
    C = len(idx)
    Y = patch_tokens(sd, xs, P)  # [B,T,D]
    n = Y.shape[1] // C
    extra = torch.zeros((), dtype=Y.dtype, device=Y.device)
    lam_o = cfg.get("ortho_loss_v1_lambda", 0) or 0
    lam_p = cfg.get("proxy_loss_lambda", 0) or 0
    if lam_o > 0:  # :378-389
        extra = extra + lam_o * ortho_loss_linear(
            Y, C, n, cfg["gamma_s"], cfg["gamma_d"], cfg["reverse_pos_pairs"], cfg["use_square"])
    if not cfg.get("use_channelvit_channels", True):  # :83-95, 121, 409: no channel_embed parameter, no channel offset on the tokens
        assert lam_p == 0, "the reference leaves channel_embed unbound in this mode: the proxy term cannot be evaluated"
        E = torch.zeros(C, Y.shape[-1], dtype=Y.dtype, device=Y.device)
    else:
        E_all = sd["feature_extractor.patch_embed.channel_embed.weight"]
        E = E_all[list(ch_ids)] if channel_embed_rows is None else channel_embed_rows  # :122,136/212
    if lam_p > 0:  # :399-402
        Pr = sd["feature_extractor.patch_embed.channel_emb_proxies"][list(ch_ids)]
        s = math.sqrt(1.0 / cfg["temperature"])  # :60
        extra = extra + lam_p * proxy_loss(Pr, E, torch.eye(C, dtype=Y.dtype, device=Y.device), s)
    Z = Y + E.repeat_interleave(n, dim=0)[None]  # :409-411 (token t = c*n + i)
    cls = sd["feature_extractor.cls_token"].expand(B, -1, -1)
    Z = torch.cat([cls, Z], dim=1)  # :561-562
    Z = Z + pos_embed_for(sd["feature_extractor.pos_embed"], C * n, Hi, Wi, C, P)  # :565
    return Z, extra


# --------------------------------------------------------------------------------------
# encoder (models/vit.py:59-82, 101-144, 346-399; models/dichavit.py:645-652)
# --------------------------------------------------------------------------------------
def block_forward(sd: Dict[str, Tensor], pre: str, z: Tensor, heads: int, drop: Optional[Tuple[Tensor, Tensor, float]] = None) -> Tensor:
    """drop = (keep mask of the attention branch [B], keep mask of the MLP branch [B], keep probability): stochastic depth, vit.py:37-56,
    397-398 — each residual branch of sample b is multiplied by mask_b / keep_prob."""
    B, N, D = z.shape
    hd = D // heads
    u = F.layer_norm(z, (D,), sd[pre + "norm1.weight"], sd[pre + "norm1.bias"], LN_EPS)
    qkv = u @ sd[pre + "attn.qkv.weight"].t() + sd[pre + "attn.qkv.bias"]
    qkv = qkv.reshape(B, N, 3, heads, hd).permute(2, 0, 3, 1, 4)  # vit.py:123
    q, k, v = qkv[0], qkv[1], qkv[2]
    att = (q @ k.transpose(-2, -1)) * hd ** -0.5  # vit.py:126 (scale after the product)
    att = att.softmax(dim=-1)
    o = (att @ v).transpose(1, 2).reshape(B, N, D)
    sa = sm = 1.0
    if drop is not None:
        sa = (drop[0].to(z.dtype) / drop[2]).view(B, 1, 1)
        sm = (drop[1].to(z.dtype) / drop[2]).view(B, 1, 1)
    z = z + sa * (o @ sd[pre + "attn.proj.weight"].t() + sd[pre + "attn.proj.bias"])
    u = F.layer_norm(z, (D,), sd[pre + "norm2.weight"], sd[pre + "norm2.bias"], LN_EPS)
    hdn = F.gelu(u @ sd[pre + "mlp.fc1.weight"].t() + sd[pre + "mlp.fc1.bias"])  # exact erf GELU
    return z + sm * (hdn @ sd[pre + "mlp.fc2.weight"].t() + sd[pre + "mlp.fc2.bias"])


def drop_path_rates(cfg) -> List[float]:
    """Per-block drop probabilities: linspace(0, drop_path_rate, depth) (models/dichavit.py:475); block 0 never drops."""
    _, depth, _ = MODEL_SIZES[cfg["pretrained_model_name"]]
    r = float(cfg.get("drop_path_rate", 0.0) or 0.0)
    return [float(v) for v in torch.linspace(0, r, depth)]


def encode(sd: Dict[str, Tensor], z: Tensor, cfg, drop_masks: Optional[Sequence[Tensor]] = None) -> Tensor:
    """drop_masks (training with drop_path_rate > 0): the 0/1 keep masks [B] in the order the reference draws them — for every block with
    a non-zero rate, the attention branch then the MLP branch (vit.py:397-398)."""
    D, depth, heads = MODEL_SIZES[cfg["pretrained_model_name"]]
    dpr = drop_path_rates(cfg)
    k = 0
    for i in range(depth):
        drop = None
        if drop_masks is not None and dpr[i] > 0.0:
            drop = (drop_masks[k], drop_masks[k + 1], 1.0 - dpr[i])
            k += 2
        z = block_forward(sd, f"feature_extractor.blocks.{i}.", z, heads, drop)
    f = F.layer_norm(z, (D,), sd["feature_extractor.norm.weight"], sd["feature_extractor.norm.bias"], LN_EPS)
    return f[:, 0]  # dichavit.py:651-652


def token_keep(mode: Optional[str], nc: int, n: int, rng: _pyrandom.Random) -> Optional[List[int]]:
    """dropout_tokens_hcs (models/dichavit.py:568-627): token positions kept (CLS = 0 first), training only.
    cinHW counts the CLS token (x.shape[1]), exactly as the reference does."""
    if mode in (None, "none"):
        return None
    cinHW = 1 + nc * n
    HW = cinHW // nc
    if mode in ("random", "token_random50"):
        k = (rng.randint(1, nc) if mode == "random" else int(math.ceil(0.5 * nc))) * HW  # :571 / :617
        chosen = set(rng.sample(range(cinHW), k=k))  # :572 / :618
        return [0] + [i for i in range(1, cinHW) if i in chosen]  # :574-579
    k = rng.randint(1, nc) if mode == "channel" else int(math.ceil(0.5 * nc))  # :585 / :602
    chans = set(rng.sample(range(nc), k=k))  # :587 / :604
    keep = [0]
    for c in range(nc):
        if c in chans:
            keep.extend(range(1 + c * HW, 1 + (c + 1) * HW))
    return keep


def forward(sd: Dict[str, Tensor], x: Tensor, cfg, ch_ids: Sequence[int], idx: Sequence[int],
            channel_embed_rows: Optional[Tensor] = None, keep: Optional[Sequence[int]] = None,
            drop_masks: Optional[Sequence[Tensor]] = None) -> Tuple[Tensor, Tensor]:
    """DiChaViT.forward (models/dichavit.py:844-861).  Returns (logits-or-features, extra).
    keep: token positions that survive dropout_tokens_hcs (from token_keep); drop_masks: see encode."""
    z, extra = tokenise(sd, x, cfg, ch_ids, idx, channel_embed_rows)
    if keep is not None:
        z = z[:, list(keep)]
    f = encode(sd, z, cfg, drop_masks)
    if "classifer_head.weight" in sd:  # absent for CHAMMI (dichavit.py:797-801)
        f = f @ sd["classifer_head.weight"].t() + sd["classifer_head.bias"]
    return f, extra


def train_loss(sd, x, y, cfg, ch_ids, idx, extra_loss_lambda: float = 1.0, keep=None, drop_masks=None):
    """train_one_batch_regular's loss (trainer.py:986-995)."""
    logits, extra = forward(sd, x, cfg, ch_ids, idx, keep=keep, drop_masks=drop_masks)
    main = F.cross_entropy(logits, y)
    return main + extra_loss_lambda * extra, main, extra, logits


def chammi_loss(sd, x, y, cfg, ch_ids, idx, extra_loss_lambda: float = 1.0):
    """train_one_batch_morphem70k's per-chunk loss (trainer.py:912-914): proxy loss of the
    features against DiChaViT.proxies with integer labels."""
    feat, extra = forward(sd, x, cfg, ch_ids, idx)
    s = math.sqrt(1.0 / cfg["temperature"])
    main = proxy_loss(sd["proxies"], feat, y, s)
    return main + extra_loss_lambda * extra, main, extra, feat


# --------------------------------------------------------------------------------------
# HCS channel sampling (models/dichavit.py:127-216) — host side, RNG injectable
# --------------------------------------------------------------------------------------
def proj_cosine(sd: Dict[str, Tensor], x: Tensor, P: int) -> Tensor:
    """Cosine of the PROJECTED input per channel pair, batch mean (hcs_sampling=*_proj, models/dichavit.py:156-161):
    x_sim[b,c] = normalize(flatten_(h w d)(conv(x)[b,:,c])), cos[c,e] = mean_b <x_sim[b,c], x_sim[b,e]>."""
    B, C = x.shape[:2]
    Y = patch_tokens(sd, x, P)  # [B, C*n, D], token t = c*n + i
    xs = F.normalize(Y.reshape(B, C, -1), p=2, dim=-1)
    return torch.einsum("bcd,bed->bce", xs, xs).mean(dim=0)


def hcs_sample(channel_embed_rows: Tensor, cur_channels: Sequence[int], mode: Optional[str], temp: float,
               rng: _pyrandom.Random, multinomial=None, proj_cos: Optional[Tensor] = None) -> Tuple[List[int], List[int]]:
    """Returns (sampled global channel ids in sampled order, positions within cur_channels).
    Draw order: randint(1,C) -> [randint(0,C-1) -> multinomial].
    proj_cos: the [C,C] matrix of proj_cosine() for mode 'lowest_cosine_prob_proj'."""
    C = len(cur_channels)
    k = rng.randint(1, C)  # :128
    if mode in (None, "none"):
        picked = rng.sample(list(cur_channels), k=k)  # :131
        return picked, [list(cur_channels).index(c) for c in picked]
    anchor = rng.randint(0, C - 1)  # :154
    if mode is not None and mode.endswith("_proj"):
        cos = proj_cos.detach()[anchor]  # :156-161
    else:
        e = F.normalize(channel_embed_rows.detach(), p=2, dim=-1)
        cos = (e @ e.t())[anchor]  # :169-174
    if mode == "lowest_cosine":
        ind = torch.topk(cos, k=k, largest=False).indices.tolist()  # :177
    elif mode == "highest_cosine":
        ind = torch.topk(cos, k=k, largest=True).indices.tolist()  # :183
    elif mode in ("lowest_cosine_prob", "lowest_cosine_prob_proj"):
        prob = F.softmax((1 - cos) / temp, dim=-1)  # :194-196
        ind = (multinomial or torch.multinomial)(prob, k, replacement=False).tolist()  # :199
    else:
        raise ValueError(f"Invalid hcs_sampling: '{mode}'")  # :206
    if anchor not in ind:  # :201-202
        ind[-1] = anchor
    return [cur_channels[i] for i in ind], ind


# --------------------------------------------------------------------------------------
# eval-time channel embeddings for unseen channels (models/dichavit.py:219-374)
# --------------------------------------------------------------------------------------
def eval_channel_embed(E_all: Tensor, mapper: Dict[str, List[int]], chunk_name: str,
                       training_chunks: str, new_channel_init: str) -> Tensor:
    train_ch = [c for ch in training_chunks.split("_") for c in mapper[ch]]  # :220-222
    not_seen = [c for c in train_ch if c not in mapper[chunk_name]]  # :223
    bank = not_seen if "not_in_chunk" in new_channel_init else train_ch  # :236
    rows, cur = [], 0
    for c in mapper[chunk_name]:
        if c in train_ch:
            rows.append(E_all[c][None])  # :365-366
            continue
        if new_channel_init in ("avg_2", "avg_2_not_in_chunk"):  # :240-244
            r = E_all[[bank[cur], bank[(cur + 1) % len(bank)]]].mean(0, keepdim=True)
        elif new_channel_init in ("avg_3", "avg_3_not_in_chunk"):  # :245-250
            r = E_all[[bank[cur], bank[(cur + 1) % len(bank)], bank[(cur + 2) % len(bank)]]].mean(0, keepdim=True)
        elif new_channel_init == "replicate":  # :251-254
            r = E_all[bank[cur]][None]
        elif new_channel_init == "zero":  # :255-256
            r = torch.zeros_like(E_all[0])[None]
        elif new_channel_init == "random":  # :257-258
            r = E_all[c][None]
        else:
            raise ValueError(f"Invalid new_channel_init: '{new_channel_init}'")  # :362
        cur = (cur + 1) % len(bank)  # :363
        rows.append(r)
    return torch.cat(rows, dim=0)  # :374


# --------------------------------------------------------------------------------------
# optimiser restatement (timm AdamW == decoupled-WD Adam == torch.optim.AdamW)
# --------------------------------------------------------------------------------------
def adamw_step(p: Tensor, g: Tensor, m: Tensor, v: Tensor, step: int, lr: float, b1: float, b2: float,
               eps: float, wd: float) -> None:
    """In-place decoupled AdamW update, bias-corrected (optimizers.py:20-21 -> timm AdamW;
    configs/optimizer/adamw.yaml).  p <- p*(1-lr*wd);  p <- p - lr/bc1 * m / (sqrt(v)/sqrt(bc2)+eps)."""
    p.mul_(1.0 - lr * wd)
    m.mul_(b1).add_(g, alpha=1.0 - b1)
    v.mul_(b2).addcmul_(g, g, value=1.0 - b2)
    bc1 = 1.0 - b1 ** step
    bc2 = 1.0 - b2 ** step
    denom = (v.sqrt() / math.sqrt(bc2)).add_(eps)
    p.addcdiv_(m, denom, value=-lr / bc1)


# --------------------------------------------------------------------------------------
# deterministic state / input generators shared by the golden script and the tests
# --------------------------------------------------------------------------------------
def state_shapes(cfg, n_channels: int, img: int, num_classes: int, chammi: bool = False) -> Dict[str, Tuple[int, ...]]:
    """The reference's state_dict layout (156 entries for S with both lambdas > 0)."""
    D, depth, heads = MODEL_SIZES[cfg["pretrained_model_name"]]
    P = cfg["patch_size"]
    n = (img // P) ** 2
    fe = "feature_extractor."
    sh: Dict[str, Tuple[int, ...]] = {
        "proxies": (num_classes, D),
        fe + "cls_token": (1, 1, D),
        fe + "pos_embed": (1, n + 1, D),
        fe + "patch_embed.proj.weight": (D, 1, 1, P, P),
        fe + "patch_embed.proj.bias": (D,),
        fe + "patch_embed.channel_embed.weight": (n_channels, D),
    }
    if not cfg.get("use_channelvit_channels", True):
        del sh[fe + "patch_embed.channel_embed.weight"]
    if (cfg.get("proxy_loss_lambda", 0) or 0) > 0:
        sh[fe + "patch_embed.channel_emb_proxies"] = (n_channels, D)
    for i in range(depth):
        b = f"{fe}blocks.{i}."
        sh.update({
            b + "norm1.weight": (D,), b + "norm1.bias": (D,),
            b + "attn.qkv.weight": (3 * D, D), b + "attn.qkv.bias": (3 * D,),
            b + "attn.proj.weight": (D, D), b + "attn.proj.bias": (D,),
            b + "norm2.weight": (D,), b + "norm2.bias": (D,),
            b + "mlp.fc1.weight": (4 * D, D), b + "mlp.fc1.bias": (4 * D,),
            b + "mlp.fc2.weight": (D, 4 * D), b + "mlp.fc2.bias": (D,),
        })
    sh[fe + "norm.weight"] = (D,)
    sh[fe + "norm.bias"] = (D,)
    if not chammi:
        sh["classifer_head.weight"] = (num_classes, D)
        sh["classifer_head.bias"] = (num_classes,)
    return sh


def make_state(shapes: Dict[str, Tuple[int, ...]], seed: int, dtype=torch.float32) -> Dict[str, Tensor]:
    """Version-stable pseudo-random parameters (numpy legacy RandomState) with the magnitudes
    of the reference's initialisers but non-zero biases so every term is exercised."""
    rs = np.random.RandomState(seed)
    out: Dict[str, Tensor] = {}
    for name in sorted(shapes):
        shp = shapes[name]
        if name.endswith("norm1.weight") or name.endswith("norm2.weight") or name.endswith("norm.weight"):
            a = 1.0 + 0.05 * rs.standard_normal(shp)
        elif name.endswith(".bias"):
            a = 0.02 * rs.standard_normal(shp)
        elif name.endswith("proj.weight") and len(shp) == 5:
            a = rs.uniform(-1.0, 1.0, shp) / math.sqrt(shp[-1] * shp[-2])
        elif name.endswith("channel_embed.weight"):
            a = rs.standard_normal(shp)
            a /= np.linalg.norm(a, axis=-1, keepdims=True)
        elif name == "proxies" or name.endswith("channel_emb_proxies"):
            a = rs.standard_normal(shp) / 8.0
        else:
            a = np.clip(0.02 * rs.standard_normal(shp), -0.04, 0.04)
        out[name] = torch.from_numpy(np.ascontiguousarray(a)).to(dtype)
    return out


def make_batch(seed: int, B: int, C: int, img: int, num_classes: int, dtype=torch.float32):
    rs = np.random.RandomState(seed)
    x = torch.from_numpy(rs.standard_normal((B, C, img, img))).to(dtype)
    y = torch.from_numpy(rs.randint(0, num_classes, B)).long()
    return x, y
