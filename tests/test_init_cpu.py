"""Initialisation parity: a freshly constructed diverse_channel_vit_amd.DiChaViT draws its parameters
from the distributions of the reference's constructor (models/dichavit.py:505-516 trunc_normal_(std=0.02) on every Linear
weight / pos_embed / cls_token, zero biases, LayerNorm (1, 0); :60-65 channel_emb_proxies = randn/8 (orthogonal on request);
:83-89 channel_embed orthogonal_ or trunc_normal_; :77-82 Conv3d default; :799-805 classifer_head default Linear, proxies =
randn/8; utils.py:477-517).  tests/golden/init_stats.npz holds the moments and first values of the REAL reference's
parameters under a torch seed (data only).  Checked here on the CPU (construction never touches the GPU):
  * statistically, per parameter: std / mean / extrema against the reference's own draw;
  * exactly: the constructor consumes the RNG in the reference's order, so under the same seed and the same torch build the
    two modules are identical to the last bit."""
import math

import numpy as np
import pytest
import torch

from conftest import load_golden


class Cfg(dict):
    __getattr__ = dict.get


def _build(variant):
    import diverse_channel_vit_amd as dcv
    cfg = Cfg(variant["cfg"], in_channel_names=[f"c{i}" for i in range(variant["n_channels"])], img_size=[variant["img"]],
              num_classes=variant["num_classes"])
    torch.manual_seed(variant["seed"])
    return dcv.dichavit(cfg, mapper={"train": list(range(variant["n_channels"]))})


@pytest.mark.parametrize("v", [0, 1])
def test_init_distributions_match_reference(v):
    meta, a = load_golden("init_stats")
    variant = meta["variants"][v]
    model = _build(variant)
    params = {n: p for n, p in model.named_parameters() if not n.startswith("adaptive_interface")}
    assert list(params) == variant["names"]  # same parameters, same registration order as the reference module
    exact = torch.__version__ == meta["torch_version"]
    n_exact = 0
    for name, p in params.items():
        ref = a[f"v{v}/{name}"]
        numel, mean, std, mn, mx, sm, sq = ref[:7]
        t = p.detach().double().flatten()
        assert t.numel() == int(numel), name
        if std == 0.0:  # constants: zero biases, unit LayerNorm weights, zero-initialised then overwritten tensors
            assert t.std(unbiased=False).item() == 0.0 and abs(t.mean().item() - mean) == 0.0, name
            continue
        n = t.numel()
        # statistical agreement with the reference's own draw (two independent samples of the same law):
        # std within 6 sigma of its sampling error (relative 1/sqrt(2n), with a floor for small tensors), mean within 6 std/sqrt(n)
        assert abs(t.std(unbiased=False).item() - std) <= 6.0 * std / math.sqrt(2 * n) + 0.02 * std + (4.0 / math.sqrt(n) if max(abs(mn), abs(mx)) == 2.0 or t.abs().max().item() == 2.0 else 0.0), (name, t.std().item(), std)
        assert abs(t.mean().item() - mean) <= 8.5 * std / math.sqrt(n) + 1e-12, (name, t.mean().item(), mean)
        if name.endswith("weight") and t.numel() >= 384 * 384 and ("qkv" in name or "fc" in name or "attn.proj" in name):
            # trunc_normal_(std=0.02, a=-2, b=2) = 0.02 * sqrt(2) * erfinv(U(-1, 1)) clamped to +-2: the bulk is N(0, 0.02^2) with
            # extrema ~ 0.02 * sqrt(2 ln n); the only other values possible are EXACTLY +-2.0 — where the float32 uniform draw
            # lands on an endpoint (erfinv = +-inf, then the clamp), about once per 2^24 elements.  The reference's helper
            # (utils.py:477-517) and torch's nn.init.trunc_normal_ share this; with the same seed the same elements are hit.
            big = t.abs() > 0.02 * 6.5
            assert int(big.sum()) <= 3 and bool((t[big].abs() == 2.0).all()), (name, t[big])
            bulk = t[~big]
            assert abs(bulk.std(unbiased=False).item() - 0.02) < 2e-4, name
            assert (abs(mx) == 2.0 or abs(mn) == 2.0) == bool(big.any()) or not exact, name
        if exact:
            got = np.array([t.sum().item(), (t * t).sum().item()] + t[:4].tolist())
            want = np.array([sm, sq] + ref[7:7 + min(4, n)].tolist())
            if np.allclose(got[:2 + min(4, n)], want, rtol=1e-12, atol=1e-12):
                n_exact += 1
    if exact:
        # same seed, same torch: the constructor draws in the reference's order -> identical parameters
        assert n_exact >= len([1 for nm in params if a[f"v{v}/{nm}"][2] != 0.0]), f"only {n_exact} tensors bit-identical"


def test_special_inits_and_constants():
    meta, _ = load_golden("init_stats")
    model = _build(meta["variants"][0])  # orthogonal channel embeddings (train_scripts.sh: orthogonal_channel_emb_init=True)
    fe = model.feature_extractor
    E = fe.patch_embed.channel_embed.weight.detach().double()
    assert torch.allclose(E @ E.t(), torch.eye(E.shape[0], dtype=torch.float64), atol=1e-5)  # dichavit.py:85-87
    for blk in fe.blocks:
        for lin in (blk.attn.qkv, blk.attn.proj, blk.mlp.fc1, blk.mlp.fc2):
            assert float(lin.bias.abs().max()) == 0.0
        for ln in (blk.norm1, blk.norm2):
            assert float((ln.weight - 1).abs().max()) == 0.0 and float(ln.bias.abs().max()) == 0.0
    # Conv3d keeps torch's default (kaiming-uniform, bound 1/sqrt(fan_in) = 1/16 for P = 16): the reference never re-initialises it
    w = fe.patch_embed.proj.weight.detach()
    assert w.abs().max().item() <= 1.0 / 16 + 1e-7 and w.std().item() == pytest.approx((1.0 / 16) / math.sqrt(3), rel=0.05)
    # proxies / channel proxies = randn / 8
    assert model.proxies.detach().std().item() == pytest.approx(0.125, rel=0.05)
    assert fe.patch_embed.channel_emb_proxies.detach().std().item() == pytest.approx(0.125, rel=0.1)
    assert model.proxies.requires_grad and model.scale == pytest.approx(math.sqrt(1 / 0.07))
    model2 = _build(meta["variants"][1])  # trunc-normal channel embeddings, orthogonal channel proxies
    E2 = model2.feature_extractor.patch_embed.channel_embed.weight.detach()
    assert E2.std().item() == pytest.approx(0.02, rel=0.05)
    Pm = model2.feature_extractor.patch_embed.channel_emb_proxies.detach().double()
    assert torch.allclose(Pm @ Pm.t(), torch.eye(Pm.shape[0], dtype=torch.float64), atol=1e-5)


def test_hcs_draws_of_the_reference_reproduced_by_the_product_sampler():
    """the HCS subset now stays on the device (no .cpu() per step).  The draw itself — python RNG for the subset size and
    the anchor, torch.multinomial / topk on the anchor's cosine row, the anchor forced in — is the product's `hcs_pick` / `hcs_force_anchor`
    on whatever device the embeddings live; replayed here on the CPU with the seeds recorded in tests/golden/hcs.npz it reproduces the
    reference's own picks EXACTLY for every recorded draw (torch's CPU generator is the one the reference drew from; a GPU cannot reproduce
    a CPU multinomial stream, which is why the GPU parity tests inject the subset)."""
    import random
    import torch.nn.functional as F
    from diverse_channel_vit_amd.dichavit import hcs_pick, hcs_force_anchor
    from oracle import dichavit_oracle as orc  # test infrastructure: the seeded parameter generator the goldens were built from
    meta, a = load_golden("hcs")
    shapes = orc.state_shapes(meta["cfg"], meta["n_channels"], meta["img"], meta["num_classes"])
    E = orc.make_state(shapes, meta["seed"])["feature_extractor.patch_embed.channel_embed.weight"]
    cur = meta["mapper"]["train"]
    checked = 0
    for k, d in enumerate(meta["draws"]):
        if d["mode"] == "none":
            continue
        rng = random.Random(d["pyseed"])
        torch.manual_seed(d["tseed"])
        Cin_new = rng.randint(1, len(cur))
        anchor = rng.randint(0, len(cur) - 1)
        e = F.normalize(E[cur].detach(), p=2, dim=-1)
        cos = (e @ e.t())[anchor]
        ind = hcs_force_anchor(hcs_pick(cos, Cin_new, d["mode"], d["temp"]), anchor)
        picked = [cur[i] for i in ind.tolist()]
        assert picked == a[f"d{k}_picked"].tolist(), (k, d, picked, a[f"d{k}_picked"].tolist())
        checked += 1
    assert checked >= 5
