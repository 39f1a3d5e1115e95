"""Pins the oracle (oracle/dichavit_oracle.py, our CPU restatement) against outputs of the REAL
reference captured in tests/golden/*.npz by tests/golden/make_golden.py, and against the
RNG-free known answers of SURVEY.md Appendix E.  CPU only."""
import math
import random

import numpy as np
import pytest
import torch

from conftest import load_golden
from oracle import dichavit_oracle as orc


def _state(meta, dtype=torch.float64):
    shapes = orc.state_shapes(meta["cfg"], meta["n_channels"], meta["img"], meta["num_classes"],
                              chammi="Allen" in meta["mapper"])
    sd = orc.make_state(shapes, meta["seed"], dtype=dtype)
    for v in sd.values():
        v.requires_grad_(True)
    return sd


def _check_grads(sd, arrays, rtol=2e-4, atol_frac=2e-4):
    checked = 0
    for k, v in arrays.items():
        if k.startswith("gnorm/"):
            name = k[6:]
            g = sd[name].grad
            assert g is not None, name
            gn = float(v)
            assert abs(g.norm().item() - gn) <= rtol * gn + 1e-12, (name, g.norm().item(), gn)
            samp = arrays["gsamp/" + name]
            idx = np.unique(np.linspace(0, g.numel() - 1, min(64, g.numel())).astype(np.int64))
            mine = g.flatten()[idx].numpy()
            scale = max(np.abs(samp).max(), gn / math.sqrt(g.numel()))
            assert np.abs(mine - samp).max() <= atol_frac * scale + 1e-12, (name, np.abs(mine - samp).max(), scale)
            checked += 1
        elif k.startswith("gnone/"):
            assert sd[k[6:]].grad is None, k
    assert checked > 100


# ---------------------------------------------------------------------------------------------
def test_appendix_e_known_answers():
    """SURVEY App. E1/E2 (values produced by the reference, fp64)."""
    B, C, n, D = 2, 3, 4, 8
    i = torch.arange(B * C * n * D, dtype=torch.float64).reshape(B, C * n, D)
    f = torch.sin(0.37 * i) + 0.25 * torch.cos((0.11 * i * i) % 7.0)
    labels = torch.arange(C).repeat_interleave(n)
    want = {(1.0, 4.0, True, False): -0.291791598420797, (0.5, 2.0, True, False): -0.145895799210398,
            (1.0, 0.5, False, False): 1.300793062274421, (1.0, 4.0, True, True): 0.090006307274815}
    for (gs, gd, rev, sq), val in want.items():
        assert abs(orc.ortho_loss_dense(f, labels, gs, gd, rev, sq).item() - val) < 1e-12
        assert abs(orc.ortho_loss_linear(f, C, n, gs, gd, rev, sq).item() - val) < 1e-12
    j = torch.arange(40, dtype=torch.float64).reshape(5, 8)
    prox, emb, sc = torch.cos(0.3 * j), torch.sin(0.2 * j + 1.0), math.sqrt(1 / 0.07)
    assert abs(orc.proxy_loss(prox, emb, torch.eye(5, dtype=torch.float64), sc).item() - 35.173976216460602) < 1e-10
    assert abs(orc.proxy_loss(prox, emb, torch.arange(5), sc).item() - 35.173976216460595) < 1e-10


def test_bicubic_matches_torch_interpolate():
    """the explicit tap/weight matrix equals F.interpolate(bicubic, scale=(g+.1)/g)."""
    for g, D in [(14, 8), (4, 5), (8, 3)]:
        pe = torch.randn(1, 1 + g * g, D, dtype=torch.float64, generator=torch.Generator().manual_seed(g))
        mine = orc.pos_embed_for(pe, 3 * g * g, g * 16, g * 16, 3, 16)
        s = (g + 0.1) / g
        ref = torch.nn.functional.interpolate(pe[:, 1:].reshape(1, g, g, D).permute(0, 3, 1, 2), scale_factor=(s, s), mode="bicubic")
        ref = ref.permute(0, 2, 3, 1).reshape(1, g * g, D)
        assert (mine[:, 1:1 + g * g] - ref).abs().max() < 1e-13
        assert (mine[:, 1 + g * g:1 + 2 * g * g] - ref).abs().max() < 1e-13
        assert (mine[:, 1:1 + g * g] - pe[:, 1:]).abs().max() > 1e-3  # NOT the identity
        # C == 1 -> raw table (dichavit.py:529-530)
        assert orc.pos_embed_for(pe, g * g, g * 16, g * 16, 1, 16) is pe


def test_loss_fn_goldens():
    meta, a = load_golden("loss_fns")
    for c in meta["cases"]:
        if c["kind"] != "rand":
            continue
        k = c["k"]
        feat = torch.from_numpy(a[f"r_{k}_feat"]).requires_grad_(True)
        val = orc.ortho_loss_linear(feat, c["C"], c["n"], c["gs"], c["gd"], c["rev"], c["sq"])
        val.backward()
        assert abs(val.item() - float(a[f"r_{k}_val"])) < 1e-12
        assert np.abs(feat.grad.numpy() - a[f"r_{k}_grad"]).max() < 1e-12
        labels = torch.arange(c["C"]).repeat_interleave(c["n"])
        assert abs(orc.ortho_loss_dense(feat.detach(), labels, c["gs"], c["gd"], c["rev"], c["sq"]).item() - float(a[f"r_{k}_val"])) < 1e-12


@pytest.mark.parametrize("name", ["tiny_e2e", "so2sat_s", "jumpcp_s"])
def test_train_step_goldens(name):
    """logits / losses / every parameter gradient of one training step vs the reference (fp32 CPU).
    The oracle runs in fp64, so the residual is the reference's own fp32 rounding (App. E5)."""
    meta, a = load_golden(name)
    sd = _state(meta)
    cfg = meta["cfg"]
    x, y = orc.make_batch(meta["seed"] + 1, meta["B"], meta["C_in"], meta["img"], meta["num_classes"], dtype=torch.float64)
    ch = meta["mapper"][meta["chunk"]]
    with torch.no_grad():
        tok, _ = orc.tokenise(sd, x, cfg, ch, list(range(len(ch))))
        r = min(tok.shape[1], 40)
        assert np.abs(tok[0, :r].numpy() - a["tokens_row0"]).max() < 2e-6
        assert np.abs(tok[-1, -3:].numpy() - a["tokens_last"]).max() < 2e-6
        D, depth, heads = orc.MODEL_SIZES[cfg["pretrained_model_name"]]
        z = orc.block_forward(sd, "feature_extractor.blocks.0.", tok, heads)
        assert np.abs(z[0, :r].numpy() - a["block0_row0"]).max() < 5e-6
    loss, main, extra, logits = orc.train_loss(sd, x, y, cfg, ch, list(range(len(ch))))
    loss.backward()
    assert np.abs(logits.detach().numpy() - a["logits"]).max() < 2e-5
    assert abs(extra.item() - float(a["extra"])) < 1e-6 * max(1, abs(float(a["extra"]))) + 2e-8
    assert abs(loss.item() - float(a["loss"])) < 5e-6
    _check_grads(sd, a)


def test_hcs_goldens():
    meta, a = load_golden("hcs")
    cfg = dict(meta["cfg"])
    mapper = meta["mapper"]["train"]
    x, y = orc.make_batch(42, 3, 6, 32, 7, dtype=torch.float64)
    for k, d in enumerate(meta["draws"]):
        sd = _state(meta)
        rng = random.Random(d["pyseed"])
        torch.manual_seed(d["tseed"])
        E = sd["feature_extractor.patch_embed.channel_embed.weight"].detach().float()
        picked, idx = orc.hcs_sample(E, mapper, d["mode"], d["temp"], rng)
        assert picked == a[f"d{k}_picked"].tolist()
        loss, main, extra, logits = orc.train_loss(sd, x, y, cfg, picked, idx)
        loss.backward()
        assert np.abs(logits.detach().numpy() - a[f"d{k}_logits"]).max() < 2e-5
        assert abs(loss.item() - float(a[f"d{k}_loss"])) < 5e-6
        g = sd["feature_extractor.patch_embed.channel_embed.weight"].grad.numpy()
        assert np.abs(g - a[f"d{k}_gchan"]).max() < 2e-4 * np.abs(a[f"d{k}_gchan"]).max()
        gp = sd["feature_extractor.patch_embed.proj.weight"].grad.norm().item()
        assert abs(gp - float(a[f"d{k}_gnorm_proj"])) < 2e-4 * gp


def test_chammi_goldens():
    meta, a = load_golden("chammi")
    sd = _state(meta)
    cfg = meta["cfg"]
    for chunk in ["Allen", "HPA", "CP"]:
        ch = meta["mapper"][chunk]
        x, y = orc.make_batch(meta["seed"] + len(ch), 2, len(ch), meta["img"], meta["num_classes"], dtype=torch.float64)
        loss, main, extra, feat = orc.chammi_loss(sd, x, y, cfg, ch, list(range(len(ch))))
        loss.backward()
        assert np.abs(feat.detach().numpy() - a[f"{chunk}_feat"]).max() < 2e-5
        assert abs(extra.item() - float(a[f"{chunk}_extra"])) < 2e-6
        assert abs(loss.item() - float(a[f"{chunk}_loss"])) < 1e-5
    _check_grads(sd, a)


def test_eval_newchannel_goldens():
    meta, a = load_golden("eval_newch")
    sd = _state(meta)
    cfg, mapper = meta["cfg"], meta["mapper"]
    x, _ = orc.make_batch(62, 3, 5, 32, 9, dtype=torch.float64)
    E = sd["feature_extractor.patch_embed.channel_embed.weight"]
    with torch.no_grad():
        for init in ["zero", "avg_2", "avg_3", "replicate", "avg_2_not_in_chunk", "avg_3_not_in_chunk", "random"]:
            rows = orc.eval_channel_embed(E, mapper, "test", "train", init)
            out, _ = orc.forward(sd, x, cfg, mapper["test"], list(range(5)), channel_embed_rows=rows)
            assert np.abs(out.numpy() - a["test_" + init]).max() < 2e-5, init
        out, _ = orc.forward(sd, x, cfg, mapper["valid"], list(range(5)))
        assert np.abs(out.numpy() - a["valid_none"]).max() < 2e-5


def test_curve_so2sat_golden():
    """100 optimiser steps (fp32, as the reference) reproduce the reference's loss curve."""
    meta, a = load_golden("curve100_so2sat_s")
    torch.set_num_threads(8)
    steps = 30  # the full 100 are replayed on the GPU path; 30 here keeps the CPU suite short
    shapes = orc.state_shapes(meta["cfg"], meta["n_channels"], meta["img"], meta["num_classes"])
    sd = orc.make_state(shapes, meta["seed"], dtype=torch.float32)
    names = [k for k in sd if k != "proxies"]
    for k in names:
        sd[k].requires_grad_(True)
    m = {k: torch.zeros_like(sd[k]) for k in names}
    v = {k: torch.zeros_like(sd[k]) for k in names}
    batches = [orc.make_batch(meta["seed"] + 100 + i, meta["B"], meta["n_channels"], meta["img"], meta["num_classes"]) for i in range(meta["n_batches"])]
    ch = meta["mapper"]["train"]
    for s in range(steps):
        x, y = batches[s % meta["n_batches"]]
        for k in names:
            sd[k].grad = None
        loss, main, extra, _ = orc.train_loss(sd, x, y, meta["cfg"], ch, list(range(len(ch))))
        loss.backward()
        with torch.no_grad():
            for k in names:
                orc.adamw_step(sd[k], sd[k].grad, m[k], v[k], s + 1, meta["lr"], 0.9, 0.999, meta["eps"], meta["wd"])
        assert abs(loss.item() - a["losses"][s, 0]) < 2e-4, (s, loss.item(), a["losses"][s, 0])


def test_resolution_change_goldens():
    """Other input resolution than img_size: real grid resample forward and the exact adjoint backward."""
    meta, a = load_golden("resolution")
    for img_in in (48, 24):
        sd = _state(meta)
        x, y = orc.make_batch(96 + img_in, 2, 4, img_in, 6, dtype=torch.float64)
        loss, main, extra, logits = orc.train_loss(sd, x, y, meta["cfg"], [0, 1, 2, 3], [0, 1, 2, 3])
        loss.backward()
        assert np.abs(logits.detach().numpy() - a[f"logits_{img_in}"]).max() < 2e-5
        assert abs(loss.item() - float(a[f"loss_{img_in}"])) < 5e-6
        g = sd["feature_extractor.pos_embed"].grad.numpy()
        ref = a[f"gpos_{img_in}"]
        assert np.abs(g - ref).max() < 2e-4 * np.abs(ref).max(), img_in


def test_token_drop_goldens():
    """dropout_tokens_hcs variants: same python-RNG draws as the reference, same logits / loss / gradients."""
    meta, a = load_golden("tokendrop")
    x, y = orc.make_batch(98, 2, 5, 32, 6, dtype=torch.float64)
    for k, d in enumerate(meta["draws"]):
        sd = _state(meta)
        keep = orc.token_keep(d["mode"], 5, 16, random.Random(d["pyseed"]))
        assert len(keep) == d["n_keep"] and keep[0] == 0
        loss, main, extra, logits = orc.train_loss(sd, x, y, meta["cfg"], [0, 1, 2, 3, 4], [0, 1, 2, 3, 4], keep=keep)
        loss.backward()
        assert np.abs(logits.detach().numpy() - a[f"d{k}_logits"]).max() < 2e-5, d
        assert abs(loss.item() - float(a[f"d{k}_loss"])) < 5e-6
        g = sd["feature_extractor.pos_embed"].grad.numpy()
        assert np.abs(g - a[f"d{k}_gpos"]).max() < 2e-4 * np.abs(a[f"d{k}_gpos"]).max()


def test_chammi_hcs_goldens():
    """BASELINE config 3 as specified: CHAMMI chunks WITH enable_sample=True — HCS on the non-identity mapper (global ids vs
    positions), two rounds, gradients accumulating over a round's chunks."""
    meta, a = load_golden("chammi_hcs")
    cfg = dict(meta["cfg"])
    k = 0
    for rnd in range(2):
        sd = _state(meta)
        for d in meta["draws"][3 * rnd:3 * rnd + 3]:
            ch = meta["mapper"][d["chunk"]]
            rng = random.Random(d["pyseed"])
            torch.manual_seed(d["tseed"])
            rows = sd["feature_extractor.patch_embed.channel_embed.weight"].detach().float()[ch]
            picked, idx = orc.hcs_sample(rows, ch, d["mode"], d["temp"], rng)
            assert picked == a[f"d{k}_picked"].tolist()
            assert idx == [ch.index(c) for c in picked] and (d["chunk"] == "Allen" or picked != idx)  # ids != positions off Allen
            x, y = orc.make_batch(d["batch_seed"], 2, len(ch), meta["img"], meta["num_classes"], dtype=torch.float64)
            loss, main, extra, feat = orc.chammi_loss(sd, x, y, cfg, picked, idx)
            loss.backward()
            assert np.abs(feat.detach().numpy() - a[f"d{k}_feat"]).max() < 2e-5
            assert abs(extra.item() - float(a[f"d{k}_extra"])) < 2e-6
            assert abs(loss.item() - float(a[f"d{k}_loss"])) < 1e-5
            k += 1
        _check_grads(sd, {kk[len(f"r{rnd}/"):]: v for kk, v in a.items() if kk.startswith(f"r{rnd}/")})


def test_resolution_quirk_golden():
    """interpolate_pos_encoding's early-out with several channels (4 ch x 4 patches = the model's 16 grid positions): the raw
    table is added token by token across the channels."""
    meta, a = load_golden("resolution_quirk")
    sd = _state(meta)
    x, y = orc.make_batch(196, 2, 4, meta["img_in"], 6, dtype=torch.float64)
    loss, main, extra, logits = orc.train_loss(sd, x, y, meta["cfg"], [0, 1, 2, 3], [0, 1, 2, 3])
    loss.backward()
    assert np.abs(logits.detach().numpy() - a["logits"]).max() < 2e-5
    assert abs(loss.item() - float(a["loss"])) < 5e-6
    for key, name in (("gpos", "feature_extractor.pos_embed"), ("gchan", "feature_extractor.patch_embed.channel_embed.weight"),
                      ("gcls", "feature_extractor.cls_token")):
        g = sd[name].grad.numpy()
        assert np.abs(g - a[key]).max() < 2e-4 * np.abs(a[key]).max(), key


def test_hcs_proj_golden():
    """hcs_sampling=lowest_cosine_prob_proj: the projected-input cosine matrix and the subsets the reference drew from it."""
    meta, a = load_golden("hcs_proj")
    sd = _state(meta, dtype=torch.float32)
    x, y = orc.make_batch(44, 3, 6, 32, 7)
    with torch.no_grad():
        cos = orc.proj_cosine(sd, x, meta["cfg"]["patch_size"])
    assert np.abs(cos.numpy() - a["cos"]).max() < 2e-6
    for k, d in enumerate(meta["draws"]):
        sd64 = _state(meta)
        rng = random.Random(d["pyseed"])
        torch.manual_seed(d["tseed"])
        picked, idx = orc.hcs_sample(None, meta["mapper"]["train"], "lowest_cosine_prob_proj", meta["cfg"]["hcs_sampling_temp"], rng,
                                     proj_cos=torch.from_numpy(a["cos"]))
        assert picked == a[f"d{k}_picked"].tolist()
        loss, main, extra, logits = orc.train_loss(sd64, x.double(), y, meta["cfg"], picked, idx)
        assert np.abs(logits.detach().numpy() - a[f"d{k}_logits"]).max() < 2e-5
        assert abs(loss.item() - float(a[f"d{k}_loss"])) < 5e-6


def test_headline_b16_golden_forward():
    """jumpcp_s_b16 (the headline architecture at batch 16, the fixture behind the model-level multi-round GEMM test):
    oracle forward on the first 2 images of the batch against the reference's logits (per-image independence)."""
    meta, a = load_golden("jumpcp_s_b16")
    sd = _state(meta, dtype=torch.float32)
    x, y = orc.make_batch(meta["seed"] + 1, meta["B"], 8, 224, 161)
    with torch.no_grad():
        logits, extra = orc.forward(sd, x[:2], meta["cfg"], list(range(8)), list(range(8)))
    assert np.abs(logits.numpy() - a["logits"][:2]).max() < 5e-5


def test_no_channel_embed_golden():
    """use_channelvit_channels=False (models/dichavit.py:83-95, 121, 409; tests/golden/nochannel_embed.npz from the real reference): the state
    has no channel_embed entry, the tokens carry no channel offset; logits, losses and every gradient of one train step."""
    meta, a = load_golden("nochannel_embed")
    assert not any("channel_embed" in k for k in meta["state_keys"])
    sd = _state(meta)
    x, y = orc.make_batch(meta["seed"] + 1, meta["B"], meta["C_in"], meta["img"], meta["num_classes"], dtype=torch.float64)
    ch = meta["mapper"][meta["chunk"]]
    loss, main, extra, logits = orc.train_loss(sd, x, y, meta["cfg"], ch, list(range(len(ch))))
    loss.backward()
    assert np.abs(logits.detach().numpy() - a["logits"]).max() < 2e-5
    assert abs(extra.item() - float(a["extra"])) < 1e-6 * max(1, abs(float(a["extra"]))) + 2e-8
    assert abs(loss.item() - float(a["loss"])) < 5e-6
    _check_grads(sd, a)


def test_drop_path_golden():
    """drop_path_rate = 0.4 (stochastic depth, vit.py:37-56, 397-398; tests/golden/drop_path.npz from the real reference with its torch.rand
    draws recorded): the oracle with the same keep masks reproduces logits, losses and every gradient."""
    meta, a = load_golden("drop_path")
    sd = _state(meta)
    x, y = orc.make_batch(meta["seed"] + 1, meta["B"], meta["C_in"], meta["img"], meta["num_classes"], dtype=torch.float64)
    ch = meta["mapper"][meta["chunk"]]
    masks = [torch.from_numpy(m) for m in a["keep"]]
    assert len(masks) == 22 and set(np.unique(a["keep"]).tolist()) <= {0.0, 1.0} and 0 < a["keep"].mean() < 1
    loss, main, extra, logits = orc.train_loss(sd, x, y, meta["cfg"], ch, list(range(len(ch))), drop_masks=masks)
    loss.backward()
    assert np.abs(logits.detach().numpy() - a["logits"]).max() < 2e-5
    assert abs(loss.item() - float(a["loss"])) < 5e-6
    _check_grads(sd, a)
