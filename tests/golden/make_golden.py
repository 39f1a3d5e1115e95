"""Generates the golden fixtures in this directory by running the REAL reference
(/root/reference, read-only) on CPU fp32.  Runs only in the build container — the GPU box
has no /root/reference and never needs it: tests read the committed ``*.npz`` files.

    python tests/golden/make_golden.py [case ...]      (no args = all cases)

Nothing of the reference (source, bytecode, pickles) is written anywhere: fixtures hold
only configs, seeds, and output numbers.  Parameters and inputs are regenerated in the
tests from the seeds via oracle.dichavit_oracle.make_state / make_batch (numpy legacy
RandomState: version-stable).

Import shims: omegaconf and h5py are absent here and only touched at import
time; ``models/__init__.py`` imports timm-based baselines, so ``models`` is pre-registered
as a bare namespace package.
"""
import json
import math
import os
import random
import sys
import time
import types

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import dichavit_oracle as orc  # noqa: E402

REF = "/root/reference"


def load_reference():
    sys.path.insert(0, REF)
    om = types.ModuleType("omegaconf"); om.MISSING = "???"; sys.modules["omegaconf"] = om
    sys.modules["h5py"] = types.ModuleType("h5py")
    pkg = types.ModuleType("models"); pkg.__path__ = [REF + "/models"]; sys.modules["models"] = pkg
    import warnings
    warnings.filterwarnings("ignore")
    from models.dichavit import dichavit
    from models import loss_fn
    return dichavit, loss_fn


class Cfg(dict):
    __getattr__ = dict.get


def base_cfg(**kw):
    c = dict(name="dichavit", pretrained_model_name="small", patch_size=16, temperature=0.07, learnable_temp=False,
             enable_sample=False, use_channelvit_channels=True, orthogonal_channel_emb_init=True,
             dropout_tokens_hcs="none", freeze_channel_emb=False, block_type="block", hcs_sampling="none",
             hcs_sampling_temp=0.1, proxy_loss_lambda=0.001, ortho_loss_v1_lambda=0.001, drop_path_rate=0.0,
             gamma_s=1.0, gamma_d=4.0, reverse_pos_pairs=True, use_square=False, new_channel_inits=["zero"])
    c.update(kw)
    return c


def build(dichavit, cfg, mapper, n_channels, img, num_classes, seed):
    import contextlib, io
    full = Cfg(cfg, in_channel_names=[f"c{i}" for i in range(n_channels)], img_size=[img], num_classes=num_classes)
    with contextlib.redirect_stdout(io.StringIO()):
        m = dichavit(full, mapper=mapper)
    shapes = orc.state_shapes(cfg, n_channels, img, num_classes, chammi="Allen" in mapper)
    st = orc.make_state(shapes, seed)
    sd = m.state_dict()
    want = set(sd.keys()) - {"adaptive_interface.0"}
    assert want == set(st.keys()), (want ^ set(st.keys()))
    for k, v in sd.items():
        assert k == "adaptive_interface.0" or tuple(v.shape) == tuple(st[k].shape), k
    missing = m.load_state_dict({**st, "adaptive_interface.0": st["proxies"]}, strict=True)
    return m, sorted(sd.keys())


def sample_idx(n, k=64):
    return np.unique(np.linspace(0, n - 1, min(k, n)).astype(np.int64))


def grad_summary(model):
    out = {}
    for name, p in model.named_parameters():
        if name.startswith("adaptive_interface"):
            continue
        if p.grad is None:
            out["gnone/" + name] = np.zeros(0)
            continue
        g = p.grad.detach().double().flatten().numpy()
        out["gnorm/" + name] = np.array(np.linalg.norm(g))
        out["gsamp/" + name] = g[sample_idx(g.size)].astype(np.float64)
    return out


def save(name, meta, arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, meta=np.array(json.dumps(meta)), **arrays)
    print(f"wrote {path}  ({os.path.getsize(path)/1024:.1f} KiB)")


# ------------------------------------------------------------------------------------
def case_loss_fns(dichavit, loss_fn):
    """Known answers for the two regularisers, straight from models/loss_fn.py (fp64)."""
    arrays, meta = {}, {"cases": []}
    B, C, n, D = 2, 3, 4, 8
    i = torch.arange(B * C * n * D, dtype=torch.float64).reshape(B, C * n, D)
    f = torch.sin(0.37 * i) + 0.25 * torch.cos((0.11 * i * i) % 7.0)
    labels = torch.arange(C).repeat_interleave(n)
    for k, (gs, gd, rev, sq) in enumerate([(1.0, 4.0, True, False), (0.5, 2.0, True, False),
                                           (1.0, 0.5, False, False), (1.0, 4.0, True, True)]):
        arrays[f"e1_{k}"] = np.array(loss_fn.ortho_proj_loss_fn_v2(f, labels, gs, gd, rev, sq).item())
        meta["cases"].append(dict(kind="e1", k=k, gs=gs, gd=gd, rev=rev, sq=sq))
    j = torch.arange(40, dtype=torch.float64).reshape(5, 8)
    prox, emb, sc = torch.cos(0.3 * j), torch.sin(0.2 * j + 1.0), math.sqrt(1 / 0.07)
    arrays["e2_eye"] = np.array(loss_fn.proxy_loss(prox, emb, torch.eye(5, dtype=torch.float64), sc).item())
    arrays["e2_int"] = np.array(loss_fn.proxy_loss(prox, emb, torch.arange(5), sc).item())
    # random features incl. gradient, several (C, n) incl. C == 1
    rs = np.random.RandomState(7)
    for k, (B, C, n, D, gs, gd, rev, sq) in enumerate([(2, 3, 16, 32, 1.0, 4.0, True, False),
                                                      (3, 1, 9, 16, 0.5, 2.0, True, False),
                                                      (2, 5, 4, 24, 1.0, 0.5, False, True),
                                                      (1, 8, 49, 64, 1.0, 4.0, True, False)]):
        feat = torch.from_numpy(rs.standard_normal((B, C * n, D))).requires_grad_(True)
        labels = torch.arange(C).repeat_interleave(n)
        val = loss_fn.ortho_proj_loss_fn_v2(feat, labels, gs, gd, rev, sq)
        val.backward()
        arrays[f"r_{k}_val"] = np.array(val.item())
        arrays[f"r_{k}_grad"] = feat.grad.numpy()
        arrays[f"r_{k}_feat"] = feat.detach().numpy()
        meta["cases"].append(dict(kind="rand", k=k, B=B, C=C, n=n, D=D, gs=gs, gd=gd, rev=rev, sq=sq))
    save("loss_fns", meta, arrays)


def _train_case(dichavit, name, cfg, mapper, chunk, n_channels, C_in, img, K, B, seed, stages=True):
    model, keys = build(dichavit, cfg, mapper, n_channels, img, K, seed)
    model.train()
    x, y = orc.make_batch(seed + 1, B, C_in, img, K)
    arrays = {}
    if stages:
        fe = model.feature_extractor
        with torch.no_grad():
            tok, _ = fe.prepare_tokens(x, chunk, None, None, {})
            arrays["tokens_row0"] = tok[0, : min(tok.shape[1], 40)].numpy()
            arrays["tokens_last"] = tok[-1, -3:].numpy()
            z = fe.blocks[0](tok)
            arrays["block0_row0"] = z[0, : min(z.shape[1], 40)].numpy()
    t = time.time()
    out, extra = model(x, chunk, None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    main = torch.nn.CrossEntropyLoss()(out, y)
    loss = main + 1.0 * extra
    loss.backward()
    print(f"  {name}: fwd+bwd {time.time()-t:.1f}s loss={loss.item():.6f} extra={extra.item():.6e}")
    arrays.update(logits=out.detach().numpy(), extra=np.array(extra.item()), main=np.array(main.item()),
                  loss=np.array(loss.item()))
    arrays.update(grad_summary(model))
    assert model.proxies.grad is None
    meta = dict(cfg=cfg, mapper=mapper, chunk=chunk, n_channels=n_channels, C_in=C_in, img=img, num_classes=K, B=B,
                seed=seed, state_keys=keys)
    save(name, meta, arrays)


def case_tiny(dichavit, loss_fn):
    _train_case(dichavit, "tiny_e2e", base_cfg(pretrained_model_name="tiny", patch_size=8), {"train": [0, 1, 2]},
                "train", 3, 3, 32, 5, 2, 11)


def case_so2sat(dichavit, loss_fn):
    """BASELINE config 1 shape: So2Sat S, 18ch 32x32 P8, 17 classes (train_scripts.sh:8 lambdas)."""
    cfg = base_cfg(patch_size=8, ortho_loss_v1_lambda=0.1, gamma_s=0.5)
    _train_case(dichavit, "so2sat_s", cfg, {"train": list(range(18))}, "train", 18, 18, 32, 17, 4, 21)


def case_jumpcp(dichavit, loss_fn):
    """BASELINE config 2 shape: JUMP-CP S, 8ch 224x224 P16, 161 classes, bs 2."""
    _train_case(dichavit, "jumpcp_s", base_cfg(), {"train": list(range(8))}, "train", 8, 8, 224, 161, 2, 31)


def case_resume(dichavit, loss_fn):
    """Checkpoint/resume + gradient clipping in the reference's own flow (trainer.py:1001-1006, 1292-1328): 3 steps of
    torch AdamW with clip_grad_norm_(0.5) on the tiny config, a checkpoint in the trainer's layout taken there
    (model_params / optimizer_params as arrays), then 3 more steps.  The fixture holds the checkpoint's tensors, the
    reference's parameter order, the clipped-step total norms, all 6 losses and parameter norms at the end."""
    cfg = base_cfg(pretrained_model_name="tiny", patch_size=8)
    mapper = {"train": [0, 1, 2]}
    model, keys = build(dichavit, cfg, mapper, 3, 32, 5, 51)
    model.train()
    params = [p for p in model.parameters() if p.requires_grad]
    names = [n for n, p in model.named_parameters() if p.requires_grad]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=0.04, betas=(0.9, 0.999), eps=1e-8)
    clip = 0.5
    batches = [orc.make_batch(151 + i, 4, 3, 32, 5) for i in range(3)]
    losses, norms = [], []
    arrays = {}

    def one(s):
        x, y = batches[s % 3]
        opt.zero_grad()
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        tn = torch.nn.utils.clip_grad_norm_(model.parameters(), clip)
        opt.step()
        losses.append(loss.item()); norms.append(float(tn))

    for s in range(3):
        one(s)
    osd = opt.state_dict()
    # the checkpoint's tensors themselves are NOT stored (66 MB): the test rebuilds the same trajectory from the seed,
    # passes through a save/load in the trainer's layout at this point and must land on the same remaining losses
    ck_layout = dict(opt_state_keys=sorted(osd["state"][min(osd["state"])].keys()), n_opt_state=len(osd["state"]),
                     opt_state_ids=sorted(int(i) for i in osd["state"]),
                     opt_param_ids=[int(i) for i in osd["param_groups"][0]["params"]],
                     model_keys=list(model.state_dict().keys()))
    for idx in (1, 2, 5, 20, 100):
        if idx in osd["state"]:
            arrays[f"ckpt_opt_norm/{idx}"] = np.array([osd["state"][idx]["exp_avg"].double().norm().item(),
                                                       osd["state"][idx]["exp_avg_sq"].double().norm().item()])
    for s in range(3, 6):
        one(s)
    for n_, p_ in model.named_parameters():
        if not n_.startswith("adaptive_interface"):
            arrays["final_norm/" + n_] = np.array(p_.detach().double().norm().item())
    save("resume", dict(cfg=cfg, mapper=mapper, n_channels=3, img=32, num_classes=5, B=4, seed=51, lr=1e-3, wd=0.04, clip=clip,
                        param_order=names, ck_layout=ck_layout,
                        opt_group={k: v for k, v in osd["param_groups"][0].items() if k in ("lr", "betas", "eps", "weight_decay")}),
         dict(losses=np.array(losses), total_norms=np.array(norms), **arrays))


def case_hcs(dichavit, loss_fn):
    """HCS sampling (dichavit.py:127-216): seeds recorded, resulting subset recorded."""
    cfg = base_cfg(patch_size=8, enable_sample=True, hcs_sampling="lowest_cosine_prob", hcs_sampling_temp=0.1)
    mapper = {"train": list(range(6))}
    arrays, meta = {}, dict(cfg=cfg, mapper=mapper, n_channels=6, img=32, num_classes=7, B=3, seed=41, draws=[])
    model, keys = build(dichavit, cfg, mapper, 6, 32, 7, 41)
    model.train()
    x, y = orc.make_batch(42, 3, 6, 32, 7)
    for k, (pyseed, tseed, mode, temp) in enumerate([(1, 5, "lowest_cosine_prob", 0.1), (2, 6, "lowest_cosine_prob", 1000.0),
                                                      (3, 7, "lowest_cosine", 0.1), (4, 8, "highest_cosine", 0.1),
                                                      (5, 9, "none", 0.1), (9, 10, "lowest_cosine_prob", 0.01)]):
        model.cfg["hcs_sampling"] = mode
        model.cfg["hcs_sampling_temp"] = temp
        pe = model.feature_extractor.patch_embed
        pe.counter.clear()
        random.seed(pyseed)
        torch.manual_seed(tseed)
        model.zero_grad()
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        counter = dict(pe.counter)
        # the sampled order is not exposed; recover it by replaying the documented draw order
        rng = random.Random(pyseed)
        torch.manual_seed(tseed)
        picked, idx = orc.hcs_sample(pe.channel_embed.weight.detach(), mapper["train"], mode, temp, rng)
        assert sorted(counter.keys()) == sorted(picked) or mode == "none", (counter, picked)
        arrays[f"d{k}_logits"] = out.detach().numpy()
        arrays[f"d{k}_extra"] = np.array(extra.item())
        arrays[f"d{k}_loss"] = np.array(loss.item())
        arrays[f"d{k}_picked"] = np.array(picked)
        arrays[f"d{k}_gnorm_proj"] = np.array(pe.proj.weight.grad.norm().item())
        arrays[f"d{k}_gnorm_chan"] = np.array(pe.channel_embed.weight.grad.norm().item())
        arrays[f"d{k}_gchan"] = pe.channel_embed.weight.grad.numpy().copy()
        meta["draws"].append(dict(pyseed=pyseed, tseed=tseed, mode=mode, temp=temp, counter={int(a): int(b) for a, b in counter.items()}))
        print(f"  hcs draw {k}: mode={mode} picked={picked} loss={loss.item():.6f}")
    save("hcs", meta, arrays)


def case_chammi(dichavit, loss_fn):
    """CHAMMI: 12-row channel_embed, non-identity mapper, features out, proxy main loss
    (trainer.py:130-131, 912-914; dichavit.py:797-801).  3/4/5-channel chunks -> variable N."""
    cfg = base_cfg(patch_size=16, proxy_loss_lambda=0.1, ortho_loss_v1_lambda=1.0, gamma_s=0.5, gamma_d=2.0)
    mapper = {"Allen": [0, 1, 2], "HPA": [3, 4, 5, 6], "CP": [7, 8, 9, 10, 11]}
    K, img, seed = 14, 64, 51
    model, keys = build(dichavit, cfg, mapper, 12, img, K, seed)
    model.train()
    arrays = {}
    scale = model.scale
    for chunk in ["Allen", "HPA", "CP"]:
        C = len(mapper[chunk])
        x, y = orc.make_batch(seed + C, 2, C, img, K)
        feat, extra = model(x, chunk, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = loss_fn.proxy_loss(model.proxies, feat, y, scale) + 1.0 * extra
        loss.backward()  # grads accumulate over the three chunks, one optimizer step after (trainer.py:921-935)
        arrays[f"{chunk}_feat"] = feat.detach().numpy()
        arrays[f"{chunk}_extra"] = np.array(extra.item())
        arrays[f"{chunk}_loss"] = np.array(loss.item())
        print(f"  chammi {chunk}: loss={loss.item():.6f}")
    arrays.update(grad_summary(model))
    save("chammi", dict(cfg=cfg, mapper=mapper, n_channels=12, img=img, num_classes=K, B=2, seed=seed, state_keys=keys), arrays)


def case_eval(dichavit, loss_fn):
    """Eval path: bare tensor out; leave-one-out channel synthesis (dichavit.py:219-374)."""
    cfg = base_cfg(patch_size=8)
    mapper = {"train": [0, 1, 2, 3, 4], "test": [0, 1, 5, 3, 6], "valid": [0, 1, 2, 3, 4]}
    model, keys = build(dichavit, cfg, mapper, 7, 32, 9, 61)
    model.eval()
    x, _ = orc.make_batch(62, 3, 5, 32, 9)
    arrays = {}
    with torch.inference_mode():
        for init in ["zero", "avg_2", "avg_3", "replicate", "avg_2_not_in_chunk", "avg_3_not_in_chunk", "random"]:
            out = model(x, "test", "train", init_first_layer=None, new_channel_init=init)
            assert isinstance(out, torch.Tensor)
            arrays["test_" + init] = out.numpy()
        arrays["valid_none"] = model(x, "valid", None, init_first_layer=None, new_channel_init=None).numpy()
    save("eval_newch", dict(cfg=cfg, mapper=mapper, n_channels=7, img=32, num_classes=9, B=3, seed=61), arrays)


def case_resolution(dichavit, loss_fn):
    """Input resolution different from the model's img_size: the positional grid is really resampled
    (4x4 -> 6x6 and 4x4 -> 3x3; dichavit.py:536-546) and — in training — its gradient takes the true bicubic adjoint
    (output size != input size, so ATen's same-size early-out does not apply)."""
    cfg = base_cfg(patch_size=8)
    mapper = {"train": [0, 1, 2, 3]}
    arrays = {}
    for img_in in (48, 24):
        model, keys = build(dichavit, cfg, mapper, 4, 32, 6, 95)
        model.train()
        x, y = orc.make_batch(96 + img_in, 2, 4, img_in, 6)
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        arrays[f"logits_{img_in}"] = out.detach().numpy()
        arrays[f"loss_{img_in}"] = np.array(loss.item())
        arrays[f"gpos_{img_in}"] = model.feature_extractor.pos_embed.grad.numpy().copy()
        arrays[f"gnorm_proj_{img_in}"] = np.array(model.feature_extractor.patch_embed.proj.weight.grad.norm().item())
        model.eval()
        with torch.inference_mode():
            arrays[f"eval_{img_in}"] = model(x, "train", None, new_channel_init=None).numpy()
        print(f"  resolution {img_in}: loss={loss.item():.6f}")
    save("resolution", dict(cfg=cfg, mapper=mapper, n_channels=4, img=32, num_classes=6, B=2, seed=95), arrays)


def case_tokendrop(dichavit, loss_fn):
    """dropout_tokens_hcs variants (dichavit.py:568-627) with the python RNG seeded."""
    mapper = {"train": [0, 1, 2, 3, 4]}
    arrays, meta = {}, dict(mapper=mapper, n_channels=5, img=32, num_classes=6, B=2, seed=97, draws=[])
    x, y = orc.make_batch(98, 2, 5, 32, 6)
    for k, (mode, pyseed) in enumerate([("random", 1), ("channel", 2), ("channel_random50", 3), ("token_random50", 4), ("channel", 7)]):
        cfg = base_cfg(patch_size=8, dropout_tokens_hcs=mode)
        meta["cfg"] = base_cfg(patch_size=8)
        model, keys = build(dichavit, cfg, mapper, 5, 32, 6, 97)
        model.train()
        random.seed(pyseed)
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        arrays[f"d{k}_logits"] = out.detach().numpy()
        arrays[f"d{k}_loss"] = np.array(loss.item())
        arrays[f"d{k}_gnorm_proj"] = np.array(model.feature_extractor.patch_embed.proj.weight.grad.norm().item())
        arrays[f"d{k}_gpos"] = model.feature_extractor.pos_embed.grad.numpy().copy()
        keep = orc.token_keep(mode, 5, 16, random.Random(pyseed))
        meta["draws"].append(dict(mode=mode, pyseed=pyseed, n_keep=len(keep)))
        print(f"  tokendrop {mode}: keep {len(keep)} of 81, loss={loss.item():.6f}")
    save("tokendrop", meta, arrays)


def _curve(dichavit, name, cfg, n_channels, img, K, B, seed, steps, n_batches, lr=4.9e-5, wd=0.04):
    """`steps` training steps of trainer.train_one_batch_regular's body (trainer.py:963-1006):
    zero_grad, forward, CE + extra, backward, AdamW step.  torch.optim.AdamW stands in for timm's
    AdamW (same update rule); lr is the first-epoch warm-up value of the JUMP-CP
    script: 1e-5 + (4e-4 - 1e-5)/10 (train_scripts.sh:5, configs/scheduler/cosine.yaml)."""
    mapper = {"train": list(range(n_channels))}
    model, keys = build(dichavit, cfg, mapper, n_channels, img, K, seed)
    model.train()
    opt = torch.optim.AdamW([p for p in model.parameters() if p.requires_grad], lr=lr, weight_decay=wd,
                            betas=(0.9, 0.999), eps=1e-8)
    batches = [orc.make_batch(seed + 100 + i, B, n_channels, img, K) for i in range(n_batches)]
    losses = np.zeros((steps, 3))
    t0 = time.time()
    for s in range(steps):
        x, y = batches[s % n_batches]
        opt.zero_grad()
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        main = torch.nn.CrossEntropyLoss()(out, y)
        loss = main + 1.0 * extra
        loss.backward()
        opt.step()
        losses[s] = (loss.item(), main.item(), extra.item())
        if s % 10 == 0:
            print(f"  {name} step {s}: loss={loss.item():.6f}  ({time.time()-t0:.0f}s)", flush=True)
    save(name, dict(cfg=cfg, mapper=mapper, n_channels=n_channels, img=img, num_classes=K, B=B, seed=seed, steps=steps,
                    n_batches=n_batches, lr=lr, wd=wd, betas=[0.9, 0.999], eps=1e-8), dict(losses=losses))


def case_curve_so2sat(dichavit, loss_fn):
    cfg = base_cfg(patch_size=8, ortho_loss_v1_lambda=0.1, gamma_s=0.5)
    _curve(dichavit, "curve100_so2sat_s", cfg, 18, 32, 17, 8, 71, 100, 4)


def case_curve_jumpcp(dichavit, loss_fn):
    """headline architecture (S, 8ch 224x224 P16, 161 classes), bs 2, 100 steps — ~8 min on 8 cores."""
    _curve(dichavit, "curve100_jumpcp_s", base_cfg(), 8, 224, 161, 2, 81, 100, 4)


def case_curve_so2sat_distinct(dichavit, loss_fn):
    """So2Sat-shaped model, bs 8, 100 DISTINCT batches (round 4): the repeating-batch curve above memorises its 4 batches (2.78 -> 0.026) and
    turned out to be as trajectory-sensitive as the batch-2 headline curve — one-ulp changes in 0.2 % of the LayerNorm outputs moved its
    largest error from 4.6e-3 to 1.06e-2; this one stays near ln 17 and measures the arithmetic."""
    cfg = base_cfg(patch_size=8, ortho_loss_v1_lambda=0.1, gamma_s=0.5)
    _curve(dichavit, "curve100_so2sat_s_distinct", cfg, 18, 32, 17, 8, 171, 100, 100)


def case_curve_jumpcp_b8(dichavit, loss_fn):
    """headline architecture at bs 8 over 100 DISTINCT batches (no batch is seen twice: nothing to memorise, the
    curve stays near ln(161) and is well conditioned) — the curve that carries the 1e-3 claim; ~25 min on 8 cores."""
    _curve(dichavit, "curve100_jumpcp_s_b8", base_cfg(), 8, 224, 161, 8, 91, 100, 100)


def case_curve_jumpcp_b16(dichavit, loss_fn):
    """headline architecture at bs 16 over 100 DISTINCT batches (round 5): the per-step error is a batch mean, so the spread the
    stochastic-rounding draw causes should fall like 1/sqrt(B) against the bs-8 curve; ~45 min on 6 cores, peak RSS recorded in
    profiles/r05_x1_*."""
    _curve(dichavit, "curve100_jumpcp_s_b16", base_cfg(), 8, 224, 161, 16, 191, 100, 100)


def case_curve_jumpcp_b32(dichavit, loss_fn):
    """the same at bs 32 (half the way to north_star's bs 64 on a log scale; ~50 GB of host memory and ~90 min of reference CPU time)."""
    _curve(dichavit, "curve100_jumpcp_s_b32", base_cfg(), 8, 224, 161, 32, 291, 100, 100)


def case_schedules(dichavit, loss_fn):
    """utils.cosine_scheduler (utils.py:563-574): the weight-decay schedule of trainer.py:217-228."""
    import utils as ref_utils
    arrays, meta = {}, {"cases": []}
    for k, kw in enumerate([dict(base_value=0.04, final_value=0.4, epochs=10, niter_per_ep=7),
                            dict(base_value=0.04, final_value=0.4, epochs=6, niter_per_ep=5, warmup_epochs=2, start_warmup_value=0.01),
                            dict(base_value=1.0, final_value=0.0, epochs=3, niter_per_ep=1)]):
        arrays[f"wd_{k}"] = np.asarray(ref_utils.cosine_scheduler(**kw), dtype=np.float64)
        meta["cases"].append(kw)
    save("schedules", meta, arrays)



def case_chammi_hcs(dichavit, loss_fn):
    """BASELINE config 3 as specified: CHAMMI 12-channel model WITH enable_sample=True (HCS on the NON-identity mapper:
    global channel ids 3..6 / 7..11 differ from the positions inside the chunk's tensor; dichavit.py:120-136,
    203-212, 399-402).  Two rounds over the three chunks with seeded RNGs; gradients accumulate over a round's chunks."""
    cfg = base_cfg(patch_size=16, proxy_loss_lambda=0.1, ortho_loss_v1_lambda=1.0, gamma_s=0.5, gamma_d=2.0,
                   enable_sample=True, hcs_sampling="lowest_cosine_prob", hcs_sampling_temp=0.1)
    mapper = {"Allen": [0, 1, 2], "HPA": [3, 4, 5, 6], "CP": [7, 8, 9, 10, 11]}
    K, img, seed = 14, 64, 53
    model, keys = build(dichavit, cfg, mapper, 12, img, K, seed)
    model.train()
    pe = model.feature_extractor.patch_embed
    arrays, draws = {}, []
    k = 0
    for rnd, (mode, temp) in enumerate([("lowest_cosine_prob", 0.1), ("lowest_cosine", 0.1)]):
        model.cfg["hcs_sampling"], model.cfg["hcs_sampling_temp"] = mode, temp
        model.zero_grad()
        for chunk in ["Allen", "HPA", "CP"]:
            C = len(mapper[chunk])
            x, y = orc.make_batch(seed + 10 * rnd + C, 2, C, img, K)
            pyseed, tseed = 100 + k, 200 + k
            pe.counter.clear()
            random.seed(pyseed)
            torch.manual_seed(tseed)
            feat, extra = model(x, chunk, init_first_layer=None, new_channel_init=None, cur_epoch=0)
            loss = loss_fn.proxy_loss(model.proxies, feat, y, model.scale) + 1.0 * extra
            loss.backward()
            rng = random.Random(pyseed)
            torch.manual_seed(tseed)
            rows = pe.channel_embed.weight.detach()[mapper[chunk]]
            picked, idx = orc.hcs_sample(rows, mapper[chunk], mode, temp, rng)
            assert sorted(pe.counter.keys()) == sorted(set(picked)), (dict(pe.counter), picked)
            arrays[f"d{k}_feat"] = feat.detach().numpy()
            arrays[f"d{k}_extra"] = np.array(extra.item())
            arrays[f"d{k}_loss"] = np.array(loss.item())
            arrays[f"d{k}_picked"] = np.array(picked)
            draws.append(dict(chunk=chunk, rnd=rnd, pyseed=pyseed, tseed=tseed, mode=mode, temp=temp, batch_seed=seed + 10 * rnd + C))
            print(f"  chammi_hcs {k}: {chunk} mode={mode} picked={picked} (positions {idx}) loss={loss.item():.6f}")
            k += 1
        for name_, v in grad_summary(model).items():
            arrays[f"r{rnd}/{name_}"] = v
    save("chammi_hcs", dict(cfg=cfg, mapper=mapper, n_channels=12, img=img, num_classes=K, B=2, seed=seed, draws=draws), arrays)


def case_jumpcp_b16(dichavit, loss_fn):
    """Headline architecture at batch 16: M = 16 x 1569 = 25 104 token rows -> 297 output tiles on a 256-workgroup grid: the
    persistent multi-round GEMM walk and the 256 x 384 kernel (M >= 4096) run inside a MODEL-level golden."""
    _train_case(dichavit, "jumpcp_s_b16", base_cfg(), {"train": list(range(8))}, "train", 8, 8, 224, 161, 16, 35, stages=False)


def case_base64(dichavit, loss_fn):
    """BASELINE config 5 as a whole model: DiChaViT-Base, 64 channels, 224^2 -> N = 12 545 tokens, batch 1, forward only
    (eval mode, bare logits).  The reference materialises 12 x 12 545^2 fp32 attention matrices (7.5 GB per layer)."""
    cfg = base_cfg(pretrained_model_name="base")
    mapper = {"train": list(range(64))}
    model, keys = build(dichavit, cfg, mapper, 64, 224, 161, 111)
    model.eval()
    x, _ = orc.make_batch(112, 1, 64, 224, 161)
    t = time.time()
    with torch.inference_mode():
        out = model(x, "train", None, init_first_layer=None, new_channel_init=None)
    print(f"  base64: forward {time.time()-t:.1f}s logits[:4]={out[0, :4].tolist()}")
    save("base64_fwd", dict(cfg=cfg, mapper=mapper, n_channels=64, img=224, num_classes=161, B=1, seed=111), dict(logits=out.numpy()))


def case_base32_train(dichavit, loss_fn):
    """BASELINE config 5's architecture (DiChaViT-Base, D = 768, 12 heads) as a TRAIN step: 32 channels, 224^2 -> N = 6 273 tokens, batch 1 —
    the largest channel count whose saved attention matrices (12 x 6273^2 fp32 = 1.9 GB per layer, 23 GB for the twelve) fit this container's
    62 GB next to the backward's transients.  Logits, losses and every parameter gradient (norm + samples), as in jumpcp_s_b16."""
    _train_case(dichavit, "base32_train", base_cfg(pretrained_model_name="base"), {"train": list(range(32))}, "train", 32, 32, 224, 161, 1, 131, stages=False)


def case_nochannel_embed(dichavit, loss_fn):
    """use_channelvit_channels=False (models/dichavit.py:83-95, 121, 409): no channel_embed parameter, tokens carry no channel offset.
    The reference leaves `channel_embed` unbound in this mode, so the proxy term must be off (proxy_loss_lambda = 0); the diversity loss
    stays on.  One train step of a So2Sat-shaped model + a step with random channel sampling (enable_sample, hcs_sampling = none)."""
    cfg = base_cfg(use_channelvit_channels=False, proxy_loss_lambda=0.0, patch_size=8)
    _train_case(dichavit, "nochannel_embed", cfg, {"train": list(range(6))}, "train", 6, 6, 32, 9, 3, 141, stages=False)


def case_drop_path(dichavit, loss_fn):
    """drop_path_rate > 0 (stochastic depth; vit.py:37-56, 397-398; per-block rates linspace(0, rate, depth), dichavit.py:475): one train
    step of a So2Sat-shaped model at batch 6, rate 0.4.  The reference draws its keep masks with torch.rand on the model's device; the
    draws are recorded here (torch.rand is wrapped for the duration of the forward) so that the tests can inject the same masks."""
    cfg = base_cfg(drop_path_rate=0.4, patch_size=8)
    mapper = {"train": list(range(6))}
    model, keys = build(dichavit, cfg, mapper, 6, 32, 9, 151)
    model.train()
    x, y = orc.make_batch(152, 6, 6, 32, 9)
    masks = []
    real_rand = torch.rand

    def rec_rand(*a, **kw):
        r = real_rand(*a, **kw)
        masks.append(r.clone())
        return r

    torch.manual_seed(4242)
    torch.rand = rec_rand
    try:
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    finally:
        torch.rand = real_rand
    main = torch.nn.CrossEntropyLoss()(out, y)
    loss = main + 1.0 * extra
    loss.backward()
    dpr = orc.drop_path_rates(cfg)
    rates = [r for r in dpr if r > 0 for _ in range(2)]
    assert len(masks) == len(rates) == 22 and all(m.shape == (6, 1, 1) for m in masks)
    keep = np.stack([np.floor((1.0 - r) + m.reshape(-1).numpy()) for r, m in zip(rates, masks)]).astype(np.float32)  # [22, B] of 0 / 1
    print(f"  drop_path: loss={loss.item():.6f} kept {keep.mean():.3f} of the branches")
    arrays = dict(logits=out.detach().numpy(), extra=np.array(extra.item()), main=np.array(main.item()), loss=np.array(loss.item()), keep=keep)
    arrays.update(grad_summary(model))
    save("drop_path", dict(cfg=cfg, mapper=mapper, chunk="train", n_channels=6, C_in=6, img=32, num_classes=9, B=6, seed=151, state_keys=keys), arrays)


def case_resolution_quirk(dichavit, loss_fn):
    """interpolate_pos_encoding's early-out (dichavit.py:529-530) hit with SEVERAL channels: a 32-px / P8 model (16 grid
    positions) fed 16-px images with 4 channels has 4 x 4 = 16 patch tokens = the model's own count, and H == W: the raw
    pos_embed is added token by token ACROSS the channels (no per-channel tiling, no resampling)."""
    cfg = base_cfg(patch_size=8)
    mapper = {"train": [0, 1, 2, 3]}
    model, keys = build(dichavit, cfg, mapper, 4, 32, 6, 95)
    model.train()
    x, y = orc.make_batch(196, 2, 4, 16, 6)
    out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
    loss = torch.nn.CrossEntropyLoss()(out, y) + extra
    loss.backward()
    fe = model.feature_extractor
    arrays = dict(logits=out.detach().numpy(), loss=np.array(loss.item()), extra=np.array(extra.item()),
                  gpos=fe.pos_embed.grad.numpy().copy(), gchan=fe.patch_embed.channel_embed.weight.grad.numpy().copy(),
                  gnorm_proj=np.array(fe.patch_embed.proj.weight.grad.norm().item()),
                  gcls=fe.cls_token.grad.numpy().copy())
    model.eval()
    with torch.inference_mode():
        arrays["eval"] = model(x, "train", None, new_channel_init=None).numpy()
    print(f"  resolution quirk: loss={loss.item():.6f}")
    save("resolution_quirk", dict(cfg=cfg, mapper=mapper, n_channels=4, img=32, img_in=16, num_classes=6, B=2, seed=95), arrays)


def case_hcs_proj(dichavit, loss_fn):
    """hcs_sampling=lowest_cosine_prob_proj (dichavit.py:156-161): cosine of the PROJECTED input, batch mean.  The fixture
    holds the cosine matrix (computed with the reference's own proj module), the subset the reference drew under the seeds,
    and the step's outputs."""
    from einops import rearrange
    cfg = base_cfg(patch_size=8, enable_sample=True, hcs_sampling="lowest_cosine_prob_proj", hcs_sampling_temp=0.05)
    mapper = {"train": list(range(6))}
    model, keys = build(dichavit, cfg, mapper, 6, 32, 7, 43)
    model.train()
    pe = model.feature_extractor.patch_embed
    x, y = orc.make_batch(44, 3, 6, 32, 7)
    arrays, draws = {}, []
    with torch.no_grad():
        xs = pe.proj(x.unsqueeze(1))
        xs = torch.nn.functional.normalize(rearrange(xs, "b d c h w -> b c (h w d)"), p=2, dim=-1)
        cos = torch.einsum("b c d, b e d -> b c e", xs, xs).mean(dim=0)
    arrays["cos"] = cos.numpy()
    for k, (pyseed, tseed) in enumerate([(11, 21), (12, 22), (13, 23)]):
        pe.counter.clear()
        random.seed(pyseed)
        torch.manual_seed(tseed)
        model.zero_grad()
        out, extra = model(x, "train", None, init_first_layer=None, new_channel_init=None, cur_epoch=0)
        loss = torch.nn.CrossEntropyLoss()(out, y) + extra
        loss.backward()
        # replay the documented draw order with the cosine row above
        rng = random.Random(pyseed)
        torch.manual_seed(tseed)
        kk = rng.randint(1, 6)
        anchor = rng.randint(0, 5)
        prob = torch.softmax((1 - cos[anchor]) / 0.05, dim=-1)
        ind = torch.multinomial(prob, kk, replacement=False).tolist()
        if anchor not in ind:
            ind[-1] = anchor
        picked = [mapper["train"][i] for i in ind]
        assert sorted(pe.counter.keys()) == sorted(picked), (dict(pe.counter), picked)
        arrays[f"d{k}_picked"] = np.array(picked)
        arrays[f"d{k}_logits"] = out.detach().numpy()
        arrays[f"d{k}_loss"] = np.array(loss.item())
        draws.append(dict(pyseed=pyseed, tseed=tseed, anchor=anchor, k=kk))
        print(f"  hcs_proj draw {k}: picked={picked} loss={loss.item():.6f}")
    save("hcs_proj", dict(cfg=cfg, mapper=mapper, n_channels=6, img=32, num_classes=7, B=3, seed=43, draws=draws), arrays)


def case_init_stats(dichavit, loss_fn):
    """Initialisation (dichavit.py:505-516, 60-65, 83-89, 803-805; utils.py:477-517): per-parameter moments and the first
    values of the REAL reference's freshly constructed module under torch.manual_seed(s) on the CPU.  Data only."""
    import contextlib, io
    arrays, meta = {}, dict(torch_version=torch.__version__, variants=[])
    for v, (kw, n_ch, img, K, seed) in enumerate([(dict(), 8, 224, 161, 1234),
                                                  (dict(orthogonal_channel_emb_init=False, proxy_orthogonal_init=True, patch_size=8), 18, 32, 17, 77)]):
        cfg = base_cfg(**kw)
        full = Cfg(cfg, in_channel_names=[f"c{i}" for i in range(n_ch)], img_size=[img], num_classes=K)
        torch.manual_seed(seed)
        with contextlib.redirect_stdout(io.StringIO()):
            m = dichavit(full, mapper={"train": list(range(n_ch))})
        names = []
        for name, p in m.named_parameters():
            if name.startswith("adaptive_interface"):
                continue
            t = p.detach().double().flatten()
            arrays[f"v{v}/{name}"] = np.array([t.numel(), t.mean().item(), t.std(unbiased=False).item() if t.numel() > 1 else 0.0,
                                               t.min().item(), t.max().item(), t.sum().item(), (t * t).sum().item()] + t[:4].tolist()
                                              + [0.0] * max(0, 4 - t.numel()))
            names.append(name)
        meta["variants"].append(dict(cfg=cfg, n_channels=n_ch, img=img, num_classes=K, seed=seed, names=names))
    save("init_stats", meta, arrays)


CASES = dict(tokendrop=case_tokendrop, resolution=case_resolution, schedules=case_schedules, loss_fns=case_loss_fns, tiny=case_tiny, so2sat=case_so2sat, jumpcp=case_jumpcp, hcs=case_hcs,
             chammi=case_chammi, eval=case_eval, curve_so2sat=case_curve_so2sat, curve_jumpcp=case_curve_jumpcp, curve_jumpcp_b8=case_curve_jumpcp_b8, curve_jumpcp_b16=case_curve_jumpcp_b16, curve_jumpcp_b32=case_curve_jumpcp_b32, curve_so2sat_distinct=case_curve_so2sat_distinct, resume=case_resume,
             chammi_hcs=case_chammi_hcs, jumpcp_b16=case_jumpcp_b16, base64=case_base64, resolution_quirk=case_resolution_quirk,
             hcs_proj=case_hcs_proj, init_stats=case_init_stats, base32_train=case_base32_train, nochannel_embed=case_nochannel_embed, drop_path=case_drop_path)

if __name__ == "__main__":
    torch.set_num_threads(int(os.environ.get("GOLDEN_THREADS", "8")))
    torch.manual_seed(0)
    dichavit, loss_fn = load_reference()
    for c in (sys.argv[1:] or list(CASES)):
        print("case", c)
        CASES[c](dichavit, loss_fn)
