"""Checkpoint and evaluation helpers in the reference trainer's formats.

save_checkpoint / load_checkpoint write and read the dict that Trainer._save_model / _load_model use (trainer.py:1292-1328):
``epoch, accuracy, config, optimizer_params, model_params, scheduler_params, scaler_params, datetime`` — so a run can be
resumed by either side.  Model keys may carry DataParallel/DDP's ``module.`` prefix (trainer.py:1313-1318); it is stripped or
added as the target model requires.  ``evaluate`` is eval_regular's loop (trainer.py:385-449): eval mode, inference_mode,
concatenated logits -> top-1 accuracy in percent, summed over ranks when torch.distributed is initialised (the
reference's torchmetrics Accuracy does the same reduction).  ``dump_features`` is the feature-extraction half of eval_morphem70k
(trainer.py:645-690): per chunk, eval-mode forwards with the leave-one-out ``new_channel_init`` -> one ``.npy`` per chunk in the
layout ``morphem.run_benchmark`` reads (the benchmark itself is the reference's, unchanged).  ``eval_subset_channels`` is the
channel-subset sweep of trainer.py:474-545, including its mutation of ``patch_embed.mapper[chunk]``."""
from __future__ import annotations

import datetime
import os
from itertools import combinations
from typing import Dict, Iterable, List, Optional, Sequence

import numpy as np

import torch
import torch.distributed as dist


def _plain(obj):
    """Config containers (DictConfig-likes, dict subclasses) -> plain python so that weights_only loading accepts the file."""
    if isinstance(obj, dict):
        return {str(k): _plain(v) for k, v in obj.items()}
    if isinstance(obj, (list, tuple)):
        return [_plain(v) for v in obj]
    if isinstance(obj, (int, float, str, bool)) or obj is None:
        return obj
    if hasattr(obj, "items"):
        return {str(k): _plain(v) for k, v in obj.items()}
    return str(obj)


def save_checkpoint(path: str, model, optimizer, scheduler=None, epoch: int = 0, accuracy: Optional[float] = None, config=None,
                    scaler=None) -> None:
    state = {
        "epoch": epoch,
        "accuracy": accuracy,
        "config": _plain(config) if config is not None else None,
        "optimizer_params": optimizer.state_dict(),
        "model_params": model.state_dict(),
        "scheduler_params": scheduler.state_dict() if scheduler is not None else None,
        "scaler_params": scaler.state_dict() if scaler is not None else {},
        "datetime": datetime.datetime.now().strftime("%Y-%m-%d %H:%M:%S"),
    }
    torch.save(state, path)


def _match_prefix(params: dict, model) -> dict:
    want_module = next(iter(model.state_dict().keys())).startswith("module.")
    has_module = next(iter(params.keys())).startswith("module.")
    if has_module and not want_module:
        return {k[len("module."):] if k.startswith("module.") else k: v for k, v in params.items()}
    if want_module and not has_module:
        return {"module." + k: v for k, v in params.items()}
    return params


def load_checkpoint(path: str, model, optimizer=None, scheduler=None, map_location=None) -> int:
    """Returns the stored epoch (trainer.py:1325-1328).  Files are read with ``weights_only=True``: a checkpoint whose
    ``config`` entry is a pickled OmegaConf object (what the reference itself writes) is refused by that loader — re-save it
    with a plain-dict config, this function never unpickles arbitrary objects."""
    state = torch.load(path, map_location=map_location, weights_only=True)
    model.load_state_dict(_match_prefix(state["model_params"], model))
    if optimizer is not None and state.get("optimizer_params") is not None:
        optimizer.load_state_dict(state["optimizer_params"])
    if scheduler is not None and state.get("scheduler_params") is not None:
        scheduler.load_state_dict(state["scheduler_params"])
    return int(state.get("epoch", 0))


@torch.inference_mode()
def evaluate(model, batches: Iterable, chunk_name: str, training_chunks: Optional[str] = None, new_channel_init: Optional[str] = None,
             device=None) -> float:
    """Top-1 accuracy in percent over ``batches`` (dicts with "image"/"label" as the reference's loaders yield, or (x, y)
    pairs).  With torch.distributed initialised the correct/total counts are summed over ranks before the ratio."""
    was_training = model.training
    model.eval()
    correct = total = None
    for batch in batches:
        x, y = (batch["image"], batch["label"]) if isinstance(batch, dict) else batch[:2]
        if device is not None:
            x, y = x.to(device), y.to(device)
        out = model(x, chunk_name, training_chunks, init_first_layer=None, new_channel_init=new_channel_init)
        c = (out.argmax(dim=1) == y).sum()
        correct = c if correct is None else correct + c
        total = (total or 0) + y.numel()
    if was_training:
        model.train()
    if correct is None:
        return float("nan")
    counts = torch.stack([correct.to(torch.float64), torch.tensor(float(total), dtype=torch.float64, device=correct.device)])
    if dist.is_available() and dist.is_initialized() and dist.get_world_size() > 1:
        if dist.get_backend() == "nccl":
            dist.all_reduce(counts)
        else:
            cpu = counts.cpu()
            dist.all_reduce(cpu)
            counts = cpu
    return float(counts[0] / counts[1] * 100.0)


@torch.inference_mode()
def dump_features(model, loaders: Dict[str, Iterable], feature_dir: str, feature_file: str = "features.npy",
                  training_chunks: Optional[str] = None, new_channel_init: Optional[str] = None, init_first_layer=None,
                  channel_combinations: Optional[Sequence[int]] = None, device=None) -> List[str]:
    """eval_morphem70k's feature pass (trainer.py:645-690): for every chunk in ``loaders`` (e.g. Allen / HPA / CP) run the
    model in eval mode over the chunk's batches — image tensors, as the CHAMMI test loaders yield (dicts with "image" are
    accepted too) — with ``training_chunks`` and the leave-one-out ``new_channel_init``, concatenate the [b, D] feature
    rows and write them with ``np.save`` to ``<feature_dir>/<chunk>/<feature_file>`` (utils.write_numpy, utils.py:232-235).
    Returns the written paths; the caller hands ``feature_dir`` to ``morphem.benchmark.run_benchmark`` as the reference does."""
    was_training = model.training
    model.eval()
    paths = []
    for chunk_name, loader in loaders.items():
        feats = []
        for batch in loader:
            x = batch["image"] if isinstance(batch, dict) else (batch[0] if isinstance(batch, (tuple, list)) else batch)
            if device is not None:
                x = x.to(device)
            if channel_combinations is not None:
                x = x[:, list(channel_combinations), :, :].clone()  # trainer.py:662-663
            out = model(x, chunk_name, training_chunks, init_first_layer=init_first_layer, new_channel_init=new_channel_init)
            feats.append(out.float().cpu())
        folder = os.path.join(feature_dir, chunk_name)
        os.makedirs(folder, exist_ok=True)
        path = os.path.join(folder, feature_file)
        np.save(path, torch.cat(feats, dim=0).numpy())
        paths.append(path)
    if was_training:
        model.train()
    return paths


@torch.inference_mode()
def eval_subset_channels(model, batches: Iterable, channels: Sequence[int], chunk_name: str = "test", only_sizes: Optional[Sequence[int]] = None,
                         device=None) -> Dict[int, List[float]]:
    """The channel-subset sweep of trainer.py:474-545: for n = len(channels) .. 1 and every combination of n channels, the
    chunk's entry in ``patch_embed.mapper`` is REPLACED by the selected positions (:507-514), the batch is sliced to them and
    the model evaluated (training_chunks=None, new_channel_init=""); returns {n: [top-1 % per combination]}.  The reference
    breaks after the first n (:536); pass ``only_sizes`` to restrict, default = every size.  The mapper entry stays mutated, as
    in the reference."""
    was_training = model.training
    model.eval()
    batches = list(batches)
    channels = list(channels)
    mapper = model.feature_extractor.patch_embed.mapper
    res: Dict[int, List[float]] = {}
    for n_ch in range(len(channels), 0, -1):
        if only_sizes is not None and n_ch not in only_sizes:
            continue
        accs = []
        for selected in combinations(channels, n_ch):
            sel = [channels.index(c) for c in selected]
            correct = total = 0
            for batch in batches:
                x, y = (batch["image"], batch["label"]) if isinstance(batch, dict) else batch[:2]
                if device is not None:
                    x, y = x.to(device), y.to(device)
                mapper[chunk_name] = sel
                out = model(x[:, sel, :, :].clone(), chunk_name, None, init_first_layer=None, new_channel_init="")
                correct += int((out.argmax(dim=-1) == y).sum())
                total += y.numel()
            accs.append(100.0 * correct / max(total, 1))
        res[n_ch] = accs
    if was_training:
        model.train()
    return res
