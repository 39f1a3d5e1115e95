"""MI355X-native DiChaViT training hot path (gfx950 HIP kernels behind the reference's plugin surface).

    from diverse_channel_vit_amd import dichavit          # == models.dichavit of the reference
    model = dichavit(cfg.model, mapper=mapper).to("cuda")

See DESIGN.md / INTEGRATION.md.  Importing this package never touches the GPU and never falls back to
a CPU implementation: the first forward loads diverse_channel_vit_amd/libdcv_hip.so or raises."""
from .dichavit import DiChaViT, dichavit, proxy_loss  # noqa: F401
from .dp import DataParallel  # noqa: F401
from .optim import HipAdamW, clip_grad_norm_  # noqa: F401
from .checkpoint import save_checkpoint, load_checkpoint, evaluate, dump_features, eval_subset_channels  # noqa: F401
from .graph import GraphedTrainStep  # noqa: F401
from .hip import set_deterministic, is_deterministic  # noqa: F401

__all__ = ["DiChaViT", "dichavit", "proxy_loss", "DataParallel", "HipAdamW", "GraphedTrainStep", "clip_grad_norm_", "save_checkpoint", "load_checkpoint",
           "evaluate", "dump_features", "eval_subset_channels", "set_deterministic", "is_deterministic"]
