"""Registry shim with the reference's discovery surface: ``getattr(models, cfg.model.name)(cfg.model, mapper=...)``
(trainer.py:1164, models/__init__.py:9).  INTEGRATION.md shows the one-line change in the reference."""
from .dichavit import DiChaViT, dichavit  # noqa: F401
