"""Evaluation-time image transforms feeding batched ``images_to_codes`` + ``detect``.
Mirrors ``wmar/augmentations`` for the classic transforms; neural codecs and DiffPure are out of scope."""
from .augmentation_manager import AugmentationManager  # noqa: F401
