"""``wmar/augmentations/augmentation_manager.py`` (:33-123): the (name, fn, parameter list) table the harness iterates
(generate.py:142-164).  The classic transforms run on the GPU over the whole batch; neural codecs (compressai / diffusers)
and DiffPure are out of scope and rejected loudly."""
from __future__ import annotations

from .geometric import HorizontalFlip, Rotate, UpperLeftCropWithResizeBack
from .valuemetric import JPEG, Brightness, GaussianBlur, GaussianNoise


class AugmentationManager:
    def __init__(self, include_neural_compress=False, include_diffpure=False, load_augs=True):
        if include_neural_compress or include_diffpure:
            raise NotImplementedError("neural-compression and DiffPure attacks are outside this build (DESIGN.md section 8); "
                                      "pass --include_neural_compress false --include_diffpure false")
        self.include_neural_compress = False
        self.include_diffpure = False
        L = load_augs
        self.augs = [
            ("gaussian-blur", None if not L else (lambda x, kernel_size: GaussianBlur()(x, kernel_size)),
             [0, 1, 3, 5, 7, 9, 11, 13, 15, 17, 19]),
            ("gaussian-noise", None if not L else (lambda x, std: GaussianNoise()(x, std)),
             [0, 0.025, 0.05, 0.075, 0.1, 0.125, 0.15, 0.175, 0.2]),
            ("jpeg", None if not L else (lambda x, quality: JPEG()(x, quality)), [100, 95, 85, 75, 65, 55, 45, 35, 25, 15, 5]),
            ("brightness", None if not L else (lambda x, brightness: Brightness()(x, brightness)),
             [1, 1.25, 1.5, 1.75, 2, 2.25, 2.5, 2.75, 3]),
            ("rotation", None if not L else (lambda x, angle: Rotate()(x, angle)), [-20, -15, -10, -5, 0, 5, 10, 15, 20]),
            ("flip-h", None if not L else (lambda x, do: HorizontalFlip()(x) if do else x), [0, 1]),
            ("upperleft-crop", None if not L else (lambda x, factor: UpperLeftCropWithResizeBack()(x, factor)),
             [1.0, 0.95, 0.9, 0.85, 0.8, 0.75, 0.7, 0.65, 0.6, 0.55, 0.5]),
        ]
        # the table's own callables carry a mark: the harness may replace exactly these by the fused device launch
        # (wmar_amd.augmentations.device_ops.fused), never a caller's own callable registered under the same name
        for _, fn, _ in self.augs:
            if fn is not None:
                fn._wmar_default = True
