"""MI355X-native underwater image enhancement: hand-written HIP kernels (gfx950) behind the reference's API.

Importing this package does not touch the GPU; the first call into :mod:`api` loads ``lib/libuwie.so`` and
fails loudly when it (or a ROCm device) is missing -- there is no CPU fallback.
"""
from ._lib import UwieError, UwieParams, build, load  # noqa: F401
from .api import (CONFIG_QUALITY_WEIGHTS, CONFIG_STRATEGIES, DRIVER_STRATEGIES, QUALITY_KEYS, DifferentiableEnhancement, UnsupportedInputError, EnhancementStrategies, QualityAssessment, SixStrategies, color_correction,  # noqa: F401
                  detect_image_type, enhance, enhance_all, extract_all_features, process_batch, quality_scores, select_best)
from .runtime import Device, get_device  # noqa: F401
from .streaming import StreamEnhancer  # noqa: F401

__all__ = ["enhance", "enhance_all", "process_batch", "extract_all_features", "QualityAssessment", "quality_scores", "QUALITY_KEYS", "DRIVER_STRATEGIES", "select_best", "CONFIG_STRATEGIES", "CONFIG_QUALITY_WEIGHTS", "DifferentiableEnhancement", "SixStrategies", "EnhancementStrategies", "detect_image_type", "color_correction", "Device",
           "get_device", "StreamEnhancer", "UwieError", "UnsupportedInputError", "UwieParams", "build", "load"]
