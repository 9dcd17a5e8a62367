"""ImageNet statistics on the 0-255 scale.  Mirror of salve/utils/normalization_utils.py:13-26."""

from typing import List, Tuple


def get_imagenet_mean_std() -> Tuple[List[float], List[float]]:
    mean = [m * 255 for m in (0.485, 0.456, 0.406)]
    std = [s * 255 for s in (0.229, 0.224, 0.225)]
    return mean, std
