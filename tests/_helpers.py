"""Helpers shared by CPU and GPU tests."""

import torch


def randomise_bn(model, seed=0):
    """Trained-looking BatchNorm statistics so that activations stay O(1) through the trunk (a freshly initialised
    network with identity BatchNorm has no normalisation at all: its activations grow with depth)."""
    g = torch.Generator().manual_seed(seed)
    for name, m in model.named_modules():
        if isinstance(m, torch.nn.BatchNorm2d):
            last = name.endswith("bn3") or (name.endswith("bn2") and model.resnet.block_kind == "basic")
            m.weight.data = (0.25 if last else 1.0) * (0.6 + 0.4 * torch.rand(m.num_features, generator=g))
            m.bias.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_mean.data = 0.1 * torch.randn(m.num_features, generator=g)
            m.running_var.data = 0.6 + 0.8 * torch.rand(m.num_features, generator=g)


def oracle_floor_render(args):
    """(pool worker, CPU only) canonical-mode oracle render of pano `i1` under (R, t), floor surface: returns the final BEV
    image and the BEV pixel indices (int16 [M, 2], raster order) of the points inside the window."""
    i1, R, t = args
    import numpy as np

    from oracle import bev_oracle as bo
    from salve_amd import synthetic as syn

    rgb, depth = syn.make_pano(int(i1))
    a = bo.xyzrgb_from_arrays(depth, rgb, bo.floor_ceiling_z_range("floor"))
    a, _ = bo.pose_pair(a, a[:1], R, t)
    r = bo.render_bev_image(a, mode="exact")
    if r is None:
        return None, None
    return r["bev"], np.asarray(r["img_xy"]).astype(np.int16)
