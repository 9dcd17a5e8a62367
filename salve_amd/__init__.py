"""salve_amd: MI355X-native BEV texture-map rasteriser + early-fusion ResNet verifier.

Drop-in for the one data-parallel hot path of zillow/salve.  The host
side mirrors the reference's module layout (``salve_amd.utils.bev_rendering_utils`` <->
``salve.utils.bev_rendering_utils`` ...); the arithmetic runs in hand-written HIP kernels
behind the C ABI declared in ``include/salve_hip.h``.
"""

__version__ = "0.1.0"


def install_as_salve() -> None:
    """Alias this package as `salve` so that `import salve.utils.bev_rendering_utils`, `salve.models.early_fusion`,
    `salve.common.sim2` ... in the reference's scripts resolve to the MI355X implementation."""
    import importlib
    import sys

    sys.modules.setdefault("salve", sys.modules[__name__])
    for name in ("utils", "common", "models", "utils.bev_rendering_utils", "utils.hohonet_pano_utils", "utils.rotation_utils",
                 "utils.normalization_utils", "utils.mesh_grid", "utils.zorder_utils", "utils.interpolation_utils", "common.sim2", "common.bevparams", "models.early_fusion",
                 "models.resnet_factory", "train_utils", "training_config", "dataset", "dataset.zind_data", "dataset.zind_partition",
                 "utils.pr_utils"):
        sys.modules.setdefault(f"salve.{name}", importlib.import_module(f"salve_amd.{name}"))
