"""salve_amd: MI355X-native BEV texture-map rasteriser + early-fusion ResNet verifier.

Drop-in for the one data-parallel hot path of zillow/salve.  The host
side mirrors the reference's module layout (``salve_amd.utils.bev_rendering_utils`` <->
``salve.utils.bev_rendering_utils`` ...); the arithmetic runs in hand-written HIP kernels
behind the C ABI declared in ``include/salve_hip.h``.
"""

__version__ = "0.1.0"


# the modules of the hot path that stand in for their `salve.*` namesakes
HOT_PATH_MODULES = (
    "utils.bev_rendering_utils", "utils.hohonet_pano_utils", "utils.rotation_utils", "utils.normalization_utils", "utils.mesh_grid",
    "utils.zorder_utils", "utils.interpolation_utils", "utils.pr_utils", "common.sim2", "common.bevparams", "models.early_fusion",
    "models.resnet_factory", "train_utils", "training_config", "dataset.zind_data", "dataset.zind_partition",
)


def install_as_salve() -> None:
    """Make `import salve.utils.bev_rendering_utils`, `salve.models.early_fusion`, `salve.common.sim2` ... in the
    reference's scripts resolve to the MI355X implementation.

    * A real `salve` package is importable: it stays; only the hot-path submodules listed in HOT_PATH_MODULES are
      OVERLAID (entered in sys.modules and set as attributes of their real parent packages), so the reference's other
      imports -- salve.utils.io, salve.utils.avg_meter, salve.utils.logger_utils (scripts/test.py:17-23),
      salve.common.posegraph2d, salve.dataset.hnet_prediction_loader (scripts/render_dataset_bev.py:21-26) -- keep
      resolving to the reference's own files.
    * No `salve` package anywhere: the whole package is aliased, which serves the hot path only.
    """
    import importlib
    import importlib.util
    import sys

    me = sys.modules[__name__]
    real = sys.modules.get("salve")
    if real is None:
        try:
            spec = importlib.util.find_spec("salve")
        except (ImportError, ValueError):
            spec = None
        if spec is not None:
            real = importlib.import_module("salve")
    if real is None or real is me:
        sys.modules["salve"] = me
        for name in ("utils", "common", "models", "dataset") + HOT_PATH_MODULES:
            sys.modules.setdefault(f"salve.{name}", importlib.import_module(f"salve_amd.{name}"))
        return
    for name in HOT_PATH_MODULES:
        mod = importlib.import_module(f"salve_amd.{name}")
        parent_name, _, leaf = f"salve.{name}".rpartition(".")
        try:
            parent = importlib.import_module(parent_name)  # the REAL parent package (salve.utils, salve.common, ...)
        except ImportError:
            parent = None
        sys.modules[f"salve.{name}"] = mod
        if parent is not None:
            setattr(parent, leaf, mod)
