"""salve_amd: MI355X-native BEV texture-map rasteriser + early-fusion ResNet verifier.

Drop-in for the one data-parallel hot path of zillow/salve.  The host
side mirrors the reference's module layout (``salve_amd.utils.bev_rendering_utils`` <->
``salve.utils.bev_rendering_utils`` ...); the arithmetic runs in hand-written HIP kernels
behind the C ABI declared in ``include/salve_hip.h``.
"""

__version__ = "0.1.0"
