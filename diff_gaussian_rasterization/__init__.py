"""Drop-in module name.  The reference's callers do
``from diff_gaussian_rasterization import GaussianRasterizationSettings, GaussianRasterizer``
(lib/utils/camera_utils.py:13, script/test_gaussian_rasterization.py:4); this package keeps that
import working on a ROCm machine by re-exporting the MI355X-native implementation."""
from gaussianrpg_amd.rasterizer import (  # noqa: F401
    GaussianRasterizationSettings,
    GaussianRasterizer,
    _C,
    _RasterizeGaussians,
    rasterize_gaussians,
)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians",
           "_RasterizeGaussians", "_C"]
