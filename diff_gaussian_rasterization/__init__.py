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

import sys as _sys

# The reference's package holds its extension as a real submodule (DGR/setup.py:22 builds
# ``diff_gaussian_rasterization._C``; DGR/diff_gaussian_rasterization/__init__.py:14 does
# ``from . import _C``): register ours under that dotted name so that both
# ``from diff_gaussian_rasterization import _C`` and ``import diff_gaussian_rasterization._C`` work.
_sys.modules.setdefault(__name__ + "._C", _C)

__all__ = ["GaussianRasterizationSettings", "GaussianRasterizer", "rasterize_gaussians",
           "_RasterizeGaussians", "_C"]
