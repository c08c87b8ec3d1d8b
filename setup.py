"""Packaging of the drop-in module.  The native pieces are built in-tree by
``python -m gaussianrpg_amd.build`` (explicit hipcc / g++ commands for gfx950) and shipped as
package data, so that ``pip install .`` yields the same importable names as the reference's
submodules (``diff_gaussian_rasterization`` with its ``_C`` extension, DGR/setup.py:16-34, and
``simple_knn._C.distCUDA2``, submodules/simple-knn/setup.py)."""
import os
import sys

from setuptools import find_packages, setup

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

if any(cmd in sys.argv for cmd in ("build", "build_py", "build_ext", "install", "develop",
                                   "bdist_wheel", "editable_wheel")):
    from gaussianrpg_amd import build as _b
    _b.build_all(verbose=True)

setup(
    name="gaussianrpg-amd",
    version="0.1.0",
    description="MI355X-native (gfx950) 3D-Gaussian-splatting rasterizer, drop-in for "
                "diff_gaussian_rasterization as used by GaussianRPG / Street-Gaussians",
    packages=find_packages(include=["gaussianrpg_amd", "gaussianrpg_amd.*",
                                    "diff_gaussian_rasterization", "simple_knn"]),
    package_data={"gaussianrpg_amd": ["*.so", "csrc/*"]},
    include_package_data=True,
    python_requires=">=3.8",
)
