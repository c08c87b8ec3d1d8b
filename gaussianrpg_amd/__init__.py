"""gaussianrpg_amd -- MI355X-native (gfx950 / CDNA4) 3D-Gaussian-splatting rasterizer.

Drop-in for the hot path of GimpelZhang/GaussianRPG: the ``diff_gaussian_rasterization``
operator (``GaussianRasterizer`` / ``GaussianRasterizationSettings`` over
``_C.rasterize_gaussians`` / ``_C.rasterize_gaussians_backward``).

Layout:
  csrc/           hand-written HIP kernels + the C ABI (``libgrpg_rasterizer.so``) and the torch
                  binding (``_C``)
  rasterizer.py   host-side mirror of the reference's Python operator interface
  harness.py      build-owned counterpart of the reference's callers (cameras, synthetic scenes)
  trajectory.py   frame-sharded multi-GPU trajectory rendering (one process per GPU, RCCL gather)
  build.py        in-tree build (hipcc for gfx950, g++ for the binding)

Importing this package does not load native code; ``gaussianrpg_amd.rasterizer`` does, and fails
loudly if the extension has not been built -- there is no CPU or PyTorch fallback.
"""
__version__ = "0.1.0"
