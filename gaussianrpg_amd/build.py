"""In-tree build of the native pieces (no JIT cache: the .so files must travel with the repo).

* ``libgrpg_rasterizer.so`` -- hand-written HIP for gfx950 + the C ABI (include/grpg_rasterizer.h),
  compiled with ``hipcc --offload-arch=gfx950``; no torch dependency.
* ``_C.<abi>.so`` -- the PyTorch-ROCm extension module (csrc/torch_binding.cpp) that exposes the
  reference's operator surface on top of the C ABI; compiled with g++ against torch's headers.

Both land next to this file.  ``python -m gaussianrpg_amd.build`` builds everything.
"""
import os
import shutil
import subprocess
import sys
import sysconfig
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(HERE)
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(ROOT, "build", "obj")
LIB_NAME = "libgrpg_rasterizer.so"
LIB_PATH = os.path.join(HERE, LIB_NAME)
EXT_PATH = os.path.join(HERE, "_C" + sysconfig.get_config_var("EXT_SUFFIX"))
ARCH = "gfx950"

# translation unit -> extra flags
HIP_UNITS = {
    # bit-exact integer outputs need the oracle's arithmetic: no FMA contraction
    "preprocess.hip": ["-ffp-contract=off"],
    "preprocess_bwd.hip": ["-ffp-contract=off"],
    "sort.hip": [],
    "binning.hip": [],
    "hier_binning.hip": [],
    "render_fwd.hip": [],
    # hardware global_atomic_add_f32 instead of a CAS loop; no SLP packing of the scalar gradient
    # arithmetic: a v_pk_fma_f32 occupies the gfx950 VALU 1.8x as long as a v_fma_f32
    # (tools/ubench/valu_rate.hip), so pairs only add moves
    "render_bwd.hip": ["-munsafe-fp-atomics", "-fno-slp-vectorize"],
    # bit-identical to oracle/knn_oracle.py: one rounding per operation
    "knn.hip": ["-ffp-contract=off"],
    # hardware float atomics for the cube-map gradient
    "sky.hip": ["-munsafe-fp-atomics"],
    "api.hip": [],
}
HEADERS = ["common.h", "gaussian_math.h", "blend_math.h", "compose_math.h", os.path.join(ROOT, "include", "grpg_rasterizer.h")]


def _hipcc():
    for c in (os.environ.get("HIPCC"), shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if c and os.path.exists(c):
            return c
    raise RuntimeError("hipcc not found (needed to build the gfx950 kernels)")


def _newer(target, sources):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(s) > t for s in sources)


def _run(cmd):
    p = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    if p.returncode != 0:
        raise RuntimeError("command failed: %s\n%s" % (" ".join(cmd), p.stdout))
    return p.stdout


def build_native(force=False, verbose=False):
    """hipcc -> libgrpg_rasterizer.so (gfx950 only)."""
    os.makedirs(OBJ, exist_ok=True)
    hipcc = _hipcc()
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    jobs = []
    objs = []
    for unit, extra in HIP_UNITS.items():
        src = os.path.join(CSRC, unit)
        obj = os.path.join(OBJ, unit.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs + [os.path.abspath(__file__)]):
            jobs.append([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC",
                         "-fvisibility=hidden", "-Wall", "-Wno-unused-function"] + extra +
                        os.environ.get("GRPG_EXTRA_HIPCC_FLAGS", "").split() +   # experiments (-D...)
                        ["-c", src, "-o", obj])
    if jobs:
        with ThreadPoolExecutor(max_workers=min(len(jobs), os.cpu_count() or 4)) as ex:
            for out in ex.map(_run, jobs):
                if verbose and out.strip():
                    print(out)
    if force or jobs or _newer(LIB_PATH, objs):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", LIB_PATH] + objs +
             ["-Wl,--enable-new-dtags", "-Wl,-rpath,/opt/rocm/lib"])
    return LIB_PATH


def build_binding(force=False, verbose=False):
    """g++ -> _C extension module linking libgrpg_rasterizer.so and torch."""
    import torch
    from torch.utils import cpp_extension as ce
    src = os.path.join(CSRC, "torch_binding.cpp")
    hdr = os.path.join(ROOT, "include", "grpg_rasterizer.h")
    if not (force or _newer(EXT_PATH, [src, hdr, os.path.abspath(__file__)])):
        return EXT_PATH
    rocm = os.environ.get("ROCM_PATH", "/opt/rocm")
    cmd = ["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-Wall", "-Wno-unused-variable",
           "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1", "-DTORCH_EXTENSION_NAME=_C",
           "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI)]
    for inc in ce.include_paths() + [sysconfig.get_paths()["include"], os.path.join(rocm, "include")]:
        cmd += ["-isystem", inc]
    torch_lib = os.path.join(os.path.dirname(torch.__file__), "lib")
    cmd += [src, "-o", EXT_PATH, "-L" + HERE, "-l:" + LIB_NAME, "-L" + torch_lib, "-ltorch",
            "-ltorch_cpu", "-ltorch_python", "-lc10", "-lc10_hip", "-ltorch_hip",
            "-Wl,--enable-new-dtags", "-Wl,-rpath,$ORIGIN", "-Wl,-rpath," + torch_lib]
    out = _run(cmd)
    if verbose and out.strip():
        print(out)
    return EXT_PATH


# Test variants of the C-ABI library: the same sources with one translation unit recompiled under an
# extra -D.  They live under build/variants/ (git-ignored like every built file, shipped to the GPU
# box by gpurun), are loaded through ctypes by the one test that needs them and are never imported
# by the package.
VARIANTS = {
    # every producer / consumer hand-over of the render gives up at once: the timeout -> error path
    # (render_fwd.hip pc_fail, tests/test_gpu_pc_timeout.py)
    "pcspin0": {"render_fwd.hip": ["-DGRPG_PC_SPIN_LIMIT=0"]},
}
# experiment builds (not built by build_all): python -c "from gaussianrpg_amd import build; build.build_variant('dstrace')"
EXPERIMENT_VARIANTS = {
    # phase timestamps of the depth sort's scatter passes (tools/ds_trace.py)
    "dstrace": {"sort.hip": ["-DGRPG_DS_TRACE"]},
    # per-wave trace of the render (GRPG_RENDER_TRACE=<file>, tools/trace_render.py) and the loop
    # counters / ablation switches of the blend backward (GRPG_BWD_STATS, GRPG_BWD_ABLATE)
    "trace": {"render_fwd.hip": ["-DGRPG_TRACE"], "render_bwd.hip": ["-DGRPG_TRACE"]},
    # render capped at 3 / 2 workgroups per CU by unused LDS (does a second stream's kernel co-reside?)
    # num_rendered published by a launch of its own right behind preprocess instead of by the sort's first pass
    "pubearly": {"api.hip": ["-DGRPG_PUBLISH_EARLY"]},
    # the classic (256-thread, 21 KB of LDS) depth-sort passes for every P: does a small footprint overlap better?
    "classicsort": {"api.hip": ["-DGRPG_FORCE_CLASSIC_SORT"]},
    # 96 registers / 101 KB of LDS per sort workgroup: room for another stream's workgroup on the same CU
    "sortlean": {"sort.hip": ["-DGRPG_SORT_LEAN"]},
    # round 5: only the class-0 workgroups run (images are wrong): how long is the chain of the longest
    # tiles with nothing beside it?  (+ the per-role trace of it, tools/trace_class0.py)
    # round 5: the coarse emit behind the two-launch offsets scan again (what the fused emit is measured against)
    "unfusedemit": {"api.hip": ["-DGRPG_UNFUSED_COARSE_EMIT"]},
    "only0": {"render_fwd.hip": ["-DGRPG_RENDER_ONLY_CLASS0"]},
    "no0": {"render_fwd.hip": ["-DGRPG_RENDER_NO_CLASS0"]},
    "only0trace": {"render_fwd.hip": ["-DGRPG_RENDER_ONLY_CLASS0", "-DGRPG_TRACE"], "render_bwd.hip": ["-DGRPG_TRACE"]},
    # the producer / consumer pairs from 4096 / 16384 entries instead of 8192
    "pc4096": {"render_fwd.hip": ["-DGRPG_RENDER_PC_MIN=4096"], "api.hip": ["-DGRPG_RENDER_PC_MIN=4096"],
               "hier_binning.hip": ["-DGRPG_RENDER_PC_MIN=4096"]},
    "pc16384": {"render_fwd.hip": ["-DGRPG_RENDER_PC_MIN=16384"], "api.hip": ["-DGRPG_RENDER_PC_MIN=16384"],
                "hier_binning.hip": ["-DGRPG_RENDER_PC_MIN=16384"]},
    # round 6: the layered kernel allocated for 3 waves per SIMD (168 VGPRs, no scratch) instead of 4
    # round 6 (wrong images): the layered walk without the object layer's state / without the background layer's
    # round 6: one walk per layer for object tiles from 2048 / 8192 entries instead of 4096
    "lpc2048": {"render_fwd.hip": ["-DGRPG_LAYERS_PC_MIN=2048"], "api.hip": ["-DGRPG_LAYERS_PC_MIN=2048"]},
    "lpc8192": {"render_fwd.hip": ["-DGRPG_LAYERS_PC_MIN=8192"], "api.hip": ["-DGRPG_LAYERS_PC_MIN=8192"]},
    "lay_noobj": {"render_fwd.hip": ["-DGRPG_LAYERS_ABLATE=1"]},
    "lay_nobg": {"render_fwd.hip": ["-DGRPG_LAYERS_ABLATE=2"]},
    "lay_none": {"render_fwd.hip": ["-DGRPG_LAYERS_ABLATE=3"]},
    # round 6: per-workgroup life of the point-list fill (tools/fill_trace.py)
    # round 6: the fill with every lane reading record 0 / 1 instead of its own (what does the gather cost?)
    "fill_nogather": {"hier_binning.hip": ["-DGRPG_FILL_ABLATE=2"]},
    "filltrace": {"hier_binning.hip": ["-DGRPG_FILL_TRACE"]},
    # round 6: 512-entry segments in the hierarchical binning (26 KB of LDS, 52 registers: four fill workgroups per CU)
    "hbseg512": {"hier_binning.hip": ["-DGRPG_HB_SEG=512"]},
    "layers3w": {"render_fwd.hip": ["-DGRPG_LAYERS_MIN_WAVES=3"]},
    # round 6 (wrong host bytes): a drained frame whose drain workgroups exit at once -- what the blending waves' share
    # of a host-destination frame costs (write-through staging stores, the arrival's wait and atomic)
    "drainidle": {"render_fwd.hip": ["-DGRPG_DRAIN_IDLE"]},
    # round 6: light / mid tiles of the backward by ONE wave at 4 pixels per lane (half the reductions there)
    "bwdpx4": {"render_bwd.hip": ["-DGRPG_BWD_LIGHT_SPLIT=1"]},
    "pad12": {"render_fwd.hip": ["-DGRPG_RENDER_LDS_PAD=12288"]},
    "pad26": {"render_fwd.hip": ["-DGRPG_RENDER_LDS_PAD=26624"]},
}


def variant_path(name):
    return os.path.join(ROOT, "build", "variants", "libgrpg_rasterizer_%s.so" % name)


def build_variant(name, force=False):
    """build/variants/libgrpg_rasterizer_<name>.so from the objects of build_native(), except the
    units VARIANTS[name] lists, which are recompiled with the extra flags."""
    build_native()
    hipcc = _hipcc()
    spec = VARIANTS.get(name) or EXPERIMENT_VARIANTS[name]
    out = variant_path(name)
    vobj = os.path.join(ROOT, "build", "obj_" + name)
    os.makedirs(vobj, exist_ok=True)
    os.makedirs(os.path.dirname(out), exist_ok=True)
    hdrs = [h if os.path.isabs(h) else os.path.join(CSRC, h) for h in HEADERS]
    objs = []
    for unit, extra in HIP_UNITS.items():
        if unit not in spec:
            objs.append(os.path.join(OBJ, unit.replace(".hip", ".o")))
            continue
        src = os.path.join(CSRC, unit)
        obj = os.path.join(vobj, unit.replace(".hip", ".o"))
        objs.append(obj)
        if force or _newer(obj, [src] + hdrs + [os.path.abspath(__file__)]):
            _run([hipcc, "--offload-arch=" + ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                  "-Wall", "-Wno-unused-function"] + extra + spec[unit] + ["-c", src, "-o", obj])
    if force or _newer(out, objs):
        _run([hipcc, "--offload-arch=" + ARCH, "-shared", "-fPIC", "-o", out] + objs +
             ["-Wl,--enable-new-dtags", "-Wl,-rpath,/opt/rocm/lib"])
    return out


def build_all(force=False, verbose=False):
    build_native(force=force, verbose=verbose)
    build_binding(force=force, verbose=verbose)
    for name in VARIANTS:
        build_variant(name, force=force)
    return LIB_PATH, EXT_PATH


if __name__ == "__main__":
    paths = build_all(force="--force" in sys.argv, verbose=True)
    print("built:", *paths, sep="\n  ")
