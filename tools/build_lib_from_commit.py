"""Build libgrpg_rasterizer_<name>.so from the csrc/ of another commit (with this tree's flags) for an A/B on one
box under this tree's binding:  python tools/build_lib_from_commit.py <commit> <name>
(tools/gpu_ab_variants.sh "<name>" LD_PRELOADs it).  Boxes of the pool differ by ~3 %: a delta is only trustworthy
against the previous library on ONE box."""
import os, shutil, subprocess, sys, tempfile
from concurrent.futures import ThreadPoolExecutor
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from gaussianrpg_amd import build as b
commit, name = sys.argv[1], sys.argv[2]
tmp = tempfile.mkdtemp(prefix="grpg_%s_" % name)
subprocess.check_call("git archive %s gaussianrpg_amd/csrc include | tar -x -C %s" % (commit, tmp), shell=True, cwd=ROOT)
src = os.path.join(tmp, "gaussianrpg_amd", "csrc")
def cc(item):
    u, extra = item
    if not os.path.exists(os.path.join(src, u)):
        return None
    o = os.path.join(tmp, u.replace(".hip", ".o"))
    subprocess.check_call([b._hipcc(), "--offload-arch=" + b.ARCH, "-O3", "-std=c++17", "-fPIC", "-fvisibility=hidden",
                           "-Wno-unused-function"] + extra + ["-c", os.path.join(src, u), "-o", o])
    return o
with ThreadPoolExecutor(8) as ex:
    objs = [o for o in ex.map(cc, b.HIP_UNITS.items()) if o]
out = b.variant_path(name)
os.makedirs(os.path.dirname(out), exist_ok=True)
subprocess.check_call([b._hipcc(), "--offload-arch=" + b.ARCH, "-shared", "-fPIC", "-o", out] + objs +
                      ["-Wl,--enable-new-dtags", "-Wl,-rpath,/opt/rocm/lib"])
shutil.rmtree(tmp)
print(out)
