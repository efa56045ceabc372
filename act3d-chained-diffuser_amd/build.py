"""Builds libact3d_hip.so (gfx950) in-tree with hipcc.  No cmake, no JIT cache: the .so travels with the repo."""
import os
import shutil
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

PKG_DIR = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(PKG_DIR, "csrc")
OBJ_DIR = os.path.join(CSRC, "obj")
# A3D_LIB_OUT=libact3d_hip_<tag>.so (with A3D_HIPCC_FLAGS): an A/B build next to the default library, objects in obj_<tag>/
_OUT = os.path.basename(os.environ.get("A3D_LIB_OUT", "libact3d_hip.so"))
LIB_PATH = os.path.join(PKG_DIR, _OUT)
if _OUT != "libact3d_hip.so":
    OBJ_DIR = os.path.join(CSRC, "obj_" + _OUT[len("libact3d_hip_"):-3])
INCLUDE = os.path.join(os.path.dirname(PKG_DIR), "include")
SOURCES = ["api.hip", "linear.hip", "linear_split.hip", "rope.hip", "attention.hip", "attention_bwd.hip", "attention16.hip", "attention8.hip", "scene.hip", "heads.hip", "diffusion.hip", "vision.hip", "dropout.hip", "denoise.hip", "single_query.hip", "single_query_wave.hip", "query_stream.hip", "data.hip", "conv1x1.hip", "conv1x1_deep.hip", "conv3x3.hip", "fpn_sparse.hip", "stem.hip"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-result"]
# The attention kernels never produce NaNs on their own (masked rows are handled explicitly, -inf only enters exp2):
# without IEEE-mode sNaN quieting hipcc drops the canonicalising v_max_f32 x, x it otherwise puts in front of every fmaxf
# on an MFMA result (12 of 108 VALU instructions per 64-key chunk of the VALU-bound forward loop).
_ATTN_FLAGS = ["-fno-honor-nans", "-mno-amdgpu-ieee"]
EXTRA_FLAGS = {"attention.hip": _ATTN_FLAGS, "attention_bwd.hip": _ATTN_FLAGS, "attention16.hip": _ATTN_FLAGS,
               "attention8.hip": _ATTN_FLAGS}


# development aid: extra hipcc flags for an A/B build on the GPU box, e.g. A3D_HIPCC_FLAGS="-DA3D_LIBM_SINCOS" python build.py --force
ENV_FLAGS = os.environ.get("A3D_HIPCC_FLAGS", "").split()


def _hipcc():
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libact3d_hip.so cannot be built on this machine")


def _newer(src, dst):
    return (not os.path.exists(dst)) or os.path.getmtime(src) > os.path.getmtime(dst)


def build(force=False, verbose=False):
    """Compile every csrc/*.hip for gfx950 and link libact3d_hip.so.  Returns the library path."""
    hipcc = _hipcc()
    os.makedirs(OBJ_DIR, exist_ok=True)
    deps = [os.path.join(CSRC, "a3d_common.h"), os.path.join(CSRC, "attn_ring.h"), os.path.join(INCLUDE, "act3d_hip.h"),
            os.path.abspath(__file__)]
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJ_DIR, s.replace(".hip", ".o"))
        if force or _newer(src, obj) or any(_newer(d, obj) for d in deps):
            jobs.append((src, obj))

    def compile_one(job):
        src, obj = job
        cmd = [hipcc] + FLAGS + EXTRA_FLAGS.get(os.path.basename(src), []) + ENV_FLAGS + ["-c", src, "-o", obj]
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed for %s:\n%s" % (src, r.stderr[-4000:]))
        if verbose:
            print("compiled", os.path.basename(src), file=sys.stderr)
        return obj

    if jobs:
        with ThreadPoolExecutor(max_workers=min(8, len(jobs))) as ex:
            list(ex.map(compile_one, jobs))
    objs = [os.path.join(OBJ_DIR, s.replace(".hip", ".o")) for s in srcs]
    if force or jobs or not os.path.exists(LIB_PATH):
        cmd = [hipcc, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("link failed:\n%s" % r.stderr[-4000:])
    return LIB_PATH


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True))
