"""Builds canonicalvoting_amd/_C/libcvhip.so with hipcc for gfx950 (cross-compiles without a GPU).

    python -m canonicalvoting_amd.csrc.build [--force] [--verbose]

The shared object is built in-tree so it travels with the repo snapshot to the GPU box.
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
OUT_DIR = os.path.join(os.path.dirname(HERE), "_C")
OBJ_DIR = os.path.join(OUT_DIR, "obj")
LIB = os.path.join(OUT_DIR, "libcvhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")

# No packed fp32 instructions (v_pk_add_f32 / v_pk_mul_f32 / v_pk_fma_f32) in any device code: on gfx950 a
# `v_pk_add_f32 ... op_sel:[0,1] op_sel_hi:[1,0]` (the form the SLP vectoriser emits for e.g. a 2-D rotation) returns wrong
# results while another wave on the same CU issues the 16-bit matrix instructions v_mfma_f32_32x32x16_{f16,bf16} /
# 16x16x32_f16 - found through the vote grids of scenes in flight (profiles/r3/vote_concurrency_findings.txt,
# profiles/op_check_probe.py, profiles/microbench/lds_hammer.hip).  The feature switch is a device-target feature: the
# host pass of hipcc warns that it does not know it (harmless, stderr is only shown on failure).
NO_PACKED_FP32 = ["-Xclang", "-target-feature", "-Xclang", "-packed-fp32-ops"]
COMMON = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-I" + os.path.join(ROOT, "include"),
          "-I" + HERE, "-munsafe-fp-atomics", "-Wall", "-Wno-unused-function"] + NO_PACKED_FP32
# strict fp32 (no FMA contraction) where results must match the oracle bit for bit
STRICT = ["-ffp-contract=off"]


def _sources():
    """source -> extra flags, with the experiment switches of the environment as they are NOW (read per call, and part of
    the recorded command line: an object built with other -D flags is stale)"""
    env = lambda k: os.environ.get(k, "").split()
    return {
        "cv_host.cpp": STRICT,
        "hv_vote.hip": STRICT + env("CV_HV_DEFS"),       # tile-shape experiments (-DHV_TX=16 -DHV_TW=8)
        "hv_decode.hip": STRICT + env("CV_DEC_DEFS"),    # greedy-walk experiments (-DDEC_BLOCKED=0)
        "sparse_coords.hip": [],
        "sparse_conv.hip": env("CV_SC_DEFS"),            # kernel experiments (-DCV_WP_CLAMPED_GATHER=1)
        "net_exec.cpp": [],
        "scene_exec.cpp": [],
    }



def _stale(target, deps, cmd=None):
    """target is older than a dependency, or was built by another command line (a changed -D through CV_*_DEFS, another
    HIPCC): the command that made an object is kept beside it as <object>.cmd and compared word for word."""
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    if any(os.path.getmtime(d) > t for d in deps):
        return True
    if cmd is not None:
        try:
            with open(target + ".cmd") as f:
                return f.read() != _cmd_text(cmd)
        except OSError:
            return True
    return False


def _cmd_text(cmd):
    # (the checkout's own path is written as $ROOT: the tree is built here and runs from another path on the GPU box)
    return "\n".join(c.replace(ROOT, "$ROOT") for c in cmd) + "\n"


def build(force=False, verbose=False):
    os.makedirs(OBJ_DIR, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers.append(os.path.join(ROOT, "include", "cv_hip.h"))
    headers.append(os.path.abspath(__file__))
    jobs = []
    objs = []
    for src, extra in _sources().items():
        s = os.path.join(HERE, src)
        o = os.path.join(OBJ_DIR, src + ".o")
        objs.append(o)
        cmd = [HIPCC] + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
        if force or _stale(o, [s] + headers, cmd):
            jobs.append(cmd)

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        out = cmd[cmd.index("-o") + 1]
        if os.path.exists(out + ".cmd"):
            os.remove(out + ".cmd")             # (a failed or interrupted build leaves no record behind)
        r = subprocess.run(cmd, capture_output=True, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s\n%s" % (" ".join(cmd), r.stdout, r.stderr))
        if verbose and r.stderr.strip():
            print(r.stderr)
        with open(out + ".cmd", "w") as f:
            f.write(_cmd_text(cmd))
        return True

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            list(ex.map(run, jobs))
    link = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB] + objs
    if jobs or force or _stale(LIB, objs, link):
        run(link)
    return LIB


def plan(force=False):
    """the sources build() would compile now (tests: a changed CV_*_DEFS must show up here)"""
    headers = [os.path.join(HERE, f) for f in os.listdir(HERE) if f.endswith(".h")]
    headers += [os.path.join(ROOT, "include", "cv_hip.h"), os.path.abspath(__file__)]
    todo = []
    for src, extra in _sources().items():
        s, o = os.path.join(HERE, src), os.path.join(OBJ_DIR, src + ".o")
        cmd = [HIPCC] + COMMON + extra + (["-x", "hip"] if src.endswith(".cpp") else []) + ["-c", s, "-o", o]
        if force or _stale(o, [s] + headers, cmd):
            todo.append(src)
    return todo


def ext_path():
    import sysconfig
    return os.path.join(OUT_DIR, "hv_cuda" + sysconfig.get_config_var("EXT_SUFFIX"))


def build_ext(force=False, verbose=False):
    """canonicalvoting_amd/_C/hv_cuda.<abi>.so: the compiled pybind / torch extension module `hv_cuda`
    (csrc/hv_cuda_ext.cpp = houghvoting/src/hv_cuda.cpp over the C ABI).  Host-only C++ (it calls libcvhip.so), so g++
    compiles it against torch's headers; rpath $ORIGIN finds libcvhip.so next to it."""
    import sysconfig

    import torch
    from torch.utils import cpp_extension as ce
    lib = build(force=force, verbose=verbose)
    src = os.path.join(HERE, "hv_cuda_ext.cpp")
    out = ext_path()
    if not (force or _stale(out, [src, lib, os.path.join(ROOT, "include", "cv_hip.h"), os.path.abspath(__file__)])):
        return out
    inc = ce.include_paths() + ["/opt/rocm/include", sysconfig.get_paths()["include"], os.path.join(ROOT, "include")]
    torch_lib = ce.library_paths()[0]
    cmd = [os.environ.get("CXX", "g++"), "-O2", "-std=c++17", "-fPIC", "-shared", "-D__HIP_PLATFORM_AMD__=1", "-DUSE_ROCM=1",
           "-DTORCH_EXTENSION_NAME=hv_cuda", "-DTORCH_API_INCLUDE_EXTENSION_H",
           "-D_GLIBCXX_USE_CXX11_ABI=%d" % int(torch._C._GLIBCXX_USE_CXX11_ABI), "-Wno-deprecated-declarations"]
    cmd += ["-I" + i for i in inc]
    cmd += [src, "-o", out, "-L" + torch_lib, "-L" + OUT_DIR, "-L/opt/rocm/lib", "-ltorch", "-ltorch_cpu", "-ltorch_hip",
            "-lc10", "-lc10_hip", "-ltorch_python", "-lamdhip64", "-lcvhip", "-Wl,-rpath,$ORIGIN",
            "-Wl,-rpath," + torch_lib, "-Wl,-rpath,/opt/rocm/lib"]
    if verbose:
        print(" ".join(cmd), flush=True)
    r = subprocess.run(cmd, capture_output=True, text=True)
    if r.returncode != 0:
        raise RuntimeError("building the hv_cuda extension failed:\n%s\n%s" % (" ".join(cmd), r.stderr))
    return out


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
    print(build_ext(force="--force" in sys.argv, verbose="--verbose" in sys.argv))
