"""Build recipe for libipde_hip.so (gfx950 only, in-tree).

    python -m ipde_amd.build            # incremental
    python -m ipde_amd.build --force

hipcc cross-compiles without a GPU; the resulting .so sits in ipde_amd/lib/ and
travels to the GPU box with the repo snapshot (it is git-ignored, not
gpurun-ignored).
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIBDIR = os.path.join(HERE, "lib")
OBJDIR = os.path.join(HERE, "build")
LIB = os.path.join(LIBDIR, "libipde_hip.so")
ROCM = os.environ.get("ROCM_PATH", "/opt/rocm")
HIPCC = os.path.join(ROCM, "bin", "hipcc")

SOURCES = [
    "ctx.hip",
    "layer_prepare.hip",
    "layer_laplace.hip",
    "layer_modhelm.hip",
    "layer_stokes.hip",
    "spectral.hip",
    "fft2d.hip",
    "nufft.hip",
    "annular.hip",
    "ewald.hip",
    "dense.hip",
    "lu_factor.hip",
    "geometry.hip",
    "target_plan.hip",
    "multi.hip",
]

CXXFLAGS = [
    "--offload-arch=gfx950",
    "-O3",
    "-std=c++17",
    "-fPIC",
    "-fno-fast-math",
    "-ffp-contract=on",
    "-Wall", "-Wno-unused-result", "-Wno-unused-value",
    "-Wno-unused-function",
    "-I" + os.path.join(ROCM, "include"),
]


def _newer(a, b):
    return (not os.path.exists(b)) or os.path.getmtime(a) > os.path.getmtime(b)


def _deps_mtime():
    m = 0.0
    for d in (CSRC, os.path.join(HERE, "..", "include")):
        for f in os.listdir(d):
            if f.endswith(".h"):
                m = max(m, os.path.getmtime(os.path.join(d, f)))
    return m


def build_lib(force=False, verbose=True, extra_flags=()):
    os.makedirs(LIBDIR, exist_ok=True)
    os.makedirs(OBJDIR, exist_ok=True)
    hdr_m = _deps_mtime()
    jobs = []
    objs = []
    for s in SOURCES:
        src = os.path.join(CSRC, s)
        obj = os.path.join(OBJDIR, s.replace(".hip", ".o"))
        objs.append(obj)
        stale = force or _newer(src, obj) or (os.path.exists(obj) and hdr_m > os.path.getmtime(obj))
        if stale:
            jobs.append([HIPCC, *CXXFLAGS, *extra_flags, "-c", src, "-o", obj])

    def run(cmd):
        if verbose:
            print("[ipde_amd.build]", " ".join(cmd[-4:]), flush=True)
        r = subprocess.run(cmd, capture_output=True, text=True)
        return cmd, r

    if jobs:
        with ThreadPoolExecutor(max_workers=min(4, len(jobs))) as ex:
            for cmd, r in ex.map(run, jobs):
                if r.returncode != 0:
                    sys.stderr.write(r.stdout + r.stderr)
                    raise RuntimeError("hipcc failed: " + " ".join(cmd))
                if verbose and r.stderr.strip():
                    sys.stderr.write(r.stderr)
    if jobs or not os.path.exists(LIB):
        cmd = [HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB, *objs,
               "-L" + os.path.join(ROCM, "lib"), "-lrocfft"]
        cmd, r = run(cmd)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
    return LIB


if __name__ == "__main__":
    print(build_lib(force="--force" in sys.argv))
