"""Builds libconvexadam_hip.so (gfx950) in-tree:  python -m convexadam_amd.csrc.build

hipcc cross-compiles without a GPU.  Flags that are part of the numerical contract:
  -ffp-contract=off   no implicit FMA (the kernels write fmaf where the reference's ATen build fuses)
  no -ffast-math      IEEE division / sqrt, denormals kept
"""
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
SOURCES = ["api.hip", "mind.hip", "mindmarch.hip", "pool.hip", "correlate.hip", "corrbox.hip", "corrfused.hip", "corrcert.hip", "certify.hip", "convex.hip", "adam.hip", "adamfast.hip", "warp.hip", "boxmarch.hip", "boxtile.hip", "metrics.hip", "edt.hip", "surfdist.hip", "pipeline.hip"]
PER_FILE_FLAGS = {"warp.hip": ["-fno-slp-vectorize"], "adamfast.hip": ["-fno-slp-vectorize"], "boxmarch.hip": ["-fno-slp-vectorize"], "boxtile.hip": ["-fno-slp-vectorize"], "corrbox.hip": ["-fno-slp-vectorize"], "corrfused.hip": ["-fno-slp-vectorize"], "corrcert.hip": ["-fno-slp-vectorize"], "mindmarch.hip": ["-fno-slp-vectorize"]}     # see the header of warp.hip
LIB = os.path.join(HERE, "libconvexadam_hip.so")
OBJDIR = os.path.join(HERE, "build")
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fhip-fp32-correctly-rounded-divide-sqrt", "-fPIC", "-fvisibility=hidden",
         "-Wall", "-Wno-unused-function", "-I" + os.path.join(ROOT, "include"), "-I" + HERE]


def hipcc():
    for c in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if c and (os.path.isabs(c) and os.path.exists(c) or not os.path.isabs(c)):
            return c
    return "hipcc"


def _stale(target, deps):
    if not os.path.exists(target):
        return True
    t = os.path.getmtime(target)
    return any(os.path.getmtime(d) > t for d in deps)


def build(force=False, verbose=False, jitter=False):
    """jitter=True: the race-stress variant (random per-wavefront delays around every barrier, cvx_common.h) as
    libconvexadam_hip_jitter.so; select it at run time with CONVEXADAM_HIP_LIB=<path>."""
    objdir, lib = (os.path.join(HERE, "build_jitter"), os.path.join(HERE, "libconvexadam_hip_jitter.so")) if jitter else (OBJDIR, LIB)
    os.makedirs(objdir, exist_ok=True)
    headers = [os.path.join(HERE, f) for f in sorted(os.listdir(HERE)) if f.endswith(".h")] + [os.path.join(ROOT, "include", "convexadam_hip.h"), os.path.abspath(__file__)]
    jobs = []
    for src in SOURCES:
        s = os.path.join(HERE, src)
        o = os.path.join(objdir, src.replace(".hip", ".o"))
        if force or _stale(o, [s] + headers):
            jobs.append([hipcc()] + FLAGS + PER_FILE_FLAGS.get(src, []) + ["-DCVX_BUILDING=1"] + (["-DCVX_RACE_JITTER=1"] if jitter else []) + ["-c", s, "-o", o])

    def run(cmd):
        if verbose:
            print(" ".join(cmd), flush=True)
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        if r.returncode != 0:
            raise RuntimeError("hipcc failed:\n%s\n%s" % (" ".join(cmd), r.stdout))
        return r.stdout

    with ThreadPoolExecutor(max_workers=min(8, max(1, len(jobs)))) as ex:
        outs = list(ex.map(run, jobs))
    if verbose:
        for o in outs:
            if o.strip():
                print(o)
    objs = [os.path.join(objdir, s.replace(".hip", ".o")) for s in SOURCES]
    if force or jobs or _stale(lib, objs):
        run([hipcc(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", lib] + objs)
    return lib


if __name__ == "__main__":
    print(build(force="--force" in sys.argv, verbose=True, jitter="--jitter" in sys.argv))
