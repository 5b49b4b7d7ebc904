"""Builds libfrlhip.so (all HIP kernels + the C ABI) for gfx950 with hipcc, in-tree.

Usage: python vq-vae_amd/build.py [--force]
Objects are cached under vq-vae_amd/build/ keyed by source mtime; the shared library lands in
vq-vae_amd/frl_hip/libfrlhip.so (git-ignored, but shipped to the GPU box by gpurun).
"""
from __future__ import annotations

import glob
import os
import subprocess
import sys
from concurrent.futures import ThreadPoolExecutor

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
OBJ = os.path.join(HERE, "build")
OUT = os.path.join(HERE, "frl_hip", "libfrlhip.so")
HIPCC = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
FLAGS = ["--offload-arch=gfx950", "-O3", "-fPIC", "-std=c++17", "-Wno-unused-result",
         "-I", CSRC, "-I", os.path.join(os.path.dirname(HERE), "include")]
# A/B builds: FRL_BUILD_TAG=x python build.py --force  ->  build_x/*.o, frl_hip/libfrlhip_x.so (loaded when FRL_HIP_LIB_TAG=x), compiled
# with FRL_EXTRA_FLAGS appended
TAG = os.environ.get("FRL_BUILD_TAG", "")
if TAG:
    OBJ = os.path.join(HERE, "build_" + TAG)
    OUT = os.path.join(HERE, "frl_hip", f"libfrlhip_{TAG}.so")
FLAGS += os.environ.get("FRL_EXTRA_FLAGS", "").split()


def file_flags(src: str):
    """Per-file compiler flags: a line `// build-flags: ...` among the first 40 lines of the source."""
    out = []
    with open(src) as f:
        for _, line in zip(range(40), f):
            if line.startswith("// build-flags:"):
                out += line[len("// build-flags:"):].split()
    return out


def _newer(src: str, dst: str, deps) -> bool:
    if not os.path.exists(dst):
        return True
    t = os.path.getmtime(dst)
    return any(os.path.getmtime(p) > t for p in [src] + list(deps))


def build(force: bool = False, verbose: bool = True) -> str:
    os.makedirs(OBJ, exist_ok=True)
    srcs = sorted(glob.glob(os.path.join(CSRC, "*.hip")))
    hdrs = glob.glob(os.path.join(CSRC, "*.hpp")) + glob.glob(os.path.join(os.path.dirname(HERE), "include", "*.h"))
    jobs = []
    for s in srcs:
        o = os.path.join(OBJ, os.path.basename(s)[:-4] + ".o")
        if force or _newer(s, o, hdrs):
            jobs.append((s, o))

    def cc(job):
        s, o = job
        r = subprocess.run([HIPCC] + FLAGS + file_flags(s) + ["-c", s, "-o", o], capture_output=True, text=True)
        return s, r.returncode, r.stdout + r.stderr

    if jobs:
        with ThreadPoolExecutor(max_workers=min(6, len(jobs))) as ex:
            for s, rc, log in ex.map(cc, jobs):
                if verbose:
                    print(f"[build] hipcc {os.path.basename(s)} rc={rc}", flush=True)
                if rc != 0:
                    sys.stderr.write(log)
                    raise RuntimeError(f"hipcc failed for {s}")
    objs = [os.path.join(OBJ, os.path.basename(s)[:-4] + ".o") for s in srcs]
    if jobs or not os.path.exists(OUT) or any(os.path.getmtime(o) > os.path.getmtime(OUT) for o in objs):
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-shared", "-fPIC", "-o", OUT] + objs,
                           capture_output=True, text=True)
        if r.returncode != 0:
            sys.stderr.write(r.stdout + r.stderr)
            raise RuntimeError("link failed")
        if verbose:
            print(f"[build] linked {OUT}", flush=True)
    return OUT


if __name__ == "__main__":
    build(force="--force" in sys.argv)
