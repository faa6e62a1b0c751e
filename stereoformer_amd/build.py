"""Builds libstereo_hip.so (gfx950) in-tree with hipcc.  No torch extension
machinery: the library is a plain C-ABI shared object (include/stereo_hip.h)."""
from __future__ import annotations

import glob
import os
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libstereo_hip.so")
ARCH = "gfx950"


def sources() -> list[str]:
    return sorted(glob.glob(os.path.join(CSRC, "*.hip")))


def is_stale() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = sources() + glob.glob(os.path.join(CSRC, "*.h")) + [os.path.join(HERE, "..", "include", "stereo_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps)


def build(force: bool = False, verbose: bool = False) -> str:
    """hipcc --offload-arch=gfx950 every csrc/*.hip into one shared library."""
    if not force and not is_stale():
        return LIB
    hipcc = os.environ.get("HIPCC", "/opt/rocm/bin/hipcc")
    objdir = os.path.join(HERE, "build")
    os.makedirs(objdir, exist_ok=True)
    flags = ["-O3", f"--offload-arch={ARCH}", "-fPIC", "-std=c++17", "-fno-fast-math", "-fno-slp-vectorize", "-Wall",
             "-Wno-unused-function"]
    objs = []
    procs = []
    # the split-bf16 attention kernels take their running maxima with nnan max (no NaN reaches them: a band's nearest key
    # is never masked): without the flag every MFMA result is canonicalised (v_max x, x, x) before its first v_max
    extra = {"attention_bf16.hip": ["-fno-honor-nans"], "attention_chunked.hip": ["-fno-honor-nans"],
             "attention_f16.hip": ["-fno-honor-nans"]}
    for src in sources():
        obj = os.path.join(objdir, os.path.basename(src).replace(".hip", ".o"))
        objs.append(obj)
        if (not force and os.path.exists(obj) and os.path.getmtime(obj) >= os.path.getmtime(src)
                and all(os.path.getmtime(obj) >= os.path.getmtime(h) for h in glob.glob(os.path.join(CSRC, "*.h")))
                and os.path.getmtime(obj) >= os.path.getmtime(os.path.join(HERE, "..", "include", "stereo_hip.h"))):
            continue
        cmd = [hipcc, *flags, *extra.get(os.path.basename(src), []), "-c", src, "-o", obj]
        if verbose:
            print(" ".join(cmd))
        procs.append((src, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for src, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError(f"hipcc failed on {src}:\n{out.decode()}")
        if verbose and out:
            print(out.decode())
    cmd = [hipcc, "-shared", "-fPIC", f"--offload-arch={ARCH}", "-o", LIB, *objs]
    if verbose:
        print(" ".join(cmd))
    subprocess.run(cmd, check=True)
    return LIB


if __name__ == "__main__":
    print(build(force="--force" in os.sys.argv, verbose=True))
