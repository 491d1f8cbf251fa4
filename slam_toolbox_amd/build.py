"""Builds libkartohip.so (HIP/C++ for gfx950) in-tree with hipcc.  No CPU fallback is built."""
from __future__ import annotations

import os
import shutil
import subprocess

HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(HERE, "csrc")
LIB = os.path.join(HERE, "libkartohip.so")
SOURCES = ["matcher_host.cpp", "matcher_seq.cpp", "matcher_seq.hip", "matcher_group.cpp", "matcher_kernels.hip", "spa_host.cpp", "spa_symbolic.cpp", "spa_kernels.hip", "graph.hip", "occupancy.hip", "lifelong.hip", "comm.cpp", "mapper_host.cpp"]
FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-ffp-contract=off", "-fPIC",
         "-fvisibility=hidden", "-Wno-unused-value", "-shared", "-ldl"]


def hipcc() -> str:
    for cand in (shutil.which("hipcc"), "/opt/rocm/bin/hipcc"):
        if cand and os.path.exists(cand):
            return cand
    raise RuntimeError("hipcc not found: libkartohip.so needs the ROCm toolchain")


def needs_build() -> bool:
    if not os.path.exists(LIB):
        return True
    t = os.path.getmtime(LIB)
    deps = [os.path.join(CSRC, f) for f in os.listdir(CSRC)] + [os.path.join(HERE, "..", "include", "karto_hip.h")]
    return any(os.path.getmtime(d) > t for d in deps if os.path.exists(d))


def build(force: bool = False) -> str:
    """One object per source under build/obj (git-ignored), recompiled when the source or any header of csrc/ or include/ is
    newer; `force` recompiles everything.  Objects compile in parallel (KH_BUILD_JOBS, default 4)."""
    if not (force or needs_build()):
        return LIB
    from concurrent.futures import ThreadPoolExecutor
    objdir = os.path.join(HERE, "..", "build", "obj")
    os.makedirs(objdir, exist_ok=True)
    extra = os.environ.get("KH_EXTRA_HIPCC_FLAGS", "").split()
    cflags = [f for f in FLAGS if f not in ("-shared", "-ldl")] + extra
    headers = [os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith((".hpp", ".h"))]
    headers.append(os.path.join(HERE, "..", "include", "karto_hip.h"))
    hdr_time = max(os.path.getmtime(h) for h in headers if os.path.exists(h))
    stamp = os.path.join(objdir, "flags.txt")
    flags_now = " ".join(cflags)
    if not os.path.exists(stamp) or open(stamp).read() != flags_now:
        force = True
    srcs = [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
    jobs = []
    for s in srcs:
        src, obj = os.path.join(CSRC, s), os.path.join(objdir, s + ".o")
        if force or not os.path.exists(obj) or os.path.getmtime(obj) < max(os.path.getmtime(src), hdr_time):
            jobs.append([hipcc()] + cflags + ["-c", "-o", obj, src])
    with ThreadPoolExecutor(max_workers=max(1, int(os.environ.get("KH_BUILD_JOBS", "4")))) as pool:
        for rc in pool.map(subprocess.call, jobs):
            if rc != 0:
                raise subprocess.CalledProcessError(rc, "hipcc -c")
    open(stamp, "w").write(flags_now)
    subprocess.check_call([hipcc()] + FLAGS + extra + ["-o", LIB] + [os.path.join(objdir, s + ".o") for s in srcs])
    return LIB


if __name__ == "__main__":
    print(build(force=True))
