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
    if force or needs_build():
        srcs = [os.path.join(CSRC, s) for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]
        cmd = [hipcc()] + FLAGS + os.environ.get("KH_EXTRA_HIPCC_FLAGS", "").split() + ["-o", LIB] + srcs
        subprocess.check_call(cmd)
    return LIB


if __name__ == "__main__":
    print(build(force=True))
