"""The C ABI from plain C: examples/match_scan.c must compile against include/karto_hip.h with gcc and link
libkartohip.so (CPU), and on a GPU run and improve the query pose."""
import os
import shutil
import subprocess

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _build(tmp_path):
    exe = str(tmp_path / "match_scan")
    lib_dir = os.path.join(ROOT, "slam_toolbox_amd")
    cmd = ["gcc", "-std=c99", "-O2", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "examples", "match_scan.c"), "-L", lib_dir, "-lkartohip", "-lm",
           "-Wl,-rpath," + lib_dir, "-o", exe]
    res = subprocess.run(cmd, capture_output=True, text=True)
    assert res.returncode == 0, res.stderr[-3000:]
    return exe


@pytest.mark.skipif(shutil.which("gcc") is None, reason="gcc not available")
def test_example_compiles_and_links_with_gcc(kartohip_lib, tmp_path):
    exe = _build(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True)
    if kartohip_lib.kh_device_count() < 1:
        assert res.returncode == 2 and "no GPU" in res.stderr        # no CPU fallback: it says so and stops


@pytest.mark.gpu
def test_example_runs_on_the_gpu(kartohip_lib, tmp_path):
    exe = _build(tmp_path)
    res = subprocess.run([exe], capture_output=True, text=True)
    assert res.returncode == 0, (res.stdout, res.stderr)
    assert res.stdout.startswith("response ")
