"""The library's host worker pool (slam_toolbox_amd/csrc/host_pool.hpp) under a stress program: regions from two caller
threads, back to back and with the workers asleep in between; every index exactly once.  No GPU."""
import os
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_pool_stress(tmp_path):
    exe = str(tmp_path / "host_pool_stress")
    subprocess.run(["g++", "-O2", "-std=c++17", "-pthread", os.path.join(HERE, "host_pool_stress.cpp"), "-o", exe], check=True)
    for threads in ("8", "3"):
        env = dict(os.environ, KH_HOST_THREADS=threads)
        r = subprocess.run([exe], env=env, capture_output=True, text=True, timeout=300)
        sys.stdout.write(r.stdout)
        assert r.returncode == 0, r.stdout + r.stderr
