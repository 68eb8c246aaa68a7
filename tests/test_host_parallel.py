"""The host-side helpers behind ba_set_problem's threaded set-up (visual-odometry_amd/csrc/host_parallel.h) on the CPU:
built with g++ and run as a plain program (no GPU, no HIP): the threaded merge sort against std::sort at every thread
count, the range partition, the exception path.  No reference counterpart."""
import os
import shutil
import subprocess

import pytest

HERE = os.path.dirname(os.path.abspath(__file__))


def test_host_parallel_helpers(tmp_path):
    gxx = shutil.which("g++")
    if gxx is None:
        pytest.skip("no g++")
    exe = os.path.join(tmp_path, "host_parallel_check")
    subprocess.run([gxx, "-O2", "-std=c++17", "-pthread", "-o", exe, os.path.join(HERE, "cpu", "host_parallel_check.cpp")],
                   check=True, timeout=120)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "checks passed" in out.stdout
