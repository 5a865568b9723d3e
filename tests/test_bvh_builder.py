"""Host-side BVH builder of the depth / lidar ray casters (stretch_mujoco_amd/csrc/smj_bvh.h): layout invariants and nearest-hit
equality with brute force on random meshes, through a small C++ harness (tests/bvh/bvh_check.cpp).  CPU only."""
import os
import subprocess

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_bvh_builder_invariants(tmp_path):
    exe = tmp_path / "bvh_check"
    subprocess.check_call(["g++", "-O1", "-std=c++17", "-I", os.path.join(ROOT, "stretch_mujoco_amd", "csrc"),
                           os.path.join(ROOT, "tests", "bvh", "bvh_check.cpp"), "-o", str(exe)])
    out = subprocess.run([str(exe)], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and out.stdout.strip().endswith("ok"), out.stdout + out.stderr
