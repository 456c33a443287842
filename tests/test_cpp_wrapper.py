"""include/ssf.hpp (the header-only C++ surface with the reference's method names) compiled with g++ and run against
the checker library on CPU: it must build without OpenCV / HIP headers and give what the Python mirror gives."""
import os
import subprocess

import numpy as np
import pytest

import util
from conftest import ORACLE_LIB, ROOT
from supersurfel_fusion_amd import binding, synthetic


def test_cpp_wrapper_builds_and_matches_the_python_mirror(oracle_lib, tmp_path):
    _run(oracle_lib, ORACLE_LIB, "ssf_oracle", tmp_path)


@pytest.mark.gpu
def test_cpp_wrapper_on_the_hip_library(oracle_lib, product_lib, tmp_path):
    """the same C++ program linked against libssf_hip.so on the GPU box, checked against the oracle"""
    _run(oracle_lib, os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "libssf_hip.so"), "ssf_hip", tmp_path)


def _run(oracle_lib, lib_path, lib_name, tmp_path):
    W, H, n = 160, 128, 3
    frames = [util.frame(k, W, H) for k in range(n)]
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for rgb, depth in frames:
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes()); f.write(np.ascontiguousarray(depth, np.float32).tobytes())
    exe = tmp_path / "wrapper_smoke"
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "wrapper_smoke.cpp"), "-o", str(exe), "-L", libdir, "-l" + lib_name,
           "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    K = synthetic.intrinsics(W, H)
    r = subprocess.run([str(exe), str(W), str(H), str(n), str(raw)] + [repr(float(K[k])) for k in ("fx", "fy", "cx", "cy")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    # the same arguments as wrapper_smoke.cpp's initialize(): reference defaults but for four energy weights
    fo = binding.Fusion(oracle_lib, oracle_lib.default_config(nb_supersurfels_max=50000, lambda_pos=10.0, lambda_bound=1000.0,
                                                              lambda_size=1000.0, lambda_disp=1e8,
                                                              **{k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}))
    for k, (rgb, depth) in enumerate(frames):
        res = fo.process_frame(rgb, depth)
        tok = lines[k].split()
        assert tok[0] == "frame" and int(tok[1]) == k
        assert tok[2] == "n=%d" % res["n_model"] and tok[3] == "vis=%d" % res["n_visible"], (lines[k], res)
        assert tok[5] == "icp=%d/%d" % (res["icp_valid"], res["icp_iters"])
        pose = np.array([float(x) for x in tok[7:19]], np.float32)
        assert np.array_equal(pose.view(np.uint32), res["pose"].astype(np.float32).view(np.uint32)), lines[k]
    m = fo.get_model()
    tok = lines[n].split()
    assert tok[0] == "model" and int(tok[1]) == len(m["confidences"]) and int(tok[5]) == W * H
    assert abs(float(tok[3]) - float(np.sum(m["positions"].astype(np.float64)))) < 1e-3 * max(1.0, abs(float(tok[3])))
    assert lines[n + 1].startswith("sequence n=%d" % n)
    assert lines[n + 2] == "sequence_equals_frames 1"
    assert lines[n + 3] == "uninitialised_throws 1"
