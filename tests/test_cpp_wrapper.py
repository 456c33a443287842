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


def _run_cv(oracle_lib, lib_path, lib_name, tmp_path):
    """tests/cpp/wrapper_cv_smoke.cpp: the cv::Mat overloads (against the cv::Mat test double), the two image getters
    of supersurfel_fusion.hpp:79-80, getModelDevice and initialize()'s depth_prefilter switch."""
    W, H, n = 160, 128, 3
    frames = [util.frame(k, W, H) for k in range(n)]
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for rgb, depth in frames:
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes()); f.write(np.ascontiguousarray(depth, np.float32).tobytes())
    exe = tmp_path / "wrapper_cv_smoke"
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
           os.path.join(ROOT, "tests", "cpp", "wrapper_cv_smoke.cpp"), "-o", str(exe), "-L", libdir, "-l" + lib_name, "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    K = synthetic.intrinsics(W, H)
    r = subprocess.run([str(exe), str(W), str(H), str(n), str(raw)] + [repr(float(K[k])) for k in ("fx", "fy", "cx", "cy")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    out = dict(l.split(" ", 1) for l in r.stdout.strip().splitlines())
    fo = binding.Fusion(oracle_lib, oracle_lib.default_config(nb_supersurfels_max=50000, lambda_pos=10.0, lambda_bound=1000.0, lambda_size=1000.0,
                                                              lambda_disp=1e8, depth_prefilter=0, **{k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}))
    for rgb, depth in frames:
        res = fo.process_frame(rgb, depth)
    seg = fo.preview_image()
    h = 0
    for v in seg.reshape(-1).tolist():
        h = (h * 1315423911 + v) & 0xFFFFFFFFFFFFFFFF
    assert out["seg"] == "%dx%d type=16 hash=%d" % (W, H, h)
    # the preview: boundaries white, elsewhere 0.8 x the colour in B, G, R order
    lab = fo.index_map()
    edge = np.zeros((H, W), bool); edge[:-1, :-1] = (lab[:-1, 1:] != lab[:-1, :-1]) | (lab[1:, 1:] != lab[:-1, :-1])
    assert (seg[edge] == 255).all() and 0.05 < edge.mean() < 0.5
    assert np.array_equal(seg[~edge], (np.float32(0.8) * frames[-1][0][~edge][:, ::-1].astype(np.float32)).astype(np.uint8))
    pd = fo.plane_depth()
    ok = np.isfinite(pd) & (pd > 0) & (pd < 100)
    tok = out["plane"].split()
    assert tok[0] == "%dx%d" % (W, H) and tok[1] == "type=5" and tok[2] == "finite=%d" % int(ok.sum())
    assert abs(float(tok[3].split("=")[1]) - float(pd[ok].astype(np.float64).sum())) < 1e-3 * float(ok.sum())
    assert out["same_as_raw"] == "1"
    assert out["model_device"] == "n=%d ptrs=1" % res["n_model"]
    assert out["n=%d" % res["n_model"]] == "vis=%d" % res["n_visible"]


def test_cv_mat_surface_builds_and_runs(oracle_lib, tmp_path):
    _run_cv(oracle_lib, ORACLE_LIB, "ssf_oracle", tmp_path)


@pytest.mark.gpu
def test_cv_mat_surface_on_the_hip_library(oracle_lib, product_lib, tmp_path):
    _run_cv(oracle_lib, os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "libssf_hip.so"), "ssf_hip", tmp_path)


def _run_node_call(lib_path, lib_name, tmp_path):
    """tests/cpp/node_initialize_call.cpp: the reference nodes' literal 29-argument initialize() call
    (node/supersurfel_fusion_node.cpp:256-284) compiles against include/ssf.hpp, and the eight sparse-VO / loop-closure /
    MOD arguments do not reach any library knob: same frames as the POD configuration with the same 21 path values."""
    W, H, n = 160, 128, 3
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for k in range(n):
            rgb, depth = util.frame(k, W, H)
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes()); f.write(np.ascontiguousarray(depth, np.float32).tobytes())
    exe = tmp_path / "node_initialize_call"
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"),
           os.path.join(ROOT, "tests", "cpp", "node_initialize_call.cpp"), "-o", str(exe), "-L", libdir, "-l" + lib_name, "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    K = synthetic.intrinsics(W, H)
    r = subprocess.run([str(exe), str(W), str(H), str(n), str(raw)] + [repr(float(K[k])) for k in ("fx", "fy", "cx", "cy")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    assert lines[0] == "defaults_are_the_references 1", r.stdout
    assert lines[1].startswith("node_call_equals_pod_config 1 n="), r.stdout
    assert int(lines[1].split("n=")[1]) > 0


def test_the_reference_nodes_initialize_call_compiles_and_means_the_same(oracle_lib, tmp_path):
    _run_node_call(ORACLE_LIB, "ssf_oracle", tmp_path)


@pytest.mark.gpu
def test_the_reference_nodes_initialize_call_on_the_hip_library(product_lib, tmp_path):
    _run_node_call(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "libssf_hip.so"), "ssf_hip", tmp_path)


def _run_node_pose(oracle_lib, lib_path, lib_name, tmp_path):
    """tests/cpp/node_pose_use.cpp: the pose block of both reference nodes (node/supersurfel_fusion_node.cpp:87-91,
    node/supersurfel_fusion_rgbd_benchmark_node.cpp:616-620), verbatim, against include/ssf.hpp and a tf:: double: the nine +
    three values a node would broadcast are ssf_get_pose's, and those are the oracle's."""
    W, H, n = 640, 480, 3          # at 640x480 the ICP is accepted from frame 1 on: the pose the node reads has moved
    frames = [util.frame(k, W, H) for k in range(n)]
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for rgb, depth in frames:
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes()); f.write(np.ascontiguousarray(depth, np.float32).tobytes())
    exe = tmp_path / "node_pose_use"
    libdir = os.path.dirname(lib_path)
    cmd = ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror", "-I", os.path.join(ROOT, "include"), "-I", os.path.join(ROOT, "tests", "cpp"),
           os.path.join(ROOT, "tests", "cpp", "node_pose_use.cpp"), "-o", str(exe), "-L", libdir, "-l" + lib_name, "-Wl,-rpath," + libdir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    K = synthetic.intrinsics(W, H)
    r = subprocess.run([str(exe), str(W), str(H), str(n), str(raw)] + [repr(float(K[k])) for k in ("fx", "fy", "cx", "cy")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=50000, depth_prefilter=1))
    moved = False
    for k, (rgb, depth) in enumerate(frames):
        res = fo.process_frame(rgb, depth)
        tok = lines[k].split()
        assert tok[:4] == ["frame", str(k), "node_pose_equals_abi", "1"], lines[k]
        pose = np.array([float(x) for x in tok[5:17]], np.float32)
        assert np.array_equal(pose.view(np.uint32), res["pose"].astype(np.float32).view(np.uint32)), lines[k]
        moved = moved or not np.array_equal(pose, np.array([1, 0, 0, 0, 1, 0, 0, 0, 1, 0, 0, 0], np.float32))
    assert moved, "the test sequence never left the identity pose"


def test_the_reference_nodes_pose_lines_compile_and_read_the_pose(oracle_lib, tmp_path):
    _run_node_pose(oracle_lib, ORACLE_LIB, "ssf_oracle", tmp_path)


@pytest.mark.gpu
def test_the_reference_nodes_pose_lines_on_the_hip_library(oracle_lib, product_lib, tmp_path):
    _run_node_pose(oracle_lib, os.path.join(ROOT, "supersurfel_fusion_amd", "csrc", "libssf_hip.so"), "ssf_hip", tmp_path)


def test_ssf_hpp_uses_hips_float3_when_the_translation_unit_has_one():
    """tests/cpp/node_pose_hip_float3.cpp (compile-only): with <hip/hip_vector_types.h> included first, include/ssf.hpp builds
    Transform3 / Mat33 on HIP's float3 instead of declaring its own -- the guard the header documents"""
    inc = "/opt/rocm/include"
    if not os.path.exists(os.path.join(inc, "hip", "hip_vector_types.h")):
        pytest.skip("no ROCm headers on this box")
    cmd = ["g++", "-std=c++14", "-D__HIP_PLATFORM_AMD__", "-Wall", "-Werror", "-I", inc, "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "cpp"), "-c", os.path.join(ROOT, "tests", "cpp", "node_pose_hip_float3.cpp"), "-o", os.devnull]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-3000:]


HIPCC = "/opt/rocm/bin/hipcc"


def _build_node_model_copy(lib_dir, lib_name, exe):
    """tests/cpp/node_model_copy.cpp by hipcc against rocThrust: the nodes' thrust copy-out lines of getModel() / getFrame(), verbatim"""
    if not (os.path.exists(HIPCC) and os.path.exists("/opt/rocm/include/thrust/host_vector.h")):
        pytest.skip("no hipcc / rocThrust on this box")
    cmd = [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1", "-Wall", "-Werror", "-I", os.path.join(ROOT, "include"),
           "-I", os.path.join(ROOT, "tests", "cpp"), os.path.join(ROOT, "tests", "cpp", "node_model_copy.cpp"), "-o", str(exe),
           "-L", lib_dir, "-l" + lib_name, "-Wl,-rpath," + lib_dir]
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout[-4000:]


def test_the_reference_nodes_model_copy_lines_compile_against_rocthrust(product_lib, tmp_path):
    """`thrust::host_vector<float3> positions(ssf.getModel().positions.begin(), ... + ssf.getnbSupersurfels());` and
    `thrust::host_vector<Mat33> orientations(ssf.getFrame().orientations);` (supersurfel_fusion_node.cpp:306-310,423-427;
    ...benchmark_node.cpp:189-193) compile as they stand when thrust is there: getModel() / getFrame() are `const Supersurfels&`
    views of device arrays then (include/ssf.hpp).  Compile + link here; the GPU twin runs it."""
    _build_node_model_copy(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc"), "ssf_hip", tmp_path / "node_model_copy")


@pytest.mark.gpu
def test_the_reference_nodes_model_copy_lines_on_the_hip_library(product_lib, tmp_path):
    """... and copy out exactly what ssf_get_model / ssf_get_frame hand to the host: every array, bit for bit"""
    W, H, n = 640, 480, 3
    raw = tmp_path / "frames.bin"
    with open(raw, "wb") as f:
        for k in range(n):
            rgb, depth = util.frame(k, W, H)
            f.write(np.ascontiguousarray(rgb, np.uint8).tobytes()); f.write(np.ascontiguousarray(depth, np.float32).tobytes())
    exe = tmp_path / "node_model_copy"
    _build_node_model_copy(os.path.join(ROOT, "supersurfel_fusion_amd", "csrc"), "ssf_hip", exe)
    K = synthetic.intrinsics(W, H)
    r = subprocess.run([str(exe), str(W), str(H), str(n), str(raw)] + [repr(float(K[k])) for k in ("fx", "fy", "cx", "cy")], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    lines = r.stdout.strip().splitlines()
    assert lines[0].startswith("model_range_form 1 n=") and int(lines[0].split("n=")[1]) > 500, r.stdout
    assert lines[1] == "model_whole_array_form 1" and lines[2].startswith("frame_whole_array_form 1 valid="), r.stdout


def test_ssf_hpp_layout_does_not_depend_on_the_include_order(tmp_path):
    """tests/cpp/layout_probe.cpp built twice (hipcc + rocThrust first / plain g++): the same sizeof(SupersurfelFusion), views and
    pose types in both translation units (advisor, round 4: the view members used to exist only behind THRUST_VERSION -- two
    files of one node disagreed about the object's size), and a different inline-namespace tag, so mixing the forms cannot link."""
    if not (os.path.exists(HIPCC) and os.path.exists("/opt/rocm/include/thrust/host_vector.h")):
        pytest.skip("no hipcc / rocThrust on this box")
    src = os.path.join(ROOT, "tests", "cpp", "layout_probe.cpp")
    out = {}
    for form, cmd in (("thrust", [HIPCC, "-x", "hip", "--offload-arch=gfx950", "-std=c++17", "-O1", "-DLAYOUT_WITH_THRUST"]),
                      ("plain", ["g++", "-std=c++11", "-O1", "-Wall", "-Wextra", "-Werror"])):
        exe = tmp_path / ("layout_" + form)
        r = subprocess.run(cmd + ["-I", os.path.join(ROOT, "include"), src, "-o", str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        assert r.returncode == 0, r.stdout[-4000:]
        r = subprocess.run([str(exe)], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)       # (prints sizes only: no device, no library call)
        assert r.returncode == 0, r.stdout
        out[form] = dict(t.split("=") for t in r.stdout.split())
    assert out["thrust"]["form"] == "1" and out["plain"]["form"] == "0"
    for k in ("fusion", "views", "array", "transform", "float3"):
        assert out["thrust"][k] == out["plain"][k], (k, out)
    assert out["plain"]["float3"] == "12" and out["plain"]["transform"] == "48"


def test_model_device_view_has_the_reference_layout(oracle_lib):
    """ssf_get_model_device: packed Mat33 orientations (9 floats per row), rows [visible | out of view]"""
    import ctypes as C
    W, H = 160, 128
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    for k in range(2):
        fo.process_frame(*util.frame(k, W, H))
    st, n = fo.model_device()
    m = fo.get_model()
    assert n == len(m["confidences"]) > 0
    ori = np.ctypeslib.as_array(C.cast(st.orientations, C.POINTER(C.c_float)), shape=(n, 9))
    pos = np.ctypeslib.as_array(C.cast(st.positions, C.POINTER(C.c_float)), shape=(n, 3))
    assert np.array_equal(ori, m["orientations"]) and np.array_equal(pos, m["positions"])
