"""bench.py's launcher decision (`python bench.py --gpus N` must start its own ranks, or end in ONE JSON line with "error"
-- never a traceback): the pure function, and the real script on this GPU-less box."""
import json
import os
import subprocess
import sys

import pytest
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def _plan(*a, **k):
    import bench
    return bench.launch_plan(*a, **k)


def test_one_gpu_runs_in_process():
    assert _plan(1, {}, 1, []) == ("run", None)
    assert _plan(1, {}, 8, ["--steps", "20"]) == ("run", None)


def test_more_gpus_than_the_box_has_is_an_error_record():
    kind, text = _plan(2, {}, 1, [])
    assert kind == "error" and "--gpus 2" in text and "1 GPU" in text
    kind, text = _plan(1, {}, 0, [])
    assert kind == "error" and "needs a GPU" in text
    assert _plan(0, {}, 8, [])[0] == "error"


def test_n_gpus_without_a_launcher_spawns_one_rank_per_gpu():
    kind, cmd = _plan(4, {"MASTER_PORT": "29777"}, 8, ["--gpus", "4", "--steps", "20", "--warmup", "5"])
    assert kind == "spawn"
    assert cmd[0] == sys.executable and cmd[1:3] == ["-m", "torch.distributed.run"]
    assert "--nnodes=1" in cmd and cmd[cmd.index("--nproc-per-node") + 1] == "4"
    assert cmd[cmd.index("--master-addr") + 1] == "127.0.0.1" and cmd[cmd.index("--master-port") + 1] == "29777"
    i = cmd.index(os.path.join(ROOT, "bench.py"))
    assert cmd[i + 1:] == ["--gpus", "4", "--steps", "20", "--warmup", "5"]         # the script's own arguments, untouched


def test_under_a_launcher_the_process_is_a_rank():
    assert _plan(8, {"WORLD_SIZE": "8", "RANK": "3", "LOCAL_RANK": "3"}, 8, []) == ("run", None)
    assert _plan(1, {"WORLD_SIZE": "1", "RANK": "0", "LOCAL_RANK": "0"}, 1, []) == ("run", None)
    kind, text = _plan(4, {"WORLD_SIZE": "2"}, 8, [])
    assert kind == "error" and "WORLD_SIZE=2" in text
    kind, text = _plan(2, {"WORLD_SIZE": "2", "LOCAL_RANK": "1"}, 1, [])
    assert kind == "error" and "LOCAL_RANK 1" in text


@pytest.mark.skipif(torch.cuda.is_available(), reason="box has a GPU")
def test_the_script_itself_ends_in_one_json_error_line_here():
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 2, r.stderr[-2000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout
    rec = json.loads(lines[0])
    assert rec["n_gpus"] == 2 and rec["value"] is None and "error" in rec and "Traceback" not in r.stderr


@pytest.mark.gpu
def test_two_gpus_on_a_one_gpu_box_is_the_error_record_not_a_traceback():
    if torch.cuda.device_count() >= 2:
        pytest.skip("box has two GPUs: the launch itself is the driver's scale run")
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "20", "--warmup", "5"],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=300)
    assert r.returncode == 2
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1 and json.loads(lines[0])["n_gpus"] == 2 and "2 but 1 GPU" in json.loads(lines[0])["error"]
    assert "Traceback" not in r.stderr


@pytest.mark.gpu
def test_the_bench_line_keeps_the_drivers_contract():
    """`python bench.py --gpus 1 --steps 20 --warmup 5` (the driver's form, with the slow legs shortened): ONE JSON line on stdout with
    the contract's keys, the metric's configuration, `roofline` and `cpu_baseline` objects of the prescribed shape, and figures that
    are consistent with each other (value = steps / time, the dominant kernel's share below the step time, achieved below peak)."""
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    r = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "1", "--steps", "20", "--warmup", "5", "--extras", "0",
                        "--cpu-frames", "4", "--profile-frames", "16"], stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, env=env, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    lines = [l for l in r.stdout.splitlines() if l.strip()]
    assert len(lines) == 1, r.stdout[-2000:]
    d = json.loads(lines[0])
    for key in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
                "config", "roofline", "cpu_baseline"):
        assert key in d, key
    assert d["metric"] == "frames_per_sec" and d["unit"] == "frames/s" and d["n_gpus"] == 1 and d["steps"] == 20 and d["warmup"] == 5
    assert d["higher_is_better"] is True and d["vs_baseline"] is None and d["dtype"] == "f32" and d["data"] == "synthetic"
    c = d["config"]
    assert "workload" in c and c["width"] == 640 and c["height"] == 480 and 900000 < c["n_model"] < 1100000 and c["exchange"] == "none"
    assert abs(d["value"] * d["ms_per_step"] / 1000.0 - 1.0) < 1e-6 and d["value"] > 1000
    rf = d["roofline"]
    assert rf["bound"] == "hbm" and rf["unit"] == "GB/s" and rf["peak"] == 8000.0 and "traffic" in rf
    assert abs(rf["frac"] - rf["achieved"] / rf["peak"]) < 1e-9 and 0.02 < rf["frac"] < 1.0
    assert abs(rf["achieved"] - rf["algo_bytes_per_launch"] / (rf["avg_launch_us"] * 1e-6) / 1e9) < 1e-3 * rf["achieved"]
    assert sum(rf["kernel_share_ms_per_frame"].values()) < d["ms_per_step"] * 1.5      # (per-kernel brackets are taken in a separate, unpipelined-timer run)
    cb = d["cpu_baseline"]
    assert cb["kind"] == "port" and cb["unit"] == "frames/s" and cb["cores"] >= 1 and 0.5 < cb["value"] < d["value"] and "sample" in cb
    assert d["frame_roofline"]["frac"] < 1.0 and d["kernel_source_sha"]
    # round 5: what the README leads with sits INSIDE the objects the driver's record keeps whole, and the record carries a parity
    # block -- the cpu_baseline sample's frames replayed on the product, poses and counters against the oracle's
    assert abs(rf["frame_frac"] - d["frame_roofline"]["frac"]) < 1e-12
    for key in ("steady_state_frames_per_sec", "node_call_frames_per_sec", "sequential_ms_per_frame", "pipeline_depth", "extract_batch", "extract"):
        assert key in c, key
    assert c["sequential_ms_per_frame"] > 0.05 and c["extract"] == "single rank"
    par = cb["parity"]
    assert par == d["parity"] and "error" not in par, par
    assert par["frames"] == 4 and par["frames_bit_equal"] == 4 and par["frames_counters_equal"] == 4 and par["max_abs_pose_diff"] == 0.0 and par["within_tolerance"] is True
