#!/usr/bin/env python
"""bench.py -- frames/s of the supersurfel hot path (extract | ICP | fuse) on MI355X.

Workload = the configuration BASELINE.json's metric is quoted on: a 640x480 synthetic RGB-D orbit
against a map of ~1 M live supersurfels (seeded on the scene's surfaces, seed 1234), reference
rgbd_benchmark parameters with sparse VO / MOD / loop closure off, ICP with
the reference's own early stop (icp_iter = 10).  A "step" is one frame.  Frames are rendered before
the timed region and are resident in HBM when it starts.

  python bench.py [--gpus N] [--steps K] [--warmup W]
  N > 1: launched by torch.distributed.run, one rank per GPU; the map is sharded by world-space
  tile, the frame is replicated, ICP / association are exchanged over RCCL ("strong" scaling:
  the total map size is fixed).

Prints ONE JSON line (rank 0) with the contract keys plus "roofline" (dominant kernel, measured
live with hipEvents on the library's stream) and "cpu_baseline" (the CPU oracle = a single-threaded
port of the reference algorithm, timed on this box's host cores on a bounded sample)."""
import argparse
import ctypes
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

os.environ.setdefault("NCCL_DEBUG_FILE", "/dev/stderr")   # RCCL logs to stdout by default: keep stdout for the one JSON line
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

from supersurfel_fusion_amd import binding, sharded, synthetic  # noqa: E402

W, H = 640, 480
N_MODEL = 1000000
PARAMS = dict(lambda_pos=10.0, lambda_bound=1000.0, lambda_size=1000.0, lambda_disp=1e8, thresh_disp=1e-4,
              seg_iter=10, filter_iter=3, delta_t=20, conf_thresh=2560.0, icp_iter=10, icp_cov_thresh=0.05)
HBM_PEAK_GBS = 8000.0   # MI355X HBM3E spec peak (MI355X_MICROARCH.md); the copy ceiling of the box is measured live: measured_hbm_peak()

# algorithmic bytes per unit of each kernel (DESIGN.md "Kernels")
P = W * H
ALGO_BYTES = {
    # stable partition: only the rows that change place move (visible rows are compacted, view changes cross over)
    "reorder_move": lambda c: 2.0 * c["n_model"] + 208.0 * c["n_visible"],     # state + live byte per slot, visible rows read + written
    # the same launch when it also accumulates the next frame's first ICP iteration (rows are read once for both)
    "reorder_move_icp": lambda c: 2.0 * c["n_model"] + 208.0 * c["n_visible"] + 8.0 * P + 28.0 * c["S"],
    # fuse launch: classification of every row (pos 12 + stamps 8 + conf 4 read, state written) + candidate per visible row
    "update_insert": lambda c: 28.0 * c["n_model"] + 4.0 * c["n_visible"],
    "icp_accumulate": lambda c: 36.0 * c["n_visible"] + 8.0 * P + 28.0 * c["S"],
    "match": lambda c: 44.0 * c["n_visible"],
    "update_pass_rgb": lambda c: c["batch"] * 9.0 * P,
    "update_pass_rgbd": lambda c: c["batch"] * 14.0 * P,
    # the passes of a phase in one launch (k_passes_team): the same bytes per pass and frame, c["passes"] passes per launch
    "passes_team_rgb": lambda c: c["batch"] * 9.0 * P * c["passes"],
    "passes_team_rgbd": lambda c: c["batch"] * 14.0 * P * c["passes"],
    "ingest": lambda c: c["batch"] * 23.0 * P,
    "init_disp": lambda c: c["batch"] * 9.0 * P,
    "eval_samples": lambda c: c["batch"] * 8.0 * P,
    "render_moments": lambda c: c["batch"] * 25.0 * P,
    # depth pre-filter (config 5 / depth_prefilter = 1): the frames of a batch per launch, 4 B read + 4 B written per pixel; the kernel is
    # bound by its 149 taps x a specified exp per pixel, not by these bytes
    "bilateral_prefilter": lambda c: c["batch"] * 8.0 * P,
}


EXTRACT_KERNELS = ("update_pass_rgb", "update_pass_rgbd", "passes_team_rgb", "passes_team_rgbd", "ingest", "init_disp", "eval_samples", "render_moments")


def strip_comments(src):
    """C++ source without comments and with runs of white space collapsed (string and character literals kept)"""
    out, i, n = [], 0, len(src)
    while i < n:
        c = src[i]
        if c == '"' or c == "'":
            j = i + 1
            while j < n and src[j] != c:
                j += 2 if src[j] == "\\" else 1
            out.append(src[i:j + 1]); i = j + 1
        elif src.startswith("//", i):
            j = src.find("\n", i)
            i = n if j < 0 else j
        elif src.startswith("/*", i):
            j = src.find("*/", i + 2)
            i = n if j < 0 else j + 2
            out.append(" ")
        else:
            out.append(c); i += 1
    return " ".join("".join(out).split())


def kernel_source_sha():
    """hash of the product's kernel sources (comments and white space aside: they do not reach the binary): ties a PMC
    profile under profiles/ to the code it was taken from"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "supersurfel_fusion_amd", "csrc")
    for f in ("ssf_extract.hip", "ssf_pass_tile.hpp", "ssf_track_fuse.hip", "ssf_tile_rows.inc", "ssf_host.hip", "ssf_device.hpp", "ssf_math.hpp"):
        h.update(strip_comments(open(os.path.join(d, f), "r").read()).encode())
    return h.hexdigest()[:16]


def measured_hbm_peak(dev, mib=1024, reps=12):
    """What a plain device-to-device stream copy sustains on THIS box ("report both nominal and
    measured-achievable"): bytes read + bytes written per second of torch's copy kernel over two 1 GiB buffers, best of
    `reps` (MI355X_MICROARCH.md: ~6.3 TB/s of the 8 TB/s spec)."""
    n = mib * (1 << 20) // 4
    a_, b_ = torch.empty(n, dtype=torch.float32, device=dev), torch.empty(n, dtype=torch.float32, device=dev)
    a_.fill_(1.0); b_.copy_(a_)
    torch.cuda.synchronize(dev)
    best = 0.0
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); b_.copy_(a_); e1.record()
        e1.synchronize()
        best = max(best, 2.0 * n * 4 / (e0.elapsed_time(e1) * 1e-3) / 1e9)
    del a_, b_
    return best


def make_cfg(lib, cap, rank=0, nranks=1, stream=None, force_icp=False, pipeline_depth=0, extract_batch=1, prefilter=0):
    """prefilter = 0: the metric's path starts at "depth after the pre-filter" (the
    bilateral filter is OpenCV's, third party, a "next" row); the rate WITH the library's own filter inside the frame
    is reported beside it (extra key "with_depth_prefilter")."""
    K = synthetic.intrinsics(W, H)
    kw = dict({k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}, nb_supersurfels_max=cap,
              rank=rank, nranks=nranks, icp_force_iters=1 if force_icp else 0, pipeline_depth=pipeline_depth,
              extract_batch=extract_batch, depth_prefilter=prefilter)
    kw.update(PARAMS)
    if stream is not None:
        kw["stream"] = stream
    # one process per GPU: the library works on the device this process has selected (LOCAL_RANK under torchrun)
    if lib.backend.startswith("hip") and torch.cuda.is_available():
        kw["device_id"] = torch.cuda.current_device()
    return lib.default_config(**kw)


def next_kernel_times(lib, dev, model, nvis, cap):
    """The "next" rows of SURVEY.md section 8f on their own: achieved GB/s of the deformation apply (176 B per
    supersurfel), the depth pre-filter (8 B per pixel) and one loop-closure registration call, hipEvent-timed by the
    library (cfg.profile = 1)."""
    out = {}
    rng = np.random.default_rng(5)
    f = binding.Fusion(lib, make_cfg(lib, cap))
    f.set_model(model, nvis, 30)
    n = len(model["confidences"]); m = max(n // 50, 4)
    npos = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    nrot = np.tile(np.eye(3, dtype=np.float32).reshape(1, 9), (m, 1)); ntr = rng.uniform(-1e-3, 1e-3, (m, 3)).astype(np.float32)
    w4 = rng.dirichlet(np.ones(4), n).astype(np.float32); idx = rng.integers(0, m, (n, 4)).astype(np.int32)
    f.set_profile(1)
    # two node assignments: uniformly random indices (the worst case for the node gathers; this key's meaning since round 2) and
    # indices that follow the row order (a row's four nodes near row / 50: what a time-ordered deformation graph gives rows in
    # arrival order -- the reference samples its nodes from the model and weights them sequentially, deformation_graph.cu:240-270)
    idx_seq = np.clip((np.arange(n)[:, None] // 50) + rng.integers(-2, 3, (n, 4)), 0, m - 1).astype(np.int32)
    for key, ix in (("apply_deformation", idx), ("apply_deformation_nodes_in_row_order", idx_seq)):
        for rep in range(3):
            if rep == 1:
                f.reset_kernel_times()
            f.apply_deformation(npos, nrot, ntr, w4, ix)
        ms, calls = f.kernel_times().get("apply_deformation", (0.0, 0))
        if calls:
            us = 1000.0 * ms / calls
            out[key] = dict(rows=n, nodes=m, avg_us=us, algo_bytes=176.0 * n, achieved_GBs=176.0 * n / (us * 1e-6) / 1e9,
                            frac_of_hbm_peak=176.0 * n / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                            node_indices="uniformly random" if key == "apply_deformation" else "near row / 50 (rows in arrival order, time-ordered graph)")
    R, t = synthetic.orbit_pose(0)
    rgb, depth, _ = synthetic.render(R, t, W, H, noise=True, rng=np.random.default_rng(1000))
    d_in = torch.from_numpy(depth).to(dev); d_out = torch.empty_like(d_in)
    import ctypes as C
    f.reset_kernel_times()
    for rep in range(6):
        if rep == 2:
            f.reset_kernel_times()
        f._ck(lib.lib.ssf_bilateral_filter(f.h, C.c_void_p(d_in.data_ptr()), C.c_void_p(d_out.data_ptr()), 1), "ssf_bilateral_filter")
    ms, calls = f.kernel_times().get("bilateral_prefilter", (0.0, 0))
    if calls:
        us = 1000.0 * ms / calls
        out["bilateral_prefilter"] = dict(width=W, height=H, taps=149, avg_us=us, algo_bytes=8.0 * W * H, achieved_GBs=8.0 * W * H / (us * 1e-6) / 1e9,
                                          frac_of_hbm_peak=8.0 * W * H / (us * 1e-6) / 1e9 / HBM_PEAK_GBS,
                                          note="compute bound: 149 taps (circle of radius 7) x a specified (bit-reproducible) exp per pixel, two taps per packed fp32 instruction")
        # "compute bound" as a number (round 6): vector instructions issued per launch (rocprofv3 --pmc SQ_INSTS_VALU) over the launch's
        # duration in the kernel trace, against the part's vector issue peak (tools/pmc_issue.py -> profiles/pmc_rNN_config5_issue.json;
        # used only when it was recorded at these kernel sources)
        import glob
        iss = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r[0-9][0-9]_config5_issue.json")))
        if iss:
            ij = json.load(open(iss[-1]))
            ke = next((v for k, v in ij.get("kernels", {}).items() if k.startswith("k_bilateral_r7")), None)
            if ke and ij.get("source_sha") == kernel_source_sha():
                out["bilateral_prefilter"].update(valu_issue_frac=ke["valu_issue_frac"], valu_wave_insts_per_launch=ke["valu_wave_insts_per_launch"],
                                                  valu_issue_note="profiles/%s: SQ_INSTS_VALU per launch / trace duration / (256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles)" % os.path.basename(iss[-1]))
            else:
                out["bilateral_prefilter"]["valu_issue_note"] = "profiles/%s was recorded at other kernel sources" % os.path.basename(iss[-1])
    f.process_frame(rgb, depth)
    src = f.get_frame()
    f.reset_kernel_times()
    res = f.align(src)
    ms, calls = f.kernel_times().get("align_iteration", (0.0, 0))
    if calls:
        out["align"] = dict(sources=int(f.S), iterations=int(res["iters"]), avg_us_per_iteration=1000.0 * ms / calls, pairs=int(res["pairs"]),
                            note="one single-workgroup launch per iteration (LDS-sized problem): latency, not bandwidth")
    f.close()
    return out


def drive_pipelined(eng, frame_at, first, count, on_device=True):
    """The submit-ahead / process-in-order loop over frames [first, first + count) through an engine's own entry points (a
    binding.Fusion, or the torch.distributed driver sharded.ShardedFusion of `--py-driver` / the fall-through at N > 1): as many
    frames as the extract pipeline takes are submitted ahead, then the oldest one is tracked and fused.  frame_at(i) -> (rgb, depth)
    of frame i (device addresses or host arrays).  Module level so that tests/test_sharded.py can run THIS loop at world size 2 over
    gloo on the CPU checker -- the launcher path of an N > 1 run has then executed somewhere before it meets hardware."""
    res, nsub = [], first
    for i in range(first, first + count):
        while nsub < first + count and eng.can_submit():
            rgb, depth = frame_at(nsub)
            eng.submit_frame(rgb, depth, on_device=on_device)
            nsub += 1
        r = eng.process_submitted()
        res.append(r if isinstance(r, dict) else r.as_dict())
    return res


def real_frames_leg(lib, dev, model, nvis, cap):
    """the path TIMED on real frames (every other timing of this script is the synthetic box room).  The 8
    committed TUM fr1_xyz frames (tests/golden/tum_fr1_xyz_8frames.npz: cluttered desk, ~25 % depth holes, u16 depth at 5000 / m)
    swept back and forth as the rendered ones are, resident in HBM, depth pre-filter ON, the benchmark node's launch parameters
    (replay.BENCHMARK_LAUNCH; reference caller: node/supersurfel_fusion_rgbd_benchmark_node.cpp:573-744), pipelined 2 x 12 -- (a) on
    the map the frames grow themselves and (b) against the seeded ~1 M-row map of the headline (whose walls the desk frames do not
    show: the ICP is rejected and the map is only streamed -- what the kernels cost, not a trajectory).  Beside the rates: the stage
    split, the dominant kernel's launch time, ICP iterations, and -- from the LAB build of the same sources, one frame in flight --
    the relabelling statistics the box room could hide: log entries per tile and pass (the log region holds 256), and the share of
    superpixel-row lookups whose label lay outside the tile's 7 x 7-cell LDS window (the exact global path)."""
    from supersurfel_fusion_amd import replay
    path = os.path.join(ROOT, "tests", "golden", "tum_fr1_xyz_8frames.npz")
    if not os.path.exists(path):
        return dict(error="tests/golden/tum_fr1_xyz_8frames.npz is missing")
    fr = [(np.ascontiguousarray(rgb), np.ascontiguousarray(depth, np.float32)) for _, rgb, depth in replay.frames_from_npz(path)]
    nr = len(fr)
    t_rgb = [torch.from_numpy(a_).to(dev) for a_, _ in fr]; t_dep = [torch.from_numpy(b_).to(dev) for _, b_ in fr]
    sweep = lambda i: (i % (2 * nr - 2)) if (i % (2 * nr - 2)) < nr else 2 * nr - 2 - (i % (2 * nr - 2))      # noqa: E731
    out = dict(frames="8 x TUM fr1_xyz (640x480), swept back and forth, resident in HBM; depth pre-filter on; rgbd_benchmark launch parameters",
               holes_share=float(np.mean([float((d_ == 0).mean()) for _, d_ in fr])))
    bx, dpt, n_timed, n_warm = 12, 2, 480, 60

    def cfg_for(L, cap_, depth_, batch_):
        kw = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=cap_, pipeline_depth=depth_, extract_batch=batch_)
        if L.backend.startswith("hip") and torch.cuda.is_available():
            kw["device_id"] = torch.cuda.current_device()
        return L.default_config(**kw)

    for key, seeded in (("map_grown_by_the_frames", False), ("against_the_seeded_map", True)):
        f = binding.Fusion(lib, cfg_for(lib, cap if seeded else 100000, dpt, bx))
        if seeded:
            f.set_model(model, nvis, 30)
        seq = lambda a0, n_: f.prepare_sequence([t_rgb[sweep(i)].data_ptr() for i in range(a0, a0 + n_)], [t_dep[sweep(i)].data_ptr() for i in range(a0, a0 + n_)])      # noqa: E731
        f.process_prepared(seq(0, n_warm), on_device=True)
        prep = seq(n_warm, n_timed)
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        res = [r.as_dict() for r in f.process_prepared(prep, on_device=True)]
        torch.cuda.synchronize(dev)
        dt_ = time.perf_counter() - t1
        ent = dict(frames_per_sec=n_timed / dt_, frames=n_timed, pipeline_depth=dpt, extract_batch=bx,
                   icp_iters_mean=float(np.mean([r["icp_iters"] for r in res])), icp_valid_share=float(np.mean([r["icp_valid"] for r in res])),
                   n_model=int(res[-1]["n_model"]), n_visible=int(res[-1]["n_visible"]))
        # stage split + per-kernel brackets, one batch at a time (as the headline's profile leg)
        f.set_profile(2)
        stage = np.zeros(3)
        nsub, base_ = 0, n_warm + n_timed
        rs = []
        for i in range(bx):
            while nsub < bx and f.can_submit():
                f.submit_frame(t_rgb[sweep(base_ + nsub)].data_ptr(), t_dep[sweep(base_ + nsub)].data_ptr(), on_device=True); nsub += 1
            rs.append(f.process_submitted().as_dict())
        for r in rs:
            stage += np.array(r["stage_ms"]) / bx
        f.set_profile(1); f.reset_kernel_times()
        for rep in range(2):
            nsub = 0
            for i in range(bx):
                while nsub < bx and f.can_submit():
                    f.submit_frame(t_rgb[sweep(base_ + bx * (rep + 1) + nsub)].data_ptr(), t_dep[sweep(base_ + bx * (rep + 1) + nsub)].data_ptr(), on_device=True); nsub += 1
                f.process_submitted()
            torch.cuda.synchronize(dev)
        kt = f.kernel_times()
        f.set_profile(0)
        ent["stage_ms"] = dict(extract=stage[0], icp=stage[1], fuse=stage[2])
        ent["kernel_avg_us"] = {k: round(1000.0 * ms / max(c, 1), 2) for k, (ms, c) in kt.items()
                                if k in ("update_pass_rgbd", "update_pass_rgb", "bilateral_prefilter", "render_moments", "eval_samples", "init_disp", "icp_accumulate", "match", "update_insert", "reorder_move_icp", "reorder_move")}
        ent["update_pass_rgbd_frames_per_launch"] = bx
        f.close()
        out[key] = ent
    # relabelling statistics from the lab build (the product keeps no such counters): real frames, and the rendered room beside them
    try:
        lab = binding.load_lab()
        lab.lib.ssf_dbg_pass_stats.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
        lab.lib.ssf_dbg_pass_stats_enable(1)

        def stats(frames_host, cfg_):
            fl = binding.Fusion(lab, cfg_)
            tot = np.zeros(64, np.uint64); mx = 0
            for rgb_, dep_ in frames_host:
                fl.process_frame(rgb_, dep_)
                w = np.zeros(64, np.uint32)
                if lab.lib.ssf_dbg_pass_stats(fl.h, w.ctypes.data_as(ctypes.c_void_p)) != 0:
                    raise RuntimeError("ssf_dbg_pass_stats")
                tot += w.astype(np.uint64); mx = max(mx, int(w[9]))
            fl.close()
            tiles = max(int(tot[10]), 1)
            return dict(frames=len(frames_host), log_entries_per_tile_and_pass_mean=float(tot[8]) / tiles, log_entries_per_tile_and_pass_max=mx, log_capacity=256,
                        row_lookups_outside_the_lds_window_share=float(tot[11]) / max(float(tot[12]), 1.0), row_lookups=int(tot[12]))
        real16 = [fr[sweep(i)] for i in range(16)]
        out["relabelling_statistics_real_frames"] = stats(real16, cfg_for(lab, 100000, 0, 1))
        synth = render_frames(8, tum_shaped=False)
        kw = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=100000, depth_prefilter=0)
        if torch.cuda.is_available():
            kw["device_id"] = torch.cuda.current_device()
        out["relabelling_statistics_box_room"] = stats([(np.ascontiguousarray(a_), np.ascontiguousarray(b_)) for a_, b_ in synth], lab.default_config(**kw))
    except Exception as e:           # (a box without the lab build: the timings above stand on their own)
        out["relabelling_statistics_error"] = repr(e)
    return out


def pin_to_gpu_numa_node(local):
    """Run this process on the CPUs of the NUMA node the GPU hangs off (what `numactl --cpunodebind` does): the track
    chain is a sequence of host <-> device round trips (mailbox polls, doorbells, BAR stores), and on a two-socket host a
    far node adds an inter-socket hop to every one of them.  Returns a description for the JSON line."""
    try:
        pr = torch.cuda.get_device_properties(local)
        bdf = "%04x:%02x:%02x.0" % (getattr(pr, "pci_domain_id", 0), pr.pci_bus_id, pr.pci_device_id)
        base = "/sys/bus/pci/devices/" + bdf
        node = int(open(base + "/numa_node").read())
        cpus = open(base + "/local_cpulist").read().strip()
        if node < 0 or not cpus:
            return dict(pinned=False, reason="no NUMA affinity reported for %s" % bdf)
        want = set()
        for part in cpus.split(","):
            lo, _, hi = part.partition("-")
            want.update(range(int(lo), int(hi or lo) + 1))
        want &= os.sched_getaffinity(0)
        if not want:
            return dict(pinned=False, reason="local cpus of %s not in this process's affinity mask" % bdf)
        os.sched_setaffinity(0, want)
        return dict(pinned=True, gpu=bdf, numa_node=node, cpus=cpus, n_cpus=len(want))
    except Exception as e:          # no sysfs, old torch: run unpinned
        return dict(pinned=False, reason=repr(e))


def render_frames(n, tum_shaped=False):
    """tum_shaped (BASELINE config 5): ~25 % of the pixels holes, depth quantised to 16 bits at 5000 counts per metre and
    converted back as the benchmark node does (depth_scale 0.0002)"""
    frames = []
    for k in range(n):
        R, t = synthetic.orbit_pose(k)
        rgb, depth, _ = synthetic.render(R, t, W, H, noise=True, holes=0.25 if tum_shaped else 0.0, rng=np.random.default_rng(1000 + k))
        if tum_shaped:
            d16 = np.clip(np.rint(depth.astype(np.float64) * 5000.0), 0, 65535).astype(np.uint16)
            depth = (d16.astype(np.float64) * 0.0002).astype(np.float32)
        frames.append((rgb, depth))
    return frames


def other_config(cfg_n, steps, pin):
    """bench.py --config <cfg_n> in a process of its own -> what the driver's line carries about it: frames/s, the dominant
    kernel with its roofline fraction and (when a PMC file stamped with the current kernel sources exists for that
    configuration) its HBM traffic, the whole-frame roofline fraction"""
    import subprocess
    cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--config", str(cfg_n), "--steps", str(steps), "--extras", "0",
           "--cpu-frames", "0", "--pin", str(pin)]
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    try:
        r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, timeout=900, env=env)
        lines = [l for l in r.stdout.splitlines() if l.startswith("{")]
        if r.returncode != 0 or not lines:
            return dict(error="bench.py --config %d ended with %d: %s" % (cfg_n, r.returncode, r.stderr[-400:]))
        d = json.loads(lines[-1])
    except Exception as e:                   # a sub-run must never take the headline line with it
        return dict(error=repr(e))
    rf, fr = d.get("roofline") or {}, d.get("frame_roofline") or {}
    return dict(frames_per_sec=d["value"], ms_per_frame=d["ms_per_step"], frames=d["steps"], workload=d["config"]["workload"],
                n_visible=d["config"]["n_visible"], icp_iters_mean=d["config"]["icp_iters_mean"], stage_ms=d.get("stage_ms"),
                sequential_ms_per_frame=d.get("sequential_ms_per_frame"),
                roofline=dict(kernel=rf.get("kernel"), frac=rf.get("frac"), achieved_GBs=rf.get("achieved"), avg_launch_us=rf.get("avg_launch_us"),
                              algo_bytes_per_launch=rf.get("algo_bytes_per_launch"), traffic=rf.get("traffic"), traffic_note=rf.get("traffic_note")),
                frame_roofline_frac=fr.get("frac"),
                per_kernel={k: dict(avg_us=v["avg_us"], achieved_GBs=v.get("achieved_GBs")) for k, v in (d.get("per_kernel") or {}).items()
                            if k in ("icp_accumulate", "match", "update_insert", "reorder_move_icp", "reorder_move", "update_pass_rgbd", "render_moments", "bilateral_prefilter", "apply_deformation")},
                command=" ".join(["bench.py"] + cmd[2:]))


def launch_plan(gpus, env, device_count, argv):
    """What `python bench.py --gpus N ...` has to do given how it was started (pure: tests/test_bench_launcher.py).
      ("run", None)      this process is a rank (N == 1, or launched by torch.distributed.run with WORLD_SIZE == N)
      ("spawn", cmd)     N > 1 and no WORLD_SIZE in the environment: re-exec under torch.distributed.run, one rank per GPU
      ("error", text)    the box cannot run it (fewer than N GPUs; --gpus disagrees with the launcher's WORLD_SIZE)"""
    if gpus < 1:
        return "error", "--gpus must be >= 1 (got %d)" % gpus
    if "WORLD_SIZE" in env:                                   # under a launcher (the driver's form for N > 1)
        world = int(env["WORLD_SIZE"])
        if world != gpus:
            return "error", "--gpus %d but WORLD_SIZE=%d (launch with torch.distributed.run --nproc-per-node %d)" % (gpus, world, gpus)
        if device_count < 1:
            return "error", "bench.py needs a GPU (the product has no CPU fallback)"
        local = int(env.get("LOCAL_RANK", "0"))
        if local >= device_count:
            return "error", "LOCAL_RANK %d but only %d GPU(s) visible" % (local, device_count)
        return "run", None
    if device_count < gpus:
        return "error", ("--gpus %d but %d GPU(s) visible on this box" % (gpus, device_count)) if device_count else "bench.py needs a GPU (the product has no CPU fallback)"
    if gpus == 1:
        return "run", None
    port = env.get("MASTER_PORT") or str(29500 + (os.getpid() % 2000))
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(gpus),
           "--master-addr", "127.0.0.1", "--master-port", port, os.path.abspath(__file__)] + list(argv)
    return "spawn", cmd


def error_line(text, gpus):
    """the one JSON line of a run that cannot start: same keys a reader of the contract looks for, value null"""
    return json.dumps({"metric": "frames_per_sec", "value": None, "unit": "frames/s", "n_gpus": gpus, "error": text})


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1200)
    ap.add_argument("--warmup", type=int, default=48)
    ap.add_argument("--rendered-frames", type=int, default=64,
                    help="distinct orbit frames rendered; the sequence sweeps them back and forth (consecutive frames stay 1 degree apart)")
    ap.add_argument("--seed-order", default="seeded", choices=["seeded", "image"],
                    help="config 3 only: 'image' = the seeded rows in raster order of the superpixel they project to under camera 0 "
                         "(the order a map built by the pipeline has), a measurement beside the BASELINE workload")
    ap.add_argument("--force-icp", action="store_true", help="always run icp_iter iterations (BASELINE config 3)")
    ap.add_argument("--config", type=int, default=2, choices=(2, 3, 4, 5),
                    help="2 (default): the configuration the metric is quoted on, 640x480 / ~1M supersurfels (N > 1: the same map "
                         "sharded, strong scaling); 3: the HBM-bound stress of BASELINE.json, 1280x960, ~1M supersurfels all "
                         "visible, 10 forced ICP iterations; 4: BASELINE config 4, 640x480 with 500 k supersurfels PER RANK "
                         "(2 M over 4 GPUs), weak scaling; 5: BASELINE config 5, TUM-shaped input (u16 depth at 5000 / m, ~25 %% "
                         "holes, depth pre-filter inside the frame) on the 1 M map, one loop-closure deformation (N / 50 nodes) "
                         "applied before the timed region")
    ap.add_argument("--pin", type=int, default=1, help="1 (default): bind the process to the CPUs of the GPU's NUMA node")
    ap.add_argument("--extras", type=int, default=1,
                    help="1 (default, N = 1 only): also measure the same workload with host-resident frames (PCIe-inclusive) and "
                         "with the depth pre-filter inside the frame, and the 'next' kernels (deformation, pre-filter, align); 0: skip")
    ap.add_argument("--cpu-frames", type=int, default=None,
                    help="frames of the bounded cpu_baseline sample (0 = skip; default 80, 12 at 1280x960: 10-15 s of CPU work)")
    ap.add_argument("--profile-frames", type=int, default=32,
                    help="frames of the stage split + per-kernel hipEvent profile behind `roofline` (half each; 32 = 40 launches of the dominant kernel)")
    ap.add_argument("--pipeline-depth", type=int, default=2,
                    help="batches the extract stage may run ahead of ICP/fusion (0 = strictly sequential)")
    ap.add_argument("--extract-batch", type=int, default=None, help="frames per extract launch chain (default 8; 4 at 1280x960)")
    ap.add_argument("--force-sharded", action="store_true",
                    help="run the multi-GPU exchanges (collectives included) even on one rank: exercises the N > 1 code path")
    ap.add_argument("--comm", choices=["rccl", "p2p"], default="rccl",
                    help="native exchange backend at N > 1 (or with --force-sharded): RCCL collectives on the track stream, or the "
                         "peer-to-peer exchange regions of ssf_p2p_* (one node, no collective launches)")
    ap.add_argument("--extract", choices=["replicated", "dealt"], default="replicated",
                    help="N > 1 (or --force-sharded) with the RCCL backend: 'dealt' = batch j of the frame stream is extracted by rank j %% N alone, "
                         "which broadcasts its frames' tables (ssf_comm_deal_extract): 1 / N of the extract work per rank.  Default "
                         "'replicated' (every rank extracts every frame, nothing is shipped): the dealt form has run on ONE rank only")
    ap.add_argument("--py-driver", action="store_true",
                    help="N > 1 through supersurfel_fusion_amd/sharded.py (torch.distributed collectives) instead of native RCCL")
    a = ap.parse_args()
    # stdout carries exactly one JSON line: everything the runtime libraries print there (RCCL's banner) is sent to
    # stderr until the result is ready
    saved_stdout = os.dup(1)
    os.dup2(2, 1)
    global W, H, P, N_MODEL
    if a.config == 3:
        W, H, a.force_icp = 1280, 960, True
        a.steps, a.warmup, a.rendered_frames = min(a.steps, 64), min(a.warmup, 8), min(a.rendered_frames, 16)
    P = W * H

    # how was this started?  N > 1 without a launcher: start one (one rank per GPU); a box that cannot run it gets one
    # JSON line with "error" and a non-zero exit, not a traceback
    plan, detail = launch_plan(a.gpus, os.environ, torch.cuda.device_count() if torch.cuda.is_available() else 0, sys.argv[1:])
    if plan == "error":
        if int(os.environ.get("RANK", "0")) == 0:
            os.dup2(saved_stdout, 1)
            print(error_line(detail, a.gpus), flush=True)
        sys.stderr.write("bench.py: %s\n" % detail)
        sys.exit(2)
    if plan == "spawn":
        os.dup2(saved_stdout, 1)                   # the ranks inherit the real stdout: rank 0 prints the line
        env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
        env.setdefault("OMP_NUM_THREADS", "16")     # torch.distributed.run would set 1 (and say so): the cpu_baseline leg of rank 0 uses OpenMP
        os.execve(sys.executable, detail, env)
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    if a.config == 4:
        N_MODEL = 500000 * world                       # BASELINE config 4: ~2 M supersurfels over 4 GPUs -> 500 k per rank
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    affinity = pin_to_gpu_numa_node(local) if a.pin else dict(pinned=False, reason="--pin 0")
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=dev)

    lib = binding.load_product()       # raises when libssf_hip.so is missing: no fallback
    K, Wm = a.steps, a.warmup
    nr = max(2, a.rendered_frames)
    frames = render_frames(nr, tum_shaped=a.config == 5)
    r_rgb = [torch.from_numpy(f[0]).to(dev) for f in frames]
    r_depth = [torch.from_numpy(f[1]).to(dev) for f in frames]

    class Sweep:                                   # frame i of the stream: 0,1,..,nr-1,nr-2,..,1,0,1,..
        def __init__(self, arr):
            self.arr = arr

        def __getitem__(self, i):
            j = i % (2 * nr - 2)
            return self.arr[j if j < nr else 2 * nr - 2 - j]

    d_rgb, d_depth, h_frames = Sweep(r_rgb), Sweep(r_depth), Sweep(frames)

    if a.config == 3:
        model, nvis = synthetic.seed_model_cam0_visible(N_MODEL, W, H, stamp=30)
        if a.seed_order == "image":
            # (a measurement, not the BASELINE workload: the seeded rows in the order a map BUILT by the pipeline has them -- every
            # frame appends its new supersurfels in ascending frame id, i.e. raster order of their superpixels -- instead of the
            # seeding's random order: what the gathers of k_icp / k_match cost when the rows are image-coherent by themselves)
            R0, t0 = synthetic.orbit_pose(0)
            Kc = synthetic.intrinsics(W, H)
            pc = (model["positions"].astype(np.float64) - t0) @ R0
            u = Kc["fx"] * pc[:, 0] / pc[:, 2] + Kc["cx"]; v = Kc["fy"] * pc[:, 1] / pc[:, 2] + Kc["cy"]
            key = (np.clip(v, 0, H - 1).astype(np.int64) // 16) * ((W + 15) // 16) + np.clip(u, 0, W - 1).astype(np.int64) // 16
            order = np.argsort(key[:nvis], kind="stable")
            model = {k_: np.concatenate([val[:nvis][order], val[nvis:]]) for k_, val in model.items()}
    else:
        model, nvis = synthetic.seed_model_cam0(N_MODEL, W, H, stamp=30)
    if world > 1:
        own = synthetic.tile_owner(model["positions"], world, 0.5) == rank
        vis = np.arange(N_MODEL) < nvis
        nvis_local = int((own & vis).sum())
        model_local = {k: v[own] for k, v in model.items()}
    else:
        model_local, nvis_local = model, nvis
    n_local = len(model_local["confidences"])
    cap = n_local + 65536
    if a.extract_batch is None:
        a.extract_batch = 4 if a.config == 3 else (12 if a.config == 5 else 8)        # measured optima (12 with the pre-filter in the frame: the extract stage bounds the replay then)
    if a.cpu_frames is None:
        a.cpu_frames = 12 if a.config == 3 else 80
    depth, batch = a.pipeline_depth, a.extract_batch
    # N > 1: the map is sharded and the library exchanges natively over RCCL on its track stream
    # (ssf_comm_attach; torch.distributed only ships the communicator id and provides the barrier).
    # --py-driver runs the same protocol through the stage seams with torch.distributed collectives instead.
    exchange = world > 1 or a.force_sharded
    if exchange and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
        dist.init_process_group("nccl", rank=0, world_size=1, device_id=dev)

    def make_engine(mode):
        """mode: 'none' (one rank, no exchange) | 'rccl' | 'p2p' (native exchanges) | 'py' (torch.distributed through the stage seams)"""
        py_driver = mode == "py"
        tstream, stream = None, None               # library-owned (high-priority) track stream
        if py_driver:
            tstream = torch.cuda.Stream(dev)       # the torch stream the collectives are ordered on
            stream = tstream.cuda_stream
        fus = binding.Fusion(lib, make_cfg(lib, cap, rank, world, stream, a.force_icp, depth, batch, prefilter=1 if a.config == 5 else 0))
        fus.set_model(model_local, nvis_local, 30)
        if a.config == 5:                              # one loop-closure deformation of the whole map (applyDeformation)
            rng = np.random.default_rng(5)
            n_, m_ = n_local, max(n_local // 50, 4)
            ang = rng.uniform(-0.002, 0.002, (m_, 3))
            nrot = np.stack([(synthetic.rot_y(a_[1]) @ synthetic.rot_x(a_[0])).reshape(9) for a_ in ang]).astype(np.float32)
            fus.apply_deformation(rng.uniform(-3, 3, (m_, 3)).astype(np.float32), nrot, rng.uniform(-1e-3, 1e-3, (m_, 3)).astype(np.float32),
                                  rng.dirichlet(np.ones(4), n_).astype(np.float32), rng.integers(0, m_, (n_, 4)).astype(np.int32))
            if world > 1:                              # a sharded map: the rows the deformation pushed over a tile edge go to their new owners
                sharded.rehome_over(fus, world, device=dev)
        if py_driver:
            return fus, sharded.ShardedFusion(fus, device=dev, stream=tstream, always_reduce=a.force_sharded)
        try:
            if mode == "p2p":
                # every rank on its own GPU (LOCAL_RANK): fine-grained regions; a forced one-rank run shares its GPU with itself
                fus.p2p_configure(all_ranks_on_this_device=(world == 1))
                fus.p2p_attach()                   # IPC handles of the exchange regions, all-gathered over torch.distributed
            elif mode == "rccl":
                fus.comm_attach()
                if a.extract == "dealt":
                    fus.comm_deal_extract(1)
        except Exception:
            fus.close()
            raise
        return fus, None

    # N > 1 (or --force-sharded): the native exchange asked for, then the other native one, then the torch.distributed driver --
    # every rank must take the same path, so each attempt's outcome is agreed over the ranks (MIN), and every refusal is recorded
    # in the line (`exchange_fallthrough`) instead of ending the run in a traceback
    fallthrough = []
    native_ok = 1
    if exchange:
        order = ["py"] if a.py_driver else ([a.comm] + [m_ for m_ in ("rccl", "p2p") if m_ != a.comm] + ["py"])
        f = drv = None
        for mode in order:
            ok, why = 1, ""
            try:
                f, drv = make_engine(mode)
            except Exception as e:          # binding.SsfError, a missing librccl, an IPC refusal ...
                ok, why, f, drv = 0, "%s: %s" % (type(e).__name__, e), None, None
            flag = torch.tensor([ok], dtype=torch.int32, device=dev)
            dist.all_reduce(flag, op=dist.ReduceOp.MIN)
            if int(flag.item()):
                exchange_mode = mode
                break
            fallthrough.append(dict(tried=mode, rank=rank, reason=why or "another rank could not attach"))
            sys.stderr.write("exchange '%s' refused on rank %d (%s): trying the next form\n" % (mode, rank, why or "another rank"))
            if f is not None:
                f.close(); f = drv = None
        if f is None:
            raise RuntimeError("no exchange form could be attached: %s" % fallthrough)
        native_ok = 0 if exchange_mode == "py" else 1
        a.comm = exchange_mode if exchange_mode != "py" else a.comm
    else:
        exchange_mode = "none"
        f, drv = make_engine("none")
    eng = drv if drv is not None else f
    comm_info = f.comm_info()
    if exchange and drv is None and comm_info["ranks"] != world:
        raise RuntimeError("the attached exchange reports %d rank(s), the launcher %d" % (comm_info["ranks"], world))

    def step(i):
        if drv is not None:
            return drv.process_frame(d_rgb[i].data_ptr(), d_depth[i].data_ptr(), on_device=True)
        return f.process_frame_device(d_rgb[i].data_ptr(), d_depth[i].data_ptr()).as_dict()

    def as_dict(r):
        return r if isinstance(r, dict) else r.as_dict()

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize(dev)

    cap_frames = f.pipeline_capacity()

    def run(first, count, native=True):
        """Process frames [first, first + count): every frame's extract, ICP and fusion complete inside.
        native=False: the same loop through ssf_submit_frame / ssf_process_submitted (full batches from the first
        frame on: the per-kernel profile wants every launch to cover extract_batch frames)."""
        res = []
        if depth == 0 and batch == 1:
            for i in range(first, first + count):
                res.append(step(i))
            return res
        if drv is None and native:                  # the submit-ahead / process-in-order loop, natively
            return f.process_sequence([d_rgb[i].data_ptr() for i in range(first, first + count)],
                                      [d_depth[i].data_ptr() for i in range(first, first + count)], on_device=True)
        return drive_pipelined(eng, lambda i: (d_rgb[i].data_ptr(), d_depth[i].data_ptr()), first, count, on_device=True)

    run(0, Wm)
    # the extract graph of every batch size the timed region will launch is built lazily on first use (like a JIT).  A
    # sequence starts with batches of batch/4 and batch/2 frames, continues with full ones and ends with what is left
    # (ssf_process_sequence): one short untimed sequence with exactly those sizes builds the graphs before the timing
    if batch > 1 and not (depth == 0 and batch == 1):
        ramp = [max(1, (3 * batch + 4) // 8), max(1, (5 * batch + 4) // 8)]      # the library's leading batches: 3/8 and 5/8 of a batch (seq_batch_size)
        if os.environ.get("SSF_SEQ_RAMP"):          # (experiments: the lab build reads the same variable)
            ramp = [min(batch, max(1, int(v))) for v in os.environ["SSF_SEQ_RAMP"].split(",") if v]
        tail = (K - sum(ramp)) % batch if K > sum(ramp) else 0
        # graphs are per (context, batch size): a warm-up sequence of the SAME length builds exactly the graphs a short timed
        # sequence uses; a long one needs every context to have seen a full batch (ramp + one full batch per context + tail)
        extra = K if K <= 64 else sum(ramp) + (depth + 1) * batch + tail
        run(Wm, extra)
        Wm += extra                                 # (reported as warmup_extra_frames)
    native_seq = not (depth == 0 and batch == 1) and drv is None
    if native_seq:
        # the harness's own work (argument arrays before, result structs -> dicts after) stays outside the timed
        # region: inside are exactly the K frames, submitted and processed by the library
        prepared = f.prepare_sequence([d_rgb[i].data_ptr() for i in range(Wm, Wm + K)], [d_depth[i].data_ptr() for i in range(Wm, Wm + K)])
    barrier()
    t0 = time.perf_counter()
    results = f.process_prepared(prepared, on_device=True) if native_seq else run(Wm, K)
    t_ret = time.perf_counter()
    barrier()
    dt = time.perf_counter() - t0
    fill = None
    if native_seq:
        results = [r.as_dict() for r in results]
        if K <= 64:
            # where a short timed region goes: completion time of every frame inside ssf_process_sequence (the library's
            # own clock, from its entry), the call's return and the end of the closing barrier, all in microseconds
            tt_ = np.zeros(64)
            lib.lib.ssf_sequence_times.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
            lib.lib.ssf_sequence_times(f.h, tt_.ctypes.data_as(ctypes.c_void_p))
            fill = dict(frame_done_us=[round(float(v), 1) for v in tt_[:K]], call_returned_us=round(1e6 * (t_ret - t0), 1),
                        region_us=round(1e6 * dt, 1), icp_iters=[int(r["icp_iters"]) for r in results])
            if hasattr(lib.lib, "ssf_sequence_marks"):
                mk = np.zeros(320)
                lib.lib.ssf_sequence_marks.argtypes = [ctypes.c_void_p, ctypes.c_void_p]
                lib.lib.ssf_sequence_marks(f.h, mk.ctypes.data_as(ctypes.c_void_p))
                for j, name in enumerate(("track_entered_us", "first_icp_record_us", "icp_done_us")):
                    fill[name] = [round(float(v), 1) for v in mk[64 * j:64 * j + K]]
                fill["extract_batches_launched"] = [dict(at_us=round(float(mk[256 + 2 * i]), 1), frames=int(mk[257 + 2 * i]), host_us=round(1e4 * (mk[257 + 2 * i] % 1.0), 1)) for i in range(32) if mk[256 + 2 * i] >= 0]
    iters = [r["icp_iters"] for r in results]
    last = results[-1]
    # strictly sequential latency (one frame in flight, pipeline_depth 0 / extract_batch 1): a second handle at
    # N = 1, the same handle through the one-frame entry point otherwise
    nseq = a.profile_frames
    if not exchange:
        fseq = binding.Fusion(lib, make_cfg(lib, cap, rank, world, None, a.force_icp, 0, 1, prefilter=1 if a.config == 5 else 0))
        fseq.set_model(model_local, nvis_local, 30)
        seq_step = lambda i: fseq.process_frame_device(d_rgb[i].data_ptr(), d_depth[i].data_ptr())  # noqa: E731
    else:
        fseq, seq_step = None, step
    s0 = 0 if fseq is not None else Wm + K         # the fresh handle starts at frame 0, the shared one continues
    for i in range(s0, s0 + 4):
        seq_step(i)
    barrier()
    t1 = time.perf_counter()
    for i in range(s0 + 4, s0 + 4 + nseq):
        seq_step(i)
    barrier()
    seq_ms = 1000.0 * (time.perf_counter() - t1) / max(nseq, 1)
    if fseq is not None:
        fseq.close()
    base = Wm + K + (0 if fseq is not None else 4 + nseq)
    if world > 1:
        tt = torch.tensor([dt], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())

    # ---- roofline of the dominant kernel: per-kernel hipEvent times on the library's stream ----
    # (same submission pattern as the timed region: batched / pipelined when configured)
    stage = np.zeros(3)
    ns = batch                                        # one whole batch for the stage split
    # per-kernel brackets: ONE batch at a time, the device drained in between -- a launch's duration is then the kernel's own
    # (with two batches in flight the extract launches of one run beside the track kernels of the other and every bracket
    # measures the contention too: 17 instead of 13.5 us for the dominant launch); `reps` such batches
    reps = max(1, a.profile_frames // (2 * batch))
    npk = reps * batch
    f.set_profile(2)                                  # stage split only (one event synchronise per frame)
    for r in run(base, ns, native=False):
        stage += np.array(r["stage_ms"]) / ns
    f.set_profile(1); f.reset_kernel_times()          # per-kernel hipEvent brackets
    cnt_before = f.counts()
    for rep in range(reps):
        run(base + ns + rep * batch, batch, native=False)
        torch.cuda.synchronize(dev)
    kt = f.kernel_times()
    f.set_profile(0)
    counts = dict(n_model=cnt_before["n_model"], n_visible=cnt_before["n_visible"], S=f.S, batch=batch, passes=2 * int(PARAMS["seg_iter"]))      # (4 passes per iteration, half of the iterations per phase)
    per_kernel = {}
    for name, (ms, calls) in kt.items():
        avg_us = 1000.0 * ms / max(calls, 1)
        ent = dict(total_ms_per_frame=ms / npk, launches_per_frame=calls / npk, avg_us=avg_us)
        if name in ALGO_BYTES:
            by = ALGO_BYTES[name](counts)
            ent["algo_bytes_per_launch"] = by
            ent["achieved_GBs"] = by / (avg_us * 1e-6) / 1e9
        per_kernel[name] = ent
    exchange_by_kind = {n: round(1000.0 * e["total_ms_per_frame"], 3) for n, e in per_kernel.items() if n.startswith("exchange_") or n.startswith("p2p_")}
    exchange_us = float(sum(exchange_by_kind.values()))
    # The dominant KERNEL: k_update_pass is one kernel with two instantiations (RGB / RGB-D passes, timed under two names);
    # its share is their sum.  The roofline is reported for the instantiation with the larger share of the two.
    fam = lambda n: "update_pass" if (n.startswith("update_pass") or n.startswith("passes_team")) else n
    fam_ms = {}
    for n, e in per_kernel.items():
        fam_ms[fam(n)] = fam_ms.get(fam(n), 0.0) + e["total_ms_per_frame"]
    dom_fam = max(fam_ms, key=fam_ms.get) if fam_ms else None
    dom = max((n for n in per_kernel if fam(n) == dom_fam), key=lambda n: per_kernel[n]["total_ms_per_frame"]) if dom_fam else None
    roofline = None
    if dom is not None and "achieved_GBs" in per_kernel[dom]:
        ach = per_kernel[dom]["achieved_GBs"]
        # HBM traffic of the kernel from the PMC counters (FETCH_SIZE x2 + WRITE_SIZE, KB; separate rocprofv3
        # --pmc passes of this same command, recorded in profiles/pmc_r02.json by tools/pmc_summary.py together with
        # a hash of the kernel sources it was taken at -- bench.py cannot run the profiler on itself).  A file taken
        # at other sources, or at another extract batch, does not describe this binary: traffic = null then.
        import glob
        traffic, traffic_note = None, "no profiles/pmc_r*%s.json" % ("" if a.config == 2 else "_config%d" % a.config)
        # (the default workload: profiles/pmc_rNN.json; another BASELINE configuration: profiles/pmc_rNN_config<k>.json)
        pmcs = sorted(glob.glob(os.path.join(ROOT, "profiles", "pmc_r[0-9][0-9]%s.json" % ("" if a.config == 2 else "_config%d" % a.config))))
        if pmcs:
            pmc_path = pmcs[-1]; pmc_name = "profiles/" + os.path.basename(pmc_path)
            pmc = json.load(open(pmc_path))
            if pmc.get("source_sha") != kernel_source_sha():
                traffic_note = "%s was recorded at other kernel sources (%s)" % (pmc_name, pmc.get("source_sha"))
            elif pmc.get("extract_batch", 1) != batch and dom in EXTRACT_KERNELS:     # per-launch traffic depends on the batch
                traffic_note = "%s was recorded at extract_batch %s" % (pmc_name, pmc.get("extract_batch"))
            else:
                traffic = pmc["kernels"].get(dom, {}).get("hbm_bytes_per_launch")
                traffic_note = "rocprofv3 --pmc FETCH_SIZE (x2, gfx950) + WRITE_SIZE, %s @ %s" % (pmc_name, pmc.get("source_sha"))
        roofline = dict(bound="hbm", kernel=dom, achieved=ach, peak=HBM_PEAK_GBS, unit="GB/s", frac=ach / HBM_PEAK_GBS,
                        traffic=traffic, traffic_note=traffic_note, avg_launch_us=per_kernel[dom]["avg_us"],
                        algo_bytes_per_launch=per_kernel[dom]["algo_bytes_per_launch"],
                        kernel_share_ms_per_frame={n: round(per_kernel[n]["total_ms_per_frame"], 5) for n in per_kernel if fam(n) == dom_fam})
        # ---- what the PROFILE says beside it (round 6).  `kernel` above is the dominant kernel FAMILY's larger instantiation (one
        # kernel, two template instances); by kernel NAME the largest single item of GPU time can be another one (k_icp at the
        # metric's workload: profiles/rocprof_r05.txt) -- reported here with its own roofline figures, from the same hipEvent brackets:
        tot_ms = sum(e["total_ms_per_frame"] for e in per_kernel.values()) or 1.0
        topn = max(per_kernel, key=lambda n: per_kernel[n]["total_ms_per_frame"])
        te = per_kernel[topn]
        top_traffic = None
        if pmcs and traffic is not None:
            top_traffic = pmc["kernels"].get(topn, {}).get("hbm_bytes_per_launch")
        roofline["top_kernel_by_name"] = dict(kernel=topn, share_of_kernel_time=te["total_ms_per_frame"] / tot_ms, avg_launch_us=te["avg_us"],
                                              launches_per_frame=te["launches_per_frame"], achieved=te.get("achieved_GBs"),
                                              frac=(te["achieved_GBs"] / HBM_PEAK_GBS) if te.get("achieved_GBs") else None,
                                              algo_bytes_per_launch=te.get("algo_bytes_per_launch"), traffic=top_traffic,
                                              traffic_over_algorithmic=(top_traffic / te["algo_bytes_per_launch"]) if (top_traffic and te.get("algo_bytes_per_launch")) else None)
        roofline["traffic_over_algorithmic"] = (traffic / per_kernel[dom]["algo_bytes_per_launch"]) if traffic else None
        # ... and the dominant kernel's fraction over the LAUNCH MIX of the committed rocprofv3 trace (leading batches of 3 and 5 frames
        # included, where `frac` above is quoted at full batches): profiles/rocprof_rNN.json, written by tools/rocprof_summary.py from
        # the trace's own durations and grid sizes, used only when it was taken at these kernel sources
        rp = sorted(glob.glob(os.path.join(ROOT, "profiles", "rocprof_r[0-9][0-9]%s.json" % ("" if a.config == 2 else "_config%d" % a.config))))
        roofline["frac_from_rocprof"], roofline["frac_from_rocprof_note"] = None, "no profiles/rocprof_rNN.json"
        if rp:
            rj = json.load(open(rp[-1])); rname = "profiles/" + os.path.basename(rp[-1])
            mixk = (rj.get("relabelling_launch_mix") or {}).get(dom)
            if rj.get("source_sha") != kernel_source_sha():
                roofline["frac_from_rocprof_note"] = "%s was recorded at other kernel sources (%s)" % (rname, rj.get("source_sha"))
            elif mixk:
                roofline["frac_from_rocprof"] = mixk["frac_of_8TBs"]
                roofline["frac_from_rocprof_note"] = ("%s: %d launches, %.2f frames per launch on average, %.2f us per launch (rocprofv3 --kernel-trace durations; "
                                                      "algorithmic bytes of every launch by its own frame count)" % (rname, mixk["launches"], mixk["mean_frames_per_launch"], mixk["avg_us"]))
                roofline["rocprof_top_kernels_by_name"] = [dict(kernel=t["kernel"], share=round(t["share"], 4), avg_us=round(t["avg_us"], 3)) for t in rj.get("top_kernels_by_name", [])[:4]]
                # the trace's largest single kernel NAME with its own roofline figures (its launch time in the trace, this run's
                # algorithmic bytes per launch, the stamped PMC traffic): `kernel` above is the dominant FAMILY's larger instantiation
                sym2name = (("k_icp", "icp_accumulate"), ("k_update_pass<true", "update_pass_rgbd"), ("k_update_pass<false", "update_pass_rgb"), ("k_match", "match"),
                            ("k_update_insert", "update_insert"), ("k_move_rows", "reorder_move"), ("k_render_moments", "render_moments"))
                t0 = (rj.get("top_kernels_by_name") or [None])[0]
                nm = next((b for a_, b in sym2name if t0 and t0["kernel"].startswith(a_)), None)
                if nm and nm in per_kernel and per_kernel[nm].get("algo_bytes_per_launch"):
                    ab = per_kernel[nm]["algo_bytes_per_launch"]; ach_t = ab / (t0["avg_us"] * 1e-6) / 1e9
                    tr = pmc["kernels"].get(nm, {}).get("hbm_bytes_per_launch") if (pmcs and traffic is not None) else None
                    roofline["rocprof_top_kernel"] = dict(kernel=t0["kernel"], bench_name=nm, share_of_gpu_time=t0["share"], avg_launch_us=t0["avg_us"], launches=t0["launches"],
                                                          algo_bytes_per_launch=ab, achieved=ach_t, frac=ach_t / HBM_PEAK_GBS, traffic=tr,
                                                          traffic_over_algorithmic=(tr / ab) if tr else None,
                                                          note="launch time from the trace (launches made ahead include their wait for the host's word), bytes from this run's row counts")
    # nominal AND measured-achievable peak: a stream copy on this box, outside every timed region
    hbm_measured = measured_hbm_peak(dev) if (rank == 0 and a.extras) else None          # (--extras 0: profiling runs stay free of the copy kernels)
    # ... and by a plain 16-bytes-per-lane copy kernel of the library's own (the form MI355X_MICROARCH.md quotes at 6.29 TB/s):
    # torch's copy kernel reaches ~15 % less on the same box, which is why hbm_peak_measured_GBs read 5.3 TB/s in round 3
    hbm_float4 = None
    if rank == 0 and a.extras and hasattr(lib.lib, "ssf_stream_copy_rate"):
        lib.lib.ssf_stream_copy_rate.restype = ctypes.c_double
        lib.lib.ssf_stream_copy_rate.argtypes = [ctypes.c_int, ctypes.c_int]
        v = lib.lib.ssf_stream_copy_rate(1024, 6)          # (best of six forms of the copy: unrolled 4 / 8, plain / non-temporal, grid-stride / one pass)
        hbm_float4 = v if v > 0 else None
    if roofline is not None and (hbm_measured or hbm_float4):
        best_copy = max(hbm_measured or 0.0, hbm_float4 or 0.0)
        roofline["peak_measured"] = best_copy; roofline["frac_of_measured"] = roofline["achieved"] / best_copy
        roofline["peak_measured_note"] = "best of torch's copy kernel (%s GB/s) and the library's float4 stream copy (%s GB/s: best of six forms, unrolled / non-temporal / one pass), 1 GiB read + 1 GiB written" % (
            "%.0f" % hbm_measured if hbm_measured else "n/a", "%.0f" % hbm_float4 if hbm_float4 else "n/a")

    # ---- the same workload handed over differently (N = 1): host-resident frames (what the reference's caller has: cv::Mat,
    # PCIe-inclusive) and with the library's depth pre-filter inside the frame (what the reference's processFrame does with
    # OpenCV's bilateralFilter).  Never `value`: extra keys. -------------------------------------------------------------------
    # ---- the steady-state rate: the same call over >= 600 frames, outside the timed region.  A short timed region (the
    # driver's --steps 20) is mostly pipeline fill -- the first frame can only be tracked once its batch has been extracted --;
    # this number says what the pipeline sustains (with the default --steps 1200, `value` is already that) -------------------
    steady = None
    if native_seq and world == 1 and a.extras:
        ks = 720
        prep_s = f.prepare_sequence([d_rgb[i].data_ptr() for i in range(base + ns + npk, base + ns + npk + ks)],
                                    [d_depth[i].data_ptr() for i in range(base + ns + npk, base + ns + npk + ks)])
        torch.cuda.synchronize(dev)
        t1 = time.perf_counter()
        f.process_prepared(prep_s, on_device=True)
        torch.cuda.synchronize(dev)
        steady = dict(frames_per_sec=ks / (time.perf_counter() - t1), frames=ks,
                      note="ssf_process_sequence over %d further frames of the same stream, same handle, outside the timed region" % ks)
    extras = None
    # (what the lines below still need of the timed handle, taken now: the extras run on handles of their own, and a handle that lives
    # BESIDE another one gets hardware queues of its own from the runtime -- with four or more streams already alive that halves the
    # newcomer's rate, which is what the node-call figures of rounds 2-4 measured; the timed handle is closed first and its streams go
    # back to the library's pool, so every extra runs on the queues the timed region had)
    gcounts_early = f.global_counts() if drv is None else None
    n_superpixels = f.S
    if rank == 0 and world == 1 and a.extras and not exchange and native_seq:
        f.close()
        extras = {}
        nx = 720                                     # (outside the timed region: independent of --steps; 240 until round 4: a 25 ms region, +-10 % run to run)
        host_frames = [(np.ascontiguousarray(h_frames[i][0]), np.ascontiguousarray(h_frames[i][1])) for i in range(nr)]
        hsweep = Sweep(host_frames)
        # the last one is what a node that swaps the library in gets from processFrame (supersurfel_fusion.cu:173-181): host images
        # (cv::Mat) in, depth pre-filter inside the frame -- both together
        # (host frames once more on a handle of pipeline_depth 1: their copy commands slow down with the number of hardware queues
        # the process keeps busy -- profiles/host_frames_r03.txt --, so that is what INTEGRATION.md recommends to callers that feed
        # host images; frames resident in HBM want the depth of the timed region)
        variants = [("host_frames_pageable", dict(), False, depth), ("with_depth_prefilter", dict(prefilter=1), True, depth),
                    ("host_frames_and_depth_prefilter", dict(prefilter=1), False, depth)]
        if depth > 1:
            variants += [("host_frames_pageable_depth1", dict(), False, 1), ("host_frames_and_depth_prefilter_depth1", dict(prefilter=1), False, 1)]
        if os.environ.get("BENCH_EXTRAS_AGAIN"):
            variants += [("host_frames_pageable_again", dict(), False, depth), ("host_frames_and_depth_prefilter_again", dict(prefilter=1), False, depth)]
        for key, kw, on_dev, dpt in variants:
            bx = 12 if (kw.get("prefilter") and batch == 8) else batch           # (the optimum with the pre-filter in the frame, as for config 5)
            fx = binding.Fusion(lib, make_cfg(lib, cap, 0, 1, None, a.force_icp, dpt, bx, **kw))
            fx.set_model(model_local, nvis_local, 30)
            def seq(first, count):
                if on_dev:
                    return fx.prepare_sequence([d_rgb[i].data_ptr() for i in range(first, first + count)], [d_depth[i].data_ptr() for i in range(first, first + count)])
                return fx.prepare_sequence([hsweep[i][0].ctypes.data for i in range(first, first + count)], [hsweep[i][1].ctypes.data for i in range(first, first + count)])
            fx.process_prepared(seq(0, a.warmup + 3 * bx), on_device=on_dev)
            prep = seq(a.warmup + 3 * bx, nx)
            torch.cuda.synchronize(dev)
            t1 = time.perf_counter()
            fx.process_prepared(prep, on_device=on_dev)
            torch.cuda.synchronize(dev)
            extras[key] = dict(frames_per_sec=nx / (time.perf_counter() - t1), frames=nx, pipeline_depth=dpt, extract_batch=bx)
            fx.close()
        extras["next_kernels"] = next_kernel_times(lib, dev, model_local, nvis_local, cap)
        if a.config == 2:
            try:
                extras["real_frames"] = real_frames_leg(lib, dev, model_local, nvis_local, cap)
            except Exception as e:          # (never the headline's problem)
                extras["real_frames"] = dict(error=repr(e))
        # ---- the other single-GPU BASELINE configurations beside the headline: config 3 (1280x960, 1 M rows in view, 10 forced
        # iterations: the HBM-bound stress) and config 5 (TUM-shaped input, pre-filter in the frame, one deformation), each as
        # this same script in a process of its own, outside every timed region of this one (>= 64 / >= 240 frames) -----------------
        if a.config == 2:
            for cfg_n, nsteps in ((3, 64), (5, 720)):
                extras["config%d" % cfg_n] = other_config(cfg_n, nsteps, a.pin)

    # ---- CPU baseline: the reference has no CPU implementation of this path, so the baseline
    # is the oracle restatement built -O3 -march=native ON THIS BOX, timed (i) single-threaded and (ii) with OpenMP over
    # all host cores; rank 0, bounded samples of the same workload --------------------------------------------------
    # (LAST of everything this process measures: its OpenMP teams fault memory in on every NUMA node of the host, and host frames
    # allocated afterwards -- the extras above -- were staged out of whatever the allocator recycled: the first extra ran at 4700
    # instead of 8200 frames/s behind it, round 4)
    cpu, parity = None, None
    if rank == 0 and world == 1 and a.cpu_frames > 0:            # (the contract: the CPU baseline on rank 0 at N = 1 only)
        import subprocess
        ncpu = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
        mk = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "omp", "native"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
        legs, omp_probe = {}, {}

        PARITY_KEYS = ("icp_valid", "icp_iters", "n_model", "n_visible", "n_removed", "n_inserted", "n_updated")

        def time_oracle(olib, nfr, first=1, keep=None):
            """keep: a list that receives (pose, counters) of every timed frame -- the parity leg below replays the same frames on
            the product"""
            fo = binding.Fusion(olib, make_cfg(olib, N_MODEL + 65536, 0, 1, None, a.force_icp, prefilter=1 if a.config == 5 else 0))
            fo.set_model(model, nvis, 30)
            fo.process_frame(*h_frames[0])                    # warm-up frame
            t1 = time.perf_counter()
            for i in range(first, first + nfr):
                r = fo.process_frame(*h_frames[i])
                if keep is not None:
                    keep.append((np.array(r["pose"], np.float32), [int(r[k]) for k in PARITY_KEYS]))
            dt_ = time.perf_counter() - t1
            fo.close()
            return nfr / dt_, nfr, dt_

        if mk.returncode == 0 and os.path.exists(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_native.so")):
            legs["single_thread"] = time_oracle(binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_native.so")), max(4, a.cpu_frames // 4))
        omp_threads = 1
        if mk.returncode == 0 and os.path.exists(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_omp.so")):
            olib = binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_omp.so"))
            # all host cores is not the fastest team on a many-core host (the per-frame loops are short: 1200 superpixels,
            # 480 image rows): a short probe over team sizes, then the sample with the best one; every probe is reported
            for nt in sorted({min(ncpu, 8), min(ncpu, 16), min(ncpu, 32), min(ncpu, 64), ncpu}):
                olib.lib.ssf_oracle_set_threads(nt)
                omp_probe[nt] = time_oracle(olib, 4)[0]
                if omp_probe[nt] < 0.5 * max(omp_probe.values()):      # past the knee (128 threads ran at 0.1 frames/s: 40 s of probe)
                    break
            omp_threads = max(omp_probe, key=omp_probe.get)
            olib.lib.ssf_oracle_set_threads(omp_threads)
            oracle_frames = []
            legs["openmp"] = time_oracle(olib, a.cpu_frames, keep=oracle_frames)
            # ---- parity inside the driver's record (north_star: "pose error vs reference <= 1e-4"): the frames the CPU oracle
            # has just been timed on, replayed on a product handle (the same call: one process_frame per host frame, one frame in
            # flight).  Checker only: nothing here is timed or shipped. ------------------------------------------------------------
            try:
                fp = binding.Fusion(lib, make_cfg(lib, N_MODEL + 65536, 0, 1, None, a.force_icp, 0, 1, prefilter=1 if a.config == 5 else 0))
                fp.set_model(model, nvis, 30)
                fp.process_frame(*h_frames[0])
                worst, equal, counters_equal = 0.0, 0, 0
                for j, (pose_o, cnt_o) in enumerate(oracle_frames):
                    r = fp.process_frame(*h_frames[1 + j])
                    pose_p = np.array(r["pose"], np.float32)
                    worst = max(worst, float(np.abs(pose_p.astype(np.float64) - pose_o.astype(np.float64)).max()))
                    same_cnt = [int(r[k]) for k in PARITY_KEYS] == cnt_o
                    counters_equal += 1 if same_cnt else 0
                    equal += 1 if (same_cnt and np.array_equal(pose_p.view(np.uint32), pose_o.view(np.uint32))) else 0
                fp.close()
                parity = dict(frames=len(oracle_frames), frames_bit_equal=equal, frames_counters_equal=counters_equal, max_abs_pose_diff=worst,
                              tolerance=1e-4, within_tolerance=bool(worst <= 1e-4),
                              checked="pose (12 floats, bit patterns) and %s of every frame: HIP product (one frame in flight, host frames) against "
                                      "the CPU oracle (OpenMP build; its bits equal the single-threaded checker's, tests/test_oracle.py) on the "
                                      "cpu_baseline sample of this workload" % ", ".join(PARITY_KEYS))
            except Exception as e:               # the parity leg must never take the headline line with it
                parity = dict(error=repr(e))
        if legs:
            best = "openmp" if "openmp" in legs else "single_thread"
            cpu = dict(value=legs[best][0], unit="frames/s", cores=omp_threads if best == "openmp" else 1, kind="port", host_cores=ncpu,
                       single_thread_frames_per_sec=legs.get("single_thread", (None,))[0],
                       openmp_frames_per_sec=legs.get("openmp", (None,))[0], openmp_threads=omp_threads,
                       openmp_probe_frames_per_sec_by_threads={str(k): v for k, v in omp_probe.items()},
                       parity=parity,            # (inside this object as well: the driver's record keeps config / roofline / cpu_baseline whole)
                       sample="the same %dx%d / ~%d-supersurfel workload on the CPU oracle (the build's restatement of the reference "
                              "algorithm; the reference has no CPU path), g++ -O3 -march=native: %s" %
                              (W, H, N_MODEL, "; ".join("%s %d frames in %.1f s" % (k, v[1], v[2]) for k, v in legs.items())))

    # whole-frame view: algorithmic bytes of one frame over the measured frame time
    it_mean = float(np.mean(iters))
    fb = dict(extract=180.0 * P, icp=(36.0 * counts["n_visible"] + 8.0 * P + 28.0 * counts["S"]) * it_mean,
              fuse=40.0 * counts["n_visible"] + 28.0 * counts["n_model"] + 2.0 * counts["n_model"] + 208.0 * counts["n_visible"])
    fbytes = sum(fb.values())
    frame_roofline = dict(bound="hbm", algo_bytes_per_frame=fbytes, bytes_by_stage=fb, achieved=fbytes / (dt / K) / 1e9,
                          peak=HBM_PEAK_GBS, unit="GB/s", frac=fbytes / (dt / K) / 1e9 / HBM_PEAK_GBS,
                          note="extract 180 B/pixel, ICP 36 B/visible supersurfel/iteration + frame tables, fuse: association 40 B/visible, "
                               "classify 28 B/row, row moves 2 B/slot + 208 B/visible row (the reference's full reorder would be 212 B/row)")
    node_call = max([(extras or {}).get(k, {}).get("frames_per_sec") or 0.0 for k in ("host_frames_and_depth_prefilter", "host_frames_and_depth_prefilter_depth1")]) or None
    if roofline is not None:
        # the whole-frame view inside the object the driver keeps: algorithmic bytes of one frame over the frame time of the timed
        # region, and over the steady-state frame time (720 further frames) when that was measured
        roofline["frame_frac"] = frame_roofline["frac"]; roofline["frame_algo_bytes"] = fbytes; roofline["frame_achieved"] = frame_roofline["achieved"]
        if steady:
            roofline["frame_frac_steady"] = fbytes * steady["frames_per_sec"] / 1e9 / HBM_PEAK_GBS
    gcounts = gcounts_early if drv is None else dict(n_model=last["global_n_model"], n_visible=last["global_n_visible"])
    if rank == 0:
        gn, gv = gcounts["n_model"], gcounts["n_visible"]
        out = {
            "metric": "frames_per_sec", "value": K / dt, "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": a.warmup,
            "ms_per_step": 1000.0 * dt / K, "higher_is_better": True, "scaling": "weak" if a.config == 4 else "strong", "vs_baseline": None,
            "dtype": "f32", "data": "synthetic",
            "config": {"workload": "%dx%d synthetic RGB-D orbit (seed 1234), map seeded with %d supersurfels "
                                   "(~%d live, ~%d visible), reference rgbd_benchmark parameters, extract+ICP+fuse per frame%s"
                                   % (W, H, N_MODEL, gn, gv, " (BASELINE config 3: all seeded supersurfels visible, 10 forced ICP iterations)"
                                      if a.config == 3 else ""),
                       "width": W, "height": H, "n_model": int(gn), "n_visible": int(gv), "superpixels": n_superpixels,
                       "icp_iter_max": 10, "icp_iters_mean": float(np.mean(iters)), "icp_forced": bool(a.force_icp), "baseline_config": a.config,
                       "seed_order": a.seed_order,      # "seeded": the BASELINE workload; "image": a measurement beside it (see --seed-order)
                       "exchange": (a.comm if native_ok and drv is None else "torch.distributed") if exchange else "none",
                       "exchange_note": (("native peer-to-peer exchange regions (no collective launches)" if a.comm == "p2p" else "native RCCL on the track stream") if native_ok and drv is None else "torch.distributed driver") if exchange else "none (1 rank)",
                       # what the attached exchange itself reports (ncclCommCount / opened regions): must equal n_gpus
                       "exchange_ranks_reported": comm_info["ranks"], "exchange_backend_attached": comm_info["backend"],
                       # (round 6) every exchange form that was tried and refused before the one that ran, with the reason
                       "exchange_fallthrough": fallthrough,
                       # time the track stream spends inside the exchanges per frame (RCCL collectives: hipEvent brackets around each call,
                       # cfg.profile = 1; peer-to-peer: the p2p_* kernels), from the same profile leg as per_kernel.  0 on one rank without --force-sharded
                       "exchange_us_per_frame": exchange_us, "exchange_us_per_frame_by_kind": exchange_by_kind,
                       "multi_gpu_hardware_note": "no run of this build at N > 1 on real hardware exists (every box it has seen has one GPU): "
                                                  "the N > 1 path is covered by gloo world-2 / world-3 tests on CPU and by emulated ranks / one-rank communicators on one GPU",
                       # (the figures the README leads with, inside the object the driver's record keeps whole)
                       "pipeline_depth": depth, "extract_batch": batch, "extract": ("dealt" if (a.extract == "dealt" and a.comm == "rccl" and native_ok and drv is None) else "replicated") if exchange else "single rank",
                       "steady_state_frames_per_sec": steady["frames_per_sec"] if steady else None,
                       "node_call_frames_per_sec": node_call, "sequential_ms_per_frame": seq_ms,
                       "parallelism": "map sharded by world tile over %d rank(s); extract of up to %d frame(s) runs "
                                      "ahead of ICP/fusion on its own HIP streams, %d per extract launch" % (world, cap_frames if (depth or batch > 1) else 0, batch)},
            "pipeline_depth": depth, "extract_batch": batch, "warmup_extra_frames": Wm - a.warmup, "sequential_ms_per_frame": seq_ms,
            "stage_ms": {"extract": stage[0], "icp": stage[1], "fuse": stage[2]},
            "steady_state_frames_per_sec": steady["frames_per_sec"] if steady else None, "steady_state": steady,
            "pipeline_fill": fill,
            "hbm_peak_measured_GBs": max(hbm_measured or 0.0, hbm_float4 or 0.0) or None, "hbm_peak_torch_copy_GBs": hbm_measured, "hbm_peak_float4_copy_GBs": hbm_float4,
            # the reference node's real call (host images in, depth pre-filter inside the frame), beside the headline
            # whose frames are HBM-resident and already filtered
            "as_the_reference_node_calls_it_frames_per_sec": node_call,
            "parity": parity,
            "roofline": roofline, "frame_roofline": frame_roofline, "cpu_baseline": cpu, "extras": extras, "kernel_source_sha": kernel_source_sha(), "host_affinity": affinity,
            "per_kernel": per_kernel,
        }
    f.close()
    if dist.is_initialized():
        dist.destroy_process_group()
    if rank == 0:
        sys.stdout.flush()
        ctypes.CDLL(None).fflush(None)             # C-level buffers of the runtime libraries (to stderr)
        os.dup2(saved_stdout, 1)
        print(json.dumps(out), flush=True)


if __name__ == "__main__":
    main()
