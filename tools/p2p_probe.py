"""Latency of the peer-to-peer exchange protocol with every rank on ONE GPU (the stores land in the same HBM: a lower
bound for xGMI, and the only multi-rank measurement a one-GPU box allows -- RCCL refuses two ranks on one device).

    python tools/p2p_probe.py [--ranks 1 2 4] [--frames 200] [--n-model 1000000]

Every rank is its own process; the 640x480 orbit of bench.py, the map sharded by world tile, frames resident in HBM, one
frame in flight (pipeline_depth 0: the latency-bound call pattern, where the exchange is fully exposed).  All ranks
extract the same frame, so the GPU does that work `ranks` times over -- read the ICP column, not the frame rate."""
import argparse
import json
import os
import subprocess
import sys
import tempfile
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def worker(rank, world, d, nf, n_model):
    import torch
    import bench
    from supersurfel_fusion_amd import binding, synthetic
    lib = binding.load_product()
    dev = torch.device("cuda", 0)
    W, H = bench.W, bench.H
    model, nvis = synthetic.seed_model_cam0(n_model, W, H, stamp=30)
    if world > 1:
        own = synthetic.tile_owner(model["positions"], world, 0.5) == rank
        vis = np.arange(n_model) < nvis
        nvis = int((own & vis).sum())
        model = {k: v[own] for k, v in model.items()}
    f = binding.Fusion(lib, bench.make_cfg(lib, len(model["confidences"]) + 65536, rank, world, None, False, 0, 1))
    f.set_model(model, nvis, 30)
    if world > 1:
        f.p2p_configure(all_ranks_on_this_device=True)
        mine = f.p2p_export()
        mine.tofile(os.path.join(d, "h%d.tmp" % rank)); os.replace(os.path.join(d, "h%d.tmp" % rank), os.path.join(d, "h%d.bin" % rank))
        hs = []
        for r in range(world):
            p = os.path.join(d, "h%d.bin" % r)
            while not os.path.exists(p):
                time.sleep(0.01)
            hs.append(np.fromfile(p, np.uint8))
        f.p2p_attach(np.concatenate(hs))
    frames = bench.render_frames(8)
    rgb = [torch.from_numpy(a).to(dev) for a, _ in frames]
    dep = [torch.from_numpy(b).to(dev) for _, b in frames]
    order = [0, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4, 3, 2, 1]
    res = []
    for k in range(20):
        j = order[k % len(order)]
        f.process_frame_device(rgb[j].data_ptr(), dep[j].data_ptr())
    open(os.path.join(d, "ready%d" % rank), "w").close()          # start the timed part together
    while not all(os.path.exists(os.path.join(d, "ready%d" % r)) for r in range(world)):
        time.sleep(0.001)
    t0 = time.perf_counter()
    for k in range(20, 20 + nf):
        j = order[k % len(order)]
        res.append(f.process_frame_device(rgb[j].data_ptr(), dep[j].data_ptr()))
    dt = time.perf_counter() - t0
    json.dump(dict(rank=rank, ms_per_frame=1e3 * dt / nf, icp_iters_mean=float(np.mean([r.icp_iters for r in res])),
                   n_visible=int(res[-1].n_visible), n_model=int(res[-1].n_model)), open(os.path.join(d, "out%d.json" % rank), "w"))


REFERENCE_TRACE = None
KEEP_HANDLES = False
KEPT = []


def threads_mode(world, nf, n_model):
    """the same ranks as handles of ONE process, one host thread each (ssf_p2p_attach_local)"""
    import threading
    import torch
    import bench
    from supersurfel_fusion_amd import binding, synthetic
    lib = binding.load_product()
    dev = torch.device("cuda", 0)
    W, H = bench.W, bench.H
    model, nvis = synthetic.seed_model_cam0(n_model, W, H, stamp=30)
    own = synthetic.tile_owner(model["positions"], world, 0.5) if world > 1 else np.zeros(n_model, np.int64)
    vis = np.arange(n_model) < nvis
    fs = []
    for r in range(world):
        sel = own == r
        f = binding.Fusion(lib, bench.make_cfg(lib, int(sel.sum()) + 65536, r, world, None, False, 0, 1))
        f.set_model({k: v[sel] for k, v in model.items()}, int((sel & vis).sum()), 30)
        fs.append(f)
    if world > 1:
        for f in fs:
            f.p2p_configure(all_ranks_on_this_device=True)
        regions = [f.p2p_region()[0] for f in fs]
        for f in fs:
            f.p2p_attach_local(regions)
    frames = bench.render_frames(8)
    rgb = [torch.from_numpy(a).to(dev) for a, _ in frames]
    dep = [torch.from_numpy(b).to(dev) for _, b in frames]
    order = [0, 1, 2, 3, 4, 5, 6, 7, 6, 5, 4, 3, 2, 1]
    times, iters = [0.0] * world, [0.0] * world
    trace = [[] for _ in range(world)]                 # (pose bits, icp_iters, global n_visible is implied by the pose) per timed frame
    barrier = threading.Barrier(world)

    def drive(r):
        for k in range(20):
            j = order[k % len(order)]
            res = fs[r].process_frame_device(rgb[j].data_ptr(), dep[j].data_ptr())
            trace[r].append((bytes(res.pose), res.icp_iters, res.icp_valid, res.n_model, res.n_visible, res.n_removed, res.n_inserted, res.n_updated))
        barrier.wait()
        t0 = time.perf_counter()
        it = 0
        for k in range(20, 20 + nf):
            j = order[k % len(order)]
            res = fs[r].process_frame_device(rgb[j].data_ptr(), dep[j].data_ptr())
            it += res.icp_iters
            trace[r].append((bytes(res.pose), res.icp_iters, res.icp_valid, res.n_model, res.n_visible, res.n_removed, res.n_inserted, res.n_updated))
        times[r] = 1e3 * (time.perf_counter() - t0) / nf
        iters[r] = it / nf

    ts = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
    for t in ts:
        t.start()
    for t in ts:
        t.join()
    # the pose does not depend on the number of ranks: every rank of every run must reproduce the one-rank trace bit for bit
    global REFERENCE_TRACE
    bad = None
    if world == 1 and REFERENCE_TRACE is None:
        REFERENCE_TRACE = trace[0]
    elif REFERENCE_TRACE is not None:
        names = ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")
        for k, a in enumerate(REFERENCE_TRACE):
            if bad is not None:
                break
            for r in range(world):
                b = trace[r][k]
                if a[:3] != b[:3]:
                    import struct
                    bad = dict(frame=k, rank=r, what="pose / iterations / validity", want_iters=a[1], got_iters=[trace[q][k][1] for q in range(world)],
                               ranks_equal_to_reference=[trace[q][k][0] == a[0] for q in range(world)],
                               ranks_equal_to_rank0=[trace[q][k][0] == trace[0][k][0] for q in range(world)],
                               want_t=struct.unpack("3f", a[0][36:48]), got_t=[struct.unpack("3f", trace[q][k][0][36:48]) for q in range(world)],
                               later_frames_differing=sum(1 for kk in range(k, len(REFERENCE_TRACE)) if trace[0][kk][0] != REFERENCE_TRACE[kk][0]))
                    break
            if bad is None:
                for q, nm in enumerate(names):
                    tot = sum(trace[r][k][3 + q] for r in range(world))
                    if tot != a[3 + q]:
                        bad = dict(frame=k, what=nm, want=a[3 + q], got=[trace[r][k][3 + q] for r in range(world)])
                        break
    g = [f.global_counts() for f in fs] if world > 1 else []
    print(json.dumps(dict(ranks=world, mode="threads of one process", ms_per_frame=max(times), icp_iters_mean=iters[0],
                          first_mismatch_vs_one_rank=bad, global_counts_agree=all(x == g[0] for x in g) if g else None)), flush=True)
    if not KEEP_HANDLES:
        for f in fs:
            f.close()
    else:
        KEPT.extend(fs)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--ranks", type=int, nargs="+", default=[1, 2, 4])
    ap.add_argument("--frames", type=int, default=200)
    ap.add_argument("--n-model", type=int, default=1000000)
    ap.add_argument("--worker", type=int, nargs=2, default=None)
    ap.add_argument("--dir", default=None)
    ap.add_argument("--threads", action="store_true", help="the ranks as handles of one process, one host thread each")
    ap.add_argument("--keep", action="store_true", help="--threads: keep the handles of earlier runs alive (their streams keep their hardware queues)")
    a = ap.parse_args()
    if a.threads:
        global KEEP_HANDLES
        KEEP_HANDLES = a.keep
        for world in a.ranks:
            threads_mode(world, a.frames, a.n_model)
        return
    if a.worker:
        worker(a.worker[0], a.worker[1], a.dir, a.frames, a.n_model)
        return
    for world in a.ranks:
        with tempfile.TemporaryDirectory() as d:
            env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
            ps = [subprocess.Popen([sys.executable, os.path.abspath(__file__), "--worker", str(r), str(world), "--dir", d, "--frames", str(a.frames),
                                    "--n-model", str(a.n_model)], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
            logs = []
            for p in ps:
                try:
                    logs.append(p.communicate(timeout=600)[0])
                except subprocess.TimeoutExpired:
                    p.kill(); logs.append(p.communicate()[0])
            if any(p.returncode for p in ps):
                print("ranks=%d FAILED\n%s" % (world, "\n".join(l[-1500:] for l in logs)))
                continue
            outs = [json.load(open(os.path.join(d, "out%d.json" % r))) for r in range(world)]
            print(json.dumps(dict(ranks=world, ms_per_frame=max(o["ms_per_frame"] for o in outs), icp_iters_mean=outs[0]["icp_iters_mean"],
                                  n_visible=[o["n_visible"] for o in outs], n_model=[o["n_model"] for o in outs])))


if __name__ == "__main__":
    main()
