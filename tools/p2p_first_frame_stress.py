"""First-frame stress of the peer-to-peer exchange with the ranks as handles of one process (threads): many short cycles
of create -> attach -> a few frames -> destroy, every cycle compared with a one-rank run of the same frames (pose bits of
every frame, n_model summed over the ranks).  Round 2 found rare (~1 % of the cycles) differences in the FIRST frame of a
freshly created group of >= 3 ranks; this tool exists to measure that rate under a given mitigation.

    python tools/p2p_first_frame_stress.py [--cycles 300] [--ranks 3] [--frames 3] [--n-model 200000]"""
import argparse
import ctypes as C
import json
import os
import sys
import threading
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402
from supersurfel_fusion_amd import binding, synthetic  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cycles", type=int, default=300)
    ap.add_argument("--ranks", type=int, default=3)
    ap.add_argument("--frames", type=int, default=3)
    ap.add_argument("--n-model", type=int, default=200000)
    ap.add_argument("--spread-priorities", action="store_true", help="ranks 2, 3 get a normal-priority track stream (hardware queues are per priority level)")
    ap.add_argument("--records", action="store_true", help="also fetch every rank's own and reduced ICP record after every frame (a legacy-stream copy per frame)")
    ap.add_argument("--check-maps", action="store_true", help="after the last frame compare every handle's frame maps and frame supersurfels with the one-rank run's")
    ap.add_argument("--independent", action="store_true", help="the handles are NOT shards: each holds the whole map (no exchange), "
                                                               "all run side by side -- is it the exchange, or handles running concurrently?")
    a = ap.parse_args()
    lib = binding.load_lab()          # (ssf_dbg_last_icp_record / _device_icp_records: lab build)
    lib.lib.ssf_dbg_last_icp_record.argtypes = [C.c_void_p, C.c_void_p]
    lib.lib.ssf_dbg_device_icp_records.argtypes = [C.c_void_p, C.c_void_p]
    dev = torch.device("cuda", 0)
    W, H = bench.W, bench.H
    model, nvis = synthetic.seed_model_cam0(a.n_model, W, H, stamp=30)
    frames = bench.render_frames(max(a.frames, 2))
    rgb = [torch.from_numpy(x).to(dev) for x, _ in frames]
    dep = [torch.from_numpy(y).to(dev) for _, y in frames]
    vis = np.arange(a.n_model) < nvis

    def group(world):
        if a.independent and world > 1:
            fs = []
            for r in range(world):
                f = binding.Fusion(lib, bench.make_cfg(lib, a.n_model + 65536, 0, 1, None, False, 0, 1))
                f.set_model(model, nvis, 30)
                fs.append(f)
            return fs
        own = synthetic.tile_owner(model["positions"], world, 0.5) if world > 1 else np.zeros(a.n_model, np.int64)
        fs = []
        for r in range(world):
            sel = own == r
            if a.spread_priorities:                  # two track streams per priority level instead of four at the highest
                if r >= 2:
                    os.environ["SSF_TRACK_PRIORITY"] = "0"
                else:
                    os.environ.pop("SSF_TRACK_PRIORITY", None)
            f = binding.Fusion(lib, bench.make_cfg(lib, int(sel.sum()) + 65536, r, world, None, False, 0, 1))
            f.set_model({k: v[sel] for k, v in model.items()}, int((sel & vis).sum()), 30)
            fs.append(f)
        if world > 1:
            for f in fs:
                f.p2p_configure(all_ranks_on_this_device=True)
            regions = [f.p2p_region()[0] for f in fs]
            for f in fs:
                f.p2p_attach_local(regions)
        return fs

    def run(fs):
        world = len(fs)
        out, errs = [[] for _ in fs], []

        def drive(r):
            try:
                for k in range(a.frames):
                    res = fs[r].process_frame_device(rgb[k].data_ptr(), dep[k].data_ptr())
                    rec = np.zeros(29, np.int64)
                    lib.lib.ssf_dbg_last_icp_record(fs[r].h, rec.ctypes.data_as(C.c_void_p))
                    dev_rec = np.zeros(64, np.int64)
                    # (a blocking copy on the legacy stream: while another thread captures its extract graph the runtime
                    # refuses it -- "would make the legacy stream depend on a capturing stream" -- so only where asked for)
                    if a.records:
                        lib.lib.ssf_dbg_device_icp_records(fs[r].h, dev_rec.ctypes.data_as(C.c_void_p))
                    out[r].append((bytes(res.pose), res.icp_iters, res.n_model, res.n_updated, res.n_inserted, res.n_removed, rec, dev_rec))
            except Exception as e:
                errs.append(str(e))
        ts = [threading.Thread(target=drive, args=(r,)) for r in range(world)]
        for t in ts:
            t.start()
        for t in ts:
            t.join()
        return out, errs

    def maps_of(f):
        return dict(label=f.index_map(), plane_depth=f.plane_depth().view(np.uint32), inlier=f.inlier_map(),
                    frame_conf=f.get_frame()["confidences"].view(np.uint32), frame_pos=f.get_frame()["positions"].view(np.uint32),
                    frame_ori=f.get_frame()["orientations"].view(np.uint32))

    ref_fs = group(1)
    ref, _ = run(ref_fs)
    ref = ref[0]
    ref_maps = maps_of(ref_fs[0]) if a.check_maps else None
    ref_model = ref_fs[0].get_model() if a.check_maps else None
    for f in ref_fs:
        f.close()
    import hashlib
    print("reference digest (poses, counters of the one-rank run):", hashlib.sha256(b"".join(x[0] + str(x[1:6]).encode() for x in ref)).hexdigest()[:16], flush=True)
    bad, t0 = [], time.time()
    GOOD = {}
    GOOD_OWN = {}
    for c in range(a.cycles):
        fs = group(a.ranks)
        out, errs = run(fs)
        map_diff = None
        if a.check_maps and not errs:
            map_diff = {}
            for r, f in enumerate(fs):
                m = maps_of(f)
                d = {k: int((m[k] != ref_maps[k]).sum()) for k in m if (m[k] != ref_maps[k]).any()}
                if d:
                    map_diff[r] = d
        for f in fs:
            f.close()
        if map_diff:
            bad.append(dict(cycle=c, maps_differ=map_diff,
                            poses_ok=[out[r][-1][0] == ref[-1][0] for r in range(a.ranks)])); continue
        if errs:
            bad.append(dict(cycle=c, error=errs[0][:120])); continue
        for k in range(a.frames):
            if a.independent:
                okr = [out[r][k][:6] == ref[k][:6] for r in range(a.ranks)]
                if not all(okr):
                    bad.append(dict(cycle=c, frame=k, handles_ok=okr, n_removed=[out[r][k][5] for r in range(a.ranks)], want_removed=ref[k][5],
                                    poses_ok=[out[r][k][0] == ref[k][0] for r in range(a.ranks)]))
                    break
                continue
            poses_ok = all(out[r][k][0] == ref[k][0] for r in range(a.ranks))
            if poses_ok and k not in GOOD_OWN and not a.independent:
                GOOD_OWN[k] = [int(out[r][k][7][32 + 28]) for r in range(a.ranks)]
            if poses_ok and [sum(out[r][k][q] for r in range(a.ranks)) for q in (2, 3, 4, 5)] == [ref[k][q] for q in (2, 3, 4, 5)] and k not in GOOD:
                GOOD[k] = dict(n_model=[out[r][k][2] for r in range(a.ranks)], n_updated=[out[r][k][3] for r in range(a.ranks)],
                               n_inserted=[out[r][k][4] for r in range(a.ranks)], n_removed=[out[r][k][5] for r in range(a.ranks)])
            sums = [sum(out[r][k][q] for r in range(a.ranks)) for q in (2, 3, 4, 5)]
            want = [ref[k][q] for q in (2, 3, 4, 5)]
            if not poses_ok or sums != want:
                bad.append(dict(cycle=c, frame=k, poses_ok=poses_ok, ranks_agree=all(out[r][k][0] == out[0][k][0] for r in range(a.ranks)),
                                iters=[out[r][k][1] for r in range(a.ranks)], want_iters=ref[k][1],
                                counts=dict(zip(("n_model", "n_updated", "n_inserted", "n_removed"), zip(want, sums))),
                                per_rank=dict(n_model=[out[r][k][2] for r in range(a.ranks)], n_updated=[out[r][k][3] for r in range(a.ranks)],
                                              n_inserted=[out[r][k][4] for r in range(a.ranks)], n_removed=[out[r][k][5] for r in range(a.ranks)]),
                                good_per_rank=GOOD.get(k),
                                own_inliers=[int(out[r][k][7][32 + 28]) for r in range(a.ranks)], good_own_inliers=GOOD_OWN.get(k),
                                sum_of_own_equals_reduced=[bool((sum(out[q][k][7][32:61] for q in range(a.ranks)) == out[r][k][7][0:29]).all()) for r in range(a.ranks)],
                                last_record_inliers=(int(ref[k][6][28]), [int(out[r][k][6][28]) for r in range(a.ranks)]),
                                last_record_diff=[int(v) for v in (out[0][k][6] - ref[k][6])[[0, 5, 20, 21, 27, 28]]]))
                break
    print(json.dumps(dict(cycles=a.cycles, ranks=a.ranks, frames=a.frames, bad_cycles=len(bad), seconds=round(time.time() - t0, 1), first=bad[:3], kinds=sorted(set((b.get('frame'), b.get('poses_ok'), str(b.get('counts'))) for b in bad))[:8])))


if __name__ == "__main__":
    main()
