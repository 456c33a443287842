"""One rank of the cross-process peer-to-peer tests (tests/test_p2p_gpu.py): its own process, its own HIP context, the
exchange regions of the other ranks opened through their IPC handles.

    python tests/p2p_worker.py <config.json> <rank>

config: world, dir, W, H, frames, pipelined; optional: seed_n (rows of a seeded map, sharded by tile), tile, capacity,
tum (TUM-shaped frames: u16 depth at 5000 / m, 25 % holes, benchmark launch parameters, pre-filter on), deform_after (k: one
synthetic loop-closure deformation + the re-homing sweep after k frames), depth / batch (pipeline).
Handles and re-homing tables are traded through files in <dir>; the results go to <dir>/out<rank>.npz."""
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from supersurfel_fusion_amd import binding, replay, synthetic  # noqa: E402

KEYS = ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")


def trade(d, name, rank, world, arr):
    """every rank's array, in rank order, through files"""
    tmp = os.path.join(d, "%s%d.tmp.npy" % (name, rank))
    np.save(tmp, arr)
    os.replace(tmp, os.path.join(d, "%s%d.npy" % (name, rank)))
    out, t0 = [], time.time()
    for r in range(world):
        p = os.path.join(d, "%s%d.npy" % (name, r))
        while not os.path.exists(p):
            assert time.time() - t0 < 300, "rank %d never wrote %s" % (r, name)
            time.sleep(0.01)
        out.append(np.load(p))
    return out


def frames_of(cfg):
    """the frames every rank (and the oracle in the parent) processes"""
    W, H, nf = cfg["W"], cfg["H"], cfg["frames"]
    if cfg.get("seed_n"):
        base = [util.frame(k, W, H, holes=0.25 if cfg.get("tum") else 0.0) for k in range(min(nf, 6))]
        order = [(i % (2 * len(base) - 2)) if len(base) > 1 else 0 for i in range(nf)]
        fr = [base[j if j < len(base) else 2 * len(base) - 2 - j] for j in order]
    else:
        fr = [util.frame(k, W, H) for k in range(nf)]
    if cfg.get("tum"):
        fr = [(rgb, replay.convert_depth(np.clip(np.rint(d.astype(np.float64) * 5000.0), 0, 65535).astype(np.uint16), 0.0002)) for rgb, d in fr]
    return [(np.ascontiguousarray(r, np.uint8), np.ascontiguousarray(d, np.float32)) for r, d in fr]


def spread_over_devices(world):
    """On a node with at least `world` GPUs every rank process takes its own device (the stores of the exchange then cross
    xGMI and the regions are fine-grained); on the one-GPU box all ranks share device 0."""
    try:
        import torch
        return torch.cuda.is_available() and torch.cuda.device_count() >= world > 1
    except Exception:
        return False


def make_config(lib, cfg, rank, world, capacity):
    kw = dict(pipeline_depth=cfg.get("depth", 2), extract_batch=cfg.get("batch", 2)) if cfg.get("pipelined") else {}
    if cfg.get("own_device"):
        kw["device_id"] = rank
    if cfg.get("tum"):
        args = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=capacity, rank=rank, nranks=world, shard_tile=cfg.get("tile", 0.25), **kw)
        return lib.default_config(**args)
    return util.make_cfg(lib, cfg["W"], cfg["H"], nb_supersurfels_max=capacity, rank=rank, nranks=world, shard_tile=cfg.get("tile", 0.25), **kw)


def seed_shard(cfg, rank, world):
    """this rank's rows of the seeded map (None: the map starts empty) -> (model, n_visible, capacity)"""
    if not cfg.get("seed_n"):
        return None, 0, cfg.get("capacity", 4096)
    model, nvis = synthetic.seed_model_cam0(cfg["seed_n"], cfg["W"], cfg["H"], stamp=30)
    if world == 1:
        return model, nvis, cfg["seed_n"] + 65536
    sel = synthetic.tile_owner(model["positions"], world, cfg.get("tile", 0.25)) == rank
    vis = np.arange(cfg["seed_n"]) < nvis
    return {k: v[sel] for k, v in model.items()}, int((sel & vis).sum()), int(sel.sum()) + 65536


def run_frames(f, frames, pipelined):
    if pipelined:
        return f.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    return [f.process_frame(r, d) for r, d in frames]


def main():
    cfg = json.load(open(sys.argv[1])); rank = int(sys.argv[2])
    world, d = cfg["world"], cfg["dir"]
    lib = binding.load_product()
    model, nvis, cap = seed_shard(cfg, rank, world)
    own = spread_over_devices(world)                                     # a multi-GPU node: one device per rank, as in production
    f = binding.Fusion(lib, make_config(lib, dict(cfg, own_device=own), rank, world, cap))
    if model is not None:
        f.set_model(model, nvis, 30)
    f.p2p_configure(all_ranks_on_this_device=not own, timeout_s=120.0)   # (one GPU per box here: the ranks are processes that share it)
    f.p2p_attach(np.concatenate(trade(d, "h", rank, world, f.p2p_export())))
    frames = frames_of(cfg)
    k_def = cfg.get("deform_after", -1)
    res = run_frames(f, frames[:k_def] if k_def > 0 else frames, cfg.get("pipelined"))
    moved = -1
    if k_def > 0:
        f.apply_deformation(*util.deformation_for(f.get_model(), cfg.get("nodes", 12), angle=cfg.get("angle", 0.01), shift=cfg.get("shift", 0.004)))
        tables = trade(d, "t", rank, world, f.rehome_begin())
        moved = len(tables[rank])
        turned = f.rehome_end(np.concatenate(tables))
        assert turned == 0, "rank %d turned %d arrival(s) away" % (rank, turned)
        m = f.get_model()
        home = bool((synthetic.tile_owner(m["positions"][m["confidences"] > 0], world, cfg.get("tile", 0.25)) == rank).all())
        np.save(os.path.join(d, "home%d.npy" % rank), np.array([home, moved]))
        res += run_frames(f, frames[k_def:], cfg.get("pipelined"))
    g = f.global_counts()
    m = f.get_model()
    np.savez(os.path.join(d, "out%d.npz" % rank), poses=np.stack([r["pose"] for r in res]),
             counts=np.array([[r[k] for k in KEYS] for r in res], np.int64),
             iters=np.array([r["icp_iters"] for r in res]), valid=np.array([r["icp_valid"] for r in res]),
             global_counts=np.array([g[k] for k in KEYS], np.int64), **{"model_" + k: v for k, v in m.items()})


if __name__ == "__main__":
    main()
