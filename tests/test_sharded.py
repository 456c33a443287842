"""The N>1 path on CPU: two processes, torch.distributed backend "gloo", the sharded driver
(supersurfel_fusion_amd/sharded.py) over the checker engine.  Because every exchanged quantity is
an exact integer (ICP record) or order-free (MIN/MAX), the union of the rank-local maps must equal
the single-rank map bit for bit, and every rank must hold the single-rank pose."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import util
from conftest import ORACLE_LIB
from supersurfel_fusion_amd import binding, sharded

W, H, NF = 160, 128, 6


def _free_port():
    s = socket.socket(); s.bind(("127.0.0.1", 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = binding.Library(ORACLE_LIB)
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=rank, nranks=world, shard_tile=0.25))
    drv = sharded.ShardedFusion(f)
    poses, glob = [], []
    for k in range(NF):
        rgb, depth = util.frame(k, W, H)
        r = drv.process_frame(rgb, depth)
        poses.append(r["pose"]); glob.append([r["global_n_model"], r["global_n_visible"], r["icp_valid"], r["icp_iters"]])
    m = f.get_model()
    from supersurfel_fusion_amd import synthetic
    at_home = bool((synthetic.tile_owner(m["positions"][m["confidences"] > 0], world, 0.25) == rank).all())     # migration keeps rows with their tile's owner
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), poses=np.array(poses), glob=np.array(glob), at_home=at_home, **m)
    dist.destroy_process_group()


def _rows(m):
    """canonical multiset of model rows: each row as bytes, sorted"""
    n = len(m["confidences"])
    cols = [np.ascontiguousarray(m[name]).reshape(n, -1).view(np.uint32) for name, _, _ in binding.SURFEL_FIELDS]
    rows = np.concatenate(cols, axis=1)
    return rows[np.lexsort(rows.T[::-1])]


@pytest.mark.parametrize("world", [2, 3])
def test_sharded_map_equals_single_rank_map(world, oracle_lib, tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path)), nprocs=world, join=True)
    # single rank reference through the same driver
    f = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    drv = sharded.ShardedFusion(f)
    poses, glob = [], []
    for k in range(NF):
        rgb, depth = util.frame(k, W, H)
        r = drv.process_frame(rgb, depth)
        poses.append(r["pose"]); glob.append([r["n_model"], r["n_visible"], r["icp_valid"], r["icp_iters"]])
    single = f.get_model()
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in ranks:
        assert bool(r["at_home"]), "a row lives on a rank that does not own its world tile"
        assert np.array_equal(r["poses"].view(np.uint32), np.array(poses).view(np.uint32)), "pose differs across shard counts"
        assert np.array_equal(r["glob"], np.array(glob))
    merged = {name: np.concatenate([r[name] for r in ranks]) for name, _, _ in binding.SURFEL_FIELDS}
    assert len(merged["confidences"]) == len(single["confidences"])
    assert np.array_equal(_rows(merged), _rows(single)), "union of shards != single-rank map"
    sizes = [len(r["confidences"]) for r in ranks]
    assert min(sizes) > 0, sizes   # the tile hash spreads the map over every rank


def test_sharded_driver_equals_process_frame(oracle_lib):
    """world_size 1: the stage-seam driver is the same computation as ssf_process_frame."""
    fa = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    fb = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    drv = sharded.ShardedFusion(fb)
    for k in range(3):
        rgb, depth = util.frame(k, W, H)
        ra = fa.process_frame(rgb, depth); rb = drv.process_frame(rgb, depth)
        util.same_result(ra, rb)
    util.compare_state(fa, fb)
