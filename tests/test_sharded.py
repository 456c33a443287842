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


def _worker(rank, world, port, outdir, extract="replicated"):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = binding.Library(ORACLE_LIB)
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=rank, nranks=world, shard_tile=0.25))
    drv = sharded.ShardedFusion(f, extract=extract)
    poses, glob, here = [], [], 0
    for k in range(NF):
        rgb, depth = util.frame(k, W, H)
        if extract == "dealt" and k % world != rank:
            rgb, depth = np.zeros_like(rgb), np.zeros_like(depth)       # a rank that is not this frame's extractor never looks at the images
        r = drv.process_frame(rgb, depth)
        here += 1 if r.get("extracted_here", True) else 0
        poses.append(r["pose"]); glob.append([r["global_n_model"], r["global_n_visible"], r["icp_valid"], r["icp_iters"]])
    if extract == "dealt":
        assert here == len(range(rank, NF, world)), (rank, here)        # 1 / world of the extract work
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


@pytest.mark.parametrize("world,extract", [(2, "replicated"), (3, "replicated"), (2, "dealt"), (3, "dealt")])
def test_sharded_map_equals_single_rank_map(world, extract, oracle_lib, tmp_path):
    """extract = "dealt" (round 5): frame k is extracted by rank k % world only, which broadcasts label map + plane depth + frame
    supersurfels; the other ranks are handed BLACK images for that frame, so a rank that looked at its own
    input would diverge at once.  Poses, counters and the union of the shards stay the single-rank run's, bit for bit."""
    port = _free_port()
    mp.spawn(_worker, args=(world, port, str(tmp_path), extract), nprocs=world, join=True)
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


def frames_through_tables(lib_extract, lib_track, W_=W, H_=H, nf=5, **kw):
    """Handle A (library lib_extract) runs every frame and hands out what its extract stage produced; handle B (lib_track) is
    fed A's label map + plane depth + frame supersurfels through ssf_submit_frame_tables for every other frame (and the images
    for the rest) and tracks / fuses from them.  A third
    handle (lib_track, plain process_frame) is the reference.  Returns (B, reference) after comparing every frame's result."""
    fa = binding.Fusion(lib_extract, util.make_cfg(lib_extract, W_, H_, nb_supersurfels_max=4096, **kw))
    fb = binding.Fusion(lib_track, util.make_cfg(lib_track, W_, H_, nb_supersurfels_max=4096, **kw))
    fr = binding.Fusion(lib_track, util.make_cfg(lib_track, W_, H_, nb_supersurfels_max=4096))
    for k in range(nf):
        rgb, depth = util.frame(k, W_, H_, noise=True, holes=0.03)
        want = fr.process_frame(rgb, depth)
        fa.process_frame(rgb, depth)                                   # (A is a rank in step with the others: its frame carries the frame's stamp)
        words = sharded.pack_frame_tables(fa)
        assert len(words) == sharded.frame_tables_words(fa)
        if k % 2 == 0:                                                 # foreign and local frames may alternate freely
            fb.submit_frame_tables(*sharded.unpack_frame_tables(fb, words))
        else:
            fb.submit_frame(rgb, depth)
        assert fb.pending_frames() == 1
        got = fb.process_submitted()
        got = got if isinstance(got, dict) else got.as_dict()
        util.same_result(want, got)
        util.assert_same_bits(fb.index_map(), fr.index_map(), "label map of frame %d" % k)
        util.assert_same_bits(fb.plane_depth(), fr.plane_depth(), "plane depth of frame %d" % k)
        a, b = fb.get_frame(), fr.get_frame()
        for name in a:
            util.assert_same_bits(a[name], b[name], "frame.%s of frame %d" % (name, k))
    return fb, fr


def test_a_frame_extracted_elsewhere_tracks_like_a_local_one(oracle_lib):
    """ssf_submit_frame_tables on the checker: the map after five frames, three of them handed over as tables, equals the plain run's"""
    fb, fr = frames_through_tables(oracle_lib, oracle_lib)
    util.compare_state(fb, fr, maps=False)
    fb2, fr2 = frames_through_tables(oracle_lib, oracle_lib, pipeline_depth=1, extract_batch=2)       # (a pipelined handle: a foreign frame is a batch of its own)
    util.compare_state(fb2, fr2, maps=False)


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


def _drive_emulated(fs, k):
    """one frame of a sharded map whose ranks are handles of this process (exchanges on the host between the stage calls)"""
    world = len(fs)
    rgb, depth = util.frame(k, W, H)
    counts = np.array([[f.counts()["n_model"], f.counts()["n_visible"]] for f in fs], np.int64)
    for f in fs:
        f.stage_extract(rgb, depth)
    g_model, g_vis = int(counts[:, 0].sum()), int(counts[:, 1].sum())
    for r, f in enumerate(fs):
        f.set_shard(int(counts[:r, 1].sum()), g_model, g_vis)
        f.icp_begin()
    again = g_vis > 0
    while again:
        total = sum(f.icp_accumulate() for f in fs)
        again = [f.icp_update(total) for f in fs][0]
    for f in fs:
        f.icp_end()
    bm = [f.match() for f in fs]
    return util.exchange_and_fuse(fs, np.minimum.reduce([b for b, _ in bm]), np.maximum.reduce([m for _, m in bm]))


@pytest.mark.parametrize("world", [2, 3])
def test_rehoming_after_a_deformation(world, oracle_lib):
    """applyDeformation moves every row: after it a shard holds rows of other ranks' tiles until ssf_rehome_begin / _end have
    run; then the owner invariant holds at once, the union of the shards is the deformed unsharded map row for row, and the
    frames that follow give every rank the unsharded pose"""
    from supersurfel_fusion_amd import synthetic
    tile = 0.25
    single = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    fs = [binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096, rank=r, nranks=world, shard_tile=tile)) for r in range(world)]
    for k in range(3):
        single.process_frame(*util.frame(k, W, H))
        _drive_emulated(fs, k)
    for f in [single] + fs:
        m = f.get_model()
        f.apply_deformation(*util.deformation_for(m, 12, angle=0.05, shift=0.08))
    strays = sum(int((synthetic.tile_owner(f.get_model()["positions"][f.get_model()["confidences"] > 0], world, tile) != r).sum()) for r, f in enumerate(fs))
    assert strays > 0, "the deformation was meant to push rows over tile edges"
    moved = util.rehome_in_process(fs)
    assert sum(moved) == strays
    for r, f in enumerate(fs):
        m = f.get_model()
        ok = m["confidences"] > 0
        assert (synthetic.tile_owner(m["positions"][ok], world, tile) == r).all(), "a row is not on its tile's owner right after the sweep"
    merged = {name: np.concatenate([f.get_model()[name] for f in fs]) for name, _, _ in binding.SURFEL_FIELDS}
    assert np.array_equal(util.rows_multiset(merged), util.rows_multiset(single.get_model()))
    assert sum(f.counts()["n_visible"] for f in fs) == single.counts()["n_visible"]
    for k in range(3, 6):
        want = single.process_frame(*util.frame(k, W, H))
        res = _drive_emulated(fs, k)
        for r_ in res:
            util.assert_same_bits(r_["pose"], want["pose"], "pose after the sweep, frame %d" % k)
        assert sum(r_["n_model"] for r_ in res) == want["n_model"]
    assert util.rehome_in_process(fs) == [0] * world                 # nothing left to move


def rehoming_into_a_full_shard(lib):
    """ssf_rehome_end at a shard without room: the surplus arrivals are turned away in table order and their number is
    returned (never an error: the source ranks have already let the rows go, an error on one rank could not be rolled back);
    the rows that fit arrive as usual.  Shared by the CPU test (oracle) and tests/test_parity_gpu.py (product)."""
    from supersurfel_fusion_amd import synthetic
    world, tile = 2, 0.25
    whole = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096))
    whole.process_frame(*util.frame(0, W, H))
    m = whole.get_model()
    ok = m["confidences"] > 0
    src = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=1, nranks=world, shard_tile=tile))
    src.set_model(m, whole.counts()["n_visible"], 1)                             # rank 1 holding the whole map: rank 0's rows must leave
    theirs = ok & (synthetic.tile_owner(m["positions"], world, tile) == 0)
    table = src.rehome_begin()
    assert len(table) == int(theirs.sum()) > 8 and (table[:, 0] == 1).all()
    S = ((W + 15) // 16) * ((H + 15) // 16)          # (a handle's capacity cannot be below its superpixel count)
    n0 = S + 11
    room = 5
    dst = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=n0 + room, rank=0, nranks=world, shard_tile=tile))
    pick = np.flatnonzero(ok)[np.arange(n0) % int(ok.sum())]
    dst.set_model({k: v[pick] for k, v in m.items()}, n0 // 2, 1)
    before = dst.get_model()
    turned = dst.rehome_end(table)
    assert turned == len(table) - room
    after = dst.get_model()
    assert dst.counts()["n_model"] == n0 + room
    nvis_arr = int(table[:room, 1].sum())
    assert dst.counts()["n_visible"] == n0 // 2 + nvis_arr
    # the first `room` records of the table are the ones that arrived: visible-flagged behind the visible block, the others at the end
    arrived = np.concatenate([after["positions"][n0 // 2:n0 // 2 + nvis_arr], after["positions"][n0 + nvis_arr:]])
    want = np.concatenate([table[:room][table[:room, 1] == 1][:, 2:5], table[:room][table[:room, 1] == 0][:, 2:5]]).view(np.float32)
    assert np.array_equal(arrived.view(np.uint32), want.view(np.uint32))
    assert np.array_equal(after["positions"][:n0 // 2], before["positions"][:n0 // 2])
    assert dst.rehome_end(table[:0]) == 0
    return turned


def test_rehoming_into_a_full_shard_turns_the_surplus_away(oracle_lib):
    assert rehoming_into_a_full_shard(oracle_lib) > 0


def _full_shard_worker(rank, world, port, outdir):
    """the set-up of rehoming_into_a_full_shard over gloo: rank 1 holds the whole map, rank 0 has room for five arrivals"""
    import warnings
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = binding.Library(ORACLE_LIB)
    whole = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096))
    whole.process_frame(*util.frame(0, W, H))
    m = whole.get_model()
    ok = m["confidences"] > 0
    S = ((W + 15) // 16) * ((H + 15) // 16)
    n0, room = S + 11, 5
    if rank == 1:
        f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=1, nranks=world, shard_tile=0.25))
        f.set_model(m, whole.counts()["n_visible"], 1)
    else:
        f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=n0 + room, rank=0, nranks=world, shard_tile=0.25))
        from supersurfel_fusion_amd import synthetic
        mine = np.flatnonzero(ok & (synthetic.tile_owner(m["positions"], world, 0.25) == 0))      # rows that stay: no room is freed
        pick = mine[np.arange(n0) % len(mine)]
        f.set_model({k: v[pick] for k, v in m.items()}, n0 // 2, 1)
    with warnings.catch_warnings(record=True) as caught:
        warnings.simplefilter("always")
        rep = sharded.rehome_over(f, world)
    raised = False
    try:
        sharded.rehome_over(f, world, on_loss="raise")             # (second sweep: nothing left to move, nothing lost)
    except RuntimeError:
        raised = True
    np.save(os.path.join(outdir, "lost%d.npy" % rank), np.array([int(rep), rep.turned_away, len(caught), int(raised)]))
    dist.destroy_process_group()


def test_rows_lost_in_a_rehoming_sweep_are_reported_on_every_rank(tmp_path):
    """ssf_rehome_end's positive return (arrivals a full shard turned away) used to be dropped by sharded.rehome_over: rows
    that had already left their source shard vanished from the global map without a signal (advisor, round 4).  Now the
    count is summed over the ranks, every rank's report carries it and warns."""
    world = 2
    mp.spawn(_full_shard_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    got = [np.load(os.path.join(str(tmp_path), "lost%d.npy" % r)) for r in range(world)]
    assert got[0][0] == got[1][0] > 5                       # rows that changed rank
    assert got[0][1] == got[1][1] == got[0][0] - 5          # all but the five that fitted: the same number on BOTH ranks
    assert got[0][2] >= 1 and got[1][2] >= 1                # both warned
    assert got[0][3] == 0 and got[1][3] == 0                # the clean second sweep does not raise


def _rehome_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    lib = binding.Library(ORACLE_LIB)
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=rank, nranks=world, shard_tile=0.25))
    drv = sharded.ShardedFusion(f)
    for k in range(3):
        drv.process_frame(*util.frame(k, W, H))
    f.apply_deformation(*util.deformation_for(f.get_model(), 12, angle=0.05, shift=0.08))
    drv.rehome()
    poses = [drv.process_frame(*util.frame(k, W, H))["pose"] for k in range(3, 5)]
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), poses=np.array(poses), **f.get_model())
    dist.destroy_process_group()


def test_rehoming_over_torch_distributed(oracle_lib, tmp_path):
    """the same sweep through sharded.ShardedFusion.rehome (gloo, world 2): all-gather of the padded tables"""
    from supersurfel_fusion_amd import synthetic
    world = 2
    mp.spawn(_rehome_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    single = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    for k in range(3):
        single.process_frame(*util.frame(k, W, H))
    single.apply_deformation(*util.deformation_for(single.get_model(), 12, angle=0.05, shift=0.08))
    poses = [single.process_frame(*util.frame(k, W, H))["pose"] for k in range(3, 5)]
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r, d in enumerate(ranks):
        assert np.array_equal(d["poses"].view(np.uint32), np.array(poses).view(np.uint32))
        ok = d["confidences"] > 0
        assert (synthetic.tile_owner(d["positions"][ok], world, 0.25) == r).all()
    merged = {name: np.concatenate([d[name] for d in ranks]) for name, _, _ in binding.SURFEL_FIELDS}
    assert np.array_equal(util.rows_multiset(merged), util.rows_multiset(single.get_model()))


# ---- bench.py's own N > 1 loop at world size 2 (round 6) -------------------------------------------------------------------------
# `python bench.py --gpus N` falls through RCCL -> peer-to-peer regions -> the torch.distributed driver; the last form, and
# `--py-driver`, run bench.drive_pipelined over a sharded.ShardedFusion.  No box this build has seen has two GPUs, so that loop is run
# HERE: two gloo ranks, the CPU checker as the engine, frames submitted ahead (pipeline_depth 1, two frames per extract batch).
def _bench_loop_worker(rank, world, port, outdir):
    os.environ["MASTER_ADDR"] = "127.0.0.1"; os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    import bench
    lib = binding.Library(ORACLE_LIB)
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=rank, nranks=world, shard_tile=0.25, pipeline_depth=1, extract_batch=2))
    drv = sharded.ShardedFusion(f)
    frames = [util.frame(k, W, H) for k in range(NF)]
    res = bench.drive_pipelined(drv, lambda i: frames[i], 0, NF, on_device=False)
    m = f.get_model()
    np.savez(os.path.join(outdir, "rank%d.npz" % rank), poses=np.array([r["pose"] for r in res]),
             glob=np.array([[r["global_n_model"], r["global_n_visible"], r["icp_valid"], r["icp_iters"]] for r in res]), **m)
    dist.destroy_process_group()


def test_bench_py_driver_loop_at_world_size_two(oracle_lib, tmp_path):
    world = 2
    mp.spawn(_bench_loop_worker, args=(world, _free_port(), str(tmp_path)), nprocs=world, join=True)
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=4096))
    want = [fo.process_frame(*util.frame(k, W, H)) for k in range(NF)]
    ranks = [np.load(os.path.join(str(tmp_path), "rank%d.npz" % r)) for r in range(world)]
    for r in ranks:                                   # every rank holds the single-rank pose and the global counters, frame by frame
        assert np.array_equal(r["poses"].view(np.uint32), np.array([w["pose"] for w in want]).view(np.uint32))
        assert np.array_equal(r["glob"], np.array([[w["n_model"], w["n_visible"], w["icp_valid"], w["icp_iters"]] for w in want]))
    union = {name: np.concatenate([r[name] for r in ranks]) for name, _, _ in binding.SURFEL_FIELDS}
    assert np.array_equal(_rows(union), _rows(fo.get_model()))
