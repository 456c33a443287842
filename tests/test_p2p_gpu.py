"""The peer-to-peer exchange backend (ssf_p2p_* in include/ssf.h; SURVEY.md section 5 / 8e: "fixed-order P2P mailbox"):
the ranks of a sharded map trade the ICP record, the association tables, the migrant table and the shard sizes through
each other's HBM, no collective launches.  On the one GPU of this box the ranks are

  * handles of ONE process, each driven by its own host thread (ssf_p2p_attach_local), and
  * separate PROCESSES whose regions are opened through IPC handles (ssf_p2p_export / ssf_p2p_attach) --
    the arrangement of a real node, with every "remote" store landing in the same HBM.

Either way every rank must hold, bit for bit, what the CPU oracle computes when the same exchanges are done on the
host between its stage calls (test_parity_gpu._emulated_ranks).  What one GPU cannot show -- the stores crossing xGMI --
is not claimed."""
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

import util
from conftest import ROOT
from supersurfel_fusion_amd import binding
from test_parity_gpu import _emulated_ranks

pytestmark = pytest.mark.gpu
KEYS = ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")


def check_against_oracle(world, per_rank, models, oracle_out, oracle_handles):
    """per_rank[r] = (poses [nf,12], counts [nf,5]); oracle_out as _emulated_ranks returns it"""
    nf = len(oracle_out)
    for k in range(nf):
        poses, counts, frame_counters = oracle_out[k]
        for r in range(world):
            util.assert_same_bits(per_rank[r][0][k], poses[r], "pose of frame %d on rank %d" % (k, r))
            assert [int(v) for v in per_rank[r][1][k][:2]] == [int(v) for v in counts[r]], (k, r)
            assert [int(v) for v in per_rank[r][1][k][2:]] == frame_counters[r], (k, r)
    for r in range(world):
        mo = oracle_handles[r].get_model()
        assert len(mo["confidences"]) > 0
        for name in mo:
            util.assert_same_bits(models[r][name], mo[name], "%s of rank %d" % (name, r))


@pytest.mark.parametrize("world,pipelined,nf", [(2, False, 6), (3, False, 6), (2, True, 6), (3, False, 40), (3, True, 40)])
def test_ranks_in_one_process_bit_exact(world, pipelined, nf, oracle_lib, product_lib):
    W, H = 320, 240
    kw = dict(pipeline_depth=2, extract_batch=2) if pipelined else {}
    fs = [binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=4096, rank=r, nranks=world, shard_tile=0.25, **kw))
          for r in range(world)]
    for f in fs:
        f.p2p_configure(all_ranks_on_this_device=True)         # (one GPU per box: every rank is a handle on it)
    regions = [f.p2p_region()[0] for f in fs]
    for f in fs:
        f.p2p_attach_local(regions)
    assert [f.comm_info() for f in fs] == [dict(backend="p2p", ranks=world, rank=r) for r in range(world)]
    frames = [util.frame(k, W, H) for k in range(nf)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    out, errors = [None] * world, []

    def drive(r):
        try:
            if pipelined:
                out[r] = fs[r].process_sequence([a.ctypes.data for a, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
            else:
                out[r] = [fs[r].process_frame(a, d) for a, d in frames]
        except Exception as e:                      # a rank that fails leaves its peers waiting: they time out and fail too
            errors.append((r, e))

    threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
    for t in threads:
        t.start()
    for t in threads:
        t.join(180)
    assert not errors, errors
    assert all(o is not None and len(o) == nf for o in out)
    fo, oo = _emulated_ranks(oracle_lib, world, W, H, nf)
    per_rank = [(np.stack([x["pose"] for x in o]), np.array([[x[k] for k in KEYS] for x in o], np.int64)) for o in out]
    check_against_oracle(world, per_rank, [f.get_model() for f in fs], oo, fo)
    g = fs[0].global_counts()
    assert g["n_model"] == sum(int(c) for c in oo[-1][1][:, 0]) and g["n_visible"] == sum(int(c) for c in oo[-1][1][:, 1])
    assert sum(x["icp_iters"] for x in out[0]) >= nf - 1                   # the ICP exchange did run


def run_workers(cfg, tmp_path, timeout=600):
    """one process per rank (tests/p2p_worker.py) -> their output archives"""
    import json
    cfg = dict(cfg, dir=str(tmp_path))
    cpath = os.path.join(str(tmp_path), "config.json")
    json.dump(cfg, open(cpath, "w"))
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), cpath, str(r)], env=env, stdout=subprocess.PIPE,
                              stderr=subprocess.STDOUT, text=True) for r in range(cfg["world"])]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=timeout)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append(p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    return [np.load(os.path.join(str(tmp_path), "out%d.npz" % r)) for r in range(cfg["world"])]


@pytest.mark.parametrize("world,pipelined", [(2, False), (2, True), (3, False), (4, True)])
def test_ranks_in_separate_processes_bit_exact(world, pipelined, oracle_lib, tmp_path):
    W, H, nf = 320, 240, 6
    outs = run_workers(dict(world=world, W=W, H=H, frames=nf, pipelined=pipelined), tmp_path, 300)
    fo, oo = _emulated_ranks(oracle_lib, world, W, H, nf)
    per_rank = [(o["poses"], o["counts"]) for o in outs]
    models = [{k[len("model_"):]: o[k] for k in o.files if k.startswith("model_")} for o in outs]
    check_against_oracle(world, per_rank, models, oo, fo)
    for o in outs:
        assert [int(v) for v in o["global_counts"][:2]] == [int(oo[-1][1][:, 0].sum()), int(oo[-1][1][:, 1].sum())]


# ---- the native exchange at BASELINE size: every rank against the UNSHARDED oracle --------------------------------------------
@pytest.fixture(scope="module")
def fast_oracle():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "omp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_omp.so"))


def check_against_unsharded(world, tile, want, outs, single_model, rank_models):
    """want: the unsharded oracle's frame results; outs[r]: (poses, counts) of rank r; the maps as get_model() dicts"""
    from supersurfel_fusion_amd import synthetic
    for k, w in enumerate(want):
        for r in range(world):
            util.assert_same_bits(outs[r][0][k], w["pose"], "pose of frame %d on rank %d" % (k, r))
        for j, key in enumerate(KEYS):
            assert sum(int(outs[r][1][k][j]) for r in range(world)) == w[key], (k, key, [int(outs[r][1][k][j]) for r in range(world)], w[key])
    for r, m in enumerate(rank_models):
        ok = m["confidences"] > 0
        assert (synthetic.tile_owner(m["positions"][ok], world, tile) == r).all(), "a row lives on a rank that does not own its tile"
    merged = {name: np.concatenate([m[name] for m in rank_models]) for name, _, _ in binding.SURFEL_FIELDS}
    assert len(merged["confidences"]) == len(single_model["confidences"])
    assert np.array_equal(util.rows_multiset(merged), util.rows_multiset(single_model)), "union of the shards != the unsharded map"


def test_config4_two_million_rows_over_four_processes_native_exchange(fast_oracle, tmp_path):
    """BASELINE config 4 on the NATIVE exchange path: a 2 M-supersurfel map sharded by world tile over 4 ranks = 4 processes
    whose exchange regions are opened through IPC handles (ssf_p2p_export / ssf_p2p_attach), 640x480, 6 frames pipelined:
    every rank holds the UNSHARDED oracle's pose bit for bit, the per-shard counters sum to the oracle's, the union of the
    shards is the oracle's map, every row lives on the owner of its tile.  (All four on the box's one GPU: what crosses
    xGMI on a real node lands in the same HBM here.)"""
    import p2p_worker
    cfg = dict(world=4, W=640, H=480, frames=6, pipelined=True, depth=2, batch=2, seed_n=2000000, tile=0.5)
    fo = binding.Fusion(fast_oracle, p2p_worker.make_config(fast_oracle, dict(cfg, pipelined=False), 0, 1, cfg["seed_n"] + 65536))
    model, nvis, _ = p2p_worker.seed_shard(cfg, 0, 1)
    fo.set_model(model, nvis, 30)
    want = [fo.process_frame(r, d) for r, d in p2p_worker.frames_of(cfg)]
    outs = run_workers(cfg, tmp_path, 900)
    assert want[-1]["icp_valid"] == 1 and sum(w["icp_iters"] for w in want) >= 6
    models = [{k[len("model_"):]: o[k] for k in o.files if k.startswith("model_")} for o in outs]
    check_against_unsharded(4, 0.5, want, [(o["poses"], o["counts"]) for o in outs], fo.get_model(), models)
    assert min(len(m["confidences"]) for m in models) > 300000


def test_config5_tum_shaped_three_threads_deformation_and_rehoming(fast_oracle, product_lib):
    """BASELINE config 5 on the native exchange path, ranks as threads of one process (ssf_p2p_attach_local): TUM-shaped frames
    (u16 depth at 5000 / m, 25 % holes, benchmark launch parameters, pre-filter on), a 1 M-row map over 3 ranks, 3 frames,
    one loop-closure deformation (applyDeformation: every row moves) followed by the re-homing sweep (ssf_rehome_begin /
    _end) -- the owner invariant holds at once --, then 3 more frames: all against the unsharded oracle."""
    import p2p_worker
    from supersurfel_fusion_amd import synthetic
    world, tile = 3, 0.5
    cfg = dict(world=world, W=640, H=480, frames=6, pipelined=True, depth=2, batch=2, seed_n=1000000, tile=tile, tum=True)
    frames = p2p_worker.frames_of(cfg)
    fo = binding.Fusion(fast_oracle, p2p_worker.make_config(fast_oracle, dict(cfg, pipelined=False), 0, 1, cfg["seed_n"] + 65536))
    model, nvis, _ = p2p_worker.seed_shard(cfg, 0, 1)
    fo.set_model(model, nvis, 30)
    fs = []
    for r in range(world):
        m, nv, cap = p2p_worker.seed_shard(cfg, r, world)
        f = binding.Fusion(product_lib, p2p_worker.make_config(product_lib, cfg, r, world, cap))
        f.set_model(m, nv, 30)
        f.p2p_configure(all_ranks_on_this_device=True, timeout_s=20.0)
        fs.append(f)
    regions = [f.p2p_region()[0] for f in fs]
    for f in fs:
        f.p2p_attach_local(regions)

    def drive_all(part):
        out, errors = [None] * world, []

        def drive(r):
            try:
                out[r] = p2p_worker.run_frames(fs[r], part, True)
            except Exception as e:
                errors.append((r, e))
        threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(300)
        assert not errors, errors
        return out

    want = [fo.process_frame(r, d) for r, d in frames[:3]]
    got = drive_all(frames[:3])
    for f in [fo] + fs:
        f.apply_deformation(*util.deformation_for(f.get_model(), 200, angle=0.01, shift=0.02))
    strays = sum(int((synthetic.tile_owner(f.get_model()["positions"][f.get_model()["confidences"] > 0], world, tile) != r).sum()) for r, f in enumerate(fs))
    moved = util.rehome_in_process(fs)
    assert strays > 100 and sum(moved) == strays
    for r, f in enumerate(fs):
        m = f.get_model()
        assert (synthetic.tile_owner(m["positions"][m["confidences"] > 0], world, tile) == r).all(), "not at home right after the sweep"
    want += [fo.process_frame(r, d) for r, d in frames[3:]]
    got2 = drive_all(frames[3:])
    outs = [(np.stack([x["pose"] for x in got[r] + got2[r]]), np.array([[x[k] for k in KEYS] for x in got[r] + got2[r]], np.int64)) for r in range(world)]
    check_against_unsharded(world, tile, want, outs, fo.get_model(), [f.get_model() for f in fs])
    assert want[-1]["icp_valid"] == 1


def test_an_empty_rank_stays_in_step_through_deformation_and_rehoming(oracle_lib, product_lib):
    """Three ranks with 100 m tiles: the room's eight octant tiles hash to ranks 0 and 2, so rank 1's shard is EMPTY and stays
    empty.  Frames -> deformation -> re-homing sweep -> frames: rank 1 takes the early return of every one of those calls
    (nothing to deform, nothing leaves, nothing arrives) while its peers rewrite their shards; all three must still enter the
    next frame agreeing on whether the shard sizes are exchanged afresh -- a rank that kept the record of the last frame
    skipped an exchange its peers performed and the frame call timed out."""
    from supersurfel_fusion_amd import synthetic
    world, tile, W, H = 3, 100.0, 320, 240
    frames = [util.frame(k, W, H) for k in range(6)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=8192))
    fs = [binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=8192, rank=r, nranks=world, shard_tile=tile)) for r in range(world)]
    for f in fs:
        f.p2p_configure(all_ranks_on_this_device=True, timeout_s=20.0)
    regions = [f.p2p_region()[0] for f in fs]
    for f in fs:
        f.p2p_attach_local(regions)

    def drive_all(part):
        out, errors = [None] * world, []

        def drive(r):
            try:
                out[r] = [fs[r].process_frame(a, d) for a, d in part]
            except Exception as e:
                errors.append((r, e))
        threads = [threading.Thread(target=drive, args=(r,), daemon=True) for r in range(world)]
        for t in threads:
            t.start()
        for t in threads:
            t.join(120)
        assert not errors, errors
        return out

    want = [fo.process_frame(r, d) for r, d in frames[:3]]
    got = drive_all(frames[:3])
    assert fs[1].counts()["n_model"] == 0 and min(fs[0].counts()["n_model"], fs[2].counts()["n_model"]) > 0
    for f in [fo] + fs:
        f.apply_deformation(*util.deformation_for(f.get_model(), 16, angle=0.01, shift=0.02))
    util.rehome_in_process(fs)
    assert fs[1].counts()["n_model"] == 0
    want += [fo.process_frame(r, d) for r, d in frames[3:]]
    got2 = drive_all(frames[3:])
    outs = [(np.stack([x["pose"] for x in got[r] + got2[r]]), np.array([[x[k] for k in KEYS] for x in got[r] + got2[r]], np.int64)) for r in range(world)]
    models = [f.get_model() for f in fs]
    models[1] = {name: v[:0] for name, v in models[1].items()}          # (get_model of an empty shard hands back one placeholder row)
    check_against_unsharded(world, tile, want, outs, fo.get_model(), models)
    g = fs[1].global_counts()
    assert g["n_model"] == want[-1]["n_model"] and g["n_visible"] == want[-1]["n_visible"]


def test_rehoming_across_processes(fast_oracle, tmp_path):
    """the deformation + re-homing sweep with the ranks as separate processes (tables traded through files, as the IPC handles
    are), frames on the native exchange before and after it"""
    import p2p_worker
    cfg = dict(world=3, W=320, H=240, frames=6, pipelined=True, depth=1, batch=2, seed_n=60000, tile=0.25, deform_after=3, nodes=24, angle=0.02, shift=0.05)
    fo = binding.Fusion(fast_oracle, p2p_worker.make_config(fast_oracle, dict(cfg, pipelined=False), 0, 1, cfg["seed_n"] + 65536))
    model, nvis, _ = p2p_worker.seed_shard(cfg, 0, 1)
    fo.set_model(model, nvis, 30)
    frames = p2p_worker.frames_of(cfg)
    want = [fo.process_frame(r, d) for r, d in frames[:3]]
    fo.apply_deformation(*util.deformation_for(fo.get_model(), 24, angle=0.02, shift=0.05))
    want += [fo.process_frame(r, d) for r, d in frames[3:]]
    outs = run_workers(cfg, tmp_path, 600)
    homes = [np.load(os.path.join(str(tmp_path), "home%d.npy" % r)) for r in range(3)]
    assert all(int(h[0]) == 1 for h in homes) and sum(int(h[1]) for h in homes) > 0
    models = [{k[len("model_"):]: o[k] for k in o.files if k.startswith("model_")} for o in outs]
    check_against_unsharded(3, 0.25, want, [(o["poses"], o["counts"]) for o in outs], fo.get_model(), models)


def test_a_missing_peer_is_an_error_not_a_hang(product_lib):
    """rank 0 of a two-rank map whose peer never calls: the frame call fails after the bounded wait"""
    W, H = 160, 128
    fs = [binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=2048, rank=r, nranks=2, shard_tile=0.25)) for r in range(2)]
    for f in fs:
        f.p2p_configure(all_ranks_on_this_device=True, timeout_s=3.0)
    regions = [f.p2p_region()[0] for f in fs]
    fs[0].p2p_attach_local(regions)
    rgb, depth = util.frame(0, W, H)
    with pytest.raises(binding.SsfError):
        fs[0].process_frame(rgb, depth)
