"""BASELINE.json's configurations at their WORKLOAD size, HIP product against the CPU oracle, bit for bit
("configs not exercised by the driver-run suite"):

  config 2   640x480 orbit, map seeded with 200 k and with 1 M supersurfels, pipelined 2 x 8 as bench.py runs it
  config 3   1280x960, 1 M seeded supersurfels ALL in view, 10 forced ICP iterations per frame
  config 4   (single-GPU stand-in) a 2 M-supersurfel map sharded by world tile over 4 ranks -- four handles on the one
             GPU, the three exchanges done between the stage calls -- against the UNSHARDED oracle
  config 5   (single-GPU stand-in) TUM-shaped replay: u16 depth at 5000 / m with ~25 % holes through replay.py,
             depth pre-filter on, 1 M supersurfels, one loop-closure deformation (N / 50 nodes, 4 weights per
             supersurfel) in the middle of the sequence

The oracle here is the OpenMP build of the same sources (oracle/_build/libssf_oracle_omp.so, built on this box):
its results equal the single-threaded checker's bit for bit (tests/test_oracle.py) and it keeps these tests to
seconds.  What only real multi-GPU hardware can show (RCCL over xGMI at N = 4 / 8) is not claimed here."""
import os
import subprocess

import numpy as np
import pytest

import util
from conftest import ROOT
from supersurfel_fusion_amd import binding, replay, synthetic

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fast_oracle():
    r = subprocess.run(["make", "-C", os.path.join(ROOT, "oracle"), "omp"], stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
    assert r.returncode == 0, r.stdout
    return binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_omp.so"))


def contiguous(frames):
    return [(np.ascontiguousarray(r, np.uint8), np.ascontiguousarray(d, np.float32)) for r, d in frames]


def sweep(n, W, H, distinct=6, **kw):
    """n frames of the orbit, sweeping back and forth over `distinct` rendered ones (consecutive frames 1 degree apart)"""
    base = [util.frame(k, W, H, **kw) for k in range(distinct)]
    order = [(i % (2 * distinct - 2)) for i in range(n)]
    return contiguous([base[j if j < distinct else 2 * distinct - 2 - j] for j in order])


@pytest.mark.parametrize("n_seed", [200000, 1000000])
def test_config2_orbit_at_workload_size_pipelined(n_seed, fast_oracle, product_lib):
    """bench.py's own workload (640x480, seeded map, pipeline_depth 2 x extract_batch 8, ssf_process_sequence):
    10 frames = the batch ramp 2, 4 and a partial last batch; every frame result and the whole final map."""
    W, H, nf = 640, 480, 10
    model, nvis = synthetic.seed_model_cam0(n_seed, W, H, stamp=30)
    fo = binding.Fusion(fast_oracle, util.make_cfg(fast_oracle, W, H, nb_supersurfels_max=n_seed + 65536))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=n_seed + 65536, pipeline_depth=2, extract_batch=8))
    fo.set_model(model, nvis, 30); fh.set_model(model, nvis, 30)
    frames = sweep(nf, W, H)
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    assert want[-1]["icp_valid"] == 1 and want[-1]["n_visible"] > nvis // 2
    util.compare_state(fo, fh)


def test_config3_1280x960_one_million_visible_forced_iterations(fast_oracle, product_lib):
    """The HBM-bound stress: every seeded supersurfel in view (ICP / association / update over ~1 M rows per
    iteration), icp_force_iters = 1 -> 10 iterations per frame; 3 frames, pipelined 1 x 2."""
    W, H, nf, n_seed = 1280, 960, 3, 1000000
    model, nvis = synthetic.seed_model_cam0_visible(n_seed, W, H, stamp=30)
    assert nvis == n_seed
    kw = dict(nb_supersurfels_max=n_seed + 65536, icp_force_iters=1)
    fo = binding.Fusion(fast_oracle, util.make_cfg(fast_oracle, W, H, **kw))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=1, extract_batch=2, **kw))
    fo.set_model(model, nvis, 30); fh.set_model(model, nvis, 30)
    frames = sweep(nf, W, H, distinct=3)
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
        assert a["icp_iters"] == 10
    assert want[-1]["n_visible"] > 700000
    util.compare_state(fo, fh)


def rows_multiset(m):
    n = len(m["confidences"])
    cols = [np.ascontiguousarray(m[name]).reshape(n, -1).view(np.uint32) for name, _, _ in binding.SURFEL_FIELDS]
    rows = np.concatenate(cols, axis=1)
    return rows[np.lexsort(rows.T[::-1])]


def test_config4_two_million_rows_over_four_emulated_ranks(fast_oracle, product_lib):
    """2 M supersurfels sharded by world tile over 4 ranks (4 handles on this GPU; ICP record SUM, association MIN / MAX
    and the shard sizes exchanged on the host between the stage calls, exactly what the RCCL path does in HBM): every
    rank holds the pose of the UNSHARDED oracle bit for bit, the global counters agree, the union of the four shards is
    the oracle's map row for row, and after every frame each row lives on the rank that owns its world tile."""
    W, H, nf, n_seed, world, tile = 640, 480, 4, 2000000, 4, 0.5
    model, nvis = synthetic.seed_model_cam0(n_seed, W, H, stamp=30)
    fo = binding.Fusion(fast_oracle, util.make_cfg(fast_oracle, W, H, nb_supersurfels_max=n_seed + 65536))
    fo.set_model(model, nvis, 30)
    own = synthetic.tile_owner(model["positions"], world, tile)
    vis = np.arange(n_seed) < nvis
    ranks, counts = [], np.zeros((world, 2), np.int64)
    for r in range(world):
        sel = own == r
        f = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, nb_supersurfels_max=int(sel.sum()) + 65536, rank=r, nranks=world, shard_tile=tile))
        f.set_model({k: v[sel] for k, v in model.items()}, int((sel & vis).sum()), 30)
        ranks.append(f)
        counts[r] = [int(sel.sum()), int((sel & vis).sum())]
    frames = sweep(nf, W, H)
    for k, (rgb, depth) in enumerate(frames):
        want = fo.process_frame(rgb, depth)
        for f in ranks:
            f.stage_extract(rgb, depth)
        g_model, g_vis = int(counts[:, 0].sum()), int(counts[:, 1].sum())
        for r, f in enumerate(ranks):
            f.set_shard(int(counts[:r, 1].sum()), g_model, g_vis)
            f.icp_begin()
        again, iters = True, 0
        while again:
            total = sum(f.icp_accumulate() for f in ranks)
            agains = [f.icp_update(total) for f in ranks]
            assert len(set(agains)) == 1
            again = agains[0]; iters += 1
        valid = [f.icp_end() for f in ranks]
        bm = [f.match() for f in ranks]
        best, matched = np.minimum.reduce([b for b, _ in bm]), np.maximum.reduce([m for _, m in bm])
        res = util.exchange_and_fuse(ranks, best, matched)
        counts = np.array([[r_["n_model"], r_["n_visible"]] for r_ in res], np.int64)
        assert iters == want["icp_iters"] and all(v == bool(want["icp_valid"]) for v in valid)
        for r_ in res:
            util.assert_same_bits(r_["pose"], want["pose"], "pose of frame %d on a shard" % k)
        for key in ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated"):
            assert sum(r_[key] for r_ in res) == want[key], (k, key, [r_[key] for r_ in res], want[key])
    shards = [f.get_model() for f in ranks]
    for r, m in enumerate(shards):
        assert len(m["confidences"]) > 300000
        assert (synthetic.tile_owner(m["positions"], world, tile) == r).all(), "a row lives on a rank that does not own its tile"
    merged = {name: np.concatenate([m[name] for m in shards]) for name, _, _ in binding.SURFEL_FIELDS}
    single = fo.get_model()
    assert len(merged["confidences"]) == len(single["confidences"])
    assert np.array_equal(rows_multiset(merged), rows_multiset(single)), "union of the shards != the unsharded map"


def test_config5_tum_shaped_replay_with_deformation(fast_oracle, product_lib):
    """TUM-shaped input through the replay harness: 16-bit depth at 5000 counts / m (depth_scale 0.0002, as
    launch/supersurfel_fusion_rgbd_benchmark.launch:47), ~25 % of the pixels holes, the benchmark launch parameters with
    the depth pre-filter on, ~1 M supersurfels, pipelined replay on the product; after 4 frames one loop-closure
    deformation (applyDeformation, deformation_graph_kernels.cu:27-73: N / 50 nodes, 4 weights per supersurfel), then
    the replay continues on the deformed map."""
    W, H, n_seed = 640, 480, 1000000
    model, nvis = synthetic.seed_model_cam0(n_seed, W, H, stamp=30)
    cfg = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=n_seed + 65536)
    fo = binding.Fusion(fast_oracle, fast_oracle.default_config(**cfg))
    fh = binding.Fusion(product_lib, product_lib.default_config(pipeline_depth=2, extract_batch=4, **cfg))
    fo.set_model(model, nvis, 30); fh.set_model(model, nvis, 30)

    def tum_frames(ks):
        for k in ks:
            rgb, depth = util.frame(k, W, H, noise=True, holes=0.25)
            d16 = np.clip(np.rint(depth.astype(np.float64) * 5000.0), 0, 65535).astype(np.uint16)
            yield "%.6f" % (1305031102.0 + k / 30.0), rgb, replay.convert_depth(d16, 0.0002)

    lo, ro = replay.replay(fo, tum_frames(range(4)))
    lh, rh = replay.replay(fh, tum_frames(range(4)), pipelined=True)
    assert lo == lh
    for a, b in zip(ro, rh):
        util.same_result(a, b)
    assert ro[-1]["icp_valid"] == 1
    n = ro[-1]["n_model"]
    rng = np.random.default_rng(5)
    m = n // 50
    npos = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    ang = rng.uniform(-0.01, 0.01, (m, 3))
    nrot = np.stack([(synthetic.rot_y(a[1]) @ synthetic.rot_x(a[0])).reshape(9) for a in ang]).astype(np.float32)
    ntr = rng.uniform(-0.004, 0.004, (m, 3)).astype(np.float32)
    w = rng.dirichlet(np.ones(4), n).astype(np.float32); idx = rng.integers(0, m, (n, 4)).astype(np.int32)
    for f in (fo, fh):
        f.apply_deformation(npos, nrot, ntr, w, idx)
    util.compare_state(fo, fh, maps=False, frame_surfels=False)
    lo, ro = replay.replay(fo, tum_frames(range(4, 7)))
    lh, rh = replay.replay(fh, tum_frames(range(4, 7)), pipelined=True)
    assert lo == lh
    for a, b in zip(ro, rh):
        util.same_result(a, b)
    util.compare_state(fo, fh)
    assert (fo.inlier_map() > 0).mean() < 0.8                      # the holes are there
