"""GPU parity tests proper: the HIP product (through the C ABI) against the CPU oracle on the same
seeded inputs -- bit-exact for every integer/index result AND for every float (both sides execute
the same IEEE operation sequence; accumulations are exact integers), against the committed golden
vectors, and -- at BASELINE.json sizes, where the oracle would be slow -- through size-independent
properties (determinism, exact shard additivity of the ICP record, stability and counts of the
3-way partition)."""
import os

import numpy as np
import pytest

import util
from supersurfel_fusion_amd import binding, synthetic
from test_oracle import check_against_golden

pytestmark = pytest.mark.gpu


def pair(oracle_lib, product_lib, W, H, **kw):
    return (binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, **kw)),
            binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, **kw)))


def test_product_is_the_hip_library(product_lib):
    assert product_lib.backend == "hip-gfx950"
    assert product_lib.path.endswith("supersurfel_fusion_amd/csrc/libssf_hip.so")


def test_golden_vectors(product_lib):
    check_against_golden(product_lib)


@pytest.mark.parametrize("size", [(160, 128), (640, 480), (150, 100), (1280, 960)])
def test_sequence_bit_exact(size, oracle_lib, product_lib):
    W, H = size
    nf = 2 if W > 1000 else 4
    fo, fh = pair(oracle_lib, product_lib, W, H, nb_supersurfels_max=40000)
    for k in range(nf):
        rgb, depth = util.frame(k, W, H, noise=True, holes=0.03)
        ro, rh = fo.process_frame(rgb, depth), fh.process_frame(rgb, depth)
        util.same_result(ro, rh)
        util.compare_state(fo, fh)


@pytest.mark.parametrize("size,pass_list", [((320, 240), (1, 2, 3, 4, 7, 20, 21, 33, 40)), ((640, 480), (1, 20, 21, 40))])
def test_every_relabelling_pass_bit_exact(size, pass_list, oracle_lib, product_lib):
    """the state after k relabelling passes, k across both phases; at the metric's 640 x 480 too (round 6: first and last pass of each
    phase -- a window overflow or a far-label path that first shows at full size would otherwise only be seen in the end state)"""
    Wt, Ht = size
    rgb, depth = util.frame(0, Wt, Ht, holes=0.05)
    for passes in pass_list:
        fo, fh = pair(oracle_lib, product_lib, Wt, Ht)
        fo.set_max_passes(passes); fh.set_max_passes(passes)
        fo.stage_extract(rgb, depth); fh.stage_extract(rgb, depth)
        util.assert_same_bits(fo.index_map(), fh.index_map(), "labels after %d passes" % passes)
        util.assert_same_bits(fo.inlier_map(), fh.inlier_map(), "inliers after %d passes" % passes)
        util.assert_same_bits(fo.superpixels(), fh.superpixels(), "superpixels after %d passes" % passes)


@pytest.mark.parametrize("variant", ["no_ransac", "reference_defaults", "icp_forced"])
def test_parameter_variants(variant, oracle_lib, product_lib):
    kw = dict(nb_supersurfels_max=20000)
    if variant == "no_ransac":
        kw.update(seg_use_ransac=0)
    if variant == "reference_defaults":
        kw.update(lambda_pos=50.0, lambda_size=10000.0, lambda_disp=1e6, filter_iter=4, conf_thresh=2500.0, icp_cov_thresh=0.04)
    if variant == "icp_forced":
        kw.update(icp_force_iters=1)
    fo, fh = pair(oracle_lib, product_lib, 320, 240, **kw)
    for k in range(3):
        rgb, depth = util.frame(k, 320, 240)
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
    util.compare_state(fo, fh)


def test_edge_cases(oracle_lib, product_lib):
    # holes everywhere, then a normal frame, with a dynamic mask and a pose prior
    fo, fh = pair(oracle_lib, product_lib, 160, 128, nb_supersurfels_max=4096)
    rgb, depth = util.frame(0, 160, 128)
    for f in (fo, fh):
        f.process_frame(rgb, np.zeros_like(depth))
    util.compare_state(fo, fh, frame_surfels=False)
    mask = np.zeros(fo.S, np.uint8); mask[5:25] = 1
    prior = synthetic.pose12(*synthetic.relative_pose(1))
    rgb, depth = util.frame(1, 160, 128, holes=0.3)
    util.same_result(fo.process_frame(rgb, depth, prior_pose=prior, dynamic_mask=mask),
                     fh.process_frame(rgb, depth, prior_pose=prior, dynamic_mask=mask))
    util.compare_state(fo, fh)
    # capacity overflow: insertion stops at nb_supersurfels_max, highest frame ids dropped
    fo, fh = pair(oracle_lib, product_lib, 160, 128, nb_supersurfels_max=90)
    for k, prior_k in ((0, None), (25, 25)):
        rgb, depth = util.frame(k, 160, 128)
        p = None if prior_k is None else synthetic.pose12(*synthetic.relative_pose(prior_k))
        util.same_result(fo.process_frame(rgb, depth, prior_pose=p), fh.process_frame(rgb, depth, prior_pose=p))
    util.compare_state(fo, fh)


def seeded(lib, n, W, H, **kw):
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=n + 8192, **kw))
    model, nvis = synthetic.seed_model_cam0(n, W, H, stamp=30)
    f.set_model(model, nvis, 30)
    return f, nvis


def test_seeded_model_50k_bit_exact(oracle_lib, product_lib):
    """ICP over ~tens of thousands of visible supersurfels, association, update, classify, reorder."""
    fo, nv = seeded(oracle_lib, 50000, 640, 480)
    fh, _ = seeded(product_lib, 50000, 640, 480)
    assert nv > 3000
    for k in range(3):
        rgb, depth = util.frame(k, 640, 480)
        ro, rh = fo.process_frame(rgb, depth), fh.process_frame(rgb, depth)
        util.same_result(ro, rh)
        assert ro["icp_valid"] == 1
    util.compare_state(fo, fh)


def test_model_device_view_has_the_reference_layout(product_lib):
    """ssf_get_model_device on the product: dense [visible | out-of-view] rows, orientations as packed Mat33"""
    fh, _ = seeded(product_lib, 30000, 640, 480)
    for k in range(2):
        fh.process_frame(*util.frame(k, 640, 480))
    st, n = fh.model_device()
    m = fh.get_model()
    assert n == len(m["confidences"]) > 20000
    for name, k, dt in binding.SURFEL_FIELDS:
        got = util.device_to_host(getattr(st, name), (n, k) if k > 1 else (n,), dt)
        util.assert_same_bits(got, m[name], "device view " + name)


def test_deformation_bit_exact(oracle_lib, product_lib):
    fo, _ = seeded(oracle_lib, 20000, 640, 480)
    fh, _ = seeded(product_lib, 20000, 640, 480)
    rng = np.random.default_rng(5)
    m, n = 400, 20000
    npos = rng.uniform(-3, 3, (m, 3)).astype(np.float32)
    ang = rng.uniform(-0.05, 0.05, (m, 3))
    nrot = np.stack([(synthetic.rot_y(a[1]) @ synthetic.rot_x(a[0])).reshape(9) for a in ang]).astype(np.float32)
    ntr = rng.uniform(-0.02, 0.02, (m, 3)).astype(np.float32)
    w = rng.dirichlet(np.ones(4), n).astype(np.float32); idx = rng.integers(0, m, (n, 4)).astype(np.int32)
    for f in (fo, fh):
        f.apply_deformation(npos, nrot, ntr, w, idx)
    util.compare_state(fo, fh, maps=False, frame_surfels=False)


# ---- BASELINE-size properties (no oracle in the loop) ------------------------------------------------
@pytest.fixture(scope="module")
def big(product_lib):
    return seeded(product_lib, 1000000, 640, 480, icp_force_iters=1)


def test_full_size_shard_additivity_of_icp_record(big, product_lib):
    """The ICP record over the whole visible set equals the exact integer sum of the records of
    two disjoint halves (what the multi-GPU SUM all-reduce relies on)."""
    f, nvis = big
    assert nvis > 50000
    rgb, depth = util.frame(0, 640, 480)
    f.stage_extract(rgb, depth); f.icp_begin()
    whole = f.icp_accumulate()
    assert whole[28] > 10000
    m = f.get_model(0, nvis)
    parts = []
    for sl in (slice(0, nvis // 3), slice(nvis // 3, nvis)):
        sub = {k: v[sl] for k, v in m.items()}
        g = binding.Fusion(product_lib, util.make_cfg(product_lib, 640, 480, nb_supersurfels_max=nvis + 8192, icp_force_iters=1))
        g.set_model(sub, len(sub["confidences"]), 30)
        g.stage_extract(rgb, depth); g.icp_begin()
        parts.append(g.icp_accumulate())
    assert np.array_equal(parts[0] + parts[1], whole)
    f.icp_update(whole); f.icp_end(); b, mt = f.match(); f.fuse(b, mt)


def test_full_size_determinism_and_partition_properties(product_lib):
    runs = []
    for rep in range(2):
        f, nvis = seeded(product_lib, 1000000, 640, 480)
        m0 = f.get_model()
        uid = np.arange(len(m0["confidences"]), dtype=np.int32) + 1000   # > any stamp a new row can carry
        m0["stamps"][:, 0] = uid                      # t_init carries a unique id through the reorder
        f.set_model(m0, nvis, 30)
        for k in range(2):
            rgb, depth = util.frame(k, 640, 480)
            r = f.process_frame(rgb, depth)
        runs.append((r, f.get_model(), f.get_pose()))
    (ra, ma, pa), (rb, mb, pb) = runs
    util.same_result(ra, rb)
    for name in ma:
        util.assert_same_bits(ma[name], mb[name], "run-to-run " + name)
    # counts are consistent and the model is [visible | out of view], removed rows dropped
    assert ra["n_visible"] <= ra["n_model"] and ra["n_model"] > 900000
    # stability: rows inserted during the two frames carry t_init = 30/31 (< 1000); the seeded rows carry
    # their original position.  A stable partition keeps every block a concatenation of at most 2^frames
    # ascending runs of the original order, and never duplicates or invents a row.
    ids = ma["stamps"][:ra["n_model"], 0].astype(np.int64)
    old = ids[ids >= 1000]
    assert len(np.unique(old)) == len(old) and old.max() < 1000 + 1000000
    for blk in (ids[:ra["n_visible"]], ids[ra["n_visible"]:]):
        o = blk[blk >= 1000]
        assert int((np.diff(o) < 0).sum()) <= 3
    new = ids[ids < 1000]
    assert set(np.unique(new)) <= {30, 31}


@pytest.mark.parametrize("depth_ahead,batch", [(1, 1), (2, 1), (3, 1), (0, 3), (1, 2), (2, 4), (1, 8), (1, 12), (0, 16)])
def test_pipelined_equals_sequential_bit_exact(depth_ahead, batch, oracle_lib, product_lib):
    """ssf_submit_frame / ssf_process_submitted with extract running ahead on its own streams and
    `batch` frames per launch chain: every per-frame result and the final map equal the oracle's
    sequential run (the RANSAC draw counters are the only cross-frame state of extract; they are
    chained inside a batch by the kernel and between batches by events).  11 frames: the last
    batch is a partial one."""
    W, H, nf = 320, 240, 11
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=depth_ahead, extract_batch=batch))
    assert fh.pipeline_capacity() == (depth_ahead + 1) * batch
    frames = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(nf)]
    mask = np.zeros(fo.S, np.uint8); mask[7:19] = 1
    masks = [mask if k in (2, 5) else None for k in range(nf)]
    want = [fo.process_frame(fr[0], fr[1], dynamic_mask=mk) for fr, mk in zip(frames, masks)]
    got = []
    nsub = 0
    for k in range(nf):
        while nsub < nf and fh.can_submit():
            fh.submit_frame(frames[nsub][0], frames[nsub][1], dynamic_mask=masks[nsub]); nsub += 1
        got.append(fh.process_submitted().as_dict())
    assert fh.pending_frames() == 0
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)            # maps / frame supersurfels of the last frame + pose + whole model


def test_pipeline_full_and_empty_are_errors(product_lib):
    W, H = 160, 128
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=1))
    rgb, depth = util.frame(0, W, H)
    with pytest.raises(binding.SsfError):
        fh.process_submitted()
    fh.submit_frame(rgb, depth); fh.submit_frame(rgb, depth)
    with pytest.raises(binding.SsfError):
        fh.submit_frame(rgb, depth)
    with pytest.raises(binding.SsfError):      # the whole-frame call may not jump the queue
        fh.process_frame(rgb, depth)
    fh.process_submitted(); fh.process_submitted()
    fh.process_frame(rgb, depth)


def test_long_sweep_with_view_changes_and_store_upkeep(oracle_lib, product_lib):
    """A back-and-forth sweep with 4-degree steps: rows leave and re-enter the view every frame, rows are culled
    and inserted.  The product keeps the out-of-view rows in a deque-like store with holes:
    the logical model must stay equal to the oracle's stable partition, with forced compactions in between and
    with a capacity so small that the store has to recentre by itself."""
    W, H = 160, 128
    kw = dict(nb_supersurfels_max=600, delta_t=3, conf_thresh=1e9)       # aggressive culling, tiny capacity
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, **kw))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, **kw))
    ks = list(range(0, 40, 4)) + list(range(40, -1, -4)) + list(range(0, 24, 4))
    for n, k in enumerate(ks):
        rgb, depth = util.frame(k, W, H, noise=True, holes=0.02)
        prior = synthetic.pose12(*synthetic.relative_pose(k))             # large steps: give ICP the pose prior
        util.same_result(fo.process_frame(rgb, depth, prior_pose=prior), fh.process_frame(rgb, depth, prior_pose=prior))
        if n % 5 == 2:
            fh.debug_recentre()
        if n % 3 == 0 or n == len(ks) - 1:
            util.compare_state(fo, fh, maps=False, frame_surfels=False)
    c = fh.counts()
    assert c["n_model"] > 0 and c["n_visible"] > 0


def test_out_of_view_store_recentres_by_itself(oracle_lib, product_lib):
    """Same sweep without forced compactions: with a capacity this small the span of the out-of-view store runs out
    of room in front and has to be compacted automatically, several times; results stay equal to the oracle."""
    W, H = 160, 128
    kw = dict(nb_supersurfels_max=400, delta_t=1000, conf_thresh=0.0)     # nothing is culled by age: rows pile up
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, **kw))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, **kw))
    ks = (list(range(0, 48, 6)) + list(range(48, -1, -6))) * 3
    for n, k in enumerate(ks):
        rgb, depth = util.frame(k, W, H, noise=True)
        prior = synthetic.pose12(*synthetic.relative_pose(k))
        util.same_result(fo.process_frame(rgb, depth, prior_pose=prior), fh.process_frame(rgb, depth, prior_pose=prior))
    util.compare_state(fo, fh, maps=False, frame_surfels=False)
    assert fh.debug_recentre_count() >= 2, fh.debug_recentre_count()


def test_process_sequence_equals_frame_by_frame(oracle_lib, product_lib):
    W, H, nf = 320, 240, 9
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=2, extract_batch=4))
    frames = [util.frame(k, W, H, noise=True) for k in range(nf)]
    want = [fo.process_frame(*fr) for fr in frames]
    keep = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in keep], [d.ctypes.data for _, d in keep], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)


@pytest.mark.parametrize("kw", [
    dict(cell_size=8), dict(cell_size=32), dict(cell_size=12), dict(cell_size=20),
    dict(nb_samples=8), dict(nb_samples=32), dict(nb_samples=5),
    dict(seg_iter=3), dict(seg_iter=1), dict(seg_iter=2, filter_iter=0), dict(filter_iter=7),
    dict(cell_size=8, nb_samples=4, seg_iter=5),
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_segmentation_parameter_space(kw, oracle_lib, product_lib):
    """Cell sizes that change the LDS cell window (8: many cells per tile, 32: one, 12 / 20: not powers of two, image
    not a multiple of the cell), sample counts beyond the LDS table (32) and odd ones, odd / minimal iteration
    counts: the fallbacks (global slow paths, plane filter in global memory) must equal the oracle too."""
    W, H = 320, 240
    cfg = dict(nb_supersurfels_max=40000); cfg.update(kw)
    fo, fh = pair(oracle_lib, product_lib, W, H, **cfg)
    for k in range(3):
        rgb, depth = util.frame(k, W, H, noise=True, holes=0.03)
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)


@pytest.mark.parametrize("size,cell,filter_iter", [((640, 480), 12, 3), ((640, 480), 12, 7), ((1280, 960), 16, 3), ((1280, 960), 16, 0), ((640, 480), 8, 3), ((960, 720), 8, 4),
                                                   ((320, 240), 16, 1), ((330, 250), 16, 5), ((640, 480), 16, 12), ((640, 480), 12, 12), ((1280, 960), 16, 12), ((960, 720), 8, 12)],
                         ids=lambda v: "x".join(str(t) for t in v) if isinstance(v, tuple) else str(v))
def test_plane_filter_at_every_grid_size(size, cell, filter_iter, oracle_lib, product_lib):
    """TPS_RGBD::filter by the size of the superpixel grid and the number of sweeps.  Up to filter_iter 9 every grid larger than one
    16 x 12 tile takes the round-5 form (k_plane_filter_tiled): many workgroups per frame, each computing its core tile and
    everything within filter_iter of it -- including, for windows that touch the last column, the first columns one row down
    (the reference's `x<gridSizeX` typo makes those neighbours).  More sweeps than a 1024-thread window holds fall back to one
    workgroup per frame: all eleven floats per node in LDS up to 1396 nodes (k_plane_filter<true>), states + centroids in up to
    160 KB of LDS with the data term in registers up to 5120 nodes (k_plane_filter_regs<3> / <5>: 2160 and 4800 nodes here), the
    global scratch beyond (10 800 nodes).  Planes and everything rendered from them must be the oracle's, bit for bit."""
    W, H = size
    fo, fh = pair(oracle_lib, product_lib, W, H, nb_supersurfels_max=40000, cell_size=cell, filter_iter=filter_iter)
    rgb, depth = util.frame(1, W, H, noise=True, holes=0.03)
    fo.stage_extract(rgb, depth); fh.stage_extract(rgb, depth)
    util.assert_same_bits(fo.superpixels(), fh.superpixels(), "superpixel planes after the filter")
    util.assert_same_bits(fo.plane_depth(), fh.plane_depth(), "plane-rendered depth")
    util.assert_same_bits(fo.index_map(), fh.index_map(), "labels")
    a, b = fo.get_frame(), fh.get_frame()
    for name in a:
        util.assert_same_bits(a[name], b[name], "frame." + name)


def test_segmentation_parameter_space_batched(oracle_lib, product_lib):
    """The same fallbacks through the batched / pipelined extract (cell 8 at 150x100: partial cells and tiles)."""
    W, H, nf = 150, 100, 7
    kw = dict(cell_size=8, nb_samples=32, seg_iter=3, nb_supersurfels_max=20000)
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, **kw))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=1, extract_batch=3, **kw))
    frames = [util.frame(k, W, H, noise=True, holes=0.05) for k in range(nf)]
    want = [fo.process_frame(*fr) for fr in frames]
    keep = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in keep], [d.ctypes.data for _, d in keep], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)


@pytest.mark.parametrize("kw", [
    dict(lambda_pos=0.01, lambda_size=0.0, lambda_bound=0.0),          # superpixels free to drift far from their cells
    dict(lambda_pos=0.01, lambda_size=0.0, lambda_bound=0.0, cell_size=8),
    dict(lambda_disp=0.0), dict(thresh_disp=1e-7), dict(thresh_disp=1.0),
    dict(filter_alpha=10.0, filter_beta=0.01, filter_threshold=1.0),
    dict(range_min=1.5, range_max=2.5),
], ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_energy_parameter_space(kw, oracle_lib, product_lib):
    """Energy weights that let superpixels wander out of the LDS cell window of their tile (exact global slow
    paths), disparity thresholds at both extremes, a strong plane filter and a narrow depth range."""
    W, H = 320, 240
    cfg = dict(nb_supersurfels_max=40000); cfg.update(kw)
    fo, fh = pair(oracle_lib, product_lib, W, H, **cfg)
    for k in range(3):
        rgb, depth = util.frame(k, W, H, noise=True, holes=0.03)
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)


@pytest.mark.parametrize("kw", [dict(icp_iter=1), dict(icp_iter=3), dict(icp_iter=0), dict(icp_cov_thresh=1e-9),
                                dict(icp_force_iters=1, icp_iter=6), dict(delta_t=1, conf_thresh=1e9)],
                         ids=lambda kw: ",".join("%s=%s" % kv for kv in kw.items()))
def test_tracking_parameter_space(kw, oracle_lib, product_lib):
    """Iteration caps (0 = no ICP at all), an unreachable covariance threshold (ICP always rejected), forced
    iterations, immediate culling: 640x480 so that ICP has enough pairs to be valid where it can be."""
    W, H = 640, 480
    cfg = dict(nb_supersurfels_max=60000); cfg.update(kw)
    fo, fh = pair(oracle_lib, product_lib, W, H, **cfg)
    for k in range(4):
        rgb, depth = util.frame(k, W, H, noise=True)
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
    util.compare_state(fo, fh)


@pytest.mark.parametrize("size", [(48, 32), (33, 17), (16, 16), (641, 479), (96, 250)])
def test_odd_image_sizes(size, oracle_lib, product_lib):
    """Images smaller than a relabelling tile, not multiples of the cell or of the tile, taller than wide."""
    W, H = size
    fo, fh = pair(oracle_lib, product_lib, W, H, nb_supersurfels_max=30000)
    for k in range(3):
        rgb, depth = util.frame(k, W, H, noise=True, holes=0.05)
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)


def test_hostile_depth_values(oracle_lib, product_lib):
    """NaN, +-inf, negative, denormal and huge depths, saturated and zero colours: nothing may diverge from the oracle."""
    W, H = 160, 128
    fo, fh = pair(oracle_lib, product_lib, W, H, nb_supersurfels_max=8192)
    rng = np.random.default_rng(7)
    for k in range(3):
        rgb, depth = util.frame(k, W, H, noise=True)
        depth = depth.copy(); rgb = rgb.copy()
        m = rng.random(depth.shape)
        depth[m < 0.03] = np.nan; depth[(m >= 0.03) & (m < 0.06)] = np.inf; depth[(m >= 0.06) & (m < 0.08)] = -np.inf
        depth[(m >= 0.08) & (m < 0.11)] = -1.5; depth[(m >= 0.11) & (m < 0.13)] = 1e-42; depth[(m >= 0.13) & (m < 0.15)] = 3e38
        rgb[(m > 0.9)] = 255; rgb[(m > 0.8) & (m <= 0.9)] = 0
        util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)


def test_soak_600_frames_pipelined(oracle_lib, product_lib):
    """600 frames of a back-and-forth sweep through the pipelined / batched path (3 extract contexts, 4 frames per
    launch chain, partial batches at the end): every per-frame result and the final map equal the oracle.  Races
    between the extract streams and the track chain, or in the model-store upkeep, would show up here."""
    W, H, nf = 160, 128, 600
    kw = dict(nb_supersurfels_max=3000, delta_t=15)
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, **kw))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=2, extract_batch=4, **kw))
    base = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(0, 41, 2)]
    seq = (base + base[-2:0:-1]) * 15
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in seq[:nf]]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    assert len(got) == nf
    for i, (a, b) in enumerate(zip(want, got)):
        for key in util.RESULT_KEYS:
            assert a[key] == b[key], (i, key, a[key], b[key])
        util.assert_same_bits(a["pose"], b["pose"], "pose of frame %d" % i)
    util.compare_state(fo, fh)
    assert want[-1]["n_model"] > 100


def _emulated_ranks(lib, world, W, H, nframes, cap=4096, extract="replicated"):
    """`world` handles of one library acting as the ranks of a sharded map in ONE process: the exchanges (ICP record
    SUM, association MIN / MAX, migrant table SUM, shard sizes) are done on the host between the stage calls.
    extract = "dealt": rank k % world alone extracts frame k; the others get its label map + plane depth + frame supersurfels
    (sharded.pack_frame_tables -> ssf_submit_frame_tables) and never see the images."""
    from supersurfel_fusion_amd import sharded
    fs = [binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=cap, rank=r, nranks=world, shard_tile=0.25))
          for r in range(world)]
    counts = np.zeros((world, 2), np.int64)
    out = []
    for k in range(nframes):
        rgb, depth = util.frame(k, W, H)
        if extract == "dealt":
            owner = k % world
            fs[owner].stage_extract(rgb, depth)
            words = sharded.pack_frame_tables(fs[owner])
            for r, f in enumerate(fs):
                if r != owner:
                    f.submit_frame_tables(*sharded.unpack_frame_tables(f, words))
                    f.begin_submitted()
        else:
            for f in fs:
                f.stage_extract(rgb, depth)
        g_model, g_vis = int(counts[:, 0].sum()), int(counts[:, 1].sum())
        for r, f in enumerate(fs):
            f.set_shard(int(counts[:r, 1].sum()), g_model, g_vis)
            f.icp_begin()
        again = g_vis > 0 and fs[0].cfg.icp_iter > 0
        while again:
            total = sum(f.icp_accumulate() for f in fs)
            agains = [f.icp_update(total) for f in fs]
            assert len(set(agains)) == 1
            again = agains[0]
        valid = [f.icp_end() for f in fs]
        assert len(set(valid)) == 1
        bm = [f.match() for f in fs]
        best = np.minimum.reduce([b for b, _ in bm])
        matched = np.maximum.reduce([m for _, m in bm])
        res = util.exchange_and_fuse(fs, best, matched)       # rows that crossed a tile edge move to their new owner
        counts = np.array([[r["n_model"], r["n_visible"]] for r in res], np.int64)
        for r, f in enumerate(fs):                            # every row lives on the rank that owns its world tile
            m = f.get_model()
            ok = m["confidences"] > 0                         # (the first frame copies invalid frame rows too: they are culled next frame)
            assert (synthetic.tile_owner(m["positions"][ok], world, 0.25) == r).all(), (k, r)
        out.append(([r["pose"].copy() for r in res], counts.copy(), [[r[key] for key in ("n_removed", "n_inserted", "n_updated")] for r in res]))
    return fs, out


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_emulated_ranks_bit_exact(world, oracle_lib, product_lib):
    """nranks > 1 on the HIP engine (global ids, shard offsets, winners that live on another rank) against the
    oracle doing the same exchanges, rank by rank."""
    W, H, NF = 320, 240, 6
    fh, oh = _emulated_ranks(product_lib, world, W, H, NF)
    fo, oo = _emulated_ranks(oracle_lib, world, W, H, NF)
    for k in range(NF):
        for r in range(world):
            assert np.array_equal(oh[k][0][r].view(np.uint32), oo[k][0][r].view(np.uint32)), ("pose", k, r)
        assert np.array_equal(oh[k][1], oo[k][1]), ("counts", k, oh[k][1], oo[k][1])
        assert oh[k][2] == oo[k][2], ("frame counters", k)
    assert all(len(f.get_model()["confidences"]) > 0 for f in fh)
    for r in range(world):
        mh, mo = fh[r].get_model(), fo[r].get_model()
        for name in mh:
            assert np.array_equal(mh[name].view(np.uint32), mo[name].view(np.uint32)), (name, r)


@pytest.mark.gpu
@pytest.mark.parametrize("world", [2, 3])
def test_dealt_extract_emulated_ranks_bit_exact(world, oracle_lib, product_lib):
    """The extract stage dealt over the ranks (round 5): rank k % world extracts frame k, the others rebuild
    their private tables from its label map + plane depth + frame supersurfels (k_import_frame).  Every rank's poses, counters
    and shard equal the REPLICATED run's on the oracle -- which the gloo tests tie to the unsharded oracle -- bit for bit."""
    W, H, NF = 320, 240, 6
    fh, oh = _emulated_ranks(product_lib, world, W, H, NF, extract="dealt")
    fo, oo = _emulated_ranks(oracle_lib, world, W, H, NF)
    for k in range(NF):
        for r in range(world):
            assert np.array_equal(oh[k][0][r].view(np.uint32), oo[k][0][r].view(np.uint32)), ("pose", k, r)
        assert np.array_equal(oh[k][1], oo[k][1]), ("counts", k, oh[k][1], oo[k][1])
        assert oh[k][2] == oo[k][2], ("frame counters", k)
    for r in range(world):
        mh, mo = fh[r].get_model(), fo[r].get_model()
        for name in mh:
            assert np.array_equal(mh[name].view(np.uint32), mo[name].view(np.uint32)), (name, r)


@pytest.mark.gpu
@pytest.mark.parametrize("direction", ["oracle_extracts", "product_extracts", "product_pipelined"])
def test_a_frame_extracted_elsewhere_on_the_hip_library(direction, oracle_lib, product_lib):
    """ssf_submit_frame_tables across the two libraries: the product tracks and fuses from tables the ORACLE extracted, and the
    oracle from tables the PRODUCT extracted (the tables are the reference's own quantities, not a private format); with a
    pipelined product handle a foreign frame takes a batch context of its own between local batches."""
    import test_sharded
    if direction == "oracle_extracts":
        fb, fr = test_sharded.frames_through_tables(oracle_lib, product_lib, 320, 240)
    elif direction == "product_extracts":
        fb, fr = test_sharded.frames_through_tables(product_lib, oracle_lib, 320, 240)
    else:
        fb, fr = test_sharded.frames_through_tables(product_lib, product_lib, 320, 240, pipeline_depth=2, extract_batch=2)
    util.compare_state(fb, fr, maps=False)


@pytest.mark.gpu
@pytest.mark.parametrize("depth_ahead,batch", [(2, 2), (1, 4)])
def test_pipelined_with_priors_blank_frames_and_pose_changes(depth_ahead, batch, oracle_lib, product_lib):
    """The first ICP iteration of a frame may have been accumulated ahead by the previous frame's row moves; that
    record must be dropped whenever it is not this frame's: a pose prior on the call, a pose set in between, a frame
    in which nothing is visible (blank depth: no rows, no record), frames after it (model still there, all out of
    view or re-entering)."""
    W, H, nf = 320, 240, 14
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=depth_ahead, extract_batch=batch))
    frames = [list(util.frame(k, W, H, noise=True, holes=0.02)) for k in range(nf)]
    frames[6][1] = np.zeros_like(frames[6][1])                      # blank frame: no frame supersurfels at all
    frames[10] = list(util.frame(40, W, H))                         # a jump: most of the map leaves the view
    priors = {3: synthetic.pose12(*synthetic.relative_pose(3)), 11: synthetic.pose12(*synthetic.relative_pose(40))}
    set_pose_before = {8: synthetic.pose12(*synthetic.relative_pose(8))}
    nsub = 0
    for k in range(nf):
        while nsub < nf and fh.can_submit():
            fh.submit_frame(frames[nsub][0], frames[nsub][1]); nsub += 1
        if k in set_pose_before:
            fo.set_pose(set_pose_before[k]); fh.set_pose(set_pose_before[k])
        want = fo.process_frame(frames[k][0], frames[k][1], prior_pose=priors.get(k))
        got = fh.process_submitted(prior_pose=priors.get(k)).as_dict()
        util.same_result(want, got)
    util.compare_state(fo, fh)


@pytest.mark.gpu
def test_host_frame_sequences_on_fresh_handles(oracle_lib, product_lib):
    """ssf_process_sequence with HOST frames uploads them from a worker thread while the calling thread launches --
    and, on a fresh handle, captures -- the extract graphs: every (depth, batch) on a new handle, partial last batches,
    a second sequence on the same handle."""
    W, H, nf = 160, 128, 9
    frames = [util.frame(k, W, H, noise=True) for k in range(2 * nf)]
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    want = [fo.process_frame(*fr) for fr in frames]
    for depth_ahead, batch in [(1, 1), (1, 2), (2, 2), (2, 3), (1, 4), (2, 4), (3, 2), (1, 5), (2, 8)]:
        fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=depth_ahead, extract_batch=batch))
        keep = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
        got = fh.process_sequence([r.ctypes.data for r, _ in keep[:nf]], [d.ctypes.data for _, d in keep[:nf]], on_device=False)
        got += fh.process_sequence([r.ctypes.data for r, _ in keep[nf:]], [d.ctypes.data for _, d in keep[nf:]], on_device=False)
        for a, b in zip(want, got):
            util.same_result(a, b)
        util.compare_state(fo, fh, maps=False, frame_surfels=False)
        fh.close()


@pytest.mark.gpu
def test_tracking_survives_a_starved_host(oracle_lib, product_lib):
    """The track chain talks to the host through a polled mailbox and, for chained ICP launches, the device waits for the
    host's next transform (bounded spin on a word the host stores through the BAR).  Here the host thread is starved:
    the process is confined to ONE cpu shared with busy-looping children, so every round trip can be delayed by whole
    scheduler quanta.  Results must still equal the oracle's, frame by frame -- late words, never wrong ones."""
    import multiprocessing as mp
    import os
    W, H, nf = 160, 128, 10
    frames = [util.frame(k, W, H, noise=True) for k in range(nf)]
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    want = [fo.process_frame(*fr) for fr in frames]
    old = os.sched_getaffinity(0)
    cpu = sorted(old)[0]

    def burn(stop):
        os.sched_setaffinity(0, {cpu})
        while not stop.is_set():
            pass

    ctx = mp.get_context("fork")
    stop = ctx.Event()
    hogs = [ctx.Process(target=burn, args=(stop,), daemon=True) for _ in range(3)]
    try:
        os.sched_setaffinity(0, {cpu})
        for p in hogs:
            p.start()
        fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=2, extract_batch=2))
        keep = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
        got = fh.process_sequence([r.ctypes.data for r, _ in keep], [d.ctypes.data for _, d in keep], on_device=False)
        fl = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H))            # one frame in flight: chained ICP launches
        got_l = [fl.process_frame(*fr) for fr in frames]
    finally:
        stop.set()
        os.sched_setaffinity(0, old)
        for p in hogs:
            p.join(timeout=5)
    for a, b, c in zip(want, got, got_l):
        util.same_result(a, b)
        util.same_result(a, c)
    util.compare_state(fo, fh, maps=False, frame_surfels=False)
    util.compare_state(fo, fl, maps=False, frame_surfels=False)


def test_an_arrival_at_a_full_shard_is_counted_as_removed(oracle_lib, product_lib):
    """k_migrate_in turns a row away when its shard is at capacity: the row is counted in n_removed there (the source shard
    has already let it go), same as the oracle; everything else of the frame is untouched"""
    fh, r0h, r1h = util.arrival_at_a_full_shard(product_lib)
    fo, r0o, r1o = util.arrival_at_a_full_shard(oracle_lib)
    util.same_result(r0h, r0o); util.same_result(r1h, r1o)
    util.compare_state(fh, fo)


@pytest.mark.parametrize("pipelined", [False, True])
def test_tile_sorted_rows_bit_exact(pipelined, oracle_lib, product_lib):
    """ICP and association streaming the TILE-SORTED copy of the visible rows (csrc/ssf_tile_rows.inc, launch_icp(by_tile) /
    launch_match(orig); in the product since round 6 for visible sets of 400 k rows and more -- forced for every frame here, with the
    copy made on its own stream beside the first two iterations): exact integer sums and atomicMin keys that carry the row's own
    index make every result independent of the order of the rows."""
    fo, nv = seeded(oracle_lib, 50000, 640, 480)
    kw = dict(pipeline_depth=2, extract_batch=2) if pipelined else {}
    fh, _ = seeded(product_lib, 50000, 640, 480, **kw)
    fh.set_bin_min_rows(0)
    frames = [util.frame(k, 640, 480, noise=True, holes=0.02) for k in range(5)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False) if pipelined \
        else [fh.process_frame(r, d) for r, d in frames]
    for a, b in zip(want, got):
        util.same_result(a, b)
    assert want[-1]["icp_valid"] == 1 and want[-1]["n_updated"] > 100
    util.compare_state(fo, fh)
    # an empty map and a first frame take the same path
    fo2, fh2 = pair(oracle_lib, product_lib, 320, 240, nb_supersurfels_max=8192)
    fh2.set_bin_min_rows(0)
    for k in range(4):
        util.same_result(fo2.process_frame(*util.frame(k, 320, 240)), fh2.process_frame(*util.frame(k, 320, 240)))
    util.compare_state(fo2, fh2)


def test_association_inside_the_waiting_icp_launch(oracle_lib, product_lib):
    """When the ICP loop ends by convergence, the launch that was made ahead for the next iteration is told the frame's final
    pose and does the association on its way out (SSF_ICP_GO_MATCH, k_icp) instead of being dismissed in front of a k_match
    launch.  Same rows, same tables, same arithmetic: every result is the oracle's, and the path is really taken."""
    import ctypes as C
    fo, nv = seeded(oracle_lib, 50000, 640, 480)
    fh, _ = seeded(product_lib, 50000, 640, 480, pipeline_depth=2, extract_batch=2)
    frames = [util.frame(k, 640, 480, noise=True, holes=0.02) for k in range(8)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)
    product_lib.lib.ssf_waiter_matches.restype = C.c_longlong
    product_lib.lib.ssf_waiter_matches.argtypes = [C.c_void_p]
    n = product_lib.lib.ssf_waiter_matches(fh.h)
    iters = [r["icp_iters"] for r in got]
    assert n >= 1, ("no frame's association ran in a waiting launch", iters)
    assert n == sum(1 for r in got if r["icp_iters"] > 0), (n, iters)


def test_a_loop_that_ends_at_the_iteration_cap_associates_in_a_waiting_launch_too(oracle_lib, product_lib):
    """Behind the LAST iteration the loop allows, a launch is made ahead that can only be told to do the association: a loop
    that runs into icp_iter (here: forced, BASELINE config 3's arrangement) starts its association ~1 us after the host's last
    step, like one that converges.  Every result is the oracle's and every tracked frame took the path."""
    import ctypes as C
    fo, nv = seeded(oracle_lib, 50000, 640, 480, icp_force_iters=1)
    fh, _ = seeded(product_lib, 50000, 640, 480, icp_force_iters=1, pipeline_depth=2, extract_batch=2)
    frames = [util.frame(k, 640, 480, noise=True, holes=0.02) for k in range(6)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)
    product_lib.lib.ssf_waiter_matches.restype = C.c_longlong
    product_lib.lib.ssf_waiter_matches.argtypes = [C.c_void_p]
    iters = [r["icp_iters"] for r in got]
    assert all(i == 10 for i in iters), iters
    assert product_lib.lib.ssf_waiter_matches(fh.h) == len(got), (product_lib.lib.ssf_waiter_matches(fh.h), iters)


def test_a_late_word_to_the_waiting_launch_is_repaired_not_trusted(oracle_lib, lab_lib):
    """The host's word SSF_ICP_GO_MATCH has no acknowledgement: a calling thread stalled for longer than the waiting launch's
    0.25 s bound (descheduled, debugger, SIGSTOP -- here a test hook sleeps 0.35 s in front of the word) posts it after the
    resident workgroups have given up, and only the late-dispatched part of the grid would associate.  The host bounds the
    device's wait with its own clock and runs the association again as a launch of its own: results stay the oracle's."""
    import ctypes as C
    product_lib = lab_lib            # (the stall is a fault-injection hook: it exists in the lab build of the same sources only)
    fo, nv = seeded(oracle_lib, 50000, 640, 480)
    fh, _ = seeded(product_lib, 50000, 640, 480)
    L = product_lib.lib
    L.ssf_dbg_stall_before_match_us.argtypes = [C.c_void_p, C.c_longlong]; L.ssf_dbg_stall_before_match_us.restype = None
    L.ssf_waiter_match_repairs.argtypes = [C.c_void_p]; L.ssf_waiter_match_repairs.restype = C.c_longlong
    L.ssf_dbg_stall_before_match_us(fh.h, 350000)
    for k in range(4):
        rgb, depth = util.frame(k, 640, 480, noise=True, holes=0.02)
        ro, rh = fo.process_frame(rgb, depth), fh.process_frame(rgb, depth)
        util.same_result(ro, rh)
    util.compare_state(fo, fh)
    assert L.ssf_waiter_match_repairs(fh.h) >= 1, "no frame ended its ICP loop with a launch waiting: the path was not taken"


@pytest.mark.parametrize("switch", ["SSF_PASS_XCD=0", "SSF_PASS_TEAM=1"])
def test_relabelling_passes_bit_exact_under_the_lab_switches(switch):
    """Two arms of the relabelling pass that only exist in the LAB build of the sources (the product reads no environment variable),
    each in a process of its own (the switches are read once per process):
      SSF_PASS_XCD=0   tiles in plain grid order instead of the XCD-aware order: only speed may differ;
      SSF_PASS_TEAM=1  the passes of a phase in ONE launch whose workgroups stay, a frame per XCD, meeting inside their XCD between
                       passes and reading the previous pass' output past the L1 (lab/passes_team.inc): built in
                       round 5, exact, slower (profiles/pass_team_r05.txt).  Whole frames, batches and pipelined sequences.
    The comparison against the oracle (drift-out-of-window parameter sets included) must hold under all of them."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    name, value = switch.split("=")
    env = dict(os.environ, SSF_PRODUCT_VARIANT="lab", **{name: value})
    r = subprocess.run([sys.executable, "-m", "pytest", os.path.join(root, "tests", "test_parity_gpu.py"), "-q", "-m", "gpu", "-x",
                        "-k", ("test_segmentation_parameter_space_batched or test_golden_vectors or test_seeded_model_50k_bit_exact or "
                               "test_pipelined_equals_sequential_bit_exact") if name == "SSF_PASS_TEAM" else
                        "test_every_relabelling_pass_bit_exact or test_segmentation_parameter_space_batched"], cwd=root, env=env,
                       capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stdout[-3000:] + r.stderr[-2000:]
    assert " passed" in r.stdout and "no tests ran" not in r.stdout, r.stdout[-1000:]


def test_rehoming_into_a_full_shard_turns_the_surplus_away(product_lib):
    """ssf_rehome_end on the product: same rule as the oracle's (tests/test_sharded.py): surplus arrivals turned away in table
    order, their number returned, never an error after the source ranks have committed"""
    from test_sharded import rehoming_into_a_full_shard
    assert rehoming_into_a_full_shard(product_lib) > 0


def test_the_product_reads_no_environment(product_lib, lab_lib):
    """The product library contains no environment switch (no `SSF_...` string at all); the lab build of the same sources has them.
    (Until round 5 the product also refused ssf_debug_set_bin_min_rows: the tile-sorted copy was a lab arm then.)"""
    import subprocess
    names = lambda path: [l for l in subprocess.run(["strings", path], stdout=subprocess.PIPE, text=True).stdout.splitlines() if l.startswith("SSF_")]
    assert names(product_lib.path) == []
    assert len(names(lab_lib.path)) > 10
    f = binding.Fusion(product_lib, util.make_cfg(product_lib, 160, 128, nb_supersurfels_max=2048))
    f.set_bin_min_rows(0); f.set_bin_min_rows(-1)


def test_a_later_handle_runs_on_the_streams_of_an_earlier_one(oracle_lib, product_lib):
    """The runtime maps streams onto hardware queues at creation, depending on the process' history: the second handle of a process
    ran the same sequence at 6400 instead of 11 300 frames/s (round 4).  A destroyed handle leaves its streams in a pool and the
    next one takes them; results are the oracle's on both."""
    L = product_lib.lib
    L.ssf_pooled_streams.restype = __import__("ctypes").c_int
    frames = [util.frame(k, 160, 128) for k in range(3)]
    def run():
        fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, 160, 128))
        fh = binding.Fusion(product_lib, util.make_cfg(product_lib, 160, 128, pipeline_depth=2, extract_batch=2))
        idle_while_alive = L.ssf_pooled_streams()
        for rgb, depth in frames:
            util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)
        fh.close(); fo.close()
        return idle_while_alive
    run()
    after_first = L.ssf_pooled_streams()
    assert after_first >= 4, after_first                    # track stream + three extract contexts (+ the capture stream)
    alive = run()
    assert alive <= after_first - 4, (alive, after_first)   # the second handle took them
    assert L.ssf_pooled_streams() >= after_first        # ... and gave them back


@pytest.mark.gpu
@pytest.mark.parametrize("batch", [12, 16])
def test_sequences_with_twelve_and_sixteen_frames_per_extract_launch(batch, oracle_lib, product_lib):
    """SSF_MAX_EXTRACT_BATCH is 16 since round 4 (12 frames per launch is the optimum once the depth pre-filter is in the frame): a
    sequence of host frames with the pre-filter on, pipelined 2 x batch -- leading batches of 3/8 and 5/8, full ones, a partial last
    one -- equals the oracle's frame-by-frame run."""
    W, H, nf = 160, 128, 41
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, depth_prefilter=1))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, depth_prefilter=1, pipeline_depth=2, extract_batch=batch))
    assert fh.pipeline_capacity() == 3 * batch
    frames = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(nf)]
    frames = [(np.ascontiguousarray(r), np.ascontiguousarray(d)) for r, d in frames]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)


@pytest.mark.gpu
def test_a_long_sequence_of_host_frames_wraps_the_upload_ring(oracle_lib, product_lib):
    """Host frames of a sequence are staged and copied ahead by worker threads into a ring of (contexts + 3) x batch + 2 slots; a slot is
    reused once the frame that held it has been tracked.  130 frames through a ring of 26: five laps, every frame the oracle's."""
    W, H, nf = 160, 128, 130
    fo = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H))
    fh = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=2, extract_batch=4))
    base = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(26)]
    order = [(i % 50) if (i % 50) < 26 else 50 - (i % 50) for i in range(nf)]          # the orbit forth and back
    frames = [(np.ascontiguousarray(base[k][0]), np.ascontiguousarray(base[k][1])) for k in order]
    want = [fo.process_frame(r, d) for r, d in frames]
    got = fh.process_sequence([r.ctypes.data for r, _ in frames], [d.ctypes.data for _, d in frames], on_device=False)
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(fo, fh)
    import ctypes as C
    st = (C.c_double * 6)()
    product_lib.lib.ssf_upload_stats.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
    assert product_lib.lib.ssf_upload_stats(fh.h, st) == 0 and int(st[1]) == nf, list(st)
