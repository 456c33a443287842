"""Shared helpers of the test-suite: configurations, synthetic frames, bit-exact comparison."""
import numpy as np

from supersurfel_fusion_amd import binding, synthetic

# the reference's rgbd_benchmark launch column with VO/MOD/LC off
BENCH_PARAMS = dict(lambda_pos=10.0, lambda_bound=1000.0, lambda_size=1000.0, lambda_disp=1e8, thresh_disp=1e-4,
                    seg_iter=10, filter_iter=3, delta_t=20, conf_thresh=2560.0, icp_iter=10, icp_cov_thresh=0.05)


def make_cfg(lib, W, H, **kw):
    """Test configuration.  depth_prefilter is OFF here unless a test asks for it: the path under test takes "depth after
    the pre-filter" as its input and the filter is covered on its own (tests/test_prefilter.py,
    tests/test_replay.py run it inside process_frame); the library default is ON, as in the reference."""
    K = synthetic.intrinsics(W, H)
    args = dict({k: K[k] for k in ("width", "height", "fx", "fy", "cx", "cy")}, nb_supersurfels_max=20000, depth_prefilter=0)
    args.update(BENCH_PARAMS)
    args.update(kw)
    return lib.default_config(**args)


def frame(k, W, H, noise=True, holes=0.0):
    R, t = synthetic.orbit_pose(k)
    rgb, depth, _ = synthetic.render(R, t, W, H, noise=noise, holes=holes, rng=np.random.default_rng(1000 + k))
    return rgb, depth


def bits(a):
    a = np.ascontiguousarray(a)
    if a.dtype == np.float32:
        return a.view(np.uint32)
    if a.dtype == np.float64:
        return a.view(np.uint64)
    return a


def assert_same_bits(a, b, what):
    a, b = np.asarray(a), np.asarray(b)
    assert a.shape == b.shape, (what, a.shape, b.shape)
    if a.dtype.kind == "f":
        same = (bits(a) == bits(b)) | (np.isnan(a) & np.isnan(b))
    else:
        same = a == b
    assert bool(same.all()), "%s: %d of %d elements differ" % (what, int((~same).sum()), a.size)


def compare_state(fa, fb, maps=True, frame_surfels=True):
    """Bit-exact comparison of two Fusion handles after the same calls."""
    if maps:
        assert_same_bits(fa.index_map(), fb.index_map(), "label map")
        assert_same_bits(fa.boundary_map(), fb.boundary_map(), "boundary map")
        assert_same_bits(fa.inlier_map(), fb.inlier_map(), "inlier map")
        assert_same_bits(fa.plane_depth(), fb.plane_depth(), "plane depth")
        assert_same_bits(fa.superpixels(), fb.superpixels(), "superpixel table")
        assert_same_bits(fa.preview_image(), fb.preview_image(), "preview image (computeSuperpixelSegIm)")
    if frame_surfels:
        a, b = fa.get_frame(), fb.get_frame()
        assert_same_bits(a["confidences"], b["confidences"], "frame confidences")
        valid = a["confidences"] > 0
        for name in a:
            assert_same_bits(a[name][valid], b[name][valid], "frame " + name)
    ca, cb = fa.counts(), fb.counts()
    assert ca == cb, (ca, cb)
    assert_same_bits(fa.get_pose(), fb.get_pose(), "pose")
    ma, mb = fa.get_model(), fb.get_model()
    for name in ma:
        assert_same_bits(ma[name], mb[name], "model " + name)


RESULT_KEYS = ("icp_valid", "icp_iters", "n_model", "n_visible", "n_removed", "n_inserted", "n_updated", "stamp")


def same_result(ra, rb):
    for k in RESULT_KEYS:
        assert ra[k] == rb[k], (k, ra[k], rb[k])
    assert_same_bits(ra["pose"], rb["pose"], "result pose")


def device_to_host(ptr, shape, dtype):
    """copy a raw device pointer (ssf_get_model_device) to a numpy array through the HIP runtime already in the process"""
    import ctypes as C
    path = [l.split()[-1] for l in open("/proc/self/maps") if "libamdhip64" in l][0]
    hip = C.CDLL(path)
    out = np.zeros(shape, dtype)
    hip.hipMemcpy.argtypes = [C.c_void_p, C.c_void_p, C.c_size_t, C.c_int]
    assert hip.hipMemcpy(out.ctypes.data_as(C.c_void_p), C.c_void_p(ptr), out.nbytes, 2) == 0
    return out


def exchange_and_fuse(ranks, best, matched):
    """The fuse stage of a sharded map with the ranks emulated in one process: first halves, SUM of the migrant tables
    (what the int32 all-reduce does: at most one rank fills a slot), second halves.  Returns the per-rank results."""
    tables = [f.fuse_begin(best, matched) for f in ranks]
    filled = np.stack([t[:, 0] != 0 for t in tables]).sum(axis=0)
    assert (filled <= 1).all(), "two ranks filled the same slot of the migrant table"
    total = np.sum(np.stack(tables).astype(np.int64), axis=0).astype(np.int32)
    return [f.fuse_end(total) for f in ranks]


def arrival_at_a_full_shard(lib, W=160, H=128):
    """A shard (rank 0 of 2) filled to its capacity receives one more row through the migrant table: the row cannot be
    stored, so it is lost to the whole map (its source shard has already let it go) and must show up in n_removed of the
    frame -- otherwise the sums of the per-shard counters stop being the unsharded bookkeeping.  Returns (handle,
    result without the arrival, result with it)."""
    out = []
    for crafted in (False, True):
        S = ((W + 15) // 16) * ((H + 15) // 16)
        cap = S + 16
        f = binding.Fusion(lib, make_cfg(lib, W, H, nb_supersurfels_max=cap, rank=0, nranks=2, shard_tile=0.25))
        f.process_frame(*frame(0, W, H))
        m = f.get_model()
        valid = np.flatnonzero(m["confidences"] > 0)
        pick = valid[np.arange(cap) % len(valid)]                        # capacity rows, all valid, all in view
        full = {name: m[name][pick].copy() for name in m}
        full["confidences"][:] = 5000.0                                  # nothing is culled as unstable
        f.set_model(full, cap, 1)
        f.stage_extract(*frame(1, W, H))
        f.icp_begin()
        while f.icp_update(f.icp_accumulate()):
            pass
        f.icp_end()
        best, matched = f.match()
        table = f.fuse_begin(best, matched)
        if crafted:
            slot = int(np.flatnonzero(table[:, 0] == 0)[0])
            row = np.concatenate([full["positions"][0], full["colors"][0], full["stamps"][0].view(np.float32), full["orientations"][0],
                                  full["shapes"][0], full["dims"][0], full["confidences"][:1]]).astype(np.float32)
            table[slot, 0] = 1                                           # destination rank 0 (+ 1)
            table[slot, 2:28] = row.view(np.int32)
        out.append((f, f.fuse_end(table)))
    (f0, r0), (f1, r1) = out
    assert r1["n_removed"] == r0["n_removed"] + 1, (r0, r1)
    assert r1["n_model"] == r0["n_model"] and r1["n_visible"] == r0["n_visible"]
    return f1, r0, r1


def row_hash(model):
    """a 32-bit key per model row that depends only on the row's content (so that a shard and the unsharded map derive
    the same per-row quantities, whatever the order of their rows)"""
    n = len(model["confidences"])
    if n == 0:
        return np.zeros(0, np.uint64)
    w = np.concatenate([np.ascontiguousarray(model[name]).reshape(n, -1).view(np.uint32) for name in ("positions", "stamps", "dims")], axis=1).astype(np.uint64)
    h = np.zeros(n, np.uint64)
    for j in range(w.shape[1]):
        h = (h * np.uint64(1099511628211) + w[:, j] + np.uint64(0x9E3779B9)) & np.uint64(0xFFFFFFFFFFFF)
    return (h ^ (h >> np.uint64(17))).astype(np.uint64)


def deformation_for(model, n_nodes, seed=5, angle=0.01, shift=0.004):
    """a synthetic loop-closure deformation (applyDeformation's inputs): the node set depends only on (n_nodes, seed), a
    row's four node indices and weights only on the row itself -- the same deformation for every sharding of one map"""
    rng = np.random.default_rng(seed)
    npos = rng.uniform(-3, 3, (n_nodes, 3)).astype(np.float32)
    ang = rng.uniform(-angle, angle, (n_nodes, 3))
    nrot = np.stack([(synthetic.rot_y(a[1]) @ synthetic.rot_x(a[0])).reshape(9) for a in ang]).astype(np.float32)
    ntr = rng.uniform(-shift, shift, (n_nodes, 3)).astype(np.float32)
    h = row_hash(model)
    idx = np.stack([(h >> np.uint64(8 * j)) % np.uint64(n_nodes) for j in range(4)], axis=1).astype(np.int32)
    raw = np.stack([((h >> np.uint64(5 * j + 3)) & np.uint64(31)).astype(np.float32) + 1.0 for j in range(4)], axis=1)
    w = (raw / raw.sum(axis=1, keepdims=True)).astype(np.float32)
    return npos, nrot, ntr, w, idx


def rehome_in_process(ranks):
    """the re-homing sweep of ssf_rehome_begin / _end for ranks that live in one process: every rank's leaving rows, in
    rank order, offered to every rank"""
    tables = [f.rehome_begin() for f in ranks]
    allt = np.concatenate(tables) if sum(len(t) for t in tables) else np.zeros((0, binding.MIGRANT_WORDS), np.int32)
    for f in ranks:
        turned = f.rehome_end(allt)
        assert turned == 0, "a shard turned %d arrival(s) away: they are lost to the map" % turned
    return [len(t) for t in tables]


def rows_multiset(m):
    n = len(m["confidences"])
    cols = [np.ascontiguousarray(m[name]).reshape(n, -1).view(np.uint32) for name, _, _ in binding.SURFEL_FIELDS]
    rows = np.concatenate(cols, axis=1)
    return rows[np.lexsort(rows.T[::-1])]
