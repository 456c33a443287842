"""Stage-by-stage HIP vs oracle comparison on a short synthetic orbit (diagnostic, run on the GPU box)."""
import sys, os, time, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from supersurfel_fusion_amd import binding, synthetic

def cmp(name, a, b, exact=True):
    a = np.asarray(a); b = np.asarray(b)
    if a.dtype.kind == 'f':
        same = (a.view(np.uint32 if a.dtype == np.float32 else np.uint64) == b.view(np.uint32 if a.dtype == np.float32 else np.uint64)) | (np.isnan(a) & np.isnan(b))
        nbad = int((~same).sum())
        with np.errstate(all='ignore'):
            md = float(np.nanmax(np.abs(a.astype(np.float64) - b.astype(np.float64)))) if a.size else 0.0
        print(f"  {name:16s} bit-mismatch {nbad}/{a.size} maxabs {md:.3g}")
    else:
        nbad = int((a != b).sum())
        print(f"  {name:16s} mismatch {nbad}/{a.size}")
    return nbad

W, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (640, 480)
nframes = int(sys.argv[3]) if len(sys.argv) > 3 else 4
K = synthetic.intrinsics(W, H)
olib = binding.Library(os.path.join(os.path.dirname(__file__), '..', 'oracle', '_build', 'libssf_oracle.so'))
hlib = binding.load_product()
kw = dict({k: K[k] for k in ('width', 'height', 'fx', 'fy', 'cx', 'cy')}, lambda_pos=10., lambda_bound=1000., lambda_size=1000.,
          lambda_disp=1e8, filter_iter=3, conf_thresh=2560., nb_supersurfels_max=100000, icp_cov_thresh=0.05)
fo = binding.Fusion(olib, olib.default_config(**kw))
fh = binding.Fusion(hlib, hlib.default_config(**kw))
total_bad = 0
# bisect the relabelling passes on frame 0
R, t = synthetic.orbit_pose(0)
rgb, depth, _ = synthetic.render(R, t, W, H, noise=True, holes=0.02, rng=np.random.default_rng(0))
for mp in (1, 2, 4, 8, 20, 21, 24, 40):
    fo2 = binding.Fusion(olib, olib.default_config(**kw)); fh2 = binding.Fusion(hlib, hlib.default_config(**kw))
    fo2.set_max_passes(mp); fh2.set_max_passes(mp)
    fo2.stage_extract(rgb, depth); fh2.stage_extract(rgb, depth)
    print(f"max_passes={mp}")
    b = cmp("labels", fo2.index_map(), fh2.index_map()); b += cmp("inliers", fo2.inlier_map(), fh2.inlier_map())
    b += cmp("superpixels", fo2.superpixels(), fh2.superpixels())
    total_bad += b
    if b: break
for k in range(nframes):
    R, t = synthetic.orbit_pose(k)
    rgb, depth, _ = synthetic.render(R, t, W, H, noise=True, holes=0.02, rng=np.random.default_rng(k))
    print(f"frame {k}")
    fo.stage_extract(rgb, depth); fh.stage_extract(rgb, depth)
    total_bad += cmp("labels", fo.index_map(), fh.index_map())
    total_bad += cmp("boundary", fo.boundary_map(), fh.boundary_map())
    total_bad += cmp("inliers", fo.inlier_map(), fh.inlier_map())
    total_bad += cmp("superpixels", fo.superpixels(), fh.superpixels())
    total_bad += cmp("plane_depth", fo.plane_depth(), fh.plane_depth())
    a, b = fo.get_frame(), fh.get_frame()
    valid = a['confidences'] > 0
    total_bad += cmp("frame conf", a['confidences'], b['confidences'])
    for name in ('positions', 'colors', 'stamps', 'orientations', 'shapes', 'dims'):
        total_bad += cmp("frame " + name, a[name][valid], b[name][valid])
    fo.icp_begin(); fh.icp_begin()
    again_o = again_h = True; it = 0
    while again_o or again_h:
        so, sh = fo.icp_accumulate(), fh.icp_accumulate()
        total_bad += cmp(f"icp sums it{it}", so, sh)
        again_o, again_h = fo.icp_update(so), fh.icp_update(sh); it += 1
        if it > 12: break
    vo, vh = fo.icp_end(), fh.icp_end()
    print("  icp valid", vo, vh, "iters", it)
    total_bad += cmp("pose", fo.get_pose(), fh.get_pose())
    (bo, mo), (bh, mh) = fo.match(), fh.match()
    total_bad += cmp("match best", bo, bh); total_bad += cmp("matched", mo, mh)
    ro, rh = fo.fuse(bo, mo), fh.fuse(bh, mh)
    keys = ('n_model', 'n_visible', 'n_removed', 'n_inserted', 'n_updated', 'stamp')
    print("  counts", {q: ro[q] for q in keys}, {q: rh[q] for q in keys})
    total_bad += sum(ro[q] != rh[q] for q in keys)
    mo_, mh_ = fo.get_model(), fh.get_model()
    for name in mo_:
        total_bad += cmp("model " + name, mo_[name], mh_[name])
print("TOTAL MISMATCH", total_bad)
