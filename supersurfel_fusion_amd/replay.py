"""TUM RGB-D replay harness: the counterpart of the reference's benchmark node
(node/supersurfel_fusion_rgbd_benchmark_node.cpp:573-744) for the hot path.

It reproduces the node's call pattern exactly: parse `associations_with_gt.txt` (stamp rgb-path stamp
depth-path [stamp tx ty tz qx qy qz qw]), decode the PNGs, keep RGB order, convert the 16-bit depth
with `depth_scale` (0.0002 for TUM's 5000 counts per metre, launch/supersurfel_fusion_rgbd_benchmark.launch:47),
call process_frame once per line and append `stamp tx ty tz qx qy qz qw` to the trajectory file
(`estimated.txt`, same layout as the files the reference commits next to its datasets).  Optionally
the model is exported in the reference's text format at the end (exportModel,
core/src/supersurfel_fusion.cu:595-633).

The depth pre-filter of processFrame (cv::cuda::bilateralFilter(depth, -1, 0.03, 4.5), supersurfel_fusion.cu:180)
is ON, as in the reference (BENCHMARK_LAUNCH below).  Sparse VO, MOD and loop closure of the reference are out of
scope: the pose prior is the previous pose.  Pre-decoded frames (np.savez archives produced by `pack_frames`) replace the PNG files on
boxes without the dataset."""
import argparse
import os

import numpy as np

# The rgbd_benchmark launch column (launch/supersurfel_fusion_rgbd_benchmark.launch) with TUM fr1
# intrinsics (rgbd_benchmark/fr1_cam.yaml): what SupersurfelFusionRGBDBenchmarkNode hands to initialize().
BENCHMARK_LAUNCH = dict(width=640, height=480, fx=525.0, fy=525.0, cx=319.5, cy=239.5, cell_size=16, seg_iter=10,
                        lambda_pos=10.0, lambda_bound=1000.0, lambda_size=1000.0, lambda_disp=1e8, thresh_disp=1e-4,
                        filter_iter=3, filter_alpha=0.1, filter_beta=1.0, filter_threshold=0.05, range_min=0.2, range_max=5.0,
                        delta_t=20, conf_thresh=2560.0, nb_supersurfels_max=100000, icp_iter=10, icp_cov_thresh=0.05,
                        depth_prefilter=1, prefilter_sigma_color=0.03, prefilter_sigma_space=4.5)
# rgbd_benchmark/fr3_cam.yaml: the intrinsics the launch file loads for rgbd_dataset_freiburg3_walking_halfsphere
FR3_INTRINSICS = dict(fx=535.4, fy=539.2, cx=320.1, cy=247.6)


def read_associations(path, max_frames=None):
    """-> list of dict(stamp, rgb, depth, gt) ; gt = (t[3], q[4] xyzw) or None"""
    out = []
    with open(path) as f:
        for line in f:
            w = line.split()
            if len(w) < 4:
                break                                   # the node stops at the first short line (:592-593)
            gt = None
            if len(w) >= 12:
                v = [float(x) for x in w[5:12]]
                gt = (np.array(v[:3]), np.array(v[3:7]))
            out.append(dict(stamp=w[0], rgb=w[1], depth=w[3], gt=gt))
            if max_frames and len(out) >= max_frames:
                break
    return out


def decode_frame(dataset_dir, entry, depth_scale):
    from PIL import Image
    rgb = np.asarray(Image.open(os.path.join(dataset_dir, entry["rgb"])).convert("RGB"), np.uint8)
    d16 = np.asarray(Image.open(os.path.join(dataset_dir, entry["depth"])), np.uint16)
    return rgb, convert_depth(d16, depth_scale)


def convert_depth(d16, depth_scale):
    """depth_u16.convertTo(depth, CV_32FC1, depthScale): float(v) * scale evaluated in double, rounded to f32"""
    return (d16.astype(np.float64) * float(depth_scale)).astype(np.float32)


def pack_frames(dataset_dir, assoc_path, out_npz, n):
    """Pre-decode the first n frames into one archive (rgb u8, depth u16, association lines)."""
    ent = read_associations(assoc_path, n)
    from PIL import Image
    arrs = {"lines": np.array([l for l in open(assoc_path).read().split("\n")[:n]])}
    for i, e in enumerate(ent):
        arrs["rgb%d" % i] = np.asarray(Image.open(os.path.join(dataset_dir, e["rgb"])).convert("RGB"), np.uint8)
        arrs["depth%d" % i] = np.asarray(Image.open(os.path.join(dataset_dir, e["depth"])), np.uint16)
    np.savez_compressed(out_npz, **arrs)


def rot_to_quat_xyzw(R):
    """tf::Matrix3x3::getRotation (Shoemake), the conversion the node applies before writing"""
    R = np.asarray(R, np.float64)
    tr = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4)
    if tr > 0:
        s = np.sqrt(tr + 1.0)
        q[3] = 0.5 * s; s = 0.5 / s
        q[0] = (R[2, 1] - R[1, 2]) * s; q[1] = (R[0, 2] - R[2, 0]) * s; q[2] = (R[1, 0] - R[0, 1]) * s
    else:
        i = 0 if R[0, 0] >= R[1, 1] and R[0, 0] >= R[2, 2] else (1 if R[1, 1] >= R[2, 2] else 2)
        j, k = (i + 1) % 3, (i + 2) % 3
        s = np.sqrt(R[i, i] - R[j, j] - R[k, k] + 1.0)
        q[i] = 0.5 * s; s = 0.5 / s
        q[3] = (R[k, j] - R[j, k]) * s; q[j] = (R[j, i] + R[i, j]) * s; q[k] = (R[k, i] + R[i, k]) * s
    return q


def tum_line(stamp, pose12):
    """`stamp tx ty tz qx qy qz qw` with ostream's default 6 significant digits (:727-729)"""
    R = np.asarray(pose12[:9], np.float64).reshape(3, 3)
    t = np.asarray(pose12[9:], np.float64)
    q = rot_to_quat_xyzw(R)
    return " ".join([stamp] + ["%g" % v for v in list(t) + list(q)])


def replay(fusion, frames, out_path=None, export_model=None, pipelined=False):
    """frames: iterable of (stamp, rgb u8 HxWx3, depth f32 HxW).  Returns (lines, results).
    pipelined: decode / submit ahead while earlier frames are tracked and fused (ssf_submit_frame /
    ssf_process_submitted, for handles created with pipeline_depth / extract_batch > 0 / 1); the trajectory is the
    same, bit for bit, as with one process_frame per line."""
    lines, results = [], []
    if not pipelined:
        for stamp, rgb, depth in frames:
            r = fusion.process_frame(rgb, depth)
            results.append(r)
            lines.append(tum_line(stamp, r["pose"]))
    else:
        it, stamps, done = iter(frames), [], False
        held = []                                         # submitted host buffers stay alive until their frame is processed
        while True:
            while not done and fusion.can_submit():
                try:
                    stamp, rgb, depth = next(it)
                except StopIteration:
                    done = True
                    break
                rgb, depth = np.ascontiguousarray(rgb, np.uint8), np.ascontiguousarray(depth, np.float32)
                fusion.submit_frame(rgb, depth)           # (the copy is asynchronous: ssf.h, ssf_submit_frame)
                held.append((rgb, depth))
                stamps.append(stamp)
            if fusion.pending_frames() == 0:
                break
            r = fusion.process_submitted().as_dict()
            held.pop(0)
            results.append(r)
            lines.append(tum_line(stamps[len(lines)], r["pose"]))
    if out_path:
        with open(out_path, "w") as f:
            f.write("\n".join(lines) + "\n")
    if export_model:
        fusion.export_model_txt(export_model)
    return lines, results


def stage_figures(fusion, raw_depth):
    """Per-stage figures of the frame just processed against the RAW sensor depth it came from (a real-data sanity check
    of the extract stage that does not go through the trajectory): the plane-rendered depth of a3-a5 against the raw
    depth on the inlier pixels, the inlier share of the valid-depth pixels, the share of superpixels that became
    valid frame supersurfels (a6)."""
    plane = fusion.plane_depth().astype(np.float64)
    inl = fusion.inlier_map() != 0
    raw = np.asarray(raw_depth, np.float64)
    ok = inl & (raw > 0) & np.isfinite(plane)
    rel = np.abs(plane[ok] - raw[ok]) / raw[ok]
    fr = fusion.get_frame()
    return dict(plane_vs_raw_median=float(np.median(rel)) if rel.size else float("nan"),
                plane_vs_raw_p90=float(np.percentile(rel, 90)) if rel.size else float("nan"),
                inlier_share_of_valid_depth=float(ok.sum() / max(1, (raw > 0).sum())),
                valid_supersurfel_share=float((fr["confidences"] > 0).mean()))


def quat_xyzw_to_rot(q):
    x, y, z, w = [float(v) for v in q]
    return np.array([[1 - 2 * (y * y + z * z), 2 * (x * y - z * w), 2 * (x * z + y * w)],
                     [2 * (x * y + z * w), 1 - 2 * (x * x + z * z), 2 * (y * z - x * w)],
                     [2 * (x * z - y * w), 2 * (y * z + x * w), 1 - 2 * (x * x + y * y)]])


def prior_consistency(fusion, frames, prior_xyz, prior_quat):
    """Every frame (rgb, depth) is processed with prior_xyz[i] / prior_quat[i] (camera-to-map, TUM order) as its pose
    prior -- processFrame's `pose = vo->getPose()` -- and the size of the ICP correction is recorded where the ICP
    result was accepted.  Frame 0 builds the map at its prior."""
    dt, da, valid, n = [], [], 0, 0
    for i, (rgb, depth) in enumerate(frames):
        R = quat_xyzw_to_rot(prior_quat[i])
        prior = np.concatenate([R.reshape(-1), np.asarray(prior_xyz[i], np.float64)]).astype(np.float32)
        r = fusion.process_frame(rgb, depth, prior_pose=prior)
        n += 1
        if i == 0 or not r["icp_valid"]:
            continue
        valid += 1
        P = np.asarray(r["pose"], np.float64)
        dR = R.T @ P[:9].reshape(3, 3)
        dt.append(float(np.linalg.norm(P[9:] - prior_xyz[i])))
        da.append(float(np.degrees(np.arccos(np.clip((np.trace(dR) - 1.0) / 2.0, -1.0, 1.0)))))
    return dict(frames=n, icp_valid_frames=valid,
                correction_translation_median_m=float(np.median(dt)) if dt else None,
                correction_translation_p90_m=float(np.percentile(dt, 90)) if dt else None,
                correction_rotation_median_deg=float(np.median(da)) if da else None,
                correction_rotation_p90_deg=float(np.percentile(da, 90)) if da else None)


def frames_from_dataset(dataset_dir, depth_scale=0.0002, max_frames=None):
    for e in read_associations(os.path.join(dataset_dir, "associations_with_gt.txt"), max_frames):
        rgb, depth = decode_frame(dataset_dir, e, depth_scale)
        yield e["stamp"], rgb, depth


def frames_from_npz(path, depth_scale=0.0002):
    z = np.load(path)
    i = 0
    while "rgb%d" % i in z:
        stamp = str(z["lines"][i]).split()[0]
        yield stamp, z["rgb%d" % i], convert_depth(z["depth%d" % i], depth_scale)
        i += 1


def read_trajectory(path):
    """TUM trajectory file -> (stamps, xyz (n,3), quat xyzw (n,4)); '#' lines skipped"""
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("#")]
    v = np.array([[float(x) for x in r[1:8]] for r in rows])
    return [r[0] for r in rows], v[:, :3], v[:, 3:7]


def ate_rmse(est_xyz, gt_xyz):
    """Absolute trajectory error after Horn alignment (rigid, no scale), as the TUM tools compute it."""
    est, gt = np.asarray(est_xyz, np.float64), np.asarray(gt_xyz, np.float64)
    mu_e, mu_g = est.mean(0), gt.mean(0)
    Wm = (gt - mu_g).T @ (est - mu_e)
    U, _, Vt = np.linalg.svd(Wm)
    S = np.eye(3)
    if np.linalg.det(U) * np.linalg.det(Vt) < 0:
        S[2, 2] = -1
    R = U @ S @ Vt
    err = (gt - mu_g) - (est - mu_e) @ R.T
    return float(np.sqrt((err ** 2).sum(1).mean()))


def main():
    ap = argparse.ArgumentParser(description=__doc__.split("\n\n")[0])
    ap.add_argument("--dataset", help="TUM sequence directory with associations_with_gt.txt")
    ap.add_argument("--npz", help="pre-decoded frames (pack_frames) instead of --dataset")
    ap.add_argument("--out", default="estimated.txt")
    ap.add_argument("--depth-scale", type=float, default=0.0002)
    ap.add_argument("--max-frames", type=int, default=None)
    ap.add_argument("--export-model", default=None)
    ap.add_argument("--pipelined", action="store_true", help="extract of later frames runs ahead (pipeline_depth 2, extract_batch 4)")
    a = ap.parse_args()
    from . import binding
    lib = binding.load_product()
    cfg = lib.default_config(pipeline_depth=2 if a.pipelined else 0, extract_batch=4 if a.pipelined else 1, **BENCHMARK_LAUNCH)
    f = binding.Fusion(lib, cfg)
    frames = frames_from_npz(a.npz, a.depth_scale) if a.npz else frames_from_dataset(a.dataset, a.depth_scale, a.max_frames)
    lines, res = replay(f, frames, a.out, a.export_model, pipelined=a.pipelined)
    print("%d frames -> %s ; %d supersurfels" % (len(lines), a.out, res[-1]["n_model"] if res else 0))


if __name__ == "__main__":
    main()
