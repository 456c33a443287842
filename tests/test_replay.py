"""Replay harness: TUM association parsing, depth conversion, trajectory
line format, ATE helper -- and the real-data smoke fixture through the oracle (CPU) and, bit-exact,
through the HIP product (GPU)."""
import os

import numpy as np
import pytest

import util
from supersurfel_fusion_amd import binding, replay, synthetic

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLD = os.path.join(ROOT, "tests", "golden")
TUM = os.path.join(GOLD, "tum_fr1_xyz_8frames.npz")


def tum_fusion(lib, **kw):
    """the benchmark node's configuration (replay.BENCHMARK_LAUNCH: launch-file parameters, depth pre-filter on)"""
    cfg = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=20000); cfg.update(kw)
    return binding.Fusion(lib, lib.default_config(**cfg))


def tum_frames(n):
    for i, fr in enumerate(replay.frames_from_npz(TUM)):
        if i < n:
            yield fr


def test_association_parsing(tmp_path):
    z = np.load(TUM)
    p = tmp_path / "associations_with_gt.txt"
    p.write_text("\n".join(str(l) for l in z["lines"]) + "\n\n")
    ent = replay.read_associations(str(p))
    assert len(ent) == 8
    assert ent[0]["stamp"] == "1305031102.175304" and ent[0]["rgb"] == "rgb/1305031102.175304.png"
    assert ent[0]["depth"] == "depth/1305031102.160407.png"
    assert np.allclose(ent[0]["gt"][0], [1.3405, 0.6266, 1.6575]) and np.allclose(ent[0]["gt"][1], [0.6574, 0.6126, -0.2949, -0.3248])
    assert len(replay.read_associations(str(p), max_frames=2)) == 2


def test_depth_conversion_matches_convertTo():
    d16 = np.array([[0, 1, 5000, 65535]], np.uint16)
    d = replay.convert_depth(d16, 0.0002)
    assert d.dtype == np.float32 and d[0, 0] == 0.0 and d[0, 2] == np.float32(1.0) and d[0, 3] == np.float32(65535 * 0.0002)


def test_trajectory_line_format():
    R = synthetic.rot_y(0.3) @ synthetic.rot_x(-0.2)
    line = replay.tum_line("1305031102.211214", synthetic.pose12(R, [0.1, -0.2, 0.3]))
    w = line.split()
    assert w[0] == "1305031102.211214" and len(w) == 8
    q = np.array([float(v) for v in w[4:]])
    assert abs(np.linalg.norm(q) - 1) < 1e-5 and q[3] > 0
    # identity pose is written exactly as the reference's first estimated.txt line
    assert replay.tum_line("1305031102.175304", synthetic.pose12(np.eye(3), np.zeros(3))) == "1305031102.175304 0 0 0 0 0 0 1"


def test_ate_is_invariant_to_rigid_motion():
    rng = np.random.default_rng(3)
    gt = rng.uniform(-1, 1, (50, 3))
    R = synthetic.rot_y(0.7) @ synthetic.rot_x(0.4)
    est = gt @ R.T + [0.5, -1, 2]
    assert replay.ate_rmse(est, gt) < 1e-12
    assert abs(replay.ate_rmse(est + [0.01, 0, 0] * (np.arange(50)[:, None] % 2), gt) - 0.005) < 2e-3


def test_real_tum_frames_through_the_oracle(oracle_lib, tmp_path):
    """~25 % of the real depth image is holes; the path must stay finite and track."""
    f = tum_fusion(oracle_lib)
    out = str(tmp_path / "estimated.txt")
    lines, res = replay.replay(f, tum_frames(3), out, str(tmp_path / "model.txt"))
    assert len(lines) == 3 and lines[0].endswith(" 0 0 0 0 0 0 1")
    assert all(np.isfinite(r["pose"]).all() for r in res)
    assert res[1]["icp_valid"] == 1 and res[2]["icp_valid"] == 1
    assert (f.inlier_map() > 0).mean() < 0.8            # holes are not inliers
    # frame-to-frame motion of fr1_xyz is centimetres: compare with the ground-truth displacement
    z = np.load(TUM)
    gt = np.array([[float(v) for v in str(l).split()[5:8]] for l in z["lines"][:3]])
    est = np.array([r["pose"][9:] for r in res])
    assert abs(np.linalg.norm(est[2] - est[0]) - np.linalg.norm(gt[2] - gt[0])) < 0.02
    assert open(out).read().count("\n") == 3


# ---- the anchor to the reference's own verification artefact -----------------------------------
# tests/golden/make_fr1_xyz_trajectory.py ran the oracle over the whole fr1_xyz sequence the reference ships (790
# associated frames) exactly as the benchmark node does, and committed the trajectory next to the ground truth and
# to the reference's own committed estimated.txt.
def test_committed_fr1_xyz_trajectory_is_as_accurate_as_the_reference_s(oracle_lib):
    """ATE (Horn-aligned RMSE over all 790 frames, 8 m of path) of the oracle's hot-path-only trajectory against the
    TUM ground truth, next to the ATE of the trajectory the reference's authors committed (whole system: sparse VO +
    MOD + ICP, parameter set unknown).  Bands: the oracle within 1 cm of the reference's error, and the two
    trajectories within 1.5 cm RMSE of EACH OTHER -- a restatement that mis-read the ICP or the fusion would drift
    by decimetres over 790 frames of a hand-held sequence."""
    _, est, _ = replay.read_trajectory(os.path.join(GOLD, "fr1_xyz_oracle_estimated.txt"))
    _, gt, _ = replay.read_trajectory(os.path.join(GOLD, "fr1_xyz_gt.txt"))
    _, ref, _ = replay.read_trajectory(os.path.join(GOLD, "fr1_xyz_reference_estimated.txt"))
    assert len(est) == len(gt) == len(ref) == 790
    ate_oracle, ate_ref, between = replay.ate_rmse(est, gt), replay.ate_rmse(ref, gt), replay.ate_rmse(est, ref)
    assert abs(ate_ref - 0.0195) < 5e-4                  # the survey's figure for the reference's file
    assert ate_oracle < 0.030 and abs(ate_oracle - ate_ref) < 0.010, (ate_oracle, ate_ref)
    assert between < 0.015, between
    # local accuracy: translation over every 30-frame (1 s) window against the ground truth; the reference's own file
    # scores 0.067 max / 0.0176 mean on this measure, the oracle 0.092 / 0.021
    def window_error(t):
        return np.abs(np.linalg.norm(t[30:] - t[:-30], axis=1) - np.linalg.norm(gt[30:] - gt[:-30], axis=1))
    assert window_error(est).max() < 0.12 and window_error(est).mean() < window_error(ref).mean() + 0.01


def test_oracle_reproduces_the_committed_trajectory(oracle_lib):
    """The committed file is the output of THIS oracle: the first 8 frames (the decoded fixture), same text."""
    want = open(os.path.join(GOLD, "fr1_xyz_oracle_estimated.txt")).read().split("\n")[:8]
    lines, res = replay.replay(tum_fusion(oracle_lib, nb_supersurfels_max=100000), tum_frames(8))
    assert lines == want
    assert [r["icp_valid"] for r in res][1:] == [1] * 7


@pytest.mark.gpu
def test_real_tum_frames_bit_exact_on_gpu(oracle_lib, product_lib):
    """8 real fr1_xyz frames (u16 depth at 5000 / m, ~25 % holes, depth pre-filter on) through the HIP product: the
    trajectory text equals the committed oracle trajectory, every result and the whole state equal the oracle's."""
    want = open(os.path.join(GOLD, "fr1_xyz_oracle_estimated.txt")).read().split("\n")[:8]
    fo, fh = tum_fusion(oracle_lib), tum_fusion(product_lib)
    lo, ro = replay.replay(fo, tum_frames(8))
    lh, rh = replay.replay(fh, tum_frames(8))
    assert lh == want and lo == lh
    for a, b in zip(ro, rh):
        util.same_result(a, b)
    util.compare_state(fo, fh)


def test_pipelined_replay_gives_the_same_trajectory(oracle_lib):
    f1, f2 = tum_fusion(oracle_lib), tum_fusion(oracle_lib, pipeline_depth=1, extract_batch=2)
    l1, _ = replay.replay(f1, tum_frames(3))
    l2, _ = replay.replay(f2, tum_frames(3), pipelined=True)
    assert l1 == l2 and len(l2) == 3


@pytest.mark.gpu
def test_pipelined_replay_on_gpu_equals_the_oracle(oracle_lib, product_lib):
    fo, fh = tum_fusion(oracle_lib), tum_fusion(product_lib, pipeline_depth=2, extract_batch=4)
    lo, ro = replay.replay(fo, tum_frames(8))
    lh, rh = replay.replay(fh, tum_frames(8), pipelined=True)
    assert lo == lh
    for a, b in zip(ro, rh):
        util.same_result(a, b)
    util.compare_state(fo, fh)


# ---- second reference-held sequence: rgbd_dataset_freiburg3_walking_halfsphere (the one the benchmark launch file points at) ----
# tests/golden/make_fr3_walking_trajectory.py and make_prior_consistency.py; a DYNAMIC scene whose reference trajectory is
# produced by the reference's sparse VO + MOD -- what the hot path alone makes of it is recorded as it is.
FR3 = os.path.join(GOLD, "tum_fr3_walking_4frames.npz")


def fr3_fusion(lib, **kw):
    cfg = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=20000, **replay.FR3_INTRINSICS); cfg.update(kw)
    return binding.Fusion(lib, lib.default_config(**cfg))


def test_committed_fr3_walking_reports():
    """What the committed runs say, as bands: (i) with the launch file's covariance gate (0.05) the hot path's ICP result is
    rejected on every one of the 126 frames -- the pose never moves, the reference's trajectory there is its sparse VO's;
    (ii) with the gate at 0.1 the ICP alone, in a scene with two walking people and no MOD mask, tracks most frames but
    drifts (ATE 0.15 m over 1.2 m of path, the reference's whole system: 0.054 m); (iii) the extract stage's per-frame
    figures against the RAW sensor depth are tight and stable across the sequence -- a misread segmentation / plane fit /
    render would show here first."""
    import json
    a = json.load(open(os.path.join(GOLD, "fr3_walking_report.json")))
    b = json.load(open(os.path.join(GOLD, "fr3_walking_report_cov0.1.json")))
    assert a["frames"] == b["frames"] == 126
    assert a["icp_valid_frames"] == 0 and abs(a["ate_rmse_reference_estimated"] - 0.0541) < 5e-4
    assert b["icp_valid_frames"] >= 80 and b["ate_rmse_oracle"] < 0.20
    assert b["step_length_error_median"] < b["step_length_gt_median"]           # frame-to-frame: better than standing still
    for rep in (a, b):
        st = rep["stage"]
        assert st["plane_vs_raw_median"]["p90"] < 0.01                          # planes within 1 % of the raw depth on inliers
        assert st["plane_vs_raw_p90"]["p90"] < 0.04
        assert st["inlier_share_of_valid_depth"]["p10"] > 0.95
        assert 0.25 < st["valid_supersurfel_share"]["median"] < 0.7


def test_committed_prior_consistency():
    """Every frame started from the pose the REFERENCE committed for it (processFrame's `pose = vo->getPose()`): on the
    static fr1_xyz sequence the restated ICP accepts 771 / 790 frames and moves the reference's pose by a median of 3.8 mm
    and 0.32 degrees (median step of the sequence: 11 mm) -- per frame, nothing accumulates, so a misread Jacobian or gate
    would show as a systematic pull away from the reference's poses."""
    import json
    d = json.load(open(os.path.join(GOLD, "prior_consistency.json")))["fr1_xyz"]
    assert d["frames"] == 790 and d["icp_valid_frames"] > 750
    assert d["correction_translation_median_m"] < 0.005 and d["correction_translation_p90_m"] < 0.010
    assert d["correction_rotation_median_deg"] < 0.4 and d["correction_rotation_p90_deg"] < 0.8


def test_prior_consistency_rederived_on_the_committed_frames(oracle_lib):
    """the same measurement re-derived here on the 8 committed fr1_xyz frames with the reference's first 8 poses"""
    _, xyz, quat = replay.read_trajectory(os.path.join(GOLD, "fr1_xyz_reference_estimated.txt"))
    rep = replay.prior_consistency(tum_fusion(oracle_lib), ((rgb, d) for _, rgb, d in tum_frames(8)), xyz[:8], quat[:8])
    assert rep["frames"] == 8 and rep["icp_valid_frames"] == 7
    assert rep["correction_translation_median_m"] < 0.006 and rep["correction_rotation_median_deg"] < 0.5


def test_fr3_walking_frames_stage_figures(oracle_lib):
    """four committed frames of the dynamic sequence (two consecutive pairs, people in view): extract-stage figures against
    the raw depth, and the covariance gate that rejects the ICP result with the launch file's threshold"""
    frames = list(replay.frames_from_npz(FR3))
    for pair in (frames[:2], frames[2:]):
        f = fr3_fusion(oracle_lib)
        for k, (_, rgb, depth) in enumerate(pair):
            r = f.process_frame(rgb, depth)
            fig = replay.stage_figures(f, depth)
            assert fig["plane_vs_raw_median"] < 0.01 and fig["inlier_share_of_valid_depth"] > 0.95
        assert r["icp_iters"] >= 2 and r["icp_valid"] == 0                       # converged, then rejected by the 0.05 gate
        g = fr3_fusion(oracle_lib, icp_cov_thresh=0.1)
        for _, rgb, depth in pair:
            r2 = g.process_frame(rgb, depth)
        assert r2["icp_valid"] == 1 and np.linalg.norm(r2["pose"][9:]) < 0.2    # accepted once the gate lets it through (people walking through the view pull it: up to 8 cm here)


@pytest.mark.gpu
def test_fr3_walking_frames_bit_exact_on_gpu(oracle_lib, product_lib):
    frames = list(replay.frames_from_npz(FR3))
    for cov in (0.05, 0.1):
        fo, fh = fr3_fusion(oracle_lib, icp_cov_thresh=cov), fr3_fusion(product_lib, icp_cov_thresh=cov)
        for _, rgb, depth in frames[:2]:
            util.same_result(fo.process_frame(rgb, depth), fh.process_frame(rgb, depth))
        util.compare_state(fo, fh)


# ---- CUDA's approximate rsqrtf (round 5) ---------------------------------------------------------------------------------------
# The reference's normalize() is v * rsqrtf(dot(v, v)) (vector_math.cuh:247-252): a device intrinsic within 2 ulp of the exact value,
# built with --use_fast_math on top.  This build specifies the correctly rounded 1 / sqrt on both sides, and the generator that pins
# eigenDecomposition to the reference's own text (oracle/ref_decision_vectors.cpp) had to define rsqrtf the same way -- the one named
# stand-in of the parity chain.  The eigen-frames and the rotated normals that come out of normalize() feed integer decisions: the
# |n.n| > 0.8 gates of ICP and association (dense_registration_kernels.cuh:234, supersurfel_fusion_kernels.cu:586), and through the
# pose everything after them.  What cannot be observed (CUDA's rounding) can be bounded: the oracle's hook moves EVERY reciprocal
# square root by -2 .. +2 ulp (or by a pseudo-random amount in that range) and the sequences are replayed.
def _replayed_decisions(lib, make_fusion, frames, mode):
    lib.lib.ssf_oracle_set_rsqrt_ulp.restype = int
    assert lib.lib.ssf_oracle_set_rsqrt_ulp(mode) == mode
    try:
        f = make_fusion()
        ints, poses = [], []
        for fr in frames:
            r = f.process_frame(fr[0], fr[1])
            m = f.get_model()
            ints.append(dict(result=[int(r[k]) for k in util.RESULT_KEYS], labels=f.index_map().copy(), inliers=f.inlier_map().copy(),
                             frame_valid=(f.get_frame()["confidences"] > 0).copy(), stamps=m["stamps"].copy(), model_valid=(m["confidences"] > 0).copy(),
                             conf=m["confidences"].copy()))
            poses.append(np.asarray(r["pose"], np.float64).copy())
        return ints, poses
    finally:
        lib.lib.ssf_oracle_set_rsqrt_ulp(0)


def _count_flips(a, b):
    flips, total = 0, 0
    for fa, fb in zip(a, b):
        for k in ("result", "labels", "inliers", "frame_valid", "stamps", "model_valid", "conf"):
            va, vb = np.asarray(fa[k]), np.asarray(fb[k])
            if va.shape != vb.shape:
                flips += max(va.size, vb.size); total += max(va.size, vb.size)
            else:
                flips += int((va != vb).sum()); total += va.size
    return flips, total


@pytest.mark.parametrize("sequence", ["tum_fr1_xyz", "synthetic_orbit"])
def test_a_2_ulp_reciprocal_square_root_flips_no_integer_decision(sequence, oracle_lib):
    """Every normalize() of the oracle perturbed by -2, -1, +1, +2 ulp and pseudo-randomly within +-2 ulp, over the 8 committed real
    fr1_xyz frames (pre-filter on, launch-file parameters) and over 6 synthetic orbit frames against a seeded 20 k-row map: every
    integer the path decides -- ICP accepted / iterations, the five counters, label and inlier maps, which frame supersurfels are
    valid, the stamps of every model row (who was matched when), which rows survive, the confidences (integer pixel counts summed
    by the updates) -- is IDENTICAL to the unperturbed run (0 of ~5 M values per arm), and the poses move by less than 1e-5 (north_star's budget: 1e-4).  So the
    parity chain does not hang on how CUDA rounds rsqrtf."""
    if sequence == "tum_fr1_xyz":
        frames = [(rgb, depth) for _, rgb, depth in tum_frames(8)]
        make = lambda: tum_fusion(oracle_lib, nb_supersurfels_max=100000)      # noqa: E731
    else:
        W, H = 320, 240
        frames = [util.frame(k, W, H, noise=True, holes=0.03) for k in range(6)]
        model, nvis = synthetic.seed_model_cam0(20000, W, H, stamp=30)

        def make():
            f = binding.Fusion(oracle_lib, util.make_cfg(oracle_lib, W, H, nb_supersurfels_max=40000))
            f.set_model(model, nvis, 30)
            return f
    base_i, base_p = _replayed_decisions(oracle_lib, make, frames, 0)
    assert sum(r["result"][0] for r in base_i) >= len(frames) - 2, "the ICP must be accepted on these frames for the gates to matter"
    report = {}
    for mode in (-2, -1, 1, 2, 3):
        ints, poses = _replayed_decisions(oracle_lib, make, frames, mode)
        flips, total = _count_flips(base_i, ints)
        drift = max(float(np.abs(p - q).max()) for p, q in zip(base_p, poses))
        report[mode] = (flips, total, drift)
    assert all(v[0] == 0 for v in report.values()), report
    assert all(v[2] < 1e-5 for v in report.values()), report          # (measured: <= 1.6e-6 over 8 real frames)
    assert any(v[2] > 0 for v in report.values()), "the hook did not reach the pose at all"


# ---- the stated tolerance against the CUDA path (round 6) ------------------------------------------------------------------------
# north_star: "match the reference CUDA path ... within a stated float tolerance (bit-exact for surfel indexing/assignment)".  The
# product is bit-equal to the oracle (the specification); the reference itself is --use_fast_math and racy, i.e. a FAMILY of executions
# none of which can be produced here.  tools/cuda_tolerance_study.py measures the width of that family around the specification with
# the study build of the oracle (oracle/Makefile `arms`); profiles/cuda_tolerance_r06.txt is its full table (8 real + 6 synthetic
# frames) and DESIGN.md section 2 states the tolerance.  Here: the quick form (4 + 3 frames), asserting the stated bounds.
@pytest.fixture(scope="module")
def tolerance_study():
    import subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    subprocess.check_call(["make", "-C", os.path.join(root, "oracle"), "arms"], stdout=subprocess.DEVNULL)
    sys.path.insert(0, os.path.join(root, "tools"))
    import cuda_tolerance_study as study
    import io
    return study.study(quick=True, out=io.StringIO())


def test_stated_tolerance_arithmetic_arms(tolerance_study):
    """--use_fast_math's arithmetic (FMA contraction, flush-to-zero, every division / sqrt / rsqrt / powf within its documented
    error bound): at most 0.05 % of the labels and inlier flags of a frame move, no frame supersurfel's validity flips on the synthetic
    frames, the pose stays within 5e-4 (m / rotation entry) of the specification's over the sequence; flush-to-zero, the tie rule of
    the arg-min, the insertion order (as a multiset of rows) and powf / cbrtf within 8 ulp change NOTHING."""
    r = tolerance_study
    for seq in ("tum_fr1_xyz", "synthetic"):
        for arm in ("fma", "div-2", "div+2", "div~2", "rsqrt~2", "pow-2", "pow+2", "pow~2", "pow+8", "ftz"):
            d = r[(seq, arm)]
            assert d["labels"] <= 5e-4 * d["px"] and d["inliers"] <= 5e-4 * d["px"], (seq, arm, d)
            assert d["pose_t"] <= 5e-4 and d["pose_r"] <= 1e-3, (seq, arm, d)
        for arm in ("ftz", "pow-2", "pow+2", "pow~2", "pow+8", "tie-high", "insert-rev"):
            d = r[(seq, arm)]
            assert d["labels"] == 0 and d["inliers"] == 0 and d["fvalid"] == 0 and d["icp"] == 0 and d["counters"] == 0 and d["row_diff"] == 0, (seq, arm, d)
            assert d["pose_t"] == 0.0 and d["pose_r"] == 0.0, (seq, arm, d)
        d = r[(seq, "rsqrt~2")]
        assert d["labels"] == 0 and d["counters"] == 0 and d["row_diff"] == 0 and d["pose_t"] < 1e-5, (seq, d)
    assert any(r[(s, a)]["labels"] > 0 for s in ("tum_fr1_xyz", "synthetic") for a in ("fma", "div~2")), "the arithmetic arms reached nothing"


def test_stated_tolerance_schedule_arms(tolerance_study):
    """The other extreme of each race the reference leaves open (blocks of a relabelling pass strictly in sequence; the plane filter in
    place; the torn arg-min's worst outcome), another random stream for the RANSAC initialisation (cuRAND's cannot be reproduced), and
    the races + arithmetic at once: at most 2 % of the labels, 1 % of the inlier flags, the pose within 1e-2.  This -- not 1e-4 -- is how far two valid executions of the reference's own CUDA code can lie apart; the product's
    1e-4 (in fact 0-bit) agreement is with the specification, one member of that family."""
    r = tolerance_study
    for seq in ("tum_fr1_xyz", "synthetic"):
        for arm in ("schedule", "filter-gs", "argmin-torn", "rng-stream", "all"):
            d = r[(seq, arm)]
            assert d["labels"] <= 0.02 * d["px"] and d["inliers"] <= 0.01 * d["px"], (seq, arm, d)
            assert d["pose_t"] <= 1e-2 and d["pose_r"] <= 2e-2, (seq, arm, d)
        assert r[(seq, "rng-stream")]["labels"] > 0           # (another RANSAC stream is the largest single source of differing labels)
        assert r[(seq, "schedule")]["labels"] > 0 and r[(seq, "filter-gs")]["pose_t"] > 0 and r[(seq, "argmin-torn")]["row_diff"] > 0
