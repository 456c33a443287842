"""How far can a CUDA execution of the reference lie from this build's specification?  (CPU only; test infrastructure.)

    python tools/cuda_tolerance_study.py [--quick] > profiles/cuda_tolerance_r06.txt

north_star asks for agreement with "the reference CUDA path ... within a stated float tolerance (bit-exact for surfel
indexing/assignment)".  The reference cannot be executed here, is compiled --use_fast_math (/root/reference/cmake/UseCUDA.cmake:15:
FMA contraction, approximate division / sqrt / rsqrt / powf / cbrtf, flush-to-zero) and is racy by construction (cross-tile label
race TPS_RGBD_kernels.cuh:272-292,439; in-place plane filter TPS_RGBD_kernels.cu:585-612; torn arg-min supersurfel_fusion_kernels.cu:590-594;
atomic insertion order :376): "the CUDA path" is a family of executions.  The product is bit-equal to ONE member of a closely related
family (IEEE arithmetic, the double-buffered schedule, exact arg-min, ordered insertion: oracle/).  This tool measures the width of
the family around that member: the study build of the oracle (oracle/Makefile `arms`: the same sources with run-time arms, and once
more with FMA contraction) replays the committed real frames (8 x TUM fr1_xyz, benchmark-launch parameters, pre-filter on) and a
synthetic orbit against a seeded map, one arm at a time and all of them together, and reports per arm how many of the path's integer
decisions differ from the specification's and how far the pose moves.

The arms are bounds, not the reference: each replaces an unobservable CUDA behaviour by the worst case of its documented error bound
(CUDA C Programming Guide, "Intrinsic Functions": __fdividef / rsqrtf / sqrt.approx within 2 ulp; powf: up to 8 ulp in the guide's table; its arm moves nothing at 2 ulp and is reported at 8 too) or by the
opposite extreme of a race."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import ctypes as C            # noqa: E402
import numpy as np            # noqa: E402
from supersurfel_fusion_amd import binding, replay, synthetic      # noqa: E402

BUILD = os.path.join(ROOT, "oracle", "_build")
TUM = os.path.join(ROOT, "tests", "golden", "tum_fr1_xyz_8frames.npz")
RESULT_KEYS = ("icp_valid", "icp_iters", "n_model", "n_visible", "n_removed", "n_inserted", "n_updated", "stamp")

ARMS = [
    # name, library, {arm: value}, what it stands for
    ("fma", "fma", {}, "FMA contraction of the device code (--use_fast_math implies -fmad=true)"),
    ("ftz", "std", {"ftz": 1}, "flush-to-zero + denormals-are-zero (-ftz=true)"),
    ("div-2", "std", {"div_ulp": -2}, "every device float division and sqrt 2 ulp low (-prec-div=false, -prec-sqrt=false: <= 2 ulp)"),
    ("div+2", "std", {"div_ulp": 2}, "... 2 ulp high"),
    ("div~2", "std", {"div_ulp": 3}, "... pseudo-random in [-2, 2] ulp"),
    ("rsqrt~2", "std", {"rsqrt_ulp": 3}, "every normalize()'s rsqrtf pseudo-random in [-2, 2] ulp (round 5's study)"),
    ("pow-2", "std", {"pow_ulp": -2}, "powf / cbrtf of the Lab conversions 2 ulp low"),
    ("pow+2", "std", {"pow_ulp": 2}, "... 2 ulp high"),
    ("pow~2", "std", {"pow_ulp": 3}, "... pseudo-random in [-2, 2] ulp"),
    ("pow+8", "std", {"pow_ulp": 8}, "... 8 ulp high (the guide's bound for __powf is looser than 2 ulp)"),
    ("schedule", "std", {"schedule": 1}, "relabelling: the reference's 32x32 blocks strictly one after the other, in place (the other extreme of the cross-tile race)"),
    ("filter-gs", "std", {"filter_gs": 1}, "plane filter in place in node order (Gauss-Seidel: the in-place outcome of TPS_RGBD_kernels.cu:585-612)"),
    ("tie-high", "std", {"tie": 1}, "association: equal distances go to the highest model id"),
    ("argmin-torn", "std", {"tie": 2}, "association: every candidate saw the initial 0.05, the last store stays (worst valid outcome of the torn arg-min)"),
    ("insert-rev", "std", {"insert_rev": 1}, "insertion in descending frame id (another atomic arrival order)"),
    ("rng-stream", "std", {"rng_seed": 4321}, "another random stream for the RANSAC plane initialisation (the reference's cuRAND XORWOW sequence -- curand_init(1234, id, 0), "
                                              "TPS_RGBD_kernels.cu:321 -- cannot be reproduced without the toolkit's skip-ahead tables: this build's counter-based "
                                              "generator is part of its specification, and this arm is what ANY other stream does)"),
    ("all", "fma", {"ftz": 1, "div_ulp": 3, "rsqrt_ulp": 3, "pow_ulp": 3, "schedule": 1, "filter_gs": 1, "tie": 1},
     "everything above at once except argmin-torn and insert-rev"),
]


def load(tag):
    lib = binding.Library(os.path.join(BUILD, "libssf_oracle_arms%s.so" % ("_fma" if tag == "fma" else "")))
    lib.lib.ssf_oracle_set_arm.restype = C.c_int
    lib.lib.ssf_oracle_set_arm.argtypes = [C.c_char_p, C.c_int]
    lib.lib.ssf_oracle_set_rsqrt_ulp.restype = C.c_int
    return lib


def set_arms(lib, arms):
    for n in ("div_ulp", "pow_ulp", "schedule", "tie", "insert_rev", "filter_gs", "ftz"):
        assert lib.lib.ssf_oracle_set_arm(n.encode(), int(arms.get(n, 0))) == 0
    lib.lib.ssf_oracle_set_rsqrt_ulp(int(arms.get("rsqrt_ulp", 0)))


def run(lib, make, frames, arms):
    set_arms(lib, arms)
    try:
        f = make(lib, **({"rng_seed": int(arms["rng_seed"])} if "rng_seed" in arms else {}))
        out = []
        for rgb, depth in frames:
            r = f.process_frame(rgb, depth)
            m = f.get_model()
            fr = f.get_frame()
            out.append(dict(result={k: int(r[k]) for k in RESULT_KEYS}, pose=np.asarray(r["pose"], np.float64).copy(),
                            labels=f.index_map().copy(), inliers=f.inlier_map().copy(),
                            frame_valid=(fr["confidences"] > 0).copy(), frame_pos=fr["positions"].copy(),
                            stamps=m["stamps"].copy(), conf=m["confidences"].copy(), pos=m["positions"].copy()))
        return out
    finally:
        set_arms(lib, {})


def compare(base, arm):
    """Differences of an arm's run from the specification's, over the frames of a sequence."""
    d = dict(labels=0, px=0, inliers=0, fvalid=0, S=0, icp=0, counters=0, frames=len(base), rows=0, row_diff=0, n_model=0,
             pose_t=0.0, pose_r=0.0, fpos=0.0)
    for a, b in zip(base, arm):
        d["labels"] += int((a["labels"] != b["labels"]).sum()); d["px"] += a["labels"].size
        d["inliers"] += int((a["inliers"] != b["inliers"]).sum())
        d["fvalid"] += int((a["frame_valid"] != b["frame_valid"]).sum()); d["S"] += a["frame_valid"].size
        both = a["frame_valid"] & b["frame_valid"]
        if both.any():
            d["fpos"] = max(d["fpos"], float(np.abs(a["frame_pos"][both] - b["frame_pos"][both]).max()))
        d["icp"] += int(a["result"]["icp_valid"] != b["result"]["icp_valid"]) + int(a["result"]["icp_iters"] != b["result"]["icp_iters"])
        d["counters"] += sum(int(a["result"][k] != b["result"][k]) for k in ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated"))
        d["n_model"] = max(d["n_model"], abs(a["result"]["n_model"] - b["result"]["n_model"]))
        pa, pb = a["pose"].reshape(3, 4), b["pose"].reshape(3, 4)
        d["pose_t"] = max(d["pose_t"], float(np.abs(pa[:, 3] - pb[:, 3]).max()))
        d["pose_r"] = max(d["pose_r"], float(np.abs(pa[:, :3] - pb[:, :3]).max()))
    # the final map as a MULTISET of (t_init, t_last, summed pixel count) rows -- who was created when, matched when, fused with how
    # many pixels: rows that are in one run's map and not in the other's (row-for-row comparison would count every row behind
    # one extra insertion as different)
    a, b = base[-1], arm[-1]
    ka = np.concatenate([a["stamps"].reshape(len(a["conf"]), -1).astype(np.int64), a["conf"].astype(np.int64)[:, None]], axis=1)
    kb = np.concatenate([b["stamps"].reshape(len(b["conf"]), -1).astype(np.int64), b["conf"].astype(np.int64)[:, None]], axis=1)
    ua, ca = np.unique(ka, axis=0, return_counts=True)
    ub, cb = np.unique(kb, axis=0, return_counts=True)
    da = {tuple(r): int(c) for r, c in zip(ua.tolist(), ca.tolist())}
    db = {tuple(r): int(c) for r, c in zip(ub.tolist(), cb.tolist())}
    d["rows"] = max(len(ka), len(kb))
    d["row_diff"] = sum(abs(da.get(k, 0) - db.get(k, 0)) for k in set(da) | set(db))
    return d


def fmt(name, d):
    return ("%-12s labels %8.4f %%  inliers %8.4f %%  frame-valid %3d/%d  icp-flags %d  counters %2d (|dN| <= %d)  map-rows %6d/%d (%.3f %%)  "
            "frame-pos <= %.2e m  pose: t <= %.2e m, R <= %.2e" %
            (name, 100.0 * d["labels"] / d["px"], 100.0 * d["inliers"] / d["px"], d["fvalid"], d["S"], d["icp"], d["counters"], d["n_model"],
             d["row_diff"], d["rows"], 100.0 * d["row_diff"] / max(d["rows"], 1), d["fpos"], d["pose_t"], d["pose_r"]))


def sequences(quick):
    import util
    frames_tum = [(rgb, depth) for _, rgb, depth in list(replay.frames_from_npz(TUM))[: (4 if quick else 8)]]

    def make_tum(lib, **kw):
        cfg = dict(replay.BENCHMARK_LAUNCH, nb_supersurfels_max=100000); cfg.update(kw)
        return binding.Fusion(lib, lib.default_config(**cfg))
    W, H = 320, 240
    frames_syn = [util.frame(k, W, H, noise=True, holes=0.03) for k in range(3 if quick else 6)]
    model, nvis = synthetic.seed_model_cam0(20000, W, H, stamp=30)

    def make_syn(lib, **kw):
        f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=40000, **kw))
        f.set_model(model, nvis, 30)
        return f
    return [("tum_fr1_xyz (8 real frames, 640x480, benchmark-launch parameters, pre-filter on, map grown by the frames)", make_tum, frames_tum),
            ("synthetic orbit (6 frames, 320x240, noise + holes, against a seeded 20 k-row map)", make_syn, frames_syn)]


def study(quick=False, arms=None, out=sys.stdout):
    libs = {"std": load("std"), "fma": load("fma")}
    plain = binding.Library(os.path.join(BUILD, "libssf_oracle.so"))
    results = {}
    for title, make, frames in sequences(quick):
        print("== %s" % title, file=out)
        t0 = time.time()
        base = run(libs["std"], make, frames, {})
        # the study build with every arm off IS the checker: same bits
        f = make(plain)
        for k, (rgb, depth) in enumerate(frames):
            r = f.process_frame(rgb, depth)
            assert np.array_equal(np.asarray(r["pose"], np.float64), base[k]["pose"]), "study build (arms off) differs from the checker"
            assert np.array_equal(f.index_map(), base[k]["labels"])
        print("   (study build with every arm off == the checker, bit for bit; %d frames, %.0f s)" % (len(frames), time.time() - t0), file=out)
        for name, tag, a, _ in ARMS:
            if arms and name not in arms:
                continue
            d = compare(base, run(libs[tag], make, frames, a))
            results[(title.split(" ")[0], name)] = d
            print("   " + fmt(name, d), file=out)
            out.flush()
    print("\narms:", file=out)
    for name, tag, a, what in ARMS:
        print("   %-12s %s" % (name, what), file=out)
    return results


if __name__ == "__main__":
    study(quick="--quick" in sys.argv)
