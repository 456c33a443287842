"""Generates the trajectory fixtures that anchor the oracle to the one verification artefact the reference
commits for this path: rgbd_benchmark/rgbd_dataset_freiburg1_xyz/{estimated.txt, associations_with_gt.txt}.

Run in the BUILD container only (it reads /root/reference, which does not exist on the GPU box):

    python tests/golden/make_fr1_xyz_trajectory.py [--frames N]

What it does -- exactly the call pattern of SupersurfelFusionRGBDBenchmarkNode::run
(node/supersurfel_fusion_rgbd_benchmark_node.cpp:573-744) through supersurfel_fusion_amd/replay.py, on the CPU
oracle (OpenMP build: same bits as the single-threaded checker), with the rgbd_benchmark launch parameters
(launch/supersurfel_fusion_rgbd_benchmark.launch: SURVEY.md Appendix B), the depth pre-filter ON as
processFrame has it (supersurfel_fusion.cu:180), and the out-of-scope subsystems off (no sparse-VO prior: the
pose prior is the previous pose; no MOD mask; no loop closure):

  tests/golden/fr1_xyz_oracle_estimated.txt    the oracle's trajectory, TUM format, one line per association
  tests/golden/fr1_xyz_gt.txt                  `stamp tx ty tz qx qy qz qw` ground truth of every association line
                                               (columns 5-12 of associations_with_gt.txt: TUM RGB-D dataset, CC BY 4.0)
  tests/golden/fr1_xyz_reference_estimated.txt the reference's own committed whole-system output (data, 790 poses)
  tests/golden/fr1_xyz_ate.json                ATE (Horn-aligned RMSE) of both against the ground truth
  tests/golden/tum_fr1_xyz_8frames.npz         the first 8 decoded frames (the GPU replay test runs them)
"""
import argparse
import json
import os
import subprocess
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
from supersurfel_fusion_amd import binding, replay  # noqa: E402

DATASET = "/root/reference/rgbd_benchmark/rgbd_dataset_freiburg1_xyz"
GOLD = os.path.join(ROOT, "tests", "golden")
LAUNCH = replay.BENCHMARK_LAUNCH


def read_xyz(path):
    rows = [l.split() for l in open(path) if l.strip() and not l.startswith("#")]
    return [r[0] for r in rows], np.array([[float(v) for v in r[1:4]] for r in rows])


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--frames", type=int, default=None)
    a = ap.parse_args()
    subprocess.check_call(["make", "-C", os.path.join(ROOT, "oracle"), "omp"], stdout=subprocess.DEVNULL)
    lib = binding.Library(os.path.join(ROOT, "oracle", "_build", "libssf_oracle_omp.so"))
    f = binding.Fusion(lib, lib.default_config(**LAUNCH))
    assoc = os.path.join(DATASET, "associations_with_gt.txt")
    ent = replay.read_associations(assoc, a.frames)
    t0 = time.time()
    lines, res = replay.replay(f, replay.frames_from_dataset(DATASET, 0.0002, a.frames), os.path.join(GOLD, "fr1_xyz_oracle_estimated.txt"))
    dt = time.time() - t0
    with open(os.path.join(GOLD, "fr1_xyz_gt.txt"), "w") as g:
        for e in ent:
            g.write(" ".join([e["stamp"]] + ["%.4f" % v for v in list(e["gt"][0]) + list(e["gt"][1])]) + "\n")
    ref_lines = open(os.path.join(DATASET, "estimated.txt")).read()
    open(os.path.join(GOLD, "fr1_xyz_reference_estimated.txt"), "w").write(ref_lines)
    if a.frames is None or a.frames >= 8:
        replay.pack_frames(DATASET, assoc, os.path.join(GOLD, "tum_fr1_xyz_8frames.npz"), 8)
    gt = np.array([e["gt"][0] for e in ent])
    est = np.array([r["pose"][9:] for r in res], np.float64)
    _, ref_xyz = read_xyz(os.path.join(DATASET, "estimated.txt"))
    n = min(len(gt), len(ref_xyz))
    rep = dict(frames=len(lines), oracle_seconds=round(dt, 1),
               icp_valid_frames=int(sum(r["icp_valid"] for r in res)), icp_iters_mean=float(np.mean([r["icp_iters"] for r in res])),
               n_model_last=int(res[-1]["n_model"]),
               ate_rmse_oracle=replay.ate_rmse(est, gt), ate_rmse_reference_estimated=replay.ate_rmse(ref_xyz[:n], gt[:n]),
               ate_rmse_oracle_vs_reference_estimated=replay.ate_rmse(est[:n], ref_xyz[:n]),
               path_length_gt=float(np.linalg.norm(np.diff(gt, axis=0), axis=1).sum()),
               parameters=LAUNCH, note="hot path only: no sparse-VO prior, no MOD mask, no loop closure; the reference's "
                                       "estimated.txt is a whole-system output with an unknown parameter set (SURVEY.md section 4)")
    json.dump(rep, open(os.path.join(GOLD, "fr1_xyz_ate.json"), "w"), indent=1)
    print(json.dumps({k: v for k, v in rep.items() if k != "parameters"}, indent=1))


if __name__ == "__main__":
    main()
