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


@pytest.mark.parametrize("world,pipelined", [(2, False), (2, True), (3, False), (4, True)])
def test_ranks_in_separate_processes_bit_exact(world, pipelined, oracle_lib, tmp_path):
    W, H, nf = 320, 240, 6
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    procs = [subprocess.Popen([sys.executable, os.path.join(ROOT, "tests", "p2p_worker.py"), str(r), str(world), str(tmp_path), str(W), str(H),
                               str(nf), "1" if pipelined else "0"], env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True)
             for r in range(world)]
    logs = []
    for p in procs:
        try:
            logs.append(p.communicate(timeout=300)[0])
        except subprocess.TimeoutExpired:
            p.kill()
            logs.append(p.communicate()[0])
    assert all(p.returncode == 0 for p in procs), "\n".join(l[-3000:] for l in logs)
    outs = [np.load(os.path.join(str(tmp_path), "out%d.npz" % r)) for r in range(world)]
    fo, oo = _emulated_ranks(oracle_lib, world, W, H, nf)
    per_rank = [(o["poses"], o["counts"]) for o in outs]
    models = [{k[len("model_"):]: o[k] for k in o.files if k.startswith("model_")} for o in outs]
    check_against_oracle(world, per_rank, models, oo, fo)
    for o in outs:
        assert [int(v) for v in o["global_counts"][:2]] == [int(oo[-1][1][:, 0].sum()), int(oo[-1][1][:, 1].sum())]


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
