"""Native RCCL exchange path (ssf_comm_attach) on one GPU: a one-rank communicator runs every
collective of the N > 1 protocol (self-reduction), and the results must equal the plain
single-shard run bit for bit.  The multi-rank protocol itself is pinned on CPU (test_sharded.py,
gloo, world 2 and 3) through the stage seams that both drivers share."""
import os

import numpy as np
import pytest

import util
from supersurfel_fusion_amd import binding

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def one_rank_group():
    import torch
    import torch.distributed as dist
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29541")
    created = False
    if not dist.is_initialized():
        torch.cuda.set_device(0)
        try:
            dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda", 0))
        except Exception as e:                      # no usable RCCL / rendezvous on this box: nothing to exercise
            pytest.skip("torch.distributed nccl group unavailable: %s" % e)
        created = True
    yield dist
    if created:
        dist.destroy_process_group()


@pytest.mark.parametrize("depth_ahead,batch", [(0, 1), (2, 4)])
def test_one_rank_communicator_equals_plain_run(depth_ahead, batch, one_rank_group, product_lib):
    W, H, nf = 320, 240, 7
    plain = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H))
    comm = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=depth_ahead, extract_batch=batch))
    assert plain.comm_info() == dict(backend="none", ranks=1, rank=0)
    comm.comm_attach()
    assert comm.comm_info() == dict(backend="rccl", ranks=1, rank=0)      # what ncclCommCount / ncclCommUserRank say (bench.py prints it)
    frames = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(nf)]
    want = [plain.process_frame(*fr) for fr in frames]
    got, nsub = [], 0
    for k in range(nf):
        while nsub < nf and comm.can_submit():
            comm.submit_frame(*frames[nsub]); nsub += 1
        got.append(comm.process_submitted().as_dict())
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(plain, comm)
    g = comm.global_counts()
    assert g["n_model"] == want[-1]["n_model"] and g["n_visible"] == want[-1]["n_visible"]
    assert g["n_inserted"] == want[-1]["n_inserted"] and g["n_updated"] == want[-1]["n_updated"]


@pytest.mark.parametrize("depth_ahead,batch,mode", [(0, 1, 2), (2, 4, 2), (2, 4, 1)])
def test_dealt_extract_on_a_one_rank_communicator(depth_ahead, batch, mode, one_rank_group, product_lib):
    """ssf_comm_deal_extract on a one-rank communicator: every batch is this rank's, the broadcasts of its frames' tables run (on the
    batch context's own communicator and stream, ncclCommSplit) as self-broadcasts, and in mode 2 the rank REBUILDS its private
    tables (k_import_frame) from the wire buffers it has just shipped (k_export_rows) -- what a receiving rank does -- before the
    track chain reads them: results and state must equal the plain run's bit for bit.  (Two ranks of one RCCL communicator cannot
    share this box's one GPU; the receive path across libraries and emulated ranks is test_parity_gpu.py's.)"""
    W, H, nf = 320, 240, 9
    plain = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H))
    comm = binding.Fusion(product_lib, util.make_cfg(product_lib, W, H, pipeline_depth=depth_ahead, extract_batch=batch))
    comm.comm_attach()
    comm.comm_deal_extract(mode)
    frames = [util.frame(k, W, H, noise=True, holes=0.02) for k in range(nf)]
    want = [plain.process_frame(*fr) for fr in frames]
    got, nsub = [], 0
    for k in range(nf):
        while nsub < nf and comm.can_submit():
            comm.submit_frame(*frames[nsub]); nsub += 1
        got.append(comm.process_submitted().as_dict())
    for a, b in zip(want, got):
        util.same_result(a, b)
    util.compare_state(plain, comm)
