"""One rank of the cross-process peer-to-peer test (tests/test_p2p_gpu.py): its own process, its own HIP context, the
exchange regions of the other ranks opened through their IPC handles.

    python tests/p2p_worker.py <rank> <world> <dir> <W> <H> <frames> <pipelined>

Handles are traded through files in <dir> (h<rank>.bin); the results go to <dir>/out<rank>.npz."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import util  # noqa: E402
from supersurfel_fusion_amd import binding  # noqa: E402


def main():
    rank, world, d, W, H, nf, pipelined = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], int(sys.argv[4]), int(sys.argv[5]), int(sys.argv[6]), int(sys.argv[7])
    lib = binding.load_product()
    kw = dict(pipeline_depth=2, extract_batch=2) if pipelined else {}
    f = binding.Fusion(lib, util.make_cfg(lib, W, H, nb_supersurfels_max=4096, rank=rank, nranks=world, shard_tile=0.25, **kw))
    f.p2p_configure(all_ranks_on_this_device=True)             # (the ranks are processes that share the box's one GPU)
    mine = f.p2p_export()
    tmp = os.path.join(d, "h%d.tmp" % rank)
    mine.tofile(tmp)
    os.replace(tmp, os.path.join(d, "h%d.bin" % rank))
    t0 = time.time()
    handles = []
    for r in range(world):
        p = os.path.join(d, "h%d.bin" % r)
        while not os.path.exists(p):
            assert time.time() - t0 < 120, "rank %d never exported its handle" % r
            time.sleep(0.01)
        handles.append(np.fromfile(p, np.uint8))
    f.p2p_attach(np.concatenate(handles))
    frames = [util.frame(k, W, H) for k in range(nf)]
    if pipelined:
        frames = [(np.ascontiguousarray(r), np.ascontiguousarray(dd)) for r, dd in frames]
        res = f.process_sequence([r.ctypes.data for r, _ in frames], [dd.ctypes.data for _, dd in frames], on_device=False)
    else:
        res = [f.process_frame(r, dd) for r, dd in frames]
    g = f.global_counts()
    m = f.get_model()
    np.savez(os.path.join(d, "out%d.npz" % rank), poses=np.stack([r["pose"] for r in res]),
             counts=np.array([[r[k] for k in ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")] for r in res], np.int64),
             iters=np.array([r["icp_iters"] for r in res]), valid=np.array([r["icp_valid"] for r in res]),
             global_counts=np.array([g[k] for k in ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")], np.int64),
             **{"model_" + k: v for k, v in m.items()})


if __name__ == "__main__":
    main()
