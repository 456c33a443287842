"""Multi-GPU driver: one process per GPU, the map sharded by world-space tile across ranks.

Per frame every rank runs the extract stage on the (replicated) frame -- it is deterministic, so
all ranks hold identical frame tables and no broadcast is needed -- then the stages whose unit of
work is a model supersurfel run on the local shard with three real exchange steps:

  ICP         SUM all-reduce of the 29-value int64 fixed-point record per iteration
              (exact: integer addition is associative, so 1/2/4/8 ranks give identical bits)
  association MIN all-reduce of S packed (dist_bits<<32 | global id) keys + MAX of the S matched
              bytes; the shard that owns the winner applies the update
  counts      all-gather of (n_model, n_visible) to form global id offsets

Insertion is decided locally: the owner of a new supersurfel is a pure function of its world tile
(ssf_stage_fuse, shard_owner), evaluated identically on every rank.

The collectives go through torch.distributed: backend "nccl" (= RCCL over xGMI) with device
tensors on the GPU box, "gloo" with CPU tensors in the CPU tests.  The engine is any Fusion
(binding.py); the product path constructs it from load_product() and never touches the oracle.
"""
import numpy as np
import torch
import torch.distributed as dist


class ShardedFusion:
    def __init__(self, fusion, device=None, group=None):
        self.f = fusion
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert fusion.cfg.nranks == self.world and fusion.cfg.rank == self.rank, \
            "ssf_config.rank/nranks must match the process group"
        self.device = device if device is not None else torch.device("cpu")
        self._counts = torch.zeros(2, dtype=torch.int64, device=self.device)
        self._all = [torch.zeros(2, dtype=torch.int64, device=self.device) for _ in range(self.world)]
        self._icp = torch.zeros(29, dtype=torch.int64, device=self.device)
        S = fusion.S
        # uint64 MIN is not available on every backend: the key is < 2^63 or the all-ones sentinel,
        # so it is exchanged as int64 with the sentinel mapped to INT64_MAX (order preserved).
        self._best = torch.zeros(S, dtype=torch.int64, device=self.device)
        self._matched = torch.zeros(S, dtype=torch.uint8, device=self.device)

    def _sum(self, t):
        if self.world > 1:
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=self.group)

    def process_frame(self, rgb, depth, prior_pose=None, dynamic_mask=None, on_device=False):
        f = self.f
        f.stage_extract(rgb, depth, dynamic_mask, on_device=on_device)
        c = f.counts()
        self._counts[0], self._counts[1] = c["n_model"], c["n_visible"]
        if self.world > 1:
            dist.all_gather(self._all, self._counts, group=self.group)
            allc = torch.stack(self._all).cpu().numpy()
        else:
            allc = self._counts.cpu().numpy()[None]
        g_model, g_vis = int(allc[:, 0].sum()), int(allc[:, 1].sum())
        id_offset = int(allc[:self.rank, 1].sum())
        f.set_shard(id_offset, g_model, g_vis)
        # ---- ICP ----
        f.icp_begin(prior_pose)
        again = g_vis > 0 and f.cfg.icp_iter > 0
        iters = 0
        while again:
            sums = f.icp_accumulate()
            self._icp.copy_(torch.from_numpy(sums))
            self._sum(self._icp)
            again = f.icp_update(self._icp.cpu().numpy())
            iters += 1
        valid = f.icp_end()
        # ---- association ----
        best, matched = f.match()
        if self.world > 1:
            key = best.view(np.int64).copy()
            key[best == np.uint64(0xFFFFFFFFFFFFFFFF)] = np.iinfo(np.int64).max
            self._best.copy_(torch.from_numpy(key))
            self._matched.copy_(torch.from_numpy(matched))
            dist.all_reduce(self._best, op=dist.ReduceOp.MIN, group=self.group)
            dist.all_reduce(self._matched, op=dist.ReduceOp.MAX, group=self.group)
            key = self._best.cpu().numpy()
            best = key.view(np.uint64).copy()
            best[key == np.iinfo(np.int64).max] = np.uint64(0xFFFFFFFFFFFFFFFF)
            matched = self._matched.cpu().numpy()
        res = f.fuse(best, matched)
        res["icp_valid"], res["icp_iters"] = int(valid), iters
        tot = torch.tensor([res["n_model"], res["n_visible"], res["n_removed"], res["n_inserted"], res["n_updated"]],
                           dtype=torch.int64, device=self.device)
        self._sum(tot)
        tot = tot.cpu().numpy()
        res.update(global_n_model=int(tot[0]), global_n_visible=int(tot[1]), global_n_removed=int(tot[2]),
                   global_n_inserted=int(tot[3]), global_n_updated=int(tot[4]))
        return res
