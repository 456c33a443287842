"""Multi-GPU driver: one process per GPU, the map sharded by world-space tile across ranks.

Per frame every rank runs the extract stage on the (replicated) frame -- it is deterministic, so
all ranks hold identical frame tables and no broadcast is needed -- then the stages whose unit of
work is a model supersurfel run on the local shard with three real exchange steps:

  ICP         SUM all-reduce of the 29-value int64 fixed-point record per iteration
              (exact: integer addition is associative, so 1/2/4/8 ranks give identical bits)
  association MIN all-reduce of S packed (dist_bits<<32 | global id) keys + MAX of the S matched
              bytes; the shard that owns the winner applies the update
  migration   ("halo exchange") SUM all-reduce of the migrant table (S slots x 28 int32): an updated row whose
              fused position now hashes to another rank's world tile moves there; a frame supersurfel updates
              at most one row in the whole map, so at most one rank fills a slot and the sum is the union
  counts      one all-gather of the per-rank frame counters at the end of the frame: their sums
              are the global counts, their prefix gives the next frame's global id offsets

Insertion is decided locally: the owner of a new supersurfel is a pure function of its world tile
(ssf_stage_fuse, shard_owner), evaluated identically on every rank.

The exchanged records never leave HBM: the library writes them into torch tensors
(ssf_stage_*_device), the collectives run on those tensors, and only the 29 reduced ICP values
come to the host (the Gauss-Newton solve is host double arithmetic).  All library work is
enqueued on the stream given in ssf_config.stream, which must be the torch stream the collectives
are issued under (`stream=` below); extract runs ahead on the library's own streams
(submit_frame / process_submitted).

The collectives go through torch.distributed: backend "nccl" (= RCCL over xGMI) with device
tensors on the GPU box, "gloo" with CPU tensors in the CPU tests (for the CPU checker a "device"
pointer is a host pointer).  The engine is any Fusion (binding.py); the product path constructs
it from load_product() and never touches the oracle.
"""
import contextlib
import time

import numpy as np
import torch
import torch.distributed as dist

COUNT_KEYS = ("n_model", "n_visible", "n_removed", "n_inserted", "n_updated")


class RehomeReport(int):
    """what rehome_over returns: the number of rows that changed rank (an int, as before) with `.turned_away` = the arrivals
    full shards had to turn away, summed over the ranks (those rows are lost to the map: ssf_rehome_end's positive return)"""
    turned_away = 0


def rehome_over(fusion, world, group=None, device=None, on_loss="warn"):
    """the re-homing sweep of ONE rank's handle over torch.distributed (any backend): also for handles that run their
    frames through the native exchanges (ssf_comm_attach / ssf_p2p_attach).  ssf_rehome_end returns the POSITIVE number of
    arrivals a full shard turned away; those rows have already left their source shards, so the count is summed over the
    ranks and reported on every rank: on_loss = "warn" (default; warnings.warn), "raise" (RuntimeError) or "ignore"."""
    if world <= 1:
        return RehomeReport(0)
    mine = fusion.rehome_begin()
    dev = device if device is not None else torch.device("cpu")
    n_all = torch.zeros(world, dtype=torch.int64, device=dev)
    dist.all_gather_into_tensor(n_all, torch.tensor([len(mine)], dtype=torch.int64, device=dev), group=group)
    n_all = [int(v) for v in n_all.cpu()]
    cap = max(n_all)
    turned = 0
    if cap > 0:
        pad = torch.zeros((cap, mine.shape[1]), dtype=torch.int32, device=dev)
        if len(mine):
            pad[:len(mine)] = torch.from_numpy(mine).to(dev)
        got = torch.zeros((world * cap, mine.shape[1]), dtype=torch.int32, device=dev)
        dist.all_gather_into_tensor(got, pad, group=group)
        got = got.cpu().numpy().reshape(world, cap, -1)
        turned = int(fusion.rehome_end(np.concatenate([got[r, :n_all[r]] for r in range(world)])))
        lost = torch.tensor([turned], dtype=torch.int64, device=dev)
        dist.all_reduce(lost, op=dist.ReduceOp.SUM, group=group)          # (every rank makes this call: cap is the same everywhere)
        turned = int(lost.cpu()[0])
    rep = RehomeReport(sum(n_all))
    rep.turned_away = turned
    if turned > 0 and on_loss != "ignore":
        msg = "re-homing sweep: %d row(s) turned away by full shards and lost to the map (raise nb_supersurfels_max per rank)" % turned
        if on_loss == "raise":
            raise RuntimeError(msg)
        import warnings
        warnings.warn(msg, RuntimeWarning, stacklevel=2)
    return rep


def pack_frame_tables(fusion):
    """the current frame of `fusion` as ONE int32 array -- what a rank that ran the extract stage ships to the others
    ("index map + plane depth + frame SoA", 2.5 MB at 640 x 480): label map | plane depth | the seven
    arrays of the S frame supersurfels, bit patterns throughout"""
    from .binding import SURFEL_FIELDS
    fr = fusion.get_frame()
    parts = [fusion.index_map().reshape(-1).view(np.int32), fusion.plane_depth().reshape(-1).view(np.int32)]
    parts += [np.ascontiguousarray(fr[name], dt).reshape(-1).view(np.int32) for name, _, dt in SURFEL_FIELDS]
    return np.concatenate(parts)


def frame_tables_words(fusion):
    return 2 * fusion.W * fusion.H + 26 * fusion.S


def unpack_frame_tables(fusion, words):
    """-> (label, plane_depth, frame dict) for Fusion.submit_frame_tables"""
    from .binding import SURFEL_FIELDS
    P, S = fusion.W * fusion.H, fusion.S
    words = np.ascontiguousarray(words, np.int32)
    label = words[:P].reshape(fusion.H, fusion.W)
    depth = words[P:2 * P].view(np.float32).reshape(fusion.H, fusion.W)
    off, frame = 2 * P, {}
    for name, width, dt in SURFEL_FIELDS:
        a = words[off:off + width * S].view(dt)
        frame[name] = a.reshape(S, width) if width > 1 else a.reshape(S)
        off += width * S
    return label, depth, frame


class ShardedFusion:
    def __init__(self, fusion, device=None, group=None, stream=None, always_reduce=False, extract="replicated"):
        """extract = "replicated": every rank runs the extract stage of every frame (deterministic: identical tables, nothing is
        shipped).  "dealt": frame k is extracted by rank k % world only, which broadcasts its tables (pack_frame_tables) to the
        others -- 1 / world of the extract work per rank for one broadcast of 2.5 MB per frame; results are bit-identical
        (tests/test_sharded.py::test_dealt_extract_*).  Frames go one at a time in this driver (process_frame)."""
        assert extract in ("replicated", "dealt")
        self.extract = extract
        self._frame_no = 0
        self.f = fusion
        self.group = group
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        assert fusion.cfg.nranks == self.world and fusion.cfg.rank == self.rank, \
            "ssf_config.rank/nranks must match the process group"
        self.device = device if device is not None else torch.device("cpu")
        self.reduce = self.world > 1 or always_reduce   # always_reduce: exercise the collectives on one rank (tests)
        self.stream = stream                      # torch.cuda.Stream whose handle is ssf_config.stream (None on CPU)
        S = fusion.S
        with self._on_stream():
            self._icp = torch.zeros(29, dtype=torch.int64, device=self.device)
            # keys are < 2^63 (SSF_NO_MATCH = INT64_MAX): exchanged as int64, MIN order is preserved
            self._best = torch.zeros(S, dtype=torch.int64, device=self.device)
            self._matched = torch.zeros(S, dtype=torch.uint8, device=self.device)
            self._migrants = torch.zeros(S * 28, dtype=torch.int32, device=self.device)       # binding.MIGRANT_WORDS per slot
            self._mine = torch.zeros(len(COUNT_KEYS), dtype=torch.int64, device=self.device)
            self._all = torch.zeros(self.world * len(COUNT_KEYS), dtype=torch.int64, device=self.device)
        self._counts = None                       # per-rank (n_model, n_visible) after the previous frame

    def _on_stream(self):
        return torch.cuda.stream(self.stream) if self.stream is not None else contextlib.nullcontext()

    def _gather_counts(self, values):
        """all-gather of this rank's COUNT_KEYS -> (world, 5) numpy array."""
        with self._on_stream():
            self._mine.copy_(torch.tensor(values, dtype=torch.int64), non_blocking=False)
            if self.reduce:
                dist.all_gather_into_tensor(self._all, self._mine, group=self.group)
                allc = self._all.cpu().numpy()
            else:
                allc = self._mine.cpu().numpy()
        return allc.reshape(self.world, len(COUNT_KEYS))

    # ---- pipelined form: extract of submitted frames runs ahead of the exchange-bound stages ----
    def submit_frame(self, rgb, depth, dynamic_mask=None, on_device=False):
        self.f.submit_frame(rgb, depth, dynamic_mask, on_device=on_device)

    def can_submit(self):
        return self.f.can_submit()

    def pending_frames(self):
        return self.f.pending_frames()

    def process_submitted(self, prior_pose=None, _begun=False):
        f = self.f
        t0 = time.perf_counter()
        if not _begun:
            f.begin_submitted()
        if self._counts is None:                  # first frame (or after set_model): exchange the shard sizes
            c = f.counts()
            self._counts = self._gather_counts([c["n_model"], c["n_visible"], 0, 0, 0])[:, :2]
        g_model, g_vis = int(self._counts[:, 0].sum()), int(self._counts[:, 1].sum())
        f.set_shard(int(self._counts[:self.rank, 1].sum()), g_model, g_vis)
        # ---- ICP: kernel -> device record -> all-reduce in HBM -> 29 values to the host solve ----
        f.icp_begin(prior_pose)
        again = g_vis > 0 and f.cfg.icp_iter > 0
        iters = 0
        with self._on_stream():
            while again:
                f.icp_accumulate_device(self._icp.data_ptr())
                if self.reduce:
                    dist.all_reduce(self._icp, op=dist.ReduceOp.SUM, group=self.group)
                again = f.icp_update(f.icp_fetch(self._icp.data_ptr()))
                iters += 1
            valid = f.icp_end()
            t1 = time.perf_counter()
            # ---- association: tables stay in HBM ----
            f.match_device(self._best.data_ptr(), self._matched.data_ptr())
            if self.reduce:
                dist.all_reduce(self._best, op=dist.ReduceOp.MIN, group=self.group)
                dist.all_reduce(self._matched, op=dist.ReduceOp.MAX, group=self.group)
            # ---- fusion in two halves around the exchange of the rows that crossed a tile edge ----
            if self.world > 1:
                f.fuse_begin_device(self._best.data_ptr(), self._matched.data_ptr(), self._migrants.data_ptr())
                dist.all_reduce(self._migrants, op=dist.ReduceOp.SUM, group=self.group)
                res = f.fuse_end_device(self._migrants.data_ptr())
            else:
                res = f.fuse_device(self._best.data_ptr(), self._matched.data_ptr())
        res["icp_valid"], res["icp_iters"] = int(valid), iters
        allc = self._gather_counts([res[k] for k in COUNT_KEYS])
        # host-side split of the exchange-bound stages (extract: the library's own per-batch events)
        res["stage_ms"] = [res["stage_ms"][0], 1e3 * (t1 - t0), 1e3 * (time.perf_counter() - t1)]
        self._counts = allc[:, :2]
        tot = allc.sum(axis=0)
        res.update({"global_" + k: int(tot[i]) for i, k in enumerate(COUNT_KEYS)})
        return res

    def process_frame(self, rgb, depth, prior_pose=None, dynamic_mask=None, on_device=False):
        assert self.f.pending_frames() == 0, "frames are pending: use process_submitted"
        k = self._frame_no
        self._frame_no += 1
        if self.extract == "dealt" and self.world > 1:
            owner = k % self.world
            n = frame_tables_words(self.f)
            if owner == self.rank:                    # this rank's turn: extract, make the frame current, ship its tables
                self.f.submit_frame(rgb, depth, dynamic_mask, on_device=on_device)
                self.f.begin_submitted()
                buf = torch.from_numpy(pack_frame_tables(self.f)).to(self.device)
            else:
                buf = torch.empty(n, dtype=torch.int32, device=self.device)
            src = owner if self.group is None else dist.get_global_rank(self.group, owner)
            dist.broadcast(buf, src=src, group=self.group)
            if owner != self.rank:
                self.f.submit_frame_tables(*unpack_frame_tables(self.f, buf.cpu().numpy()))
            res = self.process_submitted(prior_pose, _begun=(owner == self.rank))
            res["extracted_here"] = owner == self.rank
            return res
        self.f.submit_frame(rgb, depth, dynamic_mask, on_device=on_device)
        return self.process_submitted(prior_pose)

    def rehome(self):
        """After Fusion.apply_deformation on a sharded map (a loop closure moves every row): one sweep that puts every row
        back on the rank that owns the world tile of its position (ssf_rehome_begin / _end).  The leaving rows of all ranks
        are all-gathered (padded to the largest table) and every rank keeps the records addressed to it."""
        rehome_over(self.f, self.world, group=self.group, device=self.device)
        self._counts = None

    def invalidate_counts(self):
        """Call after Fusion.set_model / apply_deformation changed the shard outside process_frame."""
        self._counts = None
