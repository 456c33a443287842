"""ctypes binding of the C ABI in include/ssf.h.

The binding is library-agnostic: it drives whatever shared object exporting this ABI it is handed.
The product entry point is load_product() (libssf_hip.so, hand-written HIP for gfx950); tests and
the bench's cpu_baseline leg hand the same class the CPU checker library.  Names follow the reference's C++ surface
(core/include/supersurfel_fusion/supersurfel_fusion.hpp:40-143): Fusion.process_frame ==
SupersurfelFusion::processFrame, get_pose == getPose, get_model == getModel, ...
"""
import ctypes as C
import os
import numpy as np

ICP_RECORD = 29
MIGRANT_WORDS = 28
_HERE = os.path.dirname(os.path.abspath(__file__))
PRODUCT_LIB = os.path.join(_HERE, "csrc", "libssf_hip.so")


class SsfConfig(C.Structure):
    _fields_ = [
        ("width", C.c_int), ("height", C.c_int),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("cell_size", C.c_int), ("lambda_pos", C.c_float), ("lambda_bound", C.c_float),
        ("lambda_size", C.c_float), ("lambda_disp", C.c_float), ("thresh_disp", C.c_float),
        ("seg_iter", C.c_int), ("seg_use_ransac", C.c_int), ("nb_samples", C.c_int),
        ("filter_iter", C.c_int), ("filter_alpha", C.c_float), ("filter_beta", C.c_float),
        ("filter_threshold", C.c_float), ("range_min", C.c_float), ("range_max", C.c_float),
        ("delta_t", C.c_int), ("conf_thresh", C.c_float), ("nb_supersurfels_max", C.c_int),
        ("icp_iter", C.c_int), ("icp_cov_thresh", C.c_double),
        ("rng_seed", C.c_uint64), ("icp_force_iters", C.c_int), ("device_id", C.c_int),
        ("stream", C.c_void_p), ("rank", C.c_int), ("nranks", C.c_int),
        ("shard_tile", C.c_float), ("depth_prefilter", C.c_int), ("prefilter_sigma_color", C.c_float),
        ("prefilter_sigma_space", C.c_float), ("profile", C.c_int), ("pipeline_depth", C.c_int), ("extract_batch", C.c_int),
    ]


class SsfSurfels(C.Structure):
    _fields_ = [("positions", C.c_void_p), ("colors", C.c_void_p), ("stamps", C.c_void_p),
                ("orientations", C.c_void_p), ("shapes", C.c_void_p), ("dims", C.c_void_p),
                ("confidences", C.c_void_p)]


class SsfFrameResult(C.Structure):
    _fields_ = [("pose", C.c_float * 12), ("icp_valid", C.c_int), ("icp_iters", C.c_int),
                ("n_model", C.c_int), ("n_visible", C.c_int), ("n_removed", C.c_int),
                ("n_inserted", C.c_int), ("n_updated", C.c_int), ("stamp", C.c_int),
                ("stage_ms", C.c_float * 3)]

    def as_dict(self):
        return dict(pose=np.array(self.pose[:], np.float32), icp_valid=self.icp_valid,
                    icp_iters=self.icp_iters, n_model=self.n_model, n_visible=self.n_visible,
                    n_removed=self.n_removed, n_inserted=self.n_inserted, n_updated=self.n_updated,
                    stamp=self.stamp, stage_ms=list(self.stage_ms[:]))


# every symbol include/ssf.h declares (tests check that each one is exported)
ABI_SYMBOLS = [
    "ssf_abi_version", "ssf_backend_name", "ssf_default_config", "ssf_create", "ssf_destroy",
    "ssf_last_error", "ssf_process_frame", "ssf_process_frame_device", "ssf_stage_extract",
    "ssf_debug_set_max_passes", "ssf_debug_set_bin_min_rows", "ssf_stage_set_shard", "ssf_stage_icp_begin",
    "ssf_stage_icp_accumulate", "ssf_stage_icp_update", "ssf_stage_icp_end", "ssf_stage_match",
    "ssf_stage_fuse", "ssf_get_pose", "ssf_set_pose", "ssf_get_counts", "ssf_get_model",
    "ssf_get_frame", "ssf_set_model", "ssf_get_index_map", "ssf_get_boundary_map",
    "ssf_get_inlier_map", "ssf_get_plane_depth", "ssf_get_superpixels", "ssf_get_model_device", "ssf_get_frame_device",
    "ssf_export_model_txt", "ssf_apply_deformation", "ssf_get_kernel_times",
    "ssf_reset_kernel_times", "ssf_set_profile", "ssf_bilateral_filter", "ssf_submit_frame",
    "ssf_process_submitted", "ssf_pending_frames", "ssf_pipeline_capacity", "ssf_can_submit", "ssf_stage_begin_submitted",
    "ssf_stage_icp_accumulate_device", "ssf_stage_icp_fetch", "ssf_stage_match_device", "ssf_stage_fuse_device",
    "ssf_comm_unique_id", "ssf_comm_attach", "ssf_comm_info", "ssf_p2p_export", "ssf_p2p_attach", "ssf_p2p_region", "ssf_p2p_attach_local", "ssf_p2p_configure", "ssf_rehome_begin", "ssf_rehome_end", "ssf_get_global_counts", "ssf_align", "ssf_fern_codes", "ssf_process_sequence", "ssf_debug_recentre", "ssf_debug_recentre_count", "ssf_get_preview_image", "ssf_stage_fuse_begin", "ssf_stage_fuse_end", "ssf_stage_fuse_begin_device", "ssf_stage_fuse_end_device",
    "ssf_sequence_times", "ssf_sequence_marks", "ssf_stream_copy_rate", "ssf_upload_stats", "ssf_pooled_streams", "ssf_waiter_matches", "ssf_waiter_match_repairs", "ssf_tuner_state", "ssf_submit_frame_tables", "ssf_comm_deal_extract",
]

SURFEL_FIELDS = (("positions", 3, np.float32), ("colors", 3, np.float32), ("stamps", 2, np.int32),
                 ("orientations", 9, np.float32), ("shapes", 6, np.float32),
                 ("dims", 2, np.float32), ("confidences", 1, np.float32))


class SsfError(RuntimeError):
    pass


class Library:
    """A loaded libssf_*.so."""

    def __init__(self, path):
        if not os.path.exists(path):
            raise SsfError("shared library not found: %s (run __graft_entry__.build())" % path)
        self.path = path
        self.lib = C.CDLL(path)
        L = self.lib
        L.ssf_backend_name.restype = C.c_char_p
        L.ssf_last_error.restype = C.c_char_p
        L.ssf_last_error.argtypes = [C.c_void_p]
        L.ssf_default_config.argtypes = [C.POINTER(SsfConfig)]
        L.ssf_create.argtypes = [C.POINTER(SsfConfig), C.POINTER(C.c_void_p)]
        L.ssf_destroy.argtypes = [C.c_void_p]
        L.ssf_destroy.restype = None
        vp = C.c_void_p
        L.ssf_process_frame.argtypes = [vp, vp, vp, vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_process_frame_device.argtypes = [vp, vp, vp, vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_stage_extract.argtypes = [vp, vp, vp, C.c_int, vp]
        L.ssf_debug_set_max_passes.argtypes = [vp, C.c_int]
        L.ssf_debug_set_bin_min_rows.argtypes = [vp, C.c_int]
        L.ssf_debug_recentre.argtypes = [vp]
        L.ssf_debug_recentre_count.argtypes = [vp]
        L.ssf_debug_recentre_count.restype = C.c_longlong
        L.ssf_stage_set_shard.argtypes = [vp, C.c_int64, C.c_int64, C.c_int64]
        L.ssf_stage_icp_begin.argtypes = [vp, vp]
        L.ssf_stage_icp_accumulate.argtypes = [vp, vp]
        L.ssf_stage_icp_update.argtypes = [vp, vp, C.POINTER(C.c_int)]
        L.ssf_stage_icp_end.argtypes = [vp, C.POINTER(C.c_int)]
        L.ssf_stage_match.argtypes = [vp, vp, vp]
        L.ssf_stage_fuse.argtypes = [vp, vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_get_pose.argtypes = [vp, vp]
        L.ssf_set_pose.argtypes = [vp, vp]
        L.ssf_get_counts.argtypes = [vp] + [C.POINTER(C.c_int)] * 4
        L.ssf_get_model.argtypes = [vp, C.c_int, C.c_int, C.POINTER(SsfSurfels)]
        L.ssf_get_frame.argtypes = [vp, C.POINTER(SsfSurfels)]
        L.ssf_set_model.argtypes = [vp, C.POINTER(SsfSurfels), C.c_int, C.c_int, C.c_int]
        for nm in ("ssf_get_index_map", "ssf_get_boundary_map", "ssf_get_inlier_map",
                   "ssf_get_plane_depth", "ssf_get_superpixels", "ssf_get_preview_image"):
            getattr(L, nm).argtypes = [vp, vp]
        L.ssf_get_model_device.argtypes = [vp, C.POINTER(SsfSurfels), C.POINTER(C.c_int)]
        L.ssf_export_model_txt.argtypes = [vp, C.c_char_p]
        L.ssf_apply_deformation.argtypes = [vp, vp, vp, vp, C.c_int, vp, vp]
        L.ssf_get_kernel_times.argtypes = [vp, vp, vp, vp, C.c_int]
        L.ssf_reset_kernel_times.argtypes = [vp]
        L.ssf_set_profile.argtypes = [vp, C.c_int]
        L.ssf_bilateral_filter.argtypes = [vp, vp, vp, C.c_int]
        L.ssf_submit_frame.argtypes = [vp, vp, vp, C.c_int, vp]
        L.ssf_process_sequence.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp]
        L.ssf_process_submitted.argtypes = [vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_pending_frames.argtypes = [vp]
        L.ssf_pipeline_capacity.argtypes = [vp]
        L.ssf_can_submit.argtypes = [vp]
        L.ssf_align.argtypes = [vp, C.POINTER(SsfSurfels), C.c_int, vp, vp, C.POINTER(C.c_int), C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ssf_fern_codes.argtypes = [vp, vp, vp, C.c_int, C.c_int, vp, vp, vp, C.c_int, vp]
        L.ssf_comm_unique_id.argtypes = [vp]
        L.ssf_comm_attach.argtypes = [vp, vp]
        L.ssf_p2p_export.argtypes = [vp, vp]
        L.ssf_p2p_attach.argtypes = [vp, vp]
        L.ssf_p2p_region.argtypes = [vp, C.POINTER(C.c_void_p), C.POINTER(C.c_size_t)]
        L.ssf_p2p_attach_local.argtypes = [vp, C.POINTER(C.c_void_p)]
        L.ssf_p2p_configure.argtypes = [vp, C.c_int, C.c_double]
        L.ssf_rehome_begin.argtypes = [vp, vp, C.c_int, C.POINTER(C.c_int)]
        L.ssf_rehome_end.argtypes = [vp, vp, C.c_int]
        L.ssf_get_global_counts.argtypes = [vp, vp]
        L.ssf_stage_begin_submitted.argtypes = [vp]
        L.ssf_stage_icp_accumulate_device.argtypes = [vp, vp]
        L.ssf_stage_icp_fetch.argtypes = [vp, vp, vp]
        L.ssf_stage_match_device.argtypes = [vp, vp, vp]
        L.ssf_stage_fuse_device.argtypes = [vp, vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_stage_fuse_begin.argtypes = [vp, vp, vp, vp]
        L.ssf_stage_fuse_end.argtypes = [vp, vp, C.POINTER(SsfFrameResult)]
        L.ssf_stage_fuse_begin_device.argtypes = [vp, vp, vp, vp]
        L.ssf_stage_fuse_end_device.argtypes = [vp, vp, C.POINTER(SsfFrameResult)]

    @property
    def backend(self):
        return self.lib.ssf_backend_name().decode()

    def default_config(self, **kw):
        cfg = SsfConfig()
        self.lib.ssf_default_config(C.byref(cfg))
        for k, v in kw.items():
            if not hasattr(cfg, k):
                raise AttributeError("ssf_config has no field %r" % k)
            setattr(cfg, k, v)
        return cfg


def load_product():
    """Load the HIP product library.  Fails loudly when it has not been built; there is no
    CPU fallback.  torch is imported first so that exactly one HIP runtime (the one torch
    ships, SONAME libamdhip64.so.7) lives in the process."""
    import torch  # noqa: F401  (loads libamdhip64 before our library resolves it)
    # SSF_PRODUCT_VARIANT=<tag>: a differently compiled build of the SAME sources next to the product (csrc/variants/<tag>/):
    # `lab` = -DSSF_EXPERIMENTS, the environment switches and measurement arms behind DESIGN.md's A/B tables (built by
    # csrc/Makefile; the product itself reads no environment variable); others by tools/build_variant.sh
    tag = os.environ.get("SSF_PRODUCT_VARIANT")
    return Library(os.path.join(_HERE, "csrc", "variants", tag, "libssf_hip.so") if tag else PRODUCT_LIB)


def load_lab():
    """The laboratory build of the product sources (csrc/variants/lab, -DSSF_EXPERIMENTS): for the tests and tools that
    exercise a measurement arm or an environment switch."""
    import torch  # noqa: F401
    return Library(os.path.join(_HERE, "csrc", "variants", "lab", "libssf_hip.so"))


def _ptr(a):
    return None if a is None else a.ctypes.data_as(C.c_void_p)


def _alloc_surfels(n):
    arrs = {name: np.zeros((n, k) if k > 1 else (n,), dt) for name, k, dt in SURFEL_FIELDS}
    st = SsfSurfels(*[arrs[name].ctypes.data_as(C.c_void_p) for name, _, _ in SURFEL_FIELDS])
    return arrs, st


class Fusion:
    """Host-side mirror of supersurfel_fusion::SupersurfelFusion for the hot path."""

    def __init__(self, library, cfg):
        self.L = library
        self.cfg = cfg
        self.h = C.c_void_p()
        rc = library.lib.ssf_create(C.byref(cfg), C.byref(self.h))
        if rc != 0:
            raise SsfError("ssf_create failed (%d): %s" % (rc, library.lib.ssf_last_error(None).decode()))
        self.W, self.H = cfg.width, cfg.height
        self.S = self.counts()["n_superpixels"]
        self._held = []          # host buffers of submitted frames: they must outlive the asynchronous copy (ssf.h)

    def close(self):
        if self.h:
            self.L.lib.ssf_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def _ck(self, rc, what):
        if rc != 0:
            raise SsfError("%s failed (%d): %s" % (what, rc, self.L.lib.ssf_last_error(self.h).decode()))

    # ---- whole frame -------------------------------------------------------------------------
    def process_frame(self, rgb, depth, prior_pose=None, dynamic_mask=None):
        rgb = np.ascontiguousarray(rgb, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        assert rgb.shape == (self.H, self.W, 3) and depth.shape == (self.H, self.W)
        prior = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float32)
        mask = None if dynamic_mask is None else np.ascontiguousarray(dynamic_mask, np.uint8)
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_process_frame(self.h, _ptr(rgb), _ptr(depth), _ptr(prior), _ptr(mask),
                                              C.byref(res)), "ssf_process_frame")
        return res.as_dict()

    def process_frame_device(self, d_rgb_ptr, d_depth_ptr, prior_pose=None):
        prior = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float32)
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_process_frame_device(self.h, C.c_void_p(d_rgb_ptr), C.c_void_p(d_depth_ptr),
                                                     _ptr(prior), None, C.byref(res)),
                 "ssf_process_frame_device")
        return res

    # ---- pipelined form ----------------------------------------------------------------------
    def submit_frame(self, rgb, depth, dynamic_mask=None, on_device=False):
        """Enqueue the extract stage of the next frame (asynchronous).  With on_device=True rgb and
        depth are device addresses that must stay valid until the frame has been processed."""
        if on_device:
            rp, dp = C.c_void_p(rgb), C.c_void_p(depth)
        else:
            rgb = np.ascontiguousarray(rgb, np.uint8)
            depth = np.ascontiguousarray(depth, np.float32)
            assert rgb.shape == (self.H, self.W, 3) and depth.shape == (self.H, self.W)
            rp, dp = _ptr(rgb), _ptr(depth)
        mask = None if dynamic_mask is None else np.ascontiguousarray(dynamic_mask, np.uint8)
        self._ck(self.L.lib.ssf_submit_frame(self.h, rp, dp, 1 if on_device else 0, _ptr(mask)), "ssf_submit_frame")
        self._held.append((rgb, depth, mask))

    def submit_frame_tables(self, label, plane_depth, frame):
        """The next frame, extracted by ANOTHER rank (ssf_submit_frame_tables): its label map (H x W int32), plane-rendered depth
        (H x W float32) and frame supersurfels (the dict get_frame() returns) take the place of submit_frame's images."""
        label = np.ascontiguousarray(label, np.int32); plane_depth = np.ascontiguousarray(plane_depth, np.float32)
        assert label.shape == (self.H, self.W) and plane_depth.shape == (self.H, self.W) and len(frame["confidences"]) == self.S
        keep = [np.ascontiguousarray(frame[name], dt) for name, _, dt in SURFEL_FIELDS]
        st = SsfSurfels(*[a.ctypes.data_as(C.c_void_p) for a in keep])
        self._ck(self.L.lib.ssf_submit_frame_tables(self.h, _ptr(label), _ptr(plane_depth), C.byref(st), 0), "ssf_submit_frame_tables")

    def prepare_sequence(self, rgb_ptrs, depth_ptrs):
        """ctypes argument arrays of a sequence (built ahead, e.g. outside a timed region): (rgb, depth, results, n)"""
        n = len(rgb_ptrs)
        return (C.c_void_p * n)(*rgb_ptrs), (C.c_void_p * n)(*depth_ptrs), (SsfFrameResult * n)(), n

    def process_prepared(self, prepared, on_device=True):
        """ssf_process_sequence on arrays from prepare_sequence; returns the raw SsfFrameResult array (as_dict() each)."""
        pr, pd, res, n = prepared
        self._ck(self.L.lib.ssf_process_sequence(self.h, pr, pd, n, 1 if on_device else 0, res), "ssf_process_sequence")
        return res

    def process_sequence(self, rgb_ptrs, depth_ptrs, on_device=True):
        """The whole submit-ahead / process-in-order loop in native code.  rgb_ptrs / depth_ptrs: raw addresses
        (device pointers when on_device, else addresses of contiguous host arrays).  Returns a list of result dicts."""
        return [r.as_dict() for r in self.process_prepared(self.prepare_sequence(rgb_ptrs, depth_ptrs), on_device)]

    def process_submitted(self, prior_pose=None):
        """ICP + association + fusion of the oldest submitted frame; returns its SsfFrameResult."""
        prior = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float32)
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_process_submitted(self.h, _ptr(prior), C.byref(res)), "ssf_process_submitted")
        if self._held:
            self._held.pop(0)
        return res

    def pending_frames(self):
        return int(self.L.lib.ssf_pending_frames(self.h))

    def can_submit(self):
        return bool(self.L.lib.ssf_can_submit(self.h))

    def pipeline_capacity(self):
        return int(self.L.lib.ssf_pipeline_capacity(self.h))

    # ---- stage seams -------------------------------------------------------------------------
    def stage_extract(self, rgb, depth, dynamic_mask=None, on_device=False):
        if on_device:
            rp, dp = C.c_void_p(rgb), C.c_void_p(depth)
        else:
            rgb = np.ascontiguousarray(rgb, np.uint8)
            depth = np.ascontiguousarray(depth, np.float32)
            rp, dp = _ptr(rgb), _ptr(depth)
        mask = None if dynamic_mask is None else np.ascontiguousarray(dynamic_mask, np.uint8)
        self._ck(self.L.lib.ssf_stage_extract(self.h, rp, dp, 1 if on_device else 0, _ptr(mask)), "ssf_stage_extract")

    def debug_recentre(self):
        self._ck(self.L.lib.ssf_debug_recentre(self.h), "ssf_debug_recentre")

    def debug_recentre_count(self):
        return int(self.L.lib.ssf_debug_recentre_count(self.h))

    def set_max_passes(self, n):
        self._ck(self.L.lib.ssf_debug_set_max_passes(self.h, n), "ssf_debug_set_max_passes")

    def set_bin_min_rows(self, n):
        """visible rows from which a frame's tracking streams a tile-sorted copy of them (product default: never; 0 = always)"""
        self._ck(self.L.lib.ssf_debug_set_bin_min_rows(self.h, int(n)), "ssf_debug_set_bin_min_rows")

    def set_shard(self, id_offset, global_n_model, global_n_visible):
        self._ck(self.L.lib.ssf_stage_set_shard(self.h, id_offset, global_n_model, global_n_visible), "ssf_stage_set_shard")

    def icp_begin(self, prior_pose=None):
        prior = None if prior_pose is None else np.ascontiguousarray(prior_pose, np.float32)
        self._ck(self.L.lib.ssf_stage_icp_begin(self.h, _ptr(prior)), "ssf_stage_icp_begin")

    def icp_accumulate(self):
        sums = np.zeros(ICP_RECORD, np.int64)
        self._ck(self.L.lib.ssf_stage_icp_accumulate(self.h, _ptr(sums)), "ssf_stage_icp_accumulate")
        return sums

    def icp_update(self, sums):
        sums = np.ascontiguousarray(sums, np.int64)
        again = C.c_int(0)
        self._ck(self.L.lib.ssf_stage_icp_update(self.h, _ptr(sums), C.byref(again)), "ssf_stage_icp_update")
        return bool(again.value)

    def icp_end(self):
        valid = C.c_int(0)
        self._ck(self.L.lib.ssf_stage_icp_end(self.h, C.byref(valid)), "ssf_stage_icp_end")
        return bool(valid.value)

    def match(self):
        best = np.zeros(self.S, np.uint64)
        matched = np.zeros(self.S, np.uint8)
        self._ck(self.L.lib.ssf_stage_match(self.h, _ptr(best), _ptr(matched)), "ssf_stage_match")
        return best, matched

    def fuse(self, best, matched):
        best = np.ascontiguousarray(best, np.uint64)
        matched = np.ascontiguousarray(matched, np.uint8)
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_stage_fuse(self.h, _ptr(best), _ptr(matched), C.byref(res)), "ssf_stage_fuse")
        return res.as_dict()

    def fuse_begin(self, best, matched):
        """first half of the fuse stage; returns this shard's migrant table (S x MIGRANT_WORDS int32, see ssf.h)"""
        best = np.ascontiguousarray(best, np.uint64)
        matched = np.ascontiguousarray(matched, np.uint8)
        table = np.zeros((self.S, MIGRANT_WORDS), np.int32)
        self._ck(self.L.lib.ssf_stage_fuse_begin(self.h, _ptr(best), _ptr(matched), _ptr(table)), "ssf_stage_fuse_begin")
        return table

    def fuse_end(self, table=None):
        """second half: rows addressed to this rank in the (rank-reduced) table arrive, then classify + reorder"""
        table = None if table is None else np.ascontiguousarray(table, np.int32)
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_stage_fuse_end(self.h, _ptr(table), C.byref(res)), "ssf_stage_fuse_end")
        return res.as_dict()

    def fuse_begin_device(self, d_best_ptr, d_matched_ptr, d_table_ptr):
        self._ck(self.L.lib.ssf_stage_fuse_begin_device(self.h, C.c_void_p(d_best_ptr), C.c_void_p(d_matched_ptr), C.c_void_p(d_table_ptr)),
                 "ssf_stage_fuse_begin_device")

    def fuse_end_device(self, d_table_ptr):
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_stage_fuse_end_device(self.h, C.c_void_p(d_table_ptr), C.byref(res)), "ssf_stage_fuse_end_device")
        return res.as_dict()

    # ---- loop closure: registration of a keyframe's supersurfels against the current frame ---------
    def align(self, source, init_pose=None):
        """source: dict with positions (n,3), colors (n,3), orientations (n,9) [, confidences (n,)].
        Returns dict(rel_pose (12,), valid, iters, pairs)."""
        pos = np.ascontiguousarray(source["positions"], np.float32)
        col = np.ascontiguousarray(source["colors"], np.float32)
        ori = np.ascontiguousarray(source["orientations"], np.float32)
        conf = source.get("confidences")
        conf = None if conf is None else np.ascontiguousarray(conf, np.float32)
        n = len(pos)
        st = SsfSurfels(_ptr(pos), _ptr(col), None, _ptr(ori), None, None, _ptr(conf))
        init = None if init_pose is None else np.ascontiguousarray(init_pose, np.float32)
        rel = np.zeros(12, np.float32)
        valid, iters, pairs = C.c_int(0), C.c_int(0), C.c_int(0)
        self._ck(self.L.lib.ssf_align(self.h, C.byref(st), n, _ptr(init), _ptr(rel), C.byref(valid), C.byref(iters),
                                      C.byref(pairs)), "ssf_align")
        return dict(rel_pose=rel, valid=bool(valid.value), iters=iters.value, pairs=pairs.value)

    def fern_codes(self, rgb, depth, fern_pos, fern_rgb, fern_depth):
        rgb = np.ascontiguousarray(rgb, np.uint8); depth = np.ascontiguousarray(depth, np.float32)
        fp = np.ascontiguousarray(fern_pos, np.uint32); fr = np.ascontiguousarray(fern_rgb, np.uint8)
        fd = np.ascontiguousarray(fern_depth, np.float32)
        n = len(fd)
        codes = np.zeros(n, np.uint8)
        self._ck(self.L.lib.ssf_fern_codes(self.h, _ptr(rgb), _ptr(depth), depth.shape[1], depth.shape[0], _ptr(fp), _ptr(fr),
                                           _ptr(fd), n, _ptr(codes)), "ssf_fern_codes")
        return codes

    # ---- multi-GPU, native RCCL ------------------------------------------------------------------
    def comm_deal_extract(self, mode=1):
        """After comm_attach, on every rank: batch j of the frame stream is extracted by rank j % nranks alone, which broadcasts
        its frames' tables (ssf_comm_deal_extract; mode 2 additionally re-imports on the extracting rank: a self-check)"""
        self._ck(self.L.lib.ssf_comm_deal_extract(self.h, int(mode)), "ssf_comm_deal_extract")

    def comm_attach(self, group=None):
        """Attach an RCCL communicator over the ranks of a torch.distributed group (which is only used
        to ship rank 0's unique id); afterwards process_frame / process_submitted exchange natively."""
        import torch
        import torch.distributed as dist
        ident, err = np.zeros(128, np.uint8), None
        if dist.get_rank(group) == 0:
            rc = self.L.lib.ssf_comm_unique_id(_ptr(ident))
            if rc != 0:
                err = "ssf_comm_unique_id failed (%d): %s" % (rc, self.L.lib.ssf_last_error(None).decode())
        obj = [None if err else ident.tobytes(), err]
        dist.broadcast_object_list(obj, src=0, group=group)      # every rank learns about a failure on rank 0
        if obj[0] is None:
            raise SsfError(obj[1])
        ident = np.frombuffer(obj[0], np.uint8).copy()
        self._ck(self.L.lib.ssf_comm_attach(self.h, _ptr(ident)), "ssf_comm_attach")

    def comm_info(self):
        """what exchange is attached and how many ranks it reports itself: dict(backend 'none' | 'rccl' | 'p2p', ranks, rank)"""
        b, n, r = C.c_int(), C.c_int(), C.c_int()
        self._ck(self.L.lib.ssf_comm_info(self.h, C.byref(b), C.byref(n), C.byref(r)), "ssf_comm_info")
        return dict(backend=("none", "rccl", "p2p")[b.value], ranks=n.value, rank=r.value)

    # peer-to-peer exchange (ssf_p2p_* in ssf.h): the ranks of one node trade their records through each other's HBM
    P2P_HANDLE_BYTES = 64

    def p2p_configure(self, all_ranks_on_this_device=False, timeout_s=30.0):
        """before p2p_export / p2p_region: plain device memory when every rank is a handle on this GPU (fine-grained
        otherwise), and the wall-clock bound of every in-kernel wait for a peer"""
        self._ck(self.L.lib.ssf_p2p_configure(self.h, 1 if all_ranks_on_this_device else 0, float(timeout_s)), "ssf_p2p_configure")

    def p2p_export(self):
        """64-byte IPC handle of this handle's exchange region (to be shipped to the other ranks)"""
        out = np.zeros(self.P2P_HANDLE_BYTES, np.uint8)
        self._ck(self.L.lib.ssf_p2p_export(self.h, _ptr(out)), "ssf_p2p_export")
        return out

    def p2p_attach(self, handles=None, group=None):
        """handles: nranks x 64 bytes in rank order; None: all-gathered over torch.distributed (group)"""
        if handles is None:
            import torch.distributed as dist
            mine = self.p2p_export().tobytes()
            got = [None] * dist.get_world_size(group)
            dist.all_gather_object(got, mine, group=group)
            handles = np.concatenate([np.frombuffer(b, np.uint8) for b in got])
        handles = np.ascontiguousarray(handles, np.uint8).reshape(-1)
        assert handles.size == self.P2P_HANDLE_BYTES * self.cfg.nranks
        self._ck(self.L.lib.ssf_p2p_attach(self.h, _ptr(handles)), "ssf_p2p_attach")

    def p2p_region(self):
        """(address, bytes) of this handle's exchange region: for ranks that live in one process"""
        reg, nb = C.c_void_p(), C.c_size_t()
        self._ck(self.L.lib.ssf_p2p_region(self.h, C.byref(reg), C.byref(nb)), "ssf_p2p_region")
        return reg.value, nb.value

    def p2p_attach_local(self, regions):
        arr = (C.c_void_p * len(regions))(*regions)
        self._ck(self.L.lib.ssf_p2p_attach_local(self.h, arr), "ssf_p2p_attach_local")

    def global_counts(self):
        out = np.zeros(5, np.int64)
        self._ck(self.L.lib.ssf_get_global_counts(self.h, _ptr(out)), "ssf_get_global_counts")
        return dict(zip(("n_model", "n_visible", "n_removed", "n_inserted", "n_updated"), (int(v) for v in out)))

    # device-resident variants: the arguments are raw addresses (torch tensor .data_ptr())
    def begin_submitted(self):
        self._ck(self.L.lib.ssf_stage_begin_submitted(self.h), "ssf_stage_begin_submitted")

    def icp_accumulate_device(self, d_sums_ptr):
        self._ck(self.L.lib.ssf_stage_icp_accumulate_device(self.h, C.c_void_p(d_sums_ptr)), "ssf_stage_icp_accumulate_device")

    def icp_fetch(self, d_sums_ptr):
        sums = np.zeros(ICP_RECORD, np.int64)
        self._ck(self.L.lib.ssf_stage_icp_fetch(self.h, C.c_void_p(d_sums_ptr), _ptr(sums)), "ssf_stage_icp_fetch")
        return sums

    def match_device(self, d_best_ptr, d_matched_ptr):
        self._ck(self.L.lib.ssf_stage_match_device(self.h, C.c_void_p(d_best_ptr), C.c_void_p(d_matched_ptr)),
                 "ssf_stage_match_device")

    def fuse_device(self, d_best_ptr, d_matched_ptr):
        res = SsfFrameResult()
        self._ck(self.L.lib.ssf_stage_fuse_device(self.h, C.c_void_p(d_best_ptr), C.c_void_p(d_matched_ptr), C.byref(res)),
                 "ssf_stage_fuse_device")
        return res.as_dict()

    # ---- read back ---------------------------------------------------------------------------
    def get_pose(self):
        p = np.zeros(12, np.float32)
        self._ck(self.L.lib.ssf_get_pose(self.h, _ptr(p)), "ssf_get_pose")
        return p

    def set_pose(self, p):
        p = np.ascontiguousarray(p, np.float32)
        self._ck(self.L.lib.ssf_set_pose(self.h, _ptr(p)), "ssf_set_pose")

    def counts(self):
        v = [C.c_int(0) for _ in range(4)]
        self._ck(self.L.lib.ssf_get_counts(self.h, *[C.byref(x) for x in v]), "ssf_get_counts")
        return dict(n_model=v[0].value, n_visible=v[1].value, stamp=v[2].value, n_superpixels=v[3].value)

    def get_model(self, first=0, count=None):
        if count is None:
            count = self.counts()["n_model"] - first
        arrs, st = _alloc_surfels(max(count, 0))
        if count > 0:
            self._ck(self.L.lib.ssf_get_model(self.h, first, count, C.byref(st)), "ssf_get_model")
        return arrs

    def get_frame(self):
        arrs, st = _alloc_surfels(self.S)
        self._ck(self.L.lib.ssf_get_frame(self.h, C.byref(st)), "ssf_get_frame")
        return arrs

    def set_model(self, arrs, n_visible, stamp):
        n = len(arrs["confidences"])
        keep = [np.ascontiguousarray(arrs[name], dt) for name, _, dt in SURFEL_FIELDS]
        st = SsfSurfels(*[a.ctypes.data_as(C.c_void_p) for a in keep])
        self._ck(self.L.lib.ssf_set_model(self.h, C.byref(st), n, n_visible, stamp), "ssf_set_model")

    def _map(self, fn, dtype, shape):
        out = np.zeros(shape, dtype)
        self._ck(getattr(self.L.lib, fn)(self.h, _ptr(out)), fn)
        return out

    def index_map(self):
        return self._map("ssf_get_index_map", np.int32, (self.H, self.W))

    def boundary_map(self):
        return self._map("ssf_get_boundary_map", np.int32, (self.H, self.W))

    def inlier_map(self):
        return self._map("ssf_get_inlier_map", np.uint8, (self.H, self.W))

    def plane_depth(self):
        return self._map("ssf_get_plane_depth", np.float32, (self.H, self.W))

    def preview_image(self):
        """computeSuperpixelSegIm: H x W x 3 uint8 (B, G, R), boundaries white"""
        return self._map("ssf_get_preview_image", np.uint8, (self.H, self.W, 3))

    def slanted_plane_image(self):
        """computeSlantedPlaneIm: the plane-rendered depth, H x W float32"""
        return self.plane_depth()

    def model_device(self):
        """ssf_get_model_device: (SsfSurfels of device pointers in the reference's layout, n_model)"""
        st, n = SsfSurfels(), C.c_int(0)
        self._ck(self.L.lib.ssf_get_model_device(self.h, C.byref(st), C.byref(n)), "ssf_get_model_device")
        return st, n.value

    def superpixels(self):
        return self._map("ssf_get_superpixels", np.float32, (self.S, 9))

    def export_model_txt(self, path):
        self._ck(self.L.lib.ssf_export_model_txt(self.h, path.encode()), "ssf_export_model_txt")

    def apply_deformation(self, node_pos, node_rot, node_trans, weights4, idx4):
        a = [np.ascontiguousarray(node_pos, np.float32), np.ascontiguousarray(node_rot, np.float32),
             np.ascontiguousarray(node_trans, np.float32), np.ascontiguousarray(weights4, np.float32),
             np.ascontiguousarray(idx4, np.int32)]
        self._ck(self.L.lib.ssf_apply_deformation(self.h, _ptr(a[0]), _ptr(a[1]), _ptr(a[2]), len(a[0]),
                                                  _ptr(a[3]), _ptr(a[4])), "ssf_apply_deformation")

    def rehome_begin(self, capacity=None):
        """rows that now belong to another rank's tile leave this shard -> (n, MIGRANT_WORDS) int32 records"""
        cap = self.counts()["n_model"] if capacity is None else int(capacity)
        table = np.zeros((max(cap, 1), MIGRANT_WORDS), np.int32)
        n = C.c_int(0)
        self._ck(self.L.lib.ssf_rehome_begin(self.h, _ptr(table), cap, C.byref(n)), "ssf_rehome_begin")
        return table[:n.value].copy()

    def rehome_end(self, table):
        """table: the records of all ranks in rank order (those addressed to this rank are appended).  Returns the number of
        arrivals a full shard had to turn away (lost to the map, like an arrival at a full shard inside a frame)"""
        table = np.ascontiguousarray(table, np.int32).reshape(-1, MIGRANT_WORDS)
        rc = self.L.lib.ssf_rehome_end(self.h, _ptr(table), len(table))
        if rc < 0:
            self._ck(rc, "ssf_rehome_end")
        return rc

    def bilateral_filter(self, depth):
        depth = np.ascontiguousarray(depth, np.float32)
        out = np.zeros_like(depth)
        self._ck(self.L.lib.ssf_bilateral_filter(self.h, _ptr(depth), _ptr(out), 0), "ssf_bilateral_filter")
        return out

    def kernel_times(self, max_k=64):
        names = (C.c_char_p * max_k)()
        ms = np.zeros(max_k, np.float64)
        calls = np.zeros(max_k, np.int64)
        n = self.L.lib.ssf_get_kernel_times(self.h, C.cast(names, C.c_void_p), _ptr(ms), _ptr(calls), max_k)
        return {names[i].decode(): (float(ms[i]), int(calls[i])) for i in range(max(n, 0))}

    def reset_kernel_times(self):
        self.L.lib.ssf_reset_kernel_times(self.h)

    def set_profile(self, level):
        """0: off, 2: stage_ms split only, 1: stage split + per-kernel hipEvent times"""
        self._ck(self.L.lib.ssf_set_profile(self.h, int(level)), "ssf_set_profile")
