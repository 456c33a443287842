/*
 * ssf.h -- C ABI of the supersurfel_fusion per-frame hot path (extract | ICP | fuse).
 *
 * This header is the drop-in boundary.  It replaces, for the hot path only, the C++
 * class supersurfel_fusion::SupersurfelFusion of the reference
 * (core/include/supersurfel_fusion/supersurfel_fusion.hpp:40-143):
 *
 *   ssf_create            <- SupersurfelFusion::initialize          (supersurfel_fusion.hpp:46-74,
 *                                                                    supersurfel_fusion.cu:49-164)
 *   ssf_process_frame     <- SupersurfelFusion::processFrame        (supersurfel_fusion.hpp:75-76,
 *                                                                    supersurfel_fusion.cu:166-530)
 *   ssf_get_pose          <- getPose()                              (supersurfel_fusion.hpp:84)
 *   ssf_get_model/_frame  <- getModel()/getFrame()                  (supersurfel_fusion.hpp:81-82,
 *                                                                    supersurfels.hpp:32-93)
 *   ssf_get_counts        <- getnbSupersurfels()/getStamp()         (supersurfel_fusion.hpp:85-90)
 *   ssf_get_index_map ... <- TPS_RGBD::getIndexImage() etc.         (TPS_RGBD.hpp:77-81)
 *   ssf_export_model_txt  <- exportModel(std::string)               (supersurfel_fusion.cu:595-633)
 *   ssf_apply_deformation <- DeformationGraph::applyGraphToModel    (deformation_graph.cu:840-861,
 *                                                                    deformation_graph_kernels.cu:27-73)
 *   ssf_stage_*           <- the stage seams inside processFrame:
 *        extract  = TPS_RGBD::compute/filter/computeDepthImage + generateSupersurfels
 *                   (supersurfel_fusion.cu:189-194, TPS_RGBD.cu:101-525)
 *        icp_*    = DenseRegistration::featureConstrainedSymmetricICP
 *                   (dense_registration.hpp:57-72, dense_registration.cu:245-424)
 *        match/fuse = the fuse block (supersurfel_fusion.cu:351-483)
 *
 * Plain pointers and sizes only; no C++/torch types.  All functions return 0 on success and a
 * negative ssf_status on failure (the reference calls exit(-1), cuda_error_check.h:30-66; this
 * ABI never exits).  A handle is thread-compatible: one frame in flight per handle, callers
 * serialise (the reference is driven from one ROS spin thread).
 *
 * Two shared libraries export exactly this ABI:
 *   - libssf_hip.so     (supersurfel_fusion_amd/csrc, hand-written HIP for gfx950)  = the product
 *   - libssf_oracle.so  (oracle/, plain C++ restatement of the reference)           = test checker
 */
#ifndef SSF_H
#define SSF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define SSF_ABI_VERSION 3
/* "no candidate" key of the association tables: the largest value that orders the same as a signed
 * and as an unsigned 64-bit integer (valid keys are < 2^63: the distance is a positive float), so a
 * MIN all-reduce works on backends without unsigned types */
#define SSF_NO_MATCH 0x7FFFFFFFFFFFFFFFull
#define SSF_MAX_PIPELINE_DEPTH 3
#define SSF_MAX_EXTRACT_BATCH 16

typedef enum ssf_status {
    SSF_OK = 0,
    SSF_ERR_INVALID_ARG = -1,
    SSF_ERR_DEVICE = -2,       /* HIP runtime error, see ssf_last_error */
    SSF_ERR_NO_DEVICE = -3,    /* product library loaded on a box without a gfx950 GPU */
    SSF_ERR_CAPACITY = -4,
    SSF_ERR_STATE = -5,        /* stage called out of order */
    SSF_ERR_IO = -6
} ssf_status;
/* Errors are NEGATIVE.  One entry point also returns positive values that are not errors: ssf_rehome_end's return is the
 * number of arrivals a full shard had to turn away (0 = none) -- test `rc < 0` there, not `rc != SSF_OK`. */

/* Number of values in one ICP normal-equation record: JtJ upper triangle (21, row-major order
 * 00,01,..,05,11,..,55), Jtr (6), sum r2^2 (1), inlier count (1).  Mirrors MotionTrackingData
 * (dense_registration_types.hpp:55-60), but carried as exact fixed-point int64:
 *   [0..20]  JtJ   scale 2^20      [21..26] Jtr  scale 2^24
 *   [27]     r     scale 2^44      [28]     inliers (count)
 * Integer sums are order independent, so the record is bit-identical for any thread/block/rank
 * decomposition; a SUM all-reduce over ranks is exact. */
#define SSF_ICP_RECORD 29
#define SSF_ICP_SCALE_JTJ 1048576.0          /* 2^20 */
#define SSF_ICP_SCALE_JTR 16777216.0         /* 2^24 */
#define SSF_ICP_SCALE_R   17592186044416.0   /* 2^44 */
/* ssf_align: fixed-point scales of the centroid sums (metres) and of the squared-distance sum */
#define SSF_ALIGN_SCALE_POS 16777216.0         /* 2^24 */
#define SSF_ALIGN_SCALE_D2  1073741824.0       /* 2^30 */
#define SSF_ALIGN_LIM       4503599627370496.0 /* 2^52 */

/* POD mirror of the path-relevant arguments of SupersurfelFusion::initialize
 * (supersurfel_fusion.hpp:46-74) + CamParam (cam_param.hpp:27-31).  Defaults (ssf_default_config)
 * are the reference's C++ default arguments. */
typedef struct ssf_config {
    /* CamParam */
    int   width, height;
    float fx, fy, cx, cy;
    /* TPS_RGBD */
    int   cell_size;           /* 16 */
    float lambda_pos;          /* 50 */
    float lambda_bound;        /* 1000 */
    float lambda_size;         /* 10000 */
    float lambda_disp;         /* 1e6 */
    float thresh_disp;         /* 1e-4 */
    int   seg_iter;            /* 10 */
    int   seg_use_ransac;      /* 1 */
    int   nb_samples;          /* 16 */
    int   filter_iter;         /* 4 */
    float filter_alpha;        /* 0.1 */
    float filter_beta;         /* 1.0 */
    float filter_threshold;    /* 0.05 */
    /* model */
    float range_min;           /* 0.2 */
    float range_max;           /* 5.0 */
    int   delta_t;             /* 20 */
    float conf_thresh;         /* 2500 */
    int   nb_supersurfels_max; /* 50000 */
    /* ICP */
    int    icp_iter;           /* 10 */
    double icp_cov_thresh;     /* 0.04 */
    /* ---- build-specific (no reference counterpart) ---- */
    uint64_t rng_seed;         /* 1234 (the reference seeds cuRAND with 1234, TPS_RGBD_kernels.cu:321) */
    int   icp_force_iters;     /* 1: never early-stop on the 0.9995 ratio (BASELINE config 3) */
    int   device_id;           /* HIP device ordinal (product only) */
    void* stream;              /* hipStream_t to launch on, NULL = library-owned stream */
    int   rank, nranks;        /* model shard of this handle; 0/1 = unsharded */
    float shard_tile;          /* world-space tile edge (m) hashed to the owning rank, 0.5 */
    int   depth_prefilter;     /* 1 (default): run the bilateral depth pre-filter inside process_frame, as the
                                  reference's processFrame always does (supersurfel_fusion.cu:180: in-place
                                  cv::cuda::bilateralFilter, third party); 0: the caller passes the depth it wants
                                  segmented as is (already filtered, or synthetic) */
    float prefilter_sigma_color;   /* 0.03 m  */
    float prefilter_sigma_space;   /* 4.5 px  */
    int   profile;             /* 0: none (fastest); 2: stage_ms split (one event synchronise per frame);
                                  1: additionally bracket every kernel with hipEvents (ssf_get_kernel_times) */
    int   pipeline_depth;      /* 0 (default): frames are processed strictly one after the other, as the reference
                                  does.  d > 0: ssf_submit_frame may run the extract stage of up to d + 1 frames
                                  ahead of ICP/fusion on separate HIP streams (replay / offline mapping: raises
                                  throughput, results are bit-identical to the sequential order).  Clamped to
                                  0..SSF_MAX_PIPELINE_DEPTH; 2 is the measured optimum on MI355X: the command
                                  processor serves 4 hardware queues concurrently = the track stream + 3
                                  extract streams, a 5th active queue halves the throughput. */
    int   extract_batch;       /* 1 (default).  b > 1: the extract stage of b submitted frames runs as ONE chain of
                                  launches (every kernel relabels the tiles of all b frames; the passes are
                                  launch-latency bound, so b frames cost little more than one).  Only the
                                  submit/process form batches; 1..SSF_MAX_EXTRACT_BATCH.  Measured optima on
                                  MI355X: 8 when the track chain bounds the replay (frames already filtered),
                                  12 with the depth pre-filter in the frame (the extract stage bounds it then). */
} ssf_config;

typedef struct ssf_handle ssf_handle;

/* Host-side view of a supersurfel set in the reference's SoA layout (supersurfels.hpp:34-40):
 * positions float3, colors float3 (sRGB 0..255), stamps int2 (t_init,t_last), orientations Mat33
 * (row-major rows = major, minor, normal), shapes Cov3 (xx,xy,xz,yy,yz,zz), dims float2,
 * confidences float (-1 = invalid).  104 bytes per supersurfel.  Any pointer may be NULL (skipped). */
typedef struct ssf_surfels {
    float*   positions;     /* 3*n */
    float*   colors;        /* 3*n */
    int32_t* stamps;        /* 2*n */
    float*   orientations;  /* 9*n */
    float*   shapes;        /* 6*n */
    float*   dims;          /* 2*n */
    float*   confidences;   /* n   */
} ssf_surfels;

typedef struct ssf_frame_result {
    float pose[12];      /* row-major 3x3 R then t: camera-to-map, as Transform3 (matrix_types.h:38-42) */
    int   icp_valid;     /* featureConstrainedSymmetricICP return value (0 also when ICP did not run) */
    int   icp_iters;     /* executed iterations */
    int   n_model;       /* nbSupersurfels after the frame (this shard) */
    int   n_visible;     /* nbVisible after the frame (this shard) */
    int   n_removed;
    int   n_inserted;
    int   n_updated;
    int   stamp;         /* stamp the frame was processed at */
    float stage_ms[3];   /* extract | icp | fuse time on the library's stream (cfg.profile != 0, else 0) */
} ssf_frame_result;

/* ---- lifetime ------------------------------------------------------------------------------- */
int  ssf_abi_version(void);
/* "hip-gfx950" for the product, "cpu-oracle" for the checker. */
const char* ssf_backend_name(void);
void ssf_default_config(ssf_config* cfg);
int  ssf_create(const ssf_config* cfg, ssf_handle** out);
void ssf_destroy(ssf_handle* h);
/* Last error text of this handle (or of ssf_create when h == NULL). */
const char* ssf_last_error(const ssf_handle* h);

/* ---- whole frame (single shard) ------------------------------------------------------------- */
/* rgb: H*W*3 uint8, RGB order (the node passes RGB, supersurfel_fusion_rgbd_benchmark_node.cpp);
 * depth_m: H*W float32 metres, 0 = hole (after the node's convertTo(CV_32FC1, depthScale)).
 * prior_pose: nullable 12 floats (the sparse-VO pose prior of supersurfel_fusion.cu:225-228);
 *             NULL = previous pose.
 * dynamic_mask: nullable S bytes, non-zero marks a dynamic superpixel: frame confidence := -1
 *             (the MOD hook, motion_detection.cu:573-578).
 * Host-pointer and device-pointer variants; the device variant reads buffers already in HBM. */
int ssf_process_frame(ssf_handle* h, const uint8_t* rgb, const float* depth_m,
                      const float* prior_pose, const uint8_t* dynamic_mask, ssf_frame_result* out);
int ssf_process_frame_device(ssf_handle* h, const void* d_rgb, const void* d_depth_m,
                             const float* prior_pose, const uint8_t* dynamic_mask,
                             ssf_frame_result* out);

/* ---- pipelined form (cfg.pipeline_depth > 0 to gain anything; valid with 0 too) ------------- */
/* The extract stage (segmentation -> frame supersurfels) of a frame depends on no earlier frame
 * except through the RANSAC draw counters, while ICP/fusion of frame k needs the map after frame
 * k-1.  ssf_submit_frame enqueues the extract stage of the NEXT frame asynchronously (its own HIP
 * stream; returns without waiting) and ssf_process_submitted runs ICP + association + fusion of the
 * OLDEST submitted frame and returns its result, exactly what ssf_process_frame would have
 * returned.  At most ssf_pipeline_capacity() frames may be pending (SSF_ERR_STATE beyond that).
 * With extract_batch = b the launches happen once b frames have been submitted (or when the
 * first of them is asked for by ssf_process_submitted).
 * Input buffers -- device (on_device = 1) AND host (on_device = 0: the copy to the device is enqueued, not waited
 * for; page-locked memory is read asynchronously) -- must stay valid and unmodified until the frame has been
 * processed; the same holds for dynamic_mask.  The
 * per-frame getters below refer to the last processed frame and are invalidated by the next
 * ssf_submit_frame once the pipeline wraps around (always valid with pipeline_depth = 0).
 * Replaces the loop body of the replay node (supersurfel_fusion_rgbd_benchmark_node.cpp, one
 * processFrame per image pair) when frames are available ahead of time. */
int ssf_submit_frame(ssf_handle* h, const void* rgb, const void* depth_m, int on_device,
                     const uint8_t* dynamic_mask);
int ssf_process_submitted(ssf_handle* h, const float* prior_pose, ssf_frame_result* out);
/* The NEXT frame, extracted ELSEWHERE (multi-GPU: the extract stage dealt over the ranks, SURVEY.md section 8e -- "compute on
 * GPU 0 and broadcast index map + plane depth + frame SoA"): takes the place of ssf_submit_frame for a frame whose extract
 * stage another rank has run.  What the track chain reads of an extracted frame, in the reference's own terms:
 *   label        H x W int32   TPS_RGBD::getIndexImage (TPS_RGBD.hpp:77): ssf_get_index_map of the rank that extracted it
 *   plane_depth  H x W float   the plane-rendered depth (TPS_RGBD::computeDepthImage, TPS_RGBD.cu:507-525): ssf_get_plane_depth
 *   frame        S rows        the frame supersurfels (generateSupersurfels, supersurfel_fusion.cu:551-593): ssf_get_frame
 * (2.5 MB at 640 x 480).  The library rebuilds its private tables from them (bit-identical to a local extract: they are pure
 * functions of these three) and queues the frame like a submitted one; the handle's frame counter / RANSAC epoch advance as
 * if it had extracted the frame itself, so local and foreign frames may alternate freely.  ssf_process_submitted,
 * ssf_stage_begin_submitted, ssf_pending_frames, ssf_can_submit treat it like any submitted frame.  A frame submitted this
 * way forms a batch of its own.  The per-pixel getters other than the two above (inlier map, superpixel table, preview)
 * are not defined for such a frame.  on_device = 1: device pointers. */
int ssf_submit_frame_tables(ssf_handle* h, const int32_t* label, const float* plane_depth, const ssf_surfels* frame, int on_device);
int ssf_pending_frames(const ssf_handle* h);
/* Frames that can be pending at once: (pipeline_depth + 1) * extract_batch. */
int ssf_pipeline_capacity(const ssf_handle* h);
/* 1 when ssf_submit_frame would accept a frame now (a batch context is open or free), else 0:
 * a batch whose frames are still being consumed keeps its context until the last one is fused. */
int ssf_can_submit(const ssf_handle* h);

/* A whole recorded sequence in one call: the submit-ahead / process-in-order loop above in native code (what
 * SupersurfelFusionRGBDBenchmarkNode::run does frame by frame).  rgb[i] / depth_m[i]: the n frames (host or
 * device pointers, see ssf_submit_frame); out[i]: result of frame i.  No pose priors, no dynamic masks.
 * Host frames (on_device = 0) of a pipelined handle are copied to the device ahead of their turn by a worker
 * thread of the library, which lives for the duration of the call; the frame buffers must stay valid until the
 * call returns (they need not be page-locked). */
int ssf_process_sequence(ssf_handle* h, const void* const* rgb, const void* const* depth_m, int n, int on_device,
                         ssf_frame_result* out);

/* ---- stage seams (used by the sharded multi-GPU driver and by the parity tests) ------------- */
/* extract: ingest + TPS segmentation + plane filter + plane depth + frame supersurfels. */
int ssf_stage_extract(ssf_handle* h, const void* rgb, const void* depth_m, int on_device,
                      const uint8_t* dynamic_mask);
/* Stop the segmentation after max_passes relabelling passes (0 = all); test/bisect aid. */
int ssf_debug_set_max_passes(ssf_handle* h, int max_passes);
/* Test / tuning hook: from how many visible rows on a frame's tracking streams a TILE-SORTED copy of their ICP /
 * association fields (default 400 000: the copy pays at BASELINE config 3's million visible rows, not at the metric's 120 k;
 * 0 = always, < 0 = never; results are the same bit for bit either way: DESIGN.md section 4.4.  The checker accepts and ignores it). */
int ssf_debug_set_bin_min_rows(ssf_handle* h, int min_rows);
/* Product-internal upkeep made callable for tests: compact the out-of-view row store now (DESIGN.md
 * section 3; a no-op for the results).  The CPU checker has no such store and returns SSF_OK. */
int ssf_debug_recentre(ssf_handle* h);
/* Number of compactions of the out-of-view store so far, forced or automatic (0 for the CPU checker). */
long long ssf_debug_recentre_count(const ssf_handle* h);
/* Global (all-shard) model counts and this shard's global id offset; must precede icp_begin when
 * nranks > 1.  Unsharded handles ignore it. */
int ssf_stage_set_shard(ssf_handle* h, int64_t id_offset, int64_t global_n_model,
                        int64_t global_n_visible);
int ssf_stage_icp_begin(ssf_handle* h, const float* prior_pose);
/* One pass over this shard's visible supersurfels with the current increment; sums[29]. */
int ssf_stage_icp_accumulate(ssf_handle* h, int64_t* sums);
/* Host Gauss-Newton step on the (rank-reduced) record; *again = 1 while iterations remain. */
int ssf_stage_icp_update(ssf_handle* h, const int64_t* sums, int* again);
int ssf_stage_icp_end(ssf_handle* h, int* valid);
/* Projective association of this shard's visible supersurfels.  best[S]: packed
 * (dist_bits << 32 | global_id), SSF_NO_MATCH = none; matched[S] as findBestMatches sets it. */
int ssf_stage_match(ssf_handle* h, uint64_t* best, uint8_t* matched);
/* update winners owned by this shard, insert owned unmatched frame surfels, classify, reorder. */
int ssf_stage_fuse(ssf_handle* h, const uint64_t* best, const uint8_t* matched,
                   ssf_frame_result* out);

/* Fusion in two halves around the exchange of rows between shards ("halo exchange": a supersurfel belongs to the rank
 * that owns the world tile of its position, cfg.shard_tile; updateSupersurfels moves the fused position,
 * supersurfel_fusion_kernels.cu:601-682, so an updated row can cross a tile edge):
 *   ssf_stage_fuse_begin  update of the winners this shard owns + ordered insertion of the unmatched frame
 *                         supersurfels it owns.  An updated row whose new position hashes to another rank -- and that this
 *                         frame's filterModel keeps -- leaves the shard: it is written to slot f (the frame supersurfel
 *                         that updated it) of `table`, SSF_MIGRANT_WORDS int32 words per slot, S slots, all other slots
 *                         zero.  Slot layout: [0] destination rank + 1 (0 = empty), [1] 0, [2..27] the row in the
 *                         reference's layout (position 3, colour 3, stamps 2, orientation 9 row-major, shape 6, dims 2,
 *                         confidence 1; floats as their bits).  A frame supersurfel updates at most ONE model row in the
 *                         whole map, so an int32 SUM all-reduce of the tables over the ranks is exactly their union.
 *   ssf_stage_fuse_end    rows addressed to this rank in the (reduced) table are appended behind this frame's
 *                         insertions in ascending f, then classification (filterModel) of every row and the stable
 *                         reorder.  table = NULL: nothing arrives.  Per-shard n_removed does not count emigrants and
 *                         n_inserted does not count immigrants, so the sums over the ranks are the unsharded counters.
 * ssf_stage_fuse = begin (no migration: every row stays where it is) + end.  The native RCCL mode
 * (ssf_comm_attach) always migrates.  After every frame each row then lives on the rank that owns its tile. */
#define SSF_MIGRANT_WORDS 28
int ssf_stage_fuse_begin(ssf_handle* h, const uint64_t* best, const uint8_t* matched, int32_t* table);
int ssf_stage_fuse_end(ssf_handle* h, const int32_t* table, ssf_frame_result* out);

/* Device-resident variants for the multi-GPU driver: the exchanged records stay in HBM, where the
 * RCCL collectives run on them (the buffers are the caller's, e.g. torch tensors; for the CPU
 * checker "device" pointers are host pointers).  All work is enqueued on the handle's stream
 * (cfg.stream), which must be the stream the caller's collectives are ordered on.
 *   ssf_stage_begin_submitted    the oldest frame submitted with ssf_submit_frame becomes the
 *                                current frame of the stage calls (extract ran ahead)
 *   ssf_stage_icp_accumulate_device  as ssf_stage_icp_accumulate, record -> d_sums[29], no wait
 *   ssf_stage_icp_fetch          bring a (reduced) device record to the host: sums[29]
 *   ssf_stage_match_device       association, tables -> d_best[S] / d_matched[S], no wait
 *   ssf_stage_fuse_device        as ssf_stage_fuse with the (reduced) tables taken from HBM */
int ssf_stage_begin_submitted(ssf_handle* h);
int ssf_stage_icp_accumulate_device(ssf_handle* h, int64_t* d_sums);
int ssf_stage_icp_fetch(ssf_handle* h, const int64_t* d_sums, int64_t* sums);
int ssf_stage_match_device(ssf_handle* h, uint64_t* d_best, uint8_t* d_matched);
int ssf_stage_fuse_device(ssf_handle* h, const uint64_t* d_best, const uint8_t* d_matched,
                          ssf_frame_result* out);
/*   ssf_stage_fuse_begin_device / _end_device   the two halves above with the migrant table in HBM (d_table:
 *                                SSF_MIGRANT_WORDS * S int32, the caller's buffer: the all-reduce runs on it in between) */
int ssf_stage_fuse_begin_device(ssf_handle* h, const uint64_t* d_best, const uint8_t* d_matched, int32_t* d_table);
int ssf_stage_fuse_end_device(ssf_handle* h, const int32_t* d_table, ssf_frame_result* out);

/* ---- loop-closure registration -----------------------------------
 * DenseRegistration::align (dense_registration.cu:52-243; makeCorrespondences,
 * dense_registration_kernels.cu:27-100; buildSymmetricPoint2PlaneSystem,
 * dense_registration_kernels.cuh:87-173): register `n` source supersurfels (a fern keyframe's
 * positions / colours / orientations in the keyframe's camera frame, host arrays in the layout of
 * ssf_surfels; confidences NULL = all 1, as closeGlobalLoop passes, supersurfel_fusion.cu:777-797)
 * against the CURRENT frame of the handle (the last processed / extracted one: its supersurfels,
 * index map and plane depth), starting from init_pose (R_init, t_init: source -> current camera,
 * the PnP result; NULL = identity).  cfg.icp_iter iterations, centroid- and scale-normalised
 * symmetric point-to-plane system, host LDLT step; no early stop.
 * rel_pose[12] = (R, t) as the reference returns them (transpose(R_inc), -R t_inc); *valid as the
 * reference's return value (>= 100 pairs every iteration, covariance diagonal <= icp_cov_thresh,
 * |t_inc| <= 0.3); *iters = executed iterations; pairs_last = pairs of the last iteration. */
int ssf_align(ssf_handle* h, const ssf_surfels* source, int n, const float* init_pose,
              float* rel_pose, int* valid, int* iters, int* pairs_last);

/* Fern encoding (computeCodes_kernel, ferns_kernels.cu:48-70): 4-bit code per fern from an RGB
 * image (H x W x 3 u8) and a depth image (H x W f32) given by the caller (the reference feeds a
 * pyramid level built with cv::cuda::resize, third party).  fern_pos: 2*n u32 (x, y),
 * fern_rgb: 3*n u8, fern_depth: n f32; codes: n bytes (bit0 r>, bit1 g>, bit2 b>, bit3 depth>). */
int ssf_fern_codes(ssf_handle* h, const uint8_t* rgb, const float* depth, int width, int height,
                   const uint32_t* fern_pos, const uint8_t* fern_rgb, const float* fern_depth,
                   int n, uint8_t* codes);

/* ---- multi-GPU, native: RCCL on the handle's stream --------------------------------------------
 * One process per GPU, cfg.rank / cfg.nranks / cfg.shard_tile describe this handle's shard of the
 * map.  Rank 0 calls ssf_comm_unique_id and ships the 128 bytes to the other ranks out of band
 * (e.g. a torch.distributed broadcast); every rank then calls ssf_comm_attach.  From then on
 * ssf_process_frame* / ssf_process_submitted run the exchange steps themselves, all in HBM on the
 * track stream: SUM all-reduce of the 29 x int64 ICP record per iteration, MIN / MAX all-reduce of
 * the association tables, SUM all-reduce of the migrant table (rows that crossed a tile edge move to their new
 * owner, see ssf_stage_fuse_begin), one all-gather of the shard sizes per frame (read lazily at the start of
 * the next frame).  All ranks must process the same frames in the same order.
 * ssf_get_global_counts: (n_model, n_visible, n_removed, n_inserted, n_updated) of the last frame
 * summed over the ranks.  The CPU checker exports these symbols and returns SSF_ERR_DEVICE (it has
 * no RCCL; the Python driver supersurfel_fusion_amd/sharded.py runs the same protocol over any
 * torch.distributed backend through the stage seams). */
int ssf_comm_unique_id(uint8_t* id128);
int ssf_comm_attach(ssf_handle* h, const uint8_t* id128);
/* The extract stage DEALT over the ranks instead of replicated: with mode 1, batch j of
 * the frame stream is extracted by rank j % nranks alone, which broadcasts each frame's label map + plane depth + frame
 * supersurfels (ssf_submit_frame_tables' three quantities, 2.5 MB per 640 x 480 frame) to the others over a communicator of
 * the batch context (ncclCommSplit of the attached one; ncclBroadcast on the context's own stream, behind its extract chain
 * and ahead of the track chain: off the critical path).  Each rank then runs 1 / nranks of the extract work; results are
 * bit-identical.  Collective: every rank calls it after ssf_comm_attach, with an empty pipeline; every rank must then
 * be handed the same frame stream in the same batches (the images of a batch another rank extracts are not read).  mode 0
 * = replicated again; mode 2 = as 1, and the extracting rank rebuilds its own tables from what it ships (a self-check).
 * RCCL backend only.  EXPERIMENTAL: like the rest of the native N > 1 path it has run on ONE rank only on this build's hardware
 * (no test with two ranks exists).  While a handle deals, ssf_submit_frame_tables is refused (SSF_ERR_STATE): the two ways of
 * receiving a frame extracted elsewhere do not mix.  A rank whose own extract launch fails still issues its broadcasts (the
 * peers do not hang) and returns the error. */
int ssf_comm_deal_extract(ssf_handle* h, int mode);
/* what is attached: *backend = 0 none, 1 RCCL, 2 peer-to-peer regions; *ranks = the number of ranks the exchange
 * itself reports (ncclCommCount of the communicator; the number of opened regions + 1), 1 when nothing is attached;
 * *my_rank likewise (ncclCommUserRank).  A launcher prints these next to cfg.nranks: they must agree. */
int ssf_comm_info(ssf_handle* h, int* backend, int* ranks, int* my_rank);
int ssf_get_global_counts(ssf_handle* h, int64_t* out5);

/* ---- multi-GPU, native: peer to peer over xGMI, no collective launches ---------------------------
 * EXPERIMENTAL until a run on two GPUs has passed tests/test_p2p_gpu.py bit for bit: every test so far had all ranks on
 * ONE GPU (threads of one process, and separate processes through IPC handles); no store has crossed xGMI yet.
 * The latency-tuned exchange SURVEY.md section 5 / 8e proposes for the ranks of ONE node (cfg.nranks <= 8): every
 * handle owns an exchange region in its HBM; a rank stores its small records (29 x int64 ICP record per iteration,
 * association tables, migrant table, shard sizes) straight into the other ranks' regions and sums what arrives in
 * its own, in rank order (integers: the result is the SUM / MIN / MAX all-reduce of ssf_comm_attach, bit for bit).
 * An ICP iteration is ONE launch whose last workgroup does the exchange -- as on a single GPU -- instead of kernel +
 * all-reduce + publication.
 *   ssf_p2p_export        allocates the region and returns its 64-byte IPC handle (hipIpcGetMemHandle); ship it to the
 *                         other ranks out of band (e.g. a torch.distributed all_gather of 64 bytes)
 *   ssf_p2p_attach        handles = nranks x 64 bytes in rank order (the own entry is ignored): opens the peers'
 *                         regions (hipIpcOpenMemHandle) and switches ssf_process_frame* / ssf_process_submitted to the
 *                         exchange protocol.  Mutually exclusive with ssf_comm_attach.
 *   ssf_p2p_region / ssf_p2p_attach_local   the same for handles that live in ONE process (several shards on one
 *                         GPU, or several GPUs driven by one process with peer access enabled): regions[r] = the
 *                         pointer ssf_p2p_region returned for rank r.  Each rank must then be driven by its own
 *                         host thread: a frame call returns only when every peer has made the same call (and every
 *                         rank's stream needs a hardware queue of its own while it waits: the runtime provides four
 *                         per priority level; with more, the ranks time-slice).
 * All ranks must process the same frames in the same order.  A peer that never arrives makes the waiting call fail
 * with SSF_ERR_DEVICE after a bounded wait (ssf_p2p_configure's timeout, wall clock); it does not hang the device.  The CPU checker exports these
 * symbols and returns SSF_ERR_DEVICE. */
#define SSF_P2P_HANDLE_BYTES 64
/* Optional, BEFORE ssf_p2p_export / ssf_p2p_region (the region is allocated there):
 *   all_ranks_on_this_device  0 (default): peers live on other GPUs -- the region is FINE-GRAINED device memory
 *                             (hipDeviceMallocFinegrained), the only kind HIP keeps coherent while a kernel that polls it is
 *                             running and another device stores into it; 1: every rank of the map is a handle on THIS
 *                             handle's GPU (several shards on one GPU): plain device memory.
 *   timeout_s                 wall-clock bound (default 30 s) of every wait for a peer inside a kernel; also settable after
 *                             the attach.  Ranks may reach their first exchange this far apart. */
int ssf_p2p_configure(ssf_handle* h, int all_ranks_on_this_device, double timeout_s);
int ssf_p2p_export(ssf_handle* h, uint8_t* handle64);
int ssf_p2p_attach(ssf_handle* h, const uint8_t* handles);
int ssf_p2p_region(ssf_handle* h, void** region, size_t* bytes);
int ssf_p2p_attach_local(ssf_handle* h, void* const* regions);

/* ---- read back -------------------------------------------------------------------------------- */
int ssf_get_pose(const ssf_handle* h, float* pose12);
int ssf_set_pose(ssf_handle* h, const float* pose12);
int ssf_get_counts(const ssf_handle* h, int* n_model, int* n_visible, int* stamp, int* n_superpixels);
/* Copy-out of model rows [first, first+count) / the S frame supersurfels into caller arrays. */
int ssf_get_model(ssf_handle* h, int first, int count, ssf_surfels* out);
int ssf_get_frame(ssf_handle* h, ssf_surfels* out);
/* Replace the model (import / synthetic seeding): rows [0,n), first n_visible rows visible. */
int ssf_set_model(ssf_handle* h, const ssf_surfels* in, int n, int n_visible, int stamp);
int ssf_get_index_map(ssf_handle* h, int32_t* out /* H*W */);
int ssf_get_boundary_map(ssf_handle* h, int32_t* out /* H*W */);
int ssf_get_inlier_map(ssf_handle* h, uint8_t* out /* H*W */);
int ssf_get_plane_depth(ssf_handle* h, float* out /* H*W */);
/* SuperpixelRGBD table (TPS_RGBD.hpp:32-37) as 9 floats per superpixel:
 * cx, cy, r, g, b, theta_a, theta_b, theta_c, size. */
int ssf_get_superpixels(ssf_handle* h, float* out /* 9*S */);
/* computeSuperpixelSegIm (supersurfel_fusion.hpp:79, supersurfel_fusion.cu:635-640 -> TPS_RGBD::computePreviewImage,
 * TPS_RGBD.cu:527-541, renderBoundaryImage_kernel TPS_RGBD_kernels.cu:616-644): the CV_8UC3 preview of the
 * segmentation of the last frame -- white where the label of the right or lower-right neighbour differs, otherwise
 * 0.8 x the pixel's colour, channels in the order the reference writes them (B, G, R).
 * computeSlantedPlaneIm (supersurfel_fusion.hpp:80, :642-647) is ssf_get_plane_depth: the CV_32FC1 plane-rendered depth. */
int ssf_get_preview_image(ssf_handle* h, uint8_t* out /* H*W*3 */);
/* Device-resident view of the whole model in the reference's order [visible | out-of-view] and in the reference's
 * layout (getModel() returns device vectors which the node copies out array by array,
 * supersurfel_fusion.hpp:87, supersurfel_fusion_node.cpp:306-310): n_model rows, device pointers in ssf_surfels,
 * `orientations` = packed row-major Mat33 (9 floats per supersurfel), as supersurfels.hpp:37.  The product keeps the
 * two visibility classes in separate stores and the three rows of the Mat33 as separate streams (DESIGN.md section
 * 3); this call materialises the dense, packed copy on demand.  Valid until the next call on the handle.  (For the
 * CPU checker "device" pointers are host pointers.) */
int ssf_get_model_device(ssf_handle* h, ssf_surfels* out_device_ptrs, int* n_model);
/* getFrame() as the reference returns it (supersurfel_fusion.hpp:86; the nodes copy its arrays out whole,
 * supersurfel_fusion_node.cpp:423-427): the frame supersurfels of the last processed frame as device arrays in the reference's
 * layout (orientations packed as Mat33), *n = the number of superpixels; valid until the next call on the handle. */
int ssf_get_frame_device(ssf_handle* h, ssf_surfels* out_device_ptrs, int* n_frame);
int ssf_export_model_txt(ssf_handle* h, const char* path);

/* ---- "next" row: depth pre-filter ----------------------------------------------------------- */
/* Out-of-place restatement of cv::cuda::bilateralFilter(depth, depth, -1, sigma_color, sigma_space)
 * (supersurfel_fusion.cu:180; OpenCV 3.4 cudaimgproc, third party): kernel radius = round(1.5 *
 * sigma_space), circular support, BORDER_REFLECT_101.  depth_in / depth_out: H*W float32. */
int ssf_bilateral_filter(ssf_handle* h, const void* depth_in, void* depth_out, int on_device);

/* ---- "next" row: loop-closure deformation apply ---------------------------------------------- */
/* nodes: positions 3*m, rotations 9*m (row-major), translations 3*m; per model surfel 4 weights
 * and 4 node indices (deformation_graph_kernels.cu:27-73).  Applies to rows [0, n_model). */
int ssf_apply_deformation(ssf_handle* h, const float* node_positions, const float* node_rotations,
                          const float* node_translations, int n_nodes, const float* weights4,
                          const int32_t* idx4);

/* ---- sharded maps: re-homing after positions changed outside a frame -------------------------------------------
 * ssf_apply_deformation moves EVERY row (applyDeformation, deformation_graph_kernels.cu:27-73); the per-frame migration
 * (ssf_stage_fuse_begin / _end, or inside ssf_process_frame with an exchange backend attached) only moves rows a frame has
 * updated.  After a loop closure on a sharded map, one sweep of these two calls puts every row back on the rank that owns
 * the world tile of its position (a rare, bulk operation; the transport between the ranks is the caller's --
 * supersurfel_fusion_amd/sharded.py does it over torch.distributed):
 *   ssf_rehome_begin   the valid rows whose position belongs to another rank's tile leave this shard, in logical order, as
 *                      records of SSF_MIGRANT_WORDS int32 words in the caller's `table` (room for table_rows records): word 0
 *                      = destination rank + 1, word 1 = 1 when the row sat in the shard's visible block, words 2..27 = the
 *                      row (layout of the migrant table above).  *n_out = their number.  SSF_ERR_CAPACITY (nothing changed)
 *                      when table_rows records do not suffice.
 *   ssf_rehome_end     `table` holds n records -- the tables of all ranks one after the other in rank order, or any selection
 *                      in that order that contains every record addressed to this rank: those are appended, the ones
 *                      flagged visible behind the visible block, the others behind the out-of-view rows.  Returns 0, a
 *                      negative ssf_status, or the POSITIVE number of arrivals a full shard had to turn away (in table
 *                      order; their source shards have already let them go, so they are lost to the map -- the same rule
 *                      as an arrival at a full shard inside a frame, which counts as removed).  Never fails for lack of
 *                      room: an error on one rank after the others have committed could not be rolled back.
 * No frame may be pending in the extract pipeline.  Both calls are no-ops for an unsharded handle. */
int ssf_rehome_begin(ssf_handle* h, int32_t* table, int table_rows, int* n_out);
int ssf_rehome_end(ssf_handle* h, const int32_t* table, int n);

/* ---- measurement ------------------------------------------------------------------------------ */
/* With cfg.profile = 1: per-kernel accumulated hipEvent time since the last reset.
 * names: up to max_k C strings (library-owned), ms / calls arrays of max_k.  Returns count. */
int ssf_get_kernel_times(ssf_handle* h, const char** names, double* ms, int64_t* calls, int max_k);
int ssf_reset_kernel_times(ssf_handle* h);
/* Change cfg.profile at run time (0, 1 or 2). */
int ssf_set_profile(ssf_handle* h, int enable);
/* Where the last ssf_process_sequence went (the library's own clock, microseconds from the call's entry; the first 64 frames):
 *   ssf_sequence_times  out64[k] = frame k's results complete
 *   ssf_sequence_marks  out320: [0..63] track loop entered, [64..127] first ICP record back, [128..191] ICP loop done,
 *                       [192..255] counters back; then 32 x (launch time of an extract batch, frames + host microseconds / 1e4)
 * bench.py's `pipeline_fill` is these.  The CPU checker returns zeros. */
int ssf_sequence_times(ssf_handle* h, double* out64);
int ssf_sequence_marks(ssf_handle* h, double* out320);
/* What a plain 16-bytes-per-lane stream copy sustains on this box, GB/s (mib MiB read + the same written, best of reps;
 * non-temporal, unrolled): the "measured-achievable" HBM figure beside the 8 TB/s spec.  mib is rounded
 * down to a multiple of 256 (every form of the copy moves whole rounds of 256 MiB); mib < 256 or reps < 1 returns -1.  < 0: n/a. */
double ssf_stream_copy_rate(int mib, int reps);
/* Counters of the host-side machinery (tests and tools/ read them; none is on the frame path):
 *   ssf_upload_stats          out6: [0] upload workers, [1] host frames uploaded, microseconds summed over the workers [2] waiting
 *                             for a ring slot, [3] in the staging memcpy, [4] in the copy enqueues, [5] the caller's wait for uploads
 *   ssf_pooled_streams        streams of destroyed handles waiting in the process-wide pool for the next handle
 *   ssf_waiter_matches        frames whose association ran inside an ICP launch that was waiting for the host's word
 *   ssf_waiter_match_repairs  ... and the ones re-run as a launch because that word came too late to be trusted
 *   ssf_tuner_state           out4: [0] 1 when the next frame's first ICP iteration is being accumulated by the row-move kernel
 *                             (0: by a launch of its own) -- the library measures which is faster on the running workload,
 *                             results are identical --, [1] pipelined frames seen, [2] / [3] the two forms' last measured means */
int ssf_upload_stats(ssf_handle* h, double* out6);
int ssf_tuner_state(ssf_handle* h, double* out4);
int ssf_pooled_streams(void);
long long ssf_waiter_matches(ssf_handle* h);
long long ssf_waiter_match_repairs(ssf_handle* h);

#ifdef __cplusplus
}
#endif
#endif /* SSF_H */
