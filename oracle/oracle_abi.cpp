// oracle_abi.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Exports the C ABI of include/ssf.h on top of the CPU restatement so that the same host code
// (tests, bench cpu_baseline leg) can drive the checker exactly like the HIP product.
// Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load this library.
#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <deque>
#include <vector>
#include <new>
#include <string>
#include "oracle.h"
#ifdef _OPENMP
#include <omp.h>
#endif

using namespace orc;
#ifdef SSF_ORACLE_ARMS
#include <xmmintrin.h>
namespace orc { StudyArms g_arms = {0, 0, 0, 0, 0, 0}; }
#endif
namespace orc { int g_oracle_rsqrt_ulp = 0; }       // (test hook of oracle_math.h normalize(): see ssf_oracle_set_rsqrt_ulp below)

struct PendingFrame {
    std::vector<uint8_t> rgb; std::vector<float> depth; std::vector<uint8_t> mask; bool has_mask = false;
    // a frame extracted elsewhere (ssf_submit_frame_tables): label map, plane depth, the S frame supersurfels
    bool tables = false; std::vector<int32_t> label; std::vector<float> plane_depth; Surfels frame;
};
struct ssf_handle { State s; std::deque<PendingFrame> pending; bool fusing = false; };
static std::string g_create_err;

static void pose_to12(const Pose& p, float* o) {
    for (int i = 0; i < 3; i++) { o[3 * i] = p.R.r[i].x; o[3 * i + 1] = p.R.r[i].y; o[3 * i + 2] = p.R.r[i].z; }
    o[9] = p.t.x; o[10] = p.t.y; o[11] = p.t.z;
}

extern "C" {

int ssf_abi_version(void) { return SSF_ABI_VERSION; }
const char* ssf_backend_name(void) { return "cpu-oracle"; }

void ssf_default_config(ssf_config* c) {      // supersurfel_fusion.hpp:46-74 default arguments
    std::memset(c, 0, sizeof(*c));
    c->width = 640; c->height = 480; c->fx = 525.f; c->fy = 525.f; c->cx = 319.5f; c->cy = 239.5f;
    c->cell_size = 16; c->lambda_pos = 50.f; c->lambda_bound = 1000.f; c->lambda_size = 10000.f;
    c->lambda_disp = 1e6f; c->thresh_disp = 1e-4f; c->seg_iter = 10; c->seg_use_ransac = 1;
    c->nb_samples = 16; c->filter_iter = 4; c->filter_alpha = 0.1f; c->filter_beta = 1.0f;
    c->filter_threshold = 0.05f; c->range_min = 0.2f; c->range_max = 5.0f; c->delta_t = 20;
    c->conf_thresh = 2500.f; c->nb_supersurfels_max = 50000; c->icp_iter = 10; c->icp_cov_thresh = 0.04;
    c->rng_seed = 1234; c->icp_force_iters = 0; c->device_id = 0; c->stream = nullptr;
    c->rank = 0; c->nranks = 1; c->shard_tile = 0.5f; c->profile = 0;
    c->depth_prefilter = 1; c->prefilter_sigma_color = 0.03f; c->prefilter_sigma_space = 4.5f;
    c->pipeline_depth = 0; c->extract_batch = 1;
}

int ssf_create(const ssf_config* cfg, ssf_handle** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return SSF_ERR_INVALID_ARG; }
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->cell_size <= 0 || cfg->nb_samples <= 0 ||
        cfg->nb_supersurfels_max <= 0 || cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks) {
        g_create_err = "invalid configuration"; return SSF_ERR_INVALID_ARG;
    }
    ssf_handle* h = new (std::nothrow) ssf_handle();
    if (!h) return SSF_ERR_DEVICE;
    State& s = h->s;
    s.cfg = *cfg;
    s.W = cfg->width; s.H = cfg->height;
    s.gx = (s.W + cfg->cell_size - 1) / cfg->cell_size;     // TPS_RGBD.cu:113-116
    s.gy = (s.H + cfg->cell_size - 1) / cfg->cell_size;
    s.S = s.gx * s.gy;
    if (cfg->nb_supersurfels_max < s.S) { delete h; g_create_err = "nb_supersurfels_max < nbSuperpixels"; return SSF_ERR_INVALID_ARG; }
    const size_t P = (size_t)s.W * s.H;
    s.rgba.assign(P, 0); s.disp.assign(P, 0.f); s.plane_depth.assign(P, 0.f);
    s.label.assign(P, 0); s.label_tmp.assign(P, 0); s.inlier.assign(P, 0);
    s.sums.resize(s.S); s.sp.resize(s.S);
    s.samples.assign((size_t)s.S * cfg->nb_samples * 4, 0.f);
    s.frame.resize(s.S); s.frame.zero(s.S); s.frame_lab.assign(s.S, mk3(0, 0, 0));
    s.model.resize(cfg->nb_supersurfels_max); s.model.zero(cfg->nb_supersurfels_max);
    s.model_lab.assign(cfg->nb_supersurfels_max, mk3(0, 0, 0));
    s.pose.R = identity33(); s.pose.t = mk3(0, 0, 0);
    *out = h;
    return SSF_OK;
}
void ssf_destroy(ssf_handle* h) { delete h; }
const char* ssf_last_error(const ssf_handle* h) { return h ? h->s.err.c_str() : g_create_err.c_str(); }

static void frame_lab_refresh(State& s) {
    for (int k = 0; k < s.S; k++) s.frame_lab[k] = rgbToLab(s.frame.col[k]);
}
static void install_frame_tables(State& s, PendingFrame& f);

int ssf_stage_extract(ssf_handle* h, const void* rgb, const void* depth, int on_device, const uint8_t* mask) {
    (void)on_device;
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    extract(h->s, (const uint8_t*)rgb, (const float*)depth, mask);
    frame_lab_refresh(h->s);
    return SSF_OK;
}
int ssf_debug_recentre(ssf_handle* h) { return h ? SSF_OK : SSF_ERR_INVALID_ARG; }
long long ssf_debug_recentre_count(const ssf_handle* h) { return h ? 0 : -1; }
int ssf_debug_set_max_passes(ssf_handle* h, int n) { if (!h) return SSF_ERR_INVALID_ARG; h->s.max_passes = n; return SSF_OK; }
int ssf_debug_set_bin_min_rows(ssf_handle* h, int n) { (void)n; return h ? SSF_OK : SSF_ERR_INVALID_ARG; }     // (a layout choice of the product; nothing to do here)
int ssf_stage_set_shard(ssf_handle* h, int64_t off, int64_t gm, int64_t gv) {
    if (!h) return SSF_ERR_INVALID_ARG;
    h->s.id_offset = off; h->s.global_n_model = gm; h->s.global_n_visible = gv; return SSF_OK;
}
int ssf_stage_icp_begin(ssf_handle* h, const float* prior) {
    if (!h || !h->s.have_frame) return SSF_ERR_STATE;
    icp_begin(h->s, prior); return SSF_OK;
}
int ssf_stage_icp_accumulate(ssf_handle* h, int64_t* sums) {
    if (!h || !sums) return SSF_ERR_INVALID_ARG;
    icp_accumulate(h->s, sums); return SSF_OK;
}
int ssf_stage_icp_update(ssf_handle* h, const int64_t* sums, int* again) {
    if (!h || !sums || !again) return SSF_ERR_INVALID_ARG;
    icp_update(h->s, sums, again); return SSF_OK;
}
int ssf_stage_icp_end(ssf_handle* h, int* valid) {
    if (!h || !valid) return SSF_ERR_INVALID_ARG;
    icp_end(h->s, valid); return SSF_OK;
}
int ssf_stage_match(ssf_handle* h, uint64_t* best, uint8_t* matched) {
    if (!h || !best || !matched) return SSF_ERR_INVALID_ARG;
    if (!h->s.have_frame) return SSF_ERR_STATE;
    match(h->s, best, matched); return SSF_OK;
}
int ssf_stage_fuse(ssf_handle* h, const uint64_t* best, const uint8_t* matched, ssf_frame_result* out) {
    if (!h || !best || !matched) return SSF_ERR_INVALID_ARG;
    if (!h->s.have_frame) return SSF_ERR_STATE;
    fuse(h->s, best, matched, out); return SSF_OK;
}

int ssf_align(ssf_handle* h, const ssf_surfels* src, int n, const float* init_pose, float* rel_pose, int* valid, int* iters, int* pairs_last) {
    if (!h || !src || n < 0 || !rel_pose || !valid || !src->positions || !src->colors || !src->orientations) return SSF_ERR_INVALID_ARG;
    std::vector<f3> pos(n), col(n); std::vector<Mat33> ori(n);
    for (int i = 0; i < n; i++) {
        pos[i] = mk3(src->positions[3 * i], src->positions[3 * i + 1], src->positions[3 * i + 2]);
        col[i] = mk3(src->colors[3 * i], src->colors[3 * i + 1], src->colors[3 * i + 2]);
        for (int r = 0; r < 3; r++) ori[i].r[r] = mk3(src->orientations[9 * i + 3 * r], src->orientations[9 * i + 3 * r + 1], src->orientations[9 * i + 3 * r + 2]);
    }
    *valid = align(h->s, pos.data(), col.data(), ori.data(), src->confidences, n, init_pose, rel_pose, iters, pairs_last) ? 1 : 0;
    return SSF_OK;
}
// computeCodes_kernel, ferns_kernels.cu:48-70
int ssf_fern_codes(ssf_handle* h, const uint8_t* rgb, const float* depth, int width, int height, const uint32_t* fern_pos,
                   const uint8_t* fern_rgb, const float* fern_depth, int n, uint8_t* codes) {
    if (!h || !rgb || !depth || !fern_pos || !fern_rgb || !fern_depth || !codes || width <= 0 || height <= 0 || n < 0) return SSF_ERR_INVALID_ARG;
    for (int k = 0; k < n; k++) {
        const int x = (int)std::min<uint32_t>(fern_pos[2 * k], (uint32_t)width - 1), y = (int)std::min<uint32_t>(fern_pos[2 * k + 1], (uint32_t)height - 1);
        const size_t q = (size_t)y * width + x;                     // point sampling, clamp (texture_impl.hpp:43-46)
        uint8_t r = 0;
        r |= rgb[3 * q] > fern_rgb[3 * k] ? 1 : 0;
        r |= rgb[3 * q + 1] > fern_rgb[3 * k + 1] ? 2 : 0;
        r |= rgb[3 * q + 2] > fern_rgb[3 * k + 2] ? 4 : 0;
        r |= depth[q] > fern_depth[k] ? 8 : 0;
        codes[k] = r;
    }
    return SSF_OK;
}
// the checker has no RCCL: the native multi-GPU entry points report that (sharded.py covers N > 1 on CPU)
int ssf_comm_unique_id(uint8_t* id128) { (void)id128; g_create_err = "the CPU checker has no RCCL"; return SSF_ERR_DEVICE; }
int ssf_comm_attach(ssf_handle* h, const uint8_t* id128) { (void)id128; if (h) h->s.err = "the CPU checker has no RCCL"; return SSF_ERR_DEVICE; }
int ssf_comm_info(ssf_handle* h, int* backend, int* ranks, int* my_rank) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (backend) *backend = 0;
    if (ranks) *ranks = 1;
    if (my_rank) *my_rank = 0;
    return SSF_OK;
}
int ssf_comm_deal_extract(ssf_handle* h, int mode) { (void)mode; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_p2p_export(ssf_handle* h, uint8_t* handle64) { (void)handle64; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_p2p_attach(ssf_handle* h, const uint8_t* handles) { (void)handles; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_p2p_region(ssf_handle* h, void** region, size_t* bytes) { (void)region; (void)bytes; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_p2p_configure(ssf_handle* h, int same_device, double timeout_s) { (void)same_device; (void)timeout_s; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_p2p_attach_local(ssf_handle* h, void* const* regions) { (void)regions; if (h) h->s.err = "the CPU checker has no peers"; return SSF_ERR_DEVICE; }
int ssf_get_global_counts(ssf_handle* h, int64_t* out5) {
    if (!h || !out5) return SSF_ERR_INVALID_ARG;
    out5[0] = h->s.n_model; out5[1] = h->s.n_visible; out5[2] = out5[3] = out5[4] = 0;
    return SSF_OK;
}
// "device" variants: for the checker a device pointer is a host pointer
int ssf_stage_begin_submitted(ssf_handle* h) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (h->pending.empty()) return SSF_ERR_STATE;
    PendingFrame f = std::move(h->pending.front());
    h->pending.pop_front();
    if (f.tables) { install_frame_tables(h->s, f); frame_lab_refresh(h->s); return SSF_OK; }
    return ssf_stage_extract(h, f.rgb.data(), f.depth.data(), 0, f.has_mask ? f.mask.data() : nullptr);
}
int ssf_stage_icp_accumulate_device(ssf_handle* h, int64_t* d_sums) { return ssf_stage_icp_accumulate(h, d_sums); }
int ssf_stage_icp_fetch(ssf_handle* h, const int64_t* d_sums, int64_t* sums) {
    if (!h || !d_sums || !sums) return SSF_ERR_INVALID_ARG;
    std::memcpy(sums, d_sums, SSF_ICP_RECORD * sizeof(int64_t)); return SSF_OK;
}
int ssf_stage_match_device(ssf_handle* h, uint64_t* d_best, uint8_t* d_matched) { return ssf_stage_match(h, d_best, d_matched); }
int ssf_stage_fuse_begin(ssf_handle* h, const uint64_t* best, const uint8_t* matched, int32_t* table) {
    if (!h || !best || !matched || !table) return SSF_ERR_INVALID_ARG;
    if (!h->s.have_frame) return SSF_ERR_STATE;
    fuse_begin(h->s, best, matched, 1, table); h->fusing = true; return SSF_OK;
}
int ssf_stage_fuse_end(ssf_handle* h, const int32_t* table, ssf_frame_result* out) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (!h->fusing) return SSF_ERR_STATE;
    fuse_end(h->s, table, out); h->fusing = false; return SSF_OK;
}
int ssf_stage_fuse_begin_device(ssf_handle* h, const uint64_t* b, const uint8_t* m, int32_t* t) { return ssf_stage_fuse_begin(h, b, m, t); }
int ssf_stage_fuse_end_device(ssf_handle* h, const int32_t* t, ssf_frame_result* out) { return ssf_stage_fuse_end(h, t, out); }
int ssf_stage_fuse_device(ssf_handle* h, const uint64_t* d_best, const uint8_t* d_matched, ssf_frame_result* out) {
    return ssf_stage_fuse(h, d_best, d_matched, out);
}

// a frame extracted by another rank becomes the current frame: what extract() leaves behind for ICP / association / fusion
static void install_frame_tables(State& s, PendingFrame& f) {
    s.label = f.label; s.plane_depth = f.plane_depth;
    for (int k = 0; k < s.S; k++) s.frame.copy_row((size_t)k, f.frame, (size_t)k);
    s.extract_ordinal++;                       // (extract()'s epoch guard: the next locally extracted frame draws the next counters)
    s.have_frame = true;
}
static int process_current_or(ssf_handle* h, PendingFrame* tables, const uint8_t* rgb, const float* depth, const float* prior,
                              const uint8_t* mask, ssf_frame_result* out) {
    State& s = h->s;
    using clk = std::chrono::steady_clock;
    auto t0 = clk::now();
    if (tables) install_frame_tables(s, *tables);
    else extract(s, rgb, depth, mask);
    frame_lab_refresh(s);
    auto t1 = clk::now();
    icp_begin(s, prior);
    int again = s.icp.active ? 1 : 0, valid = 0;
    int64_t sums[SSF_ICP_RECORD];
    while (again) { icp_accumulate(s, sums); icp_update(s, sums, &again); }
    icp_end(s, &valid);
    auto t2 = clk::now();
    std::vector<uint64_t> best(s.S); std::vector<uint8_t> matched(s.S);
    match(s, best.data(), matched.data());
    ssf_frame_result r;
    fuse(s, best.data(), matched.data(), &r);
    auto t3 = clk::now();
    r.stage_ms[0] = std::chrono::duration<float, std::milli>(t1 - t0).count();
    r.stage_ms[1] = std::chrono::duration<float, std::milli>(t2 - t1).count();
    r.stage_ms[2] = std::chrono::duration<float, std::milli>(t3 - t2).count();
    if (out) *out = r;
    return SSF_OK;
}
int ssf_process_frame(ssf_handle* h, const uint8_t* rgb, const float* depth, const float* prior,
                      const uint8_t* mask, ssf_frame_result* out) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    return process_current_or(h, nullptr, rgb, depth, prior, mask, out);
}
int ssf_pipeline_capacity(const ssf_handle* h) {
    if (!h) return 0;
    const int d = h->s.cfg.pipeline_depth, b = h->s.cfg.extract_batch;
    return ((d < 0 ? 0 : (d > SSF_MAX_PIPELINE_DEPTH ? SSF_MAX_PIPELINE_DEPTH : d)) + 1) *
           (b < 1 ? 1 : (b > SSF_MAX_EXTRACT_BATCH ? SSF_MAX_EXTRACT_BATCH : b));
}
int ssf_can_submit(const ssf_handle* h) { return (h && (int)h->pending.size() < ssf_pipeline_capacity(h)) ? 1 : 0; }
// the pipelined form is, by definition, the sequential order: the checker queues copies of the inputs
int ssf_submit_frame(ssf_handle* h, const void* rgb, const void* depth, int /*on_device*/, const uint8_t* mask) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    if ((int)h->pending.size() >= ssf_pipeline_capacity(h)) return SSF_ERR_STATE;
    const size_t P = (size_t)h->s.W * h->s.H;
    PendingFrame f;
    f.rgb.assign((const uint8_t*)rgb, (const uint8_t*)rgb + 3 * P);
    f.depth.assign((const float*)depth, (const float*)depth + P);
    f.has_mask = mask != nullptr;
    if (mask) f.mask.assign(mask, mask + h->s.S);
    h->pending.push_back(std::move(f));
    return SSF_OK;
}
int ssf_submit_frame_tables(ssf_handle* h, const int32_t* label, const float* plane_depth, const ssf_surfels* in, int /*on_device*/) {
    if (!h || !label || !plane_depth || !in) return SSF_ERR_INVALID_ARG;
    if (!in->positions || !in->colors || !in->stamps || !in->orientations || !in->shapes || !in->dims || !in->confidences) return SSF_ERR_INVALID_ARG;
    if ((int)h->pending.size() >= ssf_pipeline_capacity(h)) return SSF_ERR_STATE;
    const size_t P = (size_t)h->s.W * h->s.H; const int S = h->s.S;
    PendingFrame f;
    f.tables = true;
    f.label.assign(label, label + P); f.plane_depth.assign(plane_depth, plane_depth + P);
    f.frame.resize((size_t)S);
    for (int i = 0; i < S; i++) {
        f.frame.pos[i] = mk3(in->positions[3 * i], in->positions[3 * i + 1], in->positions[3 * i + 2]);
        f.frame.col[i] = mk3(in->colors[3 * i], in->colors[3 * i + 1], in->colors[3 * i + 2]);
        f.frame.stamps[2 * i] = in->stamps[2 * i]; f.frame.stamps[2 * i + 1] = in->stamps[2 * i + 1];
        for (int r = 0; r < 3; r++) f.frame.orient[i].r[r] = mk3(in->orientations[9 * i + 3 * r], in->orientations[9 * i + 3 * r + 1], in->orientations[9 * i + 3 * r + 2]);
        const float* c = &in->shapes[6 * i]; f.frame.shape[i] = mkcov(c[0], c[1], c[2], c[3], c[4], c[5]);
        f.frame.dims[2 * i] = in->dims[2 * i]; f.frame.dims[2 * i + 1] = in->dims[2 * i + 1];
        f.frame.conf[i] = in->confidences[i];
    }
    h->pending.push_back(std::move(f));
    return SSF_OK;
}
int ssf_process_submitted(ssf_handle* h, const float* prior, ssf_frame_result* out) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (h->pending.empty()) return SSF_ERR_STATE;
    PendingFrame f = std::move(h->pending.front());
    h->pending.pop_front();
    if (f.tables) return process_current_or(h, &f, nullptr, nullptr, prior, nullptr, out);
    return ssf_process_frame(h, f.rgb.data(), f.depth.data(), prior, f.has_mask ? f.mask.data() : nullptr, out);
}
int ssf_process_sequence(ssf_handle* h, const void* const* rgb, const void* const* depth, int n, int /*on_device*/, ssf_frame_result* out) {
    if (!h || !rgb || !depth || n < 0) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) return SSF_ERR_STATE;
    for (int k = 0; k < n; k++) {
        int rc = ssf_process_frame(h, (const uint8_t*)rgb[k], (const float*)depth[k], nullptr, nullptr, out ? &out[k] : nullptr);
        if (rc) return rc;
    }
    return SSF_OK;
}
int ssf_pending_frames(const ssf_handle* h) { return h ? (int)h->pending.size() : 0; }

int ssf_process_frame_device(ssf_handle* h, const void* rgb, const void* depth, const float* prior,
                             const uint8_t* mask, ssf_frame_result* out) {
    return ssf_process_frame(h, (const uint8_t*)rgb, (const float*)depth, prior, mask, out);
}

int ssf_get_pose(const ssf_handle* h, float* p) { if (!h || !p) return SSF_ERR_INVALID_ARG; pose_to12(h->s.pose, p); return SSF_OK; }
int ssf_set_pose(ssf_handle* h, const float* p) {
    if (!h || !p) return SSF_ERR_INVALID_ARG;
    for (int i = 0; i < 3; i++) h->s.pose.R.r[i] = mk3(p[3 * i], p[3 * i + 1], p[3 * i + 2]);
    h->s.pose.t = mk3(p[9], p[10], p[11]); return SSF_OK;
}
int ssf_get_counts(const ssf_handle* h, int* nm, int* nv, int* st, int* ns) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (nm) *nm = h->s.n_model;
    if (nv) *nv = h->s.n_visible;
    if (st) *st = h->s.stamp;
    if (ns) *ns = h->s.S;
    return SSF_OK;
}
static void copy_out(const Surfels& S, int first, int count, ssf_surfels* o) {
    for (int i = 0; i < count; i++) {
        const int k = first + i;
        if (o->positions) { o->positions[3 * i] = S.pos[k].x; o->positions[3 * i + 1] = S.pos[k].y; o->positions[3 * i + 2] = S.pos[k].z; }
        if (o->colors) { o->colors[3 * i] = S.col[k].x; o->colors[3 * i + 1] = S.col[k].y; o->colors[3 * i + 2] = S.col[k].z; }
        if (o->stamps) { o->stamps[2 * i] = S.stamps[2 * k]; o->stamps[2 * i + 1] = S.stamps[2 * k + 1]; }
        if (o->orientations) for (int r = 0; r < 3; r++) {
            o->orientations[9 * i + 3 * r] = S.orient[k].r[r].x; o->orientations[9 * i + 3 * r + 1] = S.orient[k].r[r].y; o->orientations[9 * i + 3 * r + 2] = S.orient[k].r[r].z; }
        if (o->shapes) { const Cov3& c = S.shape[k]; float v[6] = {c.xx, c.xy, c.xz, c.yy, c.yz, c.zz}; std::memcpy(&o->shapes[6 * i], v, 24); }
        if (o->dims) { o->dims[2 * i] = S.dims[2 * k]; o->dims[2 * i + 1] = S.dims[2 * k + 1]; }
        if (o->confidences) o->confidences[i] = S.conf[k];
    }
}
int ssf_get_model(ssf_handle* h, int first, int count, ssf_surfels* o) {
    if (!h || !o || first < 0 || count < 0 || first + count > h->s.cfg.nb_supersurfels_max) return SSF_ERR_INVALID_ARG;
    copy_out(h->s.model, first, count, o); return SSF_OK;
}
int ssf_get_frame(ssf_handle* h, ssf_surfels* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; copy_out(h->s.frame, 0, h->s.S, o); return SSF_OK; }
int ssf_set_model(ssf_handle* h, const ssf_surfels* in, int n, int n_visible, int stamp) {
    if (!h || !in || n < 0 || n > h->s.cfg.nb_supersurfels_max || n_visible < 0 || n_visible > n) return SSF_ERR_INVALID_ARG;
    if (!in->positions || !in->colors || !in->stamps || !in->orientations || !in->shapes || !in->dims || !in->confidences) return SSF_ERR_INVALID_ARG;
    State& s = h->s; Surfels& M = s.model;
    for (int i = 0; i < n; i++) {
        M.pos[i] = mk3(in->positions[3 * i], in->positions[3 * i + 1], in->positions[3 * i + 2]);
        M.col[i] = mk3(in->colors[3 * i], in->colors[3 * i + 1], in->colors[3 * i + 2]);
        M.stamps[2 * i] = in->stamps[2 * i]; M.stamps[2 * i + 1] = in->stamps[2 * i + 1];
        for (int r = 0; r < 3; r++) M.orient[i].r[r] = mk3(in->orientations[9 * i + 3 * r], in->orientations[9 * i + 3 * r + 1], in->orientations[9 * i + 3 * r + 2]);
        const float* c = &in->shapes[6 * i]; M.shape[i] = mkcov(c[0], c[1], c[2], c[3], c[4], c[5]);
        M.dims[2 * i] = in->dims[2 * i]; M.dims[2 * i + 1] = in->dims[2 * i + 1];
        M.conf[i] = in->confidences[i];
        s.model_lab[i] = rgbToLab(M.col[i]);
    }
    s.n_model = n; s.n_visible = n_visible; s.stamp = stamp;
    return SSF_OK;
}
int ssf_get_index_map(ssf_handle* h, int32_t* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; std::memcpy(o, h->s.label.data(), h->s.label.size() * 4); return SSF_OK; }
int ssf_get_boundary_map(ssf_handle* h, int32_t* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    State& s = h->s;
    for (int y = 0; y < s.H; y++) for (int x = 0; x < s.W; x++) o[(size_t)y * s.W + x] = boundary_at(s, s.label, x, y);
    return SSF_OK;
}
int ssf_get_inlier_map(ssf_handle* h, uint8_t* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; std::memcpy(o, h->s.inlier.data(), h->s.inlier.size()); return SSF_OK; }
int ssf_get_plane_depth(ssf_handle* h, float* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; std::memcpy(o, h->s.plane_depth.data(), h->s.plane_depth.size() * 4); return SSF_OK; }
int ssf_get_superpixels(ssf_handle* h, float* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    for (int k = 0; k < h->s.S; k++) {
        const Superpixel& sp = h->s.sp[k];
        float v[9] = {sp.cx, sp.cy, sp.r, sp.g, sp.b, sp.ta, sp.tb, sp.tc, sp.size};
        std::memcpy(&o[9 * k], v, 36);
    }
    return SSF_OK;
}
// for the checker a device pointer is a host pointer: its arrays already have the reference's layout
int ssf_get_frame_device(ssf_handle* h, ssf_surfels* o, int* n) {           // (the checker's "device" is the host)
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    Surfels& F = h->s.frame;
    o->positions = &F.pos[0].x; o->colors = &F.col[0].x; o->stamps = F.stamps.data(); o->orientations = &F.orient[0].r[0].x;
    o->shapes = &F.shape[0].xx; o->dims = F.dims.data(); o->confidences = F.conf.data();
    if (n) *n = h->s.S;
    return SSF_OK;
}
int ssf_get_model_device(ssf_handle* h, ssf_surfels* o, int* n) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    Surfels& M = h->s.model;
    static_assert(sizeof(f3) == 12 && sizeof(Mat33) == 36 && sizeof(Cov3) == 24, "SoA rows are packed floats");
    o->positions = &M.pos[0].x; o->colors = &M.col[0].x; o->stamps = M.stamps.data(); o->orientations = &M.orient[0].r[0].x;
    o->shapes = &M.shape[0].xx; o->dims = M.dims.data(); o->confidences = M.conf.data();
    if (n) *n = h->s.n_model;
    return SSF_OK;
}
// renderBoundaryImage_kernel, TPS_RGBD_kernels.cu:616-644 (computePreviewImage, TPS_RGBD.cu:527-541)
int ssf_get_preview_image(ssf_handle* h, uint8_t* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    const State& s = h->s;
    for (int y = 0; y < s.H; y++)
        for (int x = 0; x < s.W; x++) {
            const size_t p = (size_t)y * s.W + x;
            const int index = s.label[p];
            uint8_t* px = &o[3 * p];
            if (x < s.W - 1 && y < s.H - 1 && (s.label[p + 1] != index || s.label[p + s.W + 1] != index)) { px[0] = px[1] = px[2] = 255; continue; }
            const uint32_t c = s.rgba[p];       // texel (R, G, B, 255): the reference writes 0.8 * z, y, x = B, G, R
            px[0] = (uint8_t)(0.8f * (float)((c >> 16) & 255u)); px[1] = (uint8_t)(0.8f * (float)((c >> 8) & 255u));
            px[2] = (uint8_t)(0.8f * (float)(c & 255u));
        }
    return SSF_OK;
}

// exportModel, supersurfel_fusion.cu:595-633 (std::to_string == "%f")
int ssf_export_model_txt(ssf_handle* h, const char* path) {
    if (!h || !path) return SSF_ERR_INVALID_ARG;
    FILE* f = std::fopen(path, "w");
    if (!f) { h->s.err = "cannot open file"; return SSF_ERR_IO; }
    const State& s = h->s; const Surfels& M = s.model;
    for (int i = 0; i < s.n_model; i++) {
        if (!(M.conf[i] > s.cfg.conf_thresh)) continue;
        std::fprintf(f, "%d %d %f\n", M.stamps[2 * i], M.stamps[2 * i + 1], M.conf[i]);
        std::fprintf(f, "%f %f %f\n", M.pos[i].x, M.pos[i].y, M.pos[i].z);
        std::fprintf(f, "%f %f %f\n", M.col[i].x, M.col[i].y, M.col[i].z);
        std::fprintf(f, "%f %f\n", M.dims[2 * i], M.dims[2 * i + 1]);
        const Mat33& o = M.orient[i];
        std::fprintf(f, "%f %f %f %f %f %f %f %f %f\n", o.r[0].x, o.r[0].y, o.r[0].z, o.r[1].x, o.r[1].y, o.r[1].z, o.r[2].x, o.r[2].y, o.r[2].z);
        const Cov3& c = M.shape[i];
        std::fprintf(f, "%f %f %f %f %f %f\n\n", c.xx, c.xy, c.xz, c.yy, c.yz, c.zz);
    }
    std::fclose(f);
    return SSF_OK;
}

int ssf_apply_deformation(ssf_handle* h, const float* np, const float* nr, const float* nt, int m,
                          const float* w4, const int32_t* idx4) {
    if (!h || !np || !nr || !nt || !w4 || !idx4 || m <= 0) return SSF_ERR_INVALID_ARG;
    apply_deformation(h->s, np, nr, nt, m, w4, idx4); return SSF_OK;
}
int ssf_rehome_begin(ssf_handle* h, int32_t* table, int table_rows, int* n_out) {
    if (!h || !n_out || table_rows < 0 || (!table && table_rows > 0)) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) { h->s.err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    const int n = rehome_begin(h->s, table, table_rows);
    if (n < 0) { h->s.err = "ssf_rehome_begin: the table is too small for the rows that leave"; return SSF_ERR_CAPACITY; }
    *n_out = n; return SSF_OK;
}
int ssf_rehome_end(ssf_handle* h, const int32_t* table, int n) {
    if (!h || n < 0 || (!table && n > 0)) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) { h->s.err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    return rehome_end(h->s, table, n);           // >= 0: the number of arrivals a full shard turned away
}
int ssf_bilateral_filter(ssf_handle* h, const void* in, void* out, int on_device) {
    (void)on_device;
    if (!h || !in || !out) return SSF_ERR_INVALID_ARG;
    bilateral_filter((const float*)in, (float*)out, h->s.W, h->s.H, h->s.cfg.prefilter_sigma_color, h->s.cfg.prefilter_sigma_space);
    return SSF_OK;
}
int ssf_get_kernel_times(ssf_handle* h, const char** names, double* ms, int64_t* calls, int max_k) {
    (void)h; (void)names; (void)ms; (void)calls; (void)max_k; return 0;
}
int ssf_reset_kernel_times(ssf_handle* h) { (void)h; return SSF_OK; }
int ssf_set_profile(ssf_handle* h, int enable) { (void)h; (void)enable; return SSF_OK; }
// the product's measurement counters (include/ssf.h "measurement"): the checker has no pipeline, no streams, no uploads
int ssf_sequence_times(ssf_handle* h, double* out64) { if (!h || !out64) return SSF_ERR_INVALID_ARG; for (int i = 0; i < 64; i++) out64[i] = 0.0; return SSF_OK; }
int ssf_sequence_marks(ssf_handle* h, double* out320) { if (!h || !out320) return SSF_ERR_INVALID_ARG; for (int i = 0; i < 320; i++) out320[i] = i >= 256 && !(i & 1) ? -1.0 : 0.0; return SSF_OK; }
double ssf_stream_copy_rate(int mib, int reps) { (void)mib; (void)reps; return -1.0; }
int ssf_upload_stats(ssf_handle* h, double* out6) { if (!h || !out6) return SSF_ERR_INVALID_ARG; for (int i = 0; i < 6; i++) out6[i] = 0.0; return SSF_OK; }
int ssf_pooled_streams(void) { return 0; }
int ssf_tuner_state(ssf_handle* h, double* out4) { if (!h || !out4) return SSF_ERR_INVALID_ARG; for (int i = 0; i < 4; i++) out4[i] = 0.0; return SSF_OK; }
long long ssf_waiter_matches(ssf_handle* h) { return h ? 0 : -1; }
long long ssf_waiter_match_repairs(ssf_handle* h) { return h ? 0 : -1; }

// test hook (oracle only, see oracle_math.h normalize()): 0 = the specification, -2 .. 2 = every reciprocal square root moved
// by that many ulp, 3 = by a pseudo-random number of ulp in [-2, 2]
#ifdef SSF_ORACLE_ARMS
// STUDY ARMS (oracle_math.h): exported only by libssf_oracle_arms*.so.  name: div_ulp | pow_ulp | schedule | tie | insert_rev |
// filter_gs | ftz (flush-to-zero + denormals-are-zero in the calling thread's MXCSR: what --use_fast_math's -ftz=true does)
int ssf_oracle_set_arm(const char* name, int value) {
    const std::string n(name);
    if (n == "div_ulp") orc::g_arms.div_ulp = value;
    else if (n == "pow_ulp") orc::g_arms.pow_ulp = value;
    else if (n == "schedule") orc::g_arms.schedule = value;
    else if (n == "tie") orc::g_arms.tie = value;
    else if (n == "insert_rev") orc::g_arms.insert_rev = value;
    else if (n == "filter_gs") orc::g_arms.filter_gs = value;
    else if (n == "ftz") { unsigned int csr = _mm_getcsr(); csr = value ? (csr | 0x8040u) : (csr & ~0x8040u); _mm_setcsr(csr); }
    else return -1;
    return 0;
}
#endif
int ssf_oracle_set_rsqrt_ulp(int mode) { orc::g_oracle_rsqrt_ulp = (mode >= -2 && mode <= 3) ? mode : 0; return orc::g_oracle_rsqrt_ulp; }

// OpenMP build only (the timed CPU baseline): number of threads of the parallel loops; returns the number in effect
// (1 for the single-threaded checker).  Not part of ssf.h.
int ssf_oracle_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
    return omp_get_max_threads();
#else
    (void)n; return 1;
#endif
}

// test hooks of include/ssf_testing.h
int ssf_dbg_ldlt_solve6(const double* A, const double* b, double* x) { ldlt_solve6(A, b, x); return 0; }
int ssf_dbg_lu_inverse6(const double* A, double* Ainv) { lu_inverse6(A, Ainv); return 0; }
int ssf_dbg_align_increment(const double* JtJ, const double* Jtr, float scale, const float* cs, const float* ct, double* tf16) {
    align_increment(JtJ, Jtr, scale, cs, ct, tf16); return 0;
}
int ssf_dbg_renormalise_d(double* R9) { quat_normalize_rot_d(R9); return 0; }
int ssf_dbg_renormalise_f(float* R9) { quat_normalize_rot_f(R9); return 0; }
int ssf_dbg_gn_increment(const double* X, double* tf) {
    double tran[3] = {X[3], X[4], X[5]}, axis[3] = {X[0], X[1], X[2]};
    const double nrm = std::sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
    const double angle = 0.5 * std::atan(nrm);
    double Rr[9];
    if (nrm == 0.0) { for (int i = 0; i < 9; i++) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    else { for (int i = 0; i < 3; i++) axis[i] /= nrm; angle_axis_to_rot_d(angle, axis, Rr); }
    const double ca = std::cos(angle);
    for (int i = 0; i < 3; i++) tran[i] *= ca;
    double R9[9];
    for (int i = 0; i < 16; i++) tf[i] = 0.0;
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R9[i * 3 + j] = (Rr[i * 3] * Rr[j] + Rr[i * 3 + 1] * Rr[3 + j]) + Rr[i * 3 + 2] * Rr[6 + j];
        tf[i * 4 + 3] = (Rr[i * 3] * tran[0] + Rr[i * 3 + 1] * tran[1]) + Rr[i * 3 + 2] * tran[2];
    }
    quat_normalize_rot_d(R9);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tf[i * 4 + j] = R9[i * 3 + j];
    tf[15] = 1.0;
    return 0;
}

int ssf_dbg_rgb_to_lab(const float* c, float* o) { f3 r = rgbToLab(mk3(c[0], c[1], c[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_lab_to_rgb(const float* c, float* o) { f3 r = labToRgb(mk3(c[0], c[1], c[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_sym_inverse(const float* c, float* o) {
    Cov3 out; const bool ok = inverse(mkcov(c[0], c[1], c[2], c[3], c[4], c[5]), out);
    o[0] = out.xx; o[1] = out.xy; o[2] = out.xz; o[3] = out.yy; o[4] = out.yz; o[5] = out.zz; return ok ? 1 : 0;
}
int ssf_dbg_principal_frame(const float* c, float* vecs, float* vals) {
    Mat33 m; f3 v; eigenDecomposition(mkcov(c[0], c[1], c[2], c[3], c[4], c[5]), m, v, 10);
    for (int r = 0; r < 3; r++) { vecs[3 * r] = m.r[r].x; vecs[3 * r + 1] = m.r[r].y; vecs[3 * r + 2] = m.r[r].z; }
    vals[0] = v.x; vals[1] = v.y; vals[2] = v.z; return 0;
}
int ssf_dbg_connectivity_guard(const int32_t* g) {       // 3 x 3 label patch, row-major -> 1: the centre is a bridge
    const int ring[8] = {g[0], g[1], g[2], g[5], g[8], g[7], g[6], g[3]};
    return ring_unchangeable(g[4], ring) ? 1 : 0;
}
int ssf_dbg_plane_solve(const float* r, float* th) {
    float a = 0, b = 0, c = 0;
    const bool ok = solvePlaneEquations(a, b, c, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]);
    th[0] = a; th[1] = b; th[2] = c; return ok ? 1 : 0;
}

static Mat33 mat_from9(const float* a) { Mat33 m; for (int r = 0; r < 3; r++) m.r[r] = mk3(a[3 * r], a[3 * r + 1], a[3 * r + 2]); return m; }
static void mat_to9(const Mat33& m, float* o) { for (int r = 0; r < 3; r++) { o[3 * r] = m.r[r].x; o[3 * r + 1] = m.r[r].y; o[3 * r + 2] = m.r[r].z; } }
static void cov_to6(const Cov3& s, float* o) { o[0] = s.xx; o[1] = s.xy; o[2] = s.xz; o[3] = s.yy; o[4] = s.yz; o[5] = s.zz; }
int ssf_dbg_sym_square(const float* c, float* o) { cov_to6(square(mkcov(c[0], c[1], c[2], c[3], c[4], c[5])), o); return 0; }
int ssf_dbg_sym_mulv(const float* c, const float* v, float* o) { f3 r = mkcov(c[0], c[1], c[2], c[3], c[4], c[5]) * mk3(v[0], v[1], v[2]); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_mult_abat(const float* R9, const float* c, float* o) { cov_to6(mult_ABAt(mat_from9(R9), mkcov(c[0], c[1], c[2], c[3], c[4], c[5])), o); return 0; }
int ssf_dbg_m3_mul(const float* A9, const float* B9, float* o) { mat_to9(mat_from9(A9) * mat_from9(B9), o); return 0; }
int ssf_dbg_m3_mulv(const float* A9, const float* v, float* o) { f3 r = mat_from9(A9) * mk3(v[0], v[1], v[2]); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_row_mul(const float* v, const float* A9, float* o) {     // float3 * Mat33, matrix_math.cuh:491-496
    const Mat33 t = transpose(mat_from9(A9)); f3 r = t * mk3(v[0], v[1], v[2]); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0;
}
int ssf_dbg_rot_to_quat(const float* R9, float* q4) { rot_to_quat(mat_from9(R9), q4); return 0; }
int ssf_dbg_quat_to_rot(const float* q4, float* R9) { mat_to9(quat_to_rot(q4), R9); return 0; }

}  // extern "C"
