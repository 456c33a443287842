// ssf_host.hip -- host side of the product library: handle, HBM allocation, the per-frame driver
// (the C++ counterpart of SupersurfelFusion::processFrame, core/src/supersurfel_fusion.cu:166-530,
// hot-path parts only): extract contexts (pipelined / batched extract on their own streams and graphs),
// the track chain (ICP loop with the host Gauss-Newton step of core/src/dense_registration.cu:324-421,
// association, fusion, model-store upkeep), the RCCL exchanges of the multi-GPU mode, the loop-closure
// registration, and the C ABI of include/ssf.h.
//
// There is NO CPU fallback here: without a gfx950 device ssf_create fails with SSF_ERR_NO_DEVICE.
#include <algorithm>
#include <atomic>
#include <cfloat>
#include <chrono>
#include <deque>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <new>
#include <string>
#include <condition_variable>
#include <mutex>
#include <thread>
#include <type_traits>
#include <utility>
#include <vector>
#include <dlfcn.h>
#include <unistd.h>
#include <rccl/rccl.h>          // types only: the library is resolved at run time (dlopen), never linked
#include "../../include/ssf.h"
#include "ssf_device.hpp"

using namespace ssf;

// ---- kernel timer (cfg.profile) -------------------------------------------------------------------
namespace ssf {
struct KernelTimer {
    struct Rec { const char* name; hipEvent_t e0, e1; };
    std::vector<Rec> open, pool_free;
    std::vector<Rec> pending;
    std::map<std::string, std::pair<double, long long>> acc;
    const char* cur_name = nullptr; hipEvent_t cur_e0 = nullptr, cur_e1 = nullptr;
    // what an EMPTY (e0, e1) bracket measures on this device: the two event packets themselves (~5 us).  Part of it
    // overlaps with the dispatch when a kernel sits in between: 0.7 x the empty bracket is what makes back-to-back
    // launches agree with rocprofv3's kernel durations (relabelling pass: 14.3 us live vs 14.4 us rocprofv3).
    double bracket_bias_ms = -1.0;
};
static thread_local KernelTimer* g_timer = nullptr;
KernelTimer* current_timer() { return g_timer; }
void set_current_timer(KernelTimer* t) { g_timer = t; }
void timer_begin(KernelTimer* t, const char* name, hipStream_t st) {
    KernelTimer::Rec r;
    if (!t->pool_free.empty()) { r = t->pool_free.back(); t->pool_free.pop_back(); }
    else { (void)hipEventCreate(&r.e0); (void)hipEventCreate(&r.e1); }
    r.name = name;
    (void)hipEventRecord(r.e0, st);
    t->open.push_back(r);
}
void timer_end(KernelTimer* t, hipStream_t st) {
    KernelTimer::Rec r = t->open.back(); t->open.pop_back();
    (void)hipEventRecord(r.e1, st);
    t->pending.push_back(r);
}
static void timer_calibrate(KernelTimer* t, hipStream_t st) {
    if (t->bracket_bias_ms >= 0.0) return;
    hipEvent_t e[2 * 16];
    for (auto& x : e) (void)hipEventCreate(&x);
    for (int i = 0; i < 16; i++) { (void)hipEventRecord(e[2 * i], st); (void)hipEventRecord(e[2 * i + 1], st); }
    (void)hipStreamSynchronize(st);
    double sum = 0; int n = 0;
    for (int i = 4; i < 16; i++) { float ms; if (hipEventElapsedTime(&ms, e[2 * i], e[2 * i + 1]) == hipSuccess) { sum += ms; n++; } }
    for (auto& x : e) (void)hipEventDestroy(x);
    t->bracket_bias_ms = n ? 0.7 * sum / n : 0.0;
}
static void timer_collect(KernelTimer* t) {     // call after a stream sync
    const double bias = t->bracket_bias_ms > 0.0 ? t->bracket_bias_ms : 0.0;
    for (auto& r : t->pending) {
        float ms = 0.f;
        if (hipEventElapsedTime(&ms, r.e0, r.e1) == hipSuccess) { auto& a = t->acc[r.name]; a.first += std::max(0.0, (double)ms - bias); a.second += 1; }
        t->pool_free.push_back(r);
    }
    t->pending.clear();
}
}  // namespace ssf

// ---- host solvers: dependency-free counterparts of the reference's Eigen calls ---------------------
// (LDLT with diagonal pivoting as Eigen::LDLT, partial-pivot LU inverse, Shoemake quaternion
// re-normalisation, Rodrigues rotation; pinned against the reference's vendored Eigen by
// tests/test_solvers.py through the ssf_dbg_* exports below.)
namespace {

void sym6_ldlt_solve(const double* A, const double* b, double* x) {
    const int n = 6;
    double L[36]; int piv[6]; double w[6];
    std::memcpy(L, A, sizeof(L));
    bool all_zero = false;
    for (int k = 0; k < n; k++) {
        int p = k; double pm = std::fabs(L[k * n + k]);
        for (int i = k + 1; i < n; i++) if (std::fabs(L[i * n + i]) > pm) { pm = std::fabs(L[i * n + i]); p = i; }
        piv[k] = p;
        if (p != k) {                       // symmetric row/column exchange on the lower triangle
            for (int j = 0; j < k; j++) std::swap(L[k * n + j], L[p * n + j]);
            for (int i = p + 1; i < n; i++) std::swap(L[i * n + k], L[i * n + p]);
            std::swap(L[k * n + k], L[p * n + p]);
            for (int i = k + 1; i < p; i++) std::swap(L[i * n + k], L[p * n + i]);
        }
        if (k > 0) {
            for (int j = 0; j < k; j++) w[j] = L[j * n + j] * L[k * n + j];
            double s = 0.0;
            for (int j = 0; j < k; j++) s += L[k * n + j] * w[j];
            L[k * n + k] -= s;
            for (int i = k + 1; i < n; i++) {
                double s2 = 0.0;
                for (int j = 0; j < k; j++) s2 += L[i * n + j] * w[j];
                L[i * n + k] -= s2;
            }
        }
        const double d = L[k * n + k];
        const bool ok = std::fabs(d) > 0.0;
        if (k == 0 && !ok) { for (int j = 0; j < n; j++) piv[j] = j; all_zero = true; break; }
        if (ok) for (int i = k + 1; i < n; i++) L[i * n + k] /= d;
    }
    double y[6];
    for (int i = 0; i < n; i++) y[i] = b[i];
    for (int k = 0; k < n; k++) std::swap(y[k], y[piv[k]]);
    if (!all_zero) for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= L[i * n + j] * y[j];
    for (int i = 0; i < n; i++) { const double d = L[i * n + i]; y[i] = (std::fabs(d) > DBL_MIN) ? y[i] / d : 0.0; }
    if (!all_zero) for (int i = n - 1; i >= 0; i--) for (int j = i + 1; j < n; j++) y[i] -= L[j * n + i] * y[j];
    for (int k = n - 1; k >= 0; k--) std::swap(y[k], y[piv[k]]);
    for (int i = 0; i < n; i++) x[i] = y[i];
}

void mat6_inverse_lu(const double* A, double* Ainv) {
    const int n = 6;
    double U[36]; int perm[6];
    std::memcpy(U, A, sizeof(U));
    for (int i = 0; i < n; i++) perm[i] = i;
    for (int k = 0; k < n; k++) {
        int p = k; double pm = std::fabs(U[k * n + k]);
        for (int i = k + 1; i < n; i++) if (std::fabs(U[i * n + k]) > pm) { pm = std::fabs(U[i * n + k]); p = i; }
        if (p != k) { for (int j = 0; j < n; j++) std::swap(U[k * n + j], U[p * n + j]); std::swap(perm[k], perm[p]); }
        for (int i = k + 1; i < n; i++) {
            U[i * n + k] /= U[k * n + k];
            for (int j = k + 1; j < n; j++) U[i * n + j] -= U[i * n + k] * U[k * n + j];
        }
    }
    for (int c = 0; c < n; c++) {
        double y[6];
        for (int i = 0; i < n; i++) y[i] = (perm[i] == c) ? 1.0 : 0.0;
        for (int i = 0; i < n; i++) for (int j = 0; j < i; j++) y[i] -= U[i * n + j] * y[j];
        for (int i = n - 1; i >= 0; i--) { for (int j = i + 1; j < n; j++) y[i] -= U[i * n + j] * y[j]; y[i] /= U[i * n + i]; }
        for (int i = 0; i < n; i++) Ainv[i * n + c] = y[i];
    }
}

template <typename T>
void renormalise_rotation(T* R) {       // Quaternion(R).normalized().toRotationMatrix()
    T q[4];
    T t = (R[0] + R[4]) + R[8];
    if (t > T(0)) {
        t = std::sqrt(t + T(1.0)); q[3] = T(0.5) * t; t = T(0.5) / t;
        q[0] = (R[7] - R[5]) * t; q[1] = (R[2] - R[6]) * t; q[2] = (R[3] - R[1]) * t;
    } else {
        int i = 0;
        if (R[4] > R[0]) i = 1;
        if (R[8] > R[i * 4]) i = 2;
        const int j = (i + 1) % 3, k = (j + 1) % 3;
        t = std::sqrt(((R[i * 4] - R[j * 4]) - R[k * 4]) + T(1.0));
        q[i] = T(0.5) * t; t = T(0.5) / t;
        q[3] = (R[k * 3 + j] - R[j * 3 + k]) * t;
        q[j] = (R[j * 3 + i] + R[i * 3 + j]) * t;
        q[k] = (R[k * 3 + i] + R[i * 3 + k]) * t;
    }
    const T z = ((q[0] * q[0] + q[1] * q[1]) + q[2] * q[2]) + q[3] * q[3];
    if (z > T(0)) { const T nrm = std::sqrt(z); for (int a = 0; a < 4; a++) q[a] = q[a] / nrm; }
    const T tx = T(2) * q[0], ty = T(2) * q[1], tz = T(2) * q[2];
    const T twx = tx * q[3], twy = ty * q[3], twz = tz * q[3];
    const T txx = tx * q[0], txy = ty * q[0], txz = tz * q[0];
    const T tyy = ty * q[1], tyz = tz * q[1], tzz = tz * q[2];
    R[0] = T(1) - (tyy + tzz); R[1] = txy - twz; R[2] = txz + twy;
    R[3] = txy + twz; R[4] = T(1) - (txx + tzz); R[5] = tyz - twx;
    R[6] = txz - twy; R[7] = tyz + twx; R[8] = T(1) - (txx + tyy);
}

void rodrigues(double angle, const double* ax, double* R) {     // AngleAxisd::toRotationMatrix
    const double s = std::sin(angle), c = std::cos(angle);
    const double sx = s * ax[0], sy = s * ax[1], sz = s * ax[2];
    const double ox = (1.0 - c) * ax[0], oy = (1.0 - c) * ax[1], oz = (1.0 - c) * ax[2];
    double m;
    m = ox * ax[1]; R[1] = m - sz; R[3] = m + sz;
    m = ox * ax[2]; R[2] = m + sy; R[6] = m - sy;
    m = oy * ax[2]; R[5] = m - sx; R[7] = m + sx;
    R[0] = ox * ax[0] + c; R[4] = oy * ax[1] + c; R[8] = oz * ax[2] + c;
}

// one Gauss-Newton increment from the solved 6-vector: tf_iter (4x4, row-major)
void gn_increment(const double* X, double* tf_iter) {
    double tran[3] = {X[3], X[4], X[5]}, axis[3] = {X[0], X[1], X[2]};
    const double nrm = std::sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
    const double angle = 0.5 * std::atan(nrm);
    double Rr[9];
    if (nrm == 0.0) { for (int i = 0; i < 9; i++) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    else { for (int i = 0; i < 3; i++) axis[i] /= nrm; rodrigues(angle, axis, Rr); }
    const double ca = std::cos(angle);
    for (int i = 0; i < 3; i++) tran[i] *= ca;
    for (int i = 0; i < 16; i++) tf_iter[i] = 0.0;
    double R9[9];
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++) R9[i * 3 + j] = (Rr[i * 3] * Rr[j] + Rr[i * 3 + 1] * Rr[3 + j]) + Rr[i * 3 + 2] * Rr[6 + j];
        tf_iter[i * 4 + 3] = (Rr[i * 3] * tran[0] + Rr[i * 3 + 1] * tran[1]) + Rr[i * 3 + 2] * tran[2];
    }
    renormalise_rotation<double>(R9);
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tf_iter[i * 4 + j] = R9[i * 3 + j];
    tf_iter[15] = 1.0;
}

// host step of one align iteration (DenseRegistration::align, dense_registration.cu:168-210): as gn_increment, with
// the translation un-scaled and the increment conjugated by the centroid translations:
// T(ct) * Rot * T(tran) * Rot * T(-cs), Eigen Isometry products left to right
void align_increment(const double* JtJ, const double* Jtr, float scale, const float* cs, const float* ct, double* tf_iter) {
    double X[6];
    sym6_ldlt_solve(JtJ, Jtr, X);
    double tran[3] = {X[3], X[4], X[5]}, axis[3] = {X[0], X[1], X[2]};
    const double nrm = std::sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
    const double angle = 0.5 * std::atan(nrm);
    double Rr[9];
    if (nrm == 0.0) { for (int i = 0; i < 9; i++) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0; }
    else { for (int i = 0; i < 3; i++) axis[i] /= nrm; rodrigues(angle, axis, Rr); }
    const double ca = std::cos(angle);
    for (int i = 0; i < 3; i++) { tran[i] /= (double)scale; tran[i] *= ca; }
    double RR[9], t2[3], t4[3];
    for (int i = 0; i < 3; i++) {
        t2[i] = ((Rr[i * 3] * tran[0] + Rr[i * 3 + 1] * tran[1]) + Rr[i * 3 + 2] * tran[2]) + (double)ct[i];
        for (int j = 0; j < 3; j++) RR[i * 3 + j] = (Rr[i * 3] * Rr[j] + Rr[i * 3 + 1] * Rr[3 + j]) + Rr[i * 3 + 2] * Rr[6 + j];
    }
    const double ncs[3] = {-1.0 * (double)cs[0], -1.0 * (double)cs[1], -1.0 * (double)cs[2]};
    for (int i = 0; i < 3; i++) t4[i] = ((RR[i * 3] * ncs[0] + RR[i * 3 + 1] * ncs[1]) + RR[i * 3 + 2] * ncs[2]) + t2[i];
    renormalise_rotation<double>(RR);
    for (int i = 0; i < 16; i++) tf_iter[i] = 0.0;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) tf_iter[i * 4 + j] = RR[i * 3 + j]; tf_iter[i * 4 + 3] = t4[i]; }
    tf_iter[15] = 1.0;
}

void mat4_lmul(const double* a, double* b) {      // b <- a * b
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[i * 4 + j] = ((a[i * 4] * b[j] + a[i * 4 + 1] * b[4 + j]) + a[i * 4 + 2] * b[8 + j]) + a[i * 4 + 3] * b[12 + j];
    std::memcpy(b, r, sizeof(r));
}

}  // namespace

// ---- RCCL, resolved at run time -------------------------------------------------------------------------
// The multi-GPU exchanges (ssf_comm_attach) call RCCL directly on the track stream.  The symbols come from
// the librccl the process already holds (torch ships one, SONAME librccl.so.1) or from /opt/rocm; a box
// without RCCL still loads libssf_hip.so and runs single-GPU.
struct RcclApi {
    void* lib = nullptr;
    decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
    decltype(&ncclCommInitRank) CommInitRank = nullptr;
    decltype(&ncclCommDestroy) CommDestroy = nullptr;
    decltype(&ncclAllReduce) AllReduce = nullptr;
    decltype(&ncclAllGather) AllGather = nullptr;
    decltype(&ncclGetErrorString) GetErrorString = nullptr;
    decltype(&ncclCommCount) CommCount = nullptr;
    decltype(&ncclCommUserRank) CommUserRank = nullptr;
    decltype(&ncclBroadcast) Broadcast = nullptr;           // the three below: only the dealt extract stage needs them (ssf_comm_deal_extract)
    decltype(&ncclCommSplit) CommSplit = nullptr;
    decltype(&ncclGroupStart) GroupStart = nullptr;
    decltype(&ncclGroupEnd) GroupEnd = nullptr;
    std::string err;
};
static RcclApi* rccl_api() {
    static RcclApi api;
    static bool tried = false;
    if (tried) return api.lib ? &api : nullptr;
    tried = true;
    const char* names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_NOLOAD); if (api.lib) break; }   // already in the process?
    if (!api.lib) for (const char* n : names) { api.lib = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (api.lib) break; }
    if (!api.lib) { api.err = "librccl.so.1 not found"; return nullptr; }
    api.GetUniqueId = (decltype(api.GetUniqueId))dlsym(api.lib, "ncclGetUniqueId");
    api.CommInitRank = (decltype(api.CommInitRank))dlsym(api.lib, "ncclCommInitRank");
    api.CommDestroy = (decltype(api.CommDestroy))dlsym(api.lib, "ncclCommDestroy");
    api.AllReduce = (decltype(api.AllReduce))dlsym(api.lib, "ncclAllReduce");
    api.AllGather = (decltype(api.AllGather))dlsym(api.lib, "ncclAllGather");
    api.GetErrorString = (decltype(api.GetErrorString))dlsym(api.lib, "ncclGetErrorString");
    api.CommCount = (decltype(api.CommCount))dlsym(api.lib, "ncclCommCount");
    api.CommUserRank = (decltype(api.CommUserRank))dlsym(api.lib, "ncclCommUserRank");
    api.Broadcast = (decltype(api.Broadcast))dlsym(api.lib, "ncclBroadcast");
    api.CommSplit = (decltype(api.CommSplit))dlsym(api.lib, "ncclCommSplit");
    api.GroupStart = (decltype(api.GroupStart))dlsym(api.lib, "ncclGroupStart");
    api.GroupEnd = (decltype(api.GroupEnd))dlsym(api.lib, "ncclGroupEnd");
    if (!api.GetUniqueId || !api.CommInitRank || !api.CommDestroy || !api.AllReduce || !api.AllGather) {
        api.err = "librccl lacks a required symbol"; api.lib = nullptr; return nullptr;
    }
    return &api;
}

static inline double now_us() { return std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now().time_since_epoch()).count(); }
// ---- upload of host frames ahead of the pipeline (ssf_process_sequence with host buffers) --------------------------
// The caller of the reference hands over host images (cv::Mat).  Copying them inside the submit call costs the thread
// that also drives the track chain 30-70 us per frame (two hipMemcpyAsync, for pageable memory incl. the staging
// copy).  In ssf_process_sequence the frames are known ahead, so a worker thread copies them into a ring of device
// buffers on a stream of its own; the submitting thread only makes the extract stream wait for the copy's event.
// A sequence starts with small batches (3/8, then 5/8 of extract_batch, then full ones): the first frame can
// only be tracked when the whole first batch has been extracted, and a full batch of 8 takes twice as long as one of 2.
// (Round 3, at the faster relabelling pass, two sweeps of three runs each over the driver's 20 frames: 3,5 7253-7378 frames/s |
// 2,4 7168-7315 | 2,5 7243-7350 | 2,6 6920-7108 | 1,3 7162-7245 | 1,2,4 6811-6858 | 4 6665-6985: tools/ramp_probe.sh.)
// (SSF_SEQ_RAMP="a,b,..": sizes of the leading batches for experiments, each clamped to [1, batch].  Measured over the
// driver's 20 timed frames, batch 8, three runs each: 2,4 (the default) 5900-5990 frames/s | 2,2 5880-5980 | 2 5760-5820 |
// 3 5680-5810 | 2,8 5700-5810 | 1 5600-5780 | 4 5530-5560)
struct SeqRamp { int n = -1; int size[8]; };
static const SeqRamp& seq_ramp() {
    static const SeqRamp ramp = [] {              // (initialised once, also when several handles are driven by several threads)
        SeqRamp r;
        r.n = 0;
        const char* e = SSF_ENV_STR("SEQ_RAMP");
        if (e) { for (const char* q = e; *q && r.n < 8;) { r.size[r.n++] = atoi(q); while (*q && *q != ',') q++; if (*q == ',') q++; } }
        else { r.n = 2; r.size[0] = -3; r.size[1] = -5; }          // (negative: |value| eighths of the batch, rounded)
        return r;
    }();
    return ramp;
}
static inline int seq_batch_size(int b, int batch) {
    const SeqRamp& r = seq_ramp();
    if (b >= r.n) return batch;
    const int v = r.size[b] < 0 ? (-r.size[b] * batch + 4) / 8 : r.size[b];
    return std::min(batch, std::max(1, v));
}
static inline int seq_batch_of(int i, int batch) {
    int b = 0;
    for (;; b++) {
        const int sz = seq_batch_size(b, batch);
        if (b >= seq_ramp().n) return b + i / batch;
        if (i < sz) return b;
        i -= sz;
    }
}
// (a longer ramp -- 2, 4, 4, 6 before the batches of 8 -- was measured in round 2: 6100-6150 against 5960-6260 frames/s over
// 20 timed frames, i.e. nothing)
#ifndef SSF_UPLOAD_RING_BYTES
#define SSF_UPLOAD_RING_BYTES (256ull << 20)          // cap of the upload ring's device buffers (and, again, of its page-locked staging)
#endif
struct Uploader {
    // several workers, frames dealt to them in turn: a pageable hipMemcpyAsync is a host memcpy into a staging buffer, and one
    // thread sustains 6-10 GB/s of it (box to box) = 3000-5000 frames/s at 2.1 MB per frame, less than the pipeline consumes.
    // Two workers were the bottleneck on the slower hosts of the pool (6200 frames/s with host frames against 11 460 with frames in
    // HBM, page-locked caller memory no better: tools/host_buffer_probe.py, round 4) and are not on the faster ones (9900 with 2,
    // 3, 4 workers, 10 600 with 6: tools/upload_ab.sh); four.
#ifndef SSF_UPLOAD_THREADS
#define SSF_UPLOAD_THREADS 4
#endif
    static const int NTH = SSF_UPLOAD_THREADS;
    std::thread th[NTH];
    std::atomic<int> done[NTH];                // worker t: frames < done[t] of its residue class are enqueued
    std::atomic<int> processed{0};            // frames the caller has finished with (their ring slots may be reused)
    std::atomic<int> failed{0}, stop{0};
    int n = 0, ring = 0, device = 0;
    const void* const* rgb = nullptr; const void* const* depth = nullptr;
    size_t rgb_bytes = 0, depth_bytes = 0;
    // frame i is copied on the stream of the extract context that will take it (contexts take batches in turn): the
    // copy precedes that batch's launch in stream order, and no further hardware queue becomes active (a 5th one
    // halves the throughput of the others)
    std::vector<hipStream_t> ctx_stream; int ctx0 = 0, batch = 1;
    std::vector<uint8_t*> d_rgb; std::vector<float*> d_depth;
    // page-locked staging slots (one per ring slot): the worker copies the caller's pageable frame here itself and hands
    // the runtime a truly asynchronous DMA; left to the runtime, pageable copies of several threads serialise inside it
    std::vector<uint8_t*> p_rgb; std::vector<float*> p_depth;
    bool ready(int i) const { return done[i % NTH].load(std::memory_order_acquire) > i; }
    // where the workers' time went, microseconds summed over the workers since the handle was created (ssf_upload_stats):
    // waiting for a free ring slot | the staging memcpy | the two hipMemcpyAsync calls; frames
    std::atomic<long long> us_ring{0}, us_memcpy{0}, us_enqueue{0}, frames_done{0};
    void run(int t) {
        if (hipSetDevice(device) != hipSuccess) { failed.store(1); return; }
        for (int i = t; i < n && !stop.load(std::memory_order_relaxed); i += NTH) {
            const double t0 = now_us();
            while (i >= processed.load(std::memory_order_acquire) + ring) {
                if (stop.load(std::memory_order_relaxed)) return;
                std::this_thread::sleep_for(std::chrono::microseconds(20));       // (the ring is two batches ahead of the submitting thread)
            }
            const double t1 = now_us();
            const int sl = i % ring;
            hipStream_t st = ctx_stream[(size_t)(ctx0 + seq_batch_of(i, batch)) % ctx_stream.size()];
            const void* src_rgb = rgb[i]; const void* src_depth = depth[i];
            if (!p_rgb.empty()) {           // (slot sl was last used by frame i - ring, which has been processed: its DMA is done)
                std::memcpy(p_rgb[sl], rgb[i], rgb_bytes); std::memcpy(p_depth[sl], depth[i], depth_bytes);
                src_rgb = p_rgb[sl]; src_depth = p_depth[sl];
            }
            const double t2 = now_us();
            if (hipMemcpyAsync(d_rgb[sl], src_rgb, rgb_bytes, hipMemcpyHostToDevice, st) != hipSuccess ||
                hipMemcpyAsync(d_depth[sl], src_depth, depth_bytes, hipMemcpyHostToDevice, st) != hipSuccess) { failed.store(1); return; }
            done[t].store(i + 1, std::memory_order_release);
            const double t3 = now_us();
            us_ring += (long long)(t1 - t0); us_memcpy += (long long)(t2 - t1); us_enqueue += (long long)(t3 - t2); frames_done++;
        }
    }
    // The workers live as long as the handle: a thread's first HIP call initialises the runtime's per-thread state (several
    // milliseconds), which a sequence of 240 host frames used to pay anew on every call -- 200 us per frame instead of 125.
    // start() posts the sequence described by the fields above as job `gen`; join() waits until every worker has finished it.
    std::mutex mu; std::condition_variable cv;
    unsigned long long gen = 0; int finished = NTH; bool quit = false, spawned = false;
    void worker(int t) {
        unsigned long long seen = 0;
        for (;;) {
            { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return quit || gen != seen; }); if (quit) return; seen = gen; }
            run(t);
            { std::lock_guard<std::mutex> lk(mu); finished++; }
            cv.notify_all();
        }
    }
    void start() {
        { std::lock_guard<std::mutex> lk(mu); for (int t = 0; t < NTH; t++) done[t].store(0); finished = 0; gen++; }
        if (!spawned) { spawned = true; for (int t = 0; t < NTH; t++) th[t] = std::thread([this, t] { worker(t); }); }
        cv.notify_all();
    }
    void join() { std::unique_lock<std::mutex> lk(mu); cv.wait(lk, [&] { return finished >= NTH; }); }
    void shutdown() {
        { std::lock_guard<std::mutex> lk(mu); quit = true; }
        cv.notify_all();
        for (int t = 0; t < NTH; t++) if (th[t].joinable()) th[t].join();
    }
};

// ---- streams outlive handles ---------------------------------------------------------------------------
// The runtime maps streams onto hardware queues when they are created, and how it does that depends on the process' history: the
// SECOND handle of a process (first one destroyed, its streams with it) ran the very same sequence at 6400 instead of 11 300 frames/s --
// device-resident frames, nothing else changed (tools/host_buffer_probe.py with PROBE_KINDS=device,device; round 4: what had looked
// like "slow hosts" in the node-call figures of bench.py was this: those figures are taken on later handles of the process).  A handle
// therefore returns its streams to a pool when it is destroyed, and the next handle with the same device and priorities takes them:
// every handle of a process runs on the queues the first one got.  (Handles that live side by side get streams of their own.)
struct StreamPool {
    std::mutex mu;
    std::map<std::pair<int, int>, std::vector<hipStream_t>> idle;         // (device, priority) -> streams
    static const int CAPTURE = 1 << 20;                                    // "priority" of the capture-only streams
    // (dev: the handle's device, cfg.device_id -- not the calling thread's current one: a handle may be destroyed from a thread whose
    // current device is another GPU of the node)
    // A stream is created ON the handle's device (the calling thread's current device is put back afterwards: the capture stream is
    // taken lazily, from whatever thread first runs a segmentation).  The idle list of a key is bounded: a process that has run many
    // handles SIDE BY SIDE and destroyed them keeps at most MAX_IDLE streams per (device, priority) -- what a later handle can take --
    // and destroys the surplus instead of holding every hardware queue for good.
    static const size_t MAX_IDLE = 8;
    hipStream_t take(int dev, int prio) {
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = idle[std::make_pair(dev, prio)];
            if (!v.empty()) { hipStream_t st = v.back(); v.pop_back(); return st; }
        }
        int cur = -1;
        const bool switched = hipGetDevice(&cur) == hipSuccess && cur != dev && hipSetDevice(dev) == hipSuccess;
        hipStream_t st = nullptr;
        const hipError_t e = prio == CAPTURE ? hipStreamCreateWithFlags(&st, hipStreamNonBlocking) : hipStreamCreateWithPriority(&st, hipStreamNonBlocking, prio);
        if (switched) (void)hipSetDevice(cur);
        return e == hipSuccess ? st : nullptr;
    }
    void give(hipStream_t st, int dev, int prio) {
        if (!st) return;
        {
            std::lock_guard<std::mutex> lk(mu);
            auto& v = idle[std::make_pair(dev, prio)];
            if (v.size() < MAX_IDLE) { v.push_back(st); return; }
        }
        (void)hipStreamDestroy(st);                 // (the caller has synchronised it)
    }
};
static StreamPool& stream_pool() { static StreamPool* p = new StreamPool(); return *p; }      // (never destroyed: the runtime may be gone by then)

// ---- the first ICP iteration of the next frame: inside the row-move kernel, or as a launch of its own? ------------------------
// Both forms give the same record bit for bit (exact integer sums).  Which one is FASTER depends on what else the part is doing,
// and flipped sign between measurements of round 5 (profiles/track_chain_r05.txt): alone on the part the fused form saves a launch
// and a trip (first record 11 us after the frame's entry against 15); next to the extract launches of a filling pipeline its
// 3900-workgroup launch finishes late (36 us against 27) -- the driver's 20-frame form ran 4-7 % faster WITHOUT the fusion, a
// 1200-frame steady state 3-5 % faster WITH it.  So the handle measures: the period between consecutive frame completions of a
// pipelined sequence is attributed to the form that was in effect, the two forms take turns of PROBE frames, and the better mean
// holds for HOLD frames before the next probe.  A handle starts WITHOUT the fusion (short sequences are fill-bound and never leave
// that phase).  Results do not depend on any of it.
struct AheadTuner {
    static const int START = 48, PROBE = 16, ROUNDS = 3, HOLD = 1024, SKIP = 2;
    int forced = -1;                 // lab: SSF_ICP_AHEAD = 0 / 1 / 2 pins the form (2: fused and the track stream waits for the next batch)
    int mode = 0, frames = 0, left = START, round = 0, since_switch = 0;
    bool probing = false;
    double sum[2] = {0, 0}; int n[2] = {0, 0};
    double last_done_us = -1.0; int last_mode = 0;
    int current() const { return forced >= 0 ? (forced ? 1 : 0) : mode; }
    void sequence_break() { last_done_us = -1.0; }                       // (the period across a drained pipeline says nothing)
    void frame_done(double t_us, int iters) {
        if (forced >= 0) return;
        if (last_done_us >= 0.0 && since_switch >= SKIP && probing && iters > 0) { sum[last_mode] += (t_us - last_done_us) / (double)(iters + 4); n[last_mode]++; }   // (per unit of chain work: iterations + the fixed part)
        last_done_us = t_us; last_mode = mode; frames++; since_switch++;
        if (--left > 0) return;
        if (!probing) { probing = true; round = 0; sum[0] = sum[1] = 0; n[0] = n[1] = 0; mode ^= 1; left = PROBE; since_switch = 0; return; }
        if (++round < 2 * ROUNDS) { mode ^= 1; left = PROBE; since_switch = 0; return; }
        probing = false;
        if (n[0] > 0 && n[1] > 0) mode = (sum[1] / n[1] < sum[0] / n[0]) ? 1 : 0;
        left = HOLD; since_switch = 0;
    }
};

// ---- handle -----------------------------------------------------------------------------------------
struct IcpLoop {
    bool active = false, valid = true, done = true;
    int iter = 0;
    unsigned long long ahead_seq = 0;         // != 0: the first iteration's record was accumulated ahead (ssf_handle::ahead)
    double tf_inc[16], prev_error, JtJ[36];
    M3 R_init; V3 t_init, t_inc_stale;
};

// Everything the extract stage of one BATCH of frames owns (cfg.extract_batch frames, slot b of every
// buffer at + b * slab bytes).  With pipeline_depth > 0 there are pipeline_depth + 1 of these, each on
// its own stream: the extract of later batches runs while the track/fuse chain (h->stream) consumes the
// frames of an earlier one.  Extract has no cross-frame state (the RANSAC draws are keyed by the frame
// number), so batches are independent of one another.
struct ExtractCtx {
    FrameMaps maps;                               // slot 0; maps.slab = bytes to the next slot
    SurfelSoA frame;
    unsigned long long* d_best = nullptr; uint8_t* d_matched = nullptr;
    uint8_t* d_rgb_in = nullptr; float* d_depth_in = nullptr; float* d_depth_filt = nullptr; uint8_t* d_mask = nullptr;
    float* d_wire = nullptr;                      // 26 S words: the frame supersurfels of a frame extracted elsewhere (ssf_submit_frame_tables)
    char* d_pas = nullptr; char* d_team_ws = nullptr;      // relabelling passes in one launch per phase (k_passes_team): per-pass arguments, team workspace
    ncclComm_t deal_comm = nullptr;               // dealt extract stage: this context's own communicator (a collective per batch on ITS stream)
    bool mine = true; long long deal_batch = 0;   // ... whether the open batch is this rank's to extract, and its number in the frame stream
    hipStream_t stream = nullptr; bool own_stream = false; int stream_prio = 0;
    hipEvent_t ev_done = nullptr, ev_consumed = nullptr, ev_t0 = nullptr, ev_t1 = nullptr;
    bool consumed_valid = false, timed = false;
    hipGraph_t graph[SSF_MAX_BATCH + 1] = {}; hipGraphExec_t exec[SSF_MAX_BATCH + 1] = {};
    // batch state: open (count > 0, !launched) -> in flight (launched, inflight > 0) -> free
    int count = 0, inflight = 0, stamp0 = 0, nb_launched = 1; bool launched = false, waited = false;
    uint32_t epoch0 = 0;
    BatchIn in = {}; unsigned mask_bits = 0;
};
// the frame the track/fuse chain works on: slot views into its context
struct ActiveFrame {
    FrameMaps maps; SurfelSoA frame;
    unsigned long long* d_best = nullptr; uint8_t* d_matched = nullptr;
    ExtractCtx* ctx = nullptr; int slot = 0;
};

// Round 6: the tile-sorted copy (ssf_tile_rows.inc) in the product, for LARGE visible sets.  At BASELINE config 3 (940 k visible
// rows, ten iterations) it takes k_icp from 23.0 to 18.8 us per iteration and the association from 55.6 to 29.9 us for a 35 us sort
// (profiles/config3_sorted_rows_r06.txt); at the metric's 120 k visible rows and four iterations the sort costs more than it saves,
// hence the threshold.  -DSSF_BIN_MIN_ROWS_DEFAULT=-1 builds a product without it (the A/B).
#ifndef SSF_BIN_MIN_ROWS_DEFAULT
#define SSF_BIN_MIN_ROWS_DEFAULT 400000
#endif
struct ssf_handle {
    ssf_config cfg;
    int S = 0, gx = 0, gy = 0;
    std::string err;
    hipStream_t stream = nullptr; bool own_stream = false; int stream_prio = 0;
    SegParams seg; Cam cam;
    std::vector<ExtractCtx> ctx; int open_ctx = 0, batch = 1;
    std::deque<std::pair<int, int>> pending;      // (context, slot) submitted, not yet processed (oldest first)
    ActiveFrame active; ActiveFrame* cc = &active; // the frame the track/fuse chain is working on (or last worked on)
    uint32_t extract_ordinal = 0;                 // frames submitted so far = RNG epoch of the next frame
    // ssf_process_sequence: frames still to be submitted; do_fuse submits them between its launches and its wait for
    // the counters (the ~40 us of host work of a batch launch hide behind the ~55 us fuse chain on the GPU)
    const void* const* seq_rgb = nullptr; const void* const* seq_depth = nullptr; int seq_next = 0, seq_n = 0, seq_on_device = 0, stamp_bias = 0;
    long long n_waiter_matches = 0;           // frames whose association ran in a waiting ICP launch (debug)
    int seq_k = 0;                            // frame of the sequence the track loop is working on (debug marks)
    int seq_batches = 0;                      // batches launched by the running ssf_process_sequence (see seq_batch_size)
    double us_wait_upload = 0.0;                          // the submitting thread's wait for uploads (ssf_upload_stats)
    Uploader* up = nullptr; bool seq_upload = false;   // host frames of a sequence are copied ahead by a worker thread
    // multi-GPU: RCCL communicator over the ranks of cfg.nranks (ssf_comm_attach); the shard sizes of all ranks
    // are all-gathered at the end of every frame and read lazily at the start of the next one
    ncclComm_t comm = nullptr; int* d_all5 = nullptr;
    int deal = 0; long long deal_batches = 0;     // ssf_comm_deal_extract: 0 replicated, 1 dealt, 2 dealt + the extracting rank re-imports its own tables (self-check)
    // ... or the peer-to-peer exchange region of ssf_p2p_* (one node; no collective launches): own region, the peers'
    // regions as mapped into this process, and one sequence number per exchange kind (identical on every rank)
    struct P2P {
        unsigned char* region = nullptr; size_t bytes = 0; bool fine = false;
        bool same_device = false; double timeout_s = 30.0;       // ssf_p2p_configure
        P2PView view{}; bool on = false; std::vector<void*> opened;
        unsigned long long seq_icp = 0, seq_cnt = 0, seq_assoc = 0, seq_migr = 0;
    } p2p;
    unsigned long long all_seq = 0; bool all_pending = false, all_valid = false;
    long long all_cnt[5 * SSF_MAX_RANKS];
    SurfelSoA model[2]; int mcur = 0;
    std::vector<void*> allocs;
    struct Guarded { void* base; size_t bytes, guard; };
    std::vector<Guarded> guarded;         // SSF_ALLOC_GUARD (debug): see dalloc
    float* d_bf_in = nullptr; float* d_bf_out = nullptr; float* d_orient9 = nullptr; float* d_frame_orient9 = nullptr;
    long long* d_icp = nullptr;
    uint8_t* d_state = nullptr; int32_t* d_cand = nullptr; Counters* d_cnt = nullptr;
    // multi-GPU migration: this shard's migrant table (SSF_MIGRANT_WORDS x S int32), state between the two fuse halves
    int32_t* d_migrants = nullptr; PartitionWs fuse_ws{}; bool fuse_first = false, fuse_migrate = false, fusing = false;
    MoveTotals fuse_totals{0, 0, 0, 0}; bool move_totals_on = true;      // (lab: SSF_MOVE_TOTALS=0 keeps the fuse launch's tail)
    // model store: model[mcur] = dense array of the visible rows (ping-pong), oov[ocur] = out-of-view rows (deque
    // with live flags, host mirror of the span below), dense = materialised [visible | out-of-view] view for the
    // consumers of the whole model (get/set model, export, deformation)
    OovStore oov[2]; int ocur = 0; int oov_head = 0, oov_tail = 0, oov_live = 0; long long n_recentres = 0;
    uint8_t* d_state_oov = nullptr; uint32_t* d_bc_oov = nullptr;
    // sums of the per-frame partition (PartitionWs): two sets of part_words, used alternately; 128 arrival counters
    uint32_t* d_part = nullptr; uint32_t* d_part_ticket = nullptr; int part_words = 0, part_sup_vis = 0, part_sup_oov = 0, part_set = 0;
    SurfelSoA dense; uint8_t* d_live_scratch = nullptr;
    int32_t* d_scratch_map = nullptr;
    long long* d_icp_replicas = nullptr; unsigned int* d_tickets = nullptr; float* d_srgb_lut = nullptr;
    // host-mapped mailbox (fine-grained): results the host waits for are polled, not synchronised on
    Mailbox* mb_host = nullptr; Mailbox* mb_dev = nullptr;
    unsigned long long icp_seq = 0, cnt_seq = 0;
    // first ICP iteration of the next submitted frame, accumulated ahead by the row-move kernel of the frame just
    // fused (do_fuse): valid for exactly that frame, that pose and that model; anything else drops it
    struct { bool valid = false; unsigned long long seq = 0; ExtractCtx* ctx = nullptr; int slot = 0; int stamp = 0; Rt pose; } ahead;
    double wait_launched_us = 0.0; long long n_waiter_match_repairs = 0; long long dbg_stall_before_match_us = 0;     // see process_oldest: SSF_ICP_GO_MATCH has no acknowledgement
    bool icp_ahead = true; int icp_ahead_mode = 1;         // 1: when the next frame's extract has finished (the product); 2 (lab): always, the track stream waits for it
    AheadTuner ahead_tuner;
    // chained ICP launches: iteration i + 1 is launched while iteration i runs and waits on the device for the host's
    // word (launch_icp, IcpGo): slots in fine-grained device memory the host stores into directly
    IcpGo* go = nullptr; bool icp_chain = true; unsigned long long go_count = 0;
    bool graph_failed = false; hipStream_t capture_stream = nullptr;
    // tile-sorted copy of the visible rows' ICP / association fields (launch_bin_rows), made at the start of a frame's
    // tracking when the visible set is large (bin_min_rows); valid for that frame only
    SurfelSoA bins{}; int32_t* d_bin_idx = nullptr; uint32_t* d_bin_count = nullptr; uint32_t* d_bin_cursor = nullptr;
    bool bins_valid = false; int bin_min_rows = SSF_BIN_MIN_ROWS_DEFAULT;      // visible rows from which a frame's tracking streams the tile-sorted copy (< 0: never)
    // pass_team: the relabelling passes of a phase as ONE launch with a frame per XCD (k_passes_team) instead of a launch per pass.
    // Its workgroups must all be on the chip at once, so whole batches take turns across the contexts (launch_batch: a batch's
    // chain waits for the previous batch's ev_done).
    bool pass_team = false; ExtractCtx* team_prev = nullptr;
    long long h_icp_local[SSF_ICP_RECORD];
    long long* h_icp = nullptr; Counters* h_cnt = nullptr;
    int n_model = 0, n_visible = 0, stamp = 0, max_passes = 0;
    Rt pose;
    IcpLoop icp;
    long long id_offset = 0, global_n_model = -1, global_n_visible = -1;
    bool have_frame = false;
    int last_icp_valid = 0, last_icp_iters = 0;
    hipEvent_t ev[4] = {nullptr, nullptr, nullptr, nullptr};
    KernelTimer timer;
    std::vector<std::string> timer_names;
    double seq_mark_us[4][64] = {{0}}, seq_launch_us[32] = {0}, seq_launch_host_us[32] = {0}; int seq_launch_n[32] = {0}, seq_launches = 0;   // debug: entry / first ICP record / ICP done / counters back per frame, batch launches
    double seq_t0_us = 0, seq_done_us[64] = {0};      // debug: completion time of the first frames of the last ssf_process_sequence
    double host_us[8] = {0, 0, 0, 0, 0, 0, 0, 0};   // debug: submit | icp loop | match+fuse | frames | extract ready at activation | first icp iteration
};
static std::string g_create_err;

#define HCK(call)                                                                                    \
    do {                                                                                             \
        hipError_t e_ = (call);                                                                      \
        if (e_ != hipSuccess) {                                                                      \
            h->err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
            return SSF_ERR_DEVICE;                                                                   \
        }                                                                                            \
    } while (0)

#define NCK(call)                                                                                    \
    do {                                                                                             \
        ncclResult_t r_ = (call);                                                                    \
        if (r_ != ncclSuccess) {                                                                     \
            RcclApi* a_ = rccl_api();                                                                \
            h->err = std::string(#call) + ": " + ((a_ && a_->GetErrorString) ? a_->GetErrorString(r_) : "RCCL error"); \
            return SSF_ERR_DEVICE;                                                                   \
        }                                                                                            \
    } while (0)

// Wait until the device has published sequence number `want` into the host-mapped mailbox word.
// Bounded: falls back to a stream synchronise (and reports a device error) after ~5 s.
static int wait_seq(ssf_handle* h, const volatile unsigned long long* word, unsigned long long want) {
    const auto t0 = std::chrono::steady_clock::now();
    unsigned long long spins = 0;
    while (__atomic_load_n(word, __ATOMIC_ACQUIRE) != want) {
        if ((++spins & 0xFFFF) == 0) {
            if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 5.0) {
                hipError_t e = hipStreamSynchronize(h->stream);
                if (e != hipSuccess) { h->err = std::string("device error while waiting: ") + hipGetErrorString(e); return SSF_ERR_DEVICE; }
                if (__atomic_load_n(word, __ATOMIC_ACQUIRE) == want) return SSF_OK;
                h->err = "mailbox sequence number never arrived"; return SSF_ERR_DEVICE;
            }
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    return SSF_OK;
}

template <typename T>
static bool dalloc(ssf_handle* h, T** p, size_t count) {
    void* q = nullptr;
    // (SSF_ALLOC_GUARD=bytes: that much unused memory on both sides of every buffer -- a probe for out-of-bounds accesses
    // between the small buffers of handles that live side by side, tools/p2p_first_frame_stress.py)
    static const size_t guard = (size_t)SSF_ENV_INT("ALLOC_GUARD", 0) & ~(size_t)255;          // (lab: poisoned guard zones around every buffer)
    const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
    if (hipMalloc(&q, bytes + 2 * guard) != hipSuccess) return false;
    h->allocs.push_back(q);
    if (guard) {                                   // poisoned guard zones, checked by ssf_destroy
        static const int poison_all = SSF_ENV_INT("GUARD_BYTE", 0xA5) & 255;
        static const int only = SSF_ENV_INT("GUARD_ONLY", -1);      // poison this allocation's zones, zero the others'
        const int poison = (only < 0 || (int)h->guarded.size() == only) ? poison_all : 0;
        (void)hipMemset(q, poison, guard);
        (void)hipMemset((char*)q + guard + bytes, poison, guard);
        h->guarded.push_back({q, bytes, guard});
    }
    *p = (T*)((char*)q + guard);
    return true;
}
static void check_guards(ssf_handle* h) {
    int idx = 0;
    for (auto& g : h->guarded) {
        std::vector<unsigned char> host(g.guard);
        for (int side = 0; side < 2; side++) {
            const char* zone = (const char*)g.base + (side ? g.guard + g.bytes : 0);
            if (hipMemcpy(host.data(), zone, g.guard, hipMemcpyDeviceToHost) != hipSuccess) continue;
            size_t first = g.guard, last = 0, n = 0;
            static const int poison = SSF_ENV_INT("GUARD_BYTE", 0xA5) & 255;
            for (size_t i = 0; i < g.guard; i++) if (host[i] != (unsigned char)poison) { if (first == g.guard) first = i; last = i; n++; }
            if (n) std::fprintf(stderr, "[ssf guard] allocation #%d (%zu bytes): %zu bytes modified %s it, offsets %zu..%zu relative to the %s (rank %d)\n",
                                idx, g.bytes, n, side ? "BEHIND" : "IN FRONT OF", side ? first : g.guard - 1 - last, side ? last : g.guard - 1 - first,
                                side ? "end" : "start", h->cfg.rank);
        }
        idx++;
    }
}
static bool alloc_surfels(ssf_handle* h, SurfelSoA& s, size_t n) {
    return dalloc(h, &s.pos, 3 * n) && dalloc(h, &s.col, 3 * n) && dalloc(h, &s.lab, 3 * n) && dalloc(h, &s.stamps, 2 * n) &&
           dalloc(h, &s.r0, 3 * n) && dalloc(h, &s.r1, 3 * n) && dalloc(h, &s.r2, 3 * n) && dalloc(h, &s.shape, 6 * n) &&
           dalloc(h, &s.dims, 2 * n) && dalloc(h, &s.conf, n);
}
static void zero_surfels(ssf_handle* h, SurfelSoA& s, size_t n) {
    (void)hipMemsetAsync(s.pos, 0, 12 * n, h->stream); (void)hipMemsetAsync(s.col, 0, 12 * n, h->stream);
    (void)hipMemsetAsync(s.lab, 0, 12 * n, h->stream); (void)hipMemsetAsync(s.stamps, 0, 8 * n, h->stream);
    (void)hipMemsetAsync(s.r0, 0, 12 * n, h->stream); (void)hipMemsetAsync(s.r1, 0, 12 * n, h->stream);
    (void)hipMemsetAsync(s.r2, 0, 12 * n, h->stream); (void)hipMemsetAsync(s.shape, 0, 24 * n, h->stream);
    (void)hipMemsetAsync(s.dims, 0, 8 * n, h->stream); (void)hipMemsetAsync(s.conf, 0, 4 * n, h->stream);
}
static Rt pose_from12(const float* p) {
    Rt r; r.R = m3(v3(p[0], p[1], p[2]), v3(p[3], p[4], p[5]), v3(p[6], p[7], p[8])); r.t = v3(p[9], p[10], p[11]); return r;
}
static void pose_to12(const Rt& r, float* p) {
    p[0] = r.R.r0.x; p[1] = r.R.r0.y; p[2] = r.R.r0.z; p[3] = r.R.r1.x; p[4] = r.R.r1.y; p[5] = r.R.r1.z;
    p[6] = r.R.r2.x; p[7] = r.R.r2.y; p[8] = r.R.r2.z; p[9] = r.t.x; p[10] = r.t.y; p[11] = r.t.z;
}
struct TimerScope {
    ssf_handle* h;
    explicit TimerScope(ssf_handle* hh) : h(hh) {
        if (hh->cfg.profile == 1) timer_calibrate(&hh->timer, hh->stream);       // first use only
        set_current_timer(hh->cfg.profile == 1 ? &hh->timer : nullptr);
    }
    ~TimerScope() { set_current_timer(nullptr); }
};

// ---- stages -----------------------------------------------------------------------------------------
// the segmentation chain between ingest and finalize: fixed topology and arguments per context.
// Pass k reads label/sums buffer k&1 and writes the other; no merge launch between passes (the pass
// kernel rebuilds the rows it needs from the quiescent sums buffer).  The global superpixel table is
// only materialised where a later stage wants it: before the plane filter.
static void enqueue_segmentation(ssf_handle* h, ExtractCtx& c, hipStream_t st) {
    const SegParams& p = h->seg;
    const int nb = c.count;
    const int limit = h->max_passes > 0 ? h->max_passes : (1 << 30);
    const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 1, 1, 0};                 // pass order, TPS_RGBD.cu:190-268
    const int k1 = std::min(4 * (h->cfg.seg_iter / 2), limit), k2 = std::min(4 * h->cfg.seg_iter, std::max(limit, k1));
    unsigned int* abort_flag = &h->mb_dev->extract_abort;
    const bool team = h->pass_team && c.d_pas && c.d_team_ws && h->max_passes == 0;        // (lab: lab/passes_team.inc)
    if (team) {
        (void)hipMemsetAsync(c.d_team_ws, 0, pass_team_ws_bytes(), st);
        launch_update_passes_team(st, p, c.maps, nb, 0, k1, false, c.d_pas, c.d_team_ws, abort_flag);
    } else
        for (int k = 0; k < k1; k++) launch_update_pass(st, p, c.maps, nb, k, ox[k & 3], oy[k & 3], false, 0);
    // sums[k1&1] holds the exact sums after k1 passes; RANSAC and the inlier initialisation read them directly
    if (h->cfg.seg_use_ransac) {
        launch_init_samples(st, p, c.maps, nb, k1 & 1);
        launch_eval_samples(st, p, c.maps, nb);
        launch_init_disp(st, p, c.maps, nb, true);
    } else launch_init_disp(st, p, c.maps, nb, false);
    int k = k1;
    if (team) {
        (void)hipMemsetAsync(c.d_team_ws, 0, pass_team_ws_bytes(), st);
        launch_update_passes_team(st, p, c.maps, nb, k1, k2, true, c.d_pas, c.d_team_ws, abort_flag);
        k = std::max(k1, k2);
    } else
        for (; k < k2; k++) launch_update_pass(st, p, c.maps, nb, k, ox[k & 3], oy[k & 3], true, 0);
    launch_plane_filter(st, p, c.maps, nb, k & 1);             // final merge (table + planes) + smoothing sweeps
    launch_render_moments(st, p, h->cam, c.maps, nb);
}
// ~45 short dependent kernels: replayed as one captured hipGraph (launch-bound inner loop), one graph per
// batch size; eager when kernels are individually timed or the pass count is being bisected
static int run_segmentation(ssf_handle* h, ExtractCtx& c) {
    const bool use_graph = h->cfg.profile != 1 && h->max_passes == 0 && !h->graph_failed;
    if (use_graph) {
        hipGraphExec_t& ex = c.exec[c.count];
        if (!ex) {
            // captured on a stream of its own, not on the context's: the upload thread (Uploader) may be enqueueing
            // copies on the context's stream at this very moment
            // (one capture at a time in the process: with several handles driven by several threads -- shards of one map, or
            // several cameras on one GPU -- captures that ran side by side left, once in ~100 first frames, a graph whose
            // first replay differed from the eager chain: tools/p2p_probe.py, round 2)
            static std::mutex capture_mutex;
            static const bool unlocked = SSF_ENV_SET("CAPTURE_UNLOCKED");          // (control runs of that probe)
            std::unique_lock<std::mutex> capture_lock(capture_mutex, std::defer_lock);
            if (!unlocked) capture_lock.lock();
            bool ok = (h->capture_stream || (h->capture_stream = stream_pool().take(h->cfg.device_id, StreamPool::CAPTURE)) != nullptr) &&
                      hipStreamBeginCapture(h->capture_stream, hipStreamCaptureModeThreadLocal) == hipSuccess;
            if (ok) {
                enqueue_segmentation(h, c, h->capture_stream);
                ok = hipStreamEndCapture(h->capture_stream, &c.graph[c.count]) == hipSuccess && c.graph[c.count] != nullptr;
            }
            if (ok) ok = hipGraphInstantiate(&ex, c.graph[c.count], nullptr, nullptr, 0) == hipSuccess;
            if (!ok) { h->graph_failed = true; ex = nullptr; (void)hipGetLastError(); }
            if (ex) { HCK(hipGraphLaunch(ex, c.stream)); return SSF_OK; }       // (the first replay still under the lock)
        }
        if (ex) { HCK(hipGraphLaunch(ex, c.stream)); return SSF_OK; }
    }
    enqueue_segmentation(h, c, c.stream);
    return SSF_OK;
}

// Launch the extract stage of the open batch of context c (asynchronous; nothing is waited for).
static int launch_batch(ssf_handle* h, ExtractCtx& c) {
    const double t_launch0 = now_us();
    hipStream_t st = c.stream;
    const bool multi = h->ctx.size() > 1;
    const int nb = c.count;
    // the track/fuse chain must be done with the frames this context held before they are overwritten
    if (multi && c.consumed_valid) HCK(hipStreamWaitEvent(st, c.ev_consumed, 0));
    if (multi && h->pass_team && h->team_prev && h->team_prev != &c) HCK(hipStreamWaitEvent(st, h->team_prev->ev_done, 0));      // (see pass_team)
    c.timed = h->cfg.profile != 0;
    if (c.timed) HCK(hipEventRecord(c.ev_t0, st));
    // The extract stage DEALT over the ranks of a sharded map (ssf_comm_deal_extract): batch j of the frame
    // stream is extracted by rank j % nranks alone, which broadcasts every frame's label map, plane depth and supersurfels (2.5 MB
    // at 640 x 480) on THIS context's communicator and stream; the other ranks receive them into the same slots and rebuild their
    // private tables (k_import_frame).  Every rank launches the same batches in the same order (same frames, same configuration),
    // so the collectives of a context's communicator are issued in the same order everywhere.
    const bool dealt = h->deal != 0 && h->comm != nullptr;
    const bool mine = !dealt || c.mine;
    int extract_rc = SSF_OK;
    if (mine) {
        if (h->cfg.depth_prefilter) {                                      // supersurfel_fusion.cu:180 -- the batch's frames in one launch
            launch_bilateral_batch(st, c.in, c.d_depth_filt, c.maps.slab, nb, h->cfg.width, h->cfg.height, h->cfg.prefilter_sigma_color, h->cfg.prefilter_sigma_space);
            for (int b = 0; b < nb; b++) c.in.depth[b] = slab_shift(c.d_depth_filt, (size_t)b * c.maps.slab);
        }
        launch_ingest(st, h->seg, c.in, c.maps, nb, c.epoch0);
        extract_rc = run_segmentation(h, c);
        if (extract_rc && !dealt) return extract_rc;
        if (!extract_rc) launch_finalize_surfels(st, h->seg, c.maps, nb, c.frame, h->cfg.range_min, h->cfg.range_max, c.stamp0, c.d_mask, c.mask_bits,
                                                 c.d_best, c.d_matched);
    }
    if (dealt) {
        // (a local failure above must not leave the other ranks waiting in their broadcasts: the group is issued regardless -- what it
        //  ships is then meaningless, and this rank reports the error -- and GroupStart is always paired with GroupEnd)
        RcclApi* api = rccl_api();
        const int root = (int)(c.deal_batch % (long long)h->cfg.nranks);
        const size_t P = (size_t)h->cfg.width * h->cfg.height;
        if (mine && !extract_rc) launch_export_rows(st, h->seg, c.maps, nb, c.frame, c.d_wire);
        NCK(api->GroupStart());
        ncclResult_t bc_rc = ncclSuccess;
        for (int b = 0; b < nb && bc_rc == ncclSuccess; b++) {
            const size_t off = (size_t)b * c.maps.slab;
            int32_t* lab = slab_shift(c.maps.label, off); float* pd = slab_shift(c.maps.plane_depth, off); float* w = slab_shift(c.d_wire, off);
            bc_rc = api->Broadcast(lab, lab, P, ncclInt32, root, c.deal_comm, st);
            if (bc_rc == ncclSuccess) bc_rc = api->Broadcast(pd, pd, P, ncclFloat32, root, c.deal_comm, st);
            if (bc_rc == ncclSuccess) bc_rc = api->Broadcast(w, w, 26 * (size_t)h->S, ncclFloat32, root, c.deal_comm, st);
        }
        const ncclResult_t end_rc = api->GroupEnd();
        if (extract_rc) return extract_rc;
        NCK(bc_rc); NCK(end_rc);
        if (!mine || h->deal == 2) launch_import_frame(st, h->seg, c.maps, nb, c.frame, c.d_wire, c.d_best, c.d_matched);
    }
    HCK(hipGetLastError());
    if (c.timed) HCK(hipEventRecord(c.ev_t1, st));
    if (multi) HCK(hipEventRecord(c.ev_done, st));
    if (multi && h->pass_team) h->team_prev = &c;
    c.launched = true; c.waited = false; c.inflight = nb; c.nb_launched = nb;
    h->open_ctx = (int)((&c - h->ctx.data() + 1) % (ptrdiff_t)h->ctx.size());
    if (h->seq_n > 0) {
        if (h->seq_launches < 32) { h->seq_launch_us[h->seq_launches] = t_launch0 - h->seq_t0_us; h->seq_launch_host_us[h->seq_launches] = now_us() - t_launch0; h->seq_launch_n[h->seq_launches++] = nb; }
        h->seq_batches++;
    }
    return SSF_OK;
}
// Add one frame to the open batch; the batch is launched when it is full (or when its first frame is needed).
static int submit_extract(ssf_handle* h, const void* rgb, const void* depth, int on_device, const uint8_t* mask) {
    ExtractCtx& c = h->ctx[h->open_ctx];
    if (c.launched) { h->err = "extract pipeline is full: process a submitted frame first"; return SSF_ERR_STATE; }
    const int b = c.count;
    if (b == 0) {
        c.stamp0 = h->stamp + h->stamp_bias + (int)h->pending.size(); c.mask_bits = 0; c.epoch0 = h->extract_ordinal;
        if (h->deal && h->comm) { c.deal_batch = h->deal_batches++; c.mine = (int)(c.deal_batch % (long long)h->cfg.nranks) == h->cfg.rank; }
        else c.mine = true;
    }
    h->extract_ordinal++;
    const size_t P = (size_t)h->cfg.width * h->cfg.height, off = (size_t)b * c.maps.slab;
    c.in.rgb[b] = (const uint8_t*)rgb; c.in.depth[b] = (const float*)depth;
    if (!on_device && c.mine) {            // (a batch another rank extracts: its images are never looked at here)
        uint8_t* drgb = slab_shift(c.d_rgb_in, off); float* ddep = slab_shift(c.d_depth_in, off);
        HCK(hipMemcpyAsync(drgb, rgb, 3 * P, hipMemcpyHostToDevice, c.stream));
        HCK(hipMemcpyAsync(ddep, depth, 4 * P, hipMemcpyHostToDevice, c.stream));
        c.in.rgb[b] = drgb; c.in.depth[b] = ddep;
    }
    if (mask && c.mine) { HCK(hipMemcpyAsync(slab_shift(c.d_mask, off), mask, h->S, hipMemcpyHostToDevice, c.stream)); c.mask_bits |= 1u << b; }
    c.count = b + 1;
    h->pending.push_back(std::make_pair(h->open_ctx, b));
    // (inside ssf_process_sequence the first two batches are smaller: seq_batch_size)
    if (c.count == (h->seq_n > 0 ? seq_batch_size(h->seq_batches, h->batch) : h->batch)) return launch_batch(h, c);
    return SSF_OK;
}
// A frame extracted elsewhere takes a batch context of its own: its maps and rows are copied into slot 0 and the private tables
// rebuilt (k_import_frame) on the context's stream, where a local batch would run its extract chain.
static int submit_tables(ssf_handle* h, const int32_t* label, const float* plane_depth, const ssf_surfels* fr, int on_device) {
    if (h->ctx[h->open_ctx].count > 0 && !h->ctx[h->open_ctx].launched) {        // an open local batch: it goes first (frame order)
        int rc = launch_batch(h, h->ctx[h->open_ctx]);
        if (rc) return rc;
    }
    ExtractCtx& c = h->ctx[h->open_ctx];
    if (c.launched) { h->err = "extract pipeline is full: process a submitted frame first"; return SSF_ERR_STATE; }
    hipStream_t st = c.stream;
    const bool multi = h->ctx.size() > 1;
    if (multi && c.consumed_valid) HCK(hipStreamWaitEvent(st, c.ev_consumed, 0));
    c.stamp0 = h->stamp + h->stamp_bias + (int)h->pending.size(); c.mask_bits = 0; c.epoch0 = h->extract_ordinal;
    h->extract_ordinal++;                          // (the RANSAC epoch advances as if the frame had been extracted here)
    const size_t P = (size_t)h->cfg.width * h->cfg.height, S = (size_t)h->S;
    const hipMemcpyKind kind = on_device ? hipMemcpyDeviceToDevice : hipMemcpyHostToDevice;
    c.timed = false;
    HCK(hipMemcpyAsync(c.maps.label, label, 4 * P, kind, st));
    HCK(hipMemcpyAsync(c.maps.plane_depth, plane_depth, 4 * P, kind, st));
    float* w = c.d_wire;
    HCK(hipMemcpyAsync(w, fr->positions, 12 * S, kind, st)); w += 3 * S;
    HCK(hipMemcpyAsync(w, fr->colors, 12 * S, kind, st)); w += 3 * S;
    HCK(hipMemcpyAsync(w, fr->stamps, 8 * S, kind, st)); w += 2 * S;
    HCK(hipMemcpyAsync(w, fr->orientations, 36 * S, kind, st)); w += 9 * S;
    HCK(hipMemcpyAsync(w, fr->shapes, 24 * S, kind, st)); w += 6 * S;
    HCK(hipMemcpyAsync(w, fr->dims, 8 * S, kind, st)); w += 2 * S;
    HCK(hipMemcpyAsync(w, fr->confidences, 4 * S, kind, st));
    if (!on_device) HCK(hipStreamSynchronize(st));             // (pageable host buffers: the caller may reuse them on return)
    launch_import_frame(st, h->seg, c.maps, 1, c.frame, c.d_wire, c.d_best, c.d_matched);
    HCK(hipGetLastError());
    if (multi) HCK(hipEventRecord(c.ev_done, st));
    c.count = 1; c.launched = true; c.waited = false; c.inflight = 1; c.nb_launched = 1;
    h->pending.push_back(std::make_pair(h->open_ctx, 0));
    h->open_ctx = (int)((&c - h->ctx.data() + 1) % (ptrdiff_t)h->ctx.size());
    return SSF_OK;
}
// the frame held by h->active will not be fused (or has been): its slot is free again
static int retire_active(ssf_handle* h) {
    ExtractCtx* c = h->active.ctx;
    if (!c || !h->have_frame) return SSF_OK;
    h->have_frame = false;
    if (--c->inflight == 0) {
        if (h->ctx.size() > 1) { HCK(hipEventRecord(c->ev_consumed, h->stream)); c->consumed_valid = true; }
        c->launched = false; c->count = 0;
    }
    return SSF_OK;
}
// Make the oldest submitted frame the one the track/fuse chain works on.
static int activate_oldest(ssf_handle* h) {
    if (h->fusing) { h->err = "a frame is between ssf_stage_fuse_begin and ssf_stage_fuse_end"; return SSF_ERR_STATE; }
    if (h->pending.empty()) { h->err = "no submitted frame"; return SSF_ERR_STATE; }
    int rc = retire_active(h);                    // an activated frame that was never fused is dropped
    if (rc) return rc;
    const std::pair<int, int> fr = h->pending.front();
    ExtractCtx& c = h->ctx[fr.first];
    if (!c.launched) { rc = launch_batch(h, c); if (rc) return rc; }
    h->pending.pop_front();
    if (h->ctx.size() > 1 && !c.waited) { HCK(hipStreamWaitEvent(h->stream, c.ev_done, 0)); c.waited = true; }
    if (c.stamp0 + fr.second != h->stamp) { h->err = "submitted frame is out of sequence (model stamp changed while frames were pending)"; return SSF_ERR_STATE; }
    const size_t off = (size_t)fr.second * c.maps.slab;
    ActiveFrame& a = h->active;
    a.maps = batch_slot(c.maps, fr.second); a.frame = batch_slot(c.frame, off);
    a.d_best = slab_shift(c.d_best, off); a.d_matched = slab_shift(c.d_matched, off);
    a.ctx = &c; a.slot = fr.second;
    h->have_frame = true;
    return SSF_OK;
}
// submit the next frame of the sequence being processed (ssf_process_sequence)
// the last frames of a sequence form a partial batch: nothing more will join it, so it is launched at once instead of
// when the track chain gets to it (its extract would then run with nothing to hide behind: 0.7 ms at the end of a run)
static int seq_flush_tail(ssf_handle* h) {
    if (h->seq_next < h->seq_n) return SSF_OK;
    ExtractCtx& c = h->ctx[h->open_ctx];
    return (c.count > 0 && !c.launched) ? launch_batch(h, c) : SSF_OK;
}
static int seq_submit(ssf_handle* h) {
    const int i = h->seq_next;
    if (h->seq_on_device || !h->seq_upload) {
        int rc = submit_extract(h, h->seq_rgb[i], h->seq_depth[i], h->seq_on_device, nullptr);
        if (!rc) { h->seq_next++; rc = seq_flush_tail(h); }
        return rc;
    }
    Uploader& u = *h->up;
    const auto t0 = std::chrono::steady_clock::now();
    const double w0 = now_us();
    for (unsigned long long spins = 0; !u.ready(i); spins++) {
        if (u.failed.load()) { h->err = "upload of a host frame failed"; return SSF_ERR_DEVICE; }
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 10.0) {
            h->err = "upload of a host frame never finished"; return SSF_ERR_DEVICE;
        }
        std::this_thread::yield();
    }
    h->us_wait_upload += now_us() - w0;
    const int sl = i % u.ring;                    // (its copies are already in the stream of the context it goes to)
    int rc = submit_extract(h, u.d_rgb[sl], u.d_depth[sl], 1, nullptr);
    if (!rc) { h->seq_next++; rc = seq_flush_tail(h); }
    return rc;
}
static int do_extract(ssf_handle* h, const void* rgb, const void* depth, int on_device, const uint8_t* mask) {
    if (!h->pending.empty()) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    int rc = retire_active(h);
    if (!rc) rc = submit_extract(h, rgb, depth, on_device, mask);
    return rc ? rc : activate_oldest(h);
}

static void icp_start_from(IcpLoop& I, const Rt& pose) {
    I.R_init = m3_transpose(pose.R);
    I.t_init = negate(m3_mulv(I.R_init, pose.t));
    for (int i = 0; i < 16; i++) I.tf_inc[i] = (i % 5 == 0) ? 1.0 : 0.0;
}
static void icp_begin(ssf_handle* h, const float* prior) {
    h->bins_valid = false;                        // (process_oldest makes this frame's tile-sorted copy after this call)
    if (prior) h->pose = pose_from12(prior);
    IcpLoop& I = h->icp;
    // a record accumulated ahead is this frame's first iteration only if nothing it was computed from has changed
    I.ahead_seq = 0;
    if (h->ahead.valid && !prior && h->have_frame && h->active.ctx == h->ahead.ctx && h->active.slot == h->ahead.slot &&
        h->stamp == h->ahead.stamp && std::memcmp(&h->pose, &h->ahead.pose, sizeof(Rt)) == 0) I.ahead_seq = h->ahead.seq;
    h->ahead.valid = false;
    const long long nvis = (h->cfg.nranks > 1 && h->global_n_visible >= 0) ? h->global_n_visible : h->n_visible;
    I.active = nvis > 0 && h->cfg.icp_iter > 0;
    I.valid = true; I.done = !I.active; I.iter = 0;
    icp_start_from(I, h->pose);
    for (int i = 0; i < 36; i++) I.JtJ[i] = 0.0;
    I.prev_error = DBL_MAX;
    I.t_inc_stale = v3(0, 0, 0);
    h->last_icp_valid = 0; h->last_icp_iters = 0;
}
static void inc_to_float(const double* tf, M3& R, V3& t) {
    R = m3(v3((float)tf[0], (float)tf[1], (float)tf[2]), v3((float)tf[4], (float)tf[5], (float)tf[6]),
           v3((float)tf[8], (float)tf[9], (float)tf[10]));
    t = v3((float)tf[3], (float)tf[7], (float)tf[11]);
}
// device accumulate; the record lands in d_icp and in the mailbox (h_icp points at the mailbox copy)
static int icp_fetch(ssf_handle* h, unsigned long long seq, IcpGo* waiter = nullptr, unsigned long long waiter_go_seq = 0, bool* waiter_dismissed = nullptr);
// model -> camera transform of the coming iteration
static Rt icp_transform(IcpLoop& I) {
    M3 R_inc; V3 t_inc;
    inc_to_float(I.tf_inc, R_inc, t_inc);
    I.t_inc_stale = t_inc;
    Rt T; T.R = m3_mul(R_inc, I.R_init); T.t = add(m3_mulv(R_inc, I.t_init), t_inc);
    return T;
}
// the rows an ICP / association launch streams: the visible array, or its tile-sorted copy when this frame has one
static inline const SurfelSoA& icp_rows(const ssf_handle* h) { return h->bins_valid ? h->bins : h->model[h->mcur]; }
static int icp_accumulate(ssf_handle* h, bool to_host, long long* d_out = nullptr) {
    IcpLoop& I = h->icp;
    const Rt T = icp_transform(I);
    const unsigned long long seq = ++h->icp_seq;
    launch_icp(h->stream, h->cam, icp_rows(h), h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, T,
               h->d_icp_replicas, h->d_tickets + 8, d_out ? d_out : h->d_icp, h->mb_dev, seq, -1, nullptr, 0, nullptr, h->bins_valid ? 1 : 0);
    HCK(hipGetLastError());
    return to_host ? icp_fetch(h, seq) : SSF_OK;
}
// wait for mailbox record `seq` and copy it to h->h_icp_local
static void icp_release_waiting(IcpGo* slot, unsigned long long go_seq, const Rt* T, unsigned long long p2p_seq = 0, bool match = false);
// waiter: a launch made ahead that is waiting on the device for the host's word (chained ICP launches).  Before the stream
// is drained it is told to leave (*waiter_dismissed = true): it would otherwise hold the stream until its own bound expires.
static int icp_fetch(ssf_handle* h, unsigned long long seq, IcpGo* waiter, unsigned long long waiter_go_seq, bool* waiter_dismissed) {
    // the record is five 64-byte lines that each end in the sequence number (Mailbox::icp_rec): accept it when all
    // five carry `seq` and the checksum over the payload matches; anything else is a record still in flight
    const volatile unsigned long long* rec = h->mb_host->icp_rec;
    auto t0 = std::chrono::steady_clock::now();
    bool drained = false;                     // the stream has been synchronised once after a timeout
    for (unsigned long long spins = 0;; spins++) {
        bool ok = true;
        for (int j = 0; j < 5 && ok; j++) ok = __atomic_load_n(&rec[8 * j + 7], __ATOMIC_ACQUIRE) == seq;
        if (ok) {
            unsigned long long check = seq, w29 = 0;
            for (int p = 0; p < 30; p++) {
                const unsigned long long v = __atomic_load_n(&rec[8 * (p / 7) + p % 7], __ATOMIC_RELAXED);
                if (p < 29) { h->h_icp_local[p] = (long long)v; check += v; } else w29 = v;
            }
            bool still = true;                          // the lines must not have been overwritten while we read them
            for (int j = 0; j < 5 && still; j++) still = __atomic_load_n(&rec[8 * j + 7], __ATOMIC_ACQUIRE) == seq;
            if (still && check == w29) break;
        }
        if ((spins & 0xFFFF) == 0xFFFF && std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > (drained ? 1.0 : 5.0)) {
            // Nothing for 5 s.  The device may simply be slow or stalled (a cold box, a debugger, another tenant): drain the
            // stream -- however long that takes -- and look again before calling it an error; only a record that is
            // still missing once everything enqueued has run is one.
            if (drained) {
                char where[160];
                std::snprintf(where, sizeof(where), " (rank %d of %d, frame stamp %d, iteration %d, record %llu, peer exchange %llu, %d visible rows)", h->cfg.rank,
                              h->cfg.nranks, h->stamp, h->icp.iter, seq, h->p2p.seq_icp, h->n_visible);
                h->err = std::string("ICP mailbox record never arrived") + where; return SSF_ERR_DEVICE;
            }
            if (waiter) { icp_release_waiting(waiter, waiter_go_seq, nullptr); if (waiter_dismissed) *waiter_dismissed = true; waiter = nullptr; }
            hipError_t e = hipStreamSynchronize(h->stream);
            if (e != hipSuccess) { h->err = std::string("device error while waiting for the ICP record: ") + hipGetErrorString(e); return SSF_ERR_DEVICE; }
            drained = true; t0 = std::chrono::steady_clock::now();
        }
#if defined(__x86_64__)
        __builtin_ia32_pause();
#endif
    }
    h->h_icp = h->h_icp_local;
    return SSF_OK;
}
static void icp_update(ssf_handle* h, const int64_t* sums, int* again) {
    IcpLoop& I = h->icp;
    *again = 0;
    if (!I.active || I.done) return;
    I.iter++; h->last_icp_iters = I.iter;
    static const int tri[6][6] = {{0, 1, 2, 3, 4, 5}, {1, 6, 7, 8, 9, 10}, {2, 7, 11, 12, 13, 14},
                                  {3, 8, 12, 15, 16, 17}, {4, 9, 13, 16, 18, 19}, {5, 10, 14, 17, 19, 20}};
    double Jtr[6];
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) I.JtJ[i * 6 + j] = (double)sums[tri[i][j]] / SSF_ICP_SCALE_JTJ;
        Jtr[i] = (double)sums[21 + i] / SSF_ICP_SCALE_JTR;
    }
    const float r = (float)((double)sums[27] / SSF_ICP_SCALE_R);
    const float inliers = (float)sums[28];
    const double error = std::sqrt((double)(r / inliers));
    if (inliers < 100.0f) { I.valid = false; I.done = true; return; }
    double X[6], tf_iter[16];
    sym6_ldlt_solve(I.JtJ, Jtr, X);
    gn_increment(X, tf_iter);
    mat4_lmul(tf_iter, I.tf_inc);
    if (!h->cfg.icp_force_iters && error / I.prev_error > 0.9995) { I.done = true; return; }
    I.prev_error = error;
    if (I.iter >= h->cfg.icp_iter) { I.done = true; return; }
    *again = 1;
}
static void icp_end(ssf_handle* h, int* valid) {
    IcpLoop& I = h->icp;
    *valid = 0;
    if (!I.active) return;
    bool ok = I.valid;
    double cov[36];
    mat6_inverse_lu(I.JtJ, cov);
    for (int i = 0; i < 6; i++) if (cov[i * 6 + i] > h->cfg.icp_cov_thresh) { ok = false; break; }
    if (ok) {
        if (len3(I.t_inc_stale) > 0.2f) ok = false;
        else {
            M3 R_inc; V3 t_inc;
            inc_to_float(I.tf_inc, R_inc, t_inc);
            const M3 R_rel = m3_transpose(R_inc);
            const V3 t_rel = negate(m3_mulv(R_rel, t_inc));
            h->pose.t = add(m3_mulv(h->pose.R, t_rel), h->pose.t);
            h->pose.R = m3_mul(h->pose.R, R_rel);
            float R9[9] = {h->pose.R.r0.x, h->pose.R.r0.y, h->pose.R.r0.z, h->pose.R.r1.x, h->pose.R.r1.y, h->pose.R.r1.z,
                           h->pose.R.r2.x, h->pose.R.r2.y, h->pose.R.r2.z};
            renormalise_rotation<float>(R9);
            h->pose.R = m3(v3(R9[0], R9[1], R9[2]), v3(R9[3], R9[4], R9[5]), v3(R9[6], R9[7], R9[8]));
        }
    }
    *valid = ok ? 1 : 0;
    h->last_icp_valid = *valid;
    I.active = false;
}

// ---- model store upkeep -----------------------------------------------------------------------------------
static SurfelSoA soa_rows(const SurfelSoA& s, size_t r) {       // view starting at row r
    SurfelSoA v = s;
    v.pos += 3 * r; v.col += 3 * r; v.lab += 3 * r; v.stamps += 2 * r; v.r0 += 3 * r; v.r1 += 3 * r; v.r2 += 3 * r;
    v.shape += 6 * r; v.dims += 2 * r; v.conf += r;
    return v;
}
static int copy_soa(ssf_handle* h, const SurfelSoA& d, const SurfelSoA& s, size_t n) {      // device -> device, n rows
    if (n == 0) return SSF_OK;
    hipStream_t st = h->stream;
    HCK(hipMemcpyAsync(d.pos, s.pos, 12 * n, hipMemcpyDeviceToDevice, st)); HCK(hipMemcpyAsync(d.col, s.col, 12 * n, hipMemcpyDeviceToDevice, st));
    HCK(hipMemcpyAsync(d.lab, s.lab, 12 * n, hipMemcpyDeviceToDevice, st)); HCK(hipMemcpyAsync(d.stamps, s.stamps, 8 * n, hipMemcpyDeviceToDevice, st));
    HCK(hipMemcpyAsync(d.r0, s.r0, 12 * n, hipMemcpyDeviceToDevice, st)); HCK(hipMemcpyAsync(d.r1, s.r1, 12 * n, hipMemcpyDeviceToDevice, st));
    HCK(hipMemcpyAsync(d.r2, s.r2, 12 * n, hipMemcpyDeviceToDevice, st)); HCK(hipMemcpyAsync(d.shape, s.shape, 24 * n, hipMemcpyDeviceToDevice, st));
    HCK(hipMemcpyAsync(d.dims, s.dims, 8 * n, hipMemcpyDeviceToDevice, st)); HCK(hipMemcpyAsync(d.conf, s.conf, 4 * n, hipMemcpyDeviceToDevice, st));
    return SSF_OK;
}
static int oov_home(const ssf_handle* h) { return h->cfg.nb_supersurfels_max + h->S + 256; }   // head after a recentre
// compact the live out-of-view rows into the other store, span starting at oov_home (no dead slots afterwards)
static int oov_recentre(ssf_handle* h) {
    OovStore& src = h->oov[h->ocur]; OovStore& dst = h->oov[h->ocur ^ 1];
    HCK(hipMemsetAsync(dst.live, 0, (size_t)dst.cap, h->stream));
    launch_oov_compact(h->stream, src, dst, h->oov_tail - h->oov_head, oov_home(h), h->d_bc_oov, h->d_cnt, 1);
    HCK(hipGetLastError());
    h->ocur ^= 1;
    h->oov_head = oov_home(h); h->oov_tail = h->oov_head + h->oov_live;
    h->n_recentres++;
    return SSF_OK;
}
// dense [visible | out-of-view] copy of the model in h->dense (stream ordered)
static int materialise(ssf_handle* h) {
    int rc = copy_soa(h, h->dense, h->model[h->mcur], (size_t)h->n_visible);
    if (rc) return rc;
    if (h->oov_live > 0) {
        OovStore dst; dst.rows = h->dense; dst.live = h->d_live_scratch; dst.cap = h->cfg.nb_supersurfels_max;
        launch_oov_compact(h->stream, h->oov[h->ocur], dst, h->oov_tail - h->oov_head, h->n_visible, h->d_bc_oov, h->d_cnt, 0);
        HCK(hipGetLastError());
    }
    return SSF_OK;
}
// the stores <- h->dense (n rows, the first n_visible of them visible); also resets the device counters
static int store_from_dense(ssf_handle* h, int n, int n_visible) {
    h->ahead.valid = false;                       // the model is replaced: a record accumulated ahead is stale
    int rc = copy_soa(h, h->model[h->mcur], h->dense, (size_t)n_visible);
    if (rc) return rc;
    OovStore& o = h->oov[h->ocur];
    const int head = oov_home(h), n_oov = n - n_visible;
    rc = copy_soa(h, soa_rows(o.rows, (size_t)head), soa_rows(h->dense, (size_t)n_visible), (size_t)n_oov);
    if (rc) return rc;
    HCK(hipMemsetAsync(o.live, 0, (size_t)o.cap, h->stream));
    if (n_oov > 0) HCK(hipMemsetAsync(o.live + head, 1, (size_t)n_oov, h->stream));
    Counters c; std::memset(&c, 0, sizeof(c));
    c.n_model = n; c.n_visible = n_visible; c.oov_head = head; c.oov_tail = head + n_oov; c.oov_live = n_oov;
    c.last[0] = n; c.last[1] = n_visible;
    HCK(hipStreamSynchronize(h->stream));
    HCK(hipMemcpy(h->d_cnt, &c, sizeof(c), hipMemcpyHostToDevice));
    h->n_model = n; h->n_visible = n_visible; h->oov_head = c.oov_head; h->oov_tail = c.oov_tail; h->oov_live = n_oov;
    return SSF_OK;
}
// The shard sizes the ranks exchanged at the end of the last frame (read lazily at the start of the next) describe the map
// before a call that replaces it: dropped, the next frame exchanges them afresh.  Called at the TOP of every such entry
// point (ssf_set_model, ssf_apply_deformation, ssf_rehome_begin / _end), before any early return: whether a rank then
// actually rewrites its shard depends on the rank (an empty shard, nothing leaving, nothing arriving), but all ranks make
// the same call sequence, and every rank must enter the next frame in the same state -- a rank that kept the old record
// would skip an exchange its peers perform (their exchange numbers / the RCCL collective order would go out of step).
static inline void drop_shard_sizes(ssf_handle* h) { h->all_valid = false; h->all_pending = false; }

static inline P2PView p2p_view(ssf_handle* h, unsigned long long seq) { P2PView v = h->p2p.view; v.seq = seq; return v; }
// exchange != 0 (native multi-rank frame calls with the peer-to-peer backend): the association tables are traded with
// the peers by the match launch's last workgroup -- or, when no rank has anything to match, by a launch of its own
static int do_match(ssf_handle* h, int exchange = 0) {
    const long long nmodel = (h->cfg.nranks > 1 && h->global_n_model >= 0) ? h->global_n_model : h->n_model;
    const long long nvis = (h->cfg.nranks > 1 && h->global_n_visible >= 0) ? h->global_n_visible : h->n_visible;
    const bool any = nmodel > 0 && nvis > 0;               // (global quantities: the same decision on every rank)
    const int n = any ? h->n_visible : 0;
    if (exchange && h->p2p.on) {
        const P2PView pv = p2p_view(h, ++h->p2p.seq_assoc);
        launch_match(h->stream, h->cam, icp_rows(h), n, h->cc->maps.pix2, h->cc->maps.fpack, h->pose, h->cfg.range_min,
                     h->cfg.range_max, h->id_offset, h->cc->d_best, h->cc->d_matched, h->d_cand, h->S, h->bins_valid ? h->d_bin_idx : nullptr);
        launch_p2p_assoc(h->stream, pv, h->cc->d_best, h->cc->d_matched, h->mb_dev);
        HCK(hipGetLastError());
        h->bins_valid = false;
        return SSF_OK;
    }
    launch_match(h->stream, h->cam, icp_rows(h), n, h->cc->maps.pix2, h->cc->maps.fpack, h->pose, h->cfg.range_min,
                 h->cfg.range_max, h->id_offset, h->cc->d_best, h->cc->d_matched, h->d_cand, h->S, h->bins_valid ? h->d_bin_idx : nullptr);
    HCK(hipGetLastError());
    h->bins_valid = false;                        // (the fuse launch that follows rewrites the rows the copy was made from)
    return SSF_OK;
}

// update | insert | classify | reorder, all stream-ordered through the device-side counters; the
// final counters come back through the mailbox (no D2H copy, no stream synchronise).  In two halves: between them a
// sharded map exchanges the rows that crossed a tile edge (migrate: fuse_begin leaves this shard's migrant table in
// h->d_migrants; fuse_end takes the rank-reduced table, or nullptr when nothing can arrive).
static int comm_gather_counts(ssf_handle* h);
static int fuse_begin(ssf_handle* h, int migrate) {
    const long long nmodel_g = (h->cfg.nranks > 1 && h->global_n_model >= 0) ? h->global_n_model : h->n_model;
    const long long nvis_g = (h->cfg.nranks > 1 && h->global_n_visible >= 0) ? h->global_n_visible : h->n_visible;
    SurfelSoA& M = h->model[h->mcur];
    h->fuse_first = !(nmodel_g > 0);
    h->fuse_migrate = migrate && h->cfg.nranks > 1 && !h->fuse_first;
    if (nmodel_g > 0) {
        // out-of-view store upkeep before the frame's launches: room in front for the rows that leave the view (at
        // most all visible rows), room behind for out-of-view insertions, and not too many dead slots in the span
        {
            const int span = h->oov_tail - h->oov_head;
            if (h->oov_head < h->n_visible + h->S + 256 || h->oov[h->ocur].cap - h->oov_tail < 2 * h->S + 256 ||
                span > h->oov_live + h->oov_live / 4 + 65536) { int rc2 = oov_recentre(h); if (rc2) return rc2; }
        }
        // a single shard: the fuse launch ends without its three-trip tail, the move kernel works the counters out from the class
        // totals (MoveTotals: the counts the frame starts from are mirrored here).  A sharded map keeps the tail: arrivals from
        // other ranks change the counters between the two launches (launch_migrate_in).
        h->fuse_totals.from_tot = (h->cfg.nranks == 1 && !h->comm && !h->p2p.on && !h->fuse_migrate && h->move_totals_on) ? 1 : 0;
        h->fuse_totals.nv = h->n_visible; h->fuse_totals.head_old = h->oov_head; h->fuse_totals.tail_old = h->oov_tail;
        PartitionWs& ws = h->fuse_ws;
        {
            uint32_t* set = h->d_part + (size_t)h->part_set * h->part_words;
            ws.sup_vis = set; ws.sup_oov = set + h->part_sup_vis; ws.tot = ws.sup_oov + h->part_sup_oov;
            ws.ticket = h->d_part_ticket; ws.other = h->d_part + (size_t)(h->part_set ^ 1) * h->part_words; ws.words = h->part_words;
            h->part_set ^= 1;
        }
        // update | insert | classification of every row | publication of the counters: one launch
        launch_fuse(h->stream, M, h->cc->frame, h->pose, h->stamp, h->id_offset, h->n_visible, h->cc->d_best, h->cc->d_matched, h->d_cand,
                    h->S, nvis_g > 0 ? 1 : 0, h->cfg.nb_supersurfels_max, h->cfg.rank, h->cfg.nranks, h->cfg.shard_tile, h->d_cnt,
                    h->cam, h->oov[h->ocur], h->oov_tail - h->oov_head, h->cc->maps.plane_depth, h->cfg.delta_t,
                    h->cfg.conf_thresh, h->cfg.range_min, h->cfg.range_max, h->d_state, h->d_state_oov, h->d_bc_oov, ws,
                    h->fuse_migrate ? 1 : 0, h->fuse_totals.from_tot);
        if (h->fuse_migrate)
            launch_pack_emigrants(h->stream, M, h->cc->d_best, h->cc->d_matched, h->id_offset, h->n_visible, h->d_state, h->S,
                                  nvis_g > 0 ? 1 : 0, h->cfg.nranks, h->cfg.shard_tile, h->d_migrants);
    }
    HCK(hipGetLastError());
    h->fusing = true;
    return SSF_OK;
}
static int fuse_end(ssf_handle* h, const int32_t* d_table, ssf_frame_result* out) {
    SurfelSoA& M = h->model[h->mcur];
    const unsigned long long seq = ++h->cnt_seq;
    h->fusing = false;                           // (also on every error path below: the frame is over either way)
    if (!h->fuse_first) {
        const PartitionWs& ws = h->fuse_ws;
        if (h->fuse_migrate && d_table)
            launch_migrate_in(h->stream, M, d_table, h->S, h->cfg.rank, h->cfg.nb_supersurfels_max, h->d_cnt, h->cam, h->pose, h->stamp,
                              h->cc->maps.plane_depth, h->cfg.delta_t, h->cfg.conf_thresh, h->cfg.range_min, h->cfg.range_max,
                              h->d_state, ws);
        // The rows the move kernel writes to the new visible array are the rows the next frame's first ICP iteration
        // reads, under a transform that is known now (the pose just estimated, when the caller supplies no prior):
        // if that frame's extract has finished, the move kernel accumulates the record on the way (k_move_rows<true>).
        NextFrameIcp next{};
        P2PView next_pv{};
        bool have_next = false;
        if (h->icp_ahead && (h->p2p.on || h->ahead_tuner.current() == 1) && !h->comm && (h->cfg.nranks == 1 || h->p2p.on) && h->cfg.icp_iter > 0 && !h->pending.empty()) {
            ExtractCtx& nc = h->ctx[h->pending.front().first];
            const int nslot = h->pending.front().second;
            const bool multi = h->ctx.size() > 1;         // one context: extract ran on the track stream itself
            // Only when that frame's extract HAS finished (round 5).  Frames of the batch being consumed are ready by construction;
            // at a batch boundary the next context's event is asked.  Until round 4 the track stream was made to wait for it here --
            // in a sequence's fill phase that parked the row moves behind a batch that was still 100-200 us from done, and the
            // driver's 20-frame form ran 4 % slower with the fusion than without (7680-7730 against 8030-8060 frames/s, same box,
            // alternated twice: profiles/track_chain_r05.txt); in the steady state the next batch is ready and nothing changes.
            bool ready = nc.launched && (!multi || nc.waited);
            // (peer-to-peer shards: every rank must take the SAME form for a frame -- a rank that fused would spin in the exchange until
            //  its peer's later launch publishes -- so there the decision stays host-deterministic: wait for the event, as until round 4)
            if (nc.launched && !ready && (h->icp_ahead_mode == 2 || h->p2p.on || hipEventQuery(nc.ev_done) == hipSuccess)) {
                HCK(hipStreamWaitEvent(h->stream, nc.ev_done, 0)); nc.waited = true; ready = true;
            }
            (void)hipGetLastError();                      // (hipErrorNotReady of the query is not an error)
            if (ready) {
                const FrameMaps nm = batch_slot(nc.maps, nslot);
                IcpLoop first;
                icp_start_from(first, h->pose);
                next.pix2 = nm.pix2; next.fpack = nm.fpack; next.T = icp_transform(first);
                next.replicas = h->d_icp_replicas; next.ticket = h->d_tickets + 8; next.sums = h->d_icp;
                next.seq = ++h->icp_seq;
                if (h->p2p.on) { next_pv = p2p_view(h, ++h->p2p.seq_icp); next.pv = &next_pv; }
                h->ahead.valid = true; h->ahead.seq = next.seq; h->ahead.ctx = &nc; h->ahead.slot = nslot;
                h->ahead.stamp = h->stamp + 1; h->ahead.pose = h->pose;
                have_next = true;
            }
        }
        // move: the host continues once the counters arrive (published by the fuse launch), the row moves of this
        // frame overlap the host-side launch work of the next one (stream order keeps every later reader of the
        // model behind them)
        // (a sharded map: the shard sizes of all ranks are exchanged now -- the counters are final -- so that the next
        // frame does not have to wait for the row moves to learn them)
        if (h->comm || h->p2p.on) { int rg = comm_gather_counts(h); if (rg) return rg; }
        launch_move_rows(h->stream, h->cam, M, h->model[h->mcur ^ 1], h->oov[h->ocur], h->n_visible + (h->fuse_migrate ? 2 : 1) * h->S, h->oov_tail - h->oov_head,
                         h->d_state, h->d_state_oov, h->d_bc_oov, ws, h->d_cnt, h->mb_dev, seq, have_next ? &next : nullptr,
                         h->fuse_totals.from_tot ? &h->fuse_totals : nullptr);
        h->mcur ^= 1;
    } else {
        launch_first_frame(h->stream, M, h->cc->frame, h->pose, h->S, h->cfg.nb_supersurfels_max, h->cfg.rank, h->cfg.nranks,
                           h->cfg.shard_tile, h->d_cnt);
        launch_publish_counts(h->stream, h->d_cnt, 0, h->mb_dev, seq);
        if (h->comm || h->p2p.on) { int rg = comm_gather_counts(h); if (rg) return rg; }
    }
    HCK(hipGetLastError());
    { int rr = retire_active(h); if (rr) return rr; }     // last reader of this frame's buffers is enqueued
    h->stamp_bias = 1;                                     // the frame being fused still holds h->stamp
    while (h->seq_next < h->seq_n && !h->ctx[h->open_ctx].launched) {      // see seq_rgb
        int rs = seq_submit(h);
        if (rs) { h->stamp_bias = 0; return rs; }
    }
    h->stamp_bias = 0;
    int rc = wait_seq(h, &h->mb_host->cnt_seq, seq);
    if (rc) return rc;
    Counters c;
    for (int attempt = 0;; attempt++) {
        unsigned long long check = seq;
        const int* srcw = reinterpret_cast<const int*>(&h->mb_host->cnt);
        int* dstw = reinterpret_cast<int*>(&c);
        for (int i = 0; i < (int)(sizeof(Counters) / sizeof(int)); i++) {
            dstw[i] = __atomic_load_n(&srcw[i], __ATOMIC_RELAXED);
            check += (unsigned long long)(unsigned int)dstw[i];
        }
        if (check == __atomic_load_n(&h->mb_host->cnt_check, __ATOMIC_ACQUIRE)) break;
        if (attempt > 100000) { h->err = "counter mailbox record failed its checksum"; return SSF_ERR_DEVICE; }
    }
    h->n_model = c.n_model; h->n_visible = c.n_visible;
    h->oov_head = c.oov_head; h->oov_tail = c.oov_tail; h->oov_live = c.oov_live;
    if (__atomic_load_n(&h->mb_host->extract_abort, __ATOMIC_ACQUIRE) != 0u) {
        __atomic_store_n(&h->mb_host->extract_abort, 0u, __ATOMIC_RELEASE);
        h->err = "a team launch of the relabelling passes could not get all its workgroups onto the GPU and gave up (the GPU is oversubscribed: "
                 "several processes?); the frames of that batch are invalid";
        return SSF_ERR_DEVICE;
    }
    if (h->p2p.on && __atomic_load_n(&h->mb_host->p2p_timeout, __ATOMIC_ACQUIRE) != 0u) {
        __atomic_store_n(&h->mb_host->p2p_timeout, 0u, __ATOMIC_RELEASE);       // reported once; a later frame starts clean
        h->err = "a peer's association / migrant tables never arrived (peer-to-peer exchange)"; return SSF_ERR_DEVICE;
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        pose_to12(h->pose, out->pose);
        out->icp_valid = h->last_icp_valid; out->icp_iters = h->last_icp_iters;
        out->n_model = h->n_model; out->n_visible = h->n_visible; out->n_removed = c.n_removed;
        out->n_inserted = c.n_inserted; out->n_updated = c.n_updated; out->stamp = h->stamp;
        ExtractCtx* ec = h->active.ctx;            // extract time of the batch this frame came in, per frame
        float ms;
        if (h->cfg.profile != 0 && ec && ec->timed && hipEventElapsedTime(&ms, ec->ev_t0, ec->ev_t1) == hipSuccess)
            out->stage_ms[0] = ms / (float)ec->nb_launched;
    }
    h->stamp++;
    h->global_n_model = -1; h->global_n_visible = -1;
    if (h->cfg.profile == 1) { HCK(hipStreamSynchronize(h->stream)); timer_collect(&h->timer); }
    return SSF_OK;
}

static int do_fuse(ssf_handle* h, ssf_frame_result* out) {          // no exchange of rows (single shard, or ssf_stage_fuse)
    int rc = fuse_begin(h, 0);
    return rc ? rc : fuse_end(h, nullptr, out);
}

// ---- multi-GPU exchanges (native RCCL on the track stream) -------------------------------------------------
// enqueue the all-gather of every rank's Counters::last and its publication to the mailbox
static int comm_gather_counts(ssf_handle* h) {
    const unsigned long long seq = ++h->all_seq;
    if (h->p2p.on) launch_p2p_counts(h->stream, p2p_view(h, ++h->p2p.seq_cnt), h->d_cnt, h->mb_dev, seq);
    else {
        RcclApi* api = rccl_api();
        { ScopedKernel sk("exchange_counts", h->stream);      // (cfg.profile = 1: the collective's time on the track stream, bench.py's exchange_us_per_frame)
          NCK(api->AllGather(h->d_cnt->last, h->d_all5, 5, ncclInt32, h->comm, h->stream)); }
        launch_publish_all_counts(h->stream, h->d_all5, h->cfg.nranks, h->mb_dev, seq);
    }
    HCK(hipGetLastError());
    h->all_pending = true;
    return SSF_OK;
}
// the shard sizes of all ranks after the previous frame -> global counts and this shard's id offset
static int comm_counts(ssf_handle* h) {
    if (!h->all_valid && !h->all_pending) { int rc = comm_gather_counts(h); if (rc) return rc; }
    if (h->all_pending) {
        int rc = wait_seq(h, &h->mb_host->all_seq, h->all_seq);
        if (rc) return rc;
        const int n = 5 * h->cfg.nranks;
        for (int attempt = 0;; attempt++) {
            unsigned long long check = h->all_seq;
            for (int i = 0; i < n; i++) {
                const int v = __atomic_load_n(&h->mb_host->all_cnt[i], __ATOMIC_RELAXED);
                h->all_cnt[i] = v; check += (unsigned long long)(unsigned int)v;
            }
            if (check == __atomic_load_n(&h->mb_host->all_check, __ATOMIC_ACQUIRE)) break;
            if (attempt > 100000) { h->err = "shard-size mailbox record failed its checksum"; return SSF_ERR_DEVICE; }
        }
        h->all_pending = false; h->all_valid = true;
    }
    long long gm = 0, gv = 0, off = 0;
    for (int r = 0; r < h->cfg.nranks; r++) {
        gm += h->all_cnt[5 * r]; gv += h->all_cnt[5 * r + 1];
        if (r < h->cfg.rank) off += h->all_cnt[5 * r + 1];
    }
    h->global_n_model = gm; h->global_n_visible = gv; h->id_offset = off;
    return SSF_OK;
}

// ---- chained ICP launches --------------------------------------------------------------------------------
// launch the NEXT iteration now, to wait on the device for its transform; returns the sequence number of its record
// (match_capable: the launch can be told to do the frame's association instead of an iteration -- SSF_ICP_GO_MATCH)
static bool icp_waiter_can_match(const ssf_handle* h) {
    static const bool off = SSF_ENV_SET("NO_MATCH_IN_WAITER");          // (measurement switch)
    return !off && !h->p2p.on && !h->comm && h->cfg.nranks == 1 && h->cfg.profile == 0 && icp_variant_mode() == 0;
}
static int icp_launch_waiting(ssf_handle* h, unsigned long long* seq_out, IcpGo** slot_out, unsigned long long* go_seq_out) {
    const unsigned long long seq = ++h->icp_seq;
    const unsigned long long go_seq = ++h->go_count;
    IcpGo* slot = h->go + (go_seq % SSF_ICP_GO_SLOTS);
    h->wait_launched_us = now_us();               // (before the launch call: no workgroup of it can have started waiting earlier)
    Rt none; none.R = m3_identity(); none.t = v3(0, 0, 0);
    const P2PView pv = h->p2p.view;               // (the number of the peer exchange arrives with the go word)
    const MatchArgs ma{h->cfg.range_min, h->cfg.range_max, h->id_offset, h->cc->d_best, h->cc->d_matched, h->d_cand};
    launch_icp(h->stream, h->cam, icp_rows(h), h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, none,
               h->d_icp_replicas, h->d_tickets + 8, h->d_icp, h->mb_dev, seq, -1, slot, go_seq, h->p2p.on ? &pv : nullptr, h->bins_valid ? 1 : 0,
               icp_waiter_can_match(h) ? &ma : nullptr);
    HCK(hipGetLastError());
    *seq_out = seq; *slot_out = slot; *go_seq_out = go_seq;
    return SSF_OK;
}
// drain the write-combining buffers: the stores above become visible to the device in order, now
static inline void store_fence() {
#if defined(__x86_64__)
    __builtin_ia32_sfence();
#else
    __atomic_thread_fence(__ATOMIC_SEQ_CST);
#endif
}
// A chained launch that never got its word (host stalled past the kernel's bound, or the record never arrived) may
// have left the arrival counters / replica records of the ICP reduction half filled: drain the stream, put them back
// to rest and stop chaining launches on this handle.
static void icp_chain_reset(ssf_handle* h) {
    (void)hipStreamSynchronize(h->stream);
    (void)hipMemsetAsync(h->d_icp_replicas, 0, 2 * SSF_ICP_REPLICAS * 32 * sizeof(long long), h->stream);
    (void)hipMemsetAsync(h->d_tickets, 0, 512 * sizeof(unsigned int), h->stream);
    (void)hipStreamSynchronize(h->stream);
    h->icp_chain = false; h->ahead.valid = false;
}
// the host's word to a waiting launch: its transform and "go", or "no further iteration"
static void icp_release_waiting(IcpGo* slot, unsigned long long go_seq, const Rt* T, unsigned long long p2p_seq, bool match) {
    volatile IcpGo* s = slot;
    const unsigned long long want = go_seq & 0xFFFFFFFFull;
    if (T) {
        const float v[12] = {T->R.r0.x, T->R.r0.y, T->R.r0.z, T->R.r1.x, T->R.r1.y, T->R.r1.z, T->R.r2.x, T->R.r2.y, T->R.r2.z,
                             T->t.x, T->t.y, T->t.z};
        uint32_t w[12]; memcpy(w, v, sizeof w);
        uint32_t sum = (uint32_t)p2p_seq * icp_go_word_weight(14) + (uint32_t)(p2p_seq >> 32) * icp_go_word_weight(15);
        for (unsigned int i = 0; i < 12; i++) sum += w[i] * icp_go_word_weight(i);
        // the whole line, then ONE fence: the write-combining buffer goes out as one 64-byte write (were it ever split, the
        // checksum in the flag word keeps the kernel polling until the rest has landed)
        for (int i = 0; i < 12; i++) s->T[i] = v[i];
        s->x = p2p_seq;
        s->flag = want | ((unsigned long long)((sum >> 2) & SSF_ICP_GO_CHECK_MASK) << 32) | (match ? SSF_ICP_GO_MATCH : 0ull);
    } else s->flag = want | ((unsigned long long)icp_go_abort_check((unsigned int)want) << 32) | SSF_ICP_GO_ABORT;
    store_fence();
}

// ICP + association + fusion of the oldest submitted frame, on the track stream
static int process_oldest(ssf_handle* h, const float* prior, ssf_frame_result* out) {
    TimerScope ts(h);
    int rc = activate_oldest(h);
    if (rc) return rc;
    const double t_a = now_us();
    const int kf = h->seq_n > 0 ? h->seq_k : -1;          // frame number inside a sequence (debug marks)
    if (kf >= 0 && kf < 64) { h->seq_mark_us[0][kf] = t_a - h->seq_t0_us; h->seq_mark_us[1][kf] = 0; }
    if (h->ctx.size() > 1 && hipEventQuery(h->cc->ctx->ev_done) == hipSuccess) h->host_us[4] += 1;
    bool first_it = true;
    const bool timing = h->cfg.profile != 0 && h->cc->ctx->timed;     // stage split costs an event synchronise: opt-in
    if (timing) HCK(hipEventRecord(h->ev[1], h->stream));
    RcclApi* api = h->comm ? rccl_api() : nullptr;
    const bool exchanging = h->comm || h->p2p.on;          // a shard of a map that runs its exchanges natively
    if (exchanging) { rc = comm_counts(h); if (rc) return rc; }
    icp_begin(h, prior);
    // a large visible set: its ICP / association fields once more, sorted by the image tile they project to under the
    // frame's initial transform (ssf_track_fuse.hip, k_bin_*): the iterations and the association stream that copy
    h->bins_valid = false;
    // chained launches (single GPU, kernels not individually timed): while iteration i runs, iteration i + 1 is
    // already launched and waits on the device for its transform
    // (with the peer-to-peer exchange too: there an iteration is one launch as well; every rank takes the same decisions)
    const bool chain = h->icp_chain && h->go && !h->comm && (h->cfg.nranks == 1 || h->p2p.on) && h->cfg.profile != 1;
    if (h->icp.active && h->bin_min_rows >= 0 && h->n_visible >= h->bin_min_rows && h->n_visible > 0 && !exchanging && h->cfg.nranks == 1 &&
        bin_buffer_words(h->cam, 1) != 0) {
        if (!h->d_bin_idx) {                       // first use: the copy's buffers (48 B per row of capacity; d_bin_idx: only the flag "this is the sorted copy" of launch_match)
            const size_t N = (size_t)h->cfg.nb_supersurfels_max, bw = bin_buffer_words(h->cam, N);
            const bool ok = bw && dalloc(h, &h->bins.pos, 12 * N) && dalloc(h, &h->d_bin_idx, 1) && dalloc(h, &h->d_bin_count, bw) && dalloc(h, &h->d_bin_cursor, bw);
            if (!ok) { h->err = "allocation of the tile-sorted copy failed"; return SSF_ERR_DEVICE; }
        }
        Rt T0; T0.R = h->icp.R_init; T0.t = h->icp.t_init;
        // In front of the loop, on the track stream.  (Measured and removed, round 6: the sort on a stream of its own beside the
        // frame's first two iterations, the launch made ahead for iteration 3 the first to wait for it -- 2 003-2 026 frames/s at
        // BASELINE config 3 against 2 351-2 366 in front and 2 286-2 301 without the copy: launches made ahead hold their workgroups'
        // places while they wait for the host's word, and the sort's three launches queue behind them.  profiles/config3_sorted_rows_r06.txt)
        launch_bin_rows(h->stream, h->cam, h->model[h->mcur], h->n_visible, T0, h->d_bin_count, h->d_bin_cursor, h->bins);
        HCK(hipGetLastError());
        h->bins_valid = true;
    }
    int again = h->icp.active ? 1 : 0, valid = 0;
    bool waiting = false; unsigned long long wait_seq_rec = 0, wait_go_seq = 0; IcpGo* wait_slot = nullptr;
    while (again) {
        if (chain) {
            unsigned long long seq_rec;
            if (h->icp.ahead_seq) { seq_rec = h->icp.ahead_seq; h->icp.ahead_seq = 0; }     // iteration 1 came from the move kernel
            else if (waiting) {                                                                 // this iteration is already on the device
                const Rt T = icp_transform(h->icp);
                icp_release_waiting(wait_slot, wait_go_seq, &T, h->p2p.on ? ++h->p2p.seq_icp : 0);
                seq_rec = wait_seq_rec; waiting = false;
            } else {
                const Rt T = icp_transform(h->icp);
                seq_rec = ++h->icp_seq;
                P2PView pv{};
                if (h->p2p.on) pv = p2p_view(h, ++h->p2p.seq_icp);
                launch_icp(h->stream, h->cam, icp_rows(h), h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, T,
                           h->d_icp_replicas, h->d_tickets + 8, h->d_icp, h->mb_dev, seq_rec, -1, nullptr, 0, h->p2p.on ? &pv : nullptr, h->bins_valid ? 1 : 0);
                HCK(hipGetLastError());
            }
            // the next iteration, should there be one (the loop may run cfg.icp_iter iterations at most) -- and behind the LAST
            // iteration the loop allows, a launch that can only be told to do the association: a loop that ends at the cap
            // (BASELINE config 3: ten forced iterations) then starts its association ~1 us after the host's last step instead of
            // a launch latency later (11-13 us between the tenth k_icp and k_match in the round-4 traces), like one that converges
            if (h->icp.iter + 1 < h->cfg.icp_iter || (!timing && icp_waiter_can_match(h))) {
                rc = icp_launch_waiting(h, &wait_seq_rec, &wait_slot, &wait_go_seq);
                if (rc) return rc;
                waiting = true;
            }
            bool dismissed = false;
            rc = icp_fetch(h, seq_rec, waiting ? wait_slot : nullptr, wait_go_seq, &dismissed);
            if (dismissed) waiting = false;           // (the next iteration, if any, is launched afresh)
            if (rc) { if (waiting) icp_release_waiting(wait_slot, wait_go_seq, nullptr); icp_chain_reset(h); return rc; }
            if (first_it) { h->host_us[5] += now_us() - t_a; first_it = false; if (kf >= 0 && kf < 64) h->seq_mark_us[1][kf] = now_us() - h->seq_t0_us; }
            icp_update(h, (const int64_t*)h->h_icp, &again);
            continue;
        }
        if (h->p2p.on) {
            // one launch: its last workgroup trades the shard record with the peers through the exchange regions and
            // publishes the SUM over the ranks (exact: int64)
            const Rt T = icp_transform(h->icp);
            const unsigned long long seq = ++h->icp_seq;
            const P2PView pv = p2p_view(h, ++h->p2p.seq_icp);
            launch_icp(h->stream, h->cam, icp_rows(h), h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, T,
                       h->d_icp_replicas, h->d_tickets + 8, h->d_icp, h->mb_dev, seq, -1, nullptr, 0, &pv, h->bins_valid ? 1 : 0);
            HCK(hipGetLastError());
            rc = icp_fetch(h, seq);
        } else if (h->comm) {
            // shard record -> SUM over the ranks in HBM (exact: int64) -> mailbox -> host solve
            rc = icp_accumulate(h, false);
            if (rc) return rc;
            { ScopedKernel sk("exchange_icp_record", h->stream);
              NCK(api->AllReduce(h->d_icp, h->d_icp, SSF_ICP_RECORD, ncclInt64, ncclSum, h->comm, h->stream)); }
            const unsigned long long seq = ++h->icp_seq;
            launch_publish_icp(h->stream, h->d_icp, h->mb_dev, seq);
            HCK(hipGetLastError());
            rc = icp_fetch(h, seq);
        } else if (h->icp.ahead_seq) {
            // first iteration: the record was accumulated by the previous frame's move kernel
            rc = icp_fetch(h, h->icp.ahead_seq);
            h->icp.ahead_seq = 0;
        } else
            rc = icp_accumulate(h, true);
        if (rc) return rc;
        if (first_it) { h->host_us[5] += now_us() - t_a; first_it = false; }
        icp_update(h, (const int64_t*)h->h_icp, &again);
    }
    // no further iteration: the launch made ahead leaves -- or, told the frame's final pose, does the association on its way
    // out (the rows, tables and frame it was launched with are the ones the association reads)
    bool matched_by_waiter = false;
    if (waiting && !timing && icp_waiter_can_match(h)) {
        icp_end(h, &valid);
#ifdef SSF_EXPERIMENTS
        if (h->dbg_stall_before_match_us > 0) usleep((useconds_t)h->dbg_stall_before_match_us);      // (test hook of the lab build: a stalled host thread)
#endif
        icp_release_waiting(wait_slot, wait_go_seq, &h->pose, 0, true);
        matched_by_waiter = true;
        // The word has no acknowledgement.  A waiting workgroup gives up after SSF_ICP_GO_WAIT_TICKS (0.25 s) and tells the rest
        // of its launch to leave; if this thread was stalled that long (descheduled, a debugger, SIGSTOP) between the launch and
        // the store above, the word may have found only the late-dispatched part of the grid and the association would cover a
        // subset of the rows -- silently.  The host's own clock bounds the device's: no workgroup started waiting before
        // wait_launched_us, so below 0.1 s on this side nobody has given up.  Past it the association is run again as a launch
        // of its own: match_row only takes minima and sets flags, so a partial pass followed by a full one is the full one.
        if (now_us() - h->wait_launched_us > 100000.0) { matched_by_waiter = false; h->n_waiter_match_repairs++; }
    } else {
        if (waiting) icp_release_waiting(wait_slot, wait_go_seq, nullptr);
        icp_end(h, &valid);
    }
    const double t_b = now_us();
    if (kf >= 0 && kf < 64) h->seq_mark_us[2][kf] = t_b - h->seq_t0_us;
    if (timing) HCK(hipEventRecord(h->ev[2], h->stream));
    if (matched_by_waiter) h->n_waiter_matches++;
    else { rc = do_match(h, 1); if (rc) return rc; }
    if (h->comm) {
        // best key over the ranks (keys < 2^63: signed MIN == unsigned MIN), matched = OR over the ranks
        ScopedKernel sk("exchange_association", h->stream);
        NCK(api->AllReduce(h->cc->d_best, h->cc->d_best, h->S, ncclInt64, ncclMin, h->comm, h->stream));
        NCK(api->AllReduce(h->cc->d_matched, h->cc->d_matched, h->S, ncclUint8, ncclMax, h->comm, h->stream));
    }
    ssf_frame_result r;
    if (exchanging) {
        // rows whose fused position crossed a tile edge move to the rank that owns their new tile: every rank's
        // migrant table (one slot per frame supersurfel, at most one rank fills a slot) is summed in HBM
        static const int migrate = SSF_ENV_SET("NO_MIGRATE") ? 0 : 1;                  // (bisecting switch of tools/p2p_first_frame_stress.py)
        rc = fuse_begin(h, migrate);
        if (rc) { h->fusing = false; return rc; }
        if (h->fuse_migrate && h->p2p.on) launch_p2p_migrants(h->stream, p2p_view(h, ++h->p2p.seq_migr), h->d_migrants, h->d_tickets + 320, h->mb_dev);
        else if (h->fuse_migrate) {
            ScopedKernel sk("exchange_migrants", h->stream);
            const ncclResult_t nr = api->AllReduce(h->d_migrants, h->d_migrants, (size_t)SSF_MIGRANT_WORDS * h->S, ncclInt32, ncclSum, h->comm, h->stream);
            if (nr != ncclSuccess) {          // the frame cannot be completed: the handle must not stay "between the two halves"
                h->fusing = false;
                h->err = std::string("ncclAllReduce (migrant table): ") + (api->GetErrorString ? api->GetErrorString(nr) : "RCCL error");
                return SSF_ERR_DEVICE;
            }
        }
        rc = fuse_end(h, h->d_migrants, &r);
    } else
        rc = do_fuse(h, &r);
    if (rc) return rc;
    if (h->pending.empty()) h->ahead_tuner.sequence_break(); else h->ahead_tuner.frame_done(now_us(), r.icp_iters);
    h->host_us[1] += t_b - t_a; h->host_us[2] += now_us() - t_b; h->host_us[3] += 1;
    if (timing) {
        HCK(hipEventRecord(h->ev[3], h->stream));
        HCK(hipEventSynchronize(h->ev[3]));
        float ms;
        ExtractCtx* ec = h->cc->ctx;      // extract time of the batch this frame came in, per frame
        if (hipEventElapsedTime(&ms, ec->ev_t0, ec->ev_t1) == hipSuccess) r.stage_ms[0] = ms / (float)ec->nb_launched;
        if (hipEventElapsedTime(&ms, h->ev[1], h->ev[2]) == hipSuccess) r.stage_ms[1] = ms;
        if (hipEventElapsedTime(&ms, h->ev[2], h->ev[3]) == hipSuccess) r.stage_ms[2] = ms;
    }
    if (out) *out = r;
    return SSF_OK;
}
static int process_frame_impl(ssf_handle* h, const void* rgb, const void* depth, int on_device, const float* prior,
                              const uint8_t* mask, ssf_frame_result* out) {
    if (!h->pending.empty()) { h->err = "frames are pending in the extract pipeline: use ssf_process_submitted"; return SSF_ERR_STATE; }
    int rc;
    { TimerScope ts(h); rc = submit_extract(h, rgb, depth, on_device, mask); }
    return rc ? rc : process_oldest(h, prior, out);
}

// device temporaries of one call: freed on every exit path
struct DevTemps {
    std::vector<void*> p;
    template <typename T> hipError_t take(T** out, size_t bytes) {
        void* q = nullptr;
        const hipError_t e = hipMalloc(&q, bytes ? bytes : 1);
        if (e == hipSuccess) { p.push_back(q); *out = (T*)q; }
        return e;
    }
    ~DevTemps() { for (void* q : p) (void)hipFree(q); }
};

// ---- C ABI ----------------------------------------------------------------------------------------------
// (kernels of ssf_stream_copy_rate, further down)
typedef float f4v __attribute__((ext_vector_type(4)));
template <int U, bool NT, bool ONE_PASS>
__global__ __launch_bounds__(256) void k_stream_copy(const f4v* __restrict__ in, f4v* __restrict__ out, size_t n) {
    // (n is a multiple of U x the grid's threads -- ssf_stream_copy_rate rounds to 256 MiB and refuses less: the first round is unguarded)
    const size_t stride = ONE_PASS ? (size_t)256 : (size_t)gridDim.x * 256;
    size_t i = ONE_PASS ? (size_t)blockIdx.x * 256 * U + threadIdx.x : (size_t)blockIdx.x * 256 + threadIdx.x;
    do {
        f4v v[U];
#pragma unroll
        for (int k = 0; k < U; k++) v[k] = NT ? __builtin_nontemporal_load(&in[i + k * stride]) : in[i + k * stride];
#pragma unroll
        for (int k = 0; k < U; k++) { if (NT) __builtin_nontemporal_store(v[k], &out[i + k * stride]); else out[i + k * stride] = v[k]; }
        i += U * stride;
    } while (!ONE_PASS && i + (U - 1) * stride < n);
}

extern "C" {

int ssf_abi_version(void) { return SSF_ABI_VERSION; }
const char* ssf_backend_name(void) { return "hip-gfx950"; }

void ssf_default_config(ssf_config* c) {       // default arguments of initialize, supersurfel_fusion.hpp:46-74
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

void ssf_destroy(ssf_handle* h) {
    if (!h) return;
    if (h->up) {
        h->up->stop.store(1);
        h->up->join();
        h->up->shutdown();
        for (auto q : h->up->p_rgb) if (q) (void)hipHostFree(q);
        for (auto q : h->up->p_depth) if (q) (void)hipHostFree(q);
        delete h->up; h->up = nullptr;
    }
    for (auto& c : h->ctx) if (c.stream) (void)hipStreamSynchronize(c.stream);
    if (h->stream) (void)hipStreamSynchronize(h->stream);
    for (auto& c : h->ctx) if (c.deal_comm) { RcclApi* api = rccl_api(); if (api) (void)api->CommDestroy(c.deal_comm); c.deal_comm = nullptr; }
    if (h->comm) { RcclApi* api = rccl_api(); if (api) (void)api->CommDestroy(h->comm); h->comm = nullptr; }
    for (void* q : h->p2p.opened) (void)hipIpcCloseMemHandle(q);
    if (h->p2p.region) (void)hipFree(h->p2p.region);
    for (auto& c : h->ctx) {
        for (int n = 0; n <= SSF_MAX_BATCH; n++) { if (c.exec[n]) (void)hipGraphExecDestroy(c.exec[n]); if (c.graph[n]) (void)hipGraphDestroy(c.graph[n]); }
        hipEvent_t evs[4] = {c.ev_done, c.ev_consumed, c.ev_t0, c.ev_t1};
        for (hipEvent_t e : evs) if (e) (void)hipEventDestroy(e);
        if (c.own_stream && c.stream) stream_pool().give(c.stream, h->cfg.device_id, c.stream_prio);          // (synchronised above)
    }
    if (h->capture_stream) stream_pool().give(h->capture_stream, h->cfg.device_id, StreamPool::CAPTURE);
    if (!h->guarded.empty() && !SSF_ENV_SET("GUARD_ONLY")) check_guards(h);
    for (void* p : h->allocs) (void)hipFree(p);
    if (h->mb_host) (void)hipHostFree(h->mb_host);
    for (int i = 0; i < 4; i++) if (h->ev[i]) (void)hipEventDestroy(h->ev[i]);
    for (auto& r : h->timer.pool_free) { (void)hipEventDestroy(r.e0); (void)hipEventDestroy(r.e1); }
    if (h->own_stream && h->stream) { (void)hipStreamSynchronize(h->stream); stream_pool().give(h->stream, h->cfg.device_id, h->stream_prio); }
    delete h;
}

int ssf_create(const ssf_config* cfg, ssf_handle** out) {
    if (!cfg || !out) { g_create_err = "null argument"; return SSF_ERR_INVALID_ARG; }
    if (cfg->width <= 0 || cfg->height <= 0 || cfg->cell_size <= 0 || cfg->nb_samples <= 0 || cfg->nb_samples > 64 ||
        cfg->nb_supersurfels_max <= 0 || cfg->nranks < 1 || cfg->rank < 0 || cfg->rank >= cfg->nranks) {
        g_create_err = "invalid configuration"; return SSF_ERR_INVALID_ARG;
    }
    int ndev = 0;
    if (hipGetDeviceCount(&ndev) != hipSuccess || ndev <= 0) {
        g_create_err = "no HIP device: libssf_hip.so needs a gfx950 GPU (there is no CPU fallback)";
        return SSF_ERR_NO_DEVICE;
    }
    if (cfg->device_id < 0 || cfg->device_id >= ndev) { g_create_err = "device_id out of range"; return SSF_ERR_INVALID_ARG; }
    if (hipSetDevice(cfg->device_id) != hipSuccess) { g_create_err = "hipSetDevice failed"; return SSF_ERR_DEVICE; }
    ssf_handle* h = new (std::nothrow) ssf_handle();
    if (!h) return SSF_ERR_DEVICE;
    h->cfg = *cfg;
    const int W = cfg->width, H = cfg->height, c = cfg->cell_size;
    h->gx = (W + c - 1) / c; h->gy = (H + c - 1) / c; h->S = h->gx * h->gy;
    if (cfg->nb_supersurfels_max < h->S) { delete h; g_create_err = "nb_supersurfels_max < nbSuperpixels"; return SSF_ERR_INVALID_ARG; }
#ifndef SSF_MOVE_TOTALS_DEFAULT
#define SSF_MOVE_TOTALS_DEFAULT 1
#endif
    h->move_totals_on = SSF_ENV_INT("MOVE_TOTALS", SSF_MOVE_TOTALS_DEFAULT) != 0;      // (-DSSF_MOVE_TOTALS_DEFAULT=0: a product build that keeps the fuse launch's tail, for the A/B)
    h->icp_ahead_mode = SSF_ENV_INT("ICP_AHEAD", 1);      // (lab: measurement switches, tools/)
    h->icp_ahead = h->icp_ahead_mode != 0;
    h->ahead_tuner.forced = SSF_ENV_INT("ICP_AHEAD", -1);  // (lab: 0 / 1 / 2 pin the form; the product measures, see AheadTuner)
    h->icp_chain = SSF_ENV_INT("ICP_CHAIN", 1) != 0;
    if (SSF_ENV_SET("NO_GRAPH")) h->graph_failed = true;                             // extract chain launched eagerly
    if (cfg->stream) h->stream = (hipStream_t)cfg->stream;
    else {
        // own track stream: highest priority (ICP -> fuse is the serial chain of the pipeline; its short kernels
        // should not queue behind the wide extract launches of the low-priority context streams)
        int least = 0, greatest = 0;
        (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
        const int prio = SSF_ENV_INT("TRACK_PRIORITY", greatest);
        h->stream = stream_pool().take(cfg->device_id, prio);
        if (!h->stream) { delete h; g_create_err = "hipStreamCreate failed"; return SSF_ERR_DEVICE; }
        h->own_stream = true; h->stream_prio = prio;
    }
    SegParams& p = h->seg;
    p.W = W; p.H = H; p.cell = c; p.gx = h->gx; p.gy = h->gy; p.S = h->S; p.nb_samples = cfg->nb_samples;
    p.min_size = (int)((float)(c * c) / 4.f);                                   // TPS_RGBD.cu:198 (float -> int parameter)
    p.lambda_pos = cfg->lambda_pos; p.lambda_bound = cfg->lambda_bound; p.lambda_size = cfg->lambda_size;
    p.lambda_disp = cfg->lambda_disp; p.thresh_disp = cfg->thresh_disp;
    p.filter_alpha = cfg->filter_alpha; p.filter_beta = cfg->filter_beta; p.filter_threshold = cfg->filter_threshold;
    p.filter_iter = cfg->filter_iter; p.seed = cfg->rng_seed;
    p.inv_gx = 1.0f / (float)h->gx;
    p.cell_magic = c > 1 ? (uint32_t)((0x100000000ull + (uint64_t)c - 1) / (uint64_t)c) : 0u;
    p.win_cells_max = tile_window_cells_max(p);          // (selects the LDS footprint of the tile kernels: ssf_extract.hip, WCAP)
    h->cam.fx = cfg->fx; h->cam.fy = cfg->fy; h->cam.cx = cfg->cx; h->cam.cy = cfg->cy; h->cam.W = W; h->cam.H = H;
    const size_t P = (size_t)W * H, S = h->S, N = cfg->nb_supersurfels_max, NS = S * cfg->nb_samples;
    // relabelling tiles (the shifted grid has one more column): 32-wide tiles with 256 log entries each, or 64-wide
    // ones with 512 (ssf_extract.hip, k_update_pass<., NPX>); the log regions are sized for whichever needs more
    const size_t NT32 = (size_t)((W + 30 + 31) / 32) * ((H + 31) / 32), NT64 = (size_t)((W + 62 + 63) / 64) * ((H + 31) / 32);
    const size_t NT = std::max(NT32, 2 * NT64);
    const int nctx = std::max(0, std::min(cfg->pipeline_depth, SSF_MAX_PIPELINE_DEPTH)) + 1;
    h->batch = std::max(1, std::min(cfg->extract_batch, SSF_MAX_BATCH));
    h->ctx.resize(nctx);
    h->pass_team = SSF_ENV_INT("PASS_TEAM", 0) != 0;              // (lab arm, lab/passes_team.inc: the product never takes it)
    bool ok = dalloc(h, &h->d_srgb_lut, 256) && dalloc(h, &h->d_tickets, 512);
    {   // window geometry of the relabelling tiles (SegParams::pass_geom)
        const int ne = pass_geometry_entries(W, H);
        std::vector<uint2> tab((size_t)ne);
        p.pass_geom = nullptr; p.pass_ntile = ne / 2;
        pass_geometry_table(p, tab.data());
        uint2* d_geom = nullptr;
        ok = ok && dalloc(h, &d_geom, (size_t)ne) && hipMemcpy(d_geom, tab.data(), (size_t)ne * sizeof(uint2), hipMemcpyHostToDevice) == hipSuccess;
        if (ok) p.pass_geom = d_geom;
    }
    // working set of one frame, carved out of a slab (256 B aligned pieces); a context owns `batch` slabs
    auto carve = [&](ExtractCtx& c, char* base) -> size_t {
        size_t off = 0;
        auto take = [&](auto*& ptr, size_t count) {
            using T = typename std::remove_reference<decltype(*ptr)>::type;
            off = (off + 255) & ~(size_t)255;
            ptr = reinterpret_cast<T*>(base + off);
            off += std::max<size_t>(count, 1) * sizeof(T);
        };
        FrameMaps& m = c.maps;
        take(m.rgba, P); take(m.disp, P); take(m.label, P); take(m.inlier, P); take(m.plane_depth, P);
        take(m.sp, S); take(m.samples, NS); take(m.sample_score, NS); take(m.moments, 13 * S); take(m.filt, 11 * S); take(m.epoch, 64); take(m.pix2, P); take(m.fpack, 4 * S);
        for (int b = 0; b < 2; b++) take(m.sums[b].r, S);
        for (int b = 0; b < 3; b++) { take(m.log.ent[b], NT * 256); take(m.log.disp[b], NT * 256); take(m.log.count[b], NT); }
        SurfelSoA& f = c.frame;
        take(f.pos, 3 * S); take(f.col, 3 * S); take(f.lab, 3 * S); take(f.stamps, 2 * S); take(f.r0, 3 * S); take(f.r1, 3 * S);
        take(f.r2, 3 * S); take(f.shape, 6 * S); take(f.dims, 2 * S); take(f.conf, S);
        take(c.d_best, S); take(c.d_matched, S); take(c.d_rgb_in, 3 * P); take(c.d_depth_in, P); take(c.d_depth_filt, P); take(c.d_mask, S);
        take(c.d_wire, 26 * S);
        return (off + 255) & ~(size_t)255;
    };
    size_t slab_bytes = 0;
    for (int ci = 0; ci < nctx && ok; ci++) {
        ExtractCtx& c = h->ctx[ci];
        slab_bytes = carve(c, nullptr);
        char* base = nullptr;
        ok = dalloc(h, &base, slab_bytes * h->batch);
        if (!ok) break;
        (void)carve(c, base);
        FrameMaps& m = c.maps;
        m.slab = slab_bytes;
        m.srgb_lut = h->d_srgb_lut;
        (void)hipMemsetAsync(base, 0, slab_bytes * h->batch, h->stream);
        if (nctx == 1) c.stream = h->stream;                            // sequential: extract shares the track stream
        else {
            // low priority: the track chain (ICP -> fuse, on h->stream) is the critical path, and the
            // runtime keeps a separate pool of hardware queues per priority, so every context gets a
            // queue of its own instead of sharing one with the track stream (head-of-line blocking)
            // (context 0 one level above the others: a sequence that starts on an empty pipeline puts its first, small batch
            // there -- ssf_process_sequence -- and that batch is what the track chain waits for while the larger batches of the
            // other contexts, launched microseconds later, compete for the part.  SSF_CTX0_PRIORITY=0 switches it off.)
            int least = 0, greatest = 0;
            (void)hipDeviceGetStreamPriorityRange(&least, &greatest);
            static const bool ctx0_up = SSF_ENV_INT("CTX0_PRIORITY", 1) != 0;
            const int prio = (ci == 0 && ctx0_up && least - greatest >= 2) ? least - 1 : least;
            c.stream = stream_pool().take(h->cfg.device_id, prio); ok = c.stream != nullptr; c.own_stream = ok; c.stream_prio = prio;
        }
        if (ok && h->pass_team) {
            const int kmax = 4 * std::max(cfg->seg_iter, 0);
            std::vector<char> tab(pass_args_bytes(std::max(kmax, 1)));
            pass_args_table(p, c.maps, kmax, tab.data());
            ok = dalloc(h, &c.d_pas, tab.size()) && dalloc(h, &c.d_team_ws, pass_team_ws_bytes()) &&
                 hipMemcpy(c.d_pas, tab.data(), tab.size(), hipMemcpyHostToDevice) == hipSuccess;
        }
        ok = ok && hipEventCreateWithFlags(&c.ev_done, hipEventDisableTiming) == hipSuccess &&
             hipEventCreateWithFlags(&c.ev_consumed, hipEventDisableTiming) == hipSuccess &&
             hipEventCreate(&c.ev_t0) == hipSuccess && hipEventCreate(&c.ev_t1) == hipSuccess;
    }
    const size_t OC = 3 * N + 4 * S + 1024;        // out-of-view store: home of the span = N + S + 256, room for N rows either side (+ 2 S appended per frame)
    h->part_sup_vis = 6 * (int)(((N + 255) / 256 + 2) / PART_GROUP + 1);
    h->part_sup_oov = (int)(((OC + 255) / 256 + 8) / PART_GROUP + 1);
    h->part_words = h->part_sup_vis + h->part_sup_oov + 8 * PART_REPLICAS;
    ok = ok && dalloc(h, &h->d_part, 2 * (size_t)h->part_words) && dalloc(h, &h->d_part_ticket, 128);
    if (ok) {
        // host-writable device memory for the chained ICP launches; without it (no large BAR) the launches are not chained
        void* q = nullptr;
        int large_bar = 0;
        (void)hipDeviceGetAttribute(&large_bar, hipDeviceAttributeIsLargeBar, cfg->device_id);
        if (large_bar && hipExtMallocWithFlags(&q, SSF_ICP_GO_SLOTS * sizeof(IcpGo), hipDeviceMallocFinegrained) == hipSuccess) {
            h->allocs.push_back(q); h->go = (IcpGo*)q;
            (void)hipMemset(q, 0, SSF_ICP_GO_SLOTS * sizeof(IcpGo));
        } else { (void)hipGetLastError(); h->go = nullptr; }
    }
    ok = ok && alloc_surfels(h, h->model[0], N) && alloc_surfels(h, h->model[1], N) && alloc_surfels(h, h->dense, N) &&
         alloc_surfels(h, h->oov[0].rows, OC) && alloc_surfels(h, h->oov[1].rows, OC) && dalloc(h, &h->oov[0].live, OC) &&
         dalloc(h, &h->oov[1].live, OC) && dalloc(h, &h->d_state_oov, OC) && dalloc(h, &h->d_bc_oov, (OC + 255) / 256 + 8) &&
         dalloc(h, &h->d_live_scratch, N) && dalloc(h, &h->d_bf_in, P) && dalloc(h, &h->d_bf_out, P) &&
         dalloc(h, &h->d_icp, 64) && dalloc(h, &h->d_state, N + 16) && dalloc(h, &h->d_cand, N) &&
         dalloc(h, &h->d_cnt, 2) && dalloc(h, &h->d_migrants, (size_t)SSF_MIGRANT_WORDS * S) && dalloc(h, &h->d_scratch_map, P) && dalloc(h, &h->d_icp_replicas, 2 * SSF_ICP_REPLICAS * 32)     /* second half: the counted record of k_icp */;
#ifdef SSF_EXPERIMENTS
    h->bin_min_rows = SSF_ENV_INT("BIN_MIN_ROWS", SSF_BIN_MIN_ROWS_DEFAULT);          // (lab: the threshold by environment; < 0 never.  The copy's buffers are allocated on first use)
#endif
    if (ok) {
        ok = hipHostMalloc((void**)&h->mb_host, sizeof(Mailbox), hipHostMallocCoherent) == hipSuccess ||
             hipHostMalloc((void**)&h->mb_host, sizeof(Mailbox), hipHostMallocDefault) == hipSuccess;
        if (ok) { std::memset(h->mb_host, 0, sizeof(Mailbox)); ok = hipHostGetDevicePointer((void**)&h->mb_dev, h->mb_host, 0) == hipSuccess; }
    }
    for (int i = 0; i < 4 && ok; i++) ok = hipEventCreate(&h->ev[i]) == hipSuccess;
    if (!ok) { g_create_err = std::string("device allocation failed: ") + hipGetErrorString(hipGetLastError()); ssf_destroy(h); return SSF_ERR_DEVICE; }
    {   // gamma-expansion table for 8-bit colours, built with the same inline function the kernels use
        float lut[256];
        for (int c8 = 0; c8 < 256; c8++) lut[c8] = srgb_expand((float)c8 / 255.0f);
        (void)hipMemcpy(h->d_srgb_lut, lut, sizeof(lut), hipMemcpyHostToDevice);
    }
    (void)hipMemsetAsync(h->d_icp_replicas, 0, 2 * SSF_ICP_REPLICAS * 32 * sizeof(long long), h->stream);
    (void)hipMemsetAsync(h->d_tickets, 0, 512 * sizeof(unsigned int), h->stream);
    (void)hipMemsetAsync(h->d_part, 0, 2 * (size_t)h->part_words * sizeof(uint32_t), h->stream);
    (void)hipMemsetAsync(h->d_part_ticket, 0, 128 * sizeof(uint32_t), h->stream);
    (void)hipMemsetAsync(h->d_cand, 0xFF, N * sizeof(int32_t), h->stream);
    h->oov[0].cap = h->oov[1].cap = (int)OC;
    h->oov_head = h->oov_tail = oov_home(h); h->oov_live = 0;
    {
        Counters c0; std::memset(&c0, 0, sizeof(c0)); c0.oov_head = c0.oov_tail = h->oov_head;
        (void)hipMemcpy(h->d_cnt, &c0, sizeof(c0), hipMemcpyHostToDevice);
    }
    (void)hipMemsetAsync(h->oov[0].live, 0, OC, h->stream); (void)hipMemsetAsync(h->oov[1].live, 0, OC, h->stream);
    zero_surfels(h, h->model[0], N); zero_surfels(h, h->model[1], N); zero_surfels(h, h->dense, N);
    if (hipStreamSynchronize(h->stream) != hipSuccess) { g_create_err = "initialisation failed"; ssf_destroy(h); return SSF_ERR_DEVICE; }
    {   // getters before the first frame see slot 0 of context 0 (zeroed)
        ExtractCtx& c0 = h->ctx[0];
        h->active.maps = c0.maps; h->active.frame = c0.frame; h->active.d_best = c0.d_best; h->active.d_matched = c0.d_matched;
        h->active.ctx = &c0; h->active.slot = 0;
    }
    h->pose.R = m3_identity(); h->pose.t = v3(0, 0, 0);
    *out = h;
    return SSF_OK;
}
const char* ssf_last_error(const ssf_handle* h) { return h ? h->err.c_str() : g_create_err.c_str(); }

int ssf_process_frame(ssf_handle* h, const uint8_t* rgb, const float* depth, const float* prior, const uint8_t* mask, ssf_frame_result* out) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    return process_frame_impl(h, rgb, depth, 0, prior, mask, out);
}
int ssf_process_frame_device(ssf_handle* h, const void* rgb, const void* depth, const float* prior, const uint8_t* mask, ssf_frame_result* out) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    return process_frame_impl(h, rgb, depth, 1, prior, mask, out);
}

// pipelined form: extract of future frames runs ahead on its own stream(s)
int ssf_submit_frame(ssf_handle* h, const void* rgb, const void* depth, int on_device, const uint8_t* mask) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    TimerScope ts(h);
    const double t0 = now_us();
    int rc = submit_extract(h, rgb, depth, on_device, mask);
    h->host_us[0] += now_us() - t0;
    return rc;
}
int ssf_submit_frame_tables(ssf_handle* h, const int32_t* label, const float* plane_depth, const ssf_surfels* frame, int on_device) {
    if (!h || !label || !plane_depth || !frame) return SSF_ERR_INVALID_ARG;
    if (!frame->positions || !frame->colors || !frame->stamps || !frame->orientations || !frame->shapes || !frame->dims || !frame->confidences) return SSF_ERR_INVALID_ARG;
    // (dealt extract counts batches to choose the broadcasting rank; a frame handed in here would not be counted and the ranks'
    //  root choice would drift apart: the two ways of receiving a frame extracted elsewhere do not mix)
    if (h->deal != 0) { h->err = "ssf_submit_frame_tables: the handle deals its extract stage (ssf_comm_deal_extract); submit frames, not tables"; return SSF_ERR_STATE; }
    TimerScope ts(h);
    return submit_tables(h, label, plane_depth, frame, on_device);
}
int ssf_process_submitted(ssf_handle* h, const float* prior, ssf_frame_result* out) {
    if (!h) return SSF_ERR_INVALID_ARG;
    return process_oldest(h, prior, out);
}
int ssf_process_sequence(ssf_handle* h, const void* const* rgb, const void* const* depth, int n, int on_device, ssf_frame_result* out) {
    if (!h || !rgb || !depth || n < 0) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    for (int i = 0; i < n; i++) if (!rgb[i] || !depth[i]) return SSF_ERR_INVALID_ARG;
    int rc = SSF_OK;
    // an empty pipeline: the sequence's first (small) batch goes to context 0, whose stream outranks the other contexts'
    // (ssf_create): the batch the track chain is waiting for is not slowed down by the larger ones launched right behind it
    if (h->ctx.size() > 1 && !h->ctx[h->open_ctx].launched && h->ctx[h->open_ctx].count == 0) {
        bool idle = true;
        for (auto& c : h->ctx) idle = idle && !c.launched && c.count == 0;
        if (idle) h->open_ctx = 0;
    }
    const bool ahead = !on_device && h->ctx.size() > 1 && n > 1;       // host frames, pipelined: copy them ahead
    if (ahead) {
        const size_t P = (size_t)h->cfg.width * h->cfg.height;
        if (!h->up) {
            Uploader* u = new (std::nothrow) Uploader();
            if (!u) { h->err = "out of memory"; return SSF_ERR_DEVICE; }
            // The submitting thread runs up to (contexts + 1) batches ahead of the frame being tracked (every context full + the open
            // batch); a worker may start on frame i when frame i - ring has been processed and needs ~250 us for it (wake-up, 2.1 MB
            // staging memcpy, two enqueues).  With a ring of only two frames more than that window (round 1-4) the submitting thread --
            // the one that drives the track chain -- waited 18 us per frame for uploads and the replay ran at 6100-6300 frames/s on hosts
            // with a slower memcpy (8800-10 000 on faster ones) against 11 400 with frames in HBM: tools/host_buffer_probe.py.  Two
            // more batches of slack.
            // The ring is capped by BYTES as well (advisor, round 4: at depth 3 x 16 frames per launch the formula asks for 98 slots =
            // 0.85 GB of HBM and as much page-locked host memory at 1280x960, held until ssf_destroy): never more than
            // SSF_UPLOAD_RING_BYTES of device buffers (and the same again page-locked), never fewer than the window the submitting
            // thread can run ahead by + 2 (below that the workers could not keep up at all).  INTEGRATION.md section 2b has the footprint.
            const int window = ((int)h->ctx.size() + 1) * h->batch + 2;
            const int by_bytes = (int)(SSF_UPLOAD_RING_BYTES / (7 * P));
            u->ring = std::max(window, std::min(((int)h->ctx.size() + 3) * h->batch + 2, by_bytes));
            u->rgb_bytes = 3 * P; u->depth_bytes = 4 * P;
            u->device = h->cfg.device_id;
            bool ok = true;
            u->d_rgb.assign(u->ring, nullptr); u->d_depth.assign(u->ring, nullptr);
            for (int i = 0; i < u->ring && ok; i++) ok = dalloc(h, &u->d_rgb[i], 3 * P) && dalloc(h, &u->d_depth[i], P);
            if (ok && !SSF_ENV_SET("UPLOAD_PAGEABLE")) {     // page-locked staging (optional: without it the copies go through the runtime's)
                u->p_rgb.assign(u->ring, nullptr); u->p_depth.assign(u->ring, nullptr);
                bool pin = true;
                for (int i = 0; i < u->ring && pin; i++)
                    pin = hipHostMalloc((void**)&u->p_rgb[i], 3 * P, hipHostMallocDefault) == hipSuccess &&
                          hipHostMalloc((void**)&u->p_depth[i], 4 * P, hipHostMallocDefault) == hipSuccess;
                if (!pin) { for (auto q : u->p_rgb) if (q) (void)hipHostFree(q); for (auto q : u->p_depth) if (q) (void)hipHostFree(q); u->p_rgb.clear(); u->p_depth.clear(); (void)hipGetLastError(); }
            }
            for (auto& c : h->ctx) u->ctx_stream.push_back(c.stream);
            u->batch = h->batch;
            if (!ok) { delete u; h->err = "allocation of the upload ring failed"; return SSF_ERR_DEVICE; }    // (buffers taken so far stay in h->allocs)
            h->up = u;
        }
        Uploader& u = *h->up;
        u.n = n; u.rgb = rgb; u.depth = depth; u.ctx0 = h->open_ctx;
        u.processed.store(0); u.failed.store(0); u.stop.store(0);
        u.start();
    }
    h->seq_rgb = rgb; h->seq_depth = depth; h->seq_next = 0; h->seq_n = n; h->seq_on_device = on_device; h->seq_upload = ahead;
    h->seq_batches = 0; h->seq_launches = 0;
    h->seq_t0_us = now_us();
    for (int k = 0; k < n && !rc; k++) {
        while (!rc && h->seq_next < n && !h->ctx[h->open_ctx].launched) {       // fill the pipeline (later refills happen inside do_fuse)
            TimerScope ts(h);
            rc = seq_submit(h);
            // (Measured and removed, round 3: a head start of 150-450 us for the first, small batch before the larger ones join it
            // on the GPU.  Its frame is through at 0.6-0.7 ms instead of 0.83, but the 20-frame sequence takes 3.2-3.5 ms
            // instead of 3.07: the fill is bound by the extract work of the first 14 frames, not by its order -- profiles/fill_r03.txt.)
        }
        h->seq_k = k;
        if (!rc) rc = process_oldest(h, nullptr, out ? &out[k] : nullptr);
        if (k < 64) h->seq_done_us[k] = h->seq_mark_us[3][k] = now_us() - h->seq_t0_us;
        if (ahead) h->up->processed.store(k + 1, std::memory_order_release);
    }
    if (ahead) {
        h->up->stop.store(1);
        h->up->join();
    }
    h->seq_rgb = nullptr; h->seq_depth = nullptr; h->seq_n = 0; h->seq_next = 0; h->seq_upload = false;
    return rc;
}
int ssf_pending_frames(const ssf_handle* h) { return h ? (int)h->pending.size() : 0; }
int ssf_pipeline_capacity(const ssf_handle* h) { return h ? (int)h->ctx.size() * h->batch : 0; }
int ssf_can_submit(const ssf_handle* h) { return (h && !h->ctx[h->open_ctx].launched) ? 1 : 0; }

// ---- loop-closure registration + fern codes -------------------------------------
int ssf_align(ssf_handle* h, const ssf_surfels* src, int n, const float* init_pose, float* rel_pose, int* valid, int* iters,
              int* pairs_last) {
    if (!h || !src || n < 0 || !rel_pose || !valid || !src->positions || !src->colors || !src->orientations) return SSF_ERR_INVALID_ARG;
    TimerScope ts(h);
    // sources: positions, Lab of the colours (same inline function as the kernels), normals = rows[2]
    const size_t N = (size_t)std::max(n, 1);
    std::vector<float> lab(3 * N), nrm(3 * N);
    for (int i = 0; i < n; i++) {
        const V3 l = rgb_to_lab(v3(src->colors[3 * i], src->colors[3 * i + 1], src->colors[3 * i + 2]));
        lab[3 * i] = l.x; lab[3 * i + 1] = l.y; lab[3 * i + 2] = l.z;
        for (int c = 0; c < 3; c++) nrm[3 * i + c] = src->orientations[9 * i + 6 + c];
    }
    float *d_pos = nullptr, *d_lab = nullptr, *d_nrm = nullptr, *d_conf = nullptr; long long* d_out = nullptr;
    DevTemps tmp;
    HCK(tmp.take(&d_pos, 12 * N)); HCK(tmp.take(&d_lab, 12 * N)); HCK(tmp.take(&d_nrm, 12 * N));
    HCK(tmp.take(&d_out, 40 * sizeof(long long)));
    if (src->confidences) HCK(tmp.take(&d_conf, 4 * N));
    hipStream_t st = h->stream;
    int rc = SSF_OK;
    auto cleanup = [&]() {};
    if (n > 0) {
        if (hipMemcpyAsync(d_pos, src->positions, 12 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_lab, lab.data(), 12 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipMemcpyAsync(d_nrm, nrm.data(), 12 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess ||
            (d_conf && hipMemcpyAsync(d_conf, src->confidences, 4 * (size_t)n, hipMemcpyHostToDevice, st) != hipSuccess)) {
            cleanup(); h->err = "upload of the source supersurfels failed"; return SSF_ERR_DEVICE;
        }
    }
    M3 R_init = m3_identity(); V3 t_init = v3(0, 0, 0);
    if (init_pose) { const Rt p0 = pose_from12(init_pose); R_init = p0.R; t_init = p0.t; }
    double tf_inc[16], JtJ[36];
    for (int i = 0; i < 16; i++) tf_inc[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 36; i++) JtJ[i] = 0.0;
    M3 R_inc = m3_identity(); V3 t_inc = v3(0, 0, 0);
    bool ok = true;
    int it = 0, pairs = 0;
    static const int tri[6][6] = {{0, 1, 2, 3, 4, 5}, {1, 6, 7, 8, 9, 10}, {2, 7, 11, 12, 13, 14},
                                  {3, 8, 12, 15, 16, 17}, {4, 9, 13, 16, 18, 19}, {5, 10, 14, 17, 19, 20}};
    while (it < h->cfg.icp_iter) {
        it++;
        inc_to_float(tf_inc, R_inc, t_inc);
        Rt T; T.R = m3_mul(R_inc, R_init); T.t = add(m3_mulv(R_inc, t_init), t_inc);
        launch_align(st, h->cam, d_pos, d_lab, d_nrm, d_conf, n, h->cc->frame, h->cc->maps.label, h->cc->maps.plane_depth, T, d_out);
        long long rec[40];
        if (hipGetLastError() != hipSuccess || hipMemcpyAsync(rec, d_out, 37 * sizeof(long long), hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess) { rc = SSF_ERR_DEVICE; h->err = "align iteration failed on the device"; break; }
        pairs = (int)rec[29];
        if (pairs < 100) { ok = false; break; }
        float cs[3], ct[3], scale;
        for (int i = 0; i < 3; i++) {
            const uint32_t a = (uint32_t)rec[30 + i], b = (uint32_t)rec[33 + i];
            std::memcpy(&cs[i], &a, 4); std::memcpy(&ct[i], &b, 4);
        }
        { const uint32_t a = (uint32_t)rec[36]; std::memcpy(&scale, &a, 4); }
        double Jtr[6];
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) JtJ[i * 6 + j] = (double)rec[tri[i][j]] / SSF_ICP_SCALE_JTJ;
            Jtr[i] = (double)rec[21 + i] / SSF_ICP_SCALE_JTR;
        }
        double tf_iter[16];
        align_increment(JtJ, Jtr, scale, cs, ct, tf_iter);
        mat4_lmul(tf_iter, tf_inc);
    }
    cleanup();
    if (rc) return rc;
    double cov[36];
    mat6_inverse_lu(JtJ, cov);
    for (int i = 0; i < 6; i++) if (cov[i * 6 + i] > h->cfg.icp_cov_thresh) { ok = false; break; }
    Rt rel; rel.R = m3_identity(); rel.t = v3(0, 0, 0);
    if (ok) {
        if (len3(t_inc) > 0.3f) ok = false;                      // stale t_inc (start of the last iteration), :226
        else { rel.R = m3_transpose(R_inc); rel.t = negate(m3_mulv(rel.R, t_inc)); }
    }
    pose_to12(rel, rel_pose);
    *valid = ok ? 1 : 0;
    if (iters) *iters = it;
    if (pairs_last) *pairs_last = pairs;
    if (h->cfg.profile == 1) timer_collect(&h->timer);
    return SSF_OK;
}
int ssf_fern_codes(ssf_handle* h, const uint8_t* rgb, const float* depth, int width, int height, const uint32_t* fern_pos,
                   const uint8_t* fern_rgb, const float* fern_depth, int n, uint8_t* codes) {
    if (!h || !rgb || !depth || !fern_pos || !fern_rgb || !fern_depth || !codes || width <= 0 || height <= 0 || n < 0) return SSF_ERR_INVALID_ARG;
    if (n == 0) return SSF_OK;
    const size_t P = (size_t)width * height;
    uint8_t *d_rgb = nullptr, *d_frgb = nullptr, *d_codes = nullptr; float *d_depth = nullptr, *d_fd = nullptr; uint32_t* d_fp = nullptr;
    DevTemps tmp;
    HCK(tmp.take(&d_rgb, 3 * P)); HCK(tmp.take(&d_depth, 4 * P)); HCK(tmp.take(&d_fp, 8 * (size_t)n));
    HCK(tmp.take(&d_frgb, 3 * (size_t)n)); HCK(tmp.take(&d_fd, 4 * (size_t)n)); HCK(tmp.take(&d_codes, (size_t)n));
    hipStream_t st = h->stream;
    bool ok = hipMemcpyAsync(d_rgb, rgb, 3 * P, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(d_depth, depth, 4 * P, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(d_fp, fern_pos, 8 * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(d_frgb, fern_rgb, 3 * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess &&
              hipMemcpyAsync(d_fd, fern_depth, 4 * (size_t)n, hipMemcpyHostToDevice, st) == hipSuccess;
    if (ok) {
        launch_fern_codes(st, d_rgb, d_depth, width, height, d_fp, d_frgb, d_fd, n, d_codes);
        ok = hipGetLastError() == hipSuccess && hipMemcpyAsync(codes, d_codes, (size_t)n, hipMemcpyDeviceToHost, st) == hipSuccess &&
             hipStreamSynchronize(st) == hipSuccess;
    }
    if (!ok) { h->err = "fern encoding failed on the device"; return SSF_ERR_DEVICE; }
    return SSF_OK;
}

// ---- multi-GPU (native RCCL) ------------------------------------------------------------------------------
int ssf_comm_unique_id(uint8_t* id128) {
    if (!id128) return SSF_ERR_INVALID_ARG;
    RcclApi* api = rccl_api();
    if (!api) { g_create_err = "RCCL is not available in this process"; return SSF_ERR_DEVICE; }
    static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
    ncclUniqueId id;
    if (api->GetUniqueId(&id) != ncclSuccess) { g_create_err = "ncclGetUniqueId failed"; return SSF_ERR_DEVICE; }
    std::memcpy(id128, &id, 128);
    return SSF_OK;
}
int ssf_comm_attach(ssf_handle* h, const uint8_t* id128) {
    if (!h || !id128) return SSF_ERR_INVALID_ARG;
    if (h->comm) { h->err = "a communicator is already attached"; return SSF_ERR_STATE; }
    if (h->cfg.nranks > SSF_MAX_RANKS) { h->err = "too many ranks"; return SSF_ERR_INVALID_ARG; }
    RcclApi* api = rccl_api();
    if (!api) { h->err = "RCCL is not available in this process"; return SSF_ERR_DEVICE; }
    HCK(hipSetDevice(h->cfg.device_id));
    if (!h->d_all5 && !dalloc(h, &h->d_all5, 5 * SSF_MAX_RANKS)) { h->err = "allocation failed"; return SSF_ERR_DEVICE; }
    ncclUniqueId id;
    std::memcpy(&id, id128, 128);
    NCK(api->CommInitRank(&h->comm, h->cfg.nranks, id, h->cfg.rank));
    h->all_valid = false; h->all_pending = false;
    return SSF_OK;
}
// The extract stage dealt over the ranks (see launch_batch): one communicator per batch context, split off the attached one --
// a collective call, made by every rank after ssf_comm_attach and with an empty pipeline.  mode 0: back to the replicated form.
int ssf_comm_deal_extract(ssf_handle* h, int mode) {
    if (!h || mode < 0 || mode > 2) return SSF_ERR_INVALID_ARG;
    if (!h->comm) { h->err = "ssf_comm_deal_extract: attach an RCCL communicator first (the peer-to-peer backend keeps the extract stage replicated)"; return SSF_ERR_STATE; }
    if (!h->pending.empty()) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    for (auto& c : h->ctx) if (c.count > 0 || c.launched) { h->err = "a batch is open in the extract pipeline"; return SSF_ERR_STATE; }
    RcclApi* api = rccl_api();
    if (!api || !api->Broadcast || !api->CommSplit || !api->GroupStart || !api->GroupEnd) { h->err = "this RCCL has no ncclCommSplit / ncclBroadcast"; return SSF_ERR_DEVICE; }
    HCK(hipSetDevice(h->cfg.device_id));
    if (mode != 0)
        for (auto& c : h->ctx)
            if (!c.deal_comm) NCK(api->CommSplit(h->comm, 0, h->cfg.rank, &c.deal_comm, nullptr));
    h->deal = mode; h->deal_batches = 0;
    return SSF_OK;
}
int ssf_comm_info(ssf_handle* h, int* backend, int* ranks, int* my_rank) {
    if (!h) return SSF_ERR_INVALID_ARG;
    int b = 0, n = 1, r = 0;
    if (h->comm) {
        b = 1; n = h->cfg.nranks; r = h->cfg.rank;
        RcclApi* api = rccl_api();
        if (api && api->CommCount && api->CommUserRank) { NCK(api->CommCount(h->comm, &n)); NCK(api->CommUserRank(h->comm, &r)); }
    } else if (h->p2p.on) {
        b = 2; n = (int)h->p2p.opened.size() + 1; r = h->cfg.rank;
        if (h->p2p.opened.empty()) n = h->cfg.nranks;        // ranks of one process (ssf_p2p_attach_local): nothing was opened through IPC
    }
    if (backend) *backend = b;
    if (ranks) *ranks = n;
    if (my_rank) *my_rank = r;
    return SSF_OK;
}
// ---- multi-GPU (native, peer to peer: no collective launches) ---------------------------------------------------
// the bound of every in-kernel wait for a peer, in ticks of the device's constant-rate wall clock (wall_clock64)
static int p2p_set_timeout(ssf_handle* h) {
    int khz = 0;
    if (hipDeviceGetAttribute(&khz, hipDeviceAttributeWallClockRate, h->cfg.device_id) != hipSuccess || khz <= 0) { (void)hipGetLastError(); khz = 100000; }
    h->p2p.view.timeout_ticks = (unsigned long long)(h->p2p.timeout_s * 1000.0 * (double)khz);
    return SSF_OK;
}
static int p2p_region(ssf_handle* h) {
    if (h->p2p.region) return SSF_OK;
    if (h->cfg.nranks > SSF_P2P_MAX_RANKS) { h->err = "the peer-to-peer exchange serves at most 8 ranks (one node)"; return SSF_ERR_INVALID_ARG; }
    HCK(hipSetDevice(h->cfg.device_id));
    const size_t bytes = p2p_region_bytes(h->S);
    void* q = nullptr;
    // Which memory: peers store into this region and this rank's kernels poll it WHILE THEY RUN.  HIP guarantees coherence
    // of ordinary (coarse-grained) device memory across devices only at kernel boundaries -- the owner's L2 may serve its
    // polling loads stale lines while another GPU writes over xGMI -- so the region is FINE-GRAINED device memory
    // (hipDeviceMallocFinegrained) unless the caller has declared, through ssf_p2p_configure, that every rank of the map
    // lives on this handle's device (several shards on one GPU: the one arrangement this build could be run in).  There
    // plain memory is used: all accesses to a region are system-scope atomics that meet in the same memory, and round 2's
    // campaigns (profiles/p2p_campaigns_r02.txt) ran 2600 create-attach-run cycles clean with it against 211 bad ones with
    // an UNCACHED region (hipDeviceMallocUncached; SSF_P2P_REGION_UNCACHED=1 brings that mapping back for experiments).
    static const bool uncached = SSF_ENV_SET("P2P_REGION_UNCACHED");
    if (uncached && hipExtMallocWithFlags(&q, bytes, hipDeviceMallocUncached) == hipSuccess) h->p2p.fine = true;
    else if (!uncached && !h->p2p.same_device && hipExtMallocWithFlags(&q, bytes, hipDeviceMallocFinegrained) == hipSuccess) h->p2p.fine = true;
    else {
        (void)hipGetLastError();
        if (!h->p2p.same_device && !uncached) { h->err = "fine-grained device memory for the exchange region is not available"; return SSF_ERR_DEVICE; }
        HCK(hipMalloc(&q, bytes));
    }
    HCK(hipMemset(q, 0, bytes));
    HCK(hipDeviceSynchronize());
    h->p2p.region = (unsigned char*)q; h->p2p.bytes = bytes;
    return SSF_OK;
}
int ssf_p2p_configure(ssf_handle* h, int all_ranks_on_this_device, double timeout_s) {
    if (!h || !(timeout_s > 0.0)) return SSF_ERR_INVALID_ARG;
    if (h->p2p.region && (all_ranks_on_this_device != 0) != h->p2p.same_device) {
        h->err = "ssf_p2p_configure: the exchange region is already allocated (call before ssf_p2p_export / ssf_p2p_region)"; return SSF_ERR_STATE;
    }
    h->p2p.same_device = all_ranks_on_this_device != 0;
    h->p2p.timeout_s = timeout_s;
    if (h->p2p.on) { int rc = p2p_set_timeout(h); if (rc) return rc; }
    return SSF_OK;
}
int ssf_p2p_region(ssf_handle* h, void** region, size_t* bytes) {
    if (!h || !region) return SSF_ERR_INVALID_ARG;
    int rc = p2p_region(h);
    if (rc) return rc;
    *region = h->p2p.region; if (bytes) *bytes = h->p2p.bytes;
    return SSF_OK;
}
int ssf_p2p_export(ssf_handle* h, uint8_t* handle64) {
    if (!h || !handle64) return SSF_ERR_INVALID_ARG;
    int rc = p2p_region(h);
    if (rc) return rc;
    static_assert(sizeof(hipIpcMemHandle_t) == SSF_P2P_HANDLE_BYTES, "hipIpcMemHandle_t is 64 bytes");
    hipIpcMemHandle_t ih;
    HCK(hipIpcGetMemHandle(&ih, h->p2p.region));
    std::memcpy(handle64, &ih, sizeof(ih));
    return SSF_OK;
}
static int p2p_finish_attach(ssf_handle* h) {
    { int rc = p2p_set_timeout(h); if (rc) return rc; }
    h->p2p.view.me = h->cfg.rank; h->p2p.view.nranks = h->cfg.nranks; h->p2p.view.S = h->S; h->p2p.view.seq = 0;
    h->p2p.on = true;
    h->all_valid = false; h->all_pending = false;
    return SSF_OK;
}
static int p2p_attach_check(ssf_handle* h) {
    if (h->comm || h->p2p.on) { h->err = "an exchange backend is already attached"; return SSF_ERR_STATE; }
    if (h->stamp != 0 && h->cfg.nranks > 1) { /* joining later is fine as long as every rank does so at the same frame */ }
    return p2p_region(h);
}
int ssf_p2p_attach(ssf_handle* h, const uint8_t* handles) {
    if (!h || !handles) return SSF_ERR_INVALID_ARG;
    int rc = p2p_attach_check(h);
    if (rc) return rc;
    for (int r = 0; r < h->cfg.nranks; r++) {
        if (r == h->cfg.rank) { h->p2p.view.peer[r] = h->p2p.region; continue; }
        hipIpcMemHandle_t ih;
        std::memcpy(&ih, handles + (size_t)SSF_P2P_HANDLE_BYTES * r, sizeof(ih));
        void* q = nullptr;
        HCK(hipIpcOpenMemHandle(&q, ih, hipIpcMemLazyEnablePeerAccess));
        h->p2p.opened.push_back(q);
        h->p2p.view.peer[r] = (unsigned char*)q;
    }
    return p2p_finish_attach(h);
}
int ssf_p2p_attach_local(ssf_handle* h, void* const* regions) {
    if (!h || !regions) return SSF_ERR_INVALID_ARG;
    int rc = p2p_attach_check(h);
    if (rc) return rc;
    for (int r = 0; r < h->cfg.nranks; r++) {
        if (r != h->cfg.rank && !regions[r]) { h->err = "a peer region is missing"; return SSF_ERR_INVALID_ARG; }
        h->p2p.view.peer[r] = r == h->cfg.rank ? h->p2p.region : (unsigned char*)regions[r];
    }
    return p2p_finish_attach(h);
}
int ssf_get_global_counts(ssf_handle* h, int64_t* out5) {
    if (!h || !out5) return SSF_ERR_INVALID_ARG;
    if (!h->comm && !h->p2p.on) {
        int rc = hipStreamSynchronize(h->stream) == hipSuccess ? SSF_OK : SSF_ERR_DEVICE;
        Counters c;
        if (rc || hipMemcpy(&c, h->d_cnt, sizeof(c), hipMemcpyDeviceToHost) != hipSuccess) { h->err = "device error"; return SSF_ERR_DEVICE; }
        for (int i = 0; i < 5; i++) out5[i] = c.last[i];
        return SSF_OK;
    }
    int rc = comm_counts(h);
    if (rc) return rc;
    for (int i = 0; i < 5; i++) { out5[i] = 0; for (int r = 0; r < h->cfg.nranks; r++) out5[i] += h->all_cnt[5 * r + i]; }
    return SSF_OK;
}

int ssf_stage_extract(ssf_handle* h, const void* rgb, const void* depth, int on_device, const uint8_t* mask) {
    if (!h || !rgb || !depth) return SSF_ERR_INVALID_ARG;
    TimerScope ts(h);
    return do_extract(h, rgb, depth, on_device, mask);
}
// test hook: compact / recentre the out-of-view store now (normally done when its span runs out of room or holes pile up)
int ssf_debug_recentre(ssf_handle* h) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) return SSF_ERR_STATE;
    return oov_recentre(h);
}
long long ssf_debug_recentre_count(const ssf_handle* h) { return h ? h->n_recentres : -1; }
int ssf_debug_set_max_passes(ssf_handle* h, int n) { if (!h) return SSF_ERR_INVALID_ARG; h->max_passes = n; return SSF_OK; }
// visible rows from which a frame's tracking streams a tile-sorted copy of them (default: never; 0: always)
int ssf_debug_set_bin_min_rows(ssf_handle* h, int n) {
    if (!h) return SSF_ERR_INVALID_ARG;
    h->bin_min_rows = bin_buffer_words(h->cam, 1) == 0 ? -1 : n;
    return SSF_OK;
}
int ssf_stage_set_shard(ssf_handle* h, int64_t off, int64_t gm, int64_t gv) {
    if (!h) return SSF_ERR_INVALID_ARG;
    h->id_offset = off; h->global_n_model = gm; h->global_n_visible = gv; return SSF_OK;
}
int ssf_stage_icp_begin(ssf_handle* h, const float* prior) {
    if (!h || !h->have_frame) return SSF_ERR_STATE;
    icp_begin(h, prior); return SSF_OK;
}
int ssf_stage_icp_accumulate(ssf_handle* h, int64_t* sums) {
    if (!h || !sums) return SSF_ERR_INVALID_ARG;
    TimerScope ts(h);
    int rc = icp_accumulate(h, true);
    if (rc) return rc;
    std::memcpy(sums, h->h_icp, SSF_ICP_RECORD * sizeof(int64_t));
    return SSF_OK;
}
int ssf_stage_icp_update(ssf_handle* h, const int64_t* sums, int* again) {
    if (!h || !sums || !again) return SSF_ERR_INVALID_ARG;
    icp_update(h, sums, again); return SSF_OK;
}
int ssf_stage_icp_end(ssf_handle* h, int* valid) {
    if (!h || !valid) return SSF_ERR_INVALID_ARG;
    icp_end(h, valid); return SSF_OK;
}
int ssf_stage_match(ssf_handle* h, uint64_t* best, uint8_t* matched) {
    if (!h || !best || !matched) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame) return SSF_ERR_STATE;
    TimerScope ts(h);
    int rc = do_match(h);
    if (rc) return rc;
    HCK(hipMemcpyAsync(best, h->cc->d_best, (size_t)h->S * 8, hipMemcpyDeviceToHost, h->stream));
    HCK(hipMemcpyAsync(matched, h->cc->d_matched, (size_t)h->S, hipMemcpyDeviceToHost, h->stream));
    HCK(hipStreamSynchronize(h->stream));
    return SSF_OK;
}
int ssf_stage_begin_submitted(ssf_handle* h) {
    if (!h) return SSF_ERR_INVALID_ARG;
    return activate_oldest(h);
}
int ssf_stage_icp_accumulate_device(ssf_handle* h, int64_t* d_sums) {
    if (!h || !d_sums) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame) return SSF_ERR_STATE;
    TimerScope ts(h);
    return icp_accumulate(h, false, (long long*)d_sums);
}
int ssf_stage_icp_fetch(ssf_handle* h, const int64_t* d_sums, int64_t* sums) {
    if (!h || !d_sums || !sums) return SSF_ERR_INVALID_ARG;
    const unsigned long long seq = ++h->icp_seq;
    launch_publish_icp(h->stream, (const long long*)d_sums, h->mb_dev, seq);
    HCK(hipGetLastError());
    int rc = icp_fetch(h, seq);
    if (rc) return rc;
    std::memcpy(sums, h->h_icp, SSF_ICP_RECORD * sizeof(int64_t));
    return SSF_OK;
}
int ssf_stage_match_device(ssf_handle* h, uint64_t* d_best, uint8_t* d_matched) {
    if (!h || !d_best || !d_matched) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame) return SSF_ERR_STATE;
    TimerScope ts(h);
    int rc = do_match(h);
    if (rc) return rc;
    HCK(hipMemcpyAsync(d_best, h->cc->d_best, (size_t)h->S * 8, hipMemcpyDeviceToDevice, h->stream));
    HCK(hipMemcpyAsync(d_matched, h->cc->d_matched, (size_t)h->S, hipMemcpyDeviceToDevice, h->stream));
    return SSF_OK;
}
int ssf_stage_fuse_device(ssf_handle* h, const uint64_t* d_best, const uint8_t* d_matched, ssf_frame_result* out) {
    if (!h || !d_best || !d_matched) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame || h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    HCK(hipMemcpyAsync(h->cc->d_best, d_best, (size_t)h->S * 8, hipMemcpyDeviceToDevice, h->stream));
    HCK(hipMemcpyAsync(h->cc->d_matched, d_matched, (size_t)h->S, hipMemcpyDeviceToDevice, h->stream));
    return do_fuse(h, out);
}
int ssf_stage_fuse_begin_device(ssf_handle* h, const uint64_t* d_best, const uint8_t* d_matched, int32_t* d_table) {
    if (!h || !d_best || !d_matched || !d_table) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame || h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    HCK(hipMemcpyAsync(h->cc->d_best, d_best, (size_t)h->S * 8, hipMemcpyDeviceToDevice, h->stream));
    HCK(hipMemcpyAsync(h->cc->d_matched, d_matched, (size_t)h->S, hipMemcpyDeviceToDevice, h->stream));
    int rc = fuse_begin(h, 1);
    if (rc) return rc;
    const hipError_t e = h->fuse_migrate ? hipMemcpyAsync(d_table, h->d_migrants, (size_t)SSF_MIGRANT_WORDS * h->S * 4, hipMemcpyDeviceToDevice, h->stream)
                                         : hipMemsetAsync(d_table, 0, (size_t)SSF_MIGRANT_WORDS * h->S * 4, h->stream);
    if (e != hipSuccess) { h->fusing = false; h->err = std::string("migrant table copy: ") + hipGetErrorString(e); return SSF_ERR_DEVICE; }
    return SSF_OK;
}
int ssf_stage_fuse_end_device(ssf_handle* h, const int32_t* d_table, ssf_frame_result* out) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (!h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    return fuse_end(h, d_table, out);
}
int ssf_stage_fuse_begin(ssf_handle* h, const uint64_t* best, const uint8_t* matched, int32_t* table) {
    if (!h || !best || !matched || !table) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame || h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    HCK(hipMemcpyAsync(h->cc->d_best, best, (size_t)h->S * 8, hipMemcpyHostToDevice, h->stream));
    HCK(hipMemcpyAsync(h->cc->d_matched, matched, (size_t)h->S, hipMemcpyHostToDevice, h->stream));
    int rc = fuse_begin(h, 1);
    if (rc) return rc;
    const size_t bytes = (size_t)SSF_MIGRANT_WORDS * h->S * 4;
    if (h->fuse_migrate) {
        hipError_t e = hipMemcpyAsync(table, h->d_migrants, bytes, hipMemcpyDeviceToHost, h->stream);
        if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
        if (e != hipSuccess) { h->fusing = false; h->err = std::string("migrant table copy: ") + hipGetErrorString(e); return SSF_ERR_DEVICE; }
    } else std::memset(table, 0, bytes);
    return SSF_OK;
}
int ssf_stage_fuse_end(ssf_handle* h, const int32_t* table, ssf_frame_result* out) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (!h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    if (table && h->fuse_migrate) HCK(hipMemcpyAsync(h->d_migrants, table, (size_t)SSF_MIGRANT_WORDS * h->S * 4, hipMemcpyHostToDevice, h->stream));
    return fuse_end(h, (table && h->fuse_migrate) ? h->d_migrants : nullptr, out);
}
int ssf_stage_fuse(ssf_handle* h, const uint64_t* best, const uint8_t* matched, ssf_frame_result* out) {
    if (!h || !best || !matched) return SSF_ERR_INVALID_ARG;
    if (!h->have_frame || h->fusing) return SSF_ERR_STATE;
    TimerScope ts(h);
    HCK(hipMemcpyAsync(h->cc->d_best, best, (size_t)h->S * 8, hipMemcpyHostToDevice, h->stream));
    HCK(hipMemcpyAsync(h->cc->d_matched, matched, (size_t)h->S, hipMemcpyHostToDevice, h->stream));
    return do_fuse(h, out);
}

int ssf_get_pose(const ssf_handle* h, float* p) { if (!h || !p) return SSF_ERR_INVALID_ARG; pose_to12(h->pose, p); return SSF_OK; }
int ssf_set_pose(ssf_handle* h, const float* p) { if (!h || !p) return SSF_ERR_INVALID_ARG; h->pose = pose_from12(p); return SSF_OK; }
int ssf_get_counts(const ssf_handle* h, int* nm, int* nv, int* st, int* ns) {
    if (!h) return SSF_ERR_INVALID_ARG;
    if (nm) *nm = h->n_model;
    if (nv) *nv = h->n_visible;
    if (st) *st = h->stamp;
    if (ns) *ns = h->S;
    return SSF_OK;
}

static int copy_out(ssf_handle* h, const SurfelSoA& s, int first, int count, ssf_surfels* o) {
    if (count <= 0) return SSF_OK;
    const size_t n = count, f = first;
    hipStream_t st = h->stream;
    if (o->positions) HCK(hipMemcpyAsync(o->positions, s.pos + 3 * f, 12 * n, hipMemcpyDeviceToHost, st));
    if (o->colors) HCK(hipMemcpyAsync(o->colors, s.col + 3 * f, 12 * n, hipMemcpyDeviceToHost, st));
    if (o->stamps) HCK(hipMemcpyAsync(o->stamps, s.stamps + 2 * f, 8 * n, hipMemcpyDeviceToHost, st));
    if (o->shapes) HCK(hipMemcpyAsync(o->shapes, s.shape + 6 * f, 24 * n, hipMemcpyDeviceToHost, st));
    if (o->dims) HCK(hipMemcpyAsync(o->dims, s.dims + 2 * f, 8 * n, hipMemcpyDeviceToHost, st));
    if (o->confidences) HCK(hipMemcpyAsync(o->confidences, s.conf + f, 4 * n, hipMemcpyDeviceToHost, st));
    std::vector<float> rows;
    if (o->orientations) {
        rows.resize(9 * n);
        HCK(hipMemcpyAsync(rows.data(), s.r0 + 3 * f, 12 * n, hipMemcpyDeviceToHost, st));
        HCK(hipMemcpyAsync(rows.data() + 3 * n, s.r1 + 3 * f, 12 * n, hipMemcpyDeviceToHost, st));
        HCK(hipMemcpyAsync(rows.data() + 6 * n, s.r2 + 3 * f, 12 * n, hipMemcpyDeviceToHost, st));
    }
    HCK(hipStreamSynchronize(st));
    if (o->orientations)
        for (size_t i = 0; i < n; i++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) o->orientations[9 * i + 3 * r + c] = rows[(size_t)r * 3 * n + 3 * i + c];
    return SSF_OK;
}
int ssf_get_model(ssf_handle* h, int first, int count, ssf_surfels* o) {
    if (!h || !o || first < 0 || count < 0 || first + count > h->cfg.nb_supersurfels_max) return SSF_ERR_INVALID_ARG;
    int rc = materialise(h);
    return rc ? rc : copy_out(h, h->dense, first, count, o);
}
int ssf_get_frame(ssf_handle* h, ssf_surfels* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; return copy_out(h, h->cc->frame, 0, h->S, o); }
int ssf_set_model(ssf_handle* h, const ssf_surfels* in, int n, int n_visible, int stamp) {
    if (!h || !in || n < 0 || n > h->cfg.nb_supersurfels_max || n_visible < 0 || n_visible > n) return SSF_ERR_INVALID_ARG;
    if (!in->positions || !in->colors || !in->stamps || !in->orientations || !in->shapes || !in->dims || !in->confidences) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty()) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    drop_shard_sizes(h);
    SurfelSoA& s = h->dense;                       // upload the logical order, then split it into the two stores
    hipStream_t st = h->stream;
    const size_t N = n;
    if (n > 0) {
        std::vector<float> rows(9 * N);
        for (size_t i = 0; i < N; i++)
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) rows[(size_t)r * 3 * N + 3 * i + c] = in->orientations[9 * i + 3 * r + c];
        HCK(hipMemcpyAsync(s.pos, in->positions, 12 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.col, in->colors, 12 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.stamps, in->stamps, 8 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.shape, in->shapes, 24 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.dims, in->dims, 8 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.conf, in->confidences, 4 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.r0, rows.data(), 12 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.r1, rows.data() + 3 * N, 12 * N, hipMemcpyHostToDevice, st));
        HCK(hipMemcpyAsync(s.r2, rows.data() + 6 * N, 12 * N, hipMemcpyHostToDevice, st));
        launch_lab_refresh(st, s, n);
        HCK(hipStreamSynchronize(st));
    }
    h->stamp = stamp;
    { int rc = store_from_dense(h, n, n_visible); if (rc) return rc; }
    return SSF_OK;
}
static int copy_map(ssf_handle* h, void* dst, const void* src, size_t bytes) {
    HCK(hipMemcpyAsync(dst, src, bytes, hipMemcpyDeviceToHost, h->stream));
    HCK(hipStreamSynchronize(h->stream));
    return SSF_OK;
}
int ssf_get_index_map(ssf_handle* h, int32_t* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; return copy_map(h, o, h->cc->maps.label, (size_t)h->cfg.width * h->cfg.height * 4); }
int ssf_get_boundary_map(ssf_handle* h, int32_t* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    launch_boundary_map(h->stream, h->seg, h->cc->maps.label, h->d_scratch_map);
    return copy_map(h, o, h->d_scratch_map, (size_t)h->cfg.width * h->cfg.height * 4);
}
int ssf_get_inlier_map(ssf_handle* h, uint8_t* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; return copy_map(h, o, h->cc->maps.inlier, (size_t)h->cfg.width * h->cfg.height); }
int ssf_get_plane_depth(ssf_handle* h, float* o) { if (!h || !o) return SSF_ERR_INVALID_ARG; return copy_map(h, o, h->cc->maps.plane_depth, (size_t)h->cfg.width * h->cfg.height * 4); }
int ssf_get_superpixels(ssf_handle* h, float* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    std::vector<SpRow> rows(h->S);
    int rc = copy_map(h, rows.data(), h->cc->maps.sp, (size_t)h->S * sizeof(SpRow));
    if (rc) return rc;
    for (int k = 0; k < h->S; k++) {
        const SpRow& r = rows[k];
        const float v[9] = {r.cx, r.cy, r.r, r.g, r.b, r.ta, r.tb, r.tc, r.size};
        std::memcpy(&o[9 * k], v, sizeof(v));
    }
    return SSF_OK;
}
int ssf_get_model_device(ssf_handle* h, ssf_surfels* o, int* n) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    if (!h->d_orient9 && !dalloc(h, &h->d_orient9, 9 * (size_t)h->cfg.nb_supersurfels_max)) { h->err = "allocation failed"; return SSF_ERR_DEVICE; }
    { int rc = materialise(h); if (rc) return rc; }
    const SurfelSoA& s = h->dense;                 // a dense copy: [visible | out-of-view], valid until the next call
    launch_pack_orient(h->stream, s, h->n_model, h->d_orient9);
    HCK(hipGetLastError());
    HCK(hipStreamSynchronize(h->stream));
    o->positions = s.pos; o->colors = s.col; o->stamps = s.stamps; o->orientations = h->d_orient9; o->shapes = s.shape;
    o->dims = s.dims; o->confidences = s.conf;
    if (n) *n = h->n_model;
    return SSF_OK;
}
int ssf_get_frame_device(ssf_handle* h, ssf_surfels* o, int* n) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    if (!h->cc || !h->cc->frame.pos) { h->err = "no frame has been processed yet"; return SSF_ERR_STATE; }
    if (!h->d_frame_orient9 && !dalloc(h, &h->d_frame_orient9, 9 * (size_t)h->S)) { h->err = "allocation failed"; return SSF_ERR_DEVICE; }
    const SurfelSoA& s = h->cc->frame;
    launch_pack_orient(h->stream, s, h->S, h->d_frame_orient9);
    HCK(hipGetLastError());
    HCK(hipStreamSynchronize(h->stream));
    o->positions = s.pos; o->colors = s.col; o->stamps = s.stamps; o->orientations = h->d_frame_orient9; o->shapes = s.shape;
    o->dims = s.dims; o->confidences = s.conf;
    if (n) *n = h->S;
    return SSF_OK;
}
int ssf_get_preview_image(ssf_handle* h, uint8_t* o) {
    if (!h || !o) return SSF_ERR_INVALID_ARG;
    const size_t P = (size_t)h->cfg.width * h->cfg.height;
    uint8_t* d = reinterpret_cast<uint8_t*>(h->d_scratch_map);      // P x int32 of scratch: 3P bytes fit
    launch_preview(h->stream, h->cfg.width, h->cfg.height, h->cc->maps.label, h->cc->maps.rgba, d);
    HCK(hipGetLastError());
    return copy_map(h, o, d, 3 * P);
}

// exportModel, supersurfel_fusion.cu:595-633 (std::to_string == "%f"/"%d")
int ssf_export_model_txt(ssf_handle* h, const char* path) {
    if (!h || !path) return SSF_ERR_INVALID_ARG;
    const int n = h->n_model;
    std::vector<float> pos(3 * (size_t)n), col(3 * (size_t)n), ori(9 * (size_t)n), shp(6 * (size_t)n), dims(2 * (size_t)n), conf(n);
    std::vector<int32_t> stamps(2 * (size_t)n);
    ssf_surfels o = {pos.data(), col.data(), stamps.data(), ori.data(), shp.data(), dims.data(), conf.data()};
    int rc = materialise(h);
    if (!rc) rc = copy_out(h, h->dense, 0, n, &o);
    if (rc) return rc;
    FILE* f = std::fopen(path, "w");
    if (!f) { h->err = "cannot open file"; return SSF_ERR_IO; }
    for (int i = 0; i < n; i++) {
        if (!(conf[i] > h->cfg.conf_thresh)) continue;
        std::fprintf(f, "%d %d %f\n", stamps[2 * i], stamps[2 * i + 1], conf[i]);
        std::fprintf(f, "%f %f %f\n", pos[3 * i], pos[3 * i + 1], pos[3 * i + 2]);
        std::fprintf(f, "%f %f %f\n", col[3 * i], col[3 * i + 1], col[3 * i + 2]);
        std::fprintf(f, "%f %f\n", dims[2 * i], dims[2 * i + 1]);
        const float* q = &ori[9 * (size_t)i];
        std::fprintf(f, "%f %f %f %f %f %f %f %f %f\n", q[0], q[1], q[2], q[3], q[4], q[5], q[6], q[7], q[8]);
        const float* c = &shp[6 * (size_t)i];
        std::fprintf(f, "%f %f %f %f %f %f\n\n", c[0], c[1], c[2], c[3], c[4], c[5]);
    }
    std::fclose(f);
    return SSF_OK;
}

int ssf_apply_deformation(ssf_handle* h, const float* np, const float* nr, const float* nt, int m, const float* w4, const int32_t* idx4) {
    if (!h || !np || !nr || !nt || !w4 || !idx4 || m <= 0) return SSF_ERR_INVALID_ARG;
    drop_shard_sizes(h);
    h->ahead.valid = false;
    const size_t n = h->n_model;
    if (n == 0) return SSF_OK;
    float *d_np, *d_nr, *d_nt, *d_w, *d_nodes; int32_t* d_i;
    DevTemps tmp;
    HCK(tmp.take(&d_np, 12 * (size_t)m)); HCK(tmp.take(&d_nr, 36 * (size_t)m)); HCK(tmp.take(&d_nt, 12 * (size_t)m));
    HCK(tmp.take(&d_nodes, 64 * (size_t)m));
    HCK(tmp.take(&d_w, 16 * n)); HCK(tmp.take(&d_i, 16 * n));
    hipStream_t st = h->stream;
    HCK(hipMemcpyAsync(d_np, np, 12 * (size_t)m, hipMemcpyHostToDevice, st));
    HCK(hipMemcpyAsync(d_nr, nr, 36 * (size_t)m, hipMemcpyHostToDevice, st));
    HCK(hipMemcpyAsync(d_nt, nt, 12 * (size_t)m, hipMemcpyHostToDevice, st));
    HCK(hipMemcpyAsync(d_w, w4, 16 * n, hipMemcpyHostToDevice, st));
    HCK(hipMemcpyAsync(d_i, idx4, 16 * n, hipMemcpyHostToDevice, st));
    // applied to the dense logical view (weights are per logical row), then split back into the two stores
    { int rc = materialise(h); if (rc) return rc; }
    { TimerScope ts(h); launch_deformation(st, h->dense, (int)n, m, d_np, d_nr, d_nt, d_nodes, d_w, d_i); }
    { int rc = store_from_dense(h, h->n_model, h->n_visible); if (rc) return rc; }
    HCK(hipStreamSynchronize(st));
    if (h->cfg.profile == 1) timer_collect(&h->timer);
    return SSF_OK;
}

// ---- re-homing of a sharded map (see ssf.h): rows moved by ssf_apply_deformation go to the rank that owns their tile ----
// A rare, bulk operation (a loop closure): worked on the dense logical view with full-model copies; the transport between
// the ranks is the caller's (supersurfel_fusion_amd/sharded.py: torch.distributed; the tests: files / memory).
int ssf_rehome_begin(ssf_handle* h, int32_t* table, int table_rows, int* n_out) {
    if (!h || !n_out || table_rows < 0 || (!table && table_rows > 0)) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty() || h->fusing) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    *n_out = 0;
    drop_shard_sizes(h);
    const int n = h->n_model;
    if (h->cfg.nranks <= 1 || n == 0) return SSF_OK;
    hipStream_t st = h->stream;
    { int rc = materialise(h); if (rc) return rc; }
    int32_t* d_table = nullptr; int* d_tot = nullptr;
    DevTemps tmp;
    HCK(tmp.take(&d_table, (size_t)std::max(table_rows, 1) * SSF_MIGRANT_WORDS * 4)); HCK(tmp.take(&d_tot, 16));
    SurfelSoA scratch = h->oov[h->ocur ^ 1].rows;          // (the other out-of-view store is scratch between recentres)
    launch_rehome_split(st, h->dense, n, h->n_visible, h->cfg.rank, h->cfg.nranks, h->cfg.shard_tile, h->d_bc_oov, d_tot, scratch, d_table, table_rows);
    HCK(hipGetLastError());
    int tot[3] = {0, 0, 0};
    HCK(hipMemcpyAsync(tot, d_tot, sizeof(tot), hipMemcpyDeviceToHost, st));
    HCK(hipStreamSynchronize(st));
    if (tot[1] > table_rows) { h->err = "ssf_rehome_begin: the table is too small for the rows that leave"; return SSF_ERR_CAPACITY; }   // (stores untouched)
    if (tot[1] == 0) return SSF_OK;
    HCK(hipMemcpyAsync(table, d_table, (size_t)tot[1] * SSF_MIGRANT_WORDS * 4, hipMemcpyDeviceToHost, st));
    { int rc = copy_soa(h, h->dense, scratch, (size_t)tot[0]); if (rc) return rc; }
    { int rc = store_from_dense(h, tot[0], tot[2]); if (rc) return rc; }
    HCK(hipStreamSynchronize(st));
    *n_out = tot[1];
    return SSF_OK;
}
int ssf_rehome_end(ssf_handle* h, const int32_t* table, int n_rec) {
    if (!h || n_rec < 0 || (!table && n_rec > 0)) return SSF_ERR_INVALID_ARG;
    if (!h->pending.empty() || h->fusing) { h->err = "frames are pending in the extract pipeline"; return SSF_ERR_STATE; }
    drop_shard_sizes(h);
    // the records addressed to this rank, split by the block they arrive in (table order kept).  A full shard turns the
    // surplus away, in table order, as a frame's migration does (k_migrate_in): their source shards have already let them
    // go, so they are lost to the map -- the call still succeeds on every rank (an error here would leave the ranks in
    // different states with nothing to roll back) and returns their number
    std::vector<int32_t> vis, oov;
    const int n = h->n_model, nv = h->n_visible;
    int room = h->cfg.nb_supersurfels_max - n, turned_away = 0;
    for (int j = 0; j < n_rec; j++) {
        const int32_t* w = table + (size_t)SSF_MIGRANT_WORDS * j;
        if (w[0] - 1 != h->cfg.rank) continue;
        if (room <= 0) { turned_away++; continue; }
        room--;
        std::vector<int32_t>& dst = w[1] ? vis : oov;
        dst.insert(dst.end(), w, w + SSF_MIGRANT_WORDS);
    }
    const int av = (int)(vis.size() / SSF_MIGRANT_WORDS), ao = (int)(oov.size() / SSF_MIGRANT_WORDS);
    if (av + ao == 0) return turned_away;
    hipStream_t st = h->stream;
    { int rc = materialise(h); if (rc) return rc; }
    int32_t* d_rec = nullptr;
    DevTemps tmp;
    HCK(tmp.take(&d_rec, (size_t)(av + ao) * SSF_MIGRANT_WORDS * 4));
    if (av) HCK(hipMemcpyAsync(d_rec, vis.data(), vis.size() * 4, hipMemcpyHostToDevice, st));
    if (ao) HCK(hipMemcpyAsync(d_rec + vis.size(), oov.data(), oov.size() * 4, hipMemcpyHostToDevice, st));
    // [visible | arrivals flagged visible | out of view | the other arrivals], assembled in the scratch store
    SurfelSoA scratch = h->oov[h->ocur ^ 1].rows;
    { int rc = copy_soa(h, scratch, h->dense, (size_t)nv); if (rc) return rc; }
    launch_rehome_unpack(st, d_rec, av, scratch, nv);
    { int rc = copy_soa(h, soa_rows(scratch, (size_t)nv + av), soa_rows(h->dense, (size_t)nv), (size_t)(n - nv)); if (rc) return rc; }
    launch_rehome_unpack(st, d_rec + vis.size(), ao, scratch, n + av);
    HCK(hipGetLastError());
    { int rc = copy_soa(h, h->dense, scratch, (size_t)n + av + ao); if (rc) return rc; }
    { int rc = store_from_dense(h, n + av + ao, nv + av); if (rc) return rc; }
    HCK(hipStreamSynchronize(st));
    return turned_away;
}

int ssf_bilateral_filter(ssf_handle* h, const void* in, void* out, int on_device) {
    if (!h || !in || !out) return SSF_ERR_INVALID_ARG;
    const size_t P = (size_t)h->cfg.width * h->cfg.height;
    const float* d_in = (const float*)in; float* d_out = (float*)out;
    if (!on_device) { HCK(hipMemcpyAsync(h->d_bf_in, in, 4 * P, hipMemcpyHostToDevice, h->stream)); d_in = h->d_bf_in; d_out = h->d_bf_out; }
    { TimerScope ts(h); launch_bilateral(h->stream, d_in, d_out, h->cfg.width, h->cfg.height, h->cfg.prefilter_sigma_color, h->cfg.prefilter_sigma_space); }
    if (!on_device) HCK(hipMemcpyAsync(out, d_out, 4 * P, hipMemcpyDeviceToHost, h->stream));
    HCK(hipStreamSynchronize(h->stream));
    if (h->cfg.profile == 1) timer_collect(&h->timer);
    return SSF_OK;
}
int ssf_get_kernel_times(ssf_handle* h, const char** names, double* ms, int64_t* calls, int max_k) {
    if (!h || !names || !ms || !calls) return 0;
    h->timer_names.clear();
    for (auto& kv : h->timer.acc) h->timer_names.push_back(kv.first);
    int k = 0;
    for (auto& nm : h->timer_names) {
        if (k >= max_k) break;
        names[k] = nm.c_str(); ms[k] = h->timer.acc[nm].first; calls[k] = h->timer.acc[nm].second; k++;
    }
    return k;
}
int ssf_reset_kernel_times(ssf_handle* h) { if (!h) return SSF_ERR_INVALID_ARG; h->timer.acc.clear(); return SSF_OK; }
int ssf_set_profile(ssf_handle* h, int enable) {
    if (!h) return SSF_ERR_INVALID_ARG;
    h->cfg.profile = enable;
    if (enable == 1) timer_calibrate(&h->timer, h->stream);
    return SSF_OK;
}

// completion times (us since the call started) of the first 64 frames of the last ssf_process_sequence (tools/startup_probe.py)
#ifdef SSF_EXPERIMENTS          // (laboratory build only: probes of tools/, not part of the product)
// the record of the last ICP iteration the host fetched (after the exchange of a sharded map: the SUM over the ranks)
int ssf_dbg_last_icp_record(ssf_handle* h, int64_t* out29) {
    if (!h || !out29) return SSF_ERR_INVALID_ARG;
    for (int i = 0; i < 29; i++) out29[i] = h->h_icp_local[i];
    return SSF_OK;
}
// device copy of the last record [0..28] and, with the peer-to-peer exchange, of this shard's own record before it [32..60]
int ssf_dbg_device_icp_records(ssf_handle* h, int64_t* out64) {
    if (!h || !out64) return SSF_ERR_INVALID_ARG;
    HCK(hipStreamSynchronize(h->stream));
    HCK(hipMemcpy(out64, h->d_icp, 64 * sizeof(int64_t), hipMemcpyDeviceToHost));
    return SSF_OK;
}
#endif
// per frame of the last sequence (first 64): entry of the track loop, first ICP record back, ICP loop done, counters
// back [us from the call's entry]; then 32 x (time, frames) of the extract batches launched
int ssf_sequence_marks(ssf_handle* h, double* out320) {
    if (!h || !out320) return SSF_ERR_INVALID_ARG;
    for (int m = 0; m < 4; m++) for (int i = 0; i < 64; i++) out320[m * 64 + i] = h->seq_mark_us[m][i];
    for (int i = 0; i < 32; i++) { out320[256 + 2 * i] = i < h->seq_launches ? h->seq_launch_us[i] : -1.0; out320[257 + 2 * i] = i < h->seq_launches ? h->seq_launch_n[i] + h->seq_launch_host_us[i] / 1e4 : 0; }
    return SSF_OK;
}
// frames whose association ran inside a waiting ICP launch (SSF_ICP_GO_MATCH) since the handle was created
long long ssf_waiter_matches(ssf_handle* h) { return h ? h->n_waiter_matches : -1; }
// the self-tuned choice of AheadTuner: [0] form in effect (1: first ICP iteration inside the row-move kernel), [1] pipelined frames
// seen, [2] / [3] mean chain period per unit of work measured in the last probe without / with the fusion (0: not probed yet)
int ssf_tuner_state(ssf_handle* h, double* out4) {
    if (!h || !out4) return SSF_ERR_INVALID_ARG;
    const AheadTuner& t = h->ahead_tuner;
    out4[0] = t.current(); out4[1] = t.frames; out4[2] = t.n[0] ? t.sum[0] / t.n[0] : 0.0; out4[3] = t.n[1] ? t.sum[1] / t.n[1] : 0.0;
    return SSF_OK;
}
// ... and the frames whose association was run again as a launch of its own because the host's word to the waiting launch came
// too late to be trusted; the test hook that makes it late (a stall of the calling thread in front of the word)
long long ssf_waiter_match_repairs(ssf_handle* h) { return h ? h->n_waiter_match_repairs : -1; }
#ifdef SSF_EXPERIMENTS
void ssf_dbg_stall_before_match_us(ssf_handle* h, long long us) { if (h) h->dbg_stall_before_match_us = us; }      // (fault injection: lab build only)
#endif
int ssf_sequence_times(ssf_handle* h, double* out64) {
    if (!h || !out64) return SSF_ERR_INVALID_ARG;
    for (int i = 0; i < 64; i++) out64[i] = h->seq_done_us[i];
    return SSF_OK;
}
#ifdef SSF_EXPERIMENTS          // (laboratory build only: probes of tools/, not part of the product)
// host-side time split of the pipelined loop (tools/pipeline_probe.py); reset on read
int ssf_dbg_host_times(ssf_handle* h, double* out8) {
    if (!h || !out8) return SSF_ERR_INVALID_ARG;
    for (int i = 0; i < 8; i++) { out8[i] = h->host_us[i]; h->host_us[i] = 0; }
    return SSF_OK;
}

// throughput of the extract stage alone (tools/extract_only_probe.py): frames (device pointers, `nlist` of them,
// cycled) go through the batch contexts and are retired unread; returns microseconds per frame.  The handle's
// frame stamp advances as if the frames had been fused.
double ssf_dbg_extract_only(ssf_handle* h, const void* const* rgb, const void* const* depth, int nlist, int n) {
    if (!h || !rgb || !depth || nlist <= 0 || !h->pending.empty()) return -1.0;
    int nsub = 0;
    double t0 = 0;
    for (int i = 0; i < n; i++) {
        if (i == n / 4) { for (auto& c : h->ctx) (void)hipStreamSynchronize(c.stream); (void)hipStreamSynchronize(h->stream); t0 = now_us(); }
        while (nsub < n && !h->ctx[h->open_ctx].launched) {
            if (submit_extract(h, rgb[nsub % nlist], depth[nsub % nlist], 1, nullptr)) return -1.0;
            nsub++;
        }
        if (activate_oldest(h) || retire_active(h)) return -1.0;
        h->stamp++;
    }
    for (auto& c : h->ctx) (void)hipStreamSynchronize(c.stream);
    (void)hipStreamSynchronize(h->stream);
    return (now_us() - t0) / (double)(n - n / 4);
}

#endif

// What a stream copy reaches on THIS box (nominal AND measured-achievable peak): 16 bytes per lane, `mib` MiB
// read + the same written, best of `reps` over four forms of the same copy -- MI355X_MICROARCH.md quotes 6.29 TB/s (79 % of the
// 8 TB/s spec) for a float4 copy; round 4's single grid-stride form reached 4.7-4.8 TB/s on these boxes and torch's own copy kernel
// 5.2, neither tuned.  Forms: U independent 16-byte loads per lane in flight before the first store (4 or 8), plain or
// non-temporal (`nt`: streamed data is not kept in L2 / MALL, which a copy of 2 GiB only thrashes), grid-stride over 8192
// workgroups or one pass of exactly-sized workgroups.  The best form's rate is returned.
double ssf_stream_copy_rate(int mib, int reps) {
    // Every form moves WHOLE rounds: the grid-stride forms cover U x 2^21 threads x 16 B = 128 MiB (U = 4) or 256 MiB (U = 8) per
    // round of their loop and their first round is unguarded (k_stream_copy), so the buffers are a multiple of 256 MiB and sizes
    // below that are refused -- a smaller buffer would be overrun, a ragged one credited with bytes it did not move.
    if (mib < 256 || reps < 1) return -1.0;
    const size_t bytes = ((size_t)mib << 20) & ~(size_t)((1u << 28) - 1u), n = bytes / sizeof(f4v);
    if (bytes == 0) return -1.0;
    f4v *a = nullptr, *b = nullptr;
    if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) { (void)hipGetLastError(); (void)hipFree(a); return -1.0; }
    (void)hipMemset(a, 1, bytes); (void)hipMemset(b, 0, bytes);
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    double best = 0.0;
    for (int form = 0; form < 6; form++) {
        for (int r = 0; r < reps + 2; r++) {
            (void)hipEventRecord(e0, 0);
            const dim3 grid_stride(256 * 32), blk(256);
            switch (form) {
                case 0: hipLaunchKernelGGL((k_stream_copy<4, false, false>), grid_stride, blk, 0, 0, a, b, n); break;
                case 1: hipLaunchKernelGGL((k_stream_copy<4, true, false>), grid_stride, blk, 0, 0, a, b, n); break;
                case 2: hipLaunchKernelGGL((k_stream_copy<8, true, false>), grid_stride, blk, 0, 0, a, b, n); break;
                case 3: hipLaunchKernelGGL((k_stream_copy<4, true, true>), dim3((unsigned int)(n / (256 * 4))), blk, 0, 0, a, b, n); break;
                case 4: hipLaunchKernelGGL((k_stream_copy<8, true, true>), dim3((unsigned int)(n / (256 * 8))), blk, 0, 0, a, b, n); break;
                default: hipLaunchKernelGGL((k_stream_copy<8, false, true>), dim3((unsigned int)(n / (256 * 8))), blk, 0, 0, a, b, n); break;
            }
            (void)hipEventRecord(e1, 0);
            (void)hipEventSynchronize(e1);
            float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
            if (r >= 2 && ms > 0.f) best = std::max(best, 2.0 * (double)bytes / (ms * 1e-3) / 1e9);
        }
    }
    (void)hipGetLastError();
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1); (void)hipFree(a); (void)hipFree(b);
    return best;
}

#ifdef SSF_EXPERIMENTS          // (laboratory build only: probes of tools/, not part of the product)
// ablation timer for the ICP kernel (tools/icp_probe.py): `reps` back-to-back launches in mode `dbg`
// (bit0: skip the per-surfel math, bit1: skip the LDS accumulation, bit2: skip arrival counting + tail)
double ssf_dbg_time_icp(ssf_handle* h, int reps, int dbg) {
    if (!h || !h->have_frame) return -1.0;
    Rt T; T.R = m3_transpose(h->pose.R); T.t = negate(m3_mulv(T.R, h->pose.t));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3; i++) launch_icp(h->stream, h->cam, h->model[h->mcur], h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, T, h->d_icp_replicas, h->d_tickets + 8, h->d_icp, h->mb_dev, ++h->icp_seq, dbg);
    (void)hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps; i++) launch_icp(h->stream, h->cam, h->model[h->mcur], h->n_visible, h->cc->maps.pix2, h->cc->maps.fpack, T, h->d_icp_replicas, h->d_tickets + 8, h->d_icp, h->mb_dev, ++h->icp_seq, dbg);
    (void)hipEventRecord(e1, h->stream);
    (void)hipStreamSynchronize(h->stream);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipMemsetAsync(h->d_icp_replicas, 0, 2 * SSF_ICP_REPLICAS * 32 * sizeof(long long), h->stream);
    (void)hipMemsetAsync(h->d_tickets, 0, 512 * sizeof(unsigned int), h->stream);
    (void)hipStreamSynchronize(h->stream);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 1000.0 * ms / reps;
}

// ablation timer for the fuse launch (tools/fuse_probe.py; lab build; leaves model and partition sums garbage): `reps` back-to-back
// k_update_insert launches on the current frame.  mode bit 0: without the out-of-view arm, bit 1: without update + insert,
// bit 2: without the classification of the visible rows (and then without the update, which needs them)
double ssf_dbg_time_fuse(ssf_handle* h, int reps, int mode, long long* blocks2 /* out-of-view blocks of 256 slots | those with rows that move; may be null */) {
    if (!h || !h->have_frame || !h->cc) return -1.0;
    SurfelSoA& M = h->model[h->mcur];
    PartitionWs ws;
    uint32_t* set = h->d_part + (size_t)h->part_set * h->part_words;
    ws.sup_vis = set; ws.sup_oov = set + h->part_sup_vis; ws.tot = ws.sup_oov + h->part_sup_oov;
    ws.ticket = h->d_part_ticket; ws.other = h->d_part + (size_t)(h->part_set ^ 1) * h->part_words; ws.words = h->part_words;
    const int S = (mode & 2) ? 0 : h->S, nvis = (mode & 4) ? 0 : h->n_visible, span = (mode & 1) ? 0 : h->oov_tail - h->oov_head;
    auto launch = [&] {
        launch_fuse(h->stream, M, h->cc->frame, h->pose, h->stamp, h->id_offset, nvis, h->cc->d_best, h->cc->d_matched, h->d_cand,
                    S, (nvis > 0 && S > 0) ? 1 : 0, h->cfg.nb_supersurfels_max, 0, 1, h->cfg.shard_tile, h->d_cnt,
                    h->cam, h->oov[h->ocur], span, h->cc->maps.plane_depth, h->cfg.delta_t,
                    h->cfg.conf_thresh, h->cfg.range_min, h->cfg.range_max, h->d_state, h->d_state_oov, h->d_bc_oov, ws, 0, 1);
    };
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    for (int i = 0; i < 3 && reps > 0; i++) launch();              // (reps <= 0: only the block census of the last real frame)
    (void)hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps; i++) launch();
    (void)hipEventRecord(e1, h->stream);
    (void)hipStreamSynchronize(h->stream);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    if (blocks2) {
        const int nb = (span + 255) / 256;
        std::vector<uint32_t> bc((size_t)std::max(nb, 1));
        if (nb > 0) (void)hipMemcpy(bc.data(), h->d_bc_oov, (size_t)nb * 4, hipMemcpyDeviceToHost);
        blocks2[0] = nb; blocks2[1] = 0;
        for (int i = 0; i < nb; i++) blocks2[1] += bc[i] != 0u;
    }
    return reps > 0 ? 1000.0 * ms / reps : 0.0;
}
// one fuse launch on the current frame with every workgroup leaving its three ticks (g_fuse_trace in ssf_track_fuse.hip): out =
// 3 x workgroups words, arms4 = workgroups of update | insertion | visible rows | out-of-view span.  Returns the workgroups, < 0: n/a.
int ssf_dbg_trace_fuse(ssf_handle* h, unsigned long long* out, int cap_wgs, int* arms4, int mode /* as ssf_dbg_time_fuse */) {
    if (!h || !h->have_frame || !h->cc || !out || !arms4) return -1;
    arms4[0] = (mode & 2) ? 0 : (h->S + 31) / 32; arms4[1] = (mode & 2) ? 0 : (h->S + 255) / 256; arms4[2] = (mode & 4) ? 0 : (h->n_visible + 255) / 256;
    long long b2[2];
    unsigned long long* d = nullptr;
    const size_t words = (size_t)3 * 65536;
    if (hipMalloc((void**)&d, words * 8) != hipSuccess) return -2;
    (void)hipMemset(d, 0, words * 8);
    set_fuse_trace(d);
    (void)ssf_dbg_time_fuse(h, 1, mode, b2);         // (3 warm launches with the trace on, then the one whose ticks stay)
    set_fuse_trace(nullptr);
    std::vector<unsigned long long> all(words);
    (void)hipMemcpy(all.data(), d, words * 8, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    int n = 0;
    for (int i = 0; i < 65536; i++) if (all[3 * (size_t)i]) n = i + 1;
    arms4[3] = n - arms4[0] - arms4[1] - arms4[2];
    const int m = n < cap_wgs ? n : cap_wgs;
    std::memcpy(out, all.data(), (size_t)m * 24);
    return m;
}
#endif
// ablation timer for the relabelling pass (tools/pass_probe.py); leaves the segmentation state garbage
// lab build: tiles of the last extracted frame (slot 0 of the active context) that proved themselves clean, per pass
// (FrameMaps::epoch[1 + pass], counted by k_update_pass under SSF_EXPERIMENTS; tools/skip_probe.py)
// streams waiting in the process-wide pool for the next handle (StreamPool)
int ssf_pooled_streams(void) {
    StreamPool& sp = stream_pool();
    std::lock_guard<std::mutex> lk(sp.mu);
    int n = 0;
    for (auto& kv : sp.idle) n += (int)kv.second.size();
    return n;
}
// host frames of sequences: [0] workers, [1] frames uploaded, microseconds summed over the workers [2] waiting for a ring slot,
// [3] in the staging memcpy, [4] in the two hipMemcpyAsync calls, [5] the submitting thread's wait for uploads (tools/host_buffer_probe.py)
int ssf_upload_stats(ssf_handle* h, double* out6) {
    if (!h || !out6) return SSF_ERR_INVALID_ARG;
    for (int i = 0; i < 6; i++) out6[i] = 0.0;
    if (h->up) {
        out6[0] = Uploader::NTH; out6[1] = (double)h->up->frames_done.load(); out6[2] = (double)h->up->us_ring.load();
        out6[3] = (double)h->up->us_memcpy.load(); out6[4] = (double)h->up->us_enqueue.load();
    }
    out6[5] = h->us_wait_upload;
    return SSF_OK;
}
#ifdef SSF_EXPERIMENTS          // (laboratory build only: probes of tools/, not part of the product)
// the relabelling statistics of the frame just processed (FrameMaps::epoch, SSF_PASS_STAT_* in ssf_extract.hip): out64[8 .. 12];
// collected only after ssf_dbg_pass_stats_enable(1)
int ssf_dbg_pass_stats_enable(int on) { set_pass_stats(on ? 1 : 0); return SSF_OK; }
int ssf_dbg_pass_stats(ssf_handle* h, uint32_t* out64) {
    if (!h || !h->active.ctx || !out64) return SSF_ERR_INVALID_ARG;
    HCK(hipStreamSynchronize(h->active.ctx->stream));
    HCK(hipMemcpy(out64, h->active.maps.epoch, 64 * sizeof(uint32_t), hipMemcpyDeviceToHost));
    return SSF_OK;
}
double ssf_dbg_time_pass(ssf_handle* h, int reps, int rgbd, int dbg, int nb) {
    if (!h || !h->active.ctx) return -1.0;
    ExtractCtx& c = *h->active.ctx;                   // all slots of the batch context (nb <= extract_batch)
    nb = std::max(1, std::min(nb, h->batch));
    hipEvent_t e0, e1; (void)hipEventCreate(&e0); (void)hipEventCreate(&e1);
    const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 1, 1, 0};
    for (int i = 0; i < 4; i++) launch_update_pass(h->stream, h->seg, c.maps, nb, 20 + i, ox[i & 3], oy[i & 3], rgbd != 0, dbg);
    (void)hipEventRecord(e0, h->stream);
    for (int i = 0; i < reps; i++) launch_update_pass(h->stream, h->seg, c.maps, nb, 24 + i, ox[i & 3], oy[i & 3], rgbd != 0, dbg);
    (void)hipEventRecord(e1, h->stream);
    (void)hipStreamSynchronize(h->stream);
    float ms = 0.f; (void)hipEventElapsedTime(&ms, e0, e1);
    (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
    return 1000.0 * ms / reps;
}
// one pass launch over nb frames of the active context with every workgroup leaving its five ticks (g_pass_trace, ssf_extract.hip);
// `k` = the pass number (>= 20 with rgbd: the frames' state is that of a finished extract, a late pass' workload).  out: 5 words per
// workgroup, grid3 = the launch's grid.  Returns the workgroups copied, < 0: n/a.  Leaves the segmentation state advanced by five passes.
int ssf_dbg_trace_pass(ssf_handle* h, int rgbd, int nb, unsigned long long* out, int cap_wgs, int* grid3) {
    if (!h || !h->active.ctx || !out || !grid3) return -1;
    ExtractCtx& c = *h->active.ctx;
    nb = std::max(1, std::min(nb, h->batch));
    const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 1, 1, 0};
    grid3[0] = (h->cfg.width + 30 + 31) / 32; grid3[1] = (h->cfg.height + 31) / 32; grid3[2] = nb;
    const int n = grid3[0] * grid3[1] * grid3[2];
    unsigned long long* d = nullptr;
    if (hipMalloc((void**)&d, (size_t)n * 40) != hipSuccess) return -2;
    (void)hipMemset(d, 0, (size_t)n * 40);
    for (int i = 0; i < 4; i++) launch_update_pass(h->stream, h->seg, c.maps, nb, 20 + i, ox[i & 3], oy[i & 3], rgbd != 0, 0);
    (void)hipStreamSynchronize(h->stream);
    set_pass_trace(d);
    launch_update_pass(h->stream, h->seg, c.maps, nb, 24, ox[0], oy[0], rgbd != 0, 0);
    (void)hipStreamSynchronize(h->stream);
    set_pass_trace(nullptr);
    const int m = std::min(n, cap_wgs);
    (void)hipMemcpy(out, d, (size_t)m * 40, hipMemcpyDeviceToHost);
    (void)hipFree(d);
    return m;
}

#endif
// ---- test hooks (include/ssf_testing.h): the host solvers, so they can be pinned on a CPU box ----------
int ssf_dbg_ldlt_solve6(const double* A, const double* b, double* x) { sym6_ldlt_solve(A, b, x); return 0; }
int ssf_dbg_lu_inverse6(const double* A, double* Ainv) { mat6_inverse_lu(A, Ainv); return 0; }
int ssf_dbg_renormalise_d(double* R9) { renormalise_rotation<double>(R9); return 0; }
int ssf_dbg_renormalise_f(float* R9) { renormalise_rotation<float>(R9); return 0; }
int ssf_dbg_gn_increment(const double* X6, double* tf16) { gn_increment(X6, tf16); return 0; }
int ssf_dbg_align_increment(const double* JtJ, const double* Jtr, float scale, const float* cs, const float* ct, double* tf16) {
    align_increment(JtJ, Jtr, scale, cs, ct, tf16); return 0;
}

int ssf_dbg_rgb_to_lab(const float* c, float* o) { V3 r = rgb_to_lab(v3(c[0], c[1], c[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_lab_to_rgb(const float* c, float* o) { V3 r = lab_to_rgb(v3(c[0], c[1], c[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_sym_inverse(const float* c, float* o) {
    Sym3 out; const bool ok = sym_inverse(sym3(c[0], c[1], c[2], c[3], c[4], c[5]), out);
    o[0] = out.xx; o[1] = out.xy; o[2] = out.xz; o[3] = out.yy; o[4] = out.yz; o[5] = out.zz; return ok ? 1 : 0;
}
int ssf_dbg_principal_frame(const float* c, float* vecs, float* vals) {
    M3 m; V3 v; principal_frame(sym3(c[0], c[1], c[2], c[3], c[4], c[5]), m, v);
    const float o[9] = {m.r0.x, m.r0.y, m.r0.z, m.r1.x, m.r1.y, m.r1.z, m.r2.x, m.r2.y, m.r2.z};
    std::memcpy(vecs, o, sizeof(o)); vals[0] = v.x; vals[1] = v.y; vals[2] = v.z; return 0;
}
// img9: a 3 x 3 label patch, row-major; returns 1 when the centre pixel is a bridge (its label may not change)
int ssf_dbg_connectivity_guard(const int32_t* g) { return guard_unchangeable(guard_ring(g[4], g[0], g[1], g[2], g[5], g[8], g[7], g[6], g[3])) ? 1 : 0; }
int ssf_dbg_plane_solve(const float* r, float* th) {
    float a = 0, b = 0, c = 0;
    const bool ok = plane_solve(a, b, c, r[0], r[1], r[2], r[3], r[4], r[5], r[6], r[7], r[8], r[9], r[10], r[11]);
    th[0] = a; th[1] = b; th[2] = c; return ok ? 1 : 0;
}

static M3 m3_from9(const float* a) { return m3(v3(a[0], a[1], a[2]), v3(a[3], a[4], a[5]), v3(a[6], a[7], a[8])); }
static void m3_to9(const M3& m, float* o) { o[0] = m.r0.x; o[1] = m.r0.y; o[2] = m.r0.z; o[3] = m.r1.x; o[4] = m.r1.y; o[5] = m.r1.z; o[6] = m.r2.x; o[7] = m.r2.y; o[8] = m.r2.z; }
static void sym_to6(const Sym3& s, float* o) { o[0] = s.xx; o[1] = s.xy; o[2] = s.xz; o[3] = s.yy; o[4] = s.yz; o[5] = s.zz; }
int ssf_dbg_sym_square(const float* c, float* o) { sym_to6(sym_square(sym3(c[0], c[1], c[2], c[3], c[4], c[5])), o); return 0; }
int ssf_dbg_sym_mulv(const float* c, const float* v, float* o) { V3 r = sym_mul(sym3(c[0], c[1], c[2], c[3], c[4], c[5]), v3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_mult_abat(const float* R9, const float* c, float* o) { sym_to6(rot_sym(m3_from9(R9), sym3(c[0], c[1], c[2], c[3], c[4], c[5])), o); return 0; }
int ssf_dbg_m3_mul(const float* A9, const float* B9, float* o) { m3_to9(m3_mul(m3_from9(A9), m3_from9(B9)), o); return 0; }
int ssf_dbg_m3_mulv(const float* A9, const float* v, float* o) { V3 r = m3_mulv(m3_from9(A9), v3(v[0], v[1], v[2])); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_row_mul(const float* v, const float* A9, float* o) { V3 r = row_mul(v3(v[0], v[1], v[2]), m3_from9(A9)); o[0] = r.x; o[1] = r.y; o[2] = r.z; return 0; }
int ssf_dbg_rot_to_quat(const float* R9, float* q4) { rot_to_quat(m3_from9(R9), q4); return 0; }
int ssf_dbg_quat_to_rot(const float* q4, float* R9) { m3_to9(quat_to_rot_quirk(q4), R9); return 0; }

}  // extern "C"
