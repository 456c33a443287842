// oracle_track_fuse.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restatement of the ICP stage (DenseRegistration::featureConstrainedSymmetricICP,
// core/src/dense_registration.cu:245-424; kernel computeSymmetricICPSystem,
// core/include/supersurfel_fusion/dense_registration_kernels.cuh:175-291), of the pose
// composition (core/src/supersurfel_fusion.cu:232-240,313-328) and of the fuse block
// (core/src/supersurfel_fusion.cu:351-483; kernels core/src/supersurfel_fusion_kernels.cu:
// 348-467,522-682), plus the "next" row applyDeformation (deformation_graph_kernels.cu:27-73).
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>
#include <vector>
#include "oracle.h"
// ORC_HOST marks the functions that restate HOST code of the reference (Eigen solves, pose composition): --use_fast_math is a
// device-code switch, so the FMA study arm (oracle/Makefile `arms`) must not contract them
#if defined(SSF_ORACLE_ARMS) && defined(__FMA__)
#define ORC_HOST __attribute__((optimize("fp-contract=off")))
#else
#define ORC_HOST
#endif

namespace orc {

// lroundf(v) for pixel coordinates; anything that cannot be an in-image pixel maps to -1
// (lroundf of inf/NaN/huge is undefined in C; the reference never guards it).
static inline int project_round(float v) {
    if (!(fabsf(v) < 8388608.0f)) return -1;
    return round_half_away(v);
}

ORC_HOST static void mat4_mul(const double* a, const double* b, double* c) {
    double r[16];
    for (int i = 0; i < 4; i++)
        for (int j = 0; j < 4; j++)
            r[i * 4 + j] = ((a[i * 4 + 0] * b[0 * 4 + j] + a[i * 4 + 1] * b[1 * 4 + j]) + a[i * 4 + 2] * b[2 * 4 + j]) + a[i * 4 + 3] * b[3 * 4 + j];
    std::memcpy(c, r, sizeof(r));
}

static void refresh_inc(State& s) {                       // dense_registration.cu:291-299
    IcpState& I = s.icp;
    for (int i = 0; i < 3; i++)
        I.R_inc.r[i] = mk3((float)I.tf_inc[i * 4 + 0], (float)I.tf_inc[i * 4 + 1], (float)I.tf_inc[i * 4 + 2]);
    I.t_inc = mk3((float)I.tf_inc[3], (float)I.tf_inc[7], (float)I.tf_inc[11]);
}

ORC_HOST void icp_begin(State& s, const float* prior) {
    IcpState& I = s.icp;
    if (prior) {                                          // pose = vo->getPose(), supersurfel_fusion.cu:228
        for (int i = 0; i < 3; i++) s.pose.R.r[i] = mk3(prior[3 * i], prior[3 * i + 1], prior[3 * i + 2]);
        s.pose.t = mk3(prior[9], prior[10], prior[11]);
    }
    const int64_t nvis = (s.cfg.nranks > 1 && s.global_n_visible >= 0) ? s.global_n_visible : s.n_visible;
    I.active = nvis > 0 && s.cfg.icp_iter > 0;            // if(nbVisible > 0), supersurfel_fusion.cu:232
    I.valid = true; I.done = !I.active; I.iter = 0;
    I.R_init = transpose(s.pose.R);                        // :234-235
    I.t_init = neg(I.R_init * s.pose.t);
    for (int i = 0; i < 16; i++) I.tf_inc[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 36; i++) I.JtJ[i] = 0.0;
    I.prev_error = DBL_MAX;
    I.R_rel = identity33(); I.t_rel = mk3(0, 0, 0);
    I.R_inc = identity33(); I.t_inc = mk3(0, 0, 0);
    s.last_icp_valid = 0; s.last_icp_iters = 0;
}

// computeSymmetricICPSystem<128>, dense_registration_kernels.cuh:175-291, over this shard's
// visible rows; sums are the exact fixed-point record of ssf.h (decision A4).
void icp_accumulate(State& s, int64_t* sums) {
    IcpState& I = s.icp;
    const ssf_config& c = s.cfg;
    for (int i = 0; i < SSF_ICP_RECORD; i++) sums[i] = 0;
    refresh_inc(s);
    I.R_corres = I.R_inc * I.R_init;                       // dense_registration.cu:298-299
    I.t_corres = I.R_inc * I.t_init + I.t_inc;
    const Mat33 R = I.R_corres; const f3 t = I.t_corres;
    const int W = s.W, H = s.H;
    // (exact integer sums: the OpenMP build accumulates per thread and adds the partial records, same bits)
#pragma omp parallel
    {
    int64_t part[SSF_ICP_RECORD];
    for (int i = 0; i < SSF_ICP_RECORD; i++) part[i] = 0;
#pragma omp for schedule(static)
    for (int id = 0; id < s.n_visible; id++) {
        f3 ps = R * s.model.pos[id] + t;
        int u = project_round(fdiv(ps.x * c.fx, ps.z) + c.cx);
        int v = project_round(fdiv(ps.y * c.fy, ps.z) + c.cy);
        if (!(u >= 0 && u < W && v >= 0 && v < H)) continue;
        const size_t p = (size_t)v * W + u;
        const int tid = s.label[p];
        const float zt = s.plane_depth[p];
        if (!(s.frame.conf[tid] > 0.0f && zt >= 0.2f && zt <= 5.0f)) continue;   // hard-coded range (:224)
        const float dist_color = length(s.model_lab[id] - s.frame_lab[tid]);
        const f3 pt = mk3(fdiv(zt * ((float)u - c.cx), c.fx), fdiv(zt * ((float)v - c.cy), c.fy), zt);
        const f3 nt = s.frame.orient[tid].r[2];
        const f3 ns = normalize(R * s.model.orient[id].r[2]);
        if (!(dist_color < 20.0f && length(ps - pt) < 0.1f && fabsf(dot(nt, ns)) > 0.8f)) continue;
        const f3 d = pt - ps, c1 = cross(pt, ns), c2 = cross(ps, nt);
        const float dn1 = dot(d, ns), dn2 = dot(d, nt);
        const float x1[6] = {c1.x, c1.y, c1.z, ns.x, ns.y, ns.z};
        const float x2[6] = {c2.x, c2.y, c2.z, nt.x, nt.y, nt.z};
        // fixed point (this build's, A4): JtJ at 2^20 = the two rows scaled by 2^10 before their products, Jtr at 2^24 = the
        // two residuals scaled by 2^14 times the scaled rows -- powers of two, so the same real numbers as scaling the sums, 14 scalings per
        // row instead of 27
        float X1[6], X2[6];
        for (int i = 0; i < 6; i++) { X1[i] = x1[i] * 1024.0f; X2[i] = x2[i] * 1024.0f; }
        static_assert(SSF_ICP_SCALE_JTJ == 1024.0 * 1024.0 && SSF_ICP_SCALE_JTR == 16777216.0, "scales of the ICP record");
        const float D1 = dn1 * 16384.0f, D2 = dn2 * 16384.0f;                 // 2^14 x 2^10 (the scaled rows) = 2^24
        int k = 0;
        for (int i = 0; i < 6; i++)
            for (int j = i; j < 6; j++, k++)
                part[k] += (int64_t)fx_quant32r(X1[i] * X1[j] + X2[i] * X2[j]);
        for (int i = 0; i < 6; i++)
            part[21 + i] += (int64_t)fx_quant32r(D1 * X1[i] + D2 * X2[i]);
        part[27] += fx_quant((double)(dn2 * dn2), SSF_ICP_SCALE_R, 4611686018427387904.0);
        part[28] += 1;
    }
#pragma omp critical
    for (int i = 0; i < SSF_ICP_RECORD; i++) sums[i] += part[i];
    }
}

// host part of one iteration, dense_registration.cu:324-391
ORC_HOST void icp_update(State& s, const int64_t* sums, int* again) {
    IcpState& I = s.icp;
    *again = 0;
    if (!I.active || I.done) return;
    I.iter++;
    s.last_icp_iters = I.iter;
    static const int tri[6][6] = {{0, 1, 2, 3, 4, 5}, {1, 6, 7, 8, 9, 10}, {2, 7, 11, 12, 13, 14},
                                  {3, 8, 12, 15, 16, 17}, {4, 9, 13, 16, 18, 19}, {5, 10, 14, 17, 19, 20}};
    double Jtr[6];
    for (int i = 0; i < 6; i++) {
        for (int j = 0; j < 6; j++) I.JtJ[i * 6 + j] = (double)sums[tri[i][j]] / SSF_ICP_SCALE_JTJ;
        Jtr[i] = (double)sums[21 + i] / SSF_ICP_SCALE_JTR;
    }
    const float r = (float)((double)sums[27] / SSF_ICP_SCALE_R);
    const float inliers = (float)sums[28];
    const double error = std::sqrt((double)(r / inliers));                      // :333
    if (inliers < 100.0f) { I.valid = false; I.done = true; return; }           // :336-341
    double X[6];
    ldlt_solve6(I.JtJ, Jtr, X);                                                 // :367
    double tran[3] = {X[3], X[4], X[5]}, axis[3] = {X[0], X[1], X[2]};
    const double nrm = std::sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
    const double angle = 0.5 * std::atan(nrm);                                  // :373
    double Rr[9];
    if (nrm == 0.0) {                     // deviation: the reference divides by zero here (NaN pose)
        for (int i = 0; i < 9; i++) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0;
    } else {
        for (int i = 0; i < 3; i++) axis[i] /= nrm;
        angle_axis_to_rot_d(angle, axis, Rr);
    }
    const double ca = std::cos(angle);
    for (int i = 0; i < 3; i++) tran[i] *= ca;                                  // :375
    double tf_iter[16] = {0};                                                   // iso_rot * Trans * iso_rot (:378)
    for (int i = 0; i < 3; i++) {
        for (int j = 0; j < 3; j++)
            tf_iter[i * 4 + j] = (Rr[i * 3 + 0] * Rr[0 * 3 + j] + Rr[i * 3 + 1] * Rr[1 * 3 + j]) + Rr[i * 3 + 2] * Rr[2 * 3 + j];
        tf_iter[i * 4 + 3] = (Rr[i * 3 + 0] * tran[0] + Rr[i * 3 + 1] * tran[1]) + Rr[i * 3 + 2] * tran[2];
    }
    tf_iter[15] = 1.0;
    double R9[9];
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) R9[i * 3 + j] = tf_iter[i * 4 + j];
    quat_normalize_rot_d(R9);                                                   // :384
    for (int i = 0; i < 3; i++) for (int j = 0; j < 3; j++) tf_iter[i * 4 + j] = R9[i * 3 + j];
    mat4_mul(tf_iter, I.tf_inc, I.tf_inc);                                      // :386
    if (!s.cfg.icp_force_iters && error / I.prev_error > 0.9995) { I.done = true; return; }   // :388
    I.prev_error = error;
    if (I.iter >= s.cfg.icp_iter) { I.done = true; return; }
    *again = 1;
}

// dense_registration.cu:394-421 and supersurfel_fusion.cu:313-328
ORC_HOST void icp_end(State& s, int* valid) {
    IcpState& I = s.icp;
    *valid = 0;
    if (!I.active) return;
    bool ok = I.valid;
    double cov[36];
    lu_inverse6(I.JtJ, cov);
    for (int i = 0; i < 6; i++) if (cov[i * 6 + i] > s.cfg.icp_cov_thresh) { ok = false; break; }
    if (ok) {
        if (length(I.t_inc) > 0.2f) ok = false;            // stale t_inc (start of the last iteration)
        else {
            refresh_inc(s);
            I.R_rel = transpose(I.R_inc);
            I.t_rel = neg(I.R_rel * I.t_inc);
        }
    }
    if (ok) {
        s.pose.t = s.pose.R * I.t_rel + s.pose.t;
        s.pose.R = s.pose.R * I.R_rel;
        float R9[9];
        for (int i = 0; i < 3; i++) { R9[3 * i] = s.pose.R.r[i].x; R9[3 * i + 1] = s.pose.R.r[i].y; R9[3 * i + 2] = s.pose.R.r[i].z; }
        quat_normalize_rot_f(R9);
        for (int i = 0; i < 3; i++) s.pose.R.r[i] = mk3(R9[3 * i], R9[3 * i + 1], R9[3 * i + 2]);
    }
    *valid = ok ? 1 : 0;
    s.last_icp_valid = *valid;
    I.active = false;
}

// ---- loop-closure registration: DenseRegistration::align, dense_registration.cu:52-243 ------------------
// host step of one iteration (:168-210): LDLT solve, half-angle rotation, translation un-scaled,
// iso_iter = T(target_centroid) * Rot * T(tran) * Rot * T(-source_centroid) (Eigen Isometry products, left to
// right: linear = a.linear * b.linear, translation = a.linear * b.translation + a.translation), rotation block
// re-normalised through a quaternion
ORC_HOST void align_increment(const double* JtJ, const double* Jtr, float scale, const float* cs, const float* ct, double* tf_iter) {
    double X[6];
    ldlt_solve6(JtJ, Jtr, X);                                                   // :186
    double tran[3] = {X[3], X[4], X[5]}, axis[3] = {X[0], X[1], X[2]};
    const double nrm = std::sqrt((axis[0] * axis[0] + axis[1] * axis[1]) + axis[2] * axis[2]);
    const double angle = 0.5 * std::atan(nrm);                                  // :192
    double Rr[9];
    if (nrm == 0.0) { for (int i = 0; i < 9; i++) Rr[i] = (i % 4 == 0) ? 1.0 : 0.0; }   // deviation: reference -> NaN
    else { for (int i = 0; i < 3; i++) axis[i] /= nrm; angle_axis_to_rot_d(angle, axis, Rr); }
    const double ca = std::cos(angle);
    for (int i = 0; i < 3; i++) { tran[i] /= (double)scale; tran[i] *= ca; }    // :194-195
    double RR[9], t2[3], t4[3];
    for (int i = 0; i < 3; i++) {
        t2[i] = ((Rr[i * 3 + 0] * tran[0] + Rr[i * 3 + 1] * tran[1]) + Rr[i * 3 + 2] * tran[2]) + (double)ct[i];
        for (int j = 0; j < 3; j++)
            RR[i * 3 + j] = (Rr[i * 3 + 0] * Rr[0 * 3 + j] + Rr[i * 3 + 1] * Rr[1 * 3 + j]) + Rr[i * 3 + 2] * Rr[2 * 3 + j];
    }
    const double ncs[3] = {-1.0 * (double)cs[0], -1.0 * (double)cs[1], -1.0 * (double)cs[2]};
    for (int i = 0; i < 3; i++) t4[i] = ((RR[i * 3 + 0] * ncs[0] + RR[i * 3 + 1] * ncs[1]) + RR[i * 3 + 2] * ncs[2]) + t2[i];
    quat_normalize_rot_d(RR);                                                   // :205
    for (int i = 0; i < 16; i++) tf_iter[i] = 0.0;
    for (int i = 0; i < 3; i++) { for (int j = 0; j < 3; j++) tf_iter[i * 4 + j] = RR[i * 3 + j]; tf_iter[i * 4 + 3] = t4[i]; }
    tf_iter[15] = 1.0;
}

// One pass of makeCorrespondences (dense_registration_kernels.cu:27-100) over the sources; `visit` is called for
// every valid pair.  The reference compacts the pairs (thrust::remove_if) only to feed reductions; with exact
// integer sums the order is irrelevant, so the pairs are consumed in place (decision A4).
template <typename F>
static void for_each_pair(State& s, const f3* src_pos, const f3* src_lab, const Mat33* src_orient, const float* src_conf, int n,
                          const Mat33& R, const f3& t, F visit) {
    const ssf_config& c = s.cfg;
    for (int id = 0; id < n; id++) {
        if (src_conf && !(src_conf[id] > 0.0f)) continue;
        const f3 pv = R * src_pos[id] + t;
        const int u = project_round(fdiv(pv.x * c.fx, pv.z) + c.cx), v = project_round(fdiv(pv.y * c.fy, pv.z) + c.cy);
        if (!(u >= 0 && u < s.W && v >= 0 && v < s.H)) continue;
        const size_t p = (size_t)v * s.W + u;
        const int tid = s.label[p];
        if (!(s.frame.conf[tid] > 0.0f)) continue;
        const float dist_color = length(src_lab[id] - s.frame_lab[tid]);
        const float td = s.plane_depth[p];
        if (!std::isfinite(td)) continue;
        f3 sn = normalize(src_orient[id].r[2]);
        sn = normalize(R * sn);
        const f3 tn = normalize(s.frame.orient[tid].r[2]);
        const f3 tp = mk3(fdiv(td * ((float)u - c.cx), c.fx), fdiv(td * ((float)v - c.cy), c.fy), td);
        if (!(dist_color < 20.0f && length(pv - tp) < 0.1f && fabsf(dot(sn, tn)) > 0.8f)) continue;
        visit(pv, sn, tp, tn);
    }
}

bool align(State& s, const f3* src_pos, const f3* src_col, const Mat33* src_orient, const float* src_conf, int n,
           const float* init12, float* rel12, int* iters, int* pairs_last) {
    Mat33 R_init = identity33(); f3 t_init = mk3(0, 0, 0);
    if (init12) {
        for (int i = 0; i < 3; i++) R_init.r[i] = mk3(init12[3 * i], init12[3 * i + 1], init12[3 * i + 2]);
        t_init = mk3(init12[9], init12[10], init12[11]);
    }
    std::vector<f3> src_lab(n);
    for (int i = 0; i < n; i++) src_lab[i] = rgbToLab(src_col[i]);
    double tf_inc[16], JtJ[36];
    for (int i = 0; i < 16; i++) tf_inc[i] = (i % 5 == 0) ? 1.0 : 0.0;
    for (int i = 0; i < 36; i++) JtJ[i] = 0.0;
    Mat33 R_inc = identity33(); f3 t_inc = mk3(0, 0, 0);
    bool valid = true;
    int it = 0, pairs = 0;
    static const int tri[6][6] = {{0, 1, 2, 3, 4, 5}, {1, 6, 7, 8, 9, 10}, {2, 7, 11, 12, 13, 14},
                                  {3, 8, 12, 15, 16, 17}, {4, 9, 13, 16, 18, 19}, {5, 10, 14, 17, 19, 20}};
    while (it < s.cfg.icp_iter) {                                               // while(iter++ < nbIter), :86
        it++;
        for (int i = 0; i < 3; i++) R_inc.r[i] = mk3((float)tf_inc[i * 4], (float)tf_inc[i * 4 + 1], (float)tf_inc[i * 4 + 2]);
        t_inc = mk3((float)tf_inc[3], (float)tf_inc[7], (float)tf_inc[11]);
        const Mat33 R = R_inc * R_init;                                         // :96-97
        const f3 t = R_inc * t_init + t_inc;
        // pass 1: pair count and centroids (:131-150), exact fixed-point sums (2^24)
        int64_t cs_sum[3] = {0, 0, 0}, ct_sum[3] = {0, 0, 0}; pairs = 0;
        for_each_pair(s, src_pos, src_lab.data(), src_orient, src_conf, n, R, t, [&](const f3& ps, const f3&, const f3& pt, const f3&) {
            cs_sum[0] += fx_quant(ps.x, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM); cs_sum[1] += fx_quant(ps.y, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM);
            cs_sum[2] += fx_quant(ps.z, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM);
            ct_sum[0] += fx_quant(pt.x, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM); ct_sum[1] += fx_quant(pt.y, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM);
            ct_sum[2] += fx_quant(pt.z, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM);
            pairs++;
        });
        if (pairs < 100) { valid = false; break; }                              // :133-138
        float cs[3], ct[3];
        for (int i = 0; i < 3; i++) {
            cs[i] = (float)((double)cs_sum[i] / SSF_ALIGN_SCALE_POS) / (float)pairs;
            ct[i] = (float)((double)ct_sum[i] / SSF_ALIGN_SCALE_POS) / (float)pairs;
        }
        // pass 2: scale (:152-158)
        int64_t d2 = 0;
        for_each_pair(s, src_pos, src_lab.data(), src_orient, src_conf, n, R, t, [&](const f3& ps, const f3&, const f3& pt, const f3&) {
            const f3 a = mk3(pt.x - ct[0], pt.y - ct[1], pt.z - ct[2]), b = mk3(ps.x - cs[0], ps.y - cs[1], ps.z - cs[2]);
            d2 += fx_quant((double)((a.x * a.x + a.y * a.y) + a.z * a.z), SSF_ALIGN_SCALE_D2, SSF_ALIGN_LIM);
            d2 += fx_quant((double)((b.x * b.x + b.y * b.y) + b.z * b.z), SSF_ALIGN_SCALE_D2, SSF_ALIGN_LIM);
        });
        float scale = (float)((double)d2 / SSF_ALIGN_SCALE_D2);
        scale = std::sqrt(scale / (2.0f * (float)pairs));
        scale = 1.0f / scale;
        // pass 3: buildSymmetricPoint2PlaneSystem (dense_registration_kernels.cuh:87-173), record as in ssf.h
        int64_t sums[SSF_ICP_RECORD];
        for (int i = 0; i < SSF_ICP_RECORD; i++) sums[i] = 0;
        const f3 csv = mk3(cs[0], cs[1], cs[2]), ctv = mk3(ct[0], ct[1], ct[2]);
        for_each_pair(s, src_pos, src_lab.data(), src_orient, src_conf, n, R, t, [&](const f3& psrc, const f3& nsrc, const f3& ptgt, const f3& ntgt) {
            const f3 ps = scale * (psrc - csv), pt = scale * (ptgt - ctv);
            const f3 ns = normalize(nsrc), nt = normalize(ntgt);
            const f3 d = pt - ps, c1 = cross(pt, ns), c2 = cross(ps, nt);
            const float dn1 = dot(d, ns), dn2 = dot(d, nt);
            const float x1[6] = {c1.x, c1.y, c1.z, ns.x, ns.y, ns.z};
            const float x2[6] = {c2.x, c2.y, c2.z, nt.x, nt.y, nt.z};
            int k = 0;
            for (int i = 0; i < 6; i++)
                for (int j = i; j < 6; j++, k++)
                    sums[k] += (int64_t)fx_quant32(x1[i] * x1[j] + x2[i] * x2[j], (float)SSF_ICP_SCALE_JTJ);
            for (int i = 0; i < 6; i++)
                sums[21 + i] += (int64_t)fx_quant32(dn1 * x1[i] + dn2 * x2[i], (float)SSF_ICP_SCALE_JTR);
            sums[27] += fx_quant((double)(dn2 * dn2), SSF_ICP_SCALE_R, 4611686018427387904.0);
            sums[28] += 1;
        });
        double Jtr[6];
        for (int i = 0; i < 6; i++) {
            for (int j = 0; j < 6; j++) JtJ[i * 6 + j] = (double)sums[tri[i][j]] / SSF_ICP_SCALE_JTJ;
            Jtr[i] = (double)sums[21 + i] / SSF_ICP_SCALE_JTR;
        }
        double tf_iter[16];
        align_increment(JtJ, Jtr, scale, cs, ct, tf_iter);
        mat4_mul(tf_iter, tf_inc, tf_inc);                                      // :210
    }
    double cov[36];
    lu_inverse6(JtJ, cov);                                                      // :213
    for (int i = 0; i < 6; i++) if (cov[i * 6 + i] > s.cfg.icp_cov_thresh) { valid = false; break; }
    Mat33 Rr = identity33(); f3 tr = mk3(0, 0, 0);
    if (valid) {
        if (length(t_inc) > 0.3f) valid = false;                                // stale t_inc, :226
        else { Rr = transpose(R_inc); tr = neg(Rr * t_inc); }                   // :230-231 (R_inc of the last iteration's start)
    }
    for (int i = 0; i < 3; i++) { rel12[3 * i] = Rr.r[i].x; rel12[3 * i + 1] = Rr.r[i].y; rel12[3 * i + 2] = Rr.r[i].z; }
    rel12[9] = tr.x; rel12[10] = tr.y; rel12[11] = tr.z;
    if (iters) *iters = it;
    if (pairs_last) *pairs_last = pairs;
    return valid;
}

// findBestMatches, supersurfel_fusion_kernels.cu:522-599 (decision A13)
void match(State& s, uint64_t* best, uint8_t* matched) {
    const ssf_config& c = s.cfg;
    for (int f = 0; f < s.S; f++) { best[f] = SSF_NO_MATCH; matched[f] = 0; }
    const int64_t nmodel = (c.nranks > 1 && s.global_n_model >= 0) ? s.global_n_model : s.n_model;
    const int64_t nvis = (c.nranks > 1 && s.global_n_visible >= 0) ? s.global_n_visible : s.n_visible;
    if (!(nmodel > 0 && nvis > 0)) return;                  // supersurfel_fusion.cu:351,356
    const Mat33 R = s.pose.R; const f3 t = s.pose.t;
    const Mat33 Rview = transpose(R);
    const f3 tview = neg(Rview * t);
    const Mat33 Rt = transpose(R);
    // (MIN of packed keys / OR of flags: order-free, so the OpenMP build keeps a table per thread and folds them)
#pragma omp parallel
    {
    std::vector<uint64_t> lbest(s.S, SSF_NO_MATCH);
    std::vector<uint8_t> lmatched(s.S, 0);
#pragma omp for schedule(static)
    for (int id = 0; id < s.n_visible; id++) {
        if (!(s.model.conf[id] > 0.0f)) continue;
        const f3 mp = s.model.pos[id];
        const f3 pv = Rview * mp + tview;
        if (!(pv.z > c.range_min && pv.z < c.range_max)) continue;
        int px = project_round(fdiv(pv.x * c.fx, pv.z) + c.cx), py = project_round(fdiv(pv.y * c.fy, pv.z) + c.cy);
        if (!(px >= 0 && px < s.W && py >= 0 && py < s.H)) continue;
        const int f = s.label[(size_t)py * s.W + px];
        lmatched[f] = 1;                                    // unconditional (:570)
        if (!(s.frame.conf[f] > 0.0f)) continue;
        const f3 fp = R * s.frame.pos[f] + t;
        const Mat33 frot = s.frame.orient[f] * Rt;
        const f3 fn = normalize(frot.r[2]);
        const f3 mn = normalize(s.model.orient[id].r[2]);
        const float dist = length(mp - fp);
        const float lab_dist = length(s.model_lab[id] - s.frame_lab[f]);
        const float delta_norm = fabsf(dot(mn, fn));
        if (lab_dist < 15.0f && delta_norm > 0.8f && dist < 0.05f) {
            uint32_t bits; std::memcpy(&bits, &dist, 4);
            uint64_t key = ((uint64_t)bits << 32) | (uint64_t)(uint32_t)(s.id_offset + id);
#ifdef SSF_ORACLE_ARMS
            // STUDY ARMS of the reference's torn arg-min (supersurfel_fusion_kernels.cu:590-594: a plain compare followed by two
            // independent atomicExch): 1 = equal distances go to the HIGHEST id; 2 = every candidate saw the initial 0.05 and the
            // last store (highest id) stays -- the worst valid outcome
            if (g_arms.tie == 1 && lbest[f] != SSF_NO_MATCH && (uint32_t)(lbest[f] >> 32) == bits) { lbest[f] = key; continue; }
            if (g_arms.tie == 2) { lbest[f] = key; continue; }
#endif
            if (key < lbest[f]) lbest[f] = key;
        }
    }
#pragma omp critical
    for (int f = 0; f < s.S; f++) { if (lbest[f] < best[f]) best[f] = lbest[f]; matched[f] |= lmatched[f]; }
    }
}

// spatial-tile owner of frame surfel f (multi-GPU sharding; no reference counterpart)
int shard_owner(const State& s, int f, const Pose& pose) {
    const int n = s.cfg.nranks;
    if (n <= 1) return 0;
    if (!(s.frame.conf[f] > 0.0f)) return f % n;
    const f3 pw = pose.R * s.frame.pos[f] + pose.t;
    const float tile = s.cfg.shard_tile;
    const int32_t ix = (int32_t)floorf(pw.x / tile), iy = (int32_t)floorf(pw.y / tile), iz = (int32_t)floorf(pw.z / tile);
    const uint32_t h = ((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u);
    return (int)(h % (uint32_t)n);
}

// updateSupersurfels, supersurfel_fusion_kernels.cu:601-682
static void update_one(State& s, int f, int m) {
    const Mat33 R = s.pose.R; const f3 t = s.pose.t;
    Surfels& M = s.model; const Surfels& F = s.frame;
    const f3 model_position = M.pos[m];
    const f3 frame_position = R * F.pos[f] + t;
    const Cov3 frame_shape = mult_ABAt(R, F.shape[f]);
    const f3 frame_lab = s.frame_lab[f], model_lab = s.model_lab[m];
    const float m_conf = M.conf[m], f_conf = F.conf[f];
    const float ratio = fdiv(1.0f, m_conf + f_conf);
    M.stamps[2 * m + 1] = s.stamp;
    const f3 fused_color = labToRgb(ratio * (f_conf * frame_lab + m_conf * model_lab));
    Cov3 f1, m1, fused_shape, fused_1;
    f3 fused_position;
    const float w = ratio * f_conf;
    bool info = false;
    if (inverse(frame_shape, f1) && inverse(M.shape[m], m1)) {
        fused_1 = w * f1 + (1.0f - w) * m1;
        if (inverse(fused_1, fused_shape)) {
            fused_position = fused_shape * ((w * f1) * frame_position + ((1.0f - w) * m1) * model_position);
            info = true;
        }
    }
    if (!info) {
        fused_shape = ratio * (f_conf * frame_shape + m_conf * M.shape[m]);
        fused_position = ratio * (f_conf * frame_position + m_conf * model_position);
    }
    M.pos[m] = fused_position;
    M.conf[m] = m_conf + f_conf;
    M.shape[m] = fused_shape;
    Mat33 vecs; f3 vals;
    eigenDecomposition(fused_shape, vecs, vals, 10);
    M.orient[m] = vecs;
    M.col[m] = fused_color;
    s.model_lab[m] = rgbToLab(fused_color);
    M.dims[2 * m] = vals.x; M.dims[2 * m + 1] = vals.y;
}

// spatial-tile owner of a world position (the same hash as shard_owner)
static int owner_of(const State& s, const f3& pw) {
    const int n = s.cfg.nranks;
    if (n <= 1) return 0;
    const float tile = s.cfg.shard_tile;
    const int32_t ix = (int32_t)floorf(pw.x / tile), iy = (int32_t)floorf(pw.y / tile), iz = (int32_t)floorf(pw.z / tile);
    const uint32_t h = ((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u);
    return (int)(h % (uint32_t)n);
}

// filterModel for one row, supersurfel_fusion_kernels.cu:397-467: 0 visible, 1 out of view, 2 removed
static int classify_row(const State& s, const Surfels& M, int i, const Mat33& Rview, const f3& tview) {
    const ssf_config& c = s.cfg;
    const int time_diff = s.stamp - M.stamps[2 * i + 1];
    if ((time_diff > c.delta_t && M.conf[i] < c.conf_thresh && s.stamp > c.delta_t) || M.conf[i] <= 0.0f) return 2;
    const f3 p = Rview * M.pos[i] + tview;
    if (p.z > c.range_min && p.z < c.range_max) {
        const float u = fdiv(c.fx * p.x, p.z) + c.cx, v = fdiv(c.fy * p.y, p.z) + c.cy;
        if (u >= 0.0f && u < (float)s.W && v >= 0.0f && v < (float)s.H) {
            const float z = s.plane_depth[(size_t)((int)floorf(v)) * s.W + (int)floorf(u)];
            return (p.z < 0.8f * z) ? 2 : 0;
        }
    }
    return 1;
}

// One slot of the migrant table (multi-GPU "halo exchange", no reference counterpart): 28 words per frame
// supersurfel f -- word 0 = destination rank + 1 (0: nothing in this slot), word 1 unused, words 2..27 = the row in
// the reference's layout (pos 3, colour 3, stamps 2, orientation 9, shape 6, dims 2, confidence 1; floats as their
// bits).  A frame supersurfel updates at most one model row in the whole map, so over all ranks at most one
// contributes a non-zero slot f: an int32 SUM all-reduce of the tables is their union, exactly.
static void row_to_slot(const Surfels& M, int m, int dest, int32_t* w) {
    w[0] = dest + 1; w[1] = 0;
    float v[26];
    v[0] = M.pos[m].x; v[1] = M.pos[m].y; v[2] = M.pos[m].z; v[3] = M.col[m].x; v[4] = M.col[m].y; v[5] = M.col[m].z;
    std::memcpy(&v[6], &M.stamps[2 * m], 8);
    for (int r = 0; r < 3; r++) { v[8 + 3 * r] = M.orient[m].r[r].x; v[9 + 3 * r] = M.orient[m].r[r].y; v[10 + 3 * r] = M.orient[m].r[r].z; }
    const Cov3& c = M.shape[m];
    v[17] = c.xx; v[18] = c.xy; v[19] = c.xz; v[20] = c.yy; v[21] = c.yz; v[22] = c.zz;
    v[23] = M.dims[2 * m]; v[24] = M.dims[2 * m + 1]; v[25] = M.conf[m];
    std::memcpy(&w[2], v, sizeof(v));
}
static void slot_to_row(Surfels& M, int k, const int32_t* w) {
    float v[26];
    std::memcpy(v, &w[2], sizeof(v));
    M.pos[k] = mk3(v[0], v[1], v[2]); M.col[k] = mk3(v[3], v[4], v[5]);
    std::memcpy(&M.stamps[2 * k], &v[6], 8);
    for (int r = 0; r < 3; r++) M.orient[k].r[r] = mk3(v[8 + 3 * r], v[9 + 3 * r], v[10 + 3 * r]);
    M.shape[k] = mkcov(v[17], v[18], v[19], v[20], v[21], v[22]);
    M.dims[2 * k] = v[23]; M.dims[2 * k + 1] = v[24]; M.conf[k] = v[25];
}

// First half of the fuse block (supersurfel_fusion.cu:351-395): update of the matched rows and ordered insertion.
// migrate != 0 (sharded maps): an updated row whose fused position now hashes to another rank's world tile -- and
// that this frame's filterModel would keep -- leaves this shard: it goes to slot f of `table` (SSF_MIGRANT_WORDS * S
// words, zeroed here) and is dropped from the local model by fuse_end.
void fuse_begin(State& s, const uint64_t* best, const uint8_t* matched, int migrate, int32_t* table) {
    const ssf_config& c = s.cfg;
    const int64_t nmodel_g = (c.nranks > 1 && s.global_n_model >= 0) ? s.global_n_model : s.n_model;
    const int64_t nvis_g = (c.nranks > 1 && s.global_n_visible >= 0) ? s.global_n_visible : s.n_visible;
    s.f_updated = 0; s.f_inserted = 0; s.f_first = !(nmodel_g > 0);
    s.f_old_rows = s.n_model;
    s.emigrant.assign(s.n_model, 0);
    if (table) for (size_t i = 0; i < (size_t)SSF_MIGRANT_WORDS * s.S; i++) table[i] = 0;
    Surfels& M = s.model;
    if (s.f_first) return;
    const Mat33 R = s.pose.R; const f3 t = s.pose.t;
    const Mat33 Rview = transpose(R);
    const f3 tview = neg(Rview * t);
    if (nvis_g > 0)                                                             // :356, update (:386)
        for (int f = 0; f < s.S; f++) {
            if (!matched[f] || best[f] == SSF_NO_MATCH) continue;                 // model_id >= 0 (:626)
            int64_t local = (int64_t)(uint32_t)(best[f] & 0xFFFFFFFFull) - s.id_offset;
            if (local < 0 || local >= s.n_visible) continue;                    // owned by another shard
            const int m = (int)local;
            update_one(s, f, m); s.f_updated++;
            if (migrate && c.nranks > 1 && table) {
                const int dest = owner_of(s, M.pos[m]);
                if (dest != c.rank && classify_row(s, M, m, Rview, tview) != 2) { row_to_slot(M, m, dest, &table[(size_t)SSF_MIGRANT_WORDS * f]); s.emigrant[m] = 1; }
            }
        }
    // insertSupersurfels, supersurfel_fusion_kernels.cu:348-395 (decision A14)
    const Mat33 Rt = transpose(R);
    for (int f_ = 0; f_ < s.S; f_++) {
#ifdef SSF_ORACLE_ARMS
        const int f = g_arms.insert_rev ? s.S - 1 - f_ : f_;      // STUDY ARM: another atomic arrival order (:376)
#else
        const int f = f_;
#endif
        if (!(s.frame.conf[f] > 0.0f) || matched[f]) continue;
        if (shard_owner(s, f, s.pose) != c.rank) continue;
        if (s.n_model >= c.nb_supersurfels_max) continue;
        const int k = s.n_model++;
        M.pos[k] = R * s.frame.pos[f] + t;
        M.conf[k] = s.frame.conf[f];
        M.col[k] = s.frame.col[f];
        s.model_lab[k] = s.frame_lab[f];
        M.stamps[2 * k] = s.stamp; M.stamps[2 * k + 1] = s.stamp;
        M.dims[2 * k] = s.frame.dims[2 * f]; M.dims[2 * k + 1] = s.frame.dims[2 * f + 1];
        M.orient[k] = s.frame.orient[f] * Rt;
        M.shape[k] = mult_ABAt(R, s.frame.shape[f]);
        s.f_inserted++;
    }
}

// Second half (supersurfel_fusion.cu:397-483): rows arriving from other shards (slots of the reduced migrant table
// addressed to this rank, ascending f) are appended behind this frame's insertions, then filterModel over every row
// and the stable 3-way reorder; emigrants are dropped without counting as removed.
void fuse_end(State& s, const int32_t* table, ssf_frame_result* out) {
    const ssf_config& c = s.cfg;
    int n_removed = 0, n_lost = 0;              // n_lost: arrivals from other shards that found this one full
    Surfels& M = s.model;
    if (!s.f_first) {                                                           // supersurfel_fusion.cu:351
        const Mat33 R = s.pose.R; const f3 t = s.pose.t;
        if (table)
            for (int f = 0; f < s.S; f++) {
                const int32_t* w = &table[(size_t)SSF_MIGRANT_WORDS * f];
                if (w[0] - 1 != c.rank) continue;
                // no room: the row is lost to the whole map (its old shard has let it go) -- counted as removed HERE, so that
                // the sums of the per-shard counters stay the unsharded bookkeeping (n_model before = after + removed - inserted)
                if (s.n_model >= c.nb_supersurfels_max) { n_lost++; continue; }
                const int k = s.n_model++;
                slot_to_row(M, k, w);
                s.model_lab[k] = rgbToLab(M.col[k]);
            }
        s.emigrant.resize(s.n_model, 0);
        // filterModel, supersurfel_fusion_kernels.cu:397-467
        const Mat33 Rview = transpose(R);
        const f3 tview = neg(Rview * t);
        std::vector<int> state(s.n_model);
        int n_vis = 0, n_gone = 0;
#pragma omp parallel for schedule(static) reduction(+ : n_vis, n_removed, n_gone)
        for (int i = 0; i < s.n_model; i++) {
            int st = classify_row(s, M, i, Rview, tview);
            if (st == 2) M.conf[i] = -1.0f;
            if (s.emigrant[i] && st != 2) { st = 2; n_gone++; }                // leaves the shard: dropped, not "removed"
            else if (st == 2) n_removed++;
            if (st == 0) n_vis++;
            state[i] = st;
        }
        // thrust::sort_by_key(states, model) -- stable 3-way partition (supersurfel_fusion.cu:469-472)
        // (written as count / prefix / ordered scatter over fixed chunks of rows so that the OpenMP build can run the
        // chunks in parallel: within a state, rows keep their order -- the same permutation as three serial sweeps)
        Surfels tmp; tmp.resize(s.n_model);
        std::vector<f3> lab_tmp(s.n_model);
        const int NCH = 64, chunk = (s.n_model + NCH - 1) / NCH;
        int cnt[NCH][3], off[NCH][3];
#pragma omp parallel for schedule(static)
        for (int ch = 0; ch < NCH; ch++) {
            cnt[ch][0] = cnt[ch][1] = cnt[ch][2] = 0;
            for (int i = ch * chunk; i < std::min(s.n_model, (ch + 1) * chunk); i++) cnt[ch][state[i]]++;
        }
        int w = 0;
        for (int pass = 0; pass < 3; pass++)
            for (int ch = 0; ch < NCH; ch++) { off[ch][pass] = w; w += cnt[ch][pass]; }
#pragma omp parallel for schedule(static)
        for (int ch = 0; ch < NCH; ch++)
            for (int i = ch * chunk; i < std::min(s.n_model, (ch + 1) * chunk); i++) {
                const int d = off[ch][state[i]]++;
                tmp.copy_row(d, M, i); lab_tmp[d] = s.model_lab[i];
            }
#pragma omp parallel for schedule(static)
        for (int i = 0; i < s.n_model; i++) { M.copy_row(i, tmp, i); s.model_lab[i] = lab_tmp[i]; }
        s.n_model -= n_removed + n_gone;                                        // :474
        s.n_visible = n_vis;
    } else {
        // first frame: thrust::copy(frame -> model), supersurfel_fusion.cu:477-483
        int k = 0;
        for (int f = 0; f < s.S; f++) {
            if (shard_owner(s, f, s.pose) != c.rank) continue;
            if (k >= c.nb_supersurfels_max) break;
            M.copy_row(k, s.frame, f); s.model_lab[k] = s.frame_lab[f]; k++;
        }
        s.n_model = k; s.n_visible = k;
    }
    if (out) {
        std::memset(out, 0, sizeof(*out));
        for (int i = 0; i < 3; i++) { out->pose[3 * i] = s.pose.R.r[i].x; out->pose[3 * i + 1] = s.pose.R.r[i].y; out->pose[3 * i + 2] = s.pose.R.r[i].z; }
        out->pose[9] = s.pose.t.x; out->pose[10] = s.pose.t.y; out->pose[11] = s.pose.t.z;
        out->icp_valid = s.last_icp_valid; out->icp_iters = s.last_icp_iters;
        out->n_model = s.n_model; out->n_visible = s.n_visible; out->n_removed = n_removed + n_lost;
        out->n_inserted = s.f_inserted; out->n_updated = s.f_updated; out->stamp = s.stamp;
    }
    s.stamp++;                                                                  // :522
    s.global_n_model = -1; s.global_n_visible = -1;
    s.have_frame = false;
}

void fuse(State& s, const uint64_t* best, const uint8_t* matched, ssf_frame_result* out) {
    fuse_begin(s, best, matched, 0, nullptr);
    fuse_end(s, nullptr, out);
}

// rotMatToQuat matrix_math.cuh:529-618, quatToRotMat :512-527 (the wy = q.w*q.z quirk is kept)
ORC_HOST void rot_to_quat(const Mat33& m, float* q /*x,y,z,w*/) {
    float s, tr = (m.r[0].x + m.r[1].y) + m.r[2].z;
    if (tr > 0) {
        s = sqrtf(tr + 1);
        q[3] = 0.5f * s; s = 0.5f / s;
        q[0] = (m.r[2].y - m.r[1].z) * s; q[1] = (m.r[0].z - m.r[2].x) * s; q[2] = (m.r[1].x - m.r[0].y) * s;
    } else {
        int i = 0;
        if (m.r[1].y > m.r[0].x) i = 1;
        if (m.r[2].z > m.r[0].x || m.r[2].z > m.r[1].y) i = 2;
        if (i == 0) {
            s = sqrtf(((1.0f + m.r[0].x) - m.r[1].y) - m.r[2].z);
            q[0] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r[2].y - m.r[1].z) * s; q[1] = (m.r[0].y + m.r[1].x) * s; q[2] = (m.r[0].z + m.r[2].x) * s;
        } else if (i == 1) {
            s = sqrtf(((1.0f + m.r[1].y) - m.r[0].x) - m.r[2].z);
            q[1] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r[0].z - m.r[2].x) * s; q[0] = (m.r[0].y + m.r[1].x) * s; q[2] = (m.r[1].z + m.r[2].y) * s;
        } else {
            s = sqrtf(((1.0f + m.r[2].z) - m.r[0].x) - m.r[1].y);
            q[2] = 0.5f * s; s = 0.5f / s;
            q[3] = (m.r[1].x - m.r[0].y) * s; q[0] = (m.r[0].z + m.r[2].x) * s; q[1] = (m.r[1].z + m.r[2].y) * s;
        }
    }
}
ORC_HOST Mat33 quat_to_rot(const float* q) {
    const float x2 = q[0] * q[0], y2 = q[1] * q[1], z2 = q[2] * q[2];
    const float xy = q[0] * q[1], xz = q[0] * q[2], yz = q[1] * q[2];
    const float wx = q[3] * q[0], wy = q[3] * q[2] /* sic */, wz = q[3] * q[2];
    Mat33 m;
    m.r[0] = mk3(1.0f - 2.0f * (y2 + z2), 2.0f * (xy - wz), 2.0f * (xz + wy));
    m.r[1] = mk3(2.0f * (xy + wz), 1.0f - 2.0f * (x2 + z2), 2.0f * (yz - wx));
    m.r[2] = mk3(2.0f * (xz - wy), 2.0f * (yz + wx), 1.0f - 2.0f * (x2 + y2));
    return m;
}

// ---- re-homing of a sharded map (multi-GPU; no reference counterpart) --------------------------------------------
// applyDeformation moves EVERY row (deformation_graph_kernels.cu:27-73), the per-frame migration only rows a frame has
// updated: after a loop closure a shard holds rows whose tile belongs to another rank until this sweep has run.
// rehome_begin: the valid rows that now belong elsewhere leave, in logical order, as migrant-table records (word 1 = 1
// when the row sat in the visible block); the others close ranks, visible block first.  Returns the number of records,
// or -1 (nothing changed) when `cap` records do not suffice.
int rehome_begin(State& s, int32_t* table, int cap) {
    const ssf_config& c = s.cfg;
    if (c.nranks <= 1) return 0;
    Surfels& M = s.model;
    int n_leave = 0;
    for (int i = 0; i < s.n_model; i++) if (M.conf[i] > 0.0f && owner_of(s, M.pos[i]) != c.rank) n_leave++;
    if (n_leave > cap) return -1;
    int w = 0, k = 0, nv = 0;
    for (int i = 0; i < s.n_model; i++) {
        const int dest = M.conf[i] > 0.0f ? owner_of(s, M.pos[i]) : c.rank;
        if (dest != c.rank) {
            int32_t* rec = &table[(size_t)SSF_MIGRANT_WORDS * w++];
            row_to_slot(M, i, dest, rec);
            rec[1] = i < s.n_visible ? 1 : 0;
        } else {
            if (k != i) { M.copy_row(k, M, i); s.model_lab[k] = s.model_lab[i]; }
            if (i < s.n_visible) nv++;
            k++;
        }
    }
    s.n_model = k; s.n_visible = nv;
    return w;
}
// rehome_end: the records addressed to this rank arrive, in table order: those flagged visible behind the visible
// block, the others behind the out-of-view rows.  A full shard turns the surplus away, in table order (as migrate_in does
// inside a frame): returns their number (0 = every arrival has a row).
int rehome_end(State& s, const int32_t* table, int n) {
    const ssf_config& c = s.cfg;
    int av = 0, ao = 0, turned_away = 0, room = c.nb_supersurfels_max - s.n_model;
    std::vector<char> take(n > 0 ? n : 1, 0);
    for (int j = 0; j < n; j++) {
        const int32_t* w = &table[(size_t)SSF_MIGRANT_WORDS * j];
        if (w[0] - 1 != c.rank) continue;
        if (room <= 0) { turned_away++; continue; }
        room--; take[j] = 1;
        if (w[1]) av++; else ao++;
    }
    if (av + ao == 0) return turned_away;
    Surfels& M = s.model;
    // make room for the visible arrivals between the two blocks: the out-of-view rows move up by av (from the back)
    for (int i = s.n_model - 1; i >= s.n_visible; i--) { M.copy_row(i + av, M, i); s.model_lab[i + av] = s.model_lab[i]; }
    int kv = s.n_visible, ko = s.n_model + av;
    for (int j = 0; j < n; j++) {
        if (!take[j]) continue;
        const int32_t* w = &table[(size_t)SSF_MIGRANT_WORDS * j];
        const int k = w[1] ? kv++ : ko++;
        slot_to_row(M, k, w);
        s.model_lab[k] = rgbToLab(M.col[k]);
    }
    s.n_visible += av; s.n_model += av + ao;
    return turned_away;
}

// applyDeformation, deformation_graph_kernels.cu:27-73
void apply_deformation(State& s, const float* npos, const float* nrot, const float* ntrans, int m,
                       const float* w4, const int32_t* idx4) {
    (void)m;
#pragma omp parallel for schedule(static)
    for (int i = 0; i < s.n_model; i++) {
        const f3 pi = s.model.pos[i];
        f3 po = mk3(0, 0, 0);
        float bq[4] = {0, 0, 0, 0};
        for (int k = 0; k < 4; k++) {
            const int node = idx4[4 * i + k];
            const float wk = w4[4 * i + k];
            const f3 gk = mk3(npos[3 * node], npos[3 * node + 1], npos[3 * node + 2]);
            Mat33 Rk;
            for (int r = 0; r < 3; r++) Rk.r[r] = mk3(nrot[9 * node + 3 * r], nrot[9 * node + 3 * r + 1], nrot[9 * node + 3 * r + 2]);
            const f3 tk = mk3(ntrans[3 * node], ntrans[3 * node + 1], ntrans[3 * node + 2]);
            float qk[4]; rot_to_quat(Rk, qk);
            po = po + wk * ((Rk * (pi - gk) + gk) + tk);
            for (int a = 0; a < 4; a++) bq[a] += wk * qk[a];
        }
        const float len = sqrtf(((bq[0] * bq[0] + bq[1] * bq[1]) + bq[2] * bq[2]) + bq[3] * bq[3]);
        const float inv = 1.0f / len;                                           // operator/=(float4&, float)
        for (int a = 0; a < 4; a++) bq[a] *= inv;
        const Mat33 av = quat_to_rot(bq);
        s.model.orient[i] = s.model.orient[i] * transpose(av);
        s.model.shape[i] = mult_ABAt(av, s.model.shape[i]);
        s.model.pos[i] = po;
    }
}

}  // namespace orc
