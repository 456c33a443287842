// oracle_extract.cpp -- TEST INFRASTRUCTURE (CPU oracle), not product code.
//
// Restatement of the "extract" stage: TPS_RGBD::compute / filter / computeDepthImage
// (core/src/TPS_RGBD.cu:101-525, kernels core/src/TPS_RGBD_kernels.cu and
// core/include/supersurfel_fusion/TPS_RGBD_kernels.cuh) followed by
// SupersurfelFusion::generateSupersurfels (core/src/supersurfel_fusion.cu:551-593, kernels
// core/src/supersurfel_fusion_kernels.cu:48-224).  Schedule decisions: see oracle.h.
#include <algorithm>
#include <cmath>
#include "oracle.h"

namespace orc {

void Surfels::zero(size_t n) {
    for (size_t i = 0; i < n; i++) {
        pos[i] = mk3(0, 0, 0); col[i] = mk3(0, 0, 0);
        stamps[2 * i] = stamps[2 * i + 1] = 0;
        orient[i].r[0] = orient[i].r[1] = orient[i].r[2] = mk3(0, 0, 0);
        shape[i] = mkcov(0, 0, 0, 0, 0, 0);
        dims[2 * i] = dims[2 * i + 1] = 0.f;
        conf[i] = 0.f;
    }
}
void Surfels::copy_row(size_t d, const Surfels& s, size_t i) {
    pos[d] = s.pos[i]; col[d] = s.col[i];
    stamps[2 * d] = s.stamps[2 * i]; stamps[2 * d + 1] = s.stamps[2 * i + 1];
    orient[d] = s.orient[i]; shape[d] = s.shape[i];
    dims[2 * d] = s.dims[2 * i]; dims[2 * d + 1] = s.dims[2 * i + 1];
    conf[d] = s.conf[i];
}

static inline int lab_at(const State& s, const std::vector<int32_t>& lab, int x, int y) {
    // the shared-memory tile of the reference holds -1 outside the image (TPS_RGBD_kernels.cuh:282-288)
    if (x < 0 || x >= s.W || y < 0 || y >= s.H) return -1;
    return lab[(size_t)y * s.W + x];
}

// number of 4-neighbours whose label differs; out-of-image counts as different.  This is the
// invariant the reference maintains incrementally (initSuperpixelsRGBD_kernel,
// TPS_RGBD_kernels.cu:84-100, and the ++/-- at TPS_RGBD_kernels.cuh:400-423); decision A2.
int boundary_at(const State& s, const std::vector<int32_t>& lab, int x, int y) {
    int own = lab[(size_t)y * s.W + x], b = 0;
    if (lab_at(s, lab, x, y - 1) != own) b++;
    if (lab_at(s, lab, x - 1, y) != own) b++;
    if (lab_at(s, lab, x + 1, y) != own) b++;
    if (lab_at(s, lab, x, y + 1) != own) b++;
    return b;
}

// TPS_RGBD_kernels.cuh:178-233: ring walk NW,N,NE,E,SE,S,SW,W without closing W->NW.  ring[k]: label of ring pixel k.
bool ring_unchangeable(int index, const int ring[8]) {
    int jump = 0;
    bool prev = (ring[0] == index);
    for (int k = 1; k < 8; k++) {
        bool cur = (ring[k] == index);
        if (prev != cur) { jump++; prev = cur; }
    }
    return jump > 2;
}
static bool is_unchangeable(const State& s, const std::vector<int32_t>& lab, int x, int y) {
    const int ox[8] = {-1, 0, 1, 1, 1, 0, -1, -1}, oy[8] = {-1, -1, -1, 0, 1, 1, 1, 0};
    int ring[8];
    for (int k = 0; k < 8; k++) ring[k] = lab_at(s, lab, x + ox[k], y + oy[k]);
    return ring_unchangeable(lab_at(s, lab, x, y), ring);
}

// the 9 inlier-only disparity sums of one pixel (TPS_RGBD_kernels.cu:142-150, .cuh:445-466)
static inline void disp_terms(SpSums& c, int x, int y, float d, int sign) {
    c.dx += sign * x; c.dy += sign * y; c.dn += sign;
    c.dxx += (int64_t)sign * x * x; c.dyy += (int64_t)sign * y * y; c.dxy += (int64_t)sign * x * y;
    c.dxd += sign * fx_quant((double)((float)x * d), DISP_SCALE, DISP_LIM);
    c.dyd += sign * fx_quant((double)((float)y * d), DISP_SCALE, DISP_LIM);
    c.dd += sign * fx_quant((double)d, DISP_SCALE, DISP_LIM);
}

static inline void sums_add(SpSums& a, const SpSums& d) {
    a.sx += d.sx; a.sy += d.sy; a.sr += d.sr; a.sg += d.sg; a.sb += d.sb; a.n += d.n;
    a.dx += d.dx; a.dy += d.dy; a.dn += d.dn; a.dxx += d.dxx; a.dyy += d.dyy; a.dxy += d.dxy;
    a.dxd += d.dxd; a.dyd += d.dyd; a.dd += d.dd;
}

// mergeTPSRGBCoeffs_kernel, TPS_RGBD_kernels.cu:224-242
static void merge_rgb(State& s) {
#pragma omp parallel for schedule(static)
    for (int k = 0; k < s.S; k++) {
        const SpSums& c = s.sums[k]; Superpixel& sp = s.sp[k];
        float n = (float)c.n;
        sp.cx = fdiv((float)c.sx, n); sp.cy = fdiv((float)c.sy, n);
        sp.r = fdiv((float)c.sr, n); sp.g = fdiv((float)c.sg, n); sp.b = fdiv((float)c.sb, n);
        sp.size = n;
    }
}
// mergeTPSRGBDCoeffs_kernel, TPS_RGBD_kernels.cu:244-276
static void merge_rgbd(State& s) {
    merge_rgb(s);
    const double inv = 1.0 / DISP_SCALE;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < s.S; k++) {
        const SpSums& c = s.sums[k]; Superpixel& sp = s.sp[k];
        float dx = (float)c.dx, dy = (float)c.dy, dn = (float)c.dn;
        float dxx = (float)c.dxx, dyy = (float)c.dyy, dxy = (float)c.dxy;
        float dxd = (float)((double)c.dxd * inv), dyd = (float)((double)c.dyd * inv);
        float dd = (float)((double)c.dd * inv);
        float tx, ty, tz;
        if (!solvePlaneEquations(tx, ty, tz, dxx, dxy, dx, dxd, dxy, dyy, dy, dyd, dx, dy, dn, dd)) {
            tx = 0.f; ty = 0.f;
            uint32_t nanbits = 0xFFE00000u; std::memcpy(&tz, &nanbits, 4);
        }
        sp.ta = tx; sp.tb = ty; sp.tc = tz;
    }
}

#ifdef SSF_ORACLE_ARMS
static void update_pass_spec(State& s, int OX, int OY, bool rgbd);
// STUDY ARM g_arms.schedule = 1 (oracle_math.h): the same pass under the OTHER extreme schedule the reference's kernel admits --
// its 32 x 32-pixel blocks (16 x 16 threads, TPS_RGBD_kernels.cuh:264-292) run one after the other in raster order, each
// snapshotting its 34 x 34 label tile AFTER every earlier block has stored its labels (the specification: before any block
// stores).  Within a block all reads see the snapshot (the LDS tile), as on the GPU.  Sums, inlier bytes and the exact boundary
// recount follow the labels exactly as in the specification.  Implemented by running the specification's pass restricted to one
// block at a time on the evolving map.
static int g_tile_x0 = 0, g_tile_y0 = 0, g_tile_on = 0;
static void update_pass(State& s, int OX, int OY, bool rgbd) {
    if (g_arms.schedule == 0) { g_tile_on = 0; update_pass_spec(s, OX, OY, rgbd); return; }
    g_tile_on = 1;
    for (int ty = 0; ty < s.H; ty += 32)
        for (int tx = 0; tx < s.W; tx += 32) { g_tile_x0 = tx; g_tile_y0 = ty; update_pass_spec(s, OX, OY, rgbd); }
    g_tile_on = 0;
}
#define update_pass update_pass_spec
#endif
// updateTPSRGB_kernel / updateTPSRGBD_kernel, TPS_RGBD_kernels.cuh:235-651, one pass (OX,OY).
static void update_pass(State& s, int OX, int OY, bool rgbd) {
    const ssf_config& c = s.cfg;
    const int W = s.W, H = s.H;
    const int min_size = (int)((float)(c.cell_size * c.cell_size) / 4.f);     // TPS_RGBD.cu:198 -> int param
    const std::vector<int32_t>& src = s.label;
    std::vector<int32_t>& dst = s.label_tmp;
#ifdef SSF_ORACLE_ARMS
    std::vector<std::pair<size_t, int32_t>> tile_changes;                       // (schedule arm: this block's stores, applied when it ends)
    if (!g_tile_on)
#endif
    dst = src;                                                                  // decision A1
    const int nx[4] = {0, -1, 1, 0}, ny[4] = {-1, 0, 0, 1};                    // neighbors[8], .cuh:350
    // Rows of a pass are independent (all reads are pre-pass values, decision A1) and the sums are exact integers, so
    // the OpenMP build (libssf_oracle_omp.so, the timed CPU baseline) splits the rows over threads, each with its own
    // delta table, and adds the tables up afterwards: bit-identical to the serial order.
    int n_rows = (H - OY + 1) / 2, row0 = 0, col0 = 0, col1 = (W + 1) / 2;
#ifdef SSF_ORACLE_ARMS
    if (g_tile_on) { row0 = g_tile_y0 / 2; n_rows = std::min(n_rows, g_tile_y0 / 2 + 16); col0 = g_tile_x0 / 2; col1 = std::min(col1, g_tile_x0 / 2 + 16); }
#endif
#pragma omp parallel
    {
    std::vector<SpSums> delta(s.S, SpSums{});
#pragma omp for schedule(static)
    for (int raw_y = row0; raw_y < n_rows; raw_y++) {
        const int y = 2 * raw_y + OY;
        for (int raw_x = col0; raw_x < col1; raw_x++) {
            const int x = 2 * raw_x + ((raw_x + OX) & 1);                       // .cuh:264
            if (x >= W) continue;
#ifdef SSF_ORACLE_ARMS
            if (g_tile_on && !(x >= g_tile_x0 && x < g_tile_x0 + 32 && y >= g_tile_y0 && y < g_tile_y0 + 32)) continue;
#endif
            const size_t p = (size_t)y * W + x;
            const int index = src[p];
            int new_index = index;
            const Superpixel prev_sp = s.sp[index];
            const int bounds = boundary_at(s, src, x, y);

            float disp = 0.f, disp_energy = 0.f;
            uint8_t prev_inlier = 0, inlier = 0xff;
            if (rgbd) {                                                         // .cuh:309-321
                disp = s.disp[p];
                prev_inlier = s.inlier[p];
                float dp = (prev_sp.ta * (float)x + prev_sp.tb * (float)y) + prev_sp.tc;
                disp_energy = (dp - disp) * (dp - disp);
                if (!std::isfinite(disp_energy) || disp_energy > c.thresh_disp || dp < 0.f) {
                    disp_energy = c.thresh_disp; inlier = 0;
                }
            }
            const uint32_t px = s.rgba[p];
            const float cr = (float)(px & 255u), cg = (float)((px >> 8) & 255u), cb = (float)((px >> 16) & 255u);

            if (bounds && !is_unchangeable(s, src, x, y)) {
                const float posx = (float)x, posy = (float)y;
                const float size = prev_sp.size;
                const float sc = fdiv(size, size - 1.f);                           // .cuh:332
                const float dpx = sc * (posx - prev_sp.cx), dpy = sc * (posy - prev_sp.cy);
                const f3 dcol = mk3(sc * (cr - prev_sp.r), sc * (cg - prev_sp.g), sc * (cb - prev_sp.b));
                const float dsize = size - (float)min_size;
                float best = dot(dcol, dcol) + c.lambda_pos * (dpx * dpx + dpy * dpy);
                if (rgbd) best = best + c.lambda_disp * disp_energy;
                best = best - c.lambda_size * fminf(dsize, 0.f);
                best = best + c.lambda_bound * (float)bounds;

                for (int k = 0; k < 4; k++) {
                    const int i_n = lab_at(s, src, x + nx[k], y + ny[k]);
                    if (i_n == -1 || i_n == index) continue;
                    const Superpixel n_sp = s.sp[i_n];
                    const float ndx = posx - n_sp.cx, ndy = posy - n_sp.cy;
                    const f3 ndc = mk3(cr - n_sp.r, cg - n_sp.g, cb - n_sp.b);
                    const float ndsize = (n_sp.size + 1.f) - (float)min_size;
                    float n_de = 0.f; uint8_t n_inlier = 0xff;
                    if (rgbd) {                                                 // .cuh:365-373
                        float dp = (n_sp.ta * (float)x + n_sp.tb * (float)y) + n_sp.tc;
                        n_de = (dp - disp) * (dp - disp);
                        if (!std::isfinite(n_de) || n_de > c.thresh_disp || dp < 0.f) {
                            n_de = c.thresh_disp; n_inlier = 0;
                        }
                    }
                    int b = 0;
                    for (int q = 0; q < 4; q++)
                        if (lab_at(s, src, x + nx[q], y + ny[q]) != i_n) b++;
                    float e = dot(ndc, ndc) + c.lambda_pos * (ndx * ndx + ndy * ndy);
                    if (rgbd) e = e + c.lambda_disp * n_de;
                    e = e - c.lambda_size * fminf(ndsize, 0.f);
                    e = e + c.lambda_bound * (float)b;
                    if (e < best) { best = e; new_index = i_n; if (rgbd) inlier = n_inlier; }
                }
                if (new_index != index) {                                       // .cuh:400-440
#ifdef SSF_ORACLE_ARMS
                    if (g_tile_on) tile_changes.push_back(std::make_pair(p, (int32_t)new_index)); else
#endif
                    dst[p] = new_index;
                    SpSums& a = delta[index]; SpSums& bsum = delta[new_index];
                    const int ir = (int)(px & 255u), ig = (int)((px >> 8) & 255u), ib = (int)((px >> 16) & 255u);
                    a.sx -= x; a.sy -= y; a.sr -= ir; a.sg -= ig; a.sb -= ib; a.n -= 1;
                    bsum.sx += x; bsum.sy += y; bsum.sr += ir; bsum.sg += ig; bsum.sb += ib; bsum.n += 1;
                }
            }
            if (rgbd) {                                                         // .cuh:443-472
                if (inlier && (!prev_inlier || index != new_index)) disp_terms(delta[new_index], x, y, disp, +1);
                if (prev_inlier && (!inlier || (inlier && index != new_index))) disp_terms(delta[index], x, y, disp, -1);
                if (inlier != prev_inlier) s.inlier[p] = inlier;
            }
        }
    }
#pragma omp critical
    for (int k = 0; k < s.S; k++) sums_add(s.sums[k], delta[k]);
    }
#ifdef SSF_ORACLE_ARMS
    if (g_tile_on) { for (const auto& ch : tile_changes) s.label[ch.first] = ch.second; return; }
#endif
    s.label.swap(s.label_tmp);
}
#ifdef SSF_ORACLE_ARMS
#undef update_pass
#endif

static inline int clampi(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }
// point-sampled clamp texture fetch with float coordinates (texture_impl.hpp:43-46)
static inline size_t tex_idx(const State& s, float x, float y) {
    int ix = clampi((int)floorf(x), 0, s.W - 1), iy = clampi((int)floorf(y), 0, s.H - 1);
    return (size_t)iy * s.W + ix;
}

// initSamples_kernel, TPS_RGBD_kernels.cu:324-401
static void init_samples(State& s) {
    const int nbWalks = 10;
    const float radius = (float)s.cfg.cell_size / 2.f;                          // TPS_RGBD.cu:288
    const float wdx[4] = {-1.f, 0.f, 1.f, 0.f}, wdy[4] = {0.f, -1.f, 0.f, 1.f};
    const int ns = s.cfg.nb_samples;
#pragma omp parallel for schedule(static)
    for (int index = 0; index < s.S; index++)
        for (int t = 0; t < ns; t++) {
            const uint32_t idx = (uint32_t)(index * ns + t);
            uint32_t ctr = s.extract_ordinal * 64u;                              // epoch of this frame (oracle.h)
            const float cx = s.sp[index].cx, cy = s.sp[index].cy;
            float x = cx, y = cy;
            int i = s.label[tex_idx(s, x, y)];
            int k = 0;
            while (i != index && k++ < 10) {                                    // :345-350 (double arithmetic)
                float u1 = rng_uniform(rng_u32(s.cfg.rng_seed, idx, ctr));
                x = (float)((double)cx + ((double)radius * 2.) * (double)(u1 - 1.f));
                float u2 = rng_uniform(rng_u32(s.cfg.rng_seed, idx, ctr));
                y = (float)((double)cy + ((double)radius * 2.) * (double)(u2 - 1.f));
                i = s.label[tex_idx(s, x, y)];
            }
            float d = s.disp[tex_idx(s, x, y)];
            f3 xyd[3]; xyd[0] = xyd[1] = xyd[2] = mk3(x, y, d);
            for (int j = 0; j < 3; j++)
                for (int w = 0; w < nbWalks; w++) {
                    int dir = (int)(rng_u32(s.cfg.rng_seed, idx, ctr) & 3u);
                    float next_x = x + wdx[dir], next_y = y + wdy[dir];
                    i = s.label[tex_idx(s, x, y)];                              // label at the OLD position (:370)
                    if (i == index && next_x >= 0 && next_x < (float)s.W && next_y >= 0 && next_y < (float)s.H) {
                        x = next_x; y = next_y;
                        float dd = s.disp[tex_idx(s, x, y)];
                        if (std::isfinite(dd)) xyd[j] = mk3(x, y, dd);
                    }
                }
            float a, b, cc;
            if (!solvePlaneEquations(a, b, cc, xyd[0].x, xyd[0].y, 1.f, xyd[0].z, xyd[1].x, xyd[1].y, 1.f,
                                     xyd[1].z, xyd[2].x, xyd[2].y, 1.f, xyd[2].z)) {
                a = 0.f; b = 0.f; cc = xyd[2].z;
            }
            float* smp = &s.samples[4 * (size_t)idx];
            smp[0] = a; smp[1] = b; smp[2] = cc; smp[3] = 0.f;
        }
}

// evalSamples_kernel, TPS_RGBD_kernels.cu:403-433 (scores are integer counts)
static void eval_samples(State& s) {
    const int ns = s.cfg.nb_samples;
    // scores are counts (the reference adds 1.f per hit: exact in float below 2^24), kept as integers per thread
    std::vector<int32_t> score((size_t)s.S * ns, 0);
#pragma omp parallel
    {
    std::vector<int32_t> local((size_t)s.S * ns, 0);
#pragma omp for schedule(static)
    for (int y = 0; y < s.H; y++)
        for (int x = 0; x < s.W; x++) {
            size_t p = (size_t)y * s.W + x;
            int index = s.label[p]; float d = s.disp[p];
            for (int k = 0; k < ns; k++) {
                const float* th = &s.samples[4 * ((size_t)index * ns + k)];
                if (std::isfinite(th[2])) {
                    float dp = (th[0] * (float)x + th[1] * (float)y) + th[2];
                    float dd = (d - dp) * (d - dp);
                    if (dd < s.cfg.thresh_disp) local[(size_t)index * ns + k] += 1;
                }
            }
        }
#pragma omp critical
    for (size_t i = 0; i < score.size(); i++) score[i] += local[i];
    }
    for (size_t i = 0; i < score.size(); i++) s.samples[4 * i + 3] += (float)score[i];
}

// selectSamples_kernel, TPS_RGBD_kernels.cu:435-467
static void select_samples(State& s) {
    const int ns = s.cfg.nb_samples;
    for (int idx = 0; idx < s.S; idx++) {
        float best[4] = {0.f, 0.f, 0.f, 0.f};
        for (int k = 0; k < ns; k++) {
            const float* th = &s.samples[4 * ((size_t)idx * ns + k)];
            if (th[3] > best[3]) { best[0] = th[0]; best[1] = th[1]; best[2] = th[2]; best[3] = th[3]; }
        }
        s.sp[idx].ta = best[0]; s.sp[idx].tb = best[1]; s.sp[idx].tc = best[2];
        SpSums& c = s.sums[idx];
        c.dx = c.dy = c.dn = 0; c.dxx = c.dyy = c.dxy = 0; c.dxd = c.dyd = c.dd = 0;
    }
}

// initDispCoeffsRansacRGBD_kernel (:112-155) / initDispCoeffsRGBD_kernel (:157-190)
static void init_disp_coeffs(State& s, bool ransac) {
#pragma omp parallel
    {
    std::vector<SpSums> delta(s.S, SpSums{});
#pragma omp for schedule(static)
    for (int y = 0; y < s.H; y++)
        for (int x = 0; x < s.W; x++) {
            size_t p = (size_t)y * s.W + x;
            int index = s.label[p]; float d = s.disp[p];
            uint8_t inl = 0;
            if (std::isfinite(d)) {
                if (ransac) {
                    const Superpixel& sp = s.sp[index];
                    float dp = (sp.ta * (float)x + sp.tb * (float)y) + sp.tc;
                    float dd = (dp - d) * (dp - d);
                    if (std::isfinite(dd) && dd < s.cfg.thresh_disp && dp > 0.f) inl = 0xff;
                } else inl = 0xff;
                if (inl) disp_terms(delta[index], x, y, d, +1);
            }
            s.inlier[p] = inl;
        }
#pragma omp critical
    for (int k = 0; k < s.S; k++) sums_add(s.sums[k], delta[k]);
    }
}

// TPS_RGBD::filter, TPS_RGBD.cu:480-505; kernels TPS_RGBD_kernels.cu:510-614.  Jacobi sweeps (A9).
static void plane_filter(State& s) {
    const int S = s.S, gx = s.gx, gy = s.gy;
    std::vector<f3> X(S), Z(S), Xn(S);
    std::vector<float> px(S), py(S);
    for (int i = 0; i < S; i++) {                                               // initFilter_kernel
        const Superpixel& sp = s.sp[i];
        X[i] = mk3((sp.cx * sp.ta + sp.cy * sp.tb) + sp.tc, sp.ta, sp.tb);
        Z[i] = X[i]; px[i] = sp.cx; py[i] = sp.cy;
    }
    const float alpha = s.cfg.filter_alpha, beta = s.cfg.filter_beta, thr = s.cfg.filter_threshold;
    const int v[4] = {-1, 0, 0, 1}, u[4] = {0, -1, 1, 0};
#ifdef SSF_ORACLE_ARMS
    const bool gauss_seidel = g_arms.filter_gs != 0;        // STUDY ARM: the reference updates data[idx].X in place (:585-612)
#else
    const bool gauss_seidel = false;
#endif
    for (int it = 0; it < s.cfg.filter_iter; it++) {                            // iterateFilter_kernel
        Xn = X;
        for (int y = 0; y < gy; y++)
            for (int x = 0; x < gx; x++) {
                const int idx = y * gx + x;
                Cov3 A = mkcov(alpha, 0.f, 0.f, alpha, 0.f, alpha);
                f3 R = alpha * Z[idx];
                for (int j = 0; j < 4; j++) {
                    int yy = y + v[j], xx = x + u[j];
                    if (yy >= 0 && yy < gy && xx >= 0 && x < gx) {              // sic: x<gridSizeX (:583)
                        int nidx = yy * gx + xx;
                        if (nidx >= S) continue;                                // guard for the UB read
                        f3 Xj = X[nidx];
                        float dx = px[idx] - px[nidx], dy = py[idx] - py[nidx];
                        float dz = X[idx].x - Xj.x;
                        if (std::isfinite(dz) && dz * dz < thr * thr) {
                            A.xx += beta * 2.f;
                            A.xy += -beta * dx;
                            A.xz += -beta * dy;
                            A.yy += beta * (2.f + dx * dx);
                            A.yz += beta * (dx * dy);
                            A.zz += beta * (2.f + dy * dy);
                            R.x += beta * ((2.f * Xj.x + dx * Xj.y) + dy * Xj.z);
                            R.y += beta * (-dx * Xj.x + 2.f * Xj.y);
                            R.z += beta * (-dy * Xj.x + 2.f * Xj.z);
                        }
                    }
                }
                Cov3 A1;
                if (inverse(A, A1)) { Xn[idx] = A1 * R; if (gauss_seidel) X[idx] = Xn[idx]; }
            }
        X.swap(Xn);
    }
    for (int i = 0; i < S; i++) {                                               // finishFilter_kernel
        Superpixel& sp = s.sp[i];
        sp.ta = X[i].y; sp.tb = X[i].z;
        sp.tc = (X[i].x - sp.cx * X[i].y) - sp.cy * X[i].z;
    }
}

// renderDepthImage_kernel, TPS_RGBD_kernels.cu:469-508 (inlier mask ignored: `if(true || inlier)`)
static void render_depth(State& s) {
#pragma omp parallel for schedule(static)
    for (int y = 0; y < s.H; y++)
        for (int x = 0; x < s.W; x++) {
            size_t p = (size_t)y * s.W + x;
            const Superpixel& sp = s.sp[s.label[p]];
            float disp = ((float)x * sp.ta + (float)y * sp.tb) + sp.tc;
            s.plane_depth[p] = fdiv(1.f, disp);
        }
}

// generateSupersurfels, supersurfel_fusion.cu:551-593
static void generate_supersurfels(State& s) {
    const ssf_config& c = s.cfg;
    const int S = s.S;
    s.frame.zero(S);
    std::vector<int64_t> acc((size_t)S * 13, 0);                                // p(3) lab(3) pp(6) count
    // computeSupersurfelCoeffs, supersurfel_fusion_kernels.cu:113-167
#pragma omp parallel
    {
    std::vector<int64_t> lacc((size_t)S * 13, 0);
#pragma omp for schedule(static)
    for (int y = 0; y < s.H; y++)
        for (int x = 0; x < s.W; x++) {
            size_t p = (size_t)y * s.W + x;
            if (!s.inlier[p]) continue;
            int index = s.label[p];
            int bound = boundary_at(s, s.label, x, y);
            float depth = s.plane_depth[p];
            if (std::isfinite(depth) && depth > 0.0f && bound == 0) {
                f3 pos = mk3(fdiv(((float)x - c.cx) * depth, c.fx), fdiv(((float)y - c.cy) * depth, c.fy), depth);
                uint32_t px = s.rgba[p];
                f3 lab = rgbToLab(mk3((float)(px & 255u), (float)((px >> 8) & 255u), (float)((px >> 16) & 255u)));
                Cov3 cov = outer(pos);
                int64_t* a = &lacc[(size_t)index * 13];
                const float v[12] = {pos.x, pos.y, pos.z, lab.x, lab.y, lab.z, cov.xx, cov.xy, cov.xz, cov.yy, cov.yz, cov.zz};
                for (int k = 0; k < 12; k++) a[k] += fx_quant((double)v[k], MOM_SCALE, MOM_LIM);
                a[12] += 1;
            }
        }
#pragma omp critical
    for (size_t i = 0; i < acc.size(); i++) acc[i] += lacc[i];
    }
    // computeSupersurfels, supersurfel_fusion_kernels.cu:169-224
    const double inv = 1.0 / MOM_SCALE;
#pragma omp parallel for schedule(static)
    for (int k = 0; k < S; k++) {
        const int64_t* a = &acc[(size_t)k * 13];
        float sum[12];
        for (int j = 0; j < 12; j++) sum[j] = (float)((double)a[j] * inv);
        const float conf = (float)a[12];
        Surfels& f = s.frame;
        f.pos[k] = mk3(sum[0], sum[1], sum[2]);
        f.col[k] = mk3(sum[3], sum[4], sum[5]);
        f.shape[k] = mkcov(sum[6], sum[7], sum[8], sum[9], sum[10], sum[11]);
        f.conf[k] = conf;
        float z = fdiv(f.pos[k].z, conf);
        if (std::isfinite(z) && conf > 100.0f && z > c.range_min && z < c.range_max) {
            f.pos[k] = mk3(fdiv(f.pos[k].x, conf), fdiv(f.pos[k].y, conf), z);
            f.col[k] = labToRgb(mk3(fdiv(f.col[k].x, conf), fdiv(f.col[k].y, conf), fdiv(f.col[k].z, conf)));
            f.shape[k] = f.shape[k] / conf - outer(f.pos[k]);
            f3 vals; Mat33 vecs;
            eigenDecomposition(f.shape[k], vecs, vals, 10);
            f.orient[k] = vecs;
            f.dims[2 * k] = vals.x; f.dims[2 * k + 1] = vals.y;
            f.stamps[2 * k] = s.stamp; f.stamps[2 * k + 1] = s.stamp;
            if (fdiv(vals.x, vals.y) > 50.0f) f.conf[k] = -1.0f;
        } else
            f.conf[k] = -1.0f;
    }
}

// Depth pre-filter: cv::cuda::bilateralFilter(depth, depth, -1, 0.03, 4.5), supersurfel_fusion.cu:180.
// OpenCV (3.4, cudaimgproc/src/cuda/bilateral_filter.cu) is a third-party dependency that is not
// vendored by the reference: its published algorithm is restated here -- radius = round(1.5 sigma_s),
// taps inside the circle of that radius in row-major order, weight = exp(-d2/(2 sigma_s^2) -
// dv^2/(2 sigma_c^2)), BORDER_REFLECT_101, result = sum(w v)/sum(w) -- out of place (the reference's
// in-place call races with itself).  PARITY UNPINNED (no OpenCV here, no reference vector).
static inline int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}
void bilateral_filter(const float* in, float* out, int W, int H, float sigma_color, float sigma_space) {
    int radius = (int)std::lrint((double)sigma_space * 1.5);
    if (radius < 1) radius = 1;
    const float r2 = (float)(radius * radius);
    const float ss = -0.5f / (sigma_space * sigma_space), sc = -0.5f / (sigma_color * sigma_color);
#pragma omp parallel for schedule(static)
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            const float center = in[(size_t)y * W + x];
            float sum1 = 0.f, sum2 = 0.f;
            for (int cy = y - radius; cy <= y + radius; cy++)
                for (int cx = x - radius; cx <= x + radius; cx++) {
                    const float space2 = (float)((x - cx) * (x - cx) + (y - cy) * (y - cy));
                    if (space2 > r2) continue;
                    const float v = in[(size_t)reflect101(cy, H) * W + reflect101(cx, W)];
                    const float dv = fabsf(v - center);
                    const float w = spec_exp_neg(space2 * ss + (dv * dv) * sc);
                    sum1 = sum1 + w * v;
                    sum2 = sum2 + w;
                }
            out[(size_t)y * W + x] = fdiv(sum1, sum2);
        }
}

void extract(State& s, const uint8_t* rgb, const float* depth, const uint8_t* dynamic_mask) {
    struct Epoch { State& st; ~Epoch() { st.extract_ordinal++; } } epoch_guard{s};   // next frame, next RNG epoch
    const ssf_config& c = s.cfg;
    const int W = s.W, H = s.H;
    std::vector<float> filtered;
    if (c.depth_prefilter) {                                                    // supersurfel_fusion.cu:180
        filtered.resize((size_t)W * H);
        bilateral_filter(depth, filtered.data(), W, H, c.prefilter_sigma_color, c.prefilter_sigma_space);
        depth = filtered.data();
    }
    // ingest: cvtColor BGR2BGRA keeps channel order (TPS_RGBD.cu:136), depth2disp32F_kernel
    // (TPS_RGBD_kernels.cu:278-296), initSuperpixelsRGBD_kernel (:61-110)
    for (int k = 0; k < s.S; k++) { SpSums z{}; s.sums[k] = z; Superpixel zs{}; s.sp[k] = zs; }
    for (int y = 0; y < H; y++)
        for (int x = 0; x < W; x++) {
            size_t p = (size_t)y * W + x;
            uint32_t r = rgb[3 * p], g = rgb[3 * p + 1], b = rgb[3 * p + 2];
            s.rgba[p] = r | (g << 8) | (b << 16) | (255u << 24);
            s.disp[p] = fdiv(1.f, depth[p]);
            int index = s.gx * (y / c.cell_size) + x / c.cell_size;
            s.label[p] = index;
            s.inlier[p] = 0;
            SpSums& a = s.sums[index];
            a.sx += x; a.sy += y; a.sr += (int)r; a.sg += (int)g; a.sb += (int)b; a.n += 1;
        }
    merge_rgb(s);
    int passes = 0;
    const int limit = s.max_passes > 0 ? s.max_passes : (1 << 30);
    const int ox[4] = {0, 1, 0, 1}, oy[4] = {0, 1, 1, 0};                       // TPS_RGBD.cu:190-268
    for (int k = 0; k < c.seg_iter / 2; k++)
        for (int q = 0; q < 4; q++) {
            if (passes >= limit) break;
            update_pass(s, ox[q], oy[q], false); merge_rgb(s); passes++;
        }
    if (c.seg_use_ransac) {                                                     // TPS_RGBD.cu:279-321
        init_samples(s); eval_samples(s); select_samples(s); init_disp_coeffs(s, true);
    } else
        init_disp_coeffs(s, false);
    merge_rgbd(s);
    for (int k = c.seg_iter / 2; k < c.seg_iter; k++)
        for (int q = 0; q < 4; q++) {
            if (passes >= limit) break;
            update_pass(s, ox[q], oy[q], true); merge_rgbd(s); passes++;
        }
    plane_filter(s);
    render_depth(s);
    generate_supersurfels(s);
    if (dynamic_mask)                                                           // MOD hook, motion_detection.cu:573-578
        for (int k = 0; k < s.S; k++) if (dynamic_mask[k]) s.frame.conf[k] = -1.0f;
    s.have_frame = true;
}

}  // namespace orc
