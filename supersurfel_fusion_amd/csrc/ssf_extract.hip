// ssf_extract.hip -- "extract" stage for gfx950: RGB-D frame -> label map, slanted planes,
// plane-rendered depth and the S frame supersurfels.
//
// What is computed follows the reference (TPS_RGBD::compute/filter/computeDepthImage,
// core/src/TPS_RGBD.cu:101-525; generateSupersurfels, core/src/supersurfel_fusion.cu:551-593);
// how is this build's own:
//   * 64-wide wavefronts; every per-superpixel sum is an exact integer (int32 / int64 / fixed-point
//     int64), accumulated per 32x32 tile in LDS (integer ds_add into the tile's window of grid cells)
//     and flushed with ONE global integer atomic per non-zero (tile, superpixel, term) -- no float
//     atomics, bit-reproducible for any launch geometry.
//   * a relabelling pass snapshots its tile + 1-pixel halo of the label map in LDS and updates the map
//     in place (the tile grid is shifted so that a workgroup is the only writer of what it read); the
//     boundary count is derived from the tile, never stored; no merge launch between passes (double
//     buffered sums + replayed per-tile log, see k_update_pass).
//   * the plane filter runs the final merge and all its Jacobi sweeps in one workgroup per frame.
//   * every kernel covers all frames of an extract batch (batch index = blockIdx.z / .y, batch_slot).
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <algorithm>
#include <mutex>
#include "ssf_device.hpp"

namespace ssf {

#define HIP_CHECK_LAUNCH() (void)hipGetLastError()

// ---- wave64 helpers ----------------------------------------------------------------------------
__device__ __forceinline__ int lane_id() { return threadIdx.x & 63; }
__device__ __forceinline__ void atomic_add_i64(long long* p, long long v) {
    atomicAdd(reinterpret_cast<unsigned long long*>(p), static_cast<unsigned long long>(v));
}
// ---- ingest ------------------------------------------------------------------------------------
// One workgroup per grid cell: RGB->RGBA, disparity = 1/depth, label = cell id, and the cell's
// initial sums by an in-block reduction (depth2disp32F_kernel TPS_RGBD_kernels.cu:278-296,
// initSuperpixelsRGBD_kernel :61-110).
__global__ __launch_bounds__(256) void k_ingest(SegParams p, BatchIn in, FrameMaps m, uint32_t epoch0) {
    const uint8_t* __restrict__ rgb = batch_pick(in.rgb, (int)blockIdx.y);
    const float* __restrict__ depth = batch_pick(in.depth, (int)blockIdx.y);
    m = batch_slot(m, blockIdx.y);
    const int cell = blockIdx.x;
    if (cell == 0 && threadIdx.x == 0) m.epoch[0] = epoch0 + blockIdx.y;
#ifdef SSF_EXPERIMENTS
    if (cell == 0 && threadIdx.x >= 1 && threadIdx.x < 64) m.epoch[threadIdx.x] = 0u;       // the frame's relabelling statistics (SSF_PASS_STAT_*, ssf_dbg_pass_stats)
#endif
    const int cx0 = (cell % p.gx) * p.cell, cy0 = (cell / p.gx) * p.cell;
    const int w = min(p.cell, p.W - cx0), h = min(p.cell, p.H - cy0);
    int sx = 0, sy = 0, sr = 0, sg = 0, sb = 0, n = 0;
    for (int i = threadIdx.x; i < w * h; i += blockDim.x) {
        const int x = cx0 + i % w, y = cy0 + i / w;
        const size_t q = (size_t)y * p.W + x;
        const uint32_t r = rgb[3 * q], g = rgb[3 * q + 1], b = rgb[3 * q + 2];
        m.rgba[q] = r | (g << 8) | (b << 16) | (255u << 24);
        m.disp[q] = 1.f / depth[q];
        m.label[q] = cell;
        m.inlier[q] = 0;
        sx += x; sy += y; sr += (int)r; sg += (int)g; sb += (int)b; n += 1;
    }
    // block sums through replicated LDS counters (lane & 15): 6 integer ds_add per thread instead of 36 cross-lane
    // shuffles (which also go through the LDS pipe)
    __shared__ int acc[16][8];
    if (threadIdx.x < 128) acc[threadIdx.x >> 3][threadIdx.x & 7] = 0;
    __syncthreads();
    {
        int* a = acc[lane_id() & 15];
        atomicAdd(&a[0], sx); atomicAdd(&a[1], sy); atomicAdd(&a[2], sr); atomicAdd(&a[3], sg); atomicAdd(&a[4], sb); atomicAdd(&a[5], n);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        int t[6];
        for (int j = 0; j < 6; j++) { int v = 0; for (int r = 0; r < 16; r++) v += acc[r][j]; t[j] = v; }
        SumRec rec;                                  // whole-record stores (16-byte pieces), both buffers
        rec.sx = t[0]; rec.sy = t[1]; rec.sr = t[2]; rec.sg = t[3]; rec.sb = t[4]; rec.n = t[5];
        rec.dx = 0; rec.dy = 0; rec.dn = 0;
        rec.stamp = SSF_STAMP_NEVER;
        for (int j = 0; j < 6; j++) rec.pad0[j] = 0;
        rec.dxx = 0; rec.dyy = 0; rec.dxy = 0; rec.dxd = 0; rec.dyd = 0; rec.dd = 0; rec.pad1[0] = 0; rec.pad1[1] = 0;
        m.sums[0].r[cell] = rec; m.sums[1].r[cell] = rec;
        SpRow z = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
        m.sp[cell] = z;
    }
}

// means (+ plane) of one superpixel from its exact sums: mergeTPSRGBCoeffs_kernel /
// mergeTPSRGBDCoeffs_kernel, TPS_RGBD_kernels.cu:224-276.  `prev` supplies the plane that an RGB-only
// merge leaves untouched.
// Loads for the rare "outside the LDS window" paths.  Written as nontemporal loads so that the compiler cannot fold
// "LDS copy if inside the window, else global" into ONE load through a selected address -- that is a FLAT load, slower
// than either and tied to both wait counters.
__device__ __forceinline__ float4 ld_global_f4(const float4* p) {
    const float* q = reinterpret_cast<const float*>(p);
    return make_float4(__builtin_nontemporal_load(q), __builtin_nontemporal_load(q + 1), __builtin_nontemporal_load(q + 2),
                       __builtin_nontemporal_load(q + 3));
}
__device__ __forceinline__ SpRow ld_global_row(const SpRow* p) {
    const float4* q = reinterpret_cast<const float4*>(p);
    const float4 a = ld_global_f4(q), b = ld_global_f4(q + 1), c = ld_global_f4(q + 2);
    SpRow r; r.cx = a.x; r.cy = a.y; r.r = a.z; r.g = a.w; r.b = b.x; r.ta = b.y; r.tb = b.z; r.tc = b.w;
    r.size = c.x; r.pad0 = c.y; r.pad1 = c.z; r.pad2 = c.w;
    return r;
}
// A load that must see what ANOTHER compute unit of this XCD stored earlier in the same launch (k_passes_team: a pass reads what
// the pass before wrote): past the compute unit's vector L1, served by the XCD's L2 (the sc1 bit; an L1 invalidate at workgroup
// scope -- buffer_inv sc0 -- does nothing on this part and one at agent scope costs 8-15 us: tools/probe/xcd_team_barrier.hip).
// COH = false: an ordinary load.  16-byte values travel as two 8-byte loads (4-byte alignment suffices for the hardware).
template <bool COH, typename T> __device__ __forceinline__ T ld_coh(const T* p) {
    if (!COH) return *p;
    T out;
    if (sizeof(T) == 1) { const unsigned char v = __hip_atomic_load(reinterpret_cast<const unsigned char*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_memcpy(&out, &v, 1); }
    else if (sizeof(T) == 4) { const unsigned int v = __hip_atomic_load(reinterpret_cast<const unsigned int*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_memcpy(&out, &v, 4); }
    else if (sizeof(T) == 8) { const unsigned long long v = __hip_atomic_load(reinterpret_cast<const unsigned long long*>(p), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); __builtin_memcpy(&out, &v, 8); }
    else {
        static_assert(sizeof(T) == 1 || sizeof(T) == 4 || sizeof(T) == 8 || sizeof(T) == 16, "ld_coh: 1, 4, 8 or 16 bytes");
        const unsigned long long* q = reinterpret_cast<const unsigned long long*>(p);
        const unsigned long long v[2] = {__hip_atomic_load(q, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __hip_atomic_load(q + 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)};
        __builtin_memcpy(&out, v, 16);
    }
    return out;
}
template <bool COH = false>
__device__ __forceinline__ SpRow row_from_sums(const SpSums& s, int k, bool with_planes, SpRow prev) {
    SpRow row = prev;
    // the record is read as whole 16-byte pieces (two cache lines per superpixel)
    const int4* __restrict__ rec = reinterpret_cast<const int4*>(&s.r[k]);
    const int4 i0 = ld_coh<COH>(rec), i1 = ld_coh<COH>(rec + 1);             // sx sy sr sg | sb n dx dy
    const float n = (float)i1.y;
    row.cx = (float)i0.x / n; row.cy = (float)i0.y / n;
    row.r = (float)i0.z / n; row.g = (float)i0.w / n; row.b = (float)i1.x / n;
    row.size = n;
    if (with_planes) {
        int dn_i = ld_coh<COH>(&s.r[k].dn);
        const longlong2* __restrict__ q = reinterpret_cast<const longlong2*>(&s.r[k].dxx);
        longlong2 q0 = ld_coh<COH>(q), q1 = ld_coh<COH>(q + 1), q2 = ld_coh<COH>(q + 2);         // dxx dyy | dxy dxd | dyd dd
        // the whole record in ONE round trip: left alone, the compiler fetches dxd / dyd / dd only behind the first
        // test of the plane solve, a second dependent trip to memory in every RGB-D pass
        asm volatile("" : "+v"(dn_i), "+v"(q0.x), "+v"(q0.y), "+v"(q1.x), "+v"(q1.y), "+v"(q2.x), "+v"(q2.y));
        const double inv = 1.0 / SSF_DISP_SCALE;
        const float dx = (float)i1.z, dy = (float)i1.w, dn = (float)dn_i;
        const float dxx = (float)q0.x, dyy = (float)q0.y, dxy = (float)q1.x;
        const float dxd = (float)((double)q1.y * inv), dyd = (float)((double)q2.x * inv);
        const float dd = (float)((double)q2.y * inv);
        float ta, tb, tc;
        if (!plane_solve(ta, tb, tc, dxx, dxy, dx, dxd, dxy, dyy, dy, dyd, dx, dy, dn, dd)) {
            ta = 0.f; tb = 0.f; tc = __uint_as_float(0xFFE00000u);
        }
        row.ta = ta; row.tb = tb; row.tc = tc;
    }
    return row;
}
// the same in two independent halves, so that two waves of a workgroup can build a window row side by side (the plane
// solve is a chain of three dependent IEEE divisions; the means are five independent ones): identical arithmetic
template <bool COH = false>
__device__ __forceinline__ void row_means_from_sums(const SpSums& s, int k, SpRow& row) {
    const int4* __restrict__ rec = reinterpret_cast<const int4*>(&s.r[k]);
    const int4 i0 = ld_coh<COH>(rec), i1 = ld_coh<COH>(rec + 1);             // sx sy sr sg | sb n dx dy
    const float n = (float)i1.y;
    row.cx = (float)i0.x / n; row.cy = (float)i0.y / n;
    row.r = (float)i0.z / n; row.g = (float)i0.w / n; row.b = (float)i1.x / n;
    row.size = n;
}
template <bool COH = false>
__device__ __forceinline__ void row_plane_from_sums(const SpSums& s, int k, float& ta, float& tb, float& tc) {
    const int4* __restrict__ rec = reinterpret_cast<const int4*>(&s.r[k]);
    int4 i1 = ld_coh<COH>(rec + 1);
    int dn_i = ld_coh<COH>(&s.r[k].dn);
    const longlong2* __restrict__ q = reinterpret_cast<const longlong2*>(&s.r[k].dxx);
    longlong2 q0 = ld_coh<COH>(q), q1 = ld_coh<COH>(q + 1), q2 = ld_coh<COH>(q + 2);         // dxx dyy | dxy dxd | dyd dd
    asm volatile("" : "+v"(i1.z), "+v"(i1.w), "+v"(dn_i), "+v"(q0.x), "+v"(q0.y), "+v"(q1.x), "+v"(q1.y), "+v"(q2.x), "+v"(q2.y));
    const double inv = 1.0 / SSF_DISP_SCALE;
    const float dx = (float)i1.z, dy = (float)i1.w, dn = (float)dn_i;
    const float dxx = (float)q0.x, dyy = (float)q0.y, dxy = (float)q1.x;
    const float dxd = (float)((double)q1.y * inv), dyd = (float)((double)q2.x * inv);
    const float dd = (float)((double)q2.y * inv);
    if (!plane_solve(ta, tb, tc, dxx, dxy, dx, dxd, dxy, dyy, dy, dyd, dx, dy, dn, dd)) {
        ta = 0.f; tb = 0.f; tc = __uint_as_float(0xFFE00000u);
    }
}
// ---- relabelling pass --------------------------------------------------------------------------
#define TILE 32
#define TW (TILE + 2)
#define WIN_MAX 64
#ifndef WIN_SMALL
#define WIN_SMALL 36          // the window of a 32 x 32 tile at cell size 16 with its margin of two cells: (2 + 4)^2
#endif
static inline dim3 tile_grid(const SegParams& p) { return dim3((p.W + TILE - 1) / TILE, (p.H + TILE - 1) / TILE); }

__device__ __forceinline__ void load_label_tile(int* tile, const int32_t* __restrict__ src, int X0, int Y0, int W, int H) {
    for (int i = threadIdx.x; i < TW * TW; i += blockDim.x) {
        const int lx = i % TW, ly = i / TW;
        const int x = X0 - 1 + lx, y = Y0 - 1 + ly;
        tile[i] = (x >= 0 && x < W && y >= 0 && y < H) ? src[(size_t)y * W + x] : -1;
    }
}
// The same tile in two steps, so that its five loads per thread travel together with everything else a kernel requests
// up front: tile_request issues them unconditionally (offsets clamped into the image, `outside` = the rounds whose
// element is not in it; loads behind bounds tests are separate blocks for the compiler, which then waits for each),
// tile_commit stores the tile to LDS.
#define TILE_ROUNDS ((TW * TW + 255) / 256)
struct TileRegs { int v[TILE_ROUNDS]; unsigned int outside; };
__device__ __forceinline__ TileRegs tile_request(const int32_t* __restrict__ src, int X0, int Y0, int W, int H) {
    TileRegs t; t.outside = 0u;
    unsigned int off[TILE_ROUNDS];
#pragma unroll
    for (int k = 0; k < TILE_ROUNDS; k++) {
        const int i = 256 * (k + 1) <= TW * TW ? threadIdx.x + 256 * k : min((int)threadIdx.x + 256 * k, TW * TW - 1);
        const int ly = i / TW, lx = i - ly * TW;
        const int gx_ = X0 - 1 + lx, gy_ = Y0 - 1 + ly;
        const int cx_ = min(max(gx_, 0), W - 1), cy_ = min(max(gy_, 0), H - 1);
        if (cx_ != gx_ || cy_ != gy_) t.outside |= 1u << k;
        off[k] = __umul24((unsigned int)cy_, (unsigned int)W) + (unsigned int)cx_;
    }
#pragma unroll
    for (int k = 0; k < TILE_ROUNDS; k++) t.v[k] = src[off[k]];
    return t;
}
__device__ __forceinline__ void tile_commit(int* tile, const TileRegs& t) {
#pragma unroll
    for (int k = 0; k < TILE_ROUNDS; k++) {
        const int i = threadIdx.x + 256 * k;
        if (256 * (k + 1) <= TW * TW || i < TW * TW) tile[i] = (t.outside >> k) & 1u ? -1 : t.v[k];
    }
}
__device__ __forceinline__ int tile_boundary(const int* tile, int lx, int ly) {   // lx,ly in halo coordinates
    const int own = tile[ly * TW + lx];
    return (tile[(ly - 1) * TW + lx] != own) + (tile[ly * TW + lx - 1] != own) + (tile[ly * TW + lx + 1] != own) +
           (tile[(ly + 1) * TW + lx] != own);
}

// Window of grid cells around a 32x32 tile.  Superpixels stay close to their seed cell, so per-tile
// LDS tables indexed by window cell serve (almost) every label of the tile; a label that drifted out
// of the window takes an exact global-memory slow path.
// x / p.cell for a pixel coordinate (0 <= x < 65536) without the 20-instruction integer division
__device__ __forceinline__ int div_cell(const SegParams& p, int x) { return p.cell_magic ? (int)__umulhi((unsigned int)x, p.cell_magic) : x; }
struct CellWindow {
    int cx0, cy0, nwx, nwy, gx; float inv_gx; bool ok;
    __device__ __forceinline__ void init(const SegParams& p, int X0, int Y0, int max_entries) {
        int margin = 2;
        const int tcx0 = div_cell(p, X0), tcy0 = div_cell(p, Y0);
        const int tcx1 = div_cell(p, min(X0 + TILE - 1, p.W - 1)), tcy1 = div_cell(p, min(Y0 + TILE - 1, p.H - 1));
        while (margin > 0 && (tcx1 - tcx0 + 1 + 2 * margin) * (tcy1 - tcy0 + 1 + 2 * margin) > max_entries) margin--;
        cx0 = tcx0 - margin; cy0 = tcy0 - margin;
        nwx = tcx1 - tcx0 + 1 + 2 * margin; nwy = tcy1 - tcy0 + 1 + 2 * margin;
        ok = nwx * nwy <= max_entries; gx = p.gx; inv_gx = 1.0f / (float)p.gx;
    }
    __device__ __forceinline__ int size() const { return ok ? nwx * nwy : 0; }
    // window slot of a label, -1 when outside
    __device__ __forceinline__ int slot(int l) const {
        const int cyl = (int)(((float)l + 0.5f) * inv_gx);            // l / gx, exact for l < 2^20
        const int wx = (l - cyl * gx) - cx0, wy = cyl - cy0;
        return (ok && wx >= 0 && wx < nwx && wy >= 0 && wy < nwy) ? wy * nwx + wx : -1;
    }
    // label of window slot i, -1 when the cell is outside the grid
    __device__ __forceinline__ int label_of(int i, int gy) const {
        const int cx = cx0 + i % nwx, cy = cy0 + i / nwx;
        return (cx >= 0 && cx < gx && cy >= 0 && cy < gy) ? cy * gx + cx : -1;
    }
};
__device__ __forceinline__ void lds_add_i64(unsigned long long* p, long long v) { atomicAdd(p, (unsigned long long)v); }

// the 9 inlier-only disparity sums of one pixel moved by `sign`
__device__ __forceinline__ void disp_sums_add(const SpSums& s, int k, int x, int y, float d, int sign) {
    atomicAdd(&s.r[k].dx, sign * x); atomicAdd(&s.r[k].dy, sign * y); atomicAdd(&s.r[k].dn, sign);
    atomic_add_i64(&s.r[k].dxx, (long long)sign * x * x);
    atomic_add_i64(&s.r[k].dyy, (long long)sign * y * y);
    atomic_add_i64(&s.r[k].dxy, (long long)sign * x * y);
    atomic_add_i64(&s.r[k].dxd, sign * fx64((double)((float)x * d), SSF_DISP_SCALE, SSF_DISP_LIM));
    atomic_add_i64(&s.r[k].dyd, sign * fx64((double)((float)y * d), SSF_DISP_SCALE, SSF_DISP_LIM));
    atomic_add_i64(&s.r[k].dd, sign * fx64((double)d, SSF_DISP_SCALE, SSF_DISP_LIM));
}
// Field order of the per-window-superpixel LDS accumulators of a pass (15 exact integer sums).
enum { F_SX, F_SY, F_SR, F_SG, F_SB, F_N, F_DX, F_DY, F_DN, F_DXX, F_DYY, F_DXY, F_DXD, F_DYD, F_DD, F_COUNT };
__device__ __forceinline__ void lds_rgb(unsigned long long* a, int sign, int x, int y, uint32_t rgbf) {
    lds_add_i64(&a[F_SX], sign * x); lds_add_i64(&a[F_SY], sign * y);
    lds_add_i64(&a[F_SR], sign * (int)(rgbf & 255u)); lds_add_i64(&a[F_SG], sign * (int)((rgbf >> 8) & 255u));
    lds_add_i64(&a[F_SB], sign * (int)((rgbf >> 16) & 255u)); lds_add_i64(&a[F_N], sign);
}
__device__ __forceinline__ void lds_disp(unsigned long long* a, int sign, int x, int y, float d) {
    lds_add_i64(&a[F_DX], sign * x); lds_add_i64(&a[F_DY], sign * y); lds_add_i64(&a[F_DN], sign);
    lds_add_i64(&a[F_DXX], (long long)sign * x * x); lds_add_i64(&a[F_DYY], (long long)sign * y * y);
    lds_add_i64(&a[F_DXY], (long long)sign * x * y);
    lds_add_i64(&a[F_DXD], sign * fx64((double)((float)x * d), SSF_DISP_SCALE, SSF_DISP_LIM));
    lds_add_i64(&a[F_DYD], sign * fx64((double)((float)y * d), SSF_DISP_SCALE, SSF_DISP_LIM));
    lds_add_i64(&a[F_DD], sign * fx64((double)d, SSF_DISP_SCALE, SSF_DISP_LIM));
}
__device__ __forceinline__ void flush_field(const SpSums& s, int l, int field, long long v) {
    switch (field) {
        case F_SX: atomicAdd(&s.r[l].sx, (int)v); break;
        case F_SY: atomicAdd(&s.r[l].sy, (int)v); break;
        case F_SR: atomicAdd(&s.r[l].sr, (int)v); break;
        case F_SG: atomicAdd(&s.r[l].sg, (int)v); break;
        case F_SB: atomicAdd(&s.r[l].sb, (int)v); break;
        case F_N: atomicAdd(&s.r[l].n, (int)v); break;
        case F_DX: atomicAdd(&s.r[l].dx, (int)v); break;
        case F_DY: atomicAdd(&s.r[l].dy, (int)v); break;
        case F_DN: atomicAdd(&s.r[l].dn, (int)v); break;
        case F_DXX: atomic_add_i64(&s.r[l].dxx, v); break;
        case F_DYY: atomic_add_i64(&s.r[l].dyy, v); break;
        case F_DXY: atomic_add_i64(&s.r[l].dxy, v); break;
        case F_DXD: atomic_add_i64(&s.r[l].dxd, v); break;
        case F_DYD: atomic_add_i64(&s.r[l].dyd, v); break;
        default: atomic_add_i64(&s.r[l].dd, v); break;
    }
}

// One pass (OX,OY) of the boundary relabelling: updateTPSRGB_kernel / updateTPSRGBD_kernel,
// TPS_RGBD_kernels.cuh:235-651.  256 threads own the 256 pass pixels of a 32x32 tile.
//
// Labels are updated IN PLACE.  A pass touches rows y = OY (mod 2) and columns x = 0,3 (mod 4) for
// OX = 0 or x = 1,2 (mod 4) for OX = 1, i.e. horizontally adjacent PAIRS of pixels; everything else a
// pass pixel looks at (rows y-1, y+1, the outer neighbours of its pair) is not modified in this pass.
// The tile grid is shifted so that a pair never straddles two tiles (X0 = 2 mod 4 for OX = 0): a
// workgroup snapshots its tile + halo into LDS, decides from the snapshot, and is the only writer of
// the pass pixels it read -- exactly the "all reads before any write" schedule of the oracle (A1)
// without a second label map or a copy of the untouched pixels.
//
// No merge launch between passes.  The exact sums are double buffered: pass k builds the superpixel
// rows (means, plane) it needs straight from sums[k&1], which nothing writes during the pass, and
// applies its relabelling deltas to sums[(k+1)&1] together with the replayed log of pass k-1 (the
// deltas that buffer is still missing).  Integer adds commute, so the lagging buffer catches up
// exactly.  The rows of the superpixels around the tile (a window of grid cells) are computed by
// the first threads of the workgroup into LDS while the label tile is being staged; the energy
// then reads rows from LDS.  Labels that have drifted out of the window take an exact slow path.
// NPX = pass pixels per thread: the tile is 32 * NPX pixels wide (32 rows).  NPX = 2 halves the number of workgroups of a
// launch: with 8 frames per launch the 32-wide grid is 2520 workgroups = 10080 waves, more than the 8192 the part
// holds at once (a second, mostly empty round of workgroups); the 64-wide grid fits in one round and amortises the
// window's row computation and the flush over twice the pixels.  NPX = 1 stays for single-frame launches (latency).
// Instruction diet (round 3).  Counters + an experiment (150 extra vector instructions per wave: +2.3 us on a 20 us launch) show the
// pass is bound by instruction ISSUE as much as by latency: 6.3 M wave-level VALU instructions per 8-frame RGB-D launch are
// 10 us of the part's vector issue capacity.  Hence: the candidate labels of a pixel de-duplicated and compacted (the wave walks
// 1-2 candidates instead of 4 directions), interior tiles staged without bounds tests and through one scalar base, the
// connectivity guard as a bit mask + popcount, window slots by unsigned compares, the nine 32-bit sums in 32-bit LDS
// accumulators, the disparity terms of a pixel converted once for both signs, the flush addressed by field offset
// instead of a 15-way switch, the previous pass' log entry fetched without a branch (behind one, the compiler waited for
// it -- a second dependent trip to memory -- before it requested the window's sums).
// Global accesses of the pass as (uniform 64-bit base) + (32-bit BYTE offset in a vector register): the form the hardware
// takes directly (`global_load_dword v, v_off, s[base]`).  Indexing a typed pointer with an integer makes the compiler
// widen, shift and add in 64 bits per access (v_mad_i64_i32 / v_lshl_add_u64 and a zeroed high half): two or three
// vector instructions and a register pair each, a dozen times per thread.  Every map of a frame is far below 4 GB.
template <typename T> __device__ __forceinline__ T ld_off(const void* __restrict__ base, unsigned int byte_off) {
    return *reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off);
}
template <bool COH, typename T> __device__ __forceinline__ T ld_off_c(const void* __restrict__ base, unsigned int byte_off) {      // (see ld_coh)
    return ld_coh<COH>(reinterpret_cast<const T*>(reinterpret_cast<const char*>(base) + byte_off));
}
template <typename T> __device__ __forceinline__ void st_off(void* __restrict__ base, unsigned int byte_off, const T& v) {
    *reinterpret_cast<T*>(reinterpret_cast<char*>(base) + byte_off) = v;
}
// XCD-aware tile order.  A launch's workgroups are dealt to the eight XCDs round-robin (workgroup b of the x-fastest linear
// order runs on XCD b % 8: observed placement, only speed depends on it), so with tiles taken in grid order the neighbours
// of a tile run on seven OTHER XCDs -- and every 128-byte line that two tiles share (the halo columns, the shifted tiles'
// rows that straddle a line, the 32 inlier bytes a tile uses of a line, the window's sum records) is fetched once per L2:
// the RGB-D pass moved 58 MB per 8-frame launch for 17-34 MB of data, at 3.9 TB/s -- it had become traffic-bound.
// Here XCD x takes the x-th contiguous EIGHTH of the launch's (frame, tile row, tile column) order instead -- a whole frame
// per XCD for an 8-frame launch -- so neighbouring tiles meet in one L2.  Bijective for any grid; tile ids (the log regions
// of a tile) are those of the logical tile, the same in every pass.
struct TileOrder { unsigned int ntx, ntile, magic_ntx, magic_ntile, total; int xcd; };
__device__ __forceinline__ unsigned int xcd_share(unsigned int b, unsigned int nb) {
    // x (q + 1) for x < r, r (q + 1) + (x - r) q otherwise == x q + min(x, r): no branch (this runs in every wave's prologue)
    const unsigned int q = nb >> 3, r = nb & 7u, x = b & 7u;
    return x * q + min(x, r) + (b >> 3);
}
static inline TileOrder tile_order(dim3 grid) {
    TileOrder o;
    o.ntx = grid.x; o.ntile = grid.x * grid.y; o.total = o.ntile * grid.z;
    o.magic_ntx = (unsigned int)((0x100000000ull + o.ntx - 1) / o.ntx);        // n / d == umulhi(n, ceil(2^32 / d)) for n d < 2^32 / d ... n < 2^20 here
    o.magic_ntile = (unsigned int)((0x100000000ull + o.ntile - 1) / o.ntile);
    static const int on = SSF_ENV_INT("PASS_XCD", 1);          // (lab: 0 = tiles in grid order; the product: always the XCD-aware order)
    o.xcd = on;
    return o;
}
// the same order for the other tile kernels of the extract stage (round 6: k_eval_samples, k_init_disp, k_render_moments fetch a
// label tile WITH its halo -- three 128-byte lines per tile row instead of one -- and share the inlier lines four tiles to a line;
// in grid order the neighbours of a tile run on seven other XCDs and every one of those lines is pulled into several L2s)
#ifndef SSF_TILE_XCD
#define SSF_TILE_XCD 1
#endif
struct TileIdx { unsigned int bx, by, bz; };
__device__ __forceinline__ TileIdx tile_index(const TileOrder& ord) {
    TileIdx t; t.bx = blockIdx.x; t.by = blockIdx.y; t.bz = blockIdx.z;
    if (SSF_TILE_XCD && ord.xcd) {
        const unsigned int lin = blockIdx.x + ord.ntx * blockIdx.y + ord.ntile * blockIdx.z;
        const unsigned int u = xcd_share(lin, ord.total);
        t.bz = __umulhi(u, ord.magic_ntile);
        const unsigned int r = u - t.bz * ord.ntile;
        t.by = __umulhi(r, ord.magic_ntx); t.bx = r - t.by * ord.ntx;
    }
    return t;
}
#ifndef SSF_PASS_NPREV_RGBD
#define SSF_PASS_NPREV_RGBD 1
#endif
#ifndef SSF_PASS_NPREV_RGB
#define SSF_PASS_NPREV_RGB 1
#endif
#define PASS_F32 9                  // F_SX .. F_DN: 32-bit accumulators; F_DXX .. F_DD: 64-bit
#define PASS_F64 (F_COUNT - PASS_F32)
// accumulators of one window slot: 24 dwords = six 16-byte chunks -- the nine 32-bit sums (chunks 0-2, three dwords of padding),
// then the six 64-bit sums (chunks 3-5; RGB passes never touch them).  Zeroed and scanned a chunk at a time.
#define PASS_ACC_DW 24
#define PASS_ACC_WIDE_DW 12
// What a pass touches of a frame's working set, chosen on the HOST (round 5): the read / write sums buffer by the pass' parity, the
// previous / current log by pass mod 3.  Selected in the kernel -- from FrameMaps, after shifting all of its ~30 pointers to the
// batch slot -- it was ~50 scalar instructions of every wave's prologue (s_cselect chains + 64-bit adds); the counters put the
// prologue at 205 of the RGB-D wave's 321 scalar instructions, and 150 extra scalar instructions per wave cost the 8-frame
// launch 1.8 us (profiles/pass_issue_r05.txt): ONE scalar unit serves the four SIMDs of a compute unit, and this kernel is as
// close to its issue limit as to the vector pipes'.
struct PassArgs {
    const SumRec* sr; SumRec* sw;                                      // sums[k & 1] (read), sums[(k + 1) & 1] (written)
    const int4* pent; const float* pdis; const unsigned int* pcnt;     // log of pass k - 1
    int4* cent; float* cdis; unsigned int* ccnt;                       // log of pass k
    const uint2* geom;                                                 // window geometry per tile of this pass' grid (PassGeomEntry), or null
};
// lab: when g_pass_trace is set every workgroup of a pass launch leaves five ticks of the 100 MHz wall clock -- entry | everything it
// requested up front has arrived | staged (first barrier) | decisions taken (second barrier) | end -- at 5 x its linear block index
// (ssf_dbg_trace_pass, tools/pass_trace.py: where does a tile's 5-6 us go?).  Nothing of it in the product.
#ifdef SSF_EXPERIMENTS
__device__ unsigned long long* g_pass_trace = nullptr;
void set_pass_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pass_trace), &p, sizeof(p)); }
#define SSF_PASS_TICK_BEGIN() unsigned long long* const pass_trace = g_pass_trace; \
    const size_t pass_wg = blockIdx.x + (size_t)gridDim.x * (blockIdx.y + (size_t)gridDim.y * blockIdx.z); \
    if (pass_trace && threadIdx.x == 0) pass_trace[5 * pass_wg] = wall_clock64()
#define SSF_PASS_TICK(i) do { if (pass_trace && threadIdx.x == 0) pass_trace[5 * pass_wg + (i)] = wall_clock64(); } while (0)
#define SSF_PASS_TICK_LOADS() do { if (pass_trace) { asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); __syncthreads(); SSF_PASS_TICK(1); } } while (0)
// lab: statistics of a frame's relabelling passes in FrameMaps::epoch[8 ..] (ssf_dbg_pass_stats; bench.py's real-frame leg, round 6):
// [8] log entries written (sum over tiles and passes), [9] the largest log of a tile in a pass, [10] tiles x passes, [11] superpixel-row
// lookups that found their label OUTSIDE the tile's LDS window (the exact global path), [12] all row lookups
// (only while g_pass_stats is set -- ssf_dbg_pass_stats_enable: a global atomic per lookup is not something an A/B of the lab build should pay)
__device__ int g_pass_stats = 0;
void set_pass_stats(int on) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_pass_stats), &on, sizeof(on)); }
#define SSF_PASS_STAT_LOOKUP(far) do { if (g_pass_stats) { atomicAdd(&m.epoch[12], 1u); if (far) atomicAdd(&m.epoch[11], 1u); } } while (0)
#define SSF_PASS_STAT_END(nlog) do { if (g_pass_stats && threadIdx.x == 0) { atomicAdd(&m.epoch[8], (nlog)); atomicMax(&m.epoch[9], (nlog)); atomicAdd(&m.epoch[10], 1u); } } while (0)
#else
#define SSF_PASS_TICK_BEGIN() ((void)0)
#define SSF_PASS_TICK(i) ((void)0)
#define SSF_PASS_TICK_LOADS() ((void)0)
#define SSF_PASS_STAT_LOOKUP(far) ((void)0)
#define SSF_PASS_STAT_END(nlog) ((void)0)
#endif
template <bool RGBD, int NPX, int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_update_pass(SegParams p, FrameMaps m, PassArgs pa, int pass, int OX, int OY, int dbg, TileOrder ord) {
    constexpr bool COH = false;
    // this workgroup's (frame, tile row, tile column): see TileOrder
    unsigned int bx = blockIdx.x, by = blockIdx.y, bz = blockIdx.z;
#ifdef SSF_EXPERIMENTS
    if (ord.xcd) {              // (lab: SSF_PASS_XCD=0 = tiles in grid order)
#else
    {
#endif
        const unsigned int lin = blockIdx.x + ord.ntx * blockIdx.y + ord.ntile * blockIdx.z;
        const unsigned int t = xcd_share(lin, ord.total);
        bz = __umulhi(t, ord.magic_ntile);
        const unsigned int r = t - bz * ord.ntile;
        by = __umulhi(r, ord.magic_ntx); bx = r - by * ord.ntx;
    }
#include "ssf_pass_tile.hpp"
}
#ifdef SSF_EXPERIMENTS
// the same body as a function of (tile column, row, frame slot): lab/passes_team.inc
template <bool RGBD, int NPX, bool COH>
__device__ __forceinline__ void update_pass_tile(const SegParams& p, FrameMaps m, const PassArgs& pa, int pass, int OX, int OY, int dbg,
                                                 const TileOrder& ord, unsigned int bx, unsigned int by, unsigned int bz) {
#include "ssf_pass_tile.hpp"
}
#endif

// ---- the passes of a phase in ONE launch, a frame per XCD (round 5): a measurement arm that lost its A/B (DESIGN.md section 7,
// profiles/pass_team_r05.txt) -- lab/passes_team.inc, compiled only into the lab variant of the library
#ifdef SSF_EXPERIMENTS
#include "lab/passes_team.inc"
#else
size_t pass_team_ws_bytes() { return 0; }
size_t pass_args_bytes(int) { return 0; }
void pass_args_table(const SegParams&, const FrameMaps&, int, void*) {}
void launch_update_passes_team(hipStream_t, const SegParams&, FrameMaps&, int, int, int, bool, const void*, void*, unsigned int*) {}
#endif

// ---- RANSAC plane initialisation ---------------------------------------------------------------
__device__ __forceinline__ size_t tex_index(float x, float y, int W, int H) {   // point sampling, clamp
    const int ix = min(max((int)floorf(x), 0), W - 1), iy = min(max((int)floorf(y), 0), H - 1);
    return (size_t)iy * W + ix;
}
// initSamples_kernel, TPS_RGBD_kernels.cu:324-401: one thread per (superpixel, sample)
__global__ __launch_bounds__(256) void k_init_samples(SegParams p, FrameMaps m, int true_buf) {
    const uint32_t idx = blockIdx.x * blockDim.x + threadIdx.x;
    if (idx >= (uint32_t)(p.S * p.nb_samples)) return;
    const int index = (int)(idx / (uint32_t)p.nb_samples);
    m = batch_slot(m, blockIdx.y);
    // draws of frame number e use counters e*64 .. e*64+63 of stream idx (a frame needs at most 50)
    uint32_t ctr = m.epoch[0] * 64u;
    const float radius = (float)p.cell / 2.f;
    {
        const int32_t* __restrict__ label = m.label;
        // centroid = mergeTPSRGBCoeffs of this superpixel, straight from the exact sums
        const SpSums sm = true_buf ? m.sums[1] : m.sums[0];
        const float nn = (float)sm.r[index].n;
        const float cx = (float)sm.r[index].sx / nn, cy = (float)sm.r[index].sy / nn;
        float x = cx, y = cy;
        int i = label[tex_index(x, y, p.W, p.H)];
        int k = 0;
        while (i != index && k++ < 10) {
            const float u1 = rng_unit(rng_draw(p.seed, idx, ctr));
            x = (float)((double)cx + ((double)radius * 2.) * (double)(u1 - 1.f));
            const float u2 = rng_unit(rng_draw(p.seed, idx, ctr));
            y = (float)((double)cy + ((double)radius * 2.) * (double)(u2 - 1.f));
            i = label[tex_index(x, y, p.W, p.H)];
        }
        const float d0 = m.disp[tex_index(x, y, p.W, p.H)];
        float px[3] = {x, x, x}, py[3] = {y, y, y}, pd[3] = {d0, d0, d0};
        // The three 10-step walks were a chain of 30 dependent label reads (21 us for a kernel of 19 200 threads).  But a
        // step only ever moves while the walker stands on its own superpixel, and the directions come from the counter
        // RNG alone: the path it WOULD take if it never left the superpixel ("free path") is known without touching
        // memory.  So: free path first, then all its labels and disparities in one round of independent loads, then
        // the walk is replayed from registers -- it follows the free path up to the first foreign pixel and stays there.
        constexpr int NSTEP = 30;
        float qx[NSTEP + 1], qy[NSTEP + 1];
        uint32_t moved = 0u;                               // bit s: step s changes the position (its target is inside the image)
        qx[0] = x; qy[0] = y;
#pragma unroll
        for (int s = 1; s <= NSTEP; s++) {
            const int dir = (int)(rng_draw(p.seed, idx, ctr) & 3u);
            const float ddx = (dir == 0) ? -1.f : ((dir == 2) ? 1.f : 0.f);
            const float ddy = (dir == 1) ? -1.f : ((dir == 3) ? 1.f : 0.f);
            const float nxp = qx[s - 1] + ddx, nyp = qy[s - 1] + ddy;
            const bool inb = nxp >= 0 && nxp < (float)p.W && nyp >= 0 && nyp < (float)p.H;
            qx[s] = inb ? nxp : qx[s - 1]; qy[s] = inb ? nyp : qy[s - 1];
            moved |= inb ? (1u << s) : 0u;
        }
        int ql[NSTEP]; float qd[NSTEP + 1];
#pragma unroll
        for (int t = 0; t < NSTEP; t++) ql[t] = label[tex_index(qx[t], qy[t], p.W, p.H)];
#pragma unroll
        for (int s = 1; s <= NSTEP; s++) qd[s] = m.disp[tex_index(qx[s], qy[s], p.W, p.H)];
        bool on = true;                                    // the walker still stands on its own superpixel
#pragma unroll
        for (int j = 0; j < 3; j++)
#pragma unroll
            for (int w = 0; w < 10; w++) {
                const int s = 10 * j + w + 1;
                on = on && ql[s - 1] == index;             // label at the CURRENT position (:370)
                if (on && ((moved >> s) & 1u)) {
                    const float dd = qd[s];
                    if (isfinite(dd)) { px[j] = qx[s]; py[j] = qy[s]; pd[j] = dd; }
                }
            }
        float a, b, c;
        if (!plane_solve(a, b, c, px[0], py[0], 1.f, pd[0], px[1], py[1], 1.f, pd[1], px[2], py[2], 1.f, pd[2])) {
            a = 0.f; b = 0.f; c = pd[2];
        }
        m.samples[idx] = make_float4(a, b, c, 0.f);
        m.sample_score[idx] = 0;
    }
}

// evalSamples_kernel, TPS_RGBD_kernels.cu:403-433: integer scores.  Tile kernel: the candidate
// planes of the window's superpixels are staged in LDS, every pixel tests its label's planes and
// counts with LDS integer atomics; the tile flushes non-zero counts with one global atomic each.
#define EVAL_WIN 64
#define ACC_REP 8
#define EVAL_NS 16
#define EVAL_REP 8
#ifndef SSF_PASS_RGBD_WAVES
#define SSF_PASS_RGBD_WAVES 6          // waves per SIMD the RGB-D pass is compiled for (launch bounds)
#endif
// unused dynamic LDS of the two accumulator-heavy tile kernels: caps their workgroups per compute unit (an occupancy knob for A/Bs)
#ifndef SSF_INITDISP_DYN_LDS
#define SSF_INITDISP_DYN_LDS 0
#endif
#ifndef SSF_RENDER_DYN_LDS
#define SSF_RENDER_DYN_LDS 0
#endif
#ifndef SSF_EVAL_PLANE_STRIDE
#define SSF_EVAL_PLANE_STRIDE 17
#endif
// FAST = the reference's sixteen samples (every launch file): per pixel all planes of its superpixel are read from LDS first,
// the sixteen tests follow, and the votes go out as LDS atomics nobody waits for -- TWO samples per 32-bit counter (16-bit
// fields: a tile has 1024 pixels), in the layout [pair][window cell][replica]: eight atomics per pixel instead of sixteen,
// lanes of different superpixels on different banks (with [cell][sample][replica] the bank was (8 sample + replica) % 32
// whatever the cell: every atomic of a wave fought over eight banks), and 16 KB of counters instead of 32 (five workgroups
// per CU instead of three).
// WCAP (round 6): the window capacity the LDS tables are sized for -- 36 cells when no tile of the handle's grid needs more (cell
// size 16, every launch file: 2 x 2 cells of the tile + a margin of two), 64 otherwise.  At 64 the three accumulator kernels held
// 34-38 KB of LDS and four workgroups per compute unit, and all three get slower with fewer (25 -> 29 -> 36 us, 50 -> 54 -> 62 us at
// 4 / 3 / 2, round 5); at 36 they hold 19-23 KB: six to eight workgroups.
template <bool FAST, int WCAP>
__global__ __launch_bounds__(256) void k_eval_samples(SegParams p, FrameMaps m, TileOrder ord) {
    // (FAST: the sixteen planes of a window cell 17 float4 apart, not 16: the lanes of a wave read plane sk of a handful of DIFFERENT
    // cells in one ds_read_b128, and with a stride of 64 dwords every cell's plane sk sits on the same four banks -- the counters
    // put 53 % of this kernel's LDS cycles down to bank conflicts.  Round 5, same box, alternated: 33.2-34.4 us per 8-frame launch
    // at a stride of 16, 26.0-26.9 at 17, 32.5-32.8 at 24 (cells two apart collide again; three workgroups per compute unit instead
    // of four).)
    constexpr int PLANE_STRIDE = FAST ? SSF_EVAL_PLANE_STRIDE : EVAL_NS;
    __shared__ float4 w_plane[WCAP * (FAST ? SSF_EVAL_PLANE_STRIDE : EVAL_NS)];
    // EVAL_REP replicas of every counter (lane & 7): the 64 pixels of a wave sit in a handful of superpixels, and
    // same-address LDS atomics serialise (SQ_LDS_BANK_CONFLICT was 80 % of the LDS cycles with one replica)
    __shared__ __attribute__((aligned(16))) int w_cnt[WCAP * (FAST ? EVAL_NS / 2 : EVAL_NS) * EVAL_REP];
    const TileIdx ti = tile_index(ord);                // (XCD-aware tile order: see tile_index)
    m = batch_slot(m, ti.bz);
    const int X0 = ti.bx * TILE, Y0 = ti.by * TILE;
    const int ns = p.nb_samples;
    CellWindow win; win.init(p, X0, Y0, ns <= EVAL_NS ? WCAP : 0);
    const int32_t* __restrict__ label = m.label;
    // this thread's pixels: requested up front, in flight while the window's planes are staged
    constexpr int PX = TILE * TILE / 256;
    // (clamped addresses, no branch around the loads: pixels outside the image are skipped below; and the window's planes
    // requested in the same round, stored to LDS afterwards -- one trip to memory in front of the evaluation instead of
    // one per piece)
    int pl[PX]; float pd[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int i = threadIdx.x + 256 * k;
        const int x = min(X0 + i % TILE, p.W - 1), y = min(Y0 + i / TILE, p.H - 1);
        const unsigned int q = __umul24((unsigned int)y, (unsigned int)p.W) + (unsigned int)x;
        pl[k] = label[q]; pd[k] = m.disp[q];
    }
    constexpr int PLANE_ROUNDS = (WCAP * EVAL_NS + 255) / 256;
    float4 wp[PLANE_ROUNDS]; bool wp_ok[PLANE_ROUNDS];
    const int n_planes = win.size() * ns;
#pragma unroll
    for (int k = 0; k < PLANE_ROUNDS; k++) {
        const int i = threadIdx.x + 256 * k;
        int l = -1, sk = 0;
        if (i < n_planes) { const int wi = i / ns; sk = i - wi * ns; l = win.label_of(wi, p.gy); }
        wp_ok[k] = l >= 0;
        wp[k] = m.samples[(size_t)(l >= 0 ? l : 0) * ns + sk];
    }
    if (FAST) {
        for (int i = threadIdx.x; i < WCAP * (EVAL_NS / 2) * EVAL_REP / 4; i += 256) reinterpret_cast<int4*>(w_cnt)[i] = make_int4(0, 0, 0, 0);
    } else
        for (int i = threadIdx.x; i < n_planes * EVAL_REP; i += blockDim.x) w_cnt[i] = 0;
#pragma unroll
    for (int k = 0; k < PLANE_ROUNDS; k++) {
        const int i = threadIdx.x + 256 * k;
        if (i < n_planes) w_plane[FAST ? (i / EVAL_NS) * PLANE_STRIDE + (i % EVAL_NS) : i] = wp_ok[k] ? wp[k] : make_float4(0.f, 0.f, 0.f, 0.f);
    }
    __syncthreads();
    // The reference's sample count (16 in every launch file) has a path of its own: per pixel, ALL planes of its superpixel
    // are read from LDS first, the sixteen tests follow, then the votes go out as LDS atomics nobody waits for.  (In the
    // general loop below every (pixel, sample) step waited for its plane's LDS read AND for the previous step's atomic:
    // 64 exposed LDS round trips per thread at three waves per SIMD were what the kernel spent its time on.)
    if (FAST) {
#pragma unroll
        for (int k = 0; k < PX; k++) {
            const int i = threadIdx.x + 256 * k;
            const int x = X0 + i % TILE, y = Y0 + i / TILE;
            if (x >= p.W || y >= p.H) continue;
            const int l = pl[k];
            const float d = pd[k];
            const int ws = win.slot(l);
            if (ws >= 0) {
                const float4* __restrict__ plane = &w_plane[ws * PLANE_STRIDE];
                unsigned int votes = 0u;
#pragma unroll
                for (int sk = 0; sk < EVAL_NS; sk++) {
                    const float4 th = plane[sk];
                    const float dp = (th.x * (float)x + th.y * (float)y) + th.z;
                    const float dd = (d - dp) * (d - dp);
                    // (no short circuit: behind `isfinite(th.z) &&` the compiler read th.z, waited, tested, and only then read
                    // th.x / th.y -- two dependent LDS trips per sample.  A plane that is not finite fails `dd < thresh` anyway.)
                    const unsigned int vote = (unsigned int)(dd < p.thresh_disp) & (unsigned int)(isfinite(th.z) ? 1 : 0);
                    votes |= vote << sk;
                }
                int* __restrict__ c = &w_cnt[ws * EVAL_REP + (lane_id() & (EVAL_REP - 1))];
#pragma unroll
                for (int pr = 0; pr < EVAL_NS / 2; pr++) {        // samples pr (low half) and pr + 8 (high half) share a counter
                    const unsigned int v = ((votes >> pr) & 1u) | (((votes >> (pr + EVAL_NS / 2)) & 1u) << 16);
                    if (v) atomicAdd(&c[pr * WCAP * EVAL_REP], (int)v);
                }
            } else {
                for (int sk = 0; sk < EVAL_NS; sk++) {             // label outside the window: exact global path
                    const float4 th = ld_global_f4(&m.samples[(size_t)l * EVAL_NS + sk]);
                    if (isfinite(th.z)) {
                        const float dp = (th.x * (float)x + th.y * (float)y) + th.z;
                        const float dd = (d - dp) * (d - dp);
                        if (dd < p.thresh_disp) atomicAdd(&m.sample_score[(size_t)l * EVAL_NS + sk], 1);
                    }
                }
            }
        }
    } else
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int i = threadIdx.x + 256 * k;
        const int x = X0 + i % TILE, y = Y0 + i / TILE;
        if (x >= p.W || y >= p.H) continue;
        const int l = pl[k];
        const float d = pd[k];
        const int ws = win.slot(l);
        // (a ballot/popcount aggregation per (label, sample) instead of replicas was measured slower: 41 us)
        for (int sk = 0; sk < ns; sk++) {
            float4 th = w_plane[(ws >= 0 ? ws : 0) * ns + sk];      // (LDS read always: see k_render_moments)
            if (ws < 0) th = ld_global_f4(&m.samples[(size_t)l * ns + sk]);
            if (isfinite(th.z)) {
                const float dp = (th.x * (float)x + th.y * (float)y) + th.z;
                const float dd = (d - dp) * (d - dp);
                if (dd < p.thresh_disp) {
                    if (ws >= 0) atomicAdd(&w_cnt[(ws * ns + sk) * EVAL_REP + (lane_id() & (EVAL_REP - 1))], 1);
                    else atomicAdd(&m.sample_score[(size_t)l * ns + sk], 1);
                }
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < win.size() * ns; i += blockDim.x) {
        int c = 0;
        if (FAST) {
            const int wi = i / EVAL_NS, sk = i % EVAL_NS, pr = sk & (EVAL_NS / 2 - 1);
            unsigned int packed = 0u;
#pragma unroll
            for (int r = 0; r < EVAL_REP; r++) packed += (unsigned int)w_cnt[(pr * WCAP + wi) * EVAL_REP + r];   // (fields < 2^13 each: no carry)
            c = (int)(sk < EVAL_NS / 2 ? (packed & 0xFFFFu) : (packed >> 16));
        } else {
#pragma unroll
            for (int r = 0; r < EVAL_REP; r++) c += w_cnt[i * EVAL_REP + r];
        }
        if (c) atomicAdd(&m.sample_score[(size_t)win.label_of(i / ns, p.gy) * ns + i % ns], c);
    }
}

// selectSamples_kernel, TPS_RGBD_kernels.cu:435-467: first sample with the strictly largest score
// (all-zero scores -> theta = 0).  Evaluated where it is consumed (k_init_disp) instead of in a launch
// of its own; the disparity sums it zeroes in the reference are still zero here (nothing has
// accumulated into them since ingest).
__device__ __forceinline__ float4 select_sample(const FrameMaps& m, int l, int ns) {
    float4 best = make_float4(0.f, 0.f, 0.f, 0.f);
    if (ns == 16) {
        // the sixteen scores in ONE round of loads, the winner's plane in a second: written as the loop below, every step was
        // a dependent trip to memory (score, wait, maybe the plane) -- sixteen to thirty-two of them in front of the barrier
        typedef int32_t Score4 __attribute__((ext_vector_type(4), aligned(4)));
        const Score4* __restrict__ sc = reinterpret_cast<const Score4*>(&m.sample_score[(size_t)l * 16]);
        const Score4 s0 = sc[0], s1 = sc[1], s2 = sc[2], s3 = sc[3];
        const int sv[16] = {s0.x, s0.y, s0.z, s0.w, s1.x, s1.y, s1.z, s1.w, s2.x, s2.y, s2.z, s2.w, s3.x, s3.y, s3.z, s3.w};
        float bw = 0.f; int bk = -1;
#pragma unroll
        for (int k = 0; k < 16; k++) { const float w = (float)sv[k]; if (w > bw) { bw = w; bk = k; } }
        if (bk >= 0) { best = m.samples[(size_t)l * 16 + bk]; best.w = bw; }
        return best;
    }
    for (int k = 0; k < ns; k++) {
        float4 th = m.samples[(size_t)l * ns + k];
        th.w = (float)m.sample_score[(size_t)l * ns + k];
        if (th.w > best.w) best = th;
    }
    return best;
}

// initDispCoeffsRansacRGBD_kernel (:112-155) / initDispCoeffsRGBD_kernel (:157-190).  Tile kernel:
// the 9 exact integer sums of every inlier go into LDS accumulators of the window's superpixels and
// are flushed once per tile into BOTH sums buffers (they must agree when the RGB-D passes start).
template <int WCAP>
__global__ __launch_bounds__(256) void k_init_disp(SegParams p, FrameMaps m, int ransac, TileOrder ord) {
    __shared__ unsigned long long w_acc[WCAP * 9 * ACC_REP];     // ACC_REP replicas (lane id) against same-address serialisation
    __shared__ float4 w_theta[WCAP];
    const TileIdx ti = tile_index(ord);                // (XCD-aware tile order: see tile_index)
    m = batch_slot(m, ti.bz);
    const int X0 = ti.bx * TILE, Y0 = ti.by * TILE;
    CellWindow win; win.init(p, X0, Y0, WCAP);
    const int32_t* __restrict__ label = m.label;
    // this thread's pixels: requested up front, in flight while the window's planes are selected
    constexpr int PX = TILE * TILE / 256;
    int pl[PX]; float pd[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {                 // (clamped addresses, no branch around the loads: see k_eval_samples)
        const int i = threadIdx.x + 256 * k;
        const int x = min(X0 + i % TILE, p.W - 1), y = min(Y0 + i / TILE, p.H - 1);
        const unsigned int q = __umul24((unsigned int)y, (unsigned int)p.W) + (unsigned int)x;
        pl[k] = label[q]; pd[k] = m.disp[q];
    }
    for (int i = threadIdx.x; i < win.size() * 9 * ACC_REP; i += blockDim.x) w_acc[i] = 0ull;
    if (ransac)
        for (int i = threadIdx.x; i < win.size(); i += blockDim.x) {
            const int l = win.label_of(i, p.gy);
            if (l >= 0) w_theta[i] = select_sample(m, l, p.nb_samples);
        }
    __syncthreads();
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int i = threadIdx.x + 256 * k;
        const int x = X0 + i % TILE, y = Y0 + i / TILE;
        if (x >= p.W || y >= p.H) continue;
        const size_t q = (size_t)y * p.W + x;
        const int l = pl[k];
        const float d = pd[k];
        bool inl = false;
        if (isfinite(d)) {
            if (ransac) {
                const int wsl = win.slot(l);
                const float4 th = wsl >= 0 ? w_theta[wsl] : select_sample(m, l, p.nb_samples);
                const float dp = (th.x * (float)x + th.y * (float)y) + th.z;
                const float dd = (dp - d) * (dp - d);
                inl = isfinite(dd) && dd < p.thresh_disp && dp > 0.f;
            } else inl = true;
        }
        m.inlier[q] = inl ? 0xff : 0;
        if (!inl) continue;
        const long long t6 = fx64((double)((float)x * d), SSF_DISP_SCALE, SSF_DISP_LIM);
        const long long t7 = fx64((double)((float)y * d), SSF_DISP_SCALE, SSF_DISP_LIM);
        const long long t8 = fx64((double)d, SSF_DISP_SCALE, SSF_DISP_LIM);
        const int ws = win.slot(l);
        if (ws >= 0) {
            unsigned long long* a = &w_acc[(ws * ACC_REP + (lane_id() & (ACC_REP - 1))) * 9];
            lds_add_i64(&a[0], x); lds_add_i64(&a[1], y); lds_add_i64(&a[2], 1);
            lds_add_i64(&a[3], (long long)x * x); lds_add_i64(&a[4], (long long)y * y); lds_add_i64(&a[5], (long long)x * y);
            lds_add_i64(&a[6], t6); lds_add_i64(&a[7], t7); lds_add_i64(&a[8], t8);
        } else {
            disp_sums_add(m.sums[0], l, x, y, d, +1);
            disp_sums_add(m.sums[1], l, x, y, d, +1);
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < win.size() * 9; i += blockDim.x) {
        long long v = 0;
#pragma unroll
        for (int r = 0; r < ACC_REP; r++) v += (long long)w_acc[((i / 9) * ACC_REP + r) * 9 + i % 9];
        if (v == 0) continue;
        const int l = win.label_of(i / 9, p.gy), j = i % 9;
        for (int b = 0; b < 2; b++) {
            const SpSums s = b ? m.sums[1] : m.sums[0];
            switch (j) {
                case 0: atomicAdd(&s.r[l].dx, (int)v); break;
                case 1: atomicAdd(&s.r[l].dy, (int)v); break;
                case 2: atomicAdd(&s.r[l].dn, (int)v); break;
                case 3: atomic_add_i64(&s.r[l].dxx, v); break;
                case 4: atomic_add_i64(&s.r[l].dyy, v); break;
                case 5: atomic_add_i64(&s.r[l].dxy, v); break;
                case 6: atomic_add_i64(&s.r[l].dxd, v); break;
                case 7: atomic_add_i64(&s.r[l].dyd, v); break;
                default: atomic_add_i64(&s.r[l].dd, v); break;
            }
        }
    }
}

// ---- plane filter: TPS_RGBD::filter, TPS_RGBD.cu:480-505; kernels TPS_RGBD_kernels.cu:510-614 ----
// All sweeps in one single-workgroup launch; Jacobi (ping-pong X0/X1), the reference's
// `x<gridSizeX` bound is kept and its out-of-range read of node S is skipped.  The 11 floats per node
// live in LDS when S fits (dynamic LDS, 44 B per node), otherwise in the global scratch.
extern __shared__ __attribute__((aligned(16))) float filt_lds[];
template <bool IN_LDS>
__global__ __launch_bounds__(1024) void k_plane_filter(SegParams p, FrameMaps m, int true_buf) {
    m = batch_slot(m, blockIdx.x);                     // one workgroup per frame of the batch
    const int S = p.S;
    float* base = IN_LDS ? filt_lds : m.filt;
    float* X0 = base; float* X1 = X0 + 3 * S; float* Z = X1 + 3 * S; float* px = Z + 3 * S; float* py = px + S;
    const SpSums sm = true_buf ? m.sums[1] : m.sums[0];
    const SpRow zero_row = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        // the final merge (mergeTPSRGBDCoeffs_kernel) of superpixel i, then its filter state
        const SpRow sp = row_from_sums(sm, i, true, zero_row);
        m.sp[i] = sp;
#pragma unroll
        for (int j = 0; j < 13; j++) m.moments[(size_t)i * 13 + j] = 0;     // accumulators of k_render_moments
        const float d0 = (sp.cx * sp.ta + sp.cy * sp.tb) + sp.tc;
        X0[3 * i] = d0; X0[3 * i + 1] = sp.ta; X0[3 * i + 2] = sp.tb;
        Z[3 * i] = d0; Z[3 * i + 1] = sp.ta; Z[3 * i + 2] = sp.tb;
        px[i] = sp.cx; py[i] = sp.cy;
    }
    if (!IN_LDS) __threadfence_block();
    __syncthreads();
    const float alpha = p.filter_alpha, beta = p.filter_beta, thr = p.filter_threshold;
    float* Xa = X0; float* Xb = X1;
    for (int it = 0; it < p.filter_iter; it++) {
        for (int idx = threadIdx.x; idx < S; idx += blockDim.x) {
            const int x = idx % p.gx, y = idx / p.gx;
            Sym3 A = sym3(alpha, 0.f, 0.f, alpha, 0.f, alpha);
            const V3 Xi = v3(Xa[3 * idx], Xa[3 * idx + 1], Xa[3 * idx + 2]);
            V3 R = scale(alpha, v3(Z[3 * idx], Z[3 * idx + 1], Z[3 * idx + 2]));
            const int v[4] = {-1, 0, 0, 1}, u[4] = {0, -1, 1, 0};
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int yy = y + v[j], xx = x + u[j];
                if (yy >= 0 && yy < p.gy && xx >= 0 && x < p.gx) {
                    const int nidx = yy * p.gx + xx;
                    if (nidx >= S) continue;
                    const V3 Xj = v3(Xa[3 * nidx], Xa[3 * nidx + 1], Xa[3 * nidx + 2]);
                    const float dx = px[idx] - px[nidx], dy = py[idx] - py[nidx];
                    const float dz = Xi.x - Xj.x;
                    if (isfinite(dz) && dz * dz < thr * thr) {
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
            Sym3 A1;
            V3 Xn = Xi;
            if (sym_inverse(A, A1)) Xn = sym_mul(A1, R);
            Xb[3 * idx] = Xn.x; Xb[3 * idx + 1] = Xn.y; Xb[3 * idx + 2] = Xn.z;
        }
        if (!IN_LDS) __threadfence_block();
        __syncthreads();
        float* t = Xa; Xa = Xb; Xb = t;
    }
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        SpRow sp = m.sp[i];
        const float X = Xa[3 * i], Y = Xa[3 * i + 1], Zz = Xa[3 * i + 2];
        sp.ta = Y; sp.tb = Zz;
        sp.tc = (X - sp.cx * Y) - sp.cy * Zz;
        m.sp[i] = sp;
    }
}

// The same filter for grids whose 44 B / node do not fit the 64 KB a launch gets without asking (S = 4800 at 1280x960: 211 KB).
// What the sweeps EXCHANGE, and the centroids every node reads of its neighbours, live in LDS: 32 B / node (154 KB at S = 4800:
// gfx950 has 160 KB per CU and a single workgroup may declare all of it); the one thing a node alone reads -- its data term Z --
// stays in its thread's registers (NPT = nodes per thread = ceil(S / 1024) <= 5).  Same operations on the same values in the same
// order as k_plane_filter: bit-identical.  (Round 4's fall-back kept the eleven floats in a global scratch and ran every sweep
// through __threadfence_block: 148 us per 4-frame launch at 1280x960, 6.6 % of BASELINE config 3's GPU time for a 4800-node stencil.)
template <int NPT>
__global__ __launch_bounds__(1024) void k_plane_filter_regs(SegParams p, FrameMaps m, int true_buf) {
    m = batch_slot(m, blockIdx.x);
    const int S = p.S;
    float* Xa = filt_lds; float* Xb = Xa + 3 * S; float* px = Xb + 3 * S; float* py = px + S;
    const SpSums sm = true_buf ? m.sums[1] : m.sums[0];
    const SpRow zero_row = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    V3 Z[NPT];
    // the final merge of every node (one node's sums and plane solve in registers at a time: no interleaving across nodes)
#pragma unroll
    for (int k = 0; k < NPT; k++) {
        const int i = threadIdx.x + 1024 * k;
        Z[k] = v3(0.f, 0.f, 0.f);
        if (i < S) {
            const SpRow sp = row_from_sums(sm, i, true, zero_row);
            m.sp[i] = sp;
#pragma unroll
            for (int j = 0; j < 13; j++) m.moments[(size_t)i * 13 + j] = 0;
            const float d0 = (sp.cx * sp.ta + sp.cy * sp.tb) + sp.tc;
            Xa[3 * i] = d0; Xa[3 * i + 1] = sp.ta; Xa[3 * i + 2] = sp.tb;
            Z[k] = v3(d0, sp.ta, sp.tb);
            px[i] = sp.cx; py[i] = sp.cy;
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();
    const float alpha = p.filter_alpha, beta = p.filter_beta, thr = p.filter_threshold;
    for (int it = 0; it < p.filter_iter; it++) {
#pragma unroll
        for (int k = 0; k < NPT; k++) {
            const int idx = threadIdx.x + 1024 * k;
            if (idx < S) {
                const int x = idx % p.gx, y = idx / p.gx;
                Sym3 A = sym3(alpha, 0.f, 0.f, alpha, 0.f, alpha);
                const V3 Xi = v3(Xa[3 * idx], Xa[3 * idx + 1], Xa[3 * idx + 2]);
                V3 R = scale(alpha, Z[k]);
                const int v[4] = {-1, 0, 0, 1}, u[4] = {0, -1, 1, 0};
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int yy = y + v[j], xx = x + u[j];
                    if (yy >= 0 && yy < p.gy && xx >= 0 && x < p.gx) {
                        const int nidx = yy * p.gx + xx;
                        if (nidx >= S) continue;
                        const V3 Xj = v3(Xa[3 * nidx], Xa[3 * nidx + 1], Xa[3 * nidx + 2]);
                        const float dx = px[idx] - px[nidx], dy = py[idx] - py[nidx];
                        const float dz = Xi.x - Xj.x;
                        if (isfinite(dz) && dz * dz < thr * thr) {
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
                Sym3 A1;
                V3 Xn = Xi;
                if (sym_inverse(A, A1)) Xn = sym_mul(A1, R);
                Xb[3 * idx] = Xn.x; Xb[3 * idx + 1] = Xn.y; Xb[3 * idx + 2] = Xn.z;
            }
            __builtin_amdgcn_sched_barrier(0);
        }
        __syncthreads();
        float* t = Xa; Xa = Xb; Xb = t;
    }
#pragma unroll 1
    for (int i = threadIdx.x; i < S; i += 1024) {
        SpRow sp = m.sp[i];
        const float X = Xa[3 * i], Y = Xa[3 * i + 1], Zz = Xa[3 * i + 2];
        sp.ta = Y; sp.tb = Zz;
        sp.tc = (X - sp.cx * Y) - sp.cy * Zz;
        m.sp[i] = sp;
    }
}

// The filter on MANY workgroups per frame, without any exchange between them (round 5).  F Jacobi sweeps make a node's final
// state a function of the nodes within graph distance F of it, so a workgroup that owns a CORE tile of PF_TX x PF_TY nodes
// computes everything within F of the core redundantly (the merge of those nodes included) and nobody waits for anybody: after
// sweep s the nodes at distance > F - s from the core hold values that differ from the global iteration's -- they miss a
// neighbour outside the window -- and by construction none of them is read on the way to the core.  The graph is the
// reference's, typo included (TPS_RGBD_kernels.cu:583: `x<gridSizeX` where `xx<gridSizeX` was meant): the right-hand
// neighbour of a node in the LAST column is the FIRST node of the next row.  A window that touches the last column therefore
// takes a second rectangle along the first columns (W2: columns [0, F), one row down), whose nodes are at distance >= 1 + column
// from the core; nothing in the first columns reads the last one, so the closure ends there.  One node per thread, 32 B per
// node in LDS (two states, the centroid), the data term in a register; same operations on the same values as
// k_plane_filter: bit-identical (tests/test_parity_gpu.py::test_plane_filter_at_every_grid_size).
// Single-workgroup forms: 26.6 us per 8-frame launch at 640x480 (1200 nodes), 148 us (global scratch) / 93 us (160 KB of LDS)
// per 4-frame launch at 1280x960 (4800 nodes) -- the merge's IEEE divisions and plane solves of a whole frame on ONE compute unit.
#define PF_TX 16
#define PF_TY 12
struct PfWindow {
    int x0, y0, w, h;              // W1: core +- F, clipped to the grid
    int x2n, y2, h2;               // W2: columns [0, x2n) x rows [y2, y2 + h2) (x2n = 0: none)
    int n1, n2;
    __device__ __forceinline__ int slot(int x, int y) const {
        const unsigned int lx = (unsigned int)(x - x0), ly = (unsigned int)(y - y0);
        if (lx < (unsigned int)w && ly < (unsigned int)h) return (int)ly * w + (int)lx;
        const unsigned int l2 = (unsigned int)(y - y2);
        if ((unsigned int)x < (unsigned int)x2n && l2 < (unsigned int)h2) return n1 + (int)l2 * x2n + x;
        return -1;
    }
};
__host__ __device__ __forceinline__ PfWindow pf_window(int gx, int gy, int tx, int ty, int F) {
    PfWindow wd;
    const int cx0 = tx * PF_TX, cy0 = ty * PF_TY, cx1 = min(cx0 + PF_TX, gx) - 1, cy1 = min(cy0 + PF_TY, gy) - 1;
    wd.x0 = max(cx0 - F, 0); wd.y0 = max(cy0 - F, 0);
    const int x1 = min(cx1 + F, gx - 1), y1 = min(cy1 + F, gy - 1);
    wd.w = x1 - wd.x0 + 1; wd.h = y1 - wd.y0 + 1; wd.n1 = wd.w * wd.h;
    wd.x2n = 0; wd.y2 = 0; wd.h2 = 0;
    if (x1 == gx - 1 && wd.x0 > 0 && F > 0) {                // touches the last column without holding the first
        // rows: a node of the last column at distance d from the core has its wrap neighbour (0, y + 1) at distance d + 1, whose
        // column neighbours matter up to F - d - 1 rows away: rows [cy0 - F + 2, cy1 + F] whatever d (one more row on top kept);
        // NOT "the window's rows + 1": a window clipped by the top of the grid still needs row 0 (found by the test with
        // filter_threshold 1.0, where no neighbour is gated away)
        wd.x2n = min(F, wd.x0); wd.y2 = max(cy0 - F + 1, 0);
        wd.h2 = min(cy1 + F + 1, gy - 1) - wd.y2 + 1;
        if (wd.h2 <= 0) { wd.x2n = 0; wd.h2 = 0; }
    }
    wd.n2 = wd.x2n * wd.h2;
    return wd;
}
__global__ __launch_bounds__(1024) void k_plane_filter_tiled(SegParams p, FrameMaps m, int true_buf) {
    m = batch_slot(m, blockIdx.z);
    const int S = p.S, F = max(p.filter_iter, 0);
    const PfWindow wd = pf_window(p.gx, p.gy, (int)blockIdx.x, (int)blockIdx.y, F);
    const int nn = wd.n1 + wd.n2;                              // <= blockDim.x (launch_plane_filter)
    float* Xa = filt_lds; float* Xb = Xa + 3 * nn; float* px = Xb + 3 * nn; float* py = px + nn;
    const SpSums sm = true_buf ? m.sums[1] : m.sums[0];
    const SpRow zero_row = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    const int t = threadIdx.x;
    const bool have = t < nn;
    int x = 0, y = 0;
    if (have) {
        if (t < wd.n1) { const int ly = t / wd.w; x = wd.x0 + (t - ly * wd.w); y = wd.y0 + ly; }
        else { const int q = t - wd.n1, ly = q / wd.x2n; x = q - ly * wd.x2n; y = wd.y2 + ly; }
    }
    const int idx = y * p.gx + x;
    const int cx0 = (int)blockIdx.x * PF_TX, cy0 = (int)blockIdx.y * PF_TY;
    const bool core = have && x >= cx0 && x < cx0 + PF_TX && y >= cy0 && y < cy0 + PF_TY;
    V3 Z = v3(0.f, 0.f, 0.f);
    SpRow sp = zero_row;
    if (have) {
        sp = row_from_sums(sm, idx, true, zero_row);           // the final merge (mergeTPSRGBDCoeffs_kernel) of this node
        const float d0 = (sp.cx * sp.ta + sp.cy * sp.tb) + sp.tc;
        Xa[3 * t] = d0; Xa[3 * t + 1] = sp.ta; Xa[3 * t + 2] = sp.tb;
        Z = v3(d0, sp.ta, sp.tb);
        px[t] = sp.cx; py[t] = sp.cy;
        if (core) {
#pragma unroll
            for (int j = 0; j < 13; j++) m.moments[(size_t)idx * 13 + j] = 0;     // accumulators of k_render_moments
        }
    }
    __syncthreads();
    const float alpha = p.filter_alpha, beta = p.filter_beta, thr = p.filter_threshold;
    // the neighbours' slots (-1: no such node in the reference's graph, or outside the window -- then this node is at distance F
    // from the core and nobody reads what it computes)
    int ns[4] = {-1, -1, -1, -1};
    if (have) {
        const int v[4] = {-1, 0, 0, 1}, u[4] = {0, -1, 1, 0};
#pragma unroll
        for (int j = 0; j < 4; j++) {
            const int yy = y + v[j], xx = x + u[j];
            if (yy >= 0 && yy < p.gy && xx >= 0 && x < p.gx) {
                const int nidx = yy * p.gx + xx;
                if (nidx >= S) continue;
                const int ny = nidx / p.gx, nx = nidx - ny * p.gx;          // (xx == gx: the first node of the next row)
                ns[j] = wd.slot(nx, ny);
            }
        }
    }
    for (int it = 0; it < F; it++) {
        if (have) {
            Sym3 A = sym3(alpha, 0.f, 0.f, alpha, 0.f, alpha);
            const V3 Xi = v3(Xa[3 * t], Xa[3 * t + 1], Xa[3 * t + 2]);
            V3 R = scale(alpha, Z);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int n = ns[j];
                if (n < 0) continue;
                const V3 Xj = v3(Xa[3 * n], Xa[3 * n + 1], Xa[3 * n + 2]);
                const float dx = px[t] - px[n], dy = py[t] - py[n];
                const float dz = Xi.x - Xj.x;
                if (isfinite(dz) && dz * dz < thr * thr) {
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
            Sym3 A1;
            V3 Xn = Xi;
            if (sym_inverse(A, A1)) Xn = sym_mul(A1, R);
            Xb[3 * t] = Xn.x; Xb[3 * t + 1] = Xn.y; Xb[3 * t + 2] = Xn.z;
        }
        __syncthreads();
        float* q = Xa; Xa = Xb; Xb = q;
    }
    if (core) {
        const float X = Xa[3 * t], Y = Xa[3 * t + 1], Zz = Xa[3 * t + 2];
        sp.ta = Y; sp.tb = Zz;
        sp.tc = (X - sp.cx * Y) - sp.cy * Zz;
        m.sp[idx] = sp;
    }
}

// ---- plane depth + supersurfel moments -----------------------------------------------------------
// renderDepthImage_kernel (TPS_RGBD_kernels.cu:469-508) fused with computeSupersurfelCoeffs
// (supersurfel_fusion_kernels.cu:113-167): one read of the label tile serves the depth render, the
// boundary test and the 13 moment sums (fixed point 2^24, exact), which are accumulated with LDS
// integer atomics per window superpixel and flushed once per tile.
// replicas of the 13 moment accumulators of a window cell (lane id): at a window capacity of 64 cells 8 / 4 / 2 replicas measured
// 69.6 / 60.9 / 67.0 us per 8-frame launch (8: 53 KB of LDS, two workgroups per compute unit); at 36 cells 8 replicas are 37 KB and
// four workgroups: 47.4-47.5 against 48.9-49.8 us with 4, 58.9-59.7 with 2 (round 6, profiles/tile_kernels_r06.txt)
#ifndef MOM_REP
#define MOM_REP(wcap) ((wcap) <= WIN_SMALL ? 8 : 4)
#endif
template <int WCAP>
__global__ __launch_bounds__(256) void k_render_moments(SegParams p, Cam cam, FrameMaps m, TileOrder ord) {
    __shared__ int tile[TW * TW];
    __shared__ __attribute__((aligned(16))) SpRow w_row[WCAP];
    __shared__ unsigned long long w_acc[WCAP * 13 * MOM_REP(WCAP)];    // MOM_REP replicas (lane id) against same-address serialisation
    const TileIdx ti = tile_index(ord);                // (XCD-aware tile order: see tile_index)
    m = batch_slot(m, ti.bz);
    const int X0 = ti.bx * TILE, Y0 = ti.by * TILE;
    CellWindow win; win.init(p, X0, Y0, WCAP);
    // Everything the workgroup needs from memory is requested first -- the window's rows, the gamma table, this thread's
    // pixels, the label tile -- and stored to LDS afterwards: ONE round trip.  (Staged piece by piece, each piece was a
    // trip of its own: four of them in front of the first useful instruction.)
    static_assert(WCAP <= 256, "one window row per thread");
    const int wl = (int)threadIdx.x < win.size() ? win.label_of(threadIdx.x, p.gy) : -1;
    float4 r0 = make_float4(0.f, 0.f, 0.f, 0.f), r1 = r0, r2 = r0;
    {
        const float4* __restrict__ q = reinterpret_cast<const float4*>(&m.sp[wl >= 0 ? wl : 0]);
        r0 = q[0]; r1 = q[1]; r2 = q[2];                      // (unconditional: row 0 for the lanes without a window cell)
    }
    const float lut = m.srgb_lut[threadIdx.x];
    // this thread's pixels: inlier flag and colour (clamped addresses, used only for pixels inside the image)
    constexpr int PX = TILE * TILE / 256;
    unsigned char pin[PX]; uint32_t prgba[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int i = threadIdx.x + 256 * k;
        const int x = min(X0 + i % TILE, p.W - 1), y = min(Y0 + i / TILE, p.H - 1);
        const unsigned int q = __umul24((unsigned int)y, (unsigned int)p.W) + (unsigned int)x;
        pin[k] = m.inlier[q]; prgba[k] = m.rgba[q];
    }
    const TileRegs treg = tile_request(m.label, X0, Y0, p.W, p.H);
    for (int i = threadIdx.x; i < win.size() * 13 * MOM_REP(WCAP); i += blockDim.x) w_acc[i] = 0ull;
    // the gamma table in LDS: three lookups per pixel would otherwise be global loads queued behind this thread's
    // stores (the maps may alias as far as the compiler knows)
    __shared__ float s_lut[256];
    s_lut[threadIdx.x] = lut;
    if (wl >= 0) { float4* w = reinterpret_cast<float4*>(&w_row[threadIdx.x]); w[0] = r0; w[1] = r1; w[2] = r2; }
    tile_commit(tile, treg);
    __syncthreads();
    float out_depth[PX]; int out_label[PX];
#pragma unroll
    for (int k = 0; k < PX; k++) {
        const int i = threadIdx.x + 256 * k;
        const int lx = i % TILE, ly = i / TILE;
        const int x = X0 + lx, y = Y0 + ly;
        out_label[k] = -1; out_depth[k] = 0.f;
        if (x >= p.W || y >= p.H) continue;
        const int label = tile[(ly + 1) * TW + lx + 1];
        const int ws = win.slot(label);
        SpRow sp = w_row[ws >= 0 ? ws : 0];      // (LDS read always, global row on top when outside the window: a select
        if (ws < 0) sp = ld_global_row(&m.sp[label]);     //  between an LDS and a global ADDRESS would be one FLAT load)
        const float disp = ((float)x * sp.ta + (float)y * sp.tb) + sp.tc;
        const float depth = 1.f / disp;
        out_label[k] = label; out_depth[k] = depth;
        const int bound = tile_boundary(tile, lx + 1, ly + 1);
        if (!(pin[k] && isfinite(depth) && depth > 0.0f && bound == 0)) continue;
        const V3 pos = v3(((float)x - cam.cx) * depth / cam.fx, ((float)y - cam.cy) * depth / cam.fy, depth);
        const uint32_t px = prgba[k];
        const V3 lab = rgb8_to_lab(s_lut, px & 255u, (px >> 8) & 255u, (px >> 16) & 255u);
        const Sym3 c = sym_outer(pos);
        const float v[12] = {pos.x, pos.y, pos.z, lab.x, lab.y, lab.z, c.xx, c.xy, c.xz, c.yy, c.yz, c.zz};
        if (ws >= 0) {
            unsigned long long* a = &w_acc[(ws * MOM_REP(WCAP) + (lane_id() & (MOM_REP(WCAP) - 1))) * 13];
#pragma unroll
            for (int j = 0; j < 12; j++) lds_add_i64(&a[j], fx64((double)v[j], SSF_MOM_SCALE, SSF_MOM_LIM));
            lds_add_i64(&a[12], 1);
        } else {
#pragma unroll
            for (int j = 0; j < 12; j++) atomic_add_i64(&m.moments[(size_t)label * 13 + j], fx64((double)v[j], SSF_MOM_SCALE, SSF_MOM_LIM));
            atomic_add_i64(&m.moments[(size_t)label * 13 + 12], 1);
        }
    }
    // the rendered depth and the packed (label, depth) table of ICP / association: stored after all loads
#pragma unroll
    for (int k = 0; k < PX; k++) {
        if (out_label[k] < 0) continue;
        const int i = threadIdx.x + 256 * k;
        const size_t q = (size_t)(Y0 + i / TILE) * p.W + (X0 + i % TILE);
        m.plane_depth[q] = out_depth[k];
        m.pix2[q] = make_uint2((uint32_t)out_label[k], __float_as_uint(out_depth[k]));
    }
    __syncthreads();
    for (int i = threadIdx.x; i < win.size() * 13; i += blockDim.x) {
        long long v = 0;
#pragma unroll
        for (int r = 0; r < MOM_REP(WCAP); r++) v += (long long)w_acc[((i / 13) * MOM_REP(WCAP) + r) * 13 + i % 13];
        if (v != 0) atomic_add_i64(&m.moments[(size_t)win.label_of(i / 13, p.gy) * 13 + i % 13], v);
    }
}

// computeSupersurfels, supersurfel_fusion_kernels.cu:169-224 (+ the MOD mask hook).  64 superpixels per workgroup.  One
// thread per superpixel ran a serial chain of ~3000 dependent instructions (two ten-fold normalised squarings for the
// principal frame, double-precision roots for the colour: 11 us for 1200 threads); the three pieces are independent, so
// they run side by side with identical arithmetic: waves 0-1, two lanes per superpixel, iterate towards the largest /
// smallest axis; wave 2 converts the colour; wave 2 then assembles and stores the supersurfel.
#define FIN_PER_WG 64
__global__ __launch_bounds__(256) void k_finalize_surfels(SegParams p, FrameMaps m, SurfelSoA f, float zmin, float zmax, int stamp0,
                                                          const uint8_t* __restrict__ dyn_mask, unsigned mask_bits,
                                                          unsigned long long* __restrict__ best, uint8_t* __restrict__ matched) {
    __shared__ float s_axis[2][FIN_PER_WG][3];
    const int wv = threadIdx.x >> 6, l = threadIdx.x & 63;
    const int kk = wv < 2 ? wv * 32 + (l >> 1) : l;
    const int k = blockIdx.x * FIN_PER_WG + kk;
    const bool active = k < p.S && wv < 3;
    const int fb = blockIdx.y;
    const size_t off = (size_t)fb * m.slab;
    m = batch_slot(m, fb); f = batch_slot(f, off);
    best = slab_shift(best, off); matched = slab_shift(matched, off);
    dyn_mask = ((mask_bits >> fb) & 1u) ? slab_shift(dyn_mask, off) : nullptr;
    const int stamp = stamp0 + fb;
    float sum[12]; float conf = 0.f;
#pragma unroll
    for (int j = 0; j < 12; j++) sum[j] = 0.f;
    if (active) {
        const long long* a = &m.moments[(size_t)k * 13];
        const double inv = 1.0 / SSF_MOM_SCALE;
#pragma unroll
        for (int j = 0; j < 12; j++) sum[j] = (float)((double)a[j] * inv);
        conf = (float)a[12];
    }
    V3 pos = v3(sum[0], sum[1], sum[2]), col = v3(sum[3], sum[4], sum[5]);
    Sym3 shape = sym3(sum[6], sum[7], sum[8], sum[9], sum[10], sum[11]);
    const float z = pos.z / conf;
    const bool valid = active && isfinite(z) && conf > 100.0f && z > zmin && z < zmax;
    if (valid) {
        pos = v3(pos.x / conf, pos.y / conf, z);
        shape = sym_sub(sym_div(shape, conf), sym_outer(pos));
    }
    V3 lab = v3(0.f, 0.f, 0.f);
    if (wv < 2) {
        if (valid) {
            const V3 axis = principal_power(principal_start(shape, (l & 1) == 1));
            s_axis[l & 1][kk][0] = axis.x; s_axis[l & 1][kk][1] = axis.y; s_axis[l & 1][kk][2] = axis.z;
        }
    } else if (wv == 2 && active) {
        if (valid) col = lab_to_rgb(v3(col.x / conf, col.y / conf, col.z / conf));
        lab = rgb_to_lab(col);
    }
    __syncthreads();
    if (wv != 2 || !active) return;
    best[k] = SSF_NO_MATCH; matched[k] = 0;         // association tables of this frame (findBestMatches init)
    M3 vecs = m3(v3(0, 0, 0), v3(0, 0, 0), v3(0, 0, 0));
    float d0 = 0.f, d1 = 0.f; int s0 = 0, s1 = 0;
    if (valid) {
        V3 vals;
        principal_finish(shape, v3(s_axis[0][kk][0], s_axis[0][kk][1], s_axis[0][kk][2]),
                         v3(s_axis[1][kk][0], s_axis[1][kk][1], s_axis[1][kk][2]), vecs, vals);
        d0 = vals.x; d1 = vals.y; s0 = stamp; s1 = stamp;
        if (vals.x / vals.y > 50.0f) conf = -1.0f;
    } else
        conf = -1.0f;
    if (dyn_mask && dyn_mask[k]) conf = -1.0f;
    f.pos[3 * k] = pos.x; f.pos[3 * k + 1] = pos.y; f.pos[3 * k + 2] = pos.z;
    f.col[3 * k] = col.x; f.col[3 * k + 1] = col.y; f.col[3 * k + 2] = col.z;
    f.lab[3 * k] = lab.x; f.lab[3 * k + 1] = lab.y; f.lab[3 * k + 2] = lab.z;
    f.stamps[2 * k] = s0; f.stamps[2 * k + 1] = s1;
    f.r0[3 * k] = vecs.r0.x; f.r0[3 * k + 1] = vecs.r0.y; f.r0[3 * k + 2] = vecs.r0.z;
    f.r1[3 * k] = vecs.r1.x; f.r1[3 * k + 1] = vecs.r1.y; f.r1[3 * k + 2] = vecs.r1.z;
    f.r2[3 * k] = vecs.r2.x; f.r2[3 * k + 1] = vecs.r2.y; f.r2[3 * k + 2] = vecs.r2.z;
    f.shape[6 * k] = shape.xx; f.shape[6 * k + 1] = shape.xy; f.shape[6 * k + 2] = shape.xz;
    f.shape[6 * k + 3] = shape.yy; f.shape[6 * k + 4] = shape.yz; f.shape[6 * k + 5] = shape.zz;
    f.dims[2 * k] = d0; f.dims[2 * k + 1] = d1;
    f.conf[k] = conf;
    m.fpack[4 * k] = make_float4(conf, lab.x, lab.y, lab.z);          // one 64-byte line per frame supersurfel
    m.fpack[4 * k + 1] = make_float4(vecs.r2.x, vecs.r2.y, vecs.r2.z, 0.f);
    m.fpack[4 * k + 2] = make_float4(pos.x, pos.y, pos.z, 0.f);
}

// ---- a frame extracted elsewhere (ssf_submit_frame_tables: the extract stage dealt over the ranks of a sharded map) -------------
// What the track chain reads of a frame -- the supersurfel SoA with its Lab cache, the 64-byte lines of ICP / association (fpack),
// the packed (label, plane depth) table (pix2), the association tables' initial state -- rebuilt from the three things another
// rank ships: label map and plane depth (already in this slot's maps) and the S frame supersurfels in the reference's layout
// (`wire`: positions 3S | colours 3S | stamps 2S | orientations 9S | shapes 6S | dims 2S | confidences S).  Every value is a
// copy or a pure function of those (lab = rgb_to_lab(colour), as k_finalize_surfels computes it): bit-identical to a local extract.
__global__ __launch_bounds__(256) void k_import_frame(int P, int S, FrameMaps m, SurfelSoA f, const float* __restrict__ wire,
                                                      unsigned long long* __restrict__ best, uint8_t* __restrict__ matched) {
    {                                                                   // slot blockIdx.y of the batch context
        const size_t off = (size_t)blockIdx.y * m.slab;
        m = batch_slot(m, blockIdx.y); f = batch_slot(f, off); wire = slab_shift(wire, off);
        best = slab_shift(best, off); matched = slab_shift(matched, off);
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < P) m.pix2[i] = make_uint2((uint32_t)m.label[i], __float_as_uint(m.plane_depth[i]));
    if (i >= S) return;
    const int k = i;
    const float* wpos = wire; const float* wcol = wpos + 3 * (size_t)S; const int32_t* wst = reinterpret_cast<const int32_t*>(wcol + 3 * (size_t)S);
    const float* wori = reinterpret_cast<const float*>(wst + 2 * (size_t)S); const float* wsh = wori + 9 * (size_t)S;
    const float* wd = wsh + 6 * (size_t)S; const float* wc = wd + 2 * (size_t)S;
    const V3 pos = v3(wpos[3 * k], wpos[3 * k + 1], wpos[3 * k + 2]), col = v3(wcol[3 * k], wcol[3 * k + 1], wcol[3 * k + 2]);
    const V3 lab = rgb_to_lab(col);
    const V3 r2 = v3(wori[9 * k + 6], wori[9 * k + 7], wori[9 * k + 8]);
    const float conf = wc[k];
    f.pos[3 * k] = pos.x; f.pos[3 * k + 1] = pos.y; f.pos[3 * k + 2] = pos.z;
    f.col[3 * k] = col.x; f.col[3 * k + 1] = col.y; f.col[3 * k + 2] = col.z;
    f.lab[3 * k] = lab.x; f.lab[3 * k + 1] = lab.y; f.lab[3 * k + 2] = lab.z;
    f.stamps[2 * k] = wst[2 * k]; f.stamps[2 * k + 1] = wst[2 * k + 1];
#pragma unroll
    for (int j = 0; j < 3; j++) { f.r0[3 * k + j] = wori[9 * k + j]; f.r1[3 * k + j] = wori[9 * k + 3 + j]; f.r2[3 * k + j] = wori[9 * k + 6 + j]; }
#pragma unroll
    for (int j = 0; j < 6; j++) f.shape[6 * k + j] = wsh[6 * k + j];
    f.dims[2 * k] = wd[2 * k]; f.dims[2 * k + 1] = wd[2 * k + 1];
    f.conf[k] = conf;
    m.fpack[4 * k] = make_float4(conf, lab.x, lab.y, lab.z);
    m.fpack[4 * k + 1] = make_float4(r2.x, r2.y, r2.z, 0.f);
    m.fpack[4 * k + 2] = make_float4(pos.x, pos.y, pos.z, 0.f);
    best[k] = SSF_NO_MATCH; matched[k] = 0;
}

// the S frame supersurfels of slot blockIdx.y in the wire layout of k_import_frame (what the rank that extracted a batch ships)
__global__ __launch_bounds__(256) void k_export_rows(int S, SurfelSoA f, float* __restrict__ wire, size_t slab) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= S) return;
    const size_t off = (size_t)blockIdx.y * slab;
    f = batch_slot(f, off); wire = slab_shift(wire, off);
    float* wpos = wire; float* wcol = wpos + 3 * (size_t)S; int32_t* wst = reinterpret_cast<int32_t*>(wcol + 3 * (size_t)S);
    float* wori = reinterpret_cast<float*>(wst + 2 * (size_t)S); float* wsh = wori + 9 * (size_t)S;
    float* wd = wsh + 6 * (size_t)S; float* wc = wd + 2 * (size_t)S;
#pragma unroll
    for (int j = 0; j < 3; j++) {
        wpos[3 * k + j] = f.pos[3 * k + j]; wcol[3 * k + j] = f.col[3 * k + j];
        wori[9 * k + j] = f.r0[3 * k + j]; wori[9 * k + 3 + j] = f.r1[3 * k + j]; wori[9 * k + 6 + j] = f.r2[3 * k + j];
    }
    wst[2 * k] = f.stamps[2 * k]; wst[2 * k + 1] = f.stamps[2 * k + 1];
#pragma unroll
    for (int j = 0; j < 6; j++) wsh[6 * k + j] = f.shape[6 * k + j];
    wd[2 * k] = f.dims[2 * k]; wd[2 * k + 1] = f.dims[2 * k + 1];
    wc[k] = f.conf[k];
}
__global__ __launch_bounds__(256) void k_boundary_map(SegParams p, const int32_t* __restrict__ label, int32_t* __restrict__ out) {
    __shared__ int tile[TW * TW];
    const int X0 = blockIdx.x * TILE, Y0 = blockIdx.y * TILE;
    load_label_tile(tile, label, X0, Y0, p.W, p.H);
    __syncthreads();
    for (int i = threadIdx.x; i < TILE * TILE; i += blockDim.x) {
        const int x = X0 + i % TILE, y = Y0 + i / TILE;
        if (x < p.W && y < p.H) out[(size_t)y * p.W + x] = tile_boundary(tile, i % TILE + 1, i / TILE + 1);
    }
}

// renderBoundaryImage_kernel, TPS_RGBD_kernels.cu:616-644 (computeSuperpixelSegIm's image)
__global__ __launch_bounds__(256) void k_preview(int W, int H, const int32_t* __restrict__ label, const uint32_t* __restrict__ rgba,
                                                 uint8_t* __restrict__ out) {
    const int x = blockIdx.x * 64 + (threadIdx.x & 63), y = blockIdx.y * 4 + (threadIdx.x >> 6);
    if (x >= W || y >= H) return;
    const size_t p = (size_t)y * W + x;
    const int index = label[p];
    uint8_t b0 = 255, b1 = 255, b2 = 255;
    if (!(x < W - 1 && y < H - 1 && (label[p + 1] != index || label[p + W + 1] != index))) {
        const uint32_t c = rgba[p];
        b0 = (uint8_t)(0.8f * (float)((c >> 16) & 255u)); b1 = (uint8_t)(0.8f * (float)((c >> 8) & 255u)); b2 = (uint8_t)(0.8f * (float)(c & 255u));
    }
    out[3 * p] = b0; out[3 * p + 1] = b1; out[3 * p + 2] = b2;
}
void launch_preview(hipStream_t st, int W, int H, const int32_t* label, const uint32_t* rgba, uint8_t* out) {
    hipLaunchKernelGGL(k_preview, dim3((W + 63) / 64, (H + 3) / 4), dim3(256), 0, st, W, H, label, rgba, out);
}

// ---- depth pre-filter ("next" row f3) ---------------------------------------------------------------
// Out-of-place restatement of cv::cuda::bilateralFilter(depth, depth, -1, sigma_color, sigma_space)
// (supersurfel_fusion.cu:180; OpenCV cudaimgproc is third party, algorithm as published): circular
// support of radius round(1.5 sigma_space), BORDER_REFLECT_101, taps in row-major order.  The tile and
// its halo are staged in LDS once; every pixel then reads its (2r+1)^2 window from LDS.
#define BIL_RMAX 16
#define BIL_TILE 16      // 16 x 16 outputs per workgroup, one per thread: 1200 workgroups at 640x480 (4-5 waves per SIMD to
                         // hide the dependent chain of the specified exp; with 32 x 32 tiles and four pixels per thread a
                         // SIMD held one wave and the kernel took 0.4 ms)
__device__ __forceinline__ int reflect101(int i, int n) {
    if (n == 1) return 0;
    while (i < 0 || i >= n) i = (i < 0) ? -i : 2 * (n - 1) - i;
    return i;
}
// (every frame of an extract batch in one launch: frame = blockIdx.z, input = the caller's / the upload ring's buffer of that
// frame, output = out0 + frame * slab bytes)
__global__ __launch_bounds__(256) void k_bilateral(BatchIn bin, float* __restrict__ out0, size_t slab, int W, int H,
                                                   int radius, float ss, float sc) {
    const float* __restrict__ in = batch_pick(bin.depth, (int)blockIdx.z);
    float* __restrict__ out = slab_shift(out0, (size_t)blockIdx.z * slab);
    __shared__ float tile[(BIL_TILE + 2 * BIL_RMAX) * (BIL_TILE + 2 * BIL_RMAX)];
    __shared__ int s_ext[2 * BIL_RMAX + 1];            // half-width of the circular support in row dy
    const int X0 = blockIdx.x * BIL_TILE, Y0 = blockIdx.y * BIL_TILE;
    const int tw = BIL_TILE + 2 * radius;
    const bool staged = radius <= BIL_RMAX;
    if (staged) {
        for (int i = threadIdx.x; i < tw * tw; i += blockDim.x) {
            const int gx = reflect101(X0 - radius + i % tw, W), gy = reflect101(Y0 - radius + i / tw, H);
            tile[i] = in[(size_t)gy * W + gx];
        }
        if ((int)threadIdx.x <= 2 * radius) {
            const int dy = (int)threadIdx.x - radius;
            int e = 0;
            while ((e + 1) * (e + 1) + dy * dy <= radius * radius) e++;
            s_ext[threadIdx.x] = e;
        }
    }
    __syncthreads();
    const int lx = threadIdx.x % BIL_TILE, ly = threadIdx.x / BIL_TILE;
    const int x = X0 + lx, y = Y0 + ly;
    if (x >= W || y >= H) return;
    const float r2 = (float)(radius * radius);
    float sum1 = 0.f, sum2 = 0.f;
    if (staged) {
        // taps inside the circle in row-major order: the same sequence of additions as the definition's double loop
        const float center = tile[(ly + radius) * tw + lx + radius];
        for (int dy = -radius; dy <= radius; dy++) {
            const int e = s_ext[dy + radius];
            const float* __restrict__ row = &tile[(ly + radius + dy) * tw + lx + radius];
            const float dy2 = (float)(dy * dy);
            for (int dx = -e; dx <= e; dx++) {
                const float space2 = (float)(dx * dx) + dy2;     // = (float)(dx*dx + dy*dy): small integers, exact either way
                const float v = row[dx];
                const float dv = fabsf(v - center);
                const float w = exp_neg_spec(space2 * ss + (dv * dv) * sc);
                sum1 = sum1 + w * v;
                sum2 = sum2 + w;
            }
        }
    } else {
        const float center = in[(size_t)y * W + x];
        for (int dy = -radius; dy <= radius; dy++)
            for (int dx = -radius; dx <= radius; dx++) {
                const float space2 = (float)(dx * dx + dy * dy);
                if (space2 > r2) continue;
                const float v = in[(size_t)reflect101(y + dy, H) * W + reflect101(x + dx, W)];
                const float dv = fabsf(v - center);
                const float w = exp_neg_spec(space2 * ss + (dv * dv) * sc);
                sum1 = sum1 + w * v;
                sum2 = sum2 + w;
            }
    }
    out[(size_t)y * W + x] = sum1 / sum2;
}
// The same filter for the radius the reference's call produces (sigma_space 4.5 -> 7), the kernel is bound by its
// arithmetic (149 taps x the ~30 operations of the specified exp): rows and extents are compile-time constants (no
// loop bookkeeping, no extent table), and two taps at a time go through the packed fp32 pipe (v_pk_mul_f32 / v_pk_add_f32:
// two IEEE operations per lane and instruction, each rounded exactly like the scalar one) -- the same operations on the
// same operands in the same order per tap, so the bits do not change; (dv, dv) -> dv * dv drops the fabsf.
typedef float v2f __attribute__((ext_vector_type(2)));
__device__ __forceinline__ v2f exp_neg_spec2(v2f x) {                  // exp_neg_spec (ssf_math.hpp) on two values
    // (the scalar function also clamps x > 0 to 0: here x = space2 * ss + dv^2 * sc with ss, sc < 0 is never positive)
    const bool live0 = x.x > -87.0f, live1 = x.y > -87.0f;
    const v2f t = x * 1.44269502f;
    v2f kf; kf.x = rintf(t.x); kf.y = rintf(t.y);
    const v2f r = (x - kf * 0.693359375f) - kf * -2.12194440e-4f;
    v2f p = {1.98412701e-4f, 1.98412701e-4f};
    p = p * r + 1.38888892e-3f;
    p = p * r + 8.33333377e-3f;
    p = p * r + 4.16666679e-2f;
    p = p * r + 0.166666672f;
    p = p * r + 0.5f;
    p = p * r + 1.0f;
    p = p * r + 1.0f;
    v2f two_k;
    two_k.x = __uint_as_float((uint32_t)(127 + (int)kf.x) << 23); two_k.y = __uint_as_float((uint32_t)(127 + (int)kf.y) << 23);
    v2f w = p * two_k;
    w.x = live0 ? w.x : 0.0f; w.y = live1 ? w.y : 0.0f;
    return w;
}
// one tap pair: weights of taps dx, dx + 1 of row DY (the definition's operations, two at a time)
template <int DY>
__device__ __forceinline__ v2f bilateral_pair_w(const v2f v, const v2f c2, int dx, float ss, float sc) {
    const v2f space2 = {(float)(dx * dx + DY * DY), (float)((dx + 1) * (dx + 1) + DY * DY)};
    const v2f dv = v - c2;
    return exp_neg_spec2(space2 * ss + (dv * dv) * sc);
}
// Round 4: the row in GROUPS OF FOUR TAPS, each group ordered behind the previous one by data dependences.  Left to itself the
// compiler computed every weight of the kernel before the first sum and kept the tile in registers: 208 vector registers (two
// waves per SIMD), 833 `s_nop` behind packed operations whose dependent successor had nothing to hide behind (a v_pk_*_f32 result is
// not forwarded to the next instruction; the Horner chain of the specified exp is exactly such a chain), and the odd tap of every
// row behind a branch of its own (exp_neg_spec's early exit).  Now: 106 registers (four waves per SIMD: the other waves fill the
// gaps), no branch -- 260 -> 242 us per 8-frame launch, BASELINE config 5 9480 -> 9920 frames/s (same box, alternated).
// Measured on the way and not kept: a scheduling barrier (__builtin_amdgcn_sched_barrier) between the groups instead of the
// dependences (changes nothing: 208 registers); the chains of a group advanced in lockstep through an asm statement per Horner step
// (the `s_nop`s go, 949 -> 241, but the half-used pairs stay packed and the registers go up: 254-273 us); 5 or 6 waves by
// __launch_bounds__ (spills).  Same operations on the same operands in the same order per tap, the sums taken tap by tap in the
// definition's order: the bits do not change (tests/test_prefilter.py, tests/test_replay.py).
template <int DY>
__device__ __forceinline__ void bilateral_row7(const float* __restrict__ row, float center, float ss, float sc, float& sum1, float& sum2) {
    constexpr int R = 7;
    constexpr int E = DY * DY == 0 ? 7 : (DY * DY <= 9 ? 6 : (DY * DY == 16 ? 5 : (DY * DY == 25 ? 4 : (DY * DY == 36 ? 3 : 0))));
    static_assert((E + 1) * (E + 1) + DY * DY > R * R && E * E + DY * DY <= R * R, "half-width of the circle in this row");
    constexpr int NT = 2 * E + 1;
    const v2f c2 = {center, center};
    float val[NT + 1];
    asm volatile("" ::: "memory");                  // (this row's LDS reads stay behind the previous row's)
#pragma unroll
    for (int i = 0; i < NT; i++) val[i] = row[i - E];
    val[NT] = val[NT - 1];                          // (the last tap goes through the pair's code too, its second half unused: exp_neg_spec's
                                                    // early exit is a branch, and fifteen branches cut the kernel into blocks across which
                                                    // the compiler kept 149 tile values alive)
    // Order, said with data dependences -- a scheduling barrier alone did not keep the compiler from computing every weight of the
    // kernel first --: the sums and the NEXT group's operands pass through one empty asm statement at the end of each group.
    constexpr int NP = (NT + 1) / 2;                // pairs of the row (the last one: one tap)
#ifndef SSF_BIL_GROUP
#define SSF_BIL_GROUP 2                             // pairs per group: independent chains the scheduler can interleave
#endif
#pragma unroll
    for (int g = 0; g < NP; g += SSF_BIL_GROUP) {
#pragma unroll
        for (int q = g; q < g + SSF_BIL_GROUP && q < NP; q++) asm volatile("" : "+v"(sum1), "+v"(sum2), "+v"(val[2 * q]), "+v"(val[2 * q + 1]));
        v2f w[SSF_BIL_GROUP], wv[SSF_BIL_GROUP];
#pragma unroll
        for (int q = g; q < g + SSF_BIL_GROUP && q < NP; q++) {
            const v2f v = {val[2 * q], val[2 * q + 1]};
            w[q - g] = bilateral_pair_w<DY>(v, c2, 2 * q - E, ss, sc);
            wv[q - g] = w[q - g] * v;
        }
#pragma unroll
        for (int q = g; q < g + SSF_BIL_GROUP && q < NP; q++) {            // tap by tap: the definition's order
            sum1 = sum1 + wv[q - g].x; sum2 = sum2 + w[q - g].x;
            if (2 * q + 1 < NT) { sum1 = sum1 + wv[q - g].y; sum2 = sum2 + w[q - g].y; }
        }
    }
}
template <int WAVES>
__global__ __launch_bounds__(256, WAVES) void k_bilateral_r7(BatchIn bin, float* __restrict__ out0, size_t slab, int W, int H, float ss, float sc) {
    const float* __restrict__ in = batch_pick(bin.depth, (int)blockIdx.z);
    float* __restrict__ out = slab_shift(out0, (size_t)blockIdx.z * slab);
    constexpr int R = 7, BW = BIL_TILE + 2 * R;
    __shared__ float tile[BW * BW];
    const int X0 = blockIdx.x * BIL_TILE, Y0 = blockIdx.y * BIL_TILE;
    for (int i = threadIdx.x; i < BW * BW; i += blockDim.x) {
        const int gx = reflect101(X0 - R + i % BW, W), gy = reflect101(Y0 - R + i / BW, H);
        tile[i] = in[(size_t)gy * W + gx];
    }
    __syncthreads();
    const int lx = threadIdx.x % BIL_TILE, ly = threadIdx.x / BIL_TILE;
    const int x = X0 + lx, y = Y0 + ly;
    if (x >= W || y >= H) return;
    const float* __restrict__ c = &tile[(ly + R) * BW + lx + R];
    const float center = c[0];
    float sum1 = 0.f, sum2 = 0.f;
    bilateral_row7<-7>(c - 7 * BW, center, ss, sc, sum1, sum2); bilateral_row7<-6>(c - 6 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<-5>(c - 5 * BW, center, ss, sc, sum1, sum2); bilateral_row7<-4>(c - 4 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<-3>(c - 3 * BW, center, ss, sc, sum1, sum2); bilateral_row7<-2>(c - 2 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<-1>(c - 1 * BW, center, ss, sc, sum1, sum2); bilateral_row7<0>(c, center, ss, sc, sum1, sum2);
    bilateral_row7<1>(c + 1 * BW, center, ss, sc, sum1, sum2); bilateral_row7<2>(c + 2 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<3>(c + 3 * BW, center, ss, sc, sum1, sum2); bilateral_row7<4>(c + 4 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<5>(c + 5 * BW, center, ss, sc, sum1, sum2); bilateral_row7<6>(c + 6 * BW, center, ss, sc, sum1, sum2);
    bilateral_row7<7>(c + 7 * BW, center, ss, sc, sum1, sum2);
    out[(size_t)y * W + x] = sum1 / sum2;
}
void launch_bilateral_batch(hipStream_t st, const BatchIn& in, float* out0, size_t slab, int nb, int W, int H, float sigma_color, float sigma_space) {
    ScopedKernel sk("bilateral_prefilter", st);
    int radius = (int)lrint((double)sigma_space * 1.5);
    if (radius < 1) radius = 1;
    const float ss = -0.5f / (sigma_space * sigma_space), sc = -0.5f / (sigma_color * sigma_color);
    static const bool generic_only = SSF_ENV_SET("BILATERAL_GENERIC");
    const dim3 grid((W + BIL_TILE - 1) / BIL_TILE, (H + BIL_TILE - 1) / BIL_TILE, nb);
    if (radius == 7 && !generic_only) {
#ifdef SSF_EXPERIMENTS
        static const int waves = SSF_ENV_INT("BIL_WAVES", 2);       // (3 and 4 waves per SIMD spill: measured slower)
        if (waves == 3) { hipLaunchKernelGGL(k_bilateral_r7<3>, grid, dim3(256), 0, st, in, out0, slab, W, H, ss, sc); return; }
        if (waves == 4) { hipLaunchKernelGGL(k_bilateral_r7<4>, grid, dim3(256), 0, st, in, out0, slab, W, H, ss, sc); return; }
#endif
        hipLaunchKernelGGL(k_bilateral_r7<2>, grid, dim3(256), 0, st, in, out0, slab, W, H, ss, sc);
        return;
    }
    hipLaunchKernelGGL(k_bilateral, grid, dim3(256), 0, st, in, out0, slab, W, H, radius, ss, sc);
}
void launch_bilateral(hipStream_t st, const float* in, float* out, int W, int H, float sigma_color, float sigma_space) {
    BatchIn b{}; b.depth[0] = in;
    launch_bilateral_batch(st, b, out, 0, 1, W, H, sigma_color, sigma_space);
}

// ---- launchers -----------------------------------------------------------------------------------

void launch_ingest(hipStream_t st, const SegParams& p, const BatchIn& in, FrameMaps& m, int nb, uint32_t epoch0) {
    ScopedKernel sk("ingest", st);
    hipLaunchKernelGGL(k_ingest, dim3(p.S, nb), dim3(256), 0, st, p, in, m, epoch0);
}
// pass pixels per thread (tile width / 32) for a launch over nb frames; SSF_PASS_NPX = 1 / 2 forces it (measurement)
int pass_tile_npx(int nb) {
    (void)nb;
    return 1;       // measured (profiles/bench_r02_npx*.json): the 64-wide tiles cut the pass's HBM traffic from 1.25x to 1.01x of
                    // the algorithmic bytes but not its time -- the kernel is bound by instruction issue, not by memory or by
                    // how many workgroups are resident -- and cost a single-frame launch 50 % more (8 -> 12 us)
}
// the largest cell window CellWindow::init gives a tile of the plain 32 x 32 grid (same formulas, host integers): SegParams::win_cells_max
int tile_window_cells_max(const SegParams& p) {
    auto cell_of = [&](int x) { return p.cell_magic ? (int)(((uint64_t)(uint32_t)x * p.cell_magic) >> 32) : x; };
    int best = 0;
    for (int Y0 = 0; Y0 < p.H; Y0 += TILE)
        for (int X0 = 0; X0 < p.W; X0 += TILE) {
            int margin = 2;
            const int tcx0 = cell_of(X0), tcy0 = cell_of(Y0), tcx1 = cell_of(std::min(X0 + TILE - 1, p.W - 1)), tcy1 = cell_of(std::min(Y0 + TILE - 1, p.H - 1));
            while (margin > 0 && (tcx1 - tcx0 + 1 + 2 * margin) * (tcy1 - tcy0 + 1 + 2 * margin) > WIN_MAX) margin--;
            const int n = (tcx1 - tcx0 + 1 + 2 * margin) * (tcy1 - tcy0 + 1 + 2 * margin);
            best = std::max(best, n <= WIN_MAX ? n : WIN_MAX + 1);          // (no window at all: the large instantiation, which then takes the exact path)
        }
    return best;
}
// the table behind SegParams::pass_geom: exactly what k_update_pass<., 1, .> works out for a tile (same formulas, host integers)
int pass_geometry_entries(int W, int H) { return 2 * ((W + (TILE - 2) + TILE - 1) / TILE) * ((H + TILE - 1) / TILE); }
void pass_geometry_table(const SegParams& p, uint2* out) {
    const int ntx = (p.W + (TILE - 2) + TILE - 1) / TILE, nty = (p.H + TILE - 1) / TILE;
    auto cell_of = [&](int x) { return p.cell_magic ? (int)(((uint64_t)(uint32_t)x * p.cell_magic) >> 32) : x; };
    for (int ox = 0; ox < 2; ox++)
        for (int by = 0; by < nty; by++)
            for (int bx = 0; bx < ntx; bx++) {
                const int X0 = bx * TILE - (ox ? 0 : TILE - 2), Y0 = by * TILE;
                const int TWP1 = TILE + 4;
                const bool interior = X0 >= 1 && X0 - 1 + TWP1 <= p.W && Y0 >= 1 && Y0 + TILE < p.H;
                int margin = 2;
                const int tcx0 = cell_of(std::max(X0, 0)), tcy0 = cell_of(Y0);
                const int tcx1 = cell_of(std::min(X0 + TILE - 1, p.W - 1)), tcy1 = cell_of(std::min(Y0 + TILE - 1, p.H - 1));
                while (margin > 0 && (tcx1 - tcx0 + 1 + 2 * margin) * (tcy1 - tcy0 + 1 + 2 * margin) > WIN_MAX) margin--;
                const int nwx = tcx1 - tcx0 + 1 + 2 * margin, nwy = tcy1 - tcy0 + 1 + 2 * margin;
                const bool ok = nwx * nwy <= WIN_MAX && nwx < 256 && nwy < 256;
                uint2 e;
                e.x = ((uint32_t)(tcx0 - margin) & 0xFFFFu) | ((uint32_t)(tcy0 - margin) << 16);
                e.y = (ok ? (uint32_t)nwx | ((uint32_t)nwy << 8) : 0u) | (interior ? 1u << 16 : 0u);
                out[(size_t)ox * ntx * nty + (size_t)by * ntx + bx] = e;
            }
}
void launch_update_pass(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int k, int ox, int oy, bool rgbd, int dbg) {
#ifdef SSF_EXPERIMENTS
    static const char* per_pass_names[64] = {nullptr};
    static int per_pass = -1;
    if (per_pass < 0) {
        per_pass = SSF_ENV_SET("PROFILE_PER_PASS") ? 1 : 0;
        static char buf[64][16];
        for (int i = 0; i < 64; i++) { snprintf(buf[i], 16, "pass_%02d", i); per_pass_names[i] = buf[i]; }
    }
    ScopedKernel sk(per_pass ? per_pass_names[k & 63] : (rgbd ? "update_pass_rgbd" : "update_pass_rgb"), st);
#else
    ScopedKernel sk(rgbd ? "update_pass_rgbd" : "update_pass_rgb", st);
#endif
    // OX = 0: tiles shifted left by (tile width - 2): [-30,1], [2,33], ...  The same (larger) grid is used for OX = 1 so
    // that tile ids -- and with them the per-tile log regions replayed by the next pass -- coincide.  All passes of a
    // frame use the same tile width (the log layout depends on it): 64 when the launch covers several frames.
    const int npx = pass_tile_npx(nb);
    const int twx = TILE * npx;
    dim3 grid = tile_grid(p);
    grid.x = (p.W + (twx - 2) + twx - 1) / twx;
    grid.z = nb;
    const TileOrder ord = tile_order(grid);
    PassArgs pa;
    {
        const int lp = (k + 2) % 3, lc = k % 3, odd = k & 1;
        pa.sr = m.sums[odd].r; pa.sw = m.sums[odd ^ 1].r;
        pa.pent = m.log.ent[lp]; pa.pdis = m.log.disp[lp]; pa.pcnt = m.log.count[lp];
        pa.cent = m.log.ent[lc]; pa.cdis = m.log.disp[lc]; pa.ccnt = m.log.count[lc];
        pa.geom = npx == 1 ? p.pass_geom + (size_t)(ox ? p.pass_ntile : 0) : nullptr;        // (pass_geometry_table: this grid, by construction)
    }
    // (Measured against this kernel and removed from the source, all bit-exact, all slower -- docs/HISTORY.md section 4.1.1 / 4.1.3:
    // 64-wide tiles with two pass pixels per thread, the RGB-D variant forced to 7 / 8 waves per SIMD, one wave per tile with the
    // changeable pixels compacted, four waves that release the ones the compacted list does not need, clean-tile skipping, all
    // passes of a phase in one resident launch with the label regions in LDS.  Round 6: a 64-register body (the replay ahead of the
    // decisions) and direct-to-LDS tile loads -- no gain, profiles/pass_ldsdma_r06.txt.)
    if (rgbd) hipLaunchKernelGGL((k_update_pass<true, 1, SSF_PASS_RGBD_WAVES>), grid, dim3(256), 0, st, p, m, pa, k, ox, oy, dbg, ord);
    else hipLaunchKernelGGL((k_update_pass<false, 1, 8>), grid, dim3(256), 0, st, p, m, pa, k, ox, oy, dbg, ord);
}
void launch_init_samples(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int true_buf) {
    ScopedKernel sk("init_samples", st);
    hipLaunchKernelGGL(k_init_samples, dim3((p.S * p.nb_samples + 255) / 256, nb), dim3(256), 0, st, p, m, true_buf);
}
static inline dim3 batch_tile_grid(const SegParams& p, int nb) { dim3 g = tile_grid(p); g.z = nb; return g; }
void launch_eval_samples(hipStream_t st, const SegParams& p, FrameMaps& m, int nb) {
    ScopedKernel sk("eval_samples", st);
    const dim3 grid = batch_tile_grid(p, nb);
    const bool small = p.win_cells_max <= WIN_SMALL;
    if (p.nb_samples == EVAL_NS) {
        if (small) hipLaunchKernelGGL((k_eval_samples<true, WIN_SMALL>), grid, dim3(256), 0, st, p, m, tile_order(grid));
        else hipLaunchKernelGGL((k_eval_samples<true, EVAL_WIN>), grid, dim3(256), 0, st, p, m, tile_order(grid));
    } else hipLaunchKernelGGL((k_eval_samples<false, EVAL_WIN>), grid, dim3(256), 0, st, p, m, tile_order(grid));
}
void launch_init_disp(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, bool ransac) {
    ScopedKernel sk("init_disp", st);
    const dim3 grid = batch_tile_grid(p, nb);
    if (p.win_cells_max <= WIN_SMALL) hipLaunchKernelGGL(k_init_disp<WIN_SMALL>, grid, dim3(256), SSF_INITDISP_DYN_LDS, st, p, m, ransac ? 1 : 0, tile_order(grid));
    else hipLaunchKernelGGL(k_init_disp<WIN_MAX>, grid, dim3(256), SSF_INITDISP_DYN_LDS, st, p, m, ransac ? 1 : 0, tile_order(grid));
}
void launch_plane_filter(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int true_buf) {
    ScopedKernel sk("plane_filter", st);
    // many workgroups per frame (core tiles + everything within filter_iter of them), when the largest window fits a workgroup
    {
        const int ntx = (p.gx + PF_TX - 1) / PF_TX, nty = (p.gy + PF_TY - 1) / PF_TY;
        int worst = 0;
        for (int ty = 0; ty < nty; ty++)
            for (int tx = 0; tx < ntx; tx++) { const PfWindow wd = pf_window(p.gx, p.gy, tx, ty, std::max(p.filter_iter, 0)); worst = std::max(worst, wd.n1 + wd.n2); }
        if (worst <= 1024 && ntx * nty > 1) {
            const int threads = ((worst + 63) / 64) * 64;
            hipLaunchKernelGGL(k_plane_filter_tiled, dim3(ntx, nty, nb), dim3(threads), (size_t)worst * 8 * sizeof(float), st, p, m, true_buf);
            return;
        }
    }
    const size_t lds = (size_t)p.S * 11 * sizeof(float);
    const int threads = p.S >= 1024 ? 1024 : ((p.S + 63) / 64) * 64;
    if (lds <= 60 * 1024) { hipLaunchKernelGGL(k_plane_filter<true>, dim3(nb), dim3(threads), lds, st, p, m, true_buf); return; }
    // larger grids: states and centroids in LDS (32 B / node, up to 160 KB per workgroup), the data term in registers
    const size_t lds2 = (size_t)p.S * 8 * sizeof(float);
    const int npt = (p.S + 1023) / 1024;
    // (these instantiations ask for more than 64 KB; the attribute belongs to the function ON A DEVICE, and a process may hold handles
    //  on several GPUs: asked once per device, not once per process)
    static std::mutex big_lds_mutex;
    static int big_lds_state[64] = {};                  // per device id: 0 = not asked, 1 = granted, 2 = refused
    int dev = 0; (void)hipGetDevice(&dev);
    bool big_lds = false;
    if (dev >= 0 && dev < 64) {
        std::lock_guard<std::mutex> lk(big_lds_mutex);
        if (big_lds_state[dev] == 0) {
            bool ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plane_filter_regs<5>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess;
            ok = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_plane_filter_regs<3>), hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) == hipSuccess && ok;
            (void)hipGetLastError();
            big_lds_state[dev] = ok ? 1 : 2;
        }
        big_lds = big_lds_state[dev] == 1;
    }
    if (big_lds && lds2 <= 160 * 1024 && npt <= 5) {
        if (npt <= 3) hipLaunchKernelGGL(k_plane_filter_regs<3>, dim3(nb), dim3(1024), lds2, st, p, m, true_buf);
        else hipLaunchKernelGGL(k_plane_filter_regs<5>, dim3(nb), dim3(1024), lds2, st, p, m, true_buf);
        return;
    }
    hipLaunchKernelGGL(k_plane_filter<false>, dim3(nb), dim3(1024), 0, st, p, m, true_buf);
}
void launch_render_moments(hipStream_t st, const SegParams& p, const Cam& cam, FrameMaps& m, int nb) {
    ScopedKernel sk("render_moments", st);
    const dim3 grid = batch_tile_grid(p, nb);
    if (p.win_cells_max <= WIN_SMALL) hipLaunchKernelGGL(k_render_moments<WIN_SMALL>, grid, dim3(256), SSF_RENDER_DYN_LDS, st, p, cam, m, tile_order(grid));
    else hipLaunchKernelGGL(k_render_moments<WIN_MAX>, grid, dim3(256), SSF_RENDER_DYN_LDS, st, p, cam, m, tile_order(grid));
}
void launch_import_frame(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, SurfelSoA frame, const float* wire, unsigned long long* best, uint8_t* matched) {
    ScopedKernel sk("import_frame", st);
    const int P = p.W * p.H, n = std::max(P, p.S);
    hipLaunchKernelGGL(k_import_frame, dim3((n + 255) / 256, nb), dim3(256), 0, st, P, p.S, m, frame, wire, best, matched);
}
void launch_export_rows(hipStream_t st, const SegParams& p, const FrameMaps& m, int nb, SurfelSoA frame, float* wire) {
    ScopedKernel sk("export_rows", st);
    hipLaunchKernelGGL(k_export_rows, dim3((p.S + 255) / 256, nb), dim3(256), 0, st, p.S, frame, wire, m.slab);
}
void launch_finalize_surfels(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, SurfelSoA frame, float zmin,
                             float zmax, int stamp0, const uint8_t* dynamic_mask, unsigned mask_bits,
                             unsigned long long* best, uint8_t* matched) {
    ScopedKernel sk("finalize_surfels", st);
    hipLaunchKernelGGL(k_finalize_surfels, dim3((p.S + FIN_PER_WG - 1) / FIN_PER_WG, nb), dim3(256), 0, st, p, m, frame, zmin, zmax, stamp0,
                       dynamic_mask, mask_bits, best, matched);
}
void launch_boundary_map(hipStream_t st, const SegParams& p, const int32_t* label, int32_t* out) {
    hipLaunchKernelGGL(k_boundary_map, tile_grid(p), dim3(256), 0, st, p, label, out);
}

}  // namespace ssf
