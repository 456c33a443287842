// ssf_pass_tile.hpp -- ONE TILE OF ONE RELABELLING PASS: the body of k_update_pass (ssf_extract.hip), as text.
// Included inside a function whose scope provides: constexpr bool RGBD, COH; constexpr int NPX; SegParams p; FrameMaps m (a copy: it
// is shifted to the frame's slot here); PassArgs pa; int pass, OX, OY, dbg; TileOrder ord; unsigned int bx, by, bz (the
// tile's column, row and frame slot).  The product includes it once, in the kernel (COH = false); the lab build a second time, in
// update_pass_tile -- the body its resident arm walks tile by tile (lab/passes_team.inc; COH = true: what another workgroup wrote in
// an earlier pass of the same launch is read past the L1, ld_coh).  As a function called from the kernel the same text ran 1-1.5 %
// slower in place (gpurun r13b, alternated three times), hence the include.
    constexpr int TWX = TILE * NPX, LOGN = 256 * NPX;
    SSF_PASS_TICK_BEGIN();                        // (lab: five ticks of the wall clock per workgroup -- tools/pass_trace.py)
    __shared__ __attribute__((aligned(16))) int tile[(TWX + 4) * TW];      // rows of TWX + 4 labels: see the tile loads below
    __shared__ SpRow w_row[WIN_MAX];
    __shared__ int w_label[WIN_MAX];                          // label of a window slot (-1: outside the grid)
    __shared__ __attribute__((aligned(16))) unsigned int w_acc[WIN_MAX * PASS_ACC_DW];      // this tile's sum deltas (own + replayed), flushed once
    __shared__ unsigned int s_nlog;
    const size_t slot_off = (size_t)bz * m.slab;                 // this frame's slot of the batch context
    m.rgba = slab_shift(m.rgba, slot_off); m.disp = slab_shift(m.disp, slot_off); m.label = slab_shift(m.label, slot_off);
    m.inlier = slab_shift(m.inlier, slot_off); m.epoch = slab_shift(m.epoch, slot_off);
    SpSums sr, sw;
    sr.r = const_cast<SumRec*>(slab_shift(pa.sr, slot_off)); sw.r = slab_shift(pa.sw, slot_off);
    const int X0 = __builtin_amdgcn_readfirstlane((int)bx * TWX - (OX ? 0 : TWX - 2)), Y0 = __builtin_amdgcn_readfirstlane((int)by * TILE);  // OX = 0: tiles start at 2 (mod 4)
    int32_t* __restrict__ lab = m.label;
    // this thread's pass pixels: local columns 4j+1, 4j+2 of pass rows; pixel s of the thread is element threadIdx.x + 256 s
    int x[NPX], y[NPX], lxh[NPX], lyh[NPX]; bool in_image[NPX]; unsigned int q[NPX];
#pragma unroll
    for (int s = 0; s < NPX; s++) {
        const int e = threadIdx.x + 256 * s;
        const int tx = e % (16 * NPX), ty = e / (16 * NPX);
        const int lx0 = 4 * (tx >> 1) + 1 + (tx & 1), ly0 = 2 * ty + OY;
        x[s] = X0 + lx0; y[s] = Y0 + ly0; lxh[s] = lx0 + 1; lyh[s] = ly0 + 1;      // lxh / lyh: halo coordinates
        in_image[s] = x[s] >= 0 && x[s] < p.W && y[s] < p.H;
        q[s] = in_image[s] ? __umul24((unsigned int)y[s], (unsigned int)p.W) + (unsigned int)x[s] : 0u;      // (W, H < 2^16)
    }
    const int tile_id = (int)(by * ord.ntx + bx);
    const unsigned int* __restrict__ pcnt = slab_shift(pa.pcnt, slot_off);
    // The previous pass' entry count of this tile (uniform, needed only further down).  Behind `pass > 0` the compiler waits
    // for the word inside the branch -- a dependent trip to memory before the first vector load is issued.  Form 1 requests it
    // unconditionally (at pass 0 the word is a stale count of an earlier frame and is ignored), form 2 as a VECTOR load of
    // one address (an opaque zero in a vector register makes the address look per-lane), which travels with the pixel
    // operands and costs no scalar register.  The RGB variant lives at 8 waves per SIMD with no register to spare: which form
    // pays is measured per variant (SSF_PASS_NPREV_RGBD / SSF_PASS_NPREV_RGB).
    constexpr int NPREV_FORM = RGBD ? SSF_PASS_NPREV_RGBD : SSF_PASS_NPREV_RGB;
    unsigned int n_prev_word = 0u;
    if (COH) n_prev_word = ld_coh<true>(&pcnt[tile_id]);      // (never through the scalar cache: the count was stored by a vector store of this launch)
    else if (NPREV_FORM == 0) { if (pass > 0) n_prev_word = pcnt[tile_id]; }
    else if (NPREV_FORM == 1) n_prev_word = pcnt[tile_id];
    else { int lane_zero = 0; asm volatile("" : "+v"(lane_zero)); n_prev_word = pcnt[tile_id + lane_zero]; }
    // operands that do not depend on the label tile: in flight while the tile is staged
    uint32_t px[NPX]; float disp[NPX]; unsigned char prev_inlier[NPX];
#pragma unroll
    for (int s = 0; s < NPX; s++) {
        px[s] = ld_off<uint32_t>(m.rgba, 4u * q[s]);
        disp[s] = 0.f; prev_inlier[s] = 0;
        if (RGBD) { disp[s] = ld_off<float>(m.disp, 4u * q[s]); prev_inlier[s] = ld_off_c<COH, unsigned char>(m.inlier, q[s]); }
    }
    // the label tile + halo: requested into registers NOW (independent loads), stored to LDS after the superpixel rows
    // have been computed -- one memory round trip for tile, pixel operands, sums and log.
    // The tile travels in QUADS: a row of the LDS tile is TWP = TWX + 4 labels (the 2 + TWX halo columns and two columns of
    // padding that nothing reads), i.e. TWP / 4 16-byte quads, and quad e of the tile lies at LDS dword 4 e.  When the
    // halo (and the padding) lies inside the image -- a uniform test -- a lane fetches a quad with ONE 16-byte load (the
    // address is only 4-byte aligned: X0 - 1 is odd; global loads take that) and stores it with one ds_write_b128: two
    // rounds per thread instead of five rounds of single labels, a third of the address arithmetic.  Edge tiles fetch the
    // four labels of a quad one by one from offsets clamped into the image; `outside` = the labels that are not in it.
    // Either way the loads are unconditional (loads behind a branch made the compiler wait for the pixel operands before it
    // issued them: a second dependent trip to memory); the idle lanes of the last round re-read the tile's last quad.
    constexpr int TWP = TWX + 4, QPR = TWP / 4, NQ = QPR * TW, TILE_LOADS = (NQ + 255) / 256;
    constexpr unsigned int QPR_MAGIC = 65536u / QPR + 1u;          // e / QPR == (e * QPR_MAGIC) >> 16 for e < 1024
    static_assert(NQ <= 1024 && TILE_LOADS * 4 <= 32, "quad index / outside mask");
    uint4 tile_reg[TILE_LOADS]; unsigned int outside = 0u;
    const bool no_tile = SSF_PROBE(dbg, 32);                     // (probe: every element reads as outside the image)
    // (window geometry and the interior test: one 8-byte table entry per tile for the product's 32-wide tiles, fetched by a scalar
    // load that travels with the kernel arguments; worked out here for the lab's 64-wide tiles)
    uint2 geom = make_uint2(0u, 0u);
    constexpr bool have_geom = NPX == 1;                      // (launch_update_pass always supplies the table for 32-wide tiles)
    if (have_geom) geom = pa.geom[by * ord.ntx + bx];
    const bool interior = have_geom ? ((geom.y >> 16) & 1u) != 0u : (X0 >= 1 && X0 - 1 + TWP <= p.W && Y0 >= 1 && Y0 + TILE < p.H);
    if (interior) {
        typedef uint32_t Quad __attribute__((ext_vector_type(4), aligned(4)));     // ONE load of four labels, 4-byte aligned
        const unsigned int base_off = (unsigned int)((Y0 - 1) * p.W + (X0 - 1));
        unsigned int tile_off[TILE_LOADS];
#pragma unroll
        for (int k = 0; k < TILE_LOADS; k++) {
            const unsigned int e = 256 * (k + 1) <= NQ ? threadIdx.x + 256u * k : min(threadIdx.x + 256u * k, (unsigned int)(NQ - 1));
            const unsigned int ly = __umul24(e, QPR_MAGIC) >> 16;      // quad (ly, e - ly * QPR): label offset ly * W + 4 (e - ly * QPR)
            tile_off[k] = 4u * (base_off + 4u * e + __umul24(ly, (unsigned int)(p.W - TWP)));   // bytes (24-bit multiplies are full rate)
        }
#pragma unroll
        for (int k = 0; k < TILE_LOADS; k++) {
            const Quad v = ld_off_c<COH, Quad>(lab, tile_off[k]);
            tile_reg[k] = make_uint4(v.x, v.y, v.z, v.w);
        }
    } else {
        unsigned int tile_off[TILE_LOADS][4];
#pragma unroll
        for (int k = 0; k < TILE_LOADS; k++) {
            const unsigned int e = 256 * (k + 1) <= NQ ? threadIdx.x + 256u * k : min(threadIdx.x + 256u * k, (unsigned int)(NQ - 1));
            const int ly = (int)(__umul24(e, QPR_MAGIC) >> 16), lx = 4 * ((int)e - ly * QPR);
            const int gy_ = Y0 - 1 + ly, cy_ = min(max(gy_, 0), p.H - 1);
            const unsigned int row_off = __umul24((unsigned int)cy_, (unsigned int)p.W);
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int gx_ = X0 - 1 + lx + j, cx_ = min(max(gx_, 0), p.W - 1);
                if (cx_ != gx_ || cy_ != gy_) outside |= 1u << (4 * k + j);
                tile_off[k][j] = 4u * (row_off + (unsigned int)cx_);
            }
        }
#pragma unroll
        for (int k = 0; k < TILE_LOADS; k++)
            tile_reg[k] = make_uint4(ld_off_c<COH, uint32_t>(lab, tile_off[k][0]), ld_off_c<COH, uint32_t>(lab, tile_off[k][1]), ld_off_c<COH, uint32_t>(lab, tile_off[k][2]), ld_off_c<COH, uint32_t>(lab, tile_off[k][3]));
    }
    if (no_tile) outside = 0xFFFFFFFFu;
    // window of grid cells around the tile whose superpixel rows are cached in LDS
    int wcx0, wcy0, nwx, nwy;
    bool window_ok;
    if (have_geom) {
        wcx0 = __builtin_amdgcn_readfirstlane((int)(short)(geom.x & 0xFFFFu)); wcy0 = __builtin_amdgcn_readfirstlane((int)(short)(geom.x >> 16));
        nwx = __builtin_amdgcn_readfirstlane((int)(geom.y & 255u)); nwy = __builtin_amdgcn_readfirstlane((int)((geom.y >> 8) & 255u));
        window_ok = nwx != 0;
    } else {
        int margin = 2;
        const int tcx0 = div_cell(p, max(X0, 0)), tcy0 = div_cell(p, Y0);
        const int tcx1 = div_cell(p, min(X0 + TWX - 1, p.W - 1)), tcy1 = div_cell(p, min(Y0 + TILE - 1, p.H - 1));
        while (margin > 0 && (tcx1 - tcx0 + 1 + 2 * margin) * (tcy1 - tcy0 + 1 + 2 * margin) > WIN_MAX) margin--;
        // (uniform values, said so: without the readfirstlane the RGB-D variant computed the whole window geometry -- two
        // mul_hi, four quarter-rate mul_lo, the margin loop -- in vector registers, ~45 issue slots per wave)
        margin = __builtin_amdgcn_readfirstlane(margin);
        wcx0 = __builtin_amdgcn_readfirstlane(tcx0 - margin); wcy0 = __builtin_amdgcn_readfirstlane(tcy0 - margin);
        const int nwx_ = tcx1 - tcx0 + 1 + 2 * margin, nwy_ = tcy1 - tcy0 + 1 + 2 * margin;
        window_ok = nwx_ * nwy_ <= WIN_MAX;
        nwx = __builtin_amdgcn_readfirstlane(window_ok ? nwx_ : 0); nwy = __builtin_amdgcn_readfirstlane(window_ok ? nwy_ : 0);   // no window: every label takes the exact path
    }
    const int nslots = nwx * nwy;
    const SpRow zero_row = {0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0};
    // (window <= 64 cells: wave 0 builds the means of cell `lane`, wave 1 -- RGB-D passes -- its plane, side by side)
    if (threadIdx.x < (RGBD ? 128 : 64)) {
        const int i = threadIdx.x & 63;
        if (i < nslots) {
            // (i / nwx for i < 64, nwx <= 8: (i + 0.5) / nwx is at least 1/16 away from an integer -- an approximate reciprocal will do)
            const int wy = (int)(((float)i + 0.5f) * __builtin_amdgcn_rcpf((float)nwx)), wx = i - __mul24(wy, nwx);
            const int cx = wcx0 + wx, cy = wcy0 + wy;
            const bool inside = cx >= 0 && cx < p.gx && cy >= 0 && cy < p.gy;
            const int k = cy * p.gx + cx;
            if (threadIdx.x < 64) w_label[i] = inside ? k : -1;
            if (inside && !SSF_PROBE(dbg, 1)) {
                if (threadIdx.x < 64) {
                    SpRow row = zero_row;
                    row_means_from_sums<COH>(sr, k, row);
                    w_row[i].cx = row.cx; w_row[i].cy = row.cy; w_row[i].r = row.r; w_row[i].g = row.g; w_row[i].b = row.b; w_row[i].size = row.size;
                    w_row[i].pad0 = row.size / (row.size - 1.f);      // the pixel's own-energy scale n / (n - 1): one division per window cell instead of one per pass pixel
                    if (!RGBD) { w_row[i].ta = 0.f; w_row[i].tb = 0.f; w_row[i].tc = 0.f; }
                } else {
                    float ta, tb, tc;
                    row_plane_from_sums<COH>(sr, k, ta, tb, tc);
                    w_row[i].ta = ta; w_row[i].tb = tb; w_row[i].tc = tc;
                }
            }
        }
    }
    // this tile's log of the previous pass is replayed at the very end.  Only the valid entries are fetched (an unconditional
    // fetch of the whole region cost 5 B per pixel of HBM): the other lanes read entry 0 again -- one address, and no branch
    // around the load (behind a branch the compiler waits for the entry at once)
    const int4* __restrict__ pent = slab_shift(pa.pent, slot_off);
    const float* __restrict__ pdis = slab_shift(pa.pdis, slot_off);
    int4 prev_ent[NPX]; float prev_disp[NPX];
    const unsigned int n_prev = pass > 0 ? n_prev_word : 0u;
#pragma unroll
    for (int s = 0; s < NPX; s++) {
        const unsigned int e = threadIdx.x + 256u * s;
        const unsigned int le = (unsigned int)tile_id * LOGN + (e < n_prev ? e : 0u);
        prev_ent[s] = ld_off_c<COH, int4>(pent, 16u * le);
        prev_disp[s] = 0.f;
        if (RGBD) prev_disp[s] = ld_off_c<COH, float>(pdis, 4u * le);
    }
    if (threadIdx.x == 0) s_nlog = 0;
    constexpr int ACC_CHUNKS = RGBD ? 6 : 2;                  // 16-byte chunks of a slot that a pass of this kind can touch (RGB: sx .. n)
    // (RGB-D: every chunk of a slot, i.e. the first 6 nslots chunks of the array; RGB: chunks 0 and 1 of each slot)
    for (int i = threadIdx.x; i < nslots * ACC_CHUNKS; i += 256)
        reinterpret_cast<uint4*>(w_acc)[RGBD ? i : __mul24(i >> 1, 6) + (i & 1)] = make_uint4(0u, 0u, 0u, 0u);
    // (the idle lanes of the last round store the last quad once more: same value, same place -- no branch)
    if (!interior || no_tile) {
#pragma unroll
        for (int k = 0; k < TILE_LOADS; k++) {
            if ((outside >> (4 * k)) & 1u) tile_reg[k].x = 0xFFFFFFFFu;
            if ((outside >> (4 * k + 1)) & 1u) tile_reg[k].y = 0xFFFFFFFFu;
            if ((outside >> (4 * k + 2)) & 1u) tile_reg[k].z = 0xFFFFFFFFu;
            if ((outside >> (4 * k + 3)) & 1u) tile_reg[k].w = 0xFFFFFFFFu;
        }
    }
    SSF_PASS_TICK_LOADS();                        // (everything requested up front has arrived)
#pragma unroll
    for (int k = 0; k < TILE_LOADS; k++) {
        const unsigned int e = 256 * (k + 1) <= NQ ? threadIdx.x + 256u * k : min(threadIdx.x + 256u * k, (unsigned int)(NQ - 1));
        reinterpret_cast<uint4*>(tile)[e] = tile_reg[k];
    }
    __syncthreads();
    SSF_PASS_TICK(2);                             // (tile, window rows and accumulators staged)
    const float inv_gx = p.inv_gx;
    auto slot_of = [&](int l) -> int {
        const int cyl = (int)(((float)l + 0.5f) * inv_gx);         // l / gx, exact for l < 2^20
        const unsigned int wx = (unsigned int)((l - __mul24(cyl, p.gx)) - wcx0), wy = (unsigned int)(cyl - wcy0);
        return (wx < (unsigned int)nwx && wy < (unsigned int)nwy) ? (int)(__umul24(wy, (unsigned int)nwx) + wx) : -1;
    };
    auto row_of = [&](int l) -> SpRow {
        const int ws = slot_of(l);
        SSF_PASS_STAT_LOOKUP(ws < 0);
        if (ws >= 0) return *reinterpret_cast<const SpRow*>(reinterpret_cast<const char*>(w_row) + __umul24((unsigned int)ws, (unsigned int)sizeof(SpRow)));
        SpRow far = row_from_sums<COH>(sr, l, RGBD, zero_row);     // drifted out of the window: exact slow path
        far.pad0 = far.size / (far.size - 1.f);
        return far;
    };
    // sum deltas of one relabelled pixel: LDS accumulators of the window, global atomics outside it
    auto add_delta = [&](int from, int to, int px_x, int px_y, uint32_t rgbf, float d) {
        const unsigned fl = rgbf >> 24;
        const int wf = slot_of(from), wt = slot_of(to);
        if (fl & 1u) {
            const int ir = (int)(rgbf & 255u), ig = (int)((rgbf >> 8) & 255u), ib = (int)((rgbf >> 16) & 255u);
            if (wf >= 0) {
                unsigned int* a = &w_acc[__umul24((unsigned int)wf, PASS_ACC_DW)];
                atomicAdd(&a[F_SX], (unsigned int)-px_x); atomicAdd(&a[F_SY], (unsigned int)-px_y); atomicAdd(&a[F_SR], (unsigned int)-ir);
                atomicAdd(&a[F_SG], (unsigned int)-ig); atomicAdd(&a[F_SB], (unsigned int)-ib); atomicAdd(&a[F_N], 0xFFFFFFFFu);
            } else { atomicAdd(&sw.r[from].sx, -px_x); atomicAdd(&sw.r[from].sy, -px_y); atomicAdd(&sw.r[from].sr, -ir);
                     atomicAdd(&sw.r[from].sg, -ig); atomicAdd(&sw.r[from].sb, -ib); atomicAdd(&sw.r[from].n, -1); }
            if (wt >= 0) {
                unsigned int* a = &w_acc[__umul24((unsigned int)wt, PASS_ACC_DW)];
                atomicAdd(&a[F_SX], (unsigned int)px_x); atomicAdd(&a[F_SY], (unsigned int)px_y); atomicAdd(&a[F_SR], (unsigned int)ir);
                atomicAdd(&a[F_SG], (unsigned int)ig); atomicAdd(&a[F_SB], (unsigned int)ib); atomicAdd(&a[F_N], 1u);
            } else { atomicAdd(&sw.r[to].sx, px_x); atomicAdd(&sw.r[to].sy, px_y); atomicAdd(&sw.r[to].sr, ir);
                     atomicAdd(&sw.r[to].sg, ig); atomicAdd(&sw.r[to].sb, ib); atomicAdd(&sw.r[to].n, 1); }
        }
        if (RGBD && (fl & 6u)) {
            // the nine disparity terms of the pixel, converted once: added to `to` (flag 2), taken from `from` (flag 4)
            // (coordinates < 2^16: the products fit 32 bits unsigned, and a 24-bit multiply returns the low 32 bits of a 48-bit product)
            const long long xx = (long long)__umul24((unsigned int)px_x, (unsigned int)px_x), yy = (long long)__umul24((unsigned int)px_y, (unsigned int)px_y),
                            xy = (long long)__umul24((unsigned int)px_x, (unsigned int)px_y);
            const long long xd = fx64((double)((float)px_x * d), SSF_DISP_SCALE, SSF_DISP_LIM);
            const long long yd = fx64((double)((float)px_y * d), SSF_DISP_SCALE, SSF_DISP_LIM);
            const long long dd = fx64((double)d, SSF_DISP_SCALE, SSF_DISP_LIM);
            if (fl & 2u) {
                if (wt >= 0) {
                    unsigned int* a = &w_acc[__umul24((unsigned int)wt, PASS_ACC_DW)]; unsigned long long* b = reinterpret_cast<unsigned long long*>(a + PASS_ACC_WIDE_DW);
                    atomicAdd(&a[F_DX], (unsigned int)px_x); atomicAdd(&a[F_DY], (unsigned int)px_y); atomicAdd(&a[F_DN], 1u);
                    lds_add_i64(&b[F_DXX - PASS_F32], xx); lds_add_i64(&b[F_DYY - PASS_F32], yy); lds_add_i64(&b[F_DXY - PASS_F32], xy);
                    lds_add_i64(&b[F_DXD - PASS_F32], xd); lds_add_i64(&b[F_DYD - PASS_F32], yd); lds_add_i64(&b[F_DD - PASS_F32], dd);
                } else { disp_sums_add(sw, to, px_x, px_y, d, +1); }
            }
            if (fl & 4u) {
                if (wf >= 0) {
                    unsigned int* a = &w_acc[__umul24((unsigned int)wf, PASS_ACC_DW)]; unsigned long long* b = reinterpret_cast<unsigned long long*>(a + PASS_ACC_WIDE_DW);
                    atomicAdd(&a[F_DX], (unsigned int)-px_x); atomicAdd(&a[F_DY], (unsigned int)-px_y); atomicAdd(&a[F_DN], 0xFFFFFFFFu);
                    lds_add_i64(&b[F_DXX - PASS_F32], -xx); lds_add_i64(&b[F_DYY - PASS_F32], -yy); lds_add_i64(&b[F_DXY - PASS_F32], -xy);
                    lds_add_i64(&b[F_DXD - PASS_F32], -xd); lds_add_i64(&b[F_DYD - PASS_F32], -yd); lds_add_i64(&b[F_DD - PASS_F32], -dd);
                } else { disp_sums_add(sw, from, px_x, px_y, d, -1); }
            }
        }
    };
    if (SSF_PROBE(dbg, 2)) return;
    int4* __restrict__ cent = slab_shift(pa.cent, slot_off);
    float* __restrict__ cdis = slab_shift(pa.cdis, slot_off);
#pragma unroll
    for (int s = 0; s < NPX; s++) {
        const int lx = lxh[s], ly = lyh[s];
        const int* __restrict__ t = &tile[__mul24(ly, TWP) + lx];
        const int index = in_image[s] ? t[0] : 0;
        int new_index = index;
        const int nl[4] = {t[-TWP], t[-1], t[1], t[TWP]};                           // N, W, E, S
        const int bounds = (nl[0] != index) + (nl[1] != index) + (nl[2] != index) + (nl[3] != index);
        bool eligible = in_image[s] && bounds != 0 && !SSF_PROBE(dbg, 4);
        if (eligible) {
            // connectivity guard isUnchangeable (TPS_RGBD_kernels.cuh:178-233): ring NW,N,NE,E,SE,S,SW,W; the label changes
            // more than twice along the ring = the pixel is a bridge.  Bit k of `ring`: ring pixel k carries the pixel's label
            eligible = !guard_unchangeable(guard_ring(index, t[-TWP - 1], nl[0], t[-TWP + 1], nl[2], t[TWP + 1], nl[3], t[TWP - 1], nl[1]));
        }
        SpRow own = zero_row;
        if (in_image[s] && (RGBD || eligible)) own = row_of(index);
        float disp_energy = 0.f;
        unsigned char inlier = 0xff;
        if (RGBD && in_image[s]) {
            const float dp = (own.ta * (float)x[s] + own.tb * (float)y[s]) + own.tc;
            disp_energy = (dp - disp[s]) * (dp - disp[s]);
            if (!isfinite(disp_energy) || disp_energy > p.thresh_disp || dp < 0.f) { disp_energy = p.thresh_disp; inlier = 0; }
        }
        const float cr = (float)(px[s] & 255u), cg = (float)((px[s] >> 8) & 255u), cb = (float)((px[s] >> 16) & 255u);
        const float posx = (float)x[s], posy = (float)y[s];
        float best = 0.f;
        if (eligible) {
            const float size = own.size;
            const float sc = own.pad0;                        // size / (size - 1), divided once per superpixel (w_row / row_of)
            const float dpx = sc * (posx - own.cx), dpy = sc * (posy - own.cy);
            const V3 dcol = v3(sc * (cr - own.r), sc * (cg - own.g), sc * (cb - own.b));
            const float dsize = size - (float)p.min_size;
            best = dot3(dcol, dcol) + p.lambda_pos * (dpx * dpx + dpy * dpy);
            if (RGBD) best = best + p.lambda_disp * disp_energy;
            best = best - p.lambda_size * fminf(dsize, 0.f);
            best = best + p.lambda_bound * (float)bounds;
        }
        // the DISTINCT neighbour labels other than the pixel's own, in the order N, W, E, S (a label met a second time has the
        // same energy and cannot win against itself under the strict comparison: skipping it changes nothing).  Compacted per
        // lane, so that the wave walks max-over-lanes(count) candidates -- one or two on a boundary -- instead of four directions
        {
            const bool ok0 = eligible && nl[0] != -1 && nl[0] != index;
            const bool ok1 = eligible && nl[1] != -1 && nl[1] != index && nl[1] != nl[0];
            const bool ok2 = eligible && nl[2] != -1 && nl[2] != index && nl[2] != nl[0] && nl[2] != nl[1];
            const bool ok3 = eligible && nl[3] != -1 && nl[3] != index && nl[3] != nl[0] && nl[3] != nl[1] && nl[3] != nl[2];
            const int p1 = ok0 ? 1 : 0, p2 = p1 + (ok1 ? 1 : 0), p3 = p2 + (ok2 ? 1 : 0), ncand = p3 + (ok3 ? 1 : 0);
            const int c0 = ok0 ? nl[0] : (ok1 ? nl[1] : (ok2 ? nl[2] : nl[3]));
            const int c1 = (ok1 && p1 == 1) ? nl[1] : ((ok2 && p2 == 1) ? nl[2] : nl[3]);
            const int c2 = (ok2 && p2 == 2) ? nl[2] : nl[3];
            const int c3 = nl[3];
#pragma unroll
            for (int j = 0; j < 4; j++) {
                if (__ballot(j < ncand) == 0ull) break;               // (wave-uniform)
                if (j < ncand) {
                    const int i_n = j == 0 ? c0 : (j == 1 ? c1 : (j == 2 ? c2 : c3));
                    const SpRow nb = row_of(i_n);
                    const float ndx = posx - nb.cx, ndy = posy - nb.cy;
                    const V3 ndc = v3(cr - nb.r, cg - nb.g, cb - nb.b);
                    const float ndsize = (nb.size + 1.f) - (float)p.min_size;
                    float n_de = 0.f; unsigned char n_inlier = 0xff;
                    if (RGBD) {
                        const float dp = (nb.ta * (float)x[s] + nb.tb * (float)y[s]) + nb.tc;
                        n_de = (dp - disp[s]) * (dp - disp[s]);
                        if (!isfinite(n_de) || n_de > p.thresh_disp || dp < 0.f) { n_de = p.thresh_disp; n_inlier = 0; }
                    }
                    const int b = (nl[0] != i_n) + (nl[1] != i_n) + (nl[2] != i_n) + (nl[3] != i_n);
                    float e = dot3(ndc, ndc) + p.lambda_pos * (ndx * ndx + ndy * ndy);
                    if (RGBD) e = e + p.lambda_disp * n_de;
                    e = e - p.lambda_size * fminf(ndsize, 0.f);
                    e = e + p.lambda_bound * (float)b;
                    if (e < best) { best = e; new_index = i_n; if (RGBD) inlier = n_inlier; }
                }
            }
        }
        unsigned flags = 0u;
        if (in_image[s]) {
            if (new_index != index) st_off<int32_t>(lab, 4u * q[s], new_index);
            flags = (new_index != index) ? 1u : 0u;
            if (RGBD) {
                if (inlier && (!prev_inlier[s] || index != new_index)) flags |= 2u;
                if (prev_inlier[s] && (!inlier || (inlier && index != new_index))) flags |= 4u;
                if (inlier != prev_inlier[s]) st_off<unsigned char>(m.inlier, q[s], inlier);
            }
        }
        if (flags) {
            const uint32_t rgbf = (px[s] & 0x00FFFFFFu) | (flags << 24);
            add_delta(index, new_index, x[s], y[s], rgbf, disp[s]);
            const unsigned int slot = atomicAdd(&s_nlog, 1u);                 // LDS counter, < LOGN by construction
            const unsigned int ce = (unsigned int)tile_id * LOGN + slot;
            st_off<int4>(cent, 16u * ce, make_int4(index, new_index, x[s] | (y[s] << 16), (int)rgbf));
            if (RGBD) st_off<float>(cdis, 4u * ce, disp[s]);
        }
    }
    // replay this tile's log of the previous pass into the buffer this pass writes (it lags by exactly that)
    if (!SSF_PROBE(dbg, 8)) {
#pragma unroll
        for (int s = 0; s < NPX; s++)
            if (threadIdx.x + 256u * s < n_prev)
                add_delta(prev_ent[s].x, prev_ent[s].y, prev_ent[s].z & 0xFFFF, (prev_ent[s].z >> 16) & 0xFFFF, (uint32_t)prev_ent[s].w, prev_disp[s]);
    }
    __syncthreads();
    SSF_PASS_TICK(3);                             // (decisions taken, deltas in LDS, log written)
    // flush: the accumulators are scanned a 16-byte chunk at a time (most are zero); one global atomic per sum that is not.
    // The record's nine int32 sums and six int64 sums are addressed by field number (SumRec: int32 fields from byte 0, int64
    // fields from byte 64)
    static_assert(WIN_MAX * 6 <= 32768, "i / 6 by multiplication");
    // A tile that logged nothing and replayed nothing has nothing to flush (most tiles of the later passes): no scan.
    if (s_nlog == 0u && n_prev == 0u) {
        if (threadIdx.x == 0) {
            unsigned int* __restrict__ ccnt = slab_shift(pa.ccnt, slot_off);
            ccnt[tile_id] = 0u;
        }
        SSF_PASS_STAT_END(0u);
        SSF_PASS_TICK(4);
        return;
    }
    for (int i = threadIdx.x; i < nslots * ACC_CHUNKS; i += 256) {
        const uint4 v = reinterpret_cast<const uint4*>(w_acc)[RGBD ? i : __mul24(i >> 1, 6) + (i & 1)];
        if ((v.x | v.y | v.z | v.w) == 0u) continue;
        const int wi = RGBD ? (int)(__umul24((unsigned int)i, 10923u) >> 16) : (i >> 1);       // i / 6, i / 2
        const int c = i - __mul24(wi, ACC_CHUNKS);
        SumRec* rec = &sw.r[w_label[wi]];
        if (c < 3) {
            int* f = &rec->sx + 4 * c;                             // (chunk 2: dn and three dwords of padding, always zero)
            if (v.x) atomicAdd(f, (int)v.x);
            if (v.y) atomicAdd(f + 1, (int)v.y);
            if (v.z) atomicAdd(f + 2, (int)v.z);
            if (v.w) atomicAdd(f + 3, (int)v.w);
        } else {
            long long* f = &rec->dxx + 2 * (c - 3);
            const long long lo = (long long)(((unsigned long long)v.y << 32) | v.x), hi = (long long)(((unsigned long long)v.w << 32) | v.z);
            if (lo) atomic_add_i64(f, lo);
            if (hi) atomic_add_i64(f + 1, hi);
        }
    }
    if (threadIdx.x == 0) {
        unsigned int* __restrict__ ccnt = slab_shift(pa.ccnt, slot_off);
        ccnt[tile_id] = s_nlog;
    }
    SSF_PASS_STAT_END(s_nlog);
    SSF_PASS_TICK(4);
