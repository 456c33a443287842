// ssf_device.hpp -- HBM layout of one handle and the kernel launch entry points.
//
// Layout (all device-resident, allocated once in ssf_create; P = W*H pixels, S superpixels,
// N = nb_supersurfels_max):
//   per-frame working set ("slab", one per frame of an extract batch, see batch_slot):
//     per-pixel maps   rgba u32[P] | disp f32[P] | label i32[P] (ONE map, relabelled in place)
//                      inlier u8[P] | plane_depth f32[P]
//     superpixel sums  two sets (double buffer) of SumRec[S]: 9 x i32 (x,y,r,g,b,n,dx,dy,dn) + 6 x i64
//                      (dxx,dyy,dxy | dxd,dyd,dd fixed point 2^30) per 128-byte record: exact integers, integer atomics only
//     pass logs        3 x (int4 + f32) x 256 entries per relabelling tile + per-tile counts
//     superpixel table SpRow[S] (48 B rows, 16 B aligned), RANSAC samples, moments, filter scratch
//     frame supersurfels + association tables (best u64[S], matched u8[S]) + input staging
//   supersurfels     structure of arrays, 3-float rows kept as separate streams so that a kernel
//                    reads only what it needs (ICP: pos 12 + lab 12 + normal 12 B per surfel):
//                    pos[3n] col[3n] lab[3n] stamps[2n] r0[3n] r1[3n] r2[3n] shape[6n] dims[2n] conf[n]
//                    (r0,r1,r2 = rows of the reference's Mat33; r2 is the normal)
//   model store      visible rows: dense SoA x 2 (rebuilt every frame); out-of-view rows: OovStore x 2
//                    (deque-like span with live flags); dense SoA for whole-model consumers
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <cstddef>
#include "ssf_math.hpp"
#include "../../include/ssf.h"

// SSF_EXPERIMENTS: the laboratory build (csrc/variants/lab/libssf_hip.so, `make lab`): the measurement arms and environment
// switches behind DESIGN.md's A/B tables.  The PRODUCT library is built without it: it reads no environment variable and
// contains no kernel arm that lost its A/B; a switch evaluates to its default at compile time.
#ifdef SSF_EXPERIMENTS
#include <stdlib.h>
static inline int ssf_env_int_(const char* name, int dflt) { const char* e = getenv(name); return e ? atoi(e) : dflt; }
#define SSF_ENV_INT(name, dflt) ssf_env_int_("SSF_" name, dflt)
#define SSF_ENV_SET(name) (getenv("SSF_" name) != nullptr)
#define SSF_ENV_STR(name) getenv("SSF_" name)
// a probe bit of a kernel's `dbg` argument (ablation switches of tools/pass_probe.py, tools/icp_probe.py: loads only / no energy /
// no accumulation ...): a run-time test in the lab build, the constant `false` in the product -- the branch and its scalar
// register are compiled out of kernels that are bound by instruction issue and sensitive to their scalar-register count
#define SSF_PROBE(dbg, bits) ((((dbg)) & (bits)) != 0)
#else
#define SSF_PROBE(dbg, bits) (false)
#define SSF_ENV_INT(name, dflt) (dflt)
#define SSF_ENV_SET(name) (false)
#define SSF_ENV_STR(name) ((const char*)nullptr)
#endif

namespace ssf {

struct SurfelSoA {
    float *pos, *col, *lab;
    int32_t* stamps;
    float *r0, *r1, *r2, *shape, *dims, *conf;
};

// exact sums of one superpixel: one 128-byte record (two cache lines: the nine int32 sums, the six int64 sums), so
// that a workgroup building the rows of its cell window touches 2 lines per superpixel instead of 15
// `stamp` (in what used to be padding; no sum): the last relabelling pass that flushed a delta into THIS copy of the record --
// what lets a tile of a later pass prove that the superpixels around it have not changed (clean-tile skipping, k_update_pass)
struct alignas(64) SumRec {
    int32_t sx, sy, sr, sg, sb, n, dx, dy, dn; int32_t stamp; int32_t pad0[6];
    long long dxx, dyy, dxy, dxd, dyd, dd; long long pad1[2];
};
static_assert(offsetof(SumRec, dn) == 32 && offsetof(SumRec, stamp) == 36 && offsetof(SumRec, dxx) == 64 && offsetof(SumRec, dd) == 104 && sizeof(SumRec) == 128,
              "k_update_pass flushes its accumulators by field offset");
#define SSF_STAMP_NEVER (-1000)           // "no pass has touched this yet" (stamps are compared with pass numbers 0 .. 4 seg_iter)
struct SpSums { SumRec* r; };

// device-side counters shared by the fuse kernels (no host round trip between them)
struct Counters {
    int n_model;      // rows in the model (after insert: includes the new rows)
    int n_visible;
    int n_inserted;
    int n_updated;
    int n_removed;
    int n_state0, n_state1, n_state2;
    // out-of-view store: live rows occupy (with holes) the span [oov_head, oov_tail) of the current OOV array
    int oov_head, oov_tail, oov_live;
    // frozen inputs of the move kernel: old visible count, class totals A0 (visible staying), B0 (out-of-view
    // becoming visible), new head / old tail of the out-of-view span
    int mv_nv, mv_a0, mv_b0, mv_head_new, mv_tail_old, mv_head_old;
    // rows appended behind the old visible rows in this frame: insertions + rows that arrived from other shards
    int mv_nc;
    // multi-GPU migration: updated rows that left this shard in this frame (dropped by the partition, not "removed")
    int n_emigrated;
    // the published counters of the last frame (n_model, n_visible, n_removed, n_inserted, n_updated): never reset,
    // the source of the per-frame RCCL all-gather of the shard sizes
    int last[5];
};
#define SSF_MAX_RANKS 64

struct Cam { float fx, fy, cx, cy; int W, H; };
struct Rt { M3 R; V3 t; };

struct SegParams {
    int W, H, cell, gx, gy, S, nb_samples, min_size;
    float lambda_pos, lambda_bound, lambda_size, lambda_disp, thresh_disp;
    float filter_alpha, filter_beta, filter_threshold;
    int filter_iter;
    uint64_t seed;
    float inv_gx;         // 1.0f / (float)gx (one IEEE division, made on the host)
    uint32_t cell_magic;  // ceil(2^32 / cell) (0 when cell == 1): x / cell == mulhi(x, cell_magic) for 0 <= x < 65536
    // per tile of the relabelling passes' grid, [OX = 0 | OX = 1]: the window of grid cells around the tile and whether the tile
    // lies inside the image (pass_geometry_table: built once per handle on the host; k_update_pass fetches one 8-byte entry
    // instead of working the same ~60 scalar instructions out in every wave of every pass)
    const uint2* pass_geom; int pass_ntile;
    int win_cells_max;    // the largest cell window of a tile of the plain 32 x 32 grid (tile_window_cells_max): <= 36 selects the small-LDS instantiations of the tile kernels
};
int tile_window_cells_max(const SegParams& p);
// entry: x = (wcx0 & 0xFFFF) | (wcy0 << 16) (first window cell, may be negative), y = nwx | nwy << 8 | interior << 16 (nwx = nwy = 0: no window)
int pass_geometry_entries(int W, int H);                                  // 2 x tiles of the shifted 32-wide grid
void pass_geometry_table(const SegParams& p, uint2* host_out);

// One relabelled pixel of a pass, replayed by the next pass into the lagging sums buffer.
// flags: 1 = label moved (from -> to), 2 = add the disparity terms to `to`, 4 = remove them from `from`
// Every workgroup (tile) owns a fixed 256-entry region per log (a pass relabels at most 256 pixels
// of a tile) and replays its own region in the next pass: no global counter, no returning atomic.
struct PassLog {
    int4* ent[3];            // [tile * 256 + i] = (from, to, x | y << 16, r | g << 8 | b << 16 | flags << 24)
    float* disp[3];
    unsigned int* count[3];  // entries per tile
};

struct FrameMaps {
    uint32_t* rgba; float* disp; int32_t* label; uint8_t* inlier; float* plane_depth;   // label: ONE map, relabelled in place
    // Exact sums, double buffered: relabelling pass k reads sums[k & 1] (quiescent: nothing writes it
    // during the pass) and applies its own deltas plus the log of pass k-1 to sums[(k + 1) & 1].
    SpSums sums[2]; PassLog log; SpRow* sp; float4* samples; int32_t* sample_score;
    // gather-friendly copies for the ICP / association kernels (one cache line per lookup instead of two / five):
    // per pixel (label, plane-depth bits); per frame supersurfel one 64-byte line (conf, lab.xyz) (normal.xyz, 0)
    // (pos.xyz, 0) (unused)
    uint2* pix2; float4* fpack;
    uint32_t* epoch;      // [0] = RNG epoch of the frame = number of frames extracted before it (written by ingest)
    long long* moments;   // 13 x i64 per superpixel
    float* filt;          // plane-filter scratch: X0[3S] X1[3S] Z[3S] px[S] py[S]
    const float* srgb_lut; // srgb_expand(c/255) for c = 0..255, built on the host with the same function
    size_t slab;          // bytes between the working sets of consecutive frames of a batch (see batch_slot)
};

// Frame batching: an extract context holds up to SSF_MAX_BATCH frames whose working sets are carved
// identically out of consecutive slabs, so frame b of a batch lives at (every pointer) + b * slab.
// The extract kernels take the batch index from the grid (blockIdx.z, or .y for 1-D kernels): one
// launch relabels the tiles of all frames of the batch.  srgb_lut is shared.
#ifndef SSF_MAX_BATCH
#define SSF_MAX_BATCH 16        // (= SSF_MAX_EXTRACT_BATCH of include/ssf.h)
#endif
// (byte arithmetic on a char pointer, not on an integer: a pointer that went through an integer loses its address
// space, and every access through it becomes a FLAT instruction -- which also ties LDS waits to outstanding loads)
template <typename T> SSF_HD T* slab_shift(T* p, size_t off) {
    return reinterpret_cast<T*>(const_cast<char*>(reinterpret_cast<const char*>(p)) + off);
}
SSF_HD SpSums batch_slot(SpSums s, size_t o) { s.r = slab_shift(s.r, o); return s; }
SSF_HD FrameMaps batch_slot(FrameMaps m, int b) {
    const size_t o = (size_t)b * m.slab;
    m.rgba = slab_shift(m.rgba, o); m.disp = slab_shift(m.disp, o); m.label = slab_shift(m.label, o);
    m.inlier = slab_shift(m.inlier, o); m.plane_depth = slab_shift(m.plane_depth, o);
    m.sums[0] = batch_slot(m.sums[0], o); m.sums[1] = batch_slot(m.sums[1], o);
    for (int i = 0; i < 3; i++) {
        m.log.ent[i] = slab_shift(m.log.ent[i], o); m.log.disp[i] = slab_shift(m.log.disp[i], o);
        m.log.count[i] = slab_shift(m.log.count[i], o);
    }
    m.sp = slab_shift(m.sp, o); m.samples = slab_shift(m.samples, o); m.sample_score = slab_shift(m.sample_score, o);
    m.moments = slab_shift(m.moments, o); m.filt = slab_shift(m.filt, o); m.epoch = slab_shift(m.epoch, o);
    m.pix2 = slab_shift(m.pix2, o); m.fpack = slab_shift(m.fpack, o);
    return m;
}
SSF_HD SurfelSoA batch_slot(SurfelSoA s, size_t o) {
    s.pos = slab_shift(s.pos, o); s.col = slab_shift(s.col, o); s.lab = slab_shift(s.lab, o); s.stamps = slab_shift(s.stamps, o);
    s.r0 = slab_shift(s.r0, o); s.r1 = slab_shift(s.r1, o); s.r2 = slab_shift(s.r2, o); s.shape = slab_shift(s.shape, o);
    s.dims = slab_shift(s.dims, o); s.conf = slab_shift(s.conf, o);
    return s;
}
// caller-side inputs of the frames of one batch
struct BatchIn {
    const uint8_t* rgb[SSF_MAX_BATCH];
    const float* depth[SSF_MAX_BATCH];
};
template <typename T> SSF_HD T batch_pick(const T* arr, int b) {     // uniform select chain (no dynamic kernarg indexing)
    T v = arr[0];
#pragma unroll
    for (int i = 1; i < SSF_MAX_BATCH; i++) v = (b == i) ? arr[i] : v;
    return v;
}

// Host-mapped (fine-grained, coherent) mailbox: the last workgroup of the ICP reduction and of the
// fuse stage publish their small results here and then store a sequence number; the host polls the
// sequence number instead of paying a DMA copy + hipStreamSynchronize per ICP iteration / frame.
// Each record carries a checksum (sum of the payload words + sequence number) so that the host can
// detect a torn read without the device paying a system-scope cache write-back per publication.
struct Mailbox {
    // ICP record: five self-validating 64-byte lines (7 payload words + the sequence number each; payload = the 29
    // sums, then the checksum).  One wave store writes all 40 words; the host accepts the record when every line
    // carries the awaited sequence number and the checksum matches -- no wait for write acknowledgements in between.
    alignas(64) unsigned long long icp_rec[40];
    Counters cnt;
    unsigned long long cnt_check;
    unsigned long long cnt_seq;
    int all_cnt[5 * SSF_MAX_RANKS];       // Counters::last of every rank (multi-GPU)
    unsigned long long all_check;
    unsigned long long all_seq;
    // peer-to-peer exchanges: set (never cleared) when a bounded wait for a peer ran out inside a kernel that has no record
    // of its own to withhold (association, migrant table); the host turns it into SSF_ERR_DEVICE at the end of the frame
    unsigned int p2p_timeout;
    // the team form of the relabelling passes (lab build): set (never cleared by the device) when a launch's workgroups could not
    // all become resident within its bound and gave up (the frames of that batch are invalid; the host reports SSF_ERR_DEVICE)
    unsigned int extract_abort;
};
#ifndef SSF_ICP_REPLICAS
#define SSF_ICP_REPLICAS 8
#endif
// word w of Mailbox::icp_rec for payload p[0..29] (29 sums + checksum) and sequence number seq
#define SSF_ICP_REC_WORD(w, p, seq) ((((w) & 7) == 7) ? (unsigned long long)(seq) : ((7 * ((w) >> 3) + ((w) & 7)) < 30 ? (unsigned long long)(p)[7 * ((w) >> 3) + ((w) & 7)] : 0ull))

// ---- extract stage (ssf_extract.hip) -----------------------------------------------------------
// every extract launch processes the nb frames of a batch (m, frame, best, matched, dynamic_mask = slot 0)
// frame k of the batch is frame number epoch0 + k of this handle (keys its RANSAC draws)
void launch_ingest(hipStream_t st, const SegParams& p, const BatchIn& in, FrameMaps& m, int nb, uint32_t epoch0);
// pass number k (0-based over the whole frame) selects label/sums/log buffers: see FrameMaps
int pass_tile_npx(int nb);     // pass pixels per thread: 1 (32-wide relabelling tiles; the 64-wide form was measured and removed)
void launch_update_pass(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int k, int ox, int oy, bool rgbd, int dbg = 0);
// The passes k0 .. k1 - 1 of a batch in ONE launch whose workgroups stay, a frame per XCD, meeting inside their XCD between passes
// (k_passes_team in ssf_extract.hip).  d_pas: the context's table of per-pass arguments (pass_args_table, pass_args_bytes(kmax) bytes
// uploaded once); d_ws: pass_team_ws_bytes() of device memory ZEROED in stream order in front of every launch.
size_t pass_team_ws_bytes();
size_t pass_args_bytes(int kmax);
void pass_args_table(const SegParams& p, const FrameMaps& m, int kmax, void* host_out);
void launch_update_passes_team(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int k0, int k1, bool rgbd, const void* d_pas, void* d_ws,
                               unsigned int* abort_flag);
void launch_init_samples(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int true_buf);
void launch_eval_samples(hipStream_t st, const SegParams& p, FrameMaps& m, int nb);
void launch_init_disp(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, bool ransac);
void launch_plane_filter(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, int true_buf);   // includes the final merge
// a frame extracted by another rank (ssf_submit_frame_tables): m.label / m.plane_depth hold its maps, wire its 26 S words
void launch_import_frame(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, SurfelSoA frame, const float* wire, unsigned long long* best, uint8_t* matched);
void launch_export_rows(hipStream_t st, const SegParams& p, const FrameMaps& m, int nb, SurfelSoA frame, float* wire);      // the nb slots' supersurfels -> their wire buffers
void launch_render_moments(hipStream_t st, const SegParams& p, const Cam& cam, FrameMaps& m, int nb);
// frame k of the batch gets stamp stamp0 + k; bit k of mask_bits: dynamic_mask slot k is valid
void launch_finalize_surfels(hipStream_t st, const SegParams& p, FrameMaps& m, int nb, SurfelSoA frame, float zmin,
                             float zmax, int stamp0, const uint8_t* dynamic_mask, unsigned mask_bits,
                             unsigned long long* best, uint8_t* matched);
void launch_bilateral(hipStream_t st, const float* in, float* out, int W, int H, float sigma_color, float sigma_space);
// the nb frames of a batch in ONE launch: frame b from in.depth[b] to out0 + b * slab bytes
void launch_bilateral_batch(hipStream_t st, const BatchIn& in, float* out0, size_t slab, int nb, int W, int H, float sigma_color, float sigma_space);
void launch_boundary_map(hipStream_t st, const SegParams& p, const int32_t* label, int32_t* out);
void launch_preview(hipStream_t st, int W, int H, const int32_t* label, const uint32_t* rgba, uint8_t* out /* 3P */);

// ---- ICP + fuse (ssf_track_fuse.hip) -----------------------------------------------------------
// replicas: SSF_ICP_REPLICAS x 29 zero-initialised i64 (left zeroed again by the kernel), ticket: 65 zeroed u32 (global + 64 group arrival counters)
// pix2 / fpack: FrameMaps::pix2 / fpack of the frame
// go != nullptr: the launch is made AHEAD of its transform (while the previous iteration is still running, so that its
// launch latency is off the host round trip): the workgroups wait until the host's flag word carries go_seq (then
// the transform is in go->T) or go_seq with SSF_ICP_GO_ABORT (no further iteration: leave at once).  IcpGo lives in
// fine-grained DEVICE memory that the host writes directly (large BAR): ~1 us from the host's store to the kernel's
// eyes (tools/probe/bar_write.hip), against ~5 us for a launch to start.
// go_seq | SSF_ICP_GO_MATCH (a launch made with `match`): the loop has ended and go->T holds the frame's final POSE -- the
// waiting launch does the association (k_match's work) instead of an iteration: the association then starts ~1 us after
// the host's decision instead of a launch latency later.
// One slot = ONE 64-byte line: the host writes transform, peer-exchange number and flag word into its write-combining mapping
// and fences once (one PCIe write of the whole line), and the sixteen lanes that POLL the slot fetch all of it with every poll --
// the transform needs no trip of its own behind the word (it used to: ~1 us at the head of every chained iteration).  Should the
// line ever arrive in pieces, the flag word says so: bits 0-31 the low half of go_seq, bits 32-61 a checksum of the other
// fourteen words (icp_go_word_weight), bit 62 "associate", bit 63 "leave".  "Leave" is self-validating too: its bits 32-61 are a
// hash of go_seq (icp_go_abort_check), so a poll that sees the NEW low half next to a stale high half -- the slot's previous use
// may have ended in "leave" -- keeps polling instead of leaving a launch the host has just told to associate (the flag is read
// as two 32-bit halves of the sixteen-lane line fetch, not as one 64-bit atomic).
struct alignas(64) IcpGo { float T[12]; unsigned long long flag; unsigned long long x; };
static_assert(sizeof(IcpGo) == 64, "one line");
#define SSF_ICP_GO_ABORT (1ull << 63)
#define SSF_ICP_GO_MATCH (1ull << 62)
#define SSF_ICP_GO_CHECK_MASK 0x3FFFFFFFu
SSF_HD unsigned int icp_go_word_weight(unsigned int dword) {      // dword 0..11: the transform; 14, 15: x; 12, 13 (the flag itself): 0
    return dword < 12u ? (2u * dword + 1u) * 0x9E3779B1u : (dword == 14u ? 0x85EBCA6Bu : (dword == 15u ? 0xC2B2AE35u : 0u));
}
SSF_HD unsigned int icp_go_abort_check(unsigned int want) { return ((want * 0x9E3779B1u) >> 2) & SSF_ICP_GO_CHECK_MASK; }
#define SSF_ICP_GO_SLOTS 4
struct MatchArgs { float zmin, zmax; long long id_offset; unsigned long long* best; uint8_t* matched; int32_t* cand; };    // launch_match's arguments
int icp_variant_mode();              // 0: the product's k_icp; other values: measurement arms that cannot take SSF_ICP_GO_MATCH
void launch_icp(hipStream_t st, const Cam& cam, SurfelSoA model, int n_visible, const uint2* pix2, const float4* fpack,
                Rt T, long long* replicas, unsigned int* ticket,
                long long* sums29, Mailbox* mb, unsigned long long seq, int dbg = -1, IcpGo* go = nullptr,
                unsigned long long go_seq = 0, const struct P2PView* pv = nullptr, int by_tile = 0, const MatchArgs* match = nullptr);
// Tile-sorted copy of the ICP / association fields of the n visible rows of `model` (pos, lab, r2, conf -> the same
// streams of `out`; out_idx[j] = the row's index in the visible array) under transform T (model -> camera): see k_bin_* in
// ssf_track_fuse.hip.  count / cursor: bin_buffer_words(cam, capacity) words each.  launch_icp(by_tile = 1) /
// launch_match(orig = out_idx) then take `out` as their rows.
void launch_bin_rows(hipStream_t st, const Cam& cam, SurfelSoA model, int n, Rt T, uint32_t* count, uint32_t* cursor, SurfelSoA out);
size_t bin_buffer_words(const Cam& cam, size_t capacity);
void launch_match(hipStream_t st, const Cam& cam, SurfelSoA model, int n_visible, const uint2* pix2, const float4* fpack,
                  Rt pose, float zmin, float zmax, long long id_offset,
                  unsigned long long* best, uint8_t* matched, int32_t* cand /* per row: frame supersurfel bid for, -1 none */, int S,
                  const int32_t* orig = nullptr /* model = tile-sorted copy: index of each row in the visible array */);
struct OovStore { SurfelSoA rows; uint8_t* live; int cap; };      // out-of-view rows of the model store (see below)
// Sums the per-frame stable partition works from (k_update_insert / k_move_rows in ssf_track_fuse.hip): per group of PART_GROUP blocks
// the class counts (sup_vis: 6 per group, sup_oov: rows that come back into view), and the frame totals
// a0 a1 a2 c0 c1 c2 b0 b2 in PART_REPLICAS copies of 8 words.  Two sets alternate by frame: `other` (all `words`
// of it) is cleared by the frame that uses this one.  ticket: arrival counters of the fuse launch (65 words, zero at rest).
#define PART_GROUP 16
#define PART_REPLICAS 8
struct PartitionWs { uint32_t* sup_vis; uint32_t* sup_oov; uint32_t* tot; uint32_t* ticket; uint32_t* other; int words; };
// The fuse launch (k_update_insert in ssf_track_fuse.hip): update of the matched rows | ordered insertion of the
// unmatched frame supersurfels | classification (filterModel) of every row of the model store, and publication of the
// frame's counters by the last block to finish (cnt[0] = what the next frame starts from, cnt[1] = what
// launch_move_rows sends to the host: sequence number cnt_seq there).  do_update = 0 skips the update (no visible rows anywhere);
// span_upper = host upper bound of the out-of-view span; cand = launch_match's per-row candidate.
void launch_fuse(hipStream_t st, SurfelSoA model /* visible array */, SurfelSoA frame, Rt pose, int stamp, long long id_offset,
                 int n_visible, const unsigned long long* best, const uint8_t* matched, const int32_t* cand, int S, int do_update,
                 int capacity, int rank, int nranks, float tile, Counters* cnt, const Cam& cam, OovStore oov,
                 int span_upper, const float* plane_depth, int delta_t, float conf_thresh, float zmin, float zmax,
                 uint8_t* state_vis, uint8_t* state_oov, uint32_t* bc_oov, const PartitionWs& ws, int migrate = 0,
                 int tail_in_move = 0 /* the launch ends without turning the class totals into counters: launch_move_rows(totals) does */);
// launch_move_rows(totals != nullptr): the fuse launch ended without its tail; every block of the move kernel takes the old
// counts from here (the host mirrors them) and the class totals from the partition's replicas, block 0 finalises the counters
#ifdef SSF_EXPERIMENTS
void set_pass_stats(int on);         // (lab: the relabelling statistics of ssf_dbg_pass_stats are collected only while set)
void set_pass_trace(unsigned long long* device_words /* 5 per workgroup of a relabelling pass launch; nullptr: off */);     // (lab: tools/pass_trace.py)
void set_fuse_trace(unsigned long long* device_words /* 3 per workgroup of the fuse launch; nullptr: off */);      // (lab: tools/fuse_probe.py)
#endif
struct MoveTotals { int from_tot, nv /* visible rows before the frame */, head_old, tail_old /* out-of-view span before the frame */; };
// Multi-GPU migration (ssf_stage_fuse_begin / _end in ssf.h).  launch_fuse(migrate = 1) marks an updated row whose new
// position belongs to another rank's world tile as leaving (partition class "dropped", confidence kept);
// launch_pack_emigrants writes those rows to slot f of the migrant table (SSF_MIGRANT_WORDS words per slot, other
// slots zero); after the tables of all ranks have been summed, launch_migrate_in appends the rows addressed to this
// rank behind this frame's insertions (ascending f), classifies them and corrects the partition sums and both
// counter sets; launch_move_rows then treats them like inserted rows.
void launch_pack_emigrants(hipStream_t st, SurfelSoA model, const unsigned long long* best, const uint8_t* matched, long long id_offset,
                           int n_visible, const uint8_t* state_vis, int S, int do_update, int nranks, float tile, int32_t* table);
void launch_migrate_in(hipStream_t st, SurfelSoA model, const int32_t* table, int S, int rank, int capacity, Counters* cnt,
                       const Cam& cam, Rt pose, int stamp, const float* plane_depth, int delta_t, float conf_thresh, float zmin,
                       float zmax, uint8_t* state_vis, const PartitionWs& ws);
// re-homing of a sharded map (ssf_rehome_begin / _end): split the dense view into the rows that stay (-> stay, ranks closed,
// visible block first) and the rows that leave (-> migrant-table records); bc: 3 words per block of 256 rows, tot3: totals
// (staying, leaving, staying rows of the visible block); unpack: records -> rows [base, base + n) of dst
void launch_rehome_split(hipStream_t st, SurfelSoA dense, int n, int n_visible, int rank, int nranks, float tile, uint32_t* bc, int* tot3,
                         SurfelSoA stay, int32_t* table, int table_rows);
void launch_rehome_unpack(hipStream_t st, const int32_t* table, int n, SurfelSoA dst, int base);
void launch_first_frame(hipStream_t st, SurfelSoA model, SurfelSoA frame, Rt pose, int S, int capacity, int rank,
                        int nranks, float tile, Counters* cnt);
// first ICP iteration of the next frame, accumulated by the row-move kernel of this one (launch_classify_reorder)
struct NextFrameIcp {
    const uint2* pix2; const float4* fpack;   // packed tables of the next frame
    Rt T;                                     // model -> camera transform of that iteration
    long long* replicas; unsigned int* ticket; long long* sums; unsigned long long seq;
    const struct P2PView* pv;                 // non-null: the record is traded with the peers (see P2PView below)
};
// Model store: the visible rows are a dense array (two of them, ping-pong); the out-of-view
// rows live in a deque-like store with a live flag per row.  The per-frame stable partition
//   [visible | out-of-view] = [A0 B0 C0 | A1 B1 C1]   (A = old visible rows, B = old out-of-view rows, C = rows
//   inserted this frame; 0 = classified visible, 1 = out of view, 2 = removed)
// then only moves A0/B0/C0 into the other visible array, pushes A1 in front of the out-of-view span, appends C1
// behind it and clears the live flag of B0/B2: the (large) B1 block is never touched.
// launch_move_rows moves the rows accordingly (after launch_fuse); nv_upper / span_upper = host upper bounds of the
// visible rows (incl. insertions) and of the out-of-view span; next != nullptr: also accumulate the next frame's
// first ICP iteration
void launch_move_rows(hipStream_t st, const Cam& cam, SurfelSoA vis_src, SurfelSoA vis_dst, OovStore oov, int nv_upper, int span_upper,
                      const uint8_t* state_vis, const uint8_t* state_oov, const uint32_t* bc_oov, const PartitionWs& ws,
                      Counters* cnt /* [2]: see launch_fuse */, Mailbox* mb, unsigned long long cnt_seq, const NextFrameIcp* next,
                      const MoveTotals* totals = nullptr);
// stable compaction of the live out-of-view rows of src (span from the device counters) into dst starting at
// new_head (dst.live must be zero where it matters); set_span != 0: cnt->oov_head / oov_tail := the new span
void launch_oov_compact(hipStream_t st, OovStore src, OovStore dst, int span_upper, int new_head, uint32_t* bc_oov, Counters* cnt,
                        int set_span);
// one iteration of the loop-closure registration against a frame; out40: see k_align
void launch_align(hipStream_t st, const Cam& cam, const float* spos, const float* slab, const float* snrm, const float* sconf,
                  int n, SurfelSoA frame, const int32_t* label, const float* plane_depth, Rt T, long long* out40);
void launch_fern_codes(hipStream_t st, const uint8_t* rgb, const float* depth, int W, int H, const uint32_t* fpos,
                       const uint8_t* frgb, const float* fdepth, int n, uint8_t* codes);
// ---- peer-to-peer exchanges (ssf_p2p_* in ssf.h) ----------------------------------------------
// The ranks of one node exchange their small per-frame records through memory instead of through collective launches:
// every rank owns an exchange REGION in its HBM (plain device memory, exported to the other processes through an IPC handle)
// with one slot per (exchange kind, parity of the exchange's sequence number, source rank).  A producer stores its
// record straight into slot [.][.][me] of every peer's region (xGMI stores) and then waits in its OWN region for the
// slots of the others -- an all-gather by remote stores followed by a local, fixed-order reduction.  Two parities are
// enough: a rank can start exchange n + 2 of a kind only after every peer has produced n + 1, i.e. consumed n.
//   ICP record   40 x u64 self-validating lines exactly as Mailbox::icp_rec (no ordering between the stores needed)
//   shard sizes  one self-validating 64-byte line (Counters::last, checksum, sequence number)
//   association  best u64[S] + matched u8[S], then a flag line (release) -- consumer: MIN / OR into its own tables
//   migrants     the table of ssf_stage_fuse_begin (28 x i32 per slot; empty slots send their first word only),
//                then a flag line -- consumer: adds the peers' slots into its own table (at most one rank fills a slot)
// Everything a peer reads or writes in a region goes through system-scope atomic loads / stores (write-through,
// cache-bypassing), so no cache write-back / invalidate is part of the protocol.
#define SSF_P2P_MAX_RANKS 8
// timeout_ticks: bound of every in-kernel wait for a peer, in ticks of wall_clock64() (ssf_p2p_configure)
struct P2PView { unsigned char* peer[SSF_P2P_MAX_RANKS]; int me, nranks, S; unsigned long long seq, timeout_ticks; };
enum { P2P_FLAG_ASSOC = 0, P2P_FLAG_MIGR = 1 };
#define SSF_P2P_HEADER 8192
SSF_HD size_t p2p_slot(int par, int src) { return (size_t)(par * SSF_P2P_MAX_RANKS + src); }
SSF_HD size_t p2p_spad(int S) { return ((size_t)S + 63) & ~(size_t)63; }
SSF_HD size_t p2p_off_icp(int par, int src) { return p2p_slot(par, src) * 320; }
SSF_HD size_t p2p_off_cnt(int par, int src) { return 16 * 320 + p2p_slot(par, src) * 64; }
SSF_HD size_t p2p_off_flag(int kind, int par, int src) { return 16 * 384 + ((size_t)kind * 16 + p2p_slot(par, src)) * 64; }
SSF_HD size_t p2p_off_best(int S, int par, int src) { return SSF_P2P_HEADER + p2p_slot(par, src) * (size_t)S * 8; }
SSF_HD size_t p2p_off_matched(int S, int par, int src) { return SSF_P2P_HEADER + 16 * (size_t)S * 8 + p2p_slot(par, src) * p2p_spad(S); }
SSF_HD size_t p2p_off_migr(int S, int par, int src) {
    return SSF_P2P_HEADER + 16 * (size_t)S * 8 + 16 * p2p_spad(S) + p2p_slot(par, src) * (size_t)S * SSF_MIGRANT_WORDS * 4;
}
SSF_HD size_t p2p_region_bytes(int S) { return p2p_off_migr(S, 1, SSF_P2P_MAX_RANKS - 1) + (size_t)S * SSF_MIGRANT_WORDS * 4; }
// launch_icp with pv != nullptr: the launch's last workgroup exchanges the shard record with the peers (pv->seq) and
// publishes the SUM over the ranks (device record + mailbox), one launch per iteration as on a single GPU
void launch_p2p_counts(hipStream_t st, const P2PView& pv, const Counters* cnt, Mailbox* mb, unsigned long long all_seq);
void launch_p2p_assoc(hipStream_t st, const P2PView& pv, unsigned long long* best, uint8_t* matched, Mailbox* mb);
void launch_p2p_migrants(hipStream_t st, const P2PView& pv, int32_t* table, unsigned int* ticket /* 65 zeroed words */, Mailbox* mb);
void launch_publish_icp(hipStream_t st, const long long* rec29, Mailbox* mb, unsigned long long seq);
void launch_publish_all_counts(hipStream_t st, const int* all5, int nranks, Mailbox* mb, unsigned long long seq);
// publish the counters to the mailbox (sequence number seq) and reset the per-frame ones
void launch_publish_counts(hipStream_t st, Counters* cnt, int shrink_by_removed, Mailbox* mb, unsigned long long seq);
void launch_lab_refresh(hipStream_t st, SurfelSoA s, int n);
void launch_pack_orient(hipStream_t st, SurfelSoA s, int n, float* out9);
// m nodes; nodes16: scratch of 16 floats per node (one 64-byte record) (the packed node records, see k_pack_nodes)
void launch_deformation(hipStream_t st, SurfelSoA model, int n, int m, const float* npos, const float* nrot,
                        const float* ntrans, float* nodes16, const float* w4, const int32_t* idx4);

// profiling hook: every launch_* brackets its kernels through these (ssf_host.hip)
struct KernelTimer;
KernelTimer* current_timer();
void set_current_timer(KernelTimer* t);
void timer_begin(KernelTimer* t, const char* name, hipStream_t st);
void timer_end(KernelTimer* t, hipStream_t st);
struct ScopedKernel {
    KernelTimer* t; hipStream_t st;
    ScopedKernel(const char* name, hipStream_t s) : t(current_timer()), st(s) { if (t) timer_begin(t, name, st); }
    ~ScopedKernel() { if (t) timer_end(t, st); }
};

}  // namespace ssf
