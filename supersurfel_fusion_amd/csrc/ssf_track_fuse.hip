// ssf_track_fuse.hip -- ICP normal-equation accumulation and the model fuse kernels for gfx950.
//
// What is computed follows the reference: computeSymmetricICPSystem
// (core/include/supersurfel_fusion/dense_registration_kernels.cuh:175-291), findBestMatches /
// updateSupersurfels / insertSupersurfels / filterModel (core/src/supersurfel_fusion_kernels.cu:
// 348-467,522-682), thrust::sort_by_key over the model (core/src/supersurfel_fusion.cu:469-472) and
// applyDeformation (core/src/deformation_graph_kernels.cu:27-73).  How:
//   * ICP streams 36 B per visible supersurfel (pos, cached Lab, normal row) from separate SoA
//     streams; an inlier adds its 29 fixed-point terms with LDS integer atomics, every workgroup
//     issues 29 global integer atomics, the last workgroup (two-level arrival counter) publishes
//     the record: bit-identical for any grid, block or rank decomposition (no float atomics, no
//     14.8 KB LDS tree).
//   * association is one packed (dist_bits<<32 | id) 64-bit atomicMin per candidate: exact arg-min,
//     ties to the lowest id.
//   * insertion is an ordered block scan (ascending frame id), not atomic arrival order.
//   * classify + stable partition replaces the radix sort: one pass computes the state and
//     per-block class histograms, a one-workgroup scan turns them into offsets (and publishes the
//     frame's counters), one pass moves only the rows whose place changes (visible rows are a dense
//     array, out-of-view rows a deque-like store with live flags: OovStore, ssf_device.hpp).
//   * the loop-closure registration (DenseRegistration::align) is one single-workgroup launch per
//     iteration with exact sums; fern codes are a gather.
#include <stdlib.h>
#include <algorithm>
#include "ssf_device.hpp"

namespace ssf {

__device__ __forceinline__ int lane() { return threadIdx.x & 63; }
__device__ __forceinline__ long long wsum64(long long v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}
__device__ __forceinline__ V3 ld3(const float* __restrict__ p, size_t i) { return v3(p[3 * i], p[3 * i + 1], p[3 * i + 2]); }
__device__ __forceinline__ void st3(float* __restrict__ p, size_t i, V3 v) { p[3 * i] = v.x; p[3 * i + 1] = v.y; p[3 * i + 2] = v.z; }
__device__ __forceinline__ Sym3 ld6(const float* __restrict__ p, size_t i) {
    return sym3(p[6 * i], p[6 * i + 1], p[6 * i + 2], p[6 * i + 3], p[6 * i + 4], p[6 * i + 5]);
}
__device__ __forceinline__ void st6(float* __restrict__ p, size_t i, Sym3 c) {
    p[6 * i] = c.xx; p[6 * i + 1] = c.xy; p[6 * i + 2] = c.xz; p[6 * i + 3] = c.yy; p[6 * i + 4] = c.yz; p[6 * i + 5] = c.zz;
}

// ---- ICP ---------------------------------------------------------------------------------------
// One supersurfel per lane.  An inlier adds its 29 fixed-point terms with LDS integer atomics into a
// 29 x 16 table of the workgroup (lane & 15 spreads the same-address traffic); 29 threads fold the
// table and issue one global integer atomic each into one of SSF_ICP_REPLICAS replica records; the
// workgroup that arrives last sums the replicas (agent-scope loads), leaves them zeroed for the next
// launch and publishes the record to device memory and to the host-mapped mailbox.  All adds are
// integer, so the record is independent of every one of these decompositions.
#ifndef ICP_SLOTS
#define ICP_SLOTS 16      // lane & 15 spreads the same-address traffic (64 columns measured no faster)
#endif
// the 29 terms of one visible supersurfel (position, cached Lab, normal row) under transform (R, t), each handed to
// emit(k, value) as soon as it is computed (k is a compile-time constant after unrolling): JtJ k = 0..20 (fixed point 2^20),
// Jtr 21..26 (2^24), squared residual 27 (2^44), inlier count 28
// The 29 terms of one visible supersurfel (position, cached Lab, normal row) under transform (R, t), each handed to
// emit(k, value) as soon as it is computed (k is a compile-time constant after unrolling): JtJ k = 0..20 (fixed point 2^20),
// Jtr 21..26 (2^24), squared residual 27 (2^44), inlier count 28.  Gates: dense_registration_kernels.cuh:217-249.
// Fixed point: JtJ at 2^20 -- the two rows scaled by 2^10 once instead of 21 sums by 2^20 --, Jtr at 2^24 -- the two residuals
// scaled by 2^14 once, times the scaled rows (the specification, DESIGN.md section 2: the checker performs the same
// operations).
// UNIFORM = false: a row that is no inlier emits nothing (its lane leaves).  UNIFORM = true (the wave-reduction arm): EVERY lane
// emits all 29 terms, a row that is no inlier as zeros -- emit is then called in uniform control flow, as cross-lane
// operations need.
template <bool UNIFORM, typename Emit>
__device__ __forceinline__ void icp_row_terms_(const Cam& cam, const uint2* __restrict__ pix2, const float4* __restrict__ fpack,
                                               const M3& R, const V3& t, const V3& mpos, const V3& mlab, const V3& mnrm, int dbg, Emit emit,
                                               bool live = true) {
    const V3 ps = add(m3_mulv(R, mpos), t);
    if (SSF_PROBE(dbg, 1)) { if (ps.z > 1e30f) emit(28, 1ll); return; }
    const int u = pixel_round(ps.x * cam.fx / ps.z + cam.cx);
    const int v = pixel_round(ps.y * cam.fy / ps.z + cam.cy);
    bool ok = live && u >= 0 && u < cam.W && v >= 0 && v < cam.H;
    if (!UNIFORM && !ok) return;
    size_t q = ok ? (size_t)v * cam.W + u : 0;
    if (SSF_PROBE(dbg, 16)) q = (blockIdx.x * blockDim.x + threadIdx.x) & 0x7ffffu;          // probe: the lanes of a wave gather ADJACENT pixels (the first 2^19 of the table)
    const uint2 pl = pix2[q];                                    // (label, plane depth) of the pixel: one 8-byte gather
    const int tid = SSF_PROBE(dbg, 8) ? (int)(q & 1023) : (int)pl.x;     // probe (8): the frame supersurfel does not depend on the pixel's word -- two trips instead of three
    const float zt = __uint_as_float(pl.y);
    float4 f0 = fpack[4 * tid], f1 = fpack[4 * tid + 1];         // (conf, lab) (normal) of the frame supersurfel: one 32-byte gather
    // (keeps the gather whole: left alone, the compiler fetches the confidence first, tests it, and only then the
    // rest -- a second dependent round trip)
    asm volatile("" : "+v"(f0.y), "+v"(f1.x));
    ok = ok && (f0.x > 0.0f && zt >= 0.2f && zt <= 5.0f);
    if (!UNIFORM && !ok) return;
    const float dist_color = len3(sub(mlab, v3(f0.y, f0.z, f0.w)));
    const V3 pt = v3(zt * ((float)u - cam.cx) / cam.fx, zt * ((float)v - cam.cy) / cam.fy, zt);
    const V3 nt = v3(f1.x, f1.y, f1.z);
    const V3 ns = unit3(m3_mulv(R, mnrm));
    ok = ok && (dist_color < 20.0f && len3(sub(ps, pt)) < 0.1f && fabsf(dot3(nt, ns)) > 0.8f);
    if (!UNIFORM && !ok) return;
    const V3 d = sub(pt, ps), c1 = cross3(pt, ns), c2 = cross3(ps, nt);
    const float dn1 = dot3(d, ns), dn2 = dot3(d, nt);
    const float x1[6] = {c1.x, c1.y, c1.z, ns.x, ns.y, ns.z};
    const float x2[6] = {c2.x, c2.y, c2.z, nt.x, nt.y, nt.z};
    if (SSF_PROBE(dbg, 2)) { if (x1[0] * x2[0] > 1e30f) emit(0, 1ll); return; }
    // (UNIFORM: a lane that is no inlier carries zeros -- every term of it is then exactly 0, whatever the lane had computed)
    const float m10 = (UNIFORM && !ok) ? 0.0f : 1024.0f, m14 = (UNIFORM && !ok) ? 0.0f : 16384.0f;
    float X1[6], X2[6];
#pragma unroll
    for (int i = 0; i < 6; i++) { X1[i] = UNIFORM && !ok ? 0.0f : x1[i] * m10; X2[i] = UNIFORM && !ok ? 0.0f : x2[i] * m10; }
    const float D1 = UNIFORM && !ok ? 0.0f : dn1 * m14, D2 = UNIFORM && !ok ? 0.0f : dn2 * m14;      // 2^14 x 2^10 (the scaled rows) = 2^24
    int k = 0;
#pragma unroll
    for (int i = 0; i < 6; i++)
#pragma unroll
        for (int j = i; j < 6; j++, k++) emit(k, (long long)fx32r(X1[i] * X1[j] + X2[i] * X2[j]));
#pragma unroll
    for (int i = 0; i < 6; i++) emit(21 + i, (long long)fx32r(D1 * X1[i] + D2 * X2[i]));
    emit(27, fx64((double)(UNIFORM && !ok ? 0.0f : dn2 * dn2), 17592186044416.0, 4611686018427387904.0));
    emit(28, (UNIFORM && !ok) ? 0ll : 1ll);
}
template <typename Emit>
__device__ __forceinline__ void icp_row_terms(const Cam& cam, const uint2* __restrict__ pix2, const float4* __restrict__ fpack,
                                              const M3& R, const V3& t, const V3& mpos, const V3& mlab, const V3& mnrm, int dbg, Emit emit) {
    icp_row_terms_<false>(cam, pix2, fpack, R, t, mpos, mlab, mnrm, dbg, emit);
}
// one row per thread: every term straight into the workgroup's LDS table (lane & 15 spreads the same-address traffic)
__device__ __forceinline__ void icp_row(const Cam& cam, const uint2* __restrict__ pix2, const float4* __restrict__ fpack,
                                        const M3& R, const V3& t, const V3& mpos, const V3& mlab, const V3& mnrm,
                                        unsigned long long* red, int slot, int dbg) {
    icp_row_terms(cam, pix2, fpack, R, t, mpos, mlab, mnrm, dbg, [&](int k, long long v) { atomicAdd(&red[k * ICP_SLOTS + slot], (unsigned long long)v); });
}
template <typename T> __device__ __forceinline__ void atomic_add_done(T* p, T v);
// end of an accumulating kernel, in three steps called by every thread of a workgroup after a barrier:
// fold the workgroup's table into a replica record ...
__device__ __forceinline__ void icp_fold(unsigned long long* red, long long* __restrict__ replicas) {
    if (threadIdx.x < 29) {
        unsigned long long tot = 0;
#pragma unroll
        for (int sidx = 0; sidx < ICP_SLOTS; sidx++) tot += red[threadIdx.x * ICP_SLOTS + sidx];
        long long* rep = replicas + (size_t)(blockIdx.x % SSF_ICP_REPLICAS) * 32;
        if (tot != 0) atomic_add_done(reinterpret_cast<unsigned long long*>(&rep[threadIdx.x]), tot);     // (see atomic_add_done below)
    }
    // The replica updates above are device-scope atomic RMWs; they have completed (vmcnt(0) + barrier) before this
    // workgroup counts its arrival, and the last workgroup reads them back with device-scope atomic loads.
    // Everything exchanged between workgroups is an atomic at the coherence point, so no cache write-back /
    // invalidate fence is needed.
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}
// What a workgroup hands to the LAST workgroup of its launch (the one that sums up and publishes) must have been PERFORMED at
// the coherence point before the workgroup counts its arrival.  For an atomic without return value the wait counter
// (s_waitcnt vmcnt(0)) only says that the request has been accepted: with several handles hammering the fabric side by
// side (tools/p2p_first_frame_stress.py: four ranks of a sharded map on one GPU) the last workgroup was seen reading a
// replica record / a partition total BEFORE another workgroup's add had landed -- one frame in a few hundred came out a
// workgroup's worth short.  The value a RETURNING atomic brings back is the proof that it has been performed; these
// helpers are used for everything the last workgroup of the same launch reads.  (What the NEXT launch reads needs
// nothing: a kernel boundary completes everything.)
template <typename T> __device__ __forceinline__ void atomic_add_done(T* p, T v) {
    const T before = atomicAdd(p, v);
    asm volatile("" :: "v"(before));
}
__device__ __forceinline__ void atomic_store_done(int* p, int v) {
    const int before = __hip_atomic_exchange(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    asm volatile("" :: "v"(before));
}
// ... count the arrival (is this the last workgroup?) ...
// two-level arrival count over all workgroups of the grid (a single counter serialises thousands of same-address
// atomics at L2): 64 group counters, the last workgroup of a group reports to the global counter.  Thread 0 only.
#ifndef ARRIVE_GROUPS
#define ARRIVE_GROUPS 64u
#endif
// SSF_ARRIVE_FENCED (an experiment build, tools/build_variant.sh fenced -DSSF_ARRIVE_FENCED): real
// agent-scope release / acquire ordering around the arrival ticket -- one fence per workgroup, not per atomic -- to tell
// whether the first-frame divergence seen with UNCACHED exchange regions (round 2) was the relaxed arrival protocol's.
#if defined(SSF_EXPERIMENTS) && defined(SSF_ARRIVE_FENCED)
#define SSF_ARRIVE_RELEASE() do { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent"); asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); } while (0)
#define SSF_ARRIVE_ACQUIRE() __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent")
#else
#define SSF_ARRIVE_RELEASE() do { } while (0)
#define SSF_ARRIVE_ACQUIRE() do { } while (0)
#endif
__device__ __forceinline__ int grid_arrive(unsigned int* ticket) {
    SSF_ARRIVE_RELEASE();
    const unsigned int g = blockIdx.x & (ARRIVE_GROUPS - 1u);
    const unsigned int in_group = (gridDim.x - g + ARRIVE_GROUPS - 1u) / ARRIVE_GROUPS, groups = min(gridDim.x, ARRIVE_GROUPS);
    int last = 0;
    const unsigned int tk = __hip_atomic_fetch_add(&ticket[1 + g], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    if (tk == in_group - 1) {
        __hip_atomic_store(&ticket[1 + g], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const unsigned int tg = __hip_atomic_fetch_add(&ticket[0], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (tg == groups - 1) { last = 1; __hip_atomic_store(&ticket[0], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
    }
    if (last) SSF_ARRIVE_ACQUIRE();
    return last;
}
// ---- peer-to-peer exchange helpers (ssf_device.hpp: P2PView) ----------------------------------------------------------
__device__ __forceinline__ unsigned char* p2p_peer(const P2PView& pv, int r) {       // uniform select chain (no dynamic kernarg indexing)
    unsigned char* v = pv.peer[0];
#pragma unroll
    for (int i = 1; i < SSF_P2P_MAX_RANKS; i++) v = (r == i) ? pv.peer[i] : v;
    return v;
}
__device__ __forceinline__ unsigned long long shfl_u64(unsigned long long v, int src) {
    const unsigned int lo = (unsigned int)__shfl((int)(unsigned int)v, src), hi = (unsigned int)__shfl((int)(unsigned int)(v >> 32), src);
    return ((unsigned long long)hi << 32) | lo;
}
// A peer that never arrives must not hang the device: every wait for a peer is bounded by WALL-CLOCK time (the constant-rate
// counter behind wall_clock64(), P2PView::timeout_ticks = ssf_p2p_configure's timeout), not by a spin count -- ranks in
// separate processes reach their first exchange seconds apart (set_model of a large map, graph captures, a cold box).
struct P2PDeadline {
    unsigned long long t0, ticks; unsigned int spins;
    __device__ __forceinline__ explicit P2PDeadline(const P2PView& pv) : t0(wall_clock64()), ticks(pv.timeout_ticks), spins(0u) {}
    // call once per unsuccessful poll; true = give up
    __device__ __forceinline__ bool expired() {
        __builtin_amdgcn_s_sleep(1);
        if ((++spins & 1023u) != 0u) return false;
        return wall_clock64() - t0 > ticks;
    }
};
// Wave 0 of the last workgroup: this rank's record (tot in lanes 0..28) goes into slot [parity][me] of every peer's region
// as five self-validating lines (the format of Mailbox::icp_rec); the records of the others are awaited in this rank's
// own region and added in rank order.  Returns false when a peer's record never arrived.
__device__ __forceinline__ bool p2p_icp_exchange(const P2PView& pv, unsigned long long seq, long long& tot, unsigned long long* pay /* LDS, 30 */) {
    const int l = lane();
    const int par = (int)(seq & 1ull);
    if (l < 29) pay[l] = (unsigned long long)tot;
    const unsigned long long check = (unsigned long long)wsum64(l < 29 ? tot : 0ll) + seq;
    if (l == 0) pay[29] = check;
    __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): one wave, its LDS writes have landed
    const unsigned long long word = l < 40 ? SSF_ICP_REC_WORD(l, pay, seq) : 0ull;
#pragma unroll
    for (int r = 0; r < SSF_P2P_MAX_RANKS; r++) {
        if (r >= pv.nranks || r == pv.me) continue;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(pv.peer[r] + p2p_off_icp(par, pv.me));
        if (l < 40) __hip_atomic_store(&dst[l], word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned char* mine = p2p_peer(pv, pv.me);
    const bool is_seq_word = l < 40 && (l & 7) == 7, is_payload = l < 40 && (l & 7) != 7 && (7 * (l >> 3) + (l & 7)) < 29;
    P2PDeadline deadline(pv);
    for (int r = 0; r < pv.nranks; r++) {
        if (r == pv.me) continue;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(mine + p2p_off_icp(par, r));
        for (;;) {
            const unsigned long long w = l < 40 ? __hip_atomic_load(&src[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
            const bool lines_ok = __ballot(is_seq_word && w != seq) == 0ull;
            const unsigned long long sum = (unsigned long long)wsum64(is_payload ? (long long)w : 0ll) + seq;
            if (lines_ok && sum == shfl_u64(w, 33)) {                   // payload word 29 (the checksum) sits in lane 8 * 4 + 1
                const unsigned long long v = shfl_u64(w, l < 29 ? 8 * (l / 7) + l % 7 : 0);       // payload word l of the peer
                if (l < 29) tot += (long long)v;
                break;
            }
            if (deadline.expired()) return false;
        }
    }
    return true;
}
// ... and, in the last workgroup, sum the replicas and publish the record (P2P: the SUM of the records of all ranks)
// (the number of the peer exchange travels beside the view, not in it: a kernel that writes into its P2PView argument gets a
// private copy of the whole struct in scratch memory)
template <bool P2P>
__device__ __forceinline__ void icp_publish(long long* __restrict__ replicas, long long* __restrict__ sums, Mailbox* mb,
                                            unsigned long long seq, const P2PView& pv, unsigned long long p2p_seq) {
    // SSF_ICP_REPLICAS x 32 replica words: thread t sums field t & 31 over every 8th replica (independent
    // loads, one round trip), then 8 partial rows are folded through LDS
    __shared__ long long part[8 * 32];
    const int field = threadIdx.x & 31, group = threadIdx.x >> 5;
    long long v = 0;
#pragma unroll
    for (int j = 0; j < (SSF_ICP_REPLICAS + 7) / 8; j++) {
        const int r = group + 8 * j;
        if (r < SSF_ICP_REPLICAS) {
            v += __hip_atomic_load(&replicas[r * 32 + field], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __hip_atomic_store(&replicas[r * 32 + field], 0ll, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    part[group * 32 + field] = v;
    __syncthreads();
    if (threadIdx.x < 64) {
        __shared__ unsigned long long pay[30];
        long long tot = 0;
        if (threadIdx.x < 29)
            for (int r = 0; r < 8; r++) tot += part[r * 32 + threadIdx.x];
        if (P2P) {
            if (threadIdx.x < 29) sums[32 + threadIdx.x] = tot;        // (this shard's own record, for tools/p2p_first_frame_stress.py)
            if (!p2p_icp_exchange(pv, p2p_seq, tot, pay)) return;
        }      // (no record: the host reports the missing peer)
        if (threadIdx.x < 29) {
            sums[threadIdx.x] = tot;
            pay[threadIdx.x] = (unsigned long long)tot;
        }
        const unsigned long long check = (unsigned long long)wsum64(tot) + seq;
        if (threadIdx.x == 0) pay[29] = check;
        // one wave: LDS operations of a wave execute in order, so the reads below see the writes above; 40 lanes
        // store the five self-validating lines in one instruction
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the LDS writes have landed
        if (threadIdx.x < 40)
            __hip_atomic_store(&mb->icp_rec[threadIdx.x], SSF_ICP_REC_WORD(threadIdx.x, pay, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- the counted record: the end of an accumulating launch in ONE trip instead of three ------------------------------------
// icp_fold / grid_arrive / icp_publish above cost three DEPENDENT trips to the coherence point after the last row: the
// returning adds into a replica (their return is the proof they were performed), the arrival ticket (two levels), the last
// workgroup's read of the replicas -- 4-5 us of an 8 us launch that runs three times per frame on the critical chain.
// Here every word proves its own completeness: a workgroup adds (sum << 10) + 1 to each of the 30 words of its replica --
// fire and forget, no return, no ticket --, so the low ten bits of a word count the workgroups that have contributed to it
// (at most 512 per replica: 4096 workgroups over 8 replicas) and the rest is the exact sum (terms are 32-bit fixed point,
// at most 2^17 rows per replica: < 2^48; term 27, a 64-bit fixed-point sum, travels as its low 40 bits and the rest in word
// 29).  A fixed workgroup -- the last of the grid: it starts last -- collects: 240 threads EXCHANGE one word each with zero
// (a returning read-modify-write: what it returns has been performed, and it leaves the record clean for the next launch
// whatever lands when), add what they got to what they had, and repeat until every word's count says that all workgroups of
// its replica are in; then the record is decoded, summed over the replicas and published exactly like icp_publish's.
// Exact integers throughout: the published record is the old one bit for bit.  The wait is bounded by wall-clock time
// (SSF_ICP_COLLECT_WAIT_TICKS).
#define ICP_CNT_BITS 10
#define ICP_CNT_WORDS 30
#ifndef SSF_ICP_COLLECT_WAIT_TICKS
#define SSF_ICP_COLLECT_WAIT_TICKS 400000000ull    // 4 s of the 100 MHz wall clock: below the host's own 5 s wait for the record, far above
                                                   // anything a part shared with other processes delays a workgroup by
#endif
__device__ __forceinline__ void icp_fold_counted(const unsigned long long* red, unsigned long long* __restrict__ counted) {
    if (threadIdx.x < ICP_CNT_WORDS) {
        const int k = threadIdx.x == 29 ? 27 : (int)threadIdx.x;
        unsigned long long tot = 0;
#pragma unroll
        for (int sidx = 0; sidx < ICP_SLOTS; sidx++) tot += red[k * ICP_SLOTS + sidx];
        long long v = (long long)tot;
        if (threadIdx.x == 27) v = (long long)(tot & ((1ull << 40) - 1ull));            // term 27: low 40 bits here ...
        if (threadIdx.x == 29) v = (long long)tot >> 40;                                // ... the rest (signed) in word 29
        unsigned long long* rep = counted + (size_t)(blockIdx.x % SSF_ICP_REPLICAS) * 32;
        (void)__hip_atomic_fetch_add(&rep[threadIdx.x], ((unsigned long long)v << ICP_CNT_BITS) + 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}
__device__ __forceinline__ void icp_collect_counted(unsigned long long* __restrict__ counted, long long* __restrict__ sums, Mailbox* mb,
                                                    unsigned long long seq, unsigned int n_wg) {
    __shared__ long long part[SSF_ICP_REPLICAS * 32];
    __shared__ int s_expired;
    // word w = 256 j + thread of the SSF_ICP_REPLICAS x 30 counted words: replica w / 30, word w % 30
    constexpr int NW = SSF_ICP_REPLICAS * ICP_CNT_WORDS, WPT = (NW + 255) / 256;
    constexpr unsigned long long CNT_MASK = (1ull << ICP_CNT_BITS) - 1ull;
    int wr[WPT], wk[WPT]; bool mine[WPT]; unsigned int expect[WPT]; unsigned long long acc[WPT];
#pragma unroll
    for (int j = 0; j < WPT; j++) {
        const int w = 256 * j + (int)threadIdx.x;
        wr[j] = w / ICP_CNT_WORDS; wk[j] = w - wr[j] * ICP_CNT_WORDS;
        mine[j] = w < NW;
        // workgroups b with b % SSF_ICP_REPLICAS == replica
        expect[j] = mine[j] ? (n_wg + SSF_ICP_REPLICAS - 1u - (unsigned int)wr[j]) / SSF_ICP_REPLICAS : 0u;
        acc[j] = 0ull;
    }
    if (threadIdx.x == 0) s_expired = 0;
    __syncthreads();
    const unsigned long long t0 = wall_clock64();
    for (unsigned int round = 0;; round++) {
        bool done = true;
        unsigned long long got[WPT];
#pragma unroll
        for (int j = 0; j < WPT; j++) {           // (the exchanges of a round are independent: one trip)
            const bool need = mine[j] && (unsigned int)(acc[j] & CNT_MASK) != expect[j];
            got[j] = need ? __hip_atomic_exchange(&counted[wr[j] * 32 + wk[j]], 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0ull;
        }
#pragma unroll
        for (int j = 0; j < WPT; j++) {
            acc[j] += got[j];
            done = done && (!mine[j] || (unsigned int)(acc[j] & CNT_MASK) == expect[j]);
        }
        if (__syncthreads_and(done ? 1 : 0)) break;
        if (threadIdx.x == 0 && (round & 15u) == 15u && wall_clock64() - t0 > SSF_ICP_COLLECT_WAIT_TICKS) s_expired = 1;
        __syncthreads();
        if (s_expired) return;                   // (no record: the host's own bounded wait reports it and puts the buffers back to rest)
        __builtin_amdgcn_s_sleep(1);
    }
#pragma unroll
    for (int j = 0; j < WPT; j++)
        if (mine[j]) part[wr[j] * 32 + wk[j]] = (long long)(acc[j] - (acc[j] & CNT_MASK)) >> ICP_CNT_BITS;
    __syncthreads();
    if (threadIdx.x < 64) {
        __shared__ unsigned long long pay[30];
        long long tot = 0;
        if (threadIdx.x < ICP_CNT_WORDS)
            for (int q = 0; q < SSF_ICP_REPLICAS; q++) tot += part[q * 32 + threadIdx.x];
        // term 27 = (word 29 << 40) + word 27
        const long long hi27 = (long long)shfl_u64((unsigned long long)tot, 29);
        if (threadIdx.x == 27) tot = (long long)(((unsigned long long)hi27 << 40) + (unsigned long long)tot);
        if (threadIdx.x >= 29) tot = 0;
        if (threadIdx.x < 29) {
            sums[threadIdx.x] = tot;
            pay[threadIdx.x] = (unsigned long long)tot;
        }
        const unsigned long long check = (unsigned long long)wsum64(tot) + seq;
        if (threadIdx.x == 0) pay[29] = check;
        __builtin_amdgcn_s_waitcnt(0xc07f);       // lgkmcnt(0): the LDS writes have landed
        if (threadIdx.x < 40)
            __hip_atomic_store(&mb->icp_rec[threadIdx.x], SSF_ICP_REC_WORD(threadIdx.x, pay, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
}
// ---- rows in image-tile order (ssf_tile_rows.inc, further down): helpers of the launches that stream the tile-sorted copy ------------
// Rows sorted by image tile (k_bin_* below) are handed to the launch's workgroups so that ONE XCD works on one contiguous
// eighth of them -- an eighth of the image: its L2 then holds an eighth of the frame's (label, depth) table instead of
// all of it.  Workgroup b runs on XCD b % 8 (observed placement; only speed depends on it): logical block =
// the (b / 8)-th block of that XCD's share.  Bijective for any grid size.
__device__ __forceinline__ unsigned int xcd_block(unsigned int b, unsigned int nb) {
    const unsigned int q = nb >> 3, r = nb & 7u, x = b & 7u;
    return (x < r ? x * (q + 1u) : r * (q + 1u) + (x - r) * q) + (b >> 3);
}
// ---- association of tile-sorted rows (ssf_tile_rows.inc; round 4, in the product since round 6) -------------------------------------------------------------
// Sorted rows pile their atomicMin onto the same words (k_match 55 -> 88 us at BASELINE config 3).  Here the lanes of a wave that bid for the same frame supersurfel agree on their minimum
// first -- one global atomicMin per distinct frame supersurfel and wave.  Called by the lanes that have a bid (any subset of the
// wave: ballots and cross-lane reads only see active lanes' values for active lanes).
__device__ __forceinline__ int match_bid_wave(int f, unsigned long long key, unsigned long long* __restrict__ best) {
    unsigned long long todo = __ballot(1);                       // the lanes with a bid (the others have left match_row)
    const unsigned int klo = (unsigned int)key, khi = (unsigned int)(key >> 32);
    while (todo) {
        const int leader = __ffsll((long long)todo) - 1;
        const int f0 = __builtin_amdgcn_readlane(f, leader);
        const unsigned long long same = __ballot(f == f0) & todo;
        // minimum key of the group, on the scalar unit: one pair of lane reads per member (uniform loop: `same` is a ballot)
        unsigned long long m = ~0ull, rest = same;
        while (rest) {
            const int o = __ffsll((long long)rest) - 1;
            const unsigned long long ko = ((unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)khi, o) << 32) |
                                          (unsigned long long)(unsigned int)__builtin_amdgcn_readlane((int)klo, o);
            m = ko < m ? ko : m;
            rest &= rest - 1ull;
        }
        if (lane() == leader) atomicMin(&best[f0], m);
        todo &= ~same;
    }
    return f;
}
#ifdef SSF_EXPERIMENTS
#include "lab/icp_arms.inc"
#endif
__device__ __forceinline__ int match_row(const Cam& cam, const SurfelSoA& model, int j, int id, const uint2* __restrict__ pix2,
                                         const float4* __restrict__ fpack, const Rt& pose, float zmin, float zmax,
                                         long long id_offset, unsigned long long* __restrict__ best, uint8_t* __restrict__ matched,
                                         bool wave_agg = false);
__device__ __forceinline__ int match_values(const Cam& cam, float m_conf, const V3& mp, const V3& m_r2, const V3& m_lab, int id,
                                            const uint2* __restrict__ pix2, const float4* __restrict__ fpack, const Rt& pose, float zmin,
                                            float zmax, long long id_offset, unsigned long long* __restrict__ best,
                                            uint8_t* __restrict__ matched, bool wave_agg);
// the association over the tile-sorted copy's 48-byte records, workgroups dealt to the XCDs in contiguous shares
__device__ __forceinline__ void match_sorted_rows(const Cam& cam, const SurfelSoA& sorted, int n_visible, const uint2* __restrict__ pix2,
                                                  const float4* __restrict__ fpack, const Rt& pose, const MatchArgs& ma) {
    for (int j = (int)(xcd_block(blockIdx.x, gridDim.x) * blockDim.x + threadIdx.x); j < n_visible; j += gridDim.x * blockDim.x) {
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(sorted.pos) + 3 * (size_t)j;
        float4 a = rec[0], b = rec[1], c = rec[2];
        asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x));
        const int oid = __float_as_int(b.w);
        ma.cand[oid] = match_values(cam, a.w, v3(a.x, a.y, a.z), v3(c.x, c.y, c.z), v3(b.x, b.y, b.z), oid, pix2, fpack, pose, ma.zmin, ma.zmax, ma.id_offset, ma.best, ma.matched, true);
    }
}
#define SSF_ICP_DBG_COUNTED 0x40000000          // bit of k_icp's `dbg` argument: end the launch with the counted record
#ifndef SSF_ICP_GO_WAIT_TICKS
#define SSF_ICP_GO_WAIT_TICKS 25000000ull     // 0.25 s of the 100 MHz wall clock: how long a launch made ahead waits for the host's word
#endif
// Scalar registers decide how many 256-thread workgroups a compute unit ADMITS: min(8, floor(800 / (ceil(sgpr / 16) * 16 + 16))) --
// <= 80: 8, 82-96: 7, >= 98: 6 (MI355X_MICROARCH.md, "Residency"; the occupancy the compiler prints does not know).  Left alone this
// kernel takes 88 (seven per compute unit); round 4 saw versions at 98-102 run the whole chain 4-6 % slower "for reasons not
// understood" -- this is the reason.  Capped at 80 (78 used, 61 vector registers, no scratch): same-box A/B in profiles/track_chain_r05.txt.
#ifndef SSF_ICP_NUM_SGPR
#define SSF_ICP_NUM_SGPR 80
#endif
template <bool P2P, int MODE>          // MODE 0: rows from the visible array; 3: rows from the tile-sorted copy (by_tile launches, large visible sets); lab: 1 = rows' terms summed in registers (SSF_ICP_PER_LANE), 2 = DPP row reduction (SSF_ICP_WRED)
__global__ __launch_bounds__(256) __attribute__((amdgpu_num_sgpr(SSF_ICP_NUM_SGPR))) void k_icp(Cam cam, SurfelSoA model, int n_visible,
                                             const uint2* __restrict__ pix2, const float4* __restrict__ fpack,
                                             Rt T, long long* __restrict__ replicas, unsigned int* ticket,
                                             long long* __restrict__ sums, Mailbox* mb, unsigned long long seq, int dbg,
                                             IcpGo* go, unsigned long long go_seq, P2PView pv, int by_tile, MatchArgs ma) {
    __builtin_amdgcn_s_setprio(3);            // the track chain is the critical path: its waves issue ahead of the extract waves sharing a SIMD
    // Scalar arguments used only behind the wait for the host's word, fetched NOW: left alone the compiler re-loads them from the
    // argument segment where they are used (it is short of scalar registers) -- `s_load_dword` + `s_waitcnt lgkmcnt(0)` right behind
    // the barrier that ends the wait, a dependent trip at the head of every chained iteration (read off the ISA, round 4)
    // (gridDim.x / blockDim.x come from the implicit arguments the same way: the workgroup is 256 threads by construction)
    unsigned int n_wg = gridDim.x;
    constexpr unsigned int WG = 256;
    asm volatile("" : "+s"(dbg), "+s"(by_tile), "+s"(n_visible), "+s"(n_wg));
    __shared__ unsigned long long red[29 * ICP_SLOTS];
    __shared__ float s_T[12];
    __shared__ int s_go;
    __shared__ unsigned long long s_p2p_seq;
    for (int i = threadIdx.x; i < 29 * ICP_SLOTS; i += WG) red[i] = 0ull;
    if (go) {
        // launched ahead of its transform: wait for the host's word (bounded: a lost word must not hang the device)
        // Wave 0 polls: sixteen lanes fetch the slot's whole 64-byte line in ONE instruction -- flag word, transform and the number
        // of the peer exchange arrive together (IcpGo, ssf_device.hpp), validated by the checksum in the flag word.  (History:
        // the polling lane read the transform word by word -- twelve dependent trips to fine-grained memory at the head of every
        // chained iteration --, then twelve lanes fetched it in one trip behind the flag; now it needs no trip of its own.)
        if (threadIdx.x < 64) {
            int ok = 0, told = 0;
            const unsigned int l = threadIdx.x, want = (unsigned int)go_seq;
            const unsigned int* gw = reinterpret_cast<const unsigned int*>(go);
            const unsigned int weight = icp_go_word_weight(l);
            unsigned int w = 0u;
            // (the bound is WALL-CLOCK time -- the constant-rate counter behind wall_clock64(), 100 MHz on this part --, looked
            // at every 64 polls: a quarter of a second, not a spin count whose length in seconds depends on how long a poll
            // of fine-grained memory takes under load.  The host's own round trip is 3-5 us.)
            const unsigned long long t0 = wall_clock64();
            for (unsigned int spin = 0; spin < (1u << 24); spin++) {
                w = l < 16u ? __hip_atomic_load(&gw[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0u;
                const unsigned int flo = (unsigned int)__builtin_amdgcn_readlane((int)w, 12), fhi = (unsigned int)__builtin_amdgcn_readlane((int)w, 13);
                if (flo == want) {
                    if (fhi & 0x80000000u) {                                                     // SSF_ICP_GO_ABORT, valid only with this go_seq's hash beside it
                        if ((fhi & SSF_ICP_GO_CHECK_MASK) == icp_go_abort_check(want)) { told = 1; break; }      // (else: a stale high half -- keep polling)
                    } else {
                        unsigned int sum = w * weight;                                           // (lanes >= 16: 0)
#pragma unroll
                        for (int o = 1; o < 16; o <<= 1) sum += __shfl_xor(sum, o, 64);
                        sum = (unsigned int)__builtin_amdgcn_readfirstlane((int)sum);
                        if (((sum >> 2) & SSF_ICP_GO_CHECK_MASK) == (fhi & SSF_ICP_GO_CHECK_MASK)) {
                            ok = (fhi & 0x40000000u) ? 2 : 1; told = 1;                          // SSF_ICP_GO_MATCH: the loop is over, associate under the pose in the line
                            break;
                        }
                    }
                }
                if ((spin & 63u) == 63u && wall_clock64() - t0 > SSF_ICP_GO_WAIT_TICKS) break;
                __builtin_amdgcn_s_sleep(1);
            }
            // gave up waiting (the host stalled for seconds): make that the decision of the whole launch -- workgroups
            // dispatched later must not find a word that arrives after all and start accumulating into a record nobody
            // completes.  (Only then: the normal "leave" word is the host's, and hundreds of workgroups echoing it
            // through the BAR cost the launch behind this one 3 us per frame.)
            if (!told && l == 0u) __hip_atomic_store(&go->flag, (unsigned long long)want | ((unsigned long long)icp_go_abort_check(want) << 32) | SSF_ICP_GO_ABORT, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            if (ok) {
                if (l < 12u) s_T[l] = __uint_as_float(w);
                // (a launch made ahead learns the number of its peer exchange with its transform: a dismissed launch must
                // not use one up, the two slot parities of the exchange regions rely on consecutive numbers)
                const unsigned int xlo = (unsigned int)__builtin_amdgcn_readlane((int)w, 14), xhi = (unsigned int)__builtin_amdgcn_readlane((int)w, 15);
                if (P2P && l == 0u) s_p2p_seq = ((unsigned long long)xhi << 32) | xlo;
            }
            if (l == 0u) s_go = ok;
        }
        __syncthreads();
        if (!s_go) return;
        T.R = m3(v3(s_T[0], s_T[1], s_T[2]), v3(s_T[3], s_T[4], s_T[5]), v3(s_T[6], s_T[7], s_T[8]));
        T.t = v3(s_T[9], s_T[10], s_T[11]);
        if constexpr (!P2P && (MODE == 0 || MODE == 3)) {
            if (s_go == 2) {
                // findBestMatches in the launch that was waiting for the next iteration (k_match's rows, k_match's arithmetic)
                // (The pose is uniform, and said so HERE: read from LDS it was twelve VECTOR registers live through this branch --
                // 72 registers for the kernel, seven waves per SIMD, where the iteration needs 46.  Not for the iteration itself:
                // with its transform in scalar registers the kernel was 4 % SLOWER on the whole chain, same-box A/B round 4 --
                // 11 077 against 11 494 frames/s, config 3 2105 against 2188.)
                auto sT = [&](int i) { return __uint_as_float(__builtin_amdgcn_readfirstlane(__float_as_uint(s_T[i]))); };
                T.R = m3(v3(sT(0), sT(1), sT(2)), v3(sT(3), sT(4), sT(5)), v3(sT(6), sT(7), sT(8)));
                T.t = v3(sT(9), sT(10), sT(11));
                if constexpr (MODE == 3) { if (ma.best) match_sorted_rows(cam, model, n_visible, pix2, fpack, T, ma); return; }
                if (ma.best && !by_tile)
                    for (int id = blockIdx.x * WG + threadIdx.x; id < n_visible; id += n_wg * WG)
                        ma.cand[id] = match_row(cam, model, id, id, pix2, fpack, T, ma.zmin, ma.zmax, ma.id_offset, ma.best, ma.matched);
                return;
            }
        }
    }
    const unsigned long long p2p_seq = (P2P && go) ? s_p2p_seq : pv.seq;
    __syncthreads();
    const M3 R = T.R; const V3 t = T.t;
    const int slot = lane() & (ICP_SLOTS - 1);
    // (Measured and dropped, round 2: a launch made ahead fetching its rows BEFORE it waits for the host's word -- they do not
    // depend on the transform.  8806-8834 against 8757-8835 frames/s, nothing: the rows are L2 hits left by the previous
    // iteration, not a trip to HBM.)
    // (by_tile: `model` is the tile-sorted copy of the visible rows -- pos / lab / r2 streams only -- and the blocks are dealt
    // to the XCDs in contiguous shares; the sums are exact integers, so the order of the rows does not matter)
    unsigned int blk = blockIdx.x; (void)by_tile;
    if constexpr (MODE == 3) {
        // rows from the tile-sorted copy (ssf_tile_rows.inc): 48-byte records, (position, confidence) (Lab, index) (normal, 0);
        // one XCD works on one contiguous eighth of them -- an eighth of the image
        blk = xcd_block(blockIdx.x, n_wg);
        const float4* __restrict__ rec = reinterpret_cast<const float4*>(model.pos);
        for (int id = blk * WG + threadIdx.x; id < n_visible; id += n_wg * WG) {
            float4 a = rec[3 * id], b = rec[3 * id + 1], c = rec[3 * id + 2];
            asm volatile("" : "+v"(a.x), "+v"(b.x), "+v"(c.x));
            icp_row(cam, pix2, fpack, R, t, v3(a.x, a.y, a.z), v3(b.x, b.y, b.z), v3(c.x, c.y, c.z), red, slot, 0);
        }
    } else
#ifdef SSF_EXPERIMENTS
    if (MODE != 0) icp_lab_arm<MODE>(cam, model, n_visible, pix2, fpack, R, t, red, slot, blk);
    else
#endif
        for (int id = blk * WG + threadIdx.x; id < n_visible; id += n_wg * WG) {
            // the row's three fields in ONE round trip (left alone, the compiler fetches colour and normal only behind the
            // test of the projected position: a second dependent trip in every iteration)
            V3 mpos = ld3(model.pos, id), mlab = ld3(model.lab, id), mnrm = ld3(model.r2, id);
            asm volatile("" : "+v"(mpos.x), "+v"(mlab.x), "+v"(mnrm.x));
            icp_row(cam, pix2, fpack, R, t, mpos, mlab, mnrm, red, slot, dbg);
        }
    __syncthreads();
    __shared__ int s_last;
    if (SSF_PROBE(dbg, 4)) {                    // probe: fold only
        icp_fold(red, replicas);
        return;
    }
    if constexpr (!P2P) {
        if (dbg & SSF_ICP_DBG_COUNTED) {            // the counted record (second half of the replica buffer): see icp_fold_counted
            unsigned long long* counted = reinterpret_cast<unsigned long long*>(replicas) + SSF_ICP_REPLICAS * 32;
            icp_fold_counted(red, counted);
            if (blockIdx.x == n_wg - 1) icp_collect_counted(counted, sums, mb, seq, n_wg);
            return;
        }
    }
    icp_fold(red, replicas);
    if (threadIdx.x == 0) s_last = grid_arrive(ticket);
    __syncthreads();
    if (s_last) icp_publish<P2P>(replicas, sums, mb, seq, pv, p2p_seq);
}

// ---- loop-closure registration (DenseRegistration::align) -----------------------------------------------------
// makeCorrespondences, dense_registration_kernels.cu:27-100, for one source supersurfel
__device__ __forceinline__ bool align_pair(const Cam& cam, const float* __restrict__ spos, const float* __restrict__ slab,
                                           const float* __restrict__ snrm, const float* __restrict__ sconf, int id,
                                           const SurfelSoA& frame, const int32_t* __restrict__ label,
                                           const float* __restrict__ plane_depth, const M3& R, const V3& t,
                                           V3& pv, V3& sn, V3& tp, V3& tn) {
    if (sconf && !(sconf[id] > 0.0f)) return false;
    pv = add(m3_mulv(R, ld3(spos, id)), t);
    const int u = pixel_round(pv.x * cam.fx / pv.z + cam.cx), v = pixel_round(pv.y * cam.fy / pv.z + cam.cy);
    if (!(u >= 0 && u < cam.W && v >= 0 && v < cam.H)) return false;
    const size_t q = (size_t)v * cam.W + u;
    const int tid = label[q];
    if (!(frame.conf[tid] > 0.0f)) return false;
    const float dist_color = len3(sub(ld3(slab, id), ld3(frame.lab, tid)));
    const float td = plane_depth[q];
    if (!isfinite(td)) return false;
    sn = unit3(ld3(snrm, id));
    sn = unit3(m3_mulv(R, sn));
    tn = unit3(ld3(frame.r2, tid));
    tp = v3(td * ((float)u - cam.cx) / cam.fx, td * ((float)v - cam.cy) / cam.fy, td);
    return dist_color < 20.0f && len3(sub(pv, tp)) < 0.1f && fabsf(dot3(sn, tn)) > 0.8f;
}
// One iteration of align in ONE single-workgroup launch: pair count + centroids, scale, normalised symmetric
// point-to-plane system (buildSymmetricPoint2PlaneSystem, dense_registration_kernels.cuh:87-173).  The reference
// compacts the valid pairs (thrust::remove_if) to feed three reductions; all sums here are exact integers, so the
// pairs are simply re-derived in each phase (a keyframe has ~S sources: the whole problem is LDS-sized).
// out: [0..28] record as k_icp, [29] pairs, [30..32] source centroid bits, [33..35] target centroid bits, [36] scale bits
__global__ __launch_bounds__(1024) void k_align(Cam cam, const float* __restrict__ spos, const float* __restrict__ slab,
                                                const float* __restrict__ snrm, const float* __restrict__ sconf, int n,
                                                SurfelSoA frame, const int32_t* __restrict__ label,
                                                const float* __restrict__ plane_depth, Rt T, long long* __restrict__ out) {
    __shared__ unsigned long long acc[8];
    __shared__ unsigned long long red[29 * 16];
    __shared__ float s_c[7];
    __shared__ int s_pairs;
    for (int i = threadIdx.x; i < 29 * 16; i += blockDim.x) red[i] = 0ull;
    if (threadIdx.x < 8) acc[threadIdx.x] = 0ull;
    __syncthreads();
    const M3 R = T.R; const V3 t = T.t;
    V3 pv, sn, tp, tn;
    for (int id = threadIdx.x; id < n; id += blockDim.x) {
        if (!align_pair(cam, spos, slab, snrm, sconf, id, frame, label, plane_depth, R, t, pv, sn, tp, tn)) continue;
        atomicAdd(&acc[0], (unsigned long long)fx64((double)pv.x, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[1], (unsigned long long)fx64((double)pv.y, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[2], (unsigned long long)fx64((double)pv.z, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[3], (unsigned long long)fx64((double)tp.x, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[4], (unsigned long long)fx64((double)tp.y, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[5], (unsigned long long)fx64((double)tp.z, SSF_ALIGN_SCALE_POS, SSF_ALIGN_LIM));
        atomicAdd(&acc[6], 1ull);
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int pairs = (int)acc[6];
        s_pairs = pairs;
        for (int i = 0; i < 6; i++) s_c[i] = (float)((double)(long long)acc[i] / SSF_ALIGN_SCALE_POS) / (float)pairs;
        out[29] = pairs;
    }
    __syncthreads();
    const int pairs = s_pairs;
    if (pairs < 100) {                                       // the host stops here (dense_registration.cu:133-138)
        if (threadIdx.x < 29) out[threadIdx.x] = 0;
        if (threadIdx.x >= 30 && threadIdx.x < 37) out[threadIdx.x] = 0;
        return;
    }
    const V3 cs = v3(s_c[0], s_c[1], s_c[2]), ct = v3(s_c[3], s_c[4], s_c[5]);
    for (int id = threadIdx.x; id < n; id += blockDim.x) {
        if (!align_pair(cam, spos, slab, snrm, sconf, id, frame, label, plane_depth, R, t, pv, sn, tp, tn)) continue;
        const V3 a = v3(tp.x - ct.x, tp.y - ct.y, tp.z - ct.z), b = v3(pv.x - cs.x, pv.y - cs.y, pv.z - cs.z);
        atomicAdd(&acc[7], (unsigned long long)fx64((double)((a.x * a.x + a.y * a.y) + a.z * a.z), SSF_ALIGN_SCALE_D2, SSF_ALIGN_LIM));
        atomicAdd(&acc[7], (unsigned long long)fx64((double)((b.x * b.x + b.y * b.y) + b.z * b.z), SSF_ALIGN_SCALE_D2, SSF_ALIGN_LIM));
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        float sv = (float)((double)(long long)acc[7] / SSF_ALIGN_SCALE_D2);
        sv = sqrtf(sv / (2.0f * (float)pairs));
        s_c[6] = 1.0f / sv;
    }
    __syncthreads();
    const float sc = s_c[6];
    const int slot = lane() & 15;
    for (int id = threadIdx.x; id < n; id += blockDim.x) {
        if (!align_pair(cam, spos, slab, snrm, sconf, id, frame, label, plane_depth, R, t, pv, sn, tp, tn)) continue;
        const V3 ps = scale(sc, sub(pv, cs)), pt = scale(sc, sub(tp, ct));
        const V3 ns = unit3(sn), nt = unit3(tn);
        const V3 d = sub(pt, ps), c1 = cross3(pt, ns), c2 = cross3(ps, nt);
        const float dn1 = dot3(d, ns), dn2 = dot3(d, nt);
        const float x1[6] = {c1.x, c1.y, c1.z, ns.x, ns.y, ns.z};
        const float x2[6] = {c2.x, c2.y, c2.z, nt.x, nt.y, nt.z};
        int k = 0;
#pragma unroll
        for (int i = 0; i < 6; i++)
#pragma unroll
            for (int j = i; j < 6; j++, k++)
                atomicAdd(&red[k * 16 + slot], (unsigned long long)(long long)fx32(x1[i] * x1[j] + x2[i] * x2[j], 1048576.0f));
#pragma unroll
        for (int i = 0; i < 6; i++)
            atomicAdd(&red[(21 + i) * 16 + slot], (unsigned long long)(long long)fx32(dn1 * x1[i] + dn2 * x2[i], 16777216.0f));
        atomicAdd(&red[27 * 16 + slot], (unsigned long long)fx64((double)(dn2 * dn2), 17592186044416.0, 4611686018427387904.0));
        atomicAdd(&red[28 * 16 + slot], 1ull);
    }
    __syncthreads();
    if (threadIdx.x < 29) {
        unsigned long long tot = 0;
        for (int sidx = 0; sidx < 16; sidx++) tot += red[threadIdx.x * 16 + sidx];
        out[threadIdx.x] = (long long)tot;
    }
    if (threadIdx.x < 7) out[30 + threadIdx.x] = (long long)__float_as_uint(s_c[threadIdx.x]);
}

// computeCodes_kernel, ferns_kernels.cu:48-70 (point sampling, clamp: texture_impl.hpp:43-46)
__global__ void k_fern_codes(const uint8_t* __restrict__ rgb, const float* __restrict__ depth, int W, int H,
                             const uint32_t* __restrict__ fpos, const uint8_t* __restrict__ frgb,
                             const float* __restrict__ fdepth, int n, uint8_t* __restrict__ codes) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= n) return;
    const int x = (int)min(fpos[2 * k], (uint32_t)W - 1u), y = (int)min(fpos[2 * k + 1], (uint32_t)H - 1u);
    const size_t q = (size_t)y * W + x;
    uint8_t r = 0;
    r |= rgb[3 * q] > frgb[3 * k] ? 1 : 0;
    r |= rgb[3 * q + 1] > frgb[3 * k + 1] ? 2 : 0;
    r |= rgb[3 * q + 2] > frgb[3 * k + 2] ? 4 : 0;
    r |= depth[q] > fdepth[k] ? 8 : 0;
    codes[k] = r;
}

// ---- association ---------------------------------------------------------------------------------
// one frame supersurfel (or none) per visible model row; cand[id] = the frame supersurfel this row has bid for
// (-1: none) -- the fuse launch uses it to tell which rows the update is about to rewrite
// (j: row of the view the kernel streams -- the visible array itself, or its tile-sorted copy --; id: the row's index in
// the visible array, which is what the association key carries)
__device__ __forceinline__ int match_row(const Cam& cam, const SurfelSoA& model, int j, int id, const uint2* __restrict__ pix2,
                                         const float4* __restrict__ fpack, const Rt& pose, float zmin, float zmax,
                                         long long id_offset, unsigned long long* __restrict__ best, uint8_t* __restrict__ matched,
                                         bool wave_agg) {
    // everything this row contributes is requested at once (a visible row nearly always gets to the end): the chain is
    // row -> pixel -> frame supersurfel -> atomic, three dependent round trips instead of five
    float m_conf = model.conf[j];
    V3 mp = ld3(model.pos, j), m_r2 = ld3(model.r2, j), m_lab = ld3(model.lab, j);
    asm volatile("" : "+v"(m_conf), "+v"(mp.x), "+v"(m_r2.x), "+v"(m_lab.x));
    return match_values(cam, m_conf, mp, m_r2, m_lab, id, pix2, fpack, pose, zmin, zmax, id_offset, best, matched, wave_agg);
}
__device__ __forceinline__ int match_values(const Cam& cam, float m_conf, const V3& mp, const V3& m_r2, const V3& m_lab, int id,
                                            const uint2* __restrict__ pix2, const float4* __restrict__ fpack, const Rt& pose, float zmin,
                                            float zmax, long long id_offset, unsigned long long* __restrict__ best,
                                            uint8_t* __restrict__ matched, bool wave_agg) {
    (void)wave_agg;
    if (!(m_conf > 0.0f)) return -1;
    const M3 R = pose.R; const V3 t = pose.t;
    const M3 Rt_ = m3_transpose(R);
    const V3 tview = negate(m3_mulv(Rt_, t));
    const V3 pv = add(m3_mulv(Rt_, mp), tview);
    if (!(pv.z > zmin && pv.z < zmax)) return -1;
    const int px = pixel_round(pv.x * cam.fx / pv.z + cam.cx), py = pixel_round(pv.y * cam.fy / pv.z + cam.cy);
    if (!(px >= 0 && px < cam.W && py >= 0 && py < cam.H)) return -1;
    // (Gathering from the 4-byte label map instead cuts this kernel's fabric traffic from 15.1 to 12.9 MB per launch --
    // measured, round 2 -- but the launch got 0.3 us SLOWER: the ICP iterations that ran just before have left exactly these
    // lines of the (label, depth) table in the L2s, the label map's would be cold.)
    const int f = (int)pix2[(size_t)py * cam.W + px].x;
    matched[f] = 1;
    float4 f0 = fpack[4 * f], f1 = fpack[4 * f + 1], f2 = fpack[4 * f + 2];           // (conf, lab) (normal) (pos): one line
    asm volatile("" : "+v"(f0.y), "+v"(f1.x), "+v"(f2.x));       // (one gather, not confidence first and the rest later: see icp_row)
    if (!(f0.x > 0.0f)) return -1;
    const V3 fp = add(m3_mulv(R, v3(f2.x, f2.y, f2.z)), t);
    const V3 fn = unit3(row_mul(v3(f1.x, f1.y, f1.z), Rt_));    // third row of frame_orientation * R^T
    const V3 mn = unit3(m_r2);
    const float dist = len3(sub(mp, fp));
    const float lab_dist = len3(sub(m_lab, v3(f0.y, f0.z, f0.w)));
    const float delta_norm = fabsf(dot3(mn, fn));
    if (!(lab_dist < 15.0f && delta_norm > 0.8f && dist < 0.05f)) return -1;
    const unsigned long long key = ((unsigned long long)__float_as_uint(dist) << 32) |
                                   (unsigned long long)(uint32_t)(id_offset + id);
    // (Reading the table first to skip hopeless candidates is slower, measured twice: as a trip of its own in front of the atomic,
    // round 2, 72 against 49 us at 860 k rows; fetched past the L1 in the SAME trip as the frame supersurfel's line, round 4, 117
    // against 55 us -- a million coherent reads of 4800 hot words cost more than the atomics they save.  Test-before-set of the
    // `matched` byte: no difference, 55.0 against 55.1 us.  profiles/atomic_scope_r04.txt has the part's atomic rates.)
    // (Rows that are image-coherent by themselves -- a map the pipeline BUILT appends every frame's new supersurfels in ascending
    // frame id -- pile their bids onto the same words: 56 -> 70 us at BASELINE config 3 with the seeded rows in raster order
    // (bench.py --seed-order image; k_icp 23.4 -> 21.0 us, the frame +3 %).  Measured and removed, round 4: aggregating per wave when
    // a quarter of the bidders have their neighbour lane's supersurfel (lab's match_bid_wave behind one cross-lane compare) brings
    // that case back to 57 us, but the ballots push the WAITING launch this function is inlined into from 86 to 102 scalar
    // registers and the whole chain loses 4-6 % on every BASELINE workload (10 968 against 11 647 frames/s, config 3 2102 against
    // 2193, same box): profiles/track_chain_r04b.txt.)
    if (wave_agg) return match_bid_wave(f, key, best);        // (tile-sorted rows: the lanes of a wave that bid for one frame supersurfel agree first; a compile-time `false` everywhere else)
    atomicMin(&best[f], key);
    return f;
}
// (Measured and removed, round 2: trading the association tables with the peers in THIS launch's last workgroup instead of
// in a launch of its own.  The matched flags then have to reach that workgroup through device-scope atomics, and 100 k
// of them on 300-1200 hot addresses cost 50-100 us: memory-side atomics serialise per address at ~0.1 us -- which is also
// what bounds this kernel's own atomicMin.  The separate exchange launch costs ~8 us.)
template <bool SORTED>
__global__ __launch_bounds__(256) void k_match(Cam cam, SurfelSoA model, int n_visible, const uint2* __restrict__ pix2,
                                               const float4* __restrict__ fpack, Rt pose, float zmin, float zmax,
                                               long long id_offset, unsigned long long* __restrict__ best,
                                               uint8_t* __restrict__ matched, int32_t* __restrict__ cand, const int32_t* __restrict__ orig) {
    __builtin_amdgcn_s_setprio(3);            // the track chain is the critical path: its waves issue ahead of the extract waves sharing a SIMD
    // (orig != nullptr: `model` is the tile-sorted copy, orig[j] the row's index in the visible array; blocks dealt to the XCDs
    // in contiguous shares -- see xcd_block)
    if constexpr (SORTED) {                       // (the copy's 48-byte records: ssf_tile_rows.inc)
        (void)orig;
        const MatchArgs ma{zmin, zmax, id_offset, best, matched, cand};
        match_sorted_rows(cam, model, n_visible, pix2, fpack, pose, ma);
        return;
    } else {
        const int j = blockIdx.x * blockDim.x + threadIdx.x, id = j; (void)orig;
        if (j >= n_visible) return;
        cand[id] = match_row(cam, model, j, id, pix2, fpack, pose, zmin, zmax, id_offset, best, matched, false);
    }
}

// ---- image-space order for the model side of ICP / association: a tile-sorted copy of the visible rows (k_bin_*), made for
// large visible sets (BASELINE config 3): ssf_tile_rows.inc
#include "ssf_tile_rows.inc"

// ---- classification of one model row (used by the update/insert launch and by k_classify) --------------------
// filterModel for one row, supersurfel_fusion_kernels.cu:397-467: 0 visible, 1 out of view, 2 removed (conf := -1)
// in two steps, so that a thread that classifies several rows can issue all its plane-depth gathers together:
// classify_pre decides what it can without the frame (st >= 0) or names the pixel to look at (st < 0)
struct ClsPre { int st; int pix; float pz; };
__device__ __forceinline__ ClsPre classify_pre(const Cam& cam, float conf, int last_seen, const V3& pos, const Rt& pose, int stamp,
                                               int delta_t, float conf_thresh, float zmin, float zmax) {
    ClsPre r; r.pix = 0; r.pz = 0.0f;
    const int time_diff = stamp - last_seen;
    if ((time_diff > delta_t && conf < conf_thresh && stamp > delta_t) || conf <= 0.0f) { r.st = 2; return r; }
    const M3 Rv = m3_transpose(pose.R);
    const V3 tv = negate(m3_mulv(Rv, pose.t));
    const V3 p = add(m3_mulv(Rv, pos), tv);
    r.st = 1;
    if (p.z > zmin && p.z < zmax) {
        const float u = cam.fx * p.x / p.z + cam.cx, v = cam.fy * p.y / p.z + cam.cy;
        if (u >= 0.0f && u < (float)cam.W && v >= 0.0f && v < (float)cam.H) {
            r.st = -1; r.pix = ((int)floorf(v)) * cam.W + (int)floorf(u); r.pz = p.z;
        }
    }
    return r;
}
__device__ __forceinline__ int classify_post(const ClsPre& r, float z) { return r.st >= 0 ? r.st : (r.pz < 0.8f * z ? 2 : 0); }
__device__ __forceinline__ int classify_values(const Cam& cam, float conf, int last_seen, const V3& pos, const Rt& pose,
                                               const float* __restrict__ plane_depth, int stamp, int delta_t, float conf_thresh,
                                               float zmin, float zmax) {
    const ClsPre r = classify_pre(cam, conf, last_seen, pos, pose, stamp, delta_t, conf_thresh, zmin, zmax);
    return r.st >= 0 ? r.st : classify_post(r, plane_depth[r.pix]);
}
// the same for a row in memory; a removed row gets conf := -1
__device__ __forceinline__ int classify_row(const Cam& cam, const SurfelSoA& M, size_t idx, const Rt& pose,
                                            const float* __restrict__ plane_depth, int stamp, int delta_t, float conf_thresh,
                                            float zmin, float zmax) {
    const int st = classify_values(cam, M.conf[idx], M.stamps[2 * idx + 1], ld3(M.pos, idx), pose, plane_depth, stamp, delta_t,
                                   conf_thresh, zmin, zmax);
    if (st == 2) M.conf[idx] = -1.0f;
    return st;
}
// (4 since round 5: the launch no longer ends in an arrival count -- the move kernel works the counters out --, so fewer, longer
//  workgroups buy nothing; 8 -> 4 takes the arm alone from 8.6 to 7.2 us and the whole launch from 12.5 to 11.0 at 870 k slots,
//  +1.0 % frames/s in place, alternated three times; 2 is no better: profiles/track_chain_r05.txt item 6)
#ifndef OOV_PER_WG
#define OOV_PER_WG 4
#endif
// 256 slots of the out-of-view span (class B; dead slots are skipped).  These rows are touched by neither the
// update nor the insertion of the frame, so their blocks ride along in the update/insert launch (whose duration is
// set by the long serial chain of the updated rows) instead of lengthening the classify launch.  Of the three
// classes only B0 (rows that come back into view) needs positions: its per-block count goes to bc_oov, and to the
// sum of its group of PART_GROUP blocks; the frame totals of B0 and B2 go to the replicated totals (B1 rows stay
// where they are, their number follows from the live count).  Most blocks have neither and issue no atomic.
__device__ __forceinline__ void classify_oov_block(const Cam& cam, const OovStore& O, const Rt& pose, const float* __restrict__ plane_depth,
                                                   int stamp, int delta_t, float conf_thresh, float zmin, float zmax,
                                                   uint8_t* __restrict__ state_oov, uint32_t* __restrict__ bc_oov, PartitionWs ws,
                                                   const Counters* __restrict__ cnt, int wg, int nb_oov) {
    // OOV_PER_WG blocks of 256 slots per workgroup: their loads are in flight together
    __shared__ int h2[OOV_PER_WG][4][2];
    const int wv = threadIdx.x >> 6;
    const long long head = cnt->oov_head, tail = cnt->oov_tail;
    // Every load of a step for all OOV_PER_WG slots before the next step.  The loads are unconditional (index clamped
    // into the span; a dead or out-of-span slot's row is read and ignored) and pinned together: with `in ? load : 0`
    // the compiler made each slot's loads a branch of its own, one dependent round trip per slot and step.
    int cls[OOV_PER_WG];
    uint8_t lv[OOV_PER_WG]; float conf[OOV_PER_WG]; int seen[OOV_PER_WG]; V3 pos[OOV_PER_WG]; ClsPre pre[OOV_PER_WG]; float z[OOV_PER_WG];
    bool in[OOV_PER_WG];
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) {
        const long long phys = head + ((long long)wg * OOV_PER_WG + j) * 256 /* block size of the fuse launch */ + threadIdx.x;
        in[j] = phys < tail;
        const size_t q = (size_t)(in[j] ? phys : tail - 1);
        lv[j] = O.live[q]; conf[j] = O.rows.conf[q]; seen[j] = O.rows.stamps[2 * q + 1]; pos[j] = ld3(O.rows.pos, q);
    }
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) {
        int lvj = lv[j];
        asm volatile("" : "+v"(lvj), "+v"(conf[j]), "+v"(seen[j]), "+v"(pos[j].x), "+v"(pos[j].y), "+v"(pos[j].z));
        lv[j] = in[j] ? (uint8_t)lvj : (uint8_t)0;
    }
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) {
        pre[j] = classify_pre(cam, conf[j], seen[j], pos[j], pose, stamp, delta_t, conf_thresh, zmin, zmax);
        if (!lv[j]) { pre[j].st = 7; pre[j].pix = 0; }
        if (pre[j].st >= 0) pre[j].pix = 0;
    }
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) z[j] = plane_depth[pre[j].pix];
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) asm volatile("" : "+v"(z[j]));
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) {
        cls[j] = classify_post(pre[j], z[j]);
        if (lv[j]) {
            const long long phys = head + ((long long)wg * OOV_PER_WG + j) * 256 /* block size of the fuse launch */ + threadIdx.x;
            if (cls[j] == 2) O.rows.conf[phys] = -1.0f;
            state_oov[phys] = (uint8_t)cls[j];
        }
    }
#pragma unroll
    for (int j = 0; j < OOV_PER_WG; j++) {
        const int k0 = __popcll(__ballot(cls[j] == 0)), k2 = __popcll(__ballot(cls[j] == 2));
        if (lane() == 0) { h2[j][wv][0] = k0; h2[j][wv][1] = k2; }
    }
    __syncthreads();
    if (threadIdx.x < 2 * OOV_PER_WG) {
        const int j = threadIdx.x >> 1, which = threadIdx.x & 1, ob = wg * OOV_PER_WG + j;
        const uint32_t k = h2[j][0][which] + h2[j][1][which] + h2[j][2][which] + h2[j][3][which];
        if (ob < nb_oov) {
            // bc_oov: rows that come back into view (low half) | removed rows (high half); 0 = nothing for the move
            // kernel to do in these 256 slots
            const uint32_t other = __shfl_xor(k, 1, 64);
            if (which == 0) bc_oov[ob] = k | (other << 16);
            if (k) {
                atomic_add_done(&ws.tot[(ob & (PART_REPLICAS - 1)) * 8 + 6 + which], k);
                if (which == 0) atomicAdd(&ws.sup_oov[ob / PART_GROUP], k);
            }
        }
    }
}
// ---- update ----------------------------------------------------------------------------------------
// updateModel (supersurfel_fusion_kernels.cu:240-346) for UPD_PER_WG frame supersurfels per workgroup.  One thread
// per supersurfel spent 13 us in a serial chain of ~5000 instructions, most of them the double-precision roots of
// the two colour conversions and the twenty normalised squarings of the principal frame.  Those pieces are
// independent, so they are spread over lanes -- identical arithmetic, only who computes what differs:
//   wave 0      two lanes per supersurfel: fused shape and position (both lanes), then lane 0 iterates towards the
//               largest axis, lane 1 towards the smallest; lane 0 stores geometry and classifies the row
//   waves 1, 2  four lanes per supersurfel: one colour channel each (Lab -> sRGB compress, sRGB expand -> Lab f),
//               exchanged with shuffles inside the group; the first lane stores colour and Lab
// (the two kinds of work sit in different waves, so neither waits for the other's instruction stream)
#define UPD_PER_WG 32
struct ClassifyArgs { Cam cam; const float* plane_depth; int delta_t; float conf_thresh, zmin, zmax; };
struct ShardArgs { int rank, nranks, migrate; float tile; };
__device__ __forceinline__ int tile_owner(const V3& pw, int nranks, float tile);
__device__ __forceinline__ float pick3(const V3& v, int k) { return k == 0 ? v.x : (k == 1 ? v.y : v.z); }
__device__ __forceinline__ void update_group(SurfelSoA M, SurfelSoA F, Rt pose, int stamp, long long id_offset, int n_visible,
                                             const unsigned long long* __restrict__ best, const uint8_t* __restrict__ matched,
                                             int S, Counters* cnt, int f0, const ClassifyArgs& ca, uint8_t* __restrict__ state_vis,
                                             PartitionWs ws, const ShardArgs& sh) {
    const int wv = threadIdx.x >> 6, l = lane();
    if (wv == 3) return;
    const int f = wv == 0 ? f0 + (l >> 1) : f0 + (wv - 1) * 16 + (l >> 2);
    if (f >= S) return;                                   // (uniform over a supersurfel's lanes, like every test below)
    // Two dependent round trips instead of four: everything that hangs on the frame supersurfel alone is requested at
    // once (association result and the frame row), then everything that hangs on the model row it won.
    unsigned int mt = matched[f];
    unsigned long long bk = best[f];
    float f_conf = F.conf[f];
    V3 f_pos = v3(0.f, 0.f, 0.f), f_lab = v3(0.f, 0.f, 0.f);
    Sym3 f_shape = sym3(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    if (wv == 0) { f_pos = ld3(F.pos, f); f_shape = ld6(F.shape, f); } else f_lab = ld3(F.lab, f);
    asm volatile("" : "+v"(mt), "+v"(bk), "+v"(f_conf), "+v"(f_pos.x), "+v"(f_shape.xx), "+v"(f_lab.x));
    if (!mt || bk == SSF_NO_MATCH) return;
    const long long local = (long long)(uint32_t)(bk & 0xFFFFFFFFull) - id_offset;
    if (local < 0 || local >= n_visible) return;
    const size_t m = (size_t)local;
    float m_conf = M.conf[m];
    V3 model_position = v3(0.f, 0.f, 0.f), model_lab = v3(0.f, 0.f, 0.f);
    Sym3 model_shape = sym3(0.f, 0.f, 0.f, 0.f, 0.f, 0.f);
    if (wv == 0) { model_position = ld3(M.pos, m); model_shape = ld6(M.shape, m); } else model_lab = ld3(M.lab, m);
    asm volatile("" : "+v"(m_conf), "+v"(model_position.x), "+v"(model_shape.xx), "+v"(model_lab.x));
    const float ratio = 1.0f / (m_conf + f_conf);
    if (wv == 0) {
        const int which = l & 1;
        const M3 R = pose.R; const V3 t = pose.t;
        const V3 frame_position = add(m3_mulv(R, f_pos), t);
        const Sym3 frame_shape = rot_sym(R, f_shape);
        Sym3 f1, m1, fused_shape, fused_1;
        V3 fused_position;
        const float w = ratio * f_conf;
        bool info = false;
        if (sym_inverse(frame_shape, f1) && sym_inverse(model_shape, m1)) {
            fused_1 = sym_add(sym_scale(w, f1), sym_scale(1.0f - w, m1));
            if (sym_inverse(fused_1, fused_shape)) {
                fused_position = sym_mul(fused_shape, add(sym_mul(sym_scale(w, f1), frame_position),
                                                          sym_mul(sym_scale(1.0f - w, m1), model_position)));
                info = true;
            }
        }
        if (!info) {
            fused_shape = sym_scale(ratio, sym_add(sym_scale(f_conf, frame_shape), sym_scale(m_conf, model_shape)));
            fused_position = scale(ratio, add(scale(f_conf, frame_position), scale(m_conf, model_position)));
        }
        const V3 axis = principal_power(principal_start(fused_shape, which == 1));
        const int l0 = l & ~1;
        const V3 r0 = v3(__shfl(axis.x, l0, 64), __shfl(axis.y, l0, 64), __shfl(axis.z, l0, 64));
        const V3 r2 = v3(__shfl(axis.x, l0 + 1, 64), __shfl(axis.y, l0 + 1, 64), __shfl(axis.z, l0 + 1, 64));
        if (which == 0) {
            M3 vecs; V3 vals;
            principal_finish(fused_shape, r0, r2, vecs, vals);
            // classification (from the values in registers: the row is not read back; it was seen in this frame)
            int st = classify_values(ca.cam, m_conf + f_conf, stamp, fused_position, pose, ca.plane_depth, stamp, ca.delta_t,
                                     ca.conf_thresh, ca.zmin, ca.zmax);
            const bool removed = st == 2;
            // sharded map: a row that is kept but whose new position belongs to another rank's tile leaves this shard --
            // the partition drops it (class 2) with its confidence intact; launch_pack_emigrants ships it
            if (sh.migrate && !removed && tile_owner(fused_position, sh.nranks, sh.tile) != sh.rank) {
                st = 2;
                atomic_add_done(&cnt->n_emigrated, 1);
            }
            st3(M.pos, m, fused_position);
            M.conf[m] = removed ? -1.0f : m_conf + f_conf;
            st6(M.shape, m, fused_shape);
            st3(M.r0, m, vecs.r0); st3(M.r1, m, vecs.r1); st3(M.r2, m, vecs.r2);
            M.dims[2 * m] = vals.x; M.dims[2 * m + 1] = vals.y;
            M.stamps[2 * m + 1] = stamp;
            state_vis[m] = (uint8_t)st;
            atomic_add_done(&cnt->n_updated, 1);
            const int vb = (int)(m >> 8);
            atomicAdd(&ws.sup_vis[(vb / PART_GROUP) * 6 + st], 1u);
            atomic_add_done(&ws.tot[(vb & (PART_REPLICAS - 1)) * 8 + st], 1u);
        }
    } else {
        const int ch = min(l & 3, 2), l0 = l & ~3;        // (the fourth lane repeats the third channel; its results are unused)
        const V3 frame_lab = f_lab;
        const V3 lin = lab_to_linear_rgb(scale(ratio, add(scale(f_conf, frame_lab), scale(m_conf, model_lab))));
        const float c255 = srgb_to_255(pick3(lin, ch));
        const V3 fused_color = v3(__shfl(c255, l0, 64), __shfl(c255, l0 + 1, 64), __shfl(c255, l0 + 2, 64));
        const float e = srgb_expand(c255 / 255.0f);
        const V3 q = xyz_from_linear(__shfl(e, l0, 64), __shfl(e, l0 + 1, 64), __shfl(e, l0 + 2, 64));
        const float fq = lab_f(pick3(q, ch));
        const V3 lab = lab_from_f(__shfl(fq, l0, 64), __shfl(fq, l0 + 1, 64), __shfl(fq, l0 + 2, 64));
        if ((l & 3) == 0) { st3(M.col, m, fused_color); st3(M.lab, m, lab); }
    }
}

// spatial-tile owner of a world position / of a frame supersurfel (multi-GPU sharding)
__device__ __forceinline__ int tile_owner(const V3& pw, int nranks, float tile) {
    const int ix = (int)floorf(pw.x / tile), iy = (int)floorf(pw.y / tile), iz = (int)floorf(pw.z / tile);
    const uint32_t h = ((uint32_t)ix * 73856093u) ^ ((uint32_t)iy * 19349663u) ^ ((uint32_t)iz * 83492791u);
    return (int)(h % (uint32_t)nranks);
}
__device__ __forceinline__ int shard_owner(const SurfelSoA& F, int f, const Rt& pose, int nranks, float tile) {
    if (nranks <= 1) return 0;
    if (!(F.conf[f] > 0.0f)) return f % nranks;
    return tile_owner(add(m3_mulv(pose.R, ld3(F.pos, f)), pose.t), nranks, tile);
}

// ordered compaction helper: exclusive rank of `flag` among the block's threads (blockDim <= 1024)
__device__ __forceinline__ int block_rank_1024(bool flag, int* wave_tot, int& block_total) {
    const unsigned long long mask = __ballot(flag);
    const int wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    const int in_wave = __popcll(mask & ((1ull << lane()) - 1ull));
    if (lane() == 0) wave_tot[wv] = __popcll(mask);
    __syncthreads();
    int before = 0, total = 0;
    for (int w = 0; w < nw; w++) { const int c = wave_tot[w]; if (w < wv) before += c; total += c; }
    __syncthreads();
    block_total = total;
    return before + in_wave;
}

// insertSupersurfels (supersurfel_fusion_kernels.cu:348-395) in ascending frame id.  One workgroup per chunk of 256
// frame supersurfels: a chunk needs the number of insertions of the chunks before it, and simply re-evaluates their
// flags (a few cheap tests per thread) instead of waiting for a serial pass -- S = 4800 at 1280x960 was 19 serial
// chunks (57 us) in one workgroup.  cnt->n_visible / n_model are not written by this kernel (the scan adds
// n_inserted to n_model), so every chunk reads the same bases.
__device__ __forceinline__ bool insert_flag(const SurfelSoA& F, int f, int S, const uint8_t* __restrict__ matched, const Rt& pose,
                                            int rank, int nranks, float tile) {
    return f < S && (F.conf[f] > 0.0f) && !matched[f] && shard_owner(F, f, pose, nranks, tile) == rank;
}
__device__ __forceinline__ void insert_chunk(SurfelSoA M, SurfelSoA F, Rt pose, int stamp, const uint8_t* __restrict__ matched,
                                             int S, int capacity, int rank, int nranks, float tile, Counters* cnt, int* wave_tot,
                                             int chunk, int nchunks, const ClassifyArgs& ca, uint8_t* __restrict__ state_vis,
                                             PartitionWs ws) {
    __shared__ int s_before[4];
    __shared__ uint32_t s_cls[2][3];               // states of the rows inserted here, by partition group (a chunk spans at most two)
    if (threadIdx.x < 6) s_cls[threadIdx.x / 3][threadIdx.x % 3] = 0u;
    const int base = cnt->n_visible, base_total = cnt->n_model;
    int mine = 0;
    for (int c = 0; c < chunk; c++) mine += insert_flag(F, c * 256 + (int)threadIdx.x, S, matched, pose, rank, nranks, tile) ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) mine += __shfl_xor(mine, o, 64);
    if (lane() == 0) s_before[threadIdx.x >> 6] = mine;
    __syncthreads();
    const int before = s_before[0] + s_before[1] + s_before[2] + s_before[3];
    const M3 R = pose.R; const V3 t = pose.t;
    const M3 Rt_ = m3_transpose(R);
    const int f = chunk * 256 + threadIdx.x;
    const bool flag = insert_flag(F, f, S, matched, pose, rank, nranks, tile);
    int total;
    const int r = block_rank_1024(flag, wave_tot, total);
    const int k = base + before + r;
    const int g_first = ((base + before) >> 8) / PART_GROUP;        // partition group of the chunk's first row
    if (flag && base_total + before + r < capacity) {
        // every load (the frame row, then the classification's depth lookup) before the first store: the arrays may
        // alias as far as the compiler knows, and a load behind a store waits for it
        const V3 f_pos = ld3(F.pos, f), f_col = ld3(F.col, f), f_lab = ld3(F.lab, f);
        const V3 f_r0 = ld3(F.r0, f), f_r1 = ld3(F.r1, f), f_r2 = ld3(F.r2, f);
        const Sym3 f_shape = ld6(F.shape, f);
        const float f_d0 = F.dims[2 * f], f_d1 = F.dims[2 * f + 1];
        const float new_conf = F.conf[f];
        const V3 new_pos = add(m3_mulv(R, f_pos), t);
        // classification (class C) of the new row, for the partition
        const int st = classify_values(ca.cam, new_conf, stamp, new_pos, pose, ca.plane_depth, stamp, ca.delta_t, ca.conf_thresh,
                                       ca.zmin, ca.zmax);
        st3(M.pos, k, new_pos);
        M.conf[k] = st == 2 ? -1.0f : new_conf;
        st3(M.col, k, f_col);
        st3(M.lab, k, f_lab);
        M.stamps[2 * k] = stamp; M.stamps[2 * k + 1] = stamp;
        M.dims[2 * k] = f_d0; M.dims[2 * k + 1] = f_d1;
        st3(M.r0, k, row_mul(f_r0, Rt_));
        st3(M.r1, k, row_mul(f_r1, Rt_));
        st3(M.r2, k, row_mul(f_r2, Rt_));
        st6(M.shape, k, rot_sym(R, f_shape));
        state_vis[k] = (uint8_t)st;
        atomicAdd(&s_cls[(k >> 8) / PART_GROUP - g_first][st], 1u);
    }
    __syncthreads();
    if (threadIdx.x < 6) {
        const int seg = threadIdx.x / 3, st = threadIdx.x % 3;
        const uint32_t n = s_cls[seg][st];
        if (n) {
            atomicAdd(&ws.sup_vis[(g_first + seg) * 6 + 3 + st], n);
            atomic_add_done(&ws.tot[(chunk & (PART_REPLICAS - 1)) * 8 + 3 + st], n);
        }
    }
    if (chunk == nchunks - 1 && threadIdx.x == 0)
        atomic_store_done(&cnt->n_inserted, min(base_total + before + total, capacity) - base_total);
}
// The fuse launch.  Blocks, in this order:
//   update     nupd blocks of UPD_PER_WG frame supersurfels (updateModel, update_group); also classifies the rows
//              it rewrites
//   insert     nchunks blocks, one chunk each (insertSupersurfels); classifies the rows it inserts
//   classify   nb_vis blocks of 256 old visible rows (filterModel), minus the rows the update rewrites (a row knows
//              from cand / best whether it won its frame supersurfel)
//   classify   the remaining blocks: 256 slots of the out-of-view span each
// Every classification goes to state_vis / state_oov and, as counts, to the sums of the partition (PartitionWs:
// per group of PART_GROUP blocks, and replicated frame totals) with atomics.  The LAST block to finish turns the
// totals into the frame's counters and publishes them; the move kernel that follows derives its prefixes from the
// group sums and the states themselves.  No scan kernel, no separate classify kernel.
__device__ __forceinline__ Counters finalise_counters(Counters* cnt, Counters c, int shrink_by_removed, int keep_inserted = 0);
__device__ __forceinline__ void mailbox_counters(const Counters& c, Mailbox* mb, unsigned long long seq);
// the frame's counters from the counters it started with (+ what the fuse launch accumulated in them) and the eight class
// totals a0 a1 a2 c0 c1 c2 b0 b2 of the partition
__device__ __forceinline__ Counters frame_counters(const Counters& c_in, const uint32_t* tot) {
    const int a0 = (int)tot[0], a1 = (int)tot[1], a2 = (int)tot[2], c0 = (int)tot[3], c1 = (int)tot[4], c2 = (int)tot[5];
    const int b0 = (int)tot[6], b2 = (int)tot[7], b1 = c_in.oov_live - b0 - b2;
    Counters c = c_in;
    c.n_model = c_in.n_model + c_in.n_inserted;   // the insertion reports its rows in n_inserted only
    c.n_state0 = a0 + b0 + c0; c.n_state1 = a1 + b1 + c1; c.n_state2 = a2 + b2 + c2;
    c.n_visible = a0 + b0 + c0; c.n_removed = (a2 + b2 + c2) - c_in.n_emigrated;     // emigrants are dropped, not removed
    c.mv_nv = c_in.n_visible; c.mv_a0 = a0; c.mv_b0 = b0; c.mv_nc = c_in.n_inserted;
    c.mv_head_old = c_in.oov_head; c.mv_tail_old = c_in.oov_tail; c.mv_head_new = c_in.oov_head - a1;
    c.oov_head = c_in.oov_head - a1; c.oov_tail = c_in.oov_tail + c1;
    c.oov_live = (c_in.oov_live - b0 - b2) + a1 + c1;
    return c;
}
#ifdef SSF_EXPERIMENTS
// lab: when set, every workgroup of the fuse launch leaves three ticks of the 100 MHz wall clock -- entry, end of its arm, its
// memory operations acknowledged (tools/fuse_probe.py: which arm is late, and is it late to start or slow to run?)
__device__ unsigned long long* g_fuse_trace = nullptr;
void set_fuse_trace(unsigned long long* p) { (void)hipMemcpyToSymbol(HIP_SYMBOL(g_fuse_trace), &p, sizeof(p)); }
#endif
__global__ __launch_bounds__(256) void k_update_insert(SurfelSoA M, SurfelSoA F, Rt pose, int stamp, long long id_offset,
                                                       int n_visible, const unsigned long long* __restrict__ best,
                                                       const uint8_t* __restrict__ matched, const int32_t* __restrict__ cand, int S,
                                                       int do_update, int capacity, int rank, int nranks, float tile, Counters* cnt,
                                                       int nupd, int nchunks, int nb_vis, int nb_oov, OovStore O, ClassifyArgs ca,
                                                       uint8_t* __restrict__ state_vis, uint8_t* __restrict__ state_oov,
                                                       uint32_t* __restrict__ bc_oov, PartitionWs ws, int migrate, int tail_in_move) {
    __builtin_amdgcn_s_setprio(3);            // the track chain is the critical path: its waves issue ahead of the extract waves sharing a SIMD
    __shared__ int wave_tot[16];
    __shared__ int hist[4][6];
    __shared__ int s_last;
    __shared__ uint32_t tot[8];
    const int b = blockIdx.x;
#ifdef SSF_EXPERIMENTS
    unsigned long long* const trace = g_fuse_trace;
    if (trace && threadIdx.x == 0) trace[3 * (size_t)b] = wall_clock64();
#endif
    if (b >= nupd + nchunks + nb_vis)
        classify_oov_block(ca.cam, O, pose, ca.plane_depth, stamp, ca.delta_t, ca.conf_thresh, ca.zmin, ca.zmax, state_oov, bc_oov,
                           ws, cnt, b - nupd - nchunks - nb_vis, nb_oov);
    else if (b >= nupd + nchunks) {
        const int vb = b - nupd - nchunks, i = vb * blockDim.x + threadIdx.x, wv = threadIdx.x >> 6;
        int cls = 7;
        if (i < n_visible) {
            // the row's own fields travel together with its candidate (then: association result, then the depth lookup)
            int f = cand[i];
            float conf = M.conf[i]; int seen = M.stamps[2 * (size_t)i + 1]; V3 pos = ld3(M.pos, (size_t)i);
            asm volatile("" : "+v"(f), "+v"(conf), "+v"(seen), "+v"(pos.x));
            const bool rewritten = do_update && f >= 0 && (uint32_t)(best[f] & 0xFFFFFFFFull) == (uint32_t)(id_offset + i);
            if (!rewritten) {
                cls = classify_values(ca.cam, conf, seen, pos, pose, ca.plane_depth, stamp, ca.delta_t, ca.conf_thresh, ca.zmin, ca.zmax);
                if (cls == 2) M.conf[i] = -1.0f;
                state_vis[i] = (uint8_t)cls;
            }
        }
#pragma unroll
        for (int c = 0; c < 3; c++) { const int k = __popcll(__ballot(cls == c)); if (lane() == 0) hist[wv][c] = k; }
        __syncthreads();
        if (threadIdx.x < 3) {
            const uint32_t k = hist[0][threadIdx.x] + hist[1][threadIdx.x] + hist[2][threadIdx.x] + hist[3][threadIdx.x];
            if (k) {
                atomicAdd(&ws.sup_vis[(vb / PART_GROUP) * 6 + threadIdx.x], k);
                atomic_add_done(&ws.tot[(vb & (PART_REPLICAS - 1)) * 8 + threadIdx.x], k);
            }
        }
    } else if (b >= nupd)
        insert_chunk(M, F, pose, stamp, matched, S, capacity, rank, nranks, tile, cnt, wave_tot, b - nupd, nchunks, ca, state_vis, ws);
    else if (do_update) {
        ShardArgs sh; sh.rank = rank; sh.nranks = nranks; sh.migrate = (migrate && nranks > 1) ? 1 : 0; sh.tile = tile;
        update_group(M, F, pose, stamp, id_offset, n_visible, best, matched, S, cnt, b * UPD_PER_WG, ca, state_vis, ws, sh);
    }
    // tail_in_move (a single shard): the launch ENDS here.  What follows -- three dependent trips to the coherence point: the
    // adds' acknowledgements, the two-level arrival ticket, the last block's read-back -- only existed to turn the class totals
    // into the frame's counters before the row moves; the kernel boundary in front of k_move_rows completes every atomic of this
    // launch for free, every block of the move kernel sums the eight replicas of the totals it needs itself (64 words), and its
    // block 0 finalises and publishes the counters (move_totals / the head of k_move_rows).
#ifdef SSF_EXPERIMENTS
    if (trace) {
        __syncthreads();
        if (threadIdx.x == 0) trace[3 * (size_t)b + 1] = wall_clock64();
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) trace[3 * (size_t)b + 2] = wall_clock64();
    }
#endif
    if (tail_in_move) return;
    // the atomics above (and cnt->n_updated / n_inserted) are device-scope, complete (vmcnt(0) + barrier) before this
    // block counts its arrival; the last block reads them back with device-scope atomic loads (same protocol as the
    // ICP record)
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (threadIdx.x == 0) s_last = grid_arrive(ws.ticket);
    __syncthreads();
    if (!s_last) return;
    if (threadIdx.x < 8) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < PART_REPLICAS; r++) v += __hip_atomic_load(&ws.tot[r * 8 + threadIdx.x], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        tot[threadIdx.x] = v;
    }
    __syncthreads();
    if (threadIdx.x >= 64) {                      // (off the publishing wave's path)
        for (int i = threadIdx.x - 64; i < ws.words; i += blockDim.x - 64) ws.other[i] = 0u;      // the other set: next frame's sums
    } else if (threadIdx.x == 0) {
        Counters c_in = *cnt;
        c_in.n_inserted = __hip_atomic_load(&cnt->n_inserted, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // written in this launch
        c_in.n_updated = __hip_atomic_load(&cnt->n_updated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c_in.n_emigrated = __hip_atomic_load(&cnt->n_emigrated, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        // the frame's counters are final here; the move kernel that follows in the stream sends them to the host first
        // thing (cnt[1]), so that this launch does not end on the acknowledgement of writes to host memory
        cnt[1] = finalise_counters(cnt, frame_counters(c_in, tot), 1);
    }
}

// first frame: thrust::copy(frame -> model), supersurfel_fusion.cu:477-483 (owned rows only)
__global__ __launch_bounds__(1024) void k_first_frame(SurfelSoA M, SurfelSoA F, Rt pose, int S, int capacity, int rank,
                                                      int nranks, float tile, Counters* cnt) {
    __shared__ int wave_tot[16];
    int running = 0;
    for (int c0 = 0; c0 < S; c0 += 1024) {
        const int f = c0 + threadIdx.x;
        const bool flag = (f < S) && shard_owner(F, f, pose, nranks, tile) == rank;
        int total;
        const int r = block_rank_1024(flag, wave_tot, total);
        const int k = running + r;
        if (flag && k < capacity) {
            st3(M.pos, k, ld3(F.pos, f)); st3(M.col, k, ld3(F.col, f)); st3(M.lab, k, ld3(F.lab, f));
            M.stamps[2 * k] = F.stamps[2 * f]; M.stamps[2 * k + 1] = F.stamps[2 * f + 1];
            st3(M.r0, k, ld3(F.r0, f)); st3(M.r1, k, ld3(F.r1, f)); st3(M.r2, k, ld3(F.r2, f));
            st6(M.shape, k, ld6(F.shape, f));
            M.dims[2 * k] = F.dims[2 * f]; M.dims[2 * k + 1] = F.dims[2 * f + 1];
            M.conf[k] = F.conf[f];
        }
        running += total;
    }
    if (threadIdx.x == 0) {
        const int n = min(running, capacity);
        cnt->n_model = n; cnt->n_visible = n; cnt->n_inserted = 0; cnt->n_removed = 0;
    }
}

// ---- stable partition over the model store (see OovStore in ssf_device.hpp) ----------------------------------------
// exclusive scan of NC counters per block over nblocks blocks by one 1024-thread workgroup (one block per thread
// and round: coalesced loads); totals -> tot[NC].  (Out-of-view store compaction only.)
template <int NC>
__device__ __forceinline__ void block_scan_counts(uint32_t* __restrict__ bc, int nblocks, uint32_t* tot /* LDS, NC */,
                                                  uint32_t (*wtot)[6] /* LDS, 16 x 6 */) {
    if (threadIdx.x < NC) tot[threadIdx.x] = 0;
    __syncthreads();
    for (int b0 = 0; b0 < nblocks; b0 += 1024) {
        const int b = b0 + threadIdx.x;
        uint32_t c[NC], incl[NC];
#pragma unroll
        for (int s = 0; s < NC; s++) c[s] = (b < nblocks) ? bc[NC * b + s] : 0u;
#pragma unroll
        for (int s = 0; s < NC; s++) {
            uint32_t v = c[s];
#pragma unroll
            for (int o = 1; o < 64; o <<= 1) { const uint32_t up = __shfl_up(v, o, 64); if (lane() >= o) v += up; }
            incl[s] = v;
            if (lane() == 63) wtot[threadIdx.x >> 6][s] = v;
        }
        __syncthreads();
        uint32_t before[NC], total[NC];
#pragma unroll
        for (int s = 0; s < NC; s++) { before[s] = 0; total[s] = 0; }
        for (int w = 0; w < 16; w++)
#pragma unroll
            for (int s = 0; s < NC; s++) { const uint32_t t = wtot[w][s]; if (w < (int)(threadIdx.x >> 6)) before[s] += t; total[s] += t; }
        if (b < nblocks)
#pragma unroll
            for (int s = 0; s < NC; s++) bc[NC * b + s] = tot[s] + before[s] + incl[s] - c[s];
        __syncthreads();
        if (threadIdx.x < NC) tot[threadIdx.x] += total[threadIdx.x];
        __syncthreads();
    }
}

// one model row in registers: all loads are issued before the first store (source and destination arrays may alias as
// far as the compiler knows, so a field-by-field copy is a chain of dependent round trips)
struct RowRegs { V3 pos, col, lab, r0, r1, r2; Sym3 shape; int s0, s1; float d0, d1, conf; };
__device__ __forceinline__ RowRegs load_row(const SurfelSoA& A, size_t i) {
    RowRegs r;
    r.pos = ld3(A.pos, i); r.col = ld3(A.col, i); r.lab = ld3(A.lab, i);
    r.s0 = A.stamps[2 * i]; r.s1 = A.stamps[2 * i + 1];
    r.r0 = ld3(A.r0, i); r.r1 = ld3(A.r1, i); r.r2 = ld3(A.r2, i);
    r.shape = ld6(A.shape, i);
    r.d0 = A.dims[2 * i]; r.d1 = A.dims[2 * i + 1];
    r.conf = A.conf[i];
    return r;
}
__device__ __forceinline__ void store_row(const SurfelSoA& B, size_t j, const RowRegs& r) {
    st3(B.pos, j, r.pos); st3(B.col, j, r.col); st3(B.lab, j, r.lab);
    B.stamps[2 * j] = r.s0; B.stamps[2 * j + 1] = r.s1;
    st3(B.r0, j, r.r0); st3(B.r1, j, r.r1); st3(B.r2, j, r.r2);
    st6(B.shape, j, r.shape);
    B.dims[2 * j] = r.d0; B.dims[2 * j + 1] = r.d1;
    B.conf[j] = r.conf;
}
__device__ __forceinline__ void copy_row(const SurfelSoA& A, size_t i, const SurfelSoA& B, size_t j) { store_row(B, j, load_row(A, i)); }
// copy of a row that enters the new visible array; hands back the three fields the ICP terms read
__device__ __forceinline__ void copy_row_keep(const SurfelSoA& A, size_t i, const SurfelSoA& B, size_t j, V3& pos, V3& lab, V3& nrm) {
    const RowRegs r = load_row(A, i);
    pos = r.pos; lab = r.lab; nrm = r.r2;
    store_row(B, j, r);
}
// what the fused form of k_move_rows needs to accumulate the first ICP iteration of the NEXT frame
struct NextIcp {
    Cam cam; const uint2* pix2; const float4* fpack; Rt T;
    long long* replicas; unsigned int* ticket; long long* sums; Mailbox* mb; unsigned long long seq;
};
// move the rows whose place changes (stable within each class): A0, C0 -> new visible array, A1 -> in front of the
// out-of-view span, C1 -> behind it, B0 -> new visible array (slot freed), B2 -> slot freed.  B1 stays where it is.
// ICP = true: the rows written to the new visible array (A0, B0, C0) are exactly the rows the next frame's first ICP
// iteration reads, and its transform (inverse of the pose just estimated) is already known: accumulate that record
// here, while the rows are in registers, against the next frame's packed tables (the sums are exact integers, so
// the order of accumulation does not matter), and publish it like k_icp does.
// P2P (with ICP): the record is this shard's; the publishing workgroup trades it with the peers as k_icp<true> does.
// (Measured and removed, round 2: one more workgroup of this launch trading the shard sizes with the peers instead of a launch of
// its own in front of it -- 7030-7180 against 7260-7320 frames/s with every exchange on one rank, nothing at 2 / 4 ranks.)
template <bool ICP, bool P2P>
__global__ __launch_bounds__(256) void k_move_rows(SurfelSoA V, SurfelSoA Vn, OovStore O, const uint8_t* __restrict__ state_vis,
                                                   const uint8_t* __restrict__ state_oov,
                                                   const uint32_t* __restrict__ bc_oov, PartitionWs ws,
                                                   Counters* cnt, int nb_vis, NextIcp nx, Mailbox* mb,
                                                   unsigned long long cnt_seq, P2PView pv, MoveTotals mt) {
    __builtin_amdgcn_s_setprio(3);            // the track chain is the critical path: its waves issue ahead of the extract waves sharing a SIMD
    // the frame's counters go to the host while the rows move: finalised by the fuse launch (cnt[1]), or -- mt.from_tot, a
    // single shard: the fuse launch ended without its tail -- worked out HERE by block 0 from the class totals (the kernel
    // boundary has completed every atomic of the fuse launch), which also clears the other set of partition sums for the next
    // frame.  No other block of this launch reads a word that block 0 changes: they take the old counts from the launch's
    // arguments (the host mirrors them), the insertion count from cnt->n_inserted (stored, not accumulated: left alone) and
    // the totals from the replicas.
    if (blockIdx.x == 0 && threadIdx.x == 255) {
        Counters c;
        if (mt.from_tot) {
            uint32_t tot[8];
#pragma unroll
            for (int k = 0; k < 8; k++) {
                uint32_t v = 0;
#pragma unroll
                for (int r = 0; r < PART_REPLICAS; r++) v += __builtin_nontemporal_load(&ws.tot[r * 8 + k]);
                tot[k] = v;
            }
            const Counters c_in = *cnt;
            c = finalise_counters(cnt, frame_counters(c_in, tot), 1, /*keep_inserted=*/1);     // (the other blocks read cnt->n_inserted; the next frame's insertion stores it anew)
            cnt[1] = c;
        } else {
            int w[sizeof(Counters) / sizeof(int)];                    // (all words requested before the first store to the host)
            const int* src = reinterpret_cast<const int*>(&cnt[1]);
#pragma unroll
            for (int i = 0; i < (int)(sizeof(Counters) / sizeof(int)); i++) w[i] = __builtin_nontemporal_load(&src[i]);
            __builtin_memcpy(&c, w, sizeof(c));
        }
        mailbox_counters(c, mb, cnt_seq);
    }
    if (mt.from_tot && blockIdx.x == 0 && threadIdx.x >= 64 && threadIdx.x < 255)
        for (int i = threadIdx.x - 64; i < ws.words; i += 191) ws.other[i] = 0u;      // the other set: next frame's sums
    // out-of-view blocks: nothing moves in most of them (bc_oov: see classify_oov_block) -- leave at once
    if ((int)blockIdx.x >= nb_vis && bc_oov[blockIdx.x - nb_vis] == 0u) return;
    // the frozen inputs of the move (Counters::mv_*): from the fuse launch's last block, or from the arguments + the replicas
    __shared__ uint32_t s_tot[8];
    if (mt.from_tot && threadIdx.x < 8) {
        uint32_t v = 0;
#pragma unroll
        for (int r = 0; r < PART_REPLICAS; r++) v += ws.tot[r * 8 + threadIdx.x];
        s_tot[threadIdx.x] = v;
    }
    const int mv_nv = mt.from_tot ? mt.nv : cnt->mv_nv, mv_nc = mt.from_tot ? cnt->n_inserted : cnt->mv_nc;
    const int mv_head_old = mt.from_tot ? mt.head_old : cnt->mv_head_old, mv_tail_old = mt.from_tot ? mt.tail_old : cnt->mv_tail_old;
    auto mv_a0 = [&]() -> int { return mt.from_tot ? (int)s_tot[0] : cnt->mv_a0; };          // (s_tot: valid behind the next barrier)
    auto mv_b0 = [&]() -> int { return mt.from_tot ? (int)s_tot[6] : cnt->mv_b0; };
    auto mv_head_new = [&]() -> int { return mt.from_tot ? mt.head_old - (int)s_tot[1] : cnt->mv_head_new; };
    auto new_n_visible = [&]() -> int { return mt.from_tot ? (int)(s_tot[0] + s_tot[6] + s_tot[3]) : cnt->n_visible; };
    __shared__ int hist[4][6];
    __shared__ uint32_t base[6];                  // rows of each class in the blocks before this one
    __shared__ unsigned long long red[ICP ? 29 * ICP_SLOTS : 1];
    if (ICP) for (int i = threadIdx.x; i < 29 * ICP_SLOTS; i += blockDim.x) red[i] = 0ull;
    if (threadIdx.x < 6) base[threadIdx.x] = 0u;
    const int wv = threadIdx.x >> 6;
    int cls = 7, in_wave = 0;
    bool keep = false;
    V3 pos, lab, nrm;
    if ((int)blockIdx.x < nb_vis) {
        const int nv = mv_nv, n_rows = nv + mv_nc;                        // mv_nc = rows appended in this frame (insertions + arrivals)
        const int i = blockIdx.x * blockDim.x + threadIdx.x;
        // the row itself is requested now, together with its state byte (not behind it: a row that turns out to be removed is
        // fetched for nothing, every other row saves a dependent trip to memory): the loads travel while the prefix below
        // is worked out
        RowRegs row;
        if (i < n_rows) {
            int st = (int)state_vis[i];
            row = load_row(V, i);
            asm volatile("" : "+v"(st), "+v"(row.pos.x));
            cls = (i < nv ? 0 : 3) + st;
        }
        // prefix: the sums of the groups before this block's group (6 counters wide: word w belongs to class w % 6),
        // then the states of the rows of the earlier blocks of its own group, 16 per load
        const int g0 = (int)(blockIdx.x / PART_GROUP), ng = 6 * g0;
        const int row0 = g0 * PART_GROUP * 256, n16 = 16 * ((int)blockIdx.x - g0 * PART_GROUP);
        __syncthreads();
        for (int w = threadIdx.x; w < ng; w += blockDim.x) {
            const uint32_t v = ws.sup_vis[w];
            if (v) atomicAdd(&base[w % 6], v);
        }
        uint32_t pa0 = 0, pa1 = 0, pc0 = 0, pc1 = 0;
        for (int q = threadIdx.x; q < n16; q += blockDim.x) {
            const int r0 = row0 + 16 * q;
            if (r0 < n_rows) {
                const uint4 sv = *reinterpret_cast<const uint4*>(state_vis + r0);
                const uint32_t wd[4] = {sv.x, sv.y, sv.z, sv.w};
                uint32_t a0 = 0, a1 = 0, c0 = 0, c1 = 0;         // the classes that move
#pragma unroll
                for (int j = 0; j < 16; j++) {
                    const int rr = r0 + j;
                    const uint32_t stt = (wd[j >> 2] >> (8 * (j & 3))) & 0xFFu;
                    const bool in = rr < n_rows, old = rr < nv;
                    a0 += (in && old && stt == 0u) ? 1u : 0u; a1 += (in && old && stt == 1u) ? 1u : 0u;
                    c0 += (in && !old && stt == 0u) ? 1u : 0u; c1 += (in && !old && stt == 1u) ? 1u : 0u;
                }
                pa0 += a0; pa1 += a1; pc0 += c0; pc1 += c1;
            }
        }
        // (one LDS atomic per wave and class: 256 lanes on four addresses would serialise)
#pragma unroll
        for (int o = 32; o > 0; o >>= 1) {
            pa0 += __shfl_xor(pa0, o, 64); pa1 += __shfl_xor(pa1, o, 64); pc0 += __shfl_xor(pc0, o, 64); pc1 += __shfl_xor(pc1, o, 64);
        }
        if (lane() == 0) {
            if (pa0) atomicAdd(&base[0], pa0);
            if (pa1) atomicAdd(&base[1], pa1);
            if (pc0) atomicAdd(&base[3], pc0);
            if (pc1) atomicAdd(&base[4], pc1);
        }
#pragma unroll
        for (int c = 0; c < 6; c++) {
            const unsigned long long mask = __ballot(cls == c);
            if (cls == c) in_wave = __popcll(mask & ((1ull << lane()) - 1ull));
            if (lane() == 0) hist[wv][c] = __popcll(mask);
        }
        __syncthreads();
        if (cls == 0 || cls == 1 || cls == 3 || cls == 4) {
            int before = 0;
            for (int w = 0; w < wv; w++) before += hist[w][cls];
            const size_t r = (size_t)base[cls] + before + in_wave;
            if (cls == 0 || cls == 3) {
                const size_t j = cls == 0 ? r : (size_t)mv_a0() + mv_b0() + r;
                store_row(Vn, j, row);
                if (ICP) { pos = row.pos; lab = row.lab; nrm = row.r2; keep = true; }
            } else {
                const size_t j = (cls == 1 ? (size_t)mv_head_new() : (size_t)mv_tail_old) + r;
                store_row(O.rows, j, row);
                O.live[j] = 1;
            }
        }
    } else {
        const int ob = blockIdx.x - nb_vis;
        const long long phys = (long long)mv_head_old + (long long)ob * blockDim.x + threadIdx.x;
        if (phys < mv_tail_old) {                  // (both bytes in one round trip)
            int lv = (int)O.live[phys], st = (int)state_oov[phys];
            asm volatile("" : "+v"(lv), "+v"(st));
            if (lv) cls = st;
        }
        const unsigned long long mask = __ballot(cls == 0);
        if (cls == 0) in_wave = __popcll(mask & ((1ull << lane()) - 1ull));
        if (lane() == 0) hist[wv][0] = __popcll(mask);
        if (__syncthreads_count(cls == 0)) {      // (few blocks have rows that come back into view)
            const int g0 = ob / PART_GROUP, nw = g0 + (ob - g0 * PART_GROUP);
            for (int w = threadIdx.x; w < nw; w += blockDim.x) {
                const uint32_t v = w < g0 ? ws.sup_oov[w] : (bc_oov[g0 * PART_GROUP + (w - g0)] & 0xFFFFu);
                if (v) atomicAdd(&base[0], v);
            }
            __syncthreads();
        }
        if (cls == 0) {
            int before = 0;
            for (int w = 0; w < wv; w++) before += hist[w][0];
            const size_t j = (size_t)mv_a0() + base[0] + before + in_wave;
            if (ICP) { copy_row_keep(O.rows, (size_t)phys, Vn, j, pos, lab, nrm); keep = true; }
            else copy_row(O.rows, (size_t)phys, Vn, j);
        }
        if (cls == 0 || cls == 2) O.live[phys] = 0;
    }
    if (ICP) {
        // (the barrier after the class histogram also ordered the zeroing of `red`)
        if (keep) icp_row(nx.cam, nx.pix2, nx.fpack, nx.T.R, nx.T.t, pos, lab, nrm, red, lane() & (ICP_SLOTS - 1), 0);
        // Arrivals are counted in ROWS: the workgroup that completes cnt->n_visible (written by the fuse launch) is
        // the last.  Two levels, as in k_icp (thousands of returning atomics on one word serialise at L2): the
        // blocks of the visible array that hold rows at all (index < nbr) arrive at one of 64 group words packed
        // (rows << 32 | arrivals); the last of a group (its size follows from nbr) takes the group's rows to the
        // global row counter.  The few out-of-view blocks with rows that come back into view go there directly.
        // No rows at all: no record is published, and the host does not ask for one (ICP needs visible rows).
        const int nkeep = __syncthreads_count(keep);
        if (P2P && new_n_visible() == 0) {         // an empty shard still owes its peers a (zero) record
            if (blockIdx.x == 0) icp_publish<true>(nx.replicas, nx.sums, nx.mb, nx.seq, pv, pv.seq);
            return;
        }
        const bool vis = (int)blockIdx.x < nb_vis;
        const int nbr = (mv_nv + mv_nc + 255) / 256;
        if (vis ? (int)blockIdx.x >= nbr : nkeep == 0) return;
        __shared__ int s_last;
        icp_fold(red, nx.replicas);
        if (threadIdx.x == 0) {
            SSF_ARRIVE_RELEASE();
            unsigned int rows = (unsigned int)nkeep;
            bool report = true;
            if (vis) {
                const unsigned int g = blockIdx.x & 63u, in_group = ((unsigned int)nbr - g + 63u) / 64u;
                unsigned long long* gw = reinterpret_cast<unsigned long long*>(nx.ticket + 128) + g;
                const unsigned long long before = __hip_atomic_fetch_add(gw, ((unsigned long long)rows << 32) | 1ull, __ATOMIC_RELAXED,
                                                                         __HIP_MEMORY_SCOPE_AGENT);
                report = (unsigned int)(before & 0xFFFFFFFFull) == in_group - 1;
                if (report) {
                    rows += (unsigned int)(before >> 32);
                    __hip_atomic_store(gw, 0ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
            }
            int last = 0;
            if (report) {
                const unsigned int total = (unsigned int)new_n_visible();
                const unsigned int before = rows ? __hip_atomic_fetch_add(&nx.ticket[65], rows, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) : 0u;
                last = rows != 0u && before + rows == total;
                if (last) __hip_atomic_store(&nx.ticket[65], 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            if (last) SSF_ARRIVE_ACQUIRE();
            s_last = last;
        }
        __syncthreads();
        if (s_last) icp_publish<P2P>(nx.replicas, nx.sums, nx.mb, nx.seq, pv, pv.seq);
    }
}

// ---- multi-GPU migration: rows that crossed a tile edge move to the rank that owns their new tile -----------------
// slot f of the migrant table <- the model row that frame supersurfel f updated, if the fuse launch marked it as
// leaving (partition state 2 with its confidence intact); every other slot is zeroed.  One thread per slot.
__global__ __launch_bounds__(256) void k_pack_emigrants(SurfelSoA M, const unsigned long long* __restrict__ best,
                                                        const uint8_t* __restrict__ matched, long long id_offset, int n_visible,
                                                        const uint8_t* __restrict__ state_vis, int S, int do_update, int nranks,
                                                        float tile, int32_t* __restrict__ table) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= S) return;
    int32_t w[SSF_MIGRANT_WORDS];
#pragma unroll
    for (int i = 0; i < SSF_MIGRANT_WORDS; i++) w[i] = 0;
    if (do_update && matched[f] && best[f] != SSF_NO_MATCH) {
        const long long local = (long long)(uint32_t)(best[f] & 0xFFFFFFFFull) - id_offset;
        if (local >= 0 && local < n_visible) {
            const size_t m = (size_t)local;
            const float conf = M.conf[m];
            if (state_vis[m] == 2 && conf > 0.0f) {
                const RowRegs r = load_row(M, m);
                w[0] = tile_owner(r.pos, nranks, tile) + 1;
                const float v[26] = {r.pos.x, r.pos.y, r.pos.z, r.col.x, r.col.y, r.col.z, __int_as_float(r.s0), __int_as_float(r.s1),
                                     r.r0.x, r.r0.y, r.r0.z, r.r1.x, r.r1.y, r.r1.z, r.r2.x, r.r2.y, r.r2.z,
                                     r.shape.xx, r.shape.xy, r.shape.xz, r.shape.yy, r.shape.yz, r.shape.zz, r.d0, r.d1, r.conf};
#pragma unroll
                for (int i = 0; i < 26; i++) w[2 + i] = __float_as_int(v[i]);
            }
        }
    }
    int32_t* o = table + (size_t)SSF_MIGRANT_WORDS * f;
#pragma unroll
    for (int i = 0; i < SSF_MIGRANT_WORDS; i++) o[i] = w[i];
}
// rows of the (rank-reduced) table addressed to this rank -> appended behind this frame's insertions in the visible
// array (ascending f), classified like any inserted row; partition sums and both counter sets corrected.  One
// workgroup: at most S rows arrive, a handful in practice.
__global__ __launch_bounds__(1024) void k_migrate_in(SurfelSoA M, const int32_t* __restrict__ table, int S, int rank, int capacity,
                                                     Counters* cnt, ClassifyArgs ca, Rt pose, int stamp,
                                                     uint8_t* __restrict__ state_vis, PartitionWs ws) {
    __shared__ int wave_tot[16];
    __shared__ int s_in[4];                 // arrivals by state; [3] = arrivals turned away (shard at capacity)
    if (threadIdx.x < 4) s_in[threadIdx.x] = 0;
    const Counters c1 = cnt[1];
    const int base_row = c1.mv_nv + c1.mv_nc;                  // first free row behind the insertions
    const int total_before = c1.n_model + c1.n_state2;         // rows in the store before this frame's removals
    int running = 0;
    __syncthreads();
    for (int f0 = 0; f0 < S; f0 += 1024) {
        const int f = f0 + threadIdx.x;
        const bool flag = f < S && table[(size_t)SSF_MIGRANT_WORDS * f] - 1 == rank;
        int total;
        const int r = block_rank_1024(flag, wave_tot, total);
        if (flag && total_before + running + r < capacity) {
            const int32_t* w = table + (size_t)SSF_MIGRANT_WORDS * f;
            float v[26];
#pragma unroll
            for (int i = 0; i < 26; i++) v[i] = __int_as_float(w[2 + i]);
            RowRegs row;
            row.pos = v3(v[0], v[1], v[2]); row.col = v3(v[3], v[4], v[5]); row.lab = rgb_to_lab(row.col);
            row.s0 = __float_as_int(v[6]); row.s1 = __float_as_int(v[7]);
            row.r0 = v3(v[8], v[9], v[10]); row.r1 = v3(v[11], v[12], v[13]); row.r2 = v3(v[14], v[15], v[16]);
            row.shape = sym3(v[17], v[18], v[19], v[20], v[21], v[22]); row.d0 = v[23]; row.d1 = v[24]; row.conf = v[25];
            const int st = classify_values(ca.cam, row.conf, row.s1, row.pos, pose, ca.plane_depth, stamp, ca.delta_t, ca.conf_thresh,
                                           ca.zmin, ca.zmax);
            if (st == 2) row.conf = -1.0f;
            const size_t k = (size_t)base_row + running + r;
            store_row(M, k, row);
            state_vis[k] = (uint8_t)st;
            atomicAdd(&ws.sup_vis[((k >> 8) / PART_GROUP) * 6 + 3 + st], 1u);
            atomicAdd(&s_in[st], 1);
        } else if (flag)
            atomicAdd(&s_in[3], 1);             // no room: lost to the whole map (the source shard has let it go) -> counted as removed here
        running += min(total, max(0, capacity - total_before - running));
        __syncthreads();
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int in0 = s_in[0], in1 = s_in[1], in2 = s_in[2], n_in = in0 + in1 + in2, lost = s_in[3];
        if (n_in || lost) {
            for (int q = 0; q < 2; q++) {
                Counters c = cnt[q];
                c.n_model += in0 + in1; c.n_visible += in0;
                c.oov_tail += in1; c.oov_live += in1; c.mv_nc += n_in;
                if (q == 1) { c.n_state0 += in0; c.n_state1 += in1; c.n_state2 += in2; c.n_removed += in2 + lost; }
                c.last[0] = c.n_model; c.last[1] = c.n_visible; if (q == 1) c.last[2] = c.n_removed; else c.last[2] += in2 + lost;
                cnt[q] = c;
            }
        }
    }
}

// ---- re-homing of a sharded map after positions changed outside a frame (ssf_rehome_begin / _end in ssf.h) ---------------
// over the dense logical view [visible | out of view]: a row leaves when it is valid and its position belongs to another
// rank's tile.  Three launches: per-block counts (staying rows, leaving rows, staying rows of the visible block), their
// exclusive scan by one workgroup (totals -> tot3), and the ordered scatter: staying rows close ranks in `stay`, leaving
// rows become migrant-table records (word 1 = 1: the row sat in the visible block) in logical order.
__device__ __forceinline__ int rehome_class(const SurfelSoA& D, int i, int n, int rank, int nranks, float tile) {
    if (i >= n) return 2;                                          // 0 stays, 1 leaves
    return (D.conf[i] > 0.0f && tile_owner(ld3(D.pos, i), nranks, tile) != rank) ? 1 : 0;
}
__global__ __launch_bounds__(256) void k_rehome_count(SurfelSoA D, int n, int n_visible, int rank, int nranks, float tile, uint32_t* __restrict__ bc) {
    __shared__ int part[4][3];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    const int c = rehome_class(D, i, n, rank, nranks, tile);
    const int k0 = __popcll(__ballot(c == 0)), k1 = __popcll(__ballot(c == 1)), k2 = __popcll(__ballot(c == 0 && i < n_visible));
    if (lane() == 0) { part[threadIdx.x >> 6][0] = k0; part[threadIdx.x >> 6][1] = k1; part[threadIdx.x >> 6][2] = k2; }
    __syncthreads();
    if (threadIdx.x < 3) bc[3 * blockIdx.x + threadIdx.x] = part[0][threadIdx.x] + part[1][threadIdx.x] + part[2][threadIdx.x] + part[3][threadIdx.x];
}
__global__ __launch_bounds__(1024) void k_rehome_scan(uint32_t* __restrict__ bc, int nblocks, int* __restrict__ tot3) {
    __shared__ uint32_t wtot[16][6];
    __shared__ uint32_t tot[3];
    block_scan_counts<3>(bc, nblocks, tot, wtot);
    if (threadIdx.x < 3) tot3[threadIdx.x] = (int)tot[threadIdx.x];
}
__global__ __launch_bounds__(256) void k_rehome_scatter(SurfelSoA D, int n, int n_visible, int rank, int nranks, float tile,
                                                        const uint32_t* __restrict__ bc, SurfelSoA stay, int32_t* __restrict__ table, int table_rows) {
    __shared__ int part[4][2];
    const int i = blockIdx.x * blockDim.x + threadIdx.x, wv = threadIdx.x >> 6;
    const int c = rehome_class(D, i, n, rank, nranks, tile);
    const unsigned long long m0 = __ballot(c == 0), m1 = __ballot(c == 1);
    if (lane() == 0) { part[wv][0] = __popcll(m0); part[wv][1] = __popcll(m1); }
    __syncthreads();
    if (c > 1) return;
    int before = 0;
    for (int w = 0; w < wv; w++) before += part[w][c];
    const size_t j = (size_t)bc[3 * blockIdx.x + c] + before + __popcll((c == 0 ? m0 : m1) & ((1ull << lane()) - 1ull));
    if (c == 0) { copy_row(D, (size_t)i, stay, j); return; }
    if (j >= (size_t)table_rows) return;                               // (the host sees the total and reports SSF_ERR_CAPACITY)
    const RowRegs r = load_row(D, (size_t)i);
    int32_t* o = table + (size_t)SSF_MIGRANT_WORDS * j;
    o[0] = tile_owner(r.pos, nranks, tile) + 1; o[1] = i < n_visible ? 1 : 0;
    const float v[26] = {r.pos.x, r.pos.y, r.pos.z, r.col.x, r.col.y, r.col.z, __int_as_float(r.s0), __int_as_float(r.s1),
                         r.r0.x, r.r0.y, r.r0.z, r.r1.x, r.r1.y, r.r1.z, r.r2.x, r.r2.y, r.r2.z,
                         r.shape.xx, r.shape.xy, r.shape.xz, r.shape.yy, r.shape.yz, r.shape.zz, r.d0, r.d1, r.conf};
#pragma unroll
    for (int q = 0; q < 26; q++) o[2 + q] = __float_as_int(v[q]);
}
// records [0, n) of a packed table -> rows [base, base + n) of D (the Lab cache is rebuilt from the colour)
__global__ __launch_bounds__(256) void k_rehome_unpack(const int32_t* __restrict__ table, int n, SurfelSoA D, int base) {
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int32_t* w = table + (size_t)SSF_MIGRANT_WORDS * j;
    float v[26];
#pragma unroll
    for (int q = 0; q < 26; q++) v[q] = __int_as_float(w[2 + q]);
    RowRegs row;
    row.pos = v3(v[0], v[1], v[2]); row.col = v3(v[3], v[4], v[5]); row.lab = rgb_to_lab(row.col);
    row.s0 = __float_as_int(v[6]); row.s1 = __float_as_int(v[7]);
    row.r0 = v3(v[8], v[9], v[10]); row.r1 = v3(v[11], v[12], v[13]); row.r2 = v3(v[14], v[15], v[16]);
    row.shape = sym3(v[17], v[18], v[19], v[20], v[21], v[22]); row.d0 = v[23]; row.d1 = v[24]; row.conf = v[25];
    store_row(D, (size_t)base + j, row);
}
void launch_rehome_split(hipStream_t st, SurfelSoA dense, int n, int n_visible, int rank, int nranks, float tile, uint32_t* bc, int* tot3,
                         SurfelSoA stay, int32_t* table, int table_rows) {
    const int nb = (n + 255) / 256;
    if (nb <= 0) return;
    hipLaunchKernelGGL(k_rehome_count, dim3(nb), dim3(256), 0, st, dense, n, n_visible, rank, nranks, tile, bc);
    hipLaunchKernelGGL(k_rehome_scan, dim3(1), dim3(1024), 0, st, bc, nb, tot3);
    hipLaunchKernelGGL(k_rehome_scatter, dim3(nb), dim3(256), 0, st, dense, n, n_visible, rank, nranks, tile, bc, stay, table, table_rows);
}
void launch_rehome_unpack(hipStream_t st, const int32_t* table, int n, SurfelSoA dst, int base) {
    if (n > 0) hipLaunchKernelGGL(k_rehome_unpack, dim3((n + 255) / 256), dim3(256), 0, st, table, n, dst, base);
}

// ---- out-of-view store maintenance: stable compaction of the live rows into the other store -------------------
__global__ __launch_bounds__(256) void k_oov_count(OovStore O, uint32_t* __restrict__ bc, const Counters* __restrict__ cnt) {
    __shared__ int part[4];
    const long long phys = (long long)cnt->oov_head + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool lv = phys < cnt->oov_tail && O.live[phys];
    const int k = __popcll(__ballot(lv));
    if (lane() == 0) part[threadIdx.x >> 6] = k;
    __syncthreads();
    if (threadIdx.x == 0) bc[blockIdx.x] = part[0] + part[1] + part[2] + part[3];
}
__global__ __launch_bounds__(1024) void k_oov_scan(uint32_t* __restrict__ bc, int nb_upper, const Counters* __restrict__ cnt) {
    __shared__ uint32_t wtot[16][6];
    __shared__ uint32_t tot[1];
    const int nb = min(nb_upper, (cnt->oov_tail - cnt->oov_head + 255) / 256);
    block_scan_counts<1>(bc, nb, tot, wtot);
}
__global__ __launch_bounds__(256) void k_oov_compact(OovStore A, OovStore B, const uint32_t* __restrict__ bc, int new_head,
                                                     const Counters* __restrict__ cnt) {
    __shared__ int part[4];
    const long long phys = (long long)cnt->oov_head + (long long)blockIdx.x * blockDim.x + threadIdx.x;
    const bool lv = phys < cnt->oov_tail && A.live[phys];
    const unsigned long long mask = __ballot(lv);
    const int wv = threadIdx.x >> 6;
    if (lane() == 0) part[wv] = __popcll(mask);
    __syncthreads();
    if (lv) {
        int before = 0;
        for (int w = 0; w < wv; w++) before += part[w];
        const size_t j = (size_t)new_head + bc[blockIdx.x] + before + __popcll(mask & ((1ull << lane()) - 1ull));
        copy_row(A.rows, (size_t)phys, B.rows, j);
        B.live[j] = 1;
    }
}
__global__ void k_oov_set_span(Counters* cnt, int new_head) { cnt->oov_head = new_head; cnt->oov_tail = new_head + cnt->oov_live; }

// end of the fuse stage: nbSupersurfels -= nbRemoved (supersurfel_fusion.cu:474), publish the
// counters to the host-mapped mailbox, reset the per-frame ones for the next frame
// finalise_counters: the frame's counters become final in device memory (cnt[0] = the values the next frame starts
// from, per-frame ones reset; cnt[1] = the values to publish); mailbox_counters writes a set of values to the host
__device__ __forceinline__ Counters finalise_counters(Counters* cnt, Counters c, int shrink_by_removed, int keep_inserted) {
    if (shrink_by_removed) c.n_model = c.n_model - c.n_state2;
    c.last[0] = c.n_model; c.last[1] = c.n_visible; c.last[2] = c.n_removed; c.last[3] = c.n_inserted; c.last[4] = c.n_updated;
    Counters next = c;
    next.n_inserted = keep_inserted ? c.n_inserted : 0; next.n_updated = 0; next.n_removed = 0; next.n_state0 = 0; next.n_state1 = 0; next.n_state2 = 0;
    next.n_emigrated = 0;
    cnt[0] = next;
    return c;
}
__device__ __forceinline__ void mailbox_counters(const Counters& c, Mailbox* mb, unsigned long long seq) {
    int* dst = reinterpret_cast<int*>(&mb->cnt);
    const int* src = reinterpret_cast<const int*>(&c);
    unsigned long long check = seq;
    for (int i = 0; i < (int)(sizeof(Counters) / sizeof(int)); i++) {
        __hip_atomic_store(&dst[i], src[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        check += (unsigned long long)(unsigned int)src[i];
    }
    __hip_atomic_store(&mb->cnt_check, check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");                  // payload write-through stores acknowledged
    __hip_atomic_store(&mb->cnt_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__device__ __forceinline__ void publish_counters_value(Counters* cnt, Counters c, int shrink_by_removed, Mailbox* mb, unsigned long long seq) {
    mailbox_counters(finalise_counters(cnt, c, shrink_by_removed), mb, seq);
}
// a 29-value device record (e.g. the rank-reduced ICP system) -> mailbox, as the ICP kernel's tail does
__global__ void k_publish_icp(const long long* __restrict__ rec, Mailbox* mb, unsigned long long seq) {
    __shared__ unsigned long long pay[30];
    long long v = 0;
    if (threadIdx.x < 29) { v = rec[threadIdx.x]; pay[threadIdx.x] = (unsigned long long)v; }
    const unsigned long long check = (unsigned long long)wsum64(v) + seq;
    if (threadIdx.x == 0) pay[29] = check;
    __syncthreads();
    if (threadIdx.x < 40)
        __hip_atomic_store(&mb->icp_rec[threadIdx.x], SSF_ICP_REC_WORD(threadIdx.x, pay, seq), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_publish_all_counts(const int* __restrict__ all5, int n, Mailbox* mb, unsigned long long seq) {
    unsigned long long part = 0;
    for (int i = threadIdx.x; i < n; i += 64) {
        const int v = all5[i];
        __hip_atomic_store(&mb->all_cnt[i], v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        part += (unsigned long long)(unsigned int)v;
    }
    const unsigned long long check = (unsigned long long)wsum64((long long)part) + seq;
    if (threadIdx.x == 0) __hip_atomic_store(&mb->all_check, check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (threadIdx.x == 0) __hip_atomic_store(&mb->all_seq, seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
__global__ void k_publish_counts(Counters* cnt, int shrink_by_removed, Mailbox* mb, unsigned long long seq) {
    publish_counters_value(cnt, *cnt, shrink_by_removed, mb, seq);
}

// the three row streams of the orientations -> packed row-major Mat33 (the reference's layout, supersurfels.hpp:37)
__global__ void k_pack_orient(SurfelSoA s, int n, float* __restrict__ out9) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 a = ld3(s.r0, i), b = ld3(s.r1, i), c = ld3(s.r2, i);
    float* o = out9 + 9 * (size_t)i;
    o[0] = a.x; o[1] = a.y; o[2] = a.z; o[3] = b.x; o[4] = b.y; o[5] = b.z; o[6] = c.x; o[7] = c.y; o[8] = c.z;
}
void launch_pack_orient(hipStream_t st, SurfelSoA s, int n, float* out9) {
    if (n > 0) hipLaunchKernelGGL(k_pack_orient, dim3((n + 255) / 256), dim3(256), 0, st, s, n, out9);
}
__global__ void k_lab_refresh(SurfelSoA s, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) st3(s.lab, i, rgb_to_lab(ld3(s.col, i)));
}

// ---- deformation apply ("next" row) -----------------------------------------------------------------
// rotMatToQuat matrix_math.cuh:529-618; quatToRotMat :512-527 (its wy = q.w*q.z is reproduced)
// The nodes of the deformation graph as ONE 64-BYTE RECORD each (round 6): (g.xyz, R00) (t.xyz, R01) (R02, R10, R11, R12) (R20, R21,
// R22, 0) -- a supersurfel's four nodes are 4 x 4 whole 16-byte loads, each record one aligned 64-byte piece of a line, the table
// 1.28 MB at 20 k nodes.  The node's quaternion (rotMatToQuat: four branches, an IEEE sqrt and a divide) is worked out per row and
// node, as the reference does (deformation_graph_kernels.cu:44-52) -- the same operations on the same operands, so the same bits.
// Measured (tools/deform_probe.py, profiles/deformation_r06.txt, 1 M rows, N / 50 nodes): rounds 2-5 kept a 96-byte record with the
// quaternion precomputed (five gathers per node): 75 us with uniformly random node indices, of which 35 us were the gathers (nodes
// 0 / 1 only: 40 us) -- this form: 60 us; records padded to 128 bytes: 85 (the table's footprint in L2 matters, not lines per
// record); non-temporal row accesses: no change; one or two nodes per round at 8 / 5 waves per SIMD: 68-88.  With node indices that
// follow the row order (what a time-ordered graph gives rows in arrival order) every form runs at 40-42 us = 0.52-0.55 of HBM peak.
__global__ __launch_bounds__(256) void k_pack_nodes(int m, const float* __restrict__ npos, const float* __restrict__ nrot,
                                                    const float* __restrict__ ntrans, float4* __restrict__ nodes) {
    const int k = blockIdx.x * blockDim.x + threadIdx.x;
    if (k >= m) return;
    const V3 g = ld3(npos, k), t = ld3(ntrans, k);
    float4* o = nodes + 4 * (size_t)k;
    o[0] = make_float4(g.x, g.y, g.z, nrot[9 * k]); o[1] = make_float4(t.x, t.y, t.z, nrot[9 * k + 1]);
    o[2] = make_float4(nrot[9 * k + 2], nrot[9 * k + 3], nrot[9 * k + 4], nrot[9 * k + 5]);
    o[3] = make_float4(nrot[9 * k + 6], nrot[9 * k + 7], nrot[9 * k + 8], 0.f);
}
__global__ __launch_bounds__(256) void k_deformation(SurfelSoA M, int n, const float4* __restrict__ nodes,
                                                     const float* __restrict__ w4, const int32_t* __restrict__ idx4) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const V3 pi = ld3(M.pos, i);
    const int4 id = reinterpret_cast<const int4*>(idx4)[i];
    const float4 w = reinterpret_cast<const float4*>(w4)[i];
    const int node[4] = {id.x, id.y, id.z, id.w};
    const float wv[4] = {w.x, w.y, w.z, w.w};
    float4 rec[4][4];
#pragma unroll
    for (int k = 0; k < 4; k++)
#pragma unroll
        for (int j = 0; j < 4; j++) rec[k][j] = nodes[4 * (size_t)node[k] + j];       // all sixteen gathers in one round
    V3 po = v3(0, 0, 0);
    float bq[4] = {0, 0, 0, 0};
#pragma unroll
    for (int k = 0; k < 4; k++) {
        const float wk = wv[k];
        const V3 gk = v3(rec[k][0].x, rec[k][0].y, rec[k][0].z), tk = v3(rec[k][1].x, rec[k][1].y, rec[k][1].z);
        const M3 Rk = m3(v3(rec[k][0].w, rec[k][1].w, rec[k][2].x), v3(rec[k][2].y, rec[k][2].z, rec[k][2].w),
                         v3(rec[k][3].x, rec[k][3].y, rec[k][3].z));
        // rot_to_quat (ssf_math.hpp) with its result in named scalars: through the array the compiler kept it in scratch
        float s_, qx, qy, qz, qw; const float tr = (Rk.r0.x + Rk.r1.y) + Rk.r2.z;
        if (tr > 0) { s_ = sqrtf(tr + 1); qw = 0.5f * s_; s_ = 0.5f / s_; qx = (Rk.r2.y - Rk.r1.z) * s_; qy = (Rk.r0.z - Rk.r2.x) * s_; qz = (Rk.r1.x - Rk.r0.y) * s_; }
        else {
            int ii = 0;
            if (Rk.r1.y > Rk.r0.x) ii = 1;
            if (Rk.r2.z > Rk.r0.x || Rk.r2.z > Rk.r1.y) ii = 2;
            if (ii == 0) { s_ = sqrtf(((1.0f + Rk.r0.x) - Rk.r1.y) - Rk.r2.z); qx = 0.5f * s_; s_ = 0.5f / s_; qw = (Rk.r2.y - Rk.r1.z) * s_; qy = (Rk.r0.y + Rk.r1.x) * s_; qz = (Rk.r0.z + Rk.r2.x) * s_; }
            else if (ii == 1) { s_ = sqrtf(((1.0f + Rk.r1.y) - Rk.r0.x) - Rk.r2.z); qy = 0.5f * s_; s_ = 0.5f / s_; qw = (Rk.r0.z - Rk.r2.x) * s_; qx = (Rk.r0.y + Rk.r1.x) * s_; qz = (Rk.r1.z + Rk.r2.y) * s_; }
            else { s_ = sqrtf(((1.0f + Rk.r2.z) - Rk.r0.x) - Rk.r1.y); qz = 0.5f * s_; s_ = 0.5f / s_; qw = (Rk.r1.x - Rk.r0.y) * s_; qx = (Rk.r0.z + Rk.r2.x) * s_; qy = (Rk.r1.z + Rk.r2.y) * s_; }
        }
        po = add(po, scale(wk, add(add(m3_mulv(Rk, sub(pi, gk)), gk), tk)));
        bq[0] += wk * qx; bq[1] += wk * qy; bq[2] += wk * qz; bq[3] += wk * qw;
    }
    const float len = sqrtf(((bq[0] * bq[0] + bq[1] * bq[1]) + bq[2] * bq[2]) + bq[3] * bq[3]);
    const float inv = 1.0f / len;
#pragma unroll
    for (int a = 0; a < 4; a++) bq[a] *= inv;
    const M3 av = quat_to_rot_quirk(bq);
    const M3 avT = m3_transpose(av);
    st3(M.r0, i, row_mul(ld3(M.r0, i), avT));
    st3(M.r1, i, row_mul(ld3(M.r1, i), avT));
    st3(M.r2, i, row_mul(ld3(M.r2, i), avT));
    st6(M.shape, i, rot_sym(av, ld6(M.shape, i)));
    st3(M.pos, i, po);
}

// ---- peer-to-peer exchanges of the fuse stage (ssf_device.hpp: P2PView) ----------------------------------------------------
// shard sizes: one self-validating line per rank (Counters::last, checksum, sequence number) -> every peer; the lines of
// all ranks -> the host mailbox (as k_publish_all_counts).  One wave.
__global__ void k_p2p_counts(P2PView pv, const Counters* __restrict__ cnt, Mailbox* mb, unsigned long long all_seq) {
    const int l = lane();
    const unsigned long long seq = pv.seq;
    const int par = (int)(seq & 1ull);
    const unsigned int a = (unsigned int)cnt->last[0], b = (unsigned int)cnt->last[1], c = (unsigned int)cnt->last[2],
                       d = (unsigned int)cnt->last[3], e = (unsigned int)cnt->last[4];
    const unsigned long long w0 = a | ((unsigned long long)b << 32), w1 = c | ((unsigned long long)d << 32), w2 = e;
    const unsigned long long mine_word = l == 0 ? w0 : l == 1 ? w1 : l == 2 ? w2 : l == 6 ? w0 + w1 + w2 + seq : l == 7 ? seq : 0ull;
#pragma unroll
    for (int r = 0; r < SSF_P2P_MAX_RANKS; r++) {
        if (r >= pv.nranks || r == pv.me) continue;
        unsigned long long* dst = reinterpret_cast<unsigned long long*>(pv.peer[r] + p2p_off_cnt(par, pv.me));
        if (l < 8) __hip_atomic_store(&dst[l], mine_word, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    unsigned char* mine = p2p_peer(pv, pv.me);
    unsigned long long part = 0;
    P2PDeadline deadline(pv);
    for (int r = 0; r < pv.nranks; r++) {
        unsigned long long w = mine_word;
        if (r != pv.me) {
            const unsigned long long* src = reinterpret_cast<const unsigned long long*>(mine + p2p_off_cnt(par, r));
            for (;;) {
                w = l < 8 ? __hip_atomic_load(&src[l], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) : 0ull;
                const unsigned long long sum = (unsigned long long)wsum64(l < 3 ? (long long)w : 0ll) + seq;
                if (shfl_u64(w, 7) == seq && shfl_u64(w, 6) == sum) break;
                if (deadline.expired()) return;
            }
        }
        if (l < 3) {
            const int lo = (int)(unsigned int)w, hi = (int)(unsigned int)(w >> 32);
            __hip_atomic_store(&mb->all_cnt[5 * r + 2 * l], lo, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            part += (unsigned long long)(unsigned int)lo;
            if (l < 2) { __hip_atomic_store(&mb->all_cnt[5 * r + 2 * l + 1], hi, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM); part += (unsigned long long)(unsigned int)hi; }
        }
    }
    const unsigned long long check = (unsigned long long)wsum64((long long)part) + all_seq;
    if (l == 0) __hip_atomic_store(&mb->all_check, check, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (l == 0) __hip_atomic_store(&mb->all_seq, all_seq, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
// thread 0: every peer's flag line of `kind` carries pv.seq (its data stores were acknowledged before the flag left)
__device__ __forceinline__ bool p2p_wait_flags(const P2PView& pv, int kind) {
    unsigned char* mine = p2p_peer(pv, pv.me);
    const int par = (int)(pv.seq & 1ull);
    P2PDeadline deadline(pv);
    for (int r = 0; r < pv.nranks; r++) {
        if (r == pv.me) continue;
        const unsigned long long* f = reinterpret_cast<const unsigned long long*>(mine + p2p_off_flag(kind, par, r));
        while (__hip_atomic_load(f, __ATOMIC_ACQUIRE, __HIP_MEMORY_SCOPE_SYSTEM) != pv.seq)
            if (deadline.expired()) return false;
    }
    return true;
}
__device__ __forceinline__ void p2p_raise_flags(const P2PView& pv, int kind) {       // threads 0..nranks-1, after a barrier
    const int par = (int)(pv.seq & 1ull);
    const int r = threadIdx.x;
    if (r < pv.nranks && r != pv.me)
        __hip_atomic_store(reinterpret_cast<unsigned long long*>(p2p_peer(pv, r) + p2p_off_flag(kind, par, pv.me)), pv.seq,
                           __ATOMIC_RELEASE, __HIP_MEMORY_SCOPE_SYSTEM);
}
// association tables: this rank's best[] / matched[] -> every peer, then best := MIN, matched := OR over the ranks.
// One workgroup (S entries of 9 bytes).
__device__ __forceinline__ void p2p_assoc_exchange(const P2PView& pv, unsigned long long* __restrict__ best, uint8_t* __restrict__ matched, int* s_ok, Mailbox* mb) {
    const int par = (int)(pv.seq & 1ull), S = pv.S;
    // (own tables: device-scope atomic loads -- inside k_match they were written by other workgroups of this launch)
#pragma unroll
    for (int r = 0; r < SSF_P2P_MAX_RANKS; r++) {
        if (r >= pv.nranks || r == pv.me) continue;
        unsigned long long* db = reinterpret_cast<unsigned long long*>(pv.peer[r] + p2p_off_best(S, par, pv.me));
        uint8_t* dm = pv.peer[r] + p2p_off_matched(S, par, pv.me);
        // (returning exchanges, not stores: what comes back is the proof that the word has landed in the peer's memory
        // before this rank raises its flag -- see atomic_add_done)
        unsigned int* dm32 = reinterpret_cast<unsigned int*>(dm);
        const unsigned int* m32 = reinterpret_cast<const unsigned int*>(matched);
        unsigned long long sink = 0;
        for (int i = threadIdx.x; i < S; i += blockDim.x)
            sink ^= __hip_atomic_exchange(&db[i], __hip_atomic_load(&best[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        for (int i = threadIdx.x; i < (S + 3) / 4; i += blockDim.x)
            sink ^= __hip_atomic_exchange(&dm32[i], __hip_atomic_load(&m32[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        asm volatile("" :: "v"(sink));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    p2p_raise_flags(pv, P2P_FLAG_ASSOC);
    if (threadIdx.x == 0) {
        *s_ok = p2p_wait_flags(pv, P2P_FLAG_ASSOC) ? 1 : 0;
        if (!*s_ok) __hip_atomic_store(&mb->p2p_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (!*s_ok) return;
    unsigned char* mine = p2p_peer(pv, pv.me);
    for (int i = threadIdx.x; i < S; i += blockDim.x) {
        unsigned long long b = __hip_atomic_load(&best[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint8_t m = __hip_atomic_load(&matched[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        for (int r = 0; r < pv.nranks; r++) {
            if (r == pv.me) continue;
            const unsigned long long* sb = reinterpret_cast<const unsigned long long*>(mine + p2p_off_best(S, par, r));
            const uint8_t* sm = mine + p2p_off_matched(S, par, r);
            const unsigned long long ob = __hip_atomic_load(&sb[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            const uint8_t om = __hip_atomic_load(&sm[i], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            b = ob < b ? ob : b; m = om > m ? om : m;
        }
        best[i] = b; matched[i] = m;
    }
}
__global__ __launch_bounds__(1024) void k_p2p_assoc(P2PView pv, unsigned long long* __restrict__ best, uint8_t* __restrict__ matched, Mailbox* mb) {
    __shared__ int s_ok;
    p2p_assoc_exchange(pv, best, matched, &s_ok, mb);
}
// migrant table: slot f of this rank's table -> slot f of [parity][me] in every peer (an empty slot sends its first two
// words only); the last workgroup to finish raises the flags.  One thread per slot.
__global__ __launch_bounds__(256) void k_p2p_migr_share(P2PView pv, const int32_t* __restrict__ table, unsigned int* ticket) {
    const int par = (int)(pv.seq & 1ull), S = pv.S;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f < S) {
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(table + (size_t)SSF_MIGRANT_WORDS * f);
        unsigned long long w[SSF_MIGRANT_WORDS / 2];
        w[0] = src[0];
        const bool full = (unsigned int)w[0] != 0u;
#pragma unroll
        for (int k = 1; k < SSF_MIGRANT_WORDS / 2; k++) w[k] = full ? src[k] : 0ull;
#pragma unroll
        for (int r = 0; r < SSF_P2P_MAX_RANKS; r++) {
            if (r >= pv.nranks || r == pv.me) continue;
            unsigned long long* dst = reinterpret_cast<unsigned long long*>(pv.peer[r] + p2p_off_migr(S, par, pv.me)) + (size_t)(SSF_MIGRANT_WORDS / 2) * f;
            unsigned long long sink = __hip_atomic_exchange(&dst[0], w[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);     // (returning: see p2p_assoc_exchange)
            if (full) {
#pragma unroll
                for (int k = 1; k < SSF_MIGRANT_WORDS / 2; k++) sink ^= __hip_atomic_exchange(&dst[k], w[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            }
            asm volatile("" :: "v"(sink));
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = grid_arrive(ticket);
    __syncthreads();
    if (s_last) p2p_raise_flags(pv, P2P_FLAG_MIGR);
}
// ... and the peers' slots are added into this rank's table (at most one rank fills a slot: the sum is the union)
__global__ __launch_bounds__(256) void k_p2p_migr_gather(P2PView pv, int32_t* __restrict__ table, Mailbox* mb) {
    const int par = (int)(pv.seq & 1ull), S = pv.S;
    __shared__ int s_ok;
    if (threadIdx.x == 0) {
        s_ok = p2p_wait_flags(pv, P2P_FLAG_MIGR) ? 1 : 0;
        if (!s_ok) __hip_atomic_store(&mb->p2p_timeout, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
    }
    __syncthreads();
    if (!s_ok) return;
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= S) return;
    unsigned char* mine = p2p_peer(pv, pv.me);
    int32_t* own = table + (size_t)SSF_MIGRANT_WORDS * f;
    for (int r = 0; r < pv.nranks; r++) {
        if (r == pv.me) continue;
        const unsigned long long* src = reinterpret_cast<const unsigned long long*>(mine + p2p_off_migr(S, par, r)) + (size_t)(SSF_MIGRANT_WORDS / 2) * f;
        const unsigned long long w0 = __hip_atomic_load(&src[0], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
        if ((unsigned int)w0 == 0u) continue;
        own[0] += (int32_t)(unsigned int)w0; own[1] += (int32_t)(unsigned int)(w0 >> 32);
#pragma unroll
        for (int k = 1; k < SSF_MIGRANT_WORDS / 2; k++) {
            const unsigned long long w = __hip_atomic_load(&src[k], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
            own[2 * k] += (int32_t)(unsigned int)w; own[2 * k + 1] += (int32_t)(unsigned int)(w >> 32);
        }
    }
}

// ---- launchers -----------------------------------------------------------------------------------
// measurement arms of k_icp (lab build, environment read once): 0 = the product's form; only that one can be told to do the
// association (SSF_ICP_GO_MATCH) -- the host asks before it sends that word
int icp_variant_mode() {
    static const int mode = SSF_ENV_INT("ICP_WRED", 0) ? 2 : (SSF_ENV_INT("ICP_PER_LANE", 0) > 1 ? 1 : 0);
    return mode;
}
void launch_icp(hipStream_t st, const Cam& cam, SurfelSoA model, int n_visible, const uint2* pix2, const float4* fpack,
                Rt T, long long* replicas, unsigned int* ticket,
                long long* sums29, Mailbox* mb, unsigned long long seq, int dbg_arg, IcpGo* go, unsigned long long go_seq,
                const P2PView* pv, int by_tile, const MatchArgs* match) {
    ScopedKernel sk("icp_accumulate", st);
    const MatchArgs ma = match ? *match : MatchArgs{0.f, 0.f, 0, nullptr, nullptr, nullptr};
    // one row per thread, <= 4096 workgroups (grid-stride beyond).  (lab: SSF_ICP_PER_LANE=n rows per thread with register sums,
    // SSF_ICP_WRED=1 the DPP row reduction -- lab/icp_arms.inc, both measured slower: DESIGN.md section 4.3)
    const int mode = dbg_arg > 0 && dbg_arg != 4 ? 0 : icp_variant_mode();      // (the probe switches live in the one-row-per-thread form)
    const int per_lane = mode == 1 ? std::max(2, SSF_ENV_INT("ICP_PER_LANE", 0)) : 1;
    const int per_block = 256 * per_lane;
    int grid = (n_visible + per_block - 1) / per_block;
    if (grid < 1) grid = 1;              // an empty shard still publishes its (zero) record
    if (grid > 4096) grid = 4096;
    // the counted record (one trip at the end of the launch instead of three): single GPU, no probe switches
    static const int counted = SSF_ENV_INT("ICP_COUNTED", 1);
    const int dbg = (dbg_arg < 0 ? 0 : dbg_arg) | ((counted && !pv && dbg_arg < 0) ? SSF_ICP_DBG_COUNTED : 0);
    const P2PView none{};
    const P2PView& v = pv ? *pv : none;
    if (by_tile && !pv) { hipLaunchKernelGGL((k_icp<false, 3>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma); return; }
#ifdef SSF_EXPERIMENTS
    if (mode == 2 && !pv) { hipLaunchKernelGGL((k_icp<false, 2>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma); return; }
    if (mode == 1 && pv) { hipLaunchKernelGGL((k_icp<true, 1>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma); return; }
    if (mode == 1) { hipLaunchKernelGGL((k_icp<false, 1>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma); return; }
#endif
    if (pv) hipLaunchKernelGGL((k_icp<true, 0>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma);
    else hipLaunchKernelGGL((k_icp<false, 0>), dim3(grid), dim3(256), 0, st, cam, model, n_visible, pix2, fpack, T, replicas, ticket, sums29, mb, seq, dbg, go, go_seq, v, by_tile, ma);
}
void launch_match(hipStream_t st, const Cam& cam, SurfelSoA model, int n_visible, const uint2* pix2, const float4* fpack,
                  Rt pose, float zmin, float zmax, long long id_offset,
                  unsigned long long* best, uint8_t* matched, int32_t* cand, int S, const int32_t* orig) {
    (void)S;                                   // best/matched were initialised by k_finalize_surfels of this frame
    if (n_visible <= 0) return;
    ScopedKernel sk("match", st);
    if (orig) hipLaunchKernelGGL(k_match<true>, dim3((n_visible + 255) / 256), dim3(256), 0, st, cam, model, n_visible, pix2, fpack,
                                 pose, zmin, zmax, id_offset, best, matched, cand, orig);
    else hipLaunchKernelGGL(k_match<false>, dim3((n_visible + 255) / 256), dim3(256), 0, st, cam, model, n_visible, pix2, fpack,
                            pose, zmin, zmax, id_offset, best, matched, cand, orig);
}
void launch_fuse(hipStream_t st, SurfelSoA model, SurfelSoA frame, Rt pose, int stamp, long long id_offset,
                 int n_visible, const unsigned long long* best, const uint8_t* matched, const int32_t* cand, int S, int do_update,
                 int capacity, int rank, int nranks, float tile, Counters* cnt, const Cam& cam, OovStore oov,
                 int span_upper, const float* plane_depth, int delta_t, float conf_thresh, float zmin, float zmax,
                 uint8_t* state_vis, uint8_t* state_oov, uint32_t* bc_oov, const PartitionWs& ws, int migrate, int tail_in_move) {
    ScopedKernel sk("update_insert", st);
    const int nchunks = (S + 255) / 256, nb_oov = (span_upper + 255) / 256, nb_vis = (n_visible + 255) / 256;
    ClassifyArgs ca; ca.cam = cam; ca.plane_depth = plane_depth; ca.delta_t = delta_t; ca.conf_thresh = conf_thresh; ca.zmin = zmin; ca.zmax = zmax;
    const int nupd = (S + UPD_PER_WG - 1) / UPD_PER_WG;
    hipLaunchKernelGGL(k_update_insert, dim3(nupd + nchunks + nb_vis + (nb_oov + OOV_PER_WG - 1) / OOV_PER_WG), dim3(256), 0, st, model,
                       frame, pose, stamp, id_offset,
                       n_visible, best, matched, cand, S, do_update, capacity, rank, nranks, tile, cnt, nupd, nchunks, nb_vis, nb_oov, oov, ca,
                       state_vis, state_oov, bc_oov, ws, migrate, tail_in_move);
}
void launch_pack_emigrants(hipStream_t st, SurfelSoA model, const unsigned long long* best, const uint8_t* matched, long long id_offset,
                           int n_visible, const uint8_t* state_vis, int S, int do_update, int nranks, float tile, int32_t* table) {
    ScopedKernel sk("pack_emigrants", st);
    hipLaunchKernelGGL(k_pack_emigrants, dim3((S + 255) / 256), dim3(256), 0, st, model, best, matched, id_offset, n_visible, state_vis, S,
                       do_update, nranks, tile, table);
}
void launch_migrate_in(hipStream_t st, SurfelSoA model, const int32_t* table, int S, int rank, int capacity, Counters* cnt,
                       const Cam& cam, Rt pose, int stamp, const float* plane_depth, int delta_t, float conf_thresh, float zmin,
                       float zmax, uint8_t* state_vis, const PartitionWs& ws) {
    ScopedKernel sk("migrate_in", st);
    ClassifyArgs ca; ca.cam = cam; ca.plane_depth = plane_depth; ca.delta_t = delta_t; ca.conf_thresh = conf_thresh; ca.zmin = zmin; ca.zmax = zmax;
    hipLaunchKernelGGL(k_migrate_in, dim3(1), dim3(1024), 0, st, model, table, S, rank, capacity, cnt, ca, pose, stamp, state_vis, ws);
}
void launch_first_frame(hipStream_t st, SurfelSoA model, SurfelSoA frame, Rt pose, int S, int capacity, int rank,
                        int nranks, float tile, Counters* cnt) {
    ScopedKernel sk("first_frame", st);
    hipLaunchKernelGGL(k_first_frame, dim3(1), dim3(1024), 0, st, model, frame, pose, S, capacity, rank, nranks, tile, cnt);
}
void launch_move_rows(hipStream_t st, const Cam& cam, SurfelSoA vis_src, SurfelSoA vis_dst, OovStore oov, int nv_upper, int span_upper,
                      const uint8_t* state_vis, const uint8_t* state_oov, const uint32_t* bc_oov, const PartitionWs& ws,
                      Counters* cnt, Mailbox* mb, unsigned long long cnt_seq, const NextFrameIcp* next, const MoveTotals* totals) {
    const int nb_vis = std::max(1, (nv_upper + 255) / 256), nb_oov = (span_upper + 255) / 256;
    const MoveTotals mt = totals ? *totals : MoveTotals{0, 0, 0, 0};
    NextIcp nx{};
    const dim3 grid(nb_vis + nb_oov);
    if (next) {
        nx.cam = cam; nx.pix2 = next->pix2; nx.fpack = next->fpack; nx.T = next->T; nx.replicas = next->replicas;
        nx.ticket = next->ticket; nx.sums = next->sums; nx.mb = mb; nx.seq = next->seq;
        ScopedKernel sk("reorder_move_icp", st);
        if (next->pv)
            hipLaunchKernelGGL((k_move_rows<true, true>), grid, dim3(256), 0, st, vis_src, vis_dst, oov, state_vis, state_oov,
                               bc_oov, ws, cnt, nb_vis, nx, mb, cnt_seq, *next->pv, mt);
        else
            hipLaunchKernelGGL((k_move_rows<true, false>), grid, dim3(256), 0, st, vis_src, vis_dst, oov, state_vis, state_oov,
                               bc_oov, ws, cnt, nb_vis, nx, mb, cnt_seq, P2PView{}, mt);
    } else {
        ScopedKernel sk("reorder_move", st);
        hipLaunchKernelGGL((k_move_rows<false, false>), grid, dim3(256), 0, st, vis_src, vis_dst, oov, state_vis, state_oov,
                           bc_oov, ws, cnt, nb_vis, nx, mb, cnt_seq, P2PView{}, mt);
    }
}
void launch_oov_compact(hipStream_t st, OovStore src, OovStore dst, int span_upper, int new_head, uint32_t* bc_oov, Counters* cnt,
                        int set_span) {
    const int nb = (span_upper + 255) / 256;
    ScopedKernel sk("oov_compact", st);
    if (nb > 0) {
        hipLaunchKernelGGL(k_oov_count, dim3(nb), dim3(256), 0, st, src, bc_oov, cnt);
        hipLaunchKernelGGL(k_oov_scan, dim3(1), dim3(1024), 0, st, bc_oov, nb, cnt);
        hipLaunchKernelGGL(k_oov_compact, dim3(nb), dim3(256), 0, st, src, dst, bc_oov, new_head, cnt);
    }
    if (set_span) hipLaunchKernelGGL(k_oov_set_span, dim3(1), dim3(1), 0, st, cnt, new_head);
}
void launch_align(hipStream_t st, const Cam& cam, const float* spos, const float* slab, const float* snrm, const float* sconf,
                  int n, SurfelSoA frame, const int32_t* label, const float* plane_depth, Rt T, long long* out40) {
    ScopedKernel sk("align_iteration", st);
    hipLaunchKernelGGL(k_align, dim3(1), dim3(1024), 0, st, cam, spos, slab, snrm, sconf, n, frame, label, plane_depth, T, out40);
}
void launch_fern_codes(hipStream_t st, const uint8_t* rgb, const float* depth, int W, int H, const uint32_t* fpos,
                       const uint8_t* frgb, const float* fdepth, int n, uint8_t* codes) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_fern_codes, dim3((n + 63) / 64), dim3(64), 0, st, rgb, depth, W, H, fpos, frgb, fdepth, n, codes);
}
void launch_p2p_counts(hipStream_t st, const P2PView& pv, const Counters* cnt, Mailbox* mb, unsigned long long all_seq) {
    ScopedKernel sk("p2p_counts", st);
    hipLaunchKernelGGL(k_p2p_counts, dim3(1), dim3(64), 0, st, pv, cnt, mb, all_seq);
}
void launch_p2p_assoc(hipStream_t st, const P2PView& pv, unsigned long long* best, uint8_t* matched, Mailbox* mb) {
    ScopedKernel sk("p2p_assoc", st);
    hipLaunchKernelGGL(k_p2p_assoc, dim3(1), dim3(1024), 0, st, pv, best, matched, mb);
}
void launch_p2p_migrants(hipStream_t st, const P2PView& pv, int32_t* table, unsigned int* ticket, Mailbox* mb) {
    ScopedKernel sk("p2p_migrants", st);
    const int grid = (pv.S + 255) / 256;
    hipLaunchKernelGGL(k_p2p_migr_share, dim3(grid), dim3(256), 0, st, pv, table, ticket);
    hipLaunchKernelGGL(k_p2p_migr_gather, dim3(grid), dim3(256), 0, st, pv, table, mb);
}
void launch_publish_icp(hipStream_t st, const long long* rec, Mailbox* mb, unsigned long long seq) {
    hipLaunchKernelGGL(k_publish_icp, dim3(1), dim3(64), 0, st, rec, mb, seq);
}
void launch_publish_all_counts(hipStream_t st, const int* all5, int nranks, Mailbox* mb, unsigned long long seq) {
    hipLaunchKernelGGL(k_publish_all_counts, dim3(1), dim3(64), 0, st, all5, 5 * nranks, mb, seq);
}
void launch_publish_counts(hipStream_t st, Counters* cnt, int shrink_by_removed, Mailbox* mb, unsigned long long seq) {
    hipLaunchKernelGGL(k_publish_counts, dim3(1), dim3(1), 0, st, cnt, shrink_by_removed, mb, seq);
}
void launch_lab_refresh(hipStream_t st, SurfelSoA s, int n) {
    if (n <= 0) return;
    hipLaunchKernelGGL(k_lab_refresh, dim3((n + 255) / 256), dim3(256), 0, st, s, n);
}
void launch_deformation(hipStream_t st, SurfelSoA model, int n, int m, const float* npos, const float* nrot,
                        const float* ntrans, float* nodes16, const float* w4, const int32_t* idx4) {
    if (n <= 0) return;
    ScopedKernel sk("apply_deformation", st);
    hipLaunchKernelGGL(k_pack_nodes, dim3((m + 255) / 256), dim3(256), 0, st, m, npos, nrot, ntrans, reinterpret_cast<float4*>(nodes16));
    hipLaunchKernelGGL(k_deformation, dim3((n + 255) / 256), dim3(256), 0, st, model, n, reinterpret_cast<const float4*>(nodes16), w4, idx4);
}

}  // namespace ssf
