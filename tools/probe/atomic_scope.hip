// What does a global integer atomic cost on MI355X by SCOPE, for the association's pattern (k_match: N rows, each one
// 64-bit atomicMin onto one of S hot words chosen by a gather) and for a histogram (N 32-bit adds onto S words)?
//   agent scope   (HIP's atomicMin / atomicAdd: `global_atomic_* ... sc1`) is performed at the memory side -- the eight XCDs'
//                 L2s are not coherent with each other, so that is the only place all CUs agree on;
//   workgroup scope (no sc bits) is performed in the issuing XCD's OWN L2.  All CUs of one XCD share that L2, so the
//                 operation is atomic among them; it says nothing to the other seven XCDs.  With one replica table per
//                 XCD -- selected by the hardware's XCC_ID, not by an assumption about blockIdx -> XCD -- every word is only
//                 ever touched through ONE L2, and the kernel boundary writes the replicas back for a merge (MIN / SUM over 8).
// Also: reading the word first (agent-scope load) and skipping an atomic that cannot win.
// Checks every variant against the host's result (an atomicity failure among the CUs of an XCD would show as a wrong
// minimum / a short count), then times them.
//   hipcc --offload-arch=gfx950 -O2 tools/probe/atomic_scope.hip -o /tmp/atomic_scope && /tmp/atomic_scope
#include <hip/hip_runtime.h>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

#define NO_MATCH 0x7fffffffffffffffull
__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }   // HW_REG_XCC_ID[3:0]

__global__ void k_min_agent(const uint32_t* tgt, const uint32_t* dist, int n, unsigned long long* best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    atomicMin(&best[tgt[i]], ((unsigned long long)dist[i] << 32) | (uint32_t)i);
}
__global__ void k_min_agent_filter(const uint32_t* tgt, const uint32_t* dist, int n, unsigned long long* best) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const unsigned long long key = ((unsigned long long)dist[i] << 32) | (uint32_t)i;
    if (__hip_atomic_load(&best[tgt[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > key) atomicMin(&best[tgt[i]], key);
}
template <bool FILTER>
__global__ void k_min_xcd(const uint32_t* tgt, const uint32_t* dist, int n, unsigned long long* rep, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    unsigned long long* mine = rep + (size_t)xcc_id() * S;
    const unsigned long long key = ((unsigned long long)dist[i] << 32) | (uint32_t)i;
    // (the filter's load must not come from the CU's L1, which no other CU's atomic refreshes: sc1 = served by the L2)
    if (FILTER && !(__hip_atomic_load(&mine[tgt[i]], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) > key)) return;
    (void)__hip_atomic_fetch_min(&mine[tgt[i]], key, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_merge_min(const unsigned long long* rep, unsigned long long* best, int S) {
    const int f = blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= S) return;
    unsigned long long v = rep[f];
#pragma unroll
    for (int x = 1; x < 8; x++) v = min(v, rep[(size_t)x * S + f]);
    best[f] = v;
}
__global__ void k_fill(unsigned long long* p, unsigned long long v, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) p[i] = v;
}
__global__ void k_add_agent(const uint32_t* tgt, int n, uint32_t* cnt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) atomicAdd(&cnt[tgt[i]], 1u);
}
__global__ void k_add_xcd(const uint32_t* tgt, int n, uint32_t* rep, int S) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) (void)__hip_atomic_fetch_add(&rep[(size_t)xcc_id() * S + tgt[i]], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
}
__global__ void k_census(uint32_t* per_xcc) { if (threadIdx.x == 0) atomicAdd(&per_xcc[xcc_id()], 1u); }
// the rows' own loads alone (what every variant pays before its atomic)
__global__ void k_loads_only(const uint32_t* tgt, const uint32_t* dist, int n, unsigned long long* sink) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    if (tgt[i] == 0xffffffffu && dist[i] == 7u) sink[0] = 1;
}

template <typename F> static double time_us(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 3; i++) f();
    hipDeviceSynchronize();
    hipEventRecord(a, 0);
    for (int i = 0; i < reps; i++) f();
    hipEventRecord(b, 0); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0 / reps;
}

int main() {
    const int N = 1000000;
    uint32_t st = 12345u;
    auto rnd = [&]() { st = st * 1664525u + 1013904223u; return st >> 8; };
    uint32_t *d_tgt, *d_dist, *d_cnt, *d_cnt_rep, *d_census; unsigned long long *d_best, *d_rep;
    hipMalloc(&d_tgt, N * 4); hipMalloc(&d_dist, N * 4);
    hipMalloc(&d_best, 8192 * 8); hipMalloc(&d_rep, 8 * 8192 * 8); hipMalloc(&d_cnt, 8192 * 4); hipMalloc(&d_cnt_rep, 8 * 8192 * 4);
    hipMalloc(&d_census, 32); hipMemset(d_census, 0, 32);
    hipLaunchKernelGGL(k_census, dim3(4096), dim3(256), 0, 0, d_census);
    uint32_t census[8]; hipMemcpy(census, d_census, 32, hipMemcpyDeviceToHost);
    printf("workgroups per XCC_ID of a 4096-block launch:");
    for (int x = 0; x < 8; x++) printf(" %u", census[x]);
    printf("\n");
    const dim3 grid((N + 255) / 256), blk(256);
    int bad = 0;
    for (int S : {1200, 4800}) {
        for (int sorted = 0; sorted < 2; sorted++) {
            std::vector<uint32_t> tgt(N), dist(N);
            for (int i = 0; i < N; i++) { tgt[i] = rnd() % S; dist[i] = 0x3c000000u + (rnd() & 0xfffff); }
            if (sorted) std::sort(tgt.begin(), tgt.end());          // rows in image order: neighbours bid for the same word
            std::vector<unsigned long long> want(S, NO_MATCH); std::vector<uint32_t> wcnt(S, 0);
            for (int i = 0; i < N; i++) { want[tgt[i]] = std::min(want[tgt[i]], ((unsigned long long)dist[i] << 32) | (uint32_t)i); wcnt[tgt[i]]++; }
            hipMemcpy(d_tgt, tgt.data(), N * 4, hipMemcpyHostToDevice); hipMemcpy(d_dist, dist.data(), N * 4, hipMemcpyHostToDevice);
            std::vector<unsigned long long> got(S); std::vector<uint32_t> gcnt(S), grep(8 * S);
            auto check = [&](const char* what) {
                hipMemcpy(got.data(), d_best, S * 8, hipMemcpyDeviceToHost);
                int wrong = 0;
                for (int f = 0; f < S; f++) wrong += got[f] != want[f];
                if (wrong) { printf("  !! %s: %d of %d minima wrong\n", what, wrong, S); bad++; }
            };
            auto fill_best = [&] { hipLaunchKernelGGL(k_fill, dim3((S + 255) / 256), blk, 0, 0, d_best, NO_MATCH, S); };
            auto fill_rep = [&] { hipLaunchKernelGGL(k_fill, dim3((8 * S + 255) / 256), blk, 0, 0, d_rep, NO_MATCH, 8 * S); };
            auto merge = [&] { hipLaunchKernelGGL(k_merge_min, dim3((S + 255) / 256), blk, 0, 0, d_rep, d_best, S); };
            // correctness of every variant, 5 runs each
            for (int r = 0; r < 5; r++) {
                fill_best(); hipLaunchKernelGGL(k_min_agent, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); check("agent");
                fill_best(); hipLaunchKernelGGL(k_min_agent_filter, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); check("agent+filter");
                fill_rep(); hipLaunchKernelGGL(k_min_xcd<false>, grid, blk, 0, 0, d_tgt, d_dist, N, d_rep, S); merge(); check("xcd");
                fill_rep(); hipLaunchKernelGGL(k_min_xcd<true>, grid, blk, 0, 0, d_tgt, d_dist, N, d_rep, S); merge(); check("xcd+filter");
                hipMemset(d_cnt, 0, S * 4); hipLaunchKernelGGL(k_add_agent, grid, blk, 0, 0, d_tgt, N, d_cnt);
                hipMemcpy(gcnt.data(), d_cnt, S * 4, hipMemcpyDeviceToHost);
                int wrong = 0; for (int f = 0; f < S; f++) wrong += gcnt[f] != wcnt[f];
                if (wrong) { printf("  !! add agent: %d counts wrong\n", wrong); bad++; }
                hipMemset(d_cnt_rep, 0, 8 * S * 4); hipLaunchKernelGGL(k_add_xcd, grid, blk, 0, 0, d_tgt, N, d_cnt_rep, S);
                hipMemcpy(grep.data(), d_cnt_rep, 8 * S * 4, hipMemcpyDeviceToHost);
                wrong = 0; for (int f = 0; f < S; f++) { uint32_t t = 0; for (int x = 0; x < 8; x++) t += grep[(size_t)x * S + f]; wrong += t != wcnt[f]; }
                if (wrong) { printf("  !! add xcd: %d counts wrong (an add lost between CUs of one XCD)\n", wrong); bad++; }
            }
            const int reps = 20;
            printf("S=%d rows=%d %s\n", S, N, sorted ? "SORTED by target" : "random targets");
            printf("  loads only                               %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_loads_only, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); }, reps));
            printf("  atomicMin u64 agent scope                %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_min_agent, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); }, reps));
            printf("  (fill + that)                            %7.2f us\n", time_us([&] { fill_best(); hipLaunchKernelGGL(k_min_agent, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); }, reps));
            printf("  fill + read first, agent atomic if less  %7.2f us\n", time_us([&] { fill_best(); hipLaunchKernelGGL(k_min_agent_filter, grid, blk, 0, 0, d_tgt, d_dist, N, d_best); }, reps));
            printf("  fill + per-XCD replicas (L2) + merge     %7.2f us\n", time_us([&] { fill_rep(); hipLaunchKernelGGL(k_min_xcd<false>, grid, blk, 0, 0, d_tgt, d_dist, N, d_rep, S); merge(); }, reps));
            printf("  fill + per-XCD replicas, read first      %7.2f us\n", time_us([&] { fill_rep(); hipLaunchKernelGGL(k_min_xcd<true>, grid, blk, 0, 0, d_tgt, d_dist, N, d_rep, S); merge(); }, reps));
            printf("  per-XCD replicas kernel alone (no fill)  %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_min_xcd<false>, grid, blk, 0, 0, d_tgt, d_dist, N, d_rep, S); }, reps));
            printf("  atomicAdd u32 agent scope                %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_add_agent, grid, blk, 0, 0, d_tgt, N, d_cnt); }, reps));
            printf("  atomicAdd u32 per-XCD replicas (L2)      %7.2f us\n", time_us([&] { hipLaunchKernelGGL(k_add_xcd, grid, blk, 0, 0, d_tgt, N, d_cnt_rep, S); }, reps));
        }
    }
    printf(bad ? "FAILED: %d wrong results\n" : "all variants exact (%d)\n", bad);
    return bad ? 1 : 0;
}
