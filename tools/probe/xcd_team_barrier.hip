// What does a barrier among the workgroups of ONE XCD cost when it never leaves that XCD's L2?  (a resident
// relabelling form whose exchange is confined to one XCD -- a frame per XCD -- is the untried variant; round 3's resident form lost
// to three dependent trips to the device-wide coherence point per pass, docs/HISTORY.md section 4.1.1.)
//
// Persistent workgroups form TEAMS by the hardware's XCC_ID (not by an assumption about blockIdx -> XCD), then meet R times:
//   A  flags : every workgroup stores its own flag word (plain store: it lands in the XCD's L2), lanes poll the team's flags with
//              sc1 loads (served by that L2, bypassing the CU's L1)
//   B  team counter : one agent-scope atomic add per workgroup on the TEAM's counter, sc1 polls
//   C  device counter : the same on ONE counter for the whole grid (the round-3 form)
// and, for A, check that 256 bytes of plain-stored payload per workgroup are seen by the next team member after the barrier
// (sc1 loads): stale words are counted.  Every wait is bounded (a lost wake-up ends the run with a count, never a hang).
//   hipcc --offload-arch=gfx950 -O2 tools/probe/xcd_team_barrier.hip -o /tmp/xcd_team_barrier && /tmp/xcd_team_barrier
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

#define MAX_TEAM 128
__device__ __forceinline__ unsigned int xcc_id() { return __builtin_amdgcn_s_getreg((3 << 11) | (0 << 6) | 20) & 7u; }
__device__ __forceinline__ unsigned int ld_sc1(const unsigned int* p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

struct Shared {
    unsigned int census[8];              // workgroups per XCD (agent atomics)
    unsigned int arrived;                // one-time start barrier
    unsigned int team_counter[8 * 32];   // B: one counter per XCD, a cache line apart
    unsigned int device_counter;         // C
    unsigned int failures, stale;
};

template <int MODE>
__global__ __launch_bounds__(256) void k_meet(Shared* sh, unsigned int* flags /* 8 x MAX_TEAM */, unsigned int* payload /* 8 x MAX_TEAM x 64 */,
                                             int rounds, unsigned long long* cycles_out) {
    __shared__ unsigned int s_idx, s_team, s_fail;
    const unsigned int x = xcc_id();
    if (threadIdx.x == 0) {
        s_idx = atomicAdd(&sh->census[x], 1u); s_fail = 0;
        atomicAdd(&sh->arrived, 1u);
        unsigned int spins = 0;
        while (ld_sc1(&sh->arrived) < gridDim.x && ++spins < (1u << 22)) __builtin_amdgcn_s_sleep(1);
        if (spins >= (1u << 22)) s_fail = 1;
        s_team = ld_sc1(&sh->census[x]);
    }
    __syncthreads();
    const unsigned int idx = s_idx, team = s_team;
    if (s_fail || idx >= MAX_TEAM || team > MAX_TEAM) { if (threadIdx.x == 0) atomicAdd(&sh->failures, 1u); return; }
    unsigned int* my_flags = flags + x * MAX_TEAM;
    unsigned int* my_payload = payload + (size_t)(x * MAX_TEAM) * 64;
    unsigned int stale = 0, failed = 0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int r = 1; r <= rounds && !failed; r++) {
        if (MODE == 0 || MODE >= 3) {
            if (threadIdx.x < 64) my_payload[idx * 64 + threadIdx.x] = (unsigned int)r * 1000u + threadIdx.x;      // plain stores
            __builtin_amdgcn_s_waitcnt(0);                                                                          // ... acknowledged by the L2
            __syncthreads();
            if (threadIdx.x == 0) { my_flags[idx] = (unsigned int)r; }
            if (threadIdx.x < 64) {                                   // wave 0 polls the team's flags, two per lane
                unsigned int spins = 0;
                for (;;) {
                    const unsigned int a = threadIdx.x < team ? ld_sc1(&my_flags[threadIdx.x]) : 0xFFFFFFFFu;
                    const unsigned int b = threadIdx.x + 64 < team ? ld_sc1(&my_flags[threadIdx.x + 64]) : 0xFFFFFFFFu;
                    if (__ballot(a < (unsigned int)r || b < (unsigned int)r) == 0ull) break;
                    if (++spins > (1u << 20)) { failed = 1; break; }
                }
            }
            __syncthreads();
            if (MODE == 3) asm volatile("buffer_inv sc0\n\ts_dcache_inv" ::: "memory");        // workgroup scope
            if (MODE == 5) asm volatile("buffer_inv sc1\n\ts_dcache_inv" ::: "memory");        // agent scope: the CU's vector L1 (and what of the L2?)
            if (MODE == 6) asm volatile("buffer_inv sc0 sc1\n\ts_dcache_inv" ::: "memory");    // system scope
            if (threadIdx.x < 64) {                                   // the next member's payload, through the L2
                const unsigned int n = (idx + 1) % team;
                const unsigned int* q = &my_payload[n * 64 + threadIdx.x];
                asm volatile("" : "+v"(q));                           // (3 / 4: a PLAIN load -- no cache-policy bits -- that the compiler cannot hoist; `volatile` would make it sc0 sc1)
                const unsigned int v = MODE == 0 ? ld_sc1(q) : *q;
                if (v != (unsigned int)r * 1000u + threadIdx.x) stale++;
            }
        } else {
            unsigned int* ctr = MODE == 1 ? &sh->team_counter[x * 32] : &sh->device_counter;
            const unsigned int want = (MODE == 1 ? team : gridDim.x) * (unsigned int)r;
            __syncthreads();
            if (threadIdx.x == 0) {
                atomicAdd(ctr, 1u);
                unsigned int spins = 0;
                while (ld_sc1(ctr) < want) { if (++spins > (1u << 20)) { failed = 1; break; } __builtin_amdgcn_s_sleep(1); }
            }
            __syncthreads();
        }
        failed = __syncthreads_or((int)failed);
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    if (threadIdx.x == 0) { cycles_out[blockIdx.x] = t1 - t0; if (failed) atomicAdd(&sh->failures, 1u); }
    if (stale) atomicAdd(&sh->stale, stale);
}

int main() {
    Shared* sh; unsigned int *flags, *payload; unsigned long long* cyc;
    hipMalloc(&sh, sizeof(Shared)); hipMalloc(&flags, 8 * MAX_TEAM * 4); hipMalloc(&payload, (size_t)8 * MAX_TEAM * 64 * 4); hipMalloc(&cyc, 1024 * 8);
    const int rounds = 2000;
    for (int wgs : {256, 512}) {
        for (int mode = 0; mode < 7; mode++) {
            hipMemset(sh, 0, sizeof(Shared)); hipMemset(flags, 0, 8 * MAX_TEAM * 4); hipMemset(payload, 0, (size_t)8 * MAX_TEAM * 64 * 4);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a, 0);
            if (mode == 0) hipLaunchKernelGGL(k_meet<0>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else if (mode == 1) hipLaunchKernelGGL(k_meet<1>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else if (mode == 2) hipLaunchKernelGGL(k_meet<2>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else if (mode == 3) hipLaunchKernelGGL(k_meet<3>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else if (mode == 4) hipLaunchKernelGGL(k_meet<4>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else if (mode == 5) hipLaunchKernelGGL(k_meet<5>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            else hipLaunchKernelGGL(k_meet<6>, dim3(wgs), dim3(256), 0, 0, sh, flags, payload, rounds, cyc);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms = 0; hipEventElapsedTime(&ms, a, b);
            Shared h; hipMemcpy(&h, sh, sizeof(Shared), hipMemcpyDeviceToHost);
            const char* name[7] = {"A flags + sc1 polls (team = XCD) ", "B team counter (agent atomics)     ", "C one device-wide counter          ",
                                   "A + buffer_inv sc0, PLAIN loads   ", "A, PLAIN loads, NO invalidate     ", "A + buffer_inv sc1, PLAIN loads   ",
                                   "A + buffer_inv sc0 sc1, PLAIN     "};
            printf("%4d workgroups  %s  %.3f us per meeting  (teams %u %u %u %u %u %u %u %u; failures %u, stale payload words %u of %llu)\n", wgs, name[mode],
                   1000.0 * ms / rounds, h.census[0], h.census[1], h.census[2], h.census[3], h.census[4], h.census[5], h.census[6], h.census[7], h.failures,
                   h.stale, (mode == 0 || mode >= 3) ? (unsigned long long)wgs * 64ull * rounds : 0ull);
        }
    }
    return 0;
}
