// Can the host store directly into device memory (fine-grained allocation, large BAR), and how soon does a spinning
// kernel see it?  hipcc --offload-arch=gfx950 -O2 tools/probe/bar_write.hip -o /tmp/bar_write && /tmp/bar_write
#include <hip/hip_runtime.h>
#include <chrono>
#include <cstdio>
#include <thread>
__global__ void k_wait(volatile unsigned long long* flag, volatile unsigned long long* echo_host, unsigned long long want) {
    while (__hip_atomic_load((unsigned long long*)flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM) != want) __builtin_amdgcn_s_sleep(1);
    __hip_atomic_store((unsigned long long*)echo_host, want, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}
int main() {
    unsigned long long* d = nullptr; unsigned long long* echo = nullptr; unsigned long long* echo_dev = nullptr;
    hipError_t e = hipExtMallocWithFlags((void**)&d, 4096, hipDeviceMallocFinegrained);
    printf("hipExtMallocWithFlags(finegrained): %s\n", hipGetErrorString(e));
    if (e != hipSuccess) return 1;
    hipHostMalloc((void**)&echo, 4096, hipHostMallocCoherent); hipHostGetDevicePointer((void**)&echo_dev, echo, 0);
    hipMemset(d, 0, 4096); hipDeviceSynchronize();
    hipPointerAttribute_t attr; e = hipPointerGetAttributes(&attr, d);
    printf("attr: type %d hostPointer %p devicePointer %p\n", (int)attr.type, attr.hostPointer, attr.devicePointer);
    hipStream_t st; hipStreamCreate(&st);
    double sum = 0; int n = 0;
    for (unsigned long long it = 1; it <= 200; it++) {
        echo[0] = 0;
        hipLaunchKernelGGL(k_wait, dim3(1), dim3(64), 0, st, d, echo_dev, it);
        std::this_thread::sleep_for(std::chrono::microseconds(200));      // the kernel is resident and spinning
        auto t0 = std::chrono::steady_clock::now();
        *(volatile unsigned long long*)d = it;                              // direct host store into device memory
        __builtin_ia32_sfence();                                            // (the mapping is write-combining: push it out)
        while (*(volatile unsigned long long*)echo != it) { if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > 2.0) { printf("timeout at %llu\n", it); return 2; } }
        double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
        if (it > 20) { sum += us; n++; }
        hipStreamSynchronize(st);
    }
    printf("host store -> kernel sees it -> host sees the echo: %.2f us round trip (mean of %d)\n", sum / n, n);
    return 0;
}
