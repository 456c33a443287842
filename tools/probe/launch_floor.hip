// What does a launch of the relabelling pass' SHAPE cost with nothing in it?  Back-to-back launches of kernels that do (a)
// nothing, (b) one kernarg-dependent load per thread, (c) the same plus a workgroup barrier and an LDS allocation like the
// pass', for the grids the extract chain uses (315 tiles x nb frames x 256 threads; half as many workgroups of 512).
// hipcc --offload-arch=gfx950 -O2 tools/probe/launch_floor.hip -o /tmp/launch_floor && /tmp/launch_floor
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NT> __global__ __launch_bounds__(NT) void k_empty(const int* p, int* out) {}
template <int NT> __global__ __launch_bounds__(NT) void k_load(const int* p, int* out) {
    const unsigned int i = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NT + threadIdx.x;
    if (p[i] == 0x7fffffff) out[i] = 1;
}
template <int NT> __global__ __launch_bounds__(NT) void k_load_lds(const int* p, int* out) {
    __shared__ int lds[3600];
    const unsigned int i = (blockIdx.x + gridDim.x * (blockIdx.y + gridDim.y * blockIdx.z)) * NT + threadIdx.x;
    lds[threadIdx.x] = p[i];
    __syncthreads();
    if (lds[(threadIdx.x + 1) % NT] == 0x7fffffff) out[i] = 1;
}
template <typename F> static double time_launches(F launch, hipStream_t st, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    for (int i = 0; i < 20; i++) launch();
    hipStreamSynchronize(st);
    hipEventRecord(a, st);
    for (int i = 0; i < reps; i++) launch();
    hipEventRecord(b, st); hipEventSynchronize(b);
    float ms = 0; hipEventElapsedTime(&ms, a, b);
    return ms * 1000.0 / reps;
}
int main() {
    int* p; int* out; const size_t n = 21 * 15 * 8 * 512;
    hipMalloc(&p, n * 4); hipMalloc(&out, n * 4); hipMemset(p, 0, n * 4);
    hipStream_t st; hipStreamCreate(&st);
    for (int nb : {1, 2, 4, 8}) {
        dim3 g(21, 15, nb), g2(11, 15, nb);
        printf("nb=%d  256 threads x %4d wgs: empty %5.2f  load %5.2f  load+lds+barrier %5.2f us | 512 threads x %4d wgs: empty %5.2f  load %5.2f  load+lds+barrier %5.2f us\n",
               nb, 315 * nb,
               time_launches([&] { hipLaunchKernelGGL(k_empty<256>, g, dim3(256), 0, st, p, out); }, st, 400),
               time_launches([&] { hipLaunchKernelGGL(k_load<256>, g, dim3(256), 0, st, p, out); }, st, 400),
               time_launches([&] { hipLaunchKernelGGL(k_load_lds<256>, g, dim3(256), 0, st, p, out); }, st, 400),
               165 * nb,
               time_launches([&] { hipLaunchKernelGGL(k_empty<512>, g2, dim3(512), 0, st, p, out); }, st, 400),
               time_launches([&] { hipLaunchKernelGGL(k_load<512>, g2, dim3(512), 0, st, p, out); }, st, 400),
               time_launches([&] { hipLaunchKernelGGL(k_load_lds<512>, g2, dim3(512), 0, st, p, out); }, st, 400));
    }
    return 0;
}
