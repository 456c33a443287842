// Probe: IEEE parity of basic fp32/fp64 ops between gfx950 and the host, launch latency.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <math.h>
#include <stdio.h>
#include <chrono>
extern "C" {
__global__ void k_arith(const float* a, const float* b, float* out, int n) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    float x = a[i], y = b[i];
    out[8*i+0] = x / y;
    out[8*i+1] = sqrtf(fabsf(x));
    out[8*i+2] = x * y + x;          // must NOT contract
    out[8*i+3] = 1.0f / sqrtf(fabsf(y));
    double dx = (double)x, dy = (double)y;
    out[8*i+4] = (float)(dx / dy);
    out[8*i+5] = (float)sqrt(fabs(dx) + 1e-300);
    out[8*i+6] = (float)__double2ll_rn(dx * 1073741824.0) ;
    out[8*i+7] = rintf(x * 1048576.0f);
}
__global__ void k_empty(int* p) { if (p && threadIdx.x == 0 && blockIdx.x == 0) p[0] += 1; }
int probe_arith(const float* a, const float* b, float* out, int n) {
    float *da, *db, *dout;
    if (hipMalloc(&da, n*4) != hipSuccess) return -1;
    hipMalloc(&db, n*4); hipMalloc(&dout, n*32);
    hipMemcpy(da, a, n*4, hipMemcpyHostToDevice); hipMemcpy(db, b, n*4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k_arith, dim3((n+255)/256), dim3(256), 0, 0, da, db, dout, n);
    hipError_t e = hipDeviceSynchronize();
    hipMemcpy(out, dout, n*32, hipMemcpyDeviceToHost);
    hipFree(da); hipFree(db); hipFree(dout);
    return (int)e;
}
// returns microseconds per dependent launch in a chain of `chain` tiny kernels
double probe_launch(int chain, int reps, int grid) {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
    hipStreamSynchronize(s);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipFree(d);
    return 1000.0 * ms / (double)(chain * reps);
}
// graph replay of the same chain
double probe_graph(int chain, int reps, int grid) {
    int* d; hipMalloc(&d, 4); hipMemset(d, 0, 4);
    hipStream_t s; hipStreamCreate(&s);
    hipGraph_t g; hipGraphExec_t ge;
    hipStreamBeginCapture(s, hipStreamCaptureModeGlobal);
    for (int i = 0; i < chain; i++) hipLaunchKernelGGL(k_empty, dim3(grid), dim3(256), 0, s, d);
    hipStreamEndCapture(s, &g);
    if (hipGraphInstantiate(&ge, g, nullptr, nullptr, 0) != hipSuccess) return -1;
    hipGraphLaunch(ge, s); hipStreamSynchronize(s);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0, s);
    for (int r = 0; r < reps; r++) hipGraphLaunch(ge, s);
    hipEventRecord(e1, s); hipStreamSynchronize(s);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    return 1000.0 * ms / (double)(chain * reps);
}
// host sync round trip: launch + 232-byte D2H + sync, microseconds
double probe_roundtrip(int reps) {
    int* d; hipMalloc(&d, 256); hipMemset(d, 0, 256);
    int* hpin; hipHostMalloc(&hpin, 256);
    hipStream_t s; hipStreamCreate(&s);
    hipStreamSynchronize(s);
    auto t0 = std::chrono::steady_clock::now();
    for (int r = 0; r < reps; r++) {
        hipLaunchKernelGGL(k_empty, dim3(256), dim3(256), 0, s, d);
        hipMemcpyAsync(hpin, d, 232, hipMemcpyDeviceToHost, s);
        hipStreamSynchronize(s);
    }
    auto t1 = std::chrono::steady_clock::now();
    return std::chrono::duration<double, std::micro>(t1 - t0).count() / reps;
}
}
