// What v_cvt_i32_f32 does with NaN, infinities and out-of-range values on this part (fx32 in ssf_math.hpp relies on: NaN -> 0,
// saturation to INT_MAX / INT_MIN).   hipcc --offload-arch=gfx950 -O2 tools/probe/cvt_i32_f32.hip -o /tmp/cvt && /tmp/cvt
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstring>
__global__ void k(const float* in, int* out, int n) {
    const int i = threadIdx.x;
    if (i >= n) return;
    int r; const float t = in[i];
    asm("v_cvt_i32_f32 %0, %1" : "=v"(r) : "v"(t));
    out[i] = r;
}
int main() {
    float h[12] = {0.f, 1.5f, -1.5f, 2147483520.0f, 2147483648.0f, 3e9f, -2147483648.0f, -3e9f, INFINITY, -INFINITY, NAN, -0.0f};
    unsigned nanbits = 0xFFC00001u; std::memcpy(&h[11], &nanbits, 4);       // a negative NaN with payload
    float* d; int* o; int ho[12];
    hipMalloc(&d, sizeof(h)); hipMalloc(&o, sizeof(ho));
    hipMemcpy(d, h, sizeof(h), hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, o, 12);
    hipMemcpy(ho, o, sizeof(ho), hipMemcpyDeviceToHost);
    const int want[12] = {0, 1, -1, 2147483520, 2147483647, 2147483647, (int)0x80000000, (int)0x80000000, 2147483647, (int)0x80000000, 0, 0};
    int bad = 0;
    for (int i = 0; i < 12; i++) { printf("%g -> %d (want %d)\n", h[i], ho[i], want[i]); bad += ho[i] != want[i]; }
    printf(bad ? "MISMATCH\n" : "as the ISA manual says\n");
    return bad;
}
