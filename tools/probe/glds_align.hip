// tools/probe/glds_align.hip -- does global_load_lds_dwordx4 take a source that is only 4-byte aligned?  (round 6: the relabelling
// tile's quads start at X0 - 1, an odd label; the register path relies on global_load_dwordx4 taking that.)
//   hipcc --offload-arch=gfx950 -O2 tools/probe/glds_align.hip -o tools/probe/bin/glds_align && tools/probe/bin/glds_align
// For each source misalignment a = 0..3 labels: a 34 x 36 tile fetched as 306 quads straight into LDS, compared with the image;
// the dword form (global_load_lds_dword) beside it.  Prints mismatches per form, and the time of 1000 back-to-back tile fetches.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %s at %d\n", hipGetErrorString(e_), __LINE__); return 1; } } while (0)
constexpr int TWP = 36, TW = 34, NQ = 9 * TW;
template <int FORM>
__global__ __launch_bounds__(256) void k(const int* __restrict__ src, int* __restrict__ out, int W, int a, int reps) {
    __shared__ __attribute__((aligned(16))) int tile[320 * 4];
    const unsigned int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int X0 = 64 * (blockIdx.x % 8) + a, Y0 = 3 + 32 * (blockIdx.x / 8);
    for (int r = 0; r < reps; r++) {
        if (FORM == 0) {
            for (int k2 = 0; k2 < 2; k2++) {
                const unsigned int e = min(threadIdx.x + 256u * k2, (unsigned int)(NQ - 1)), ly = e / 9;
                const unsigned int off = 4u * ((unsigned int)(Y0 * W + X0) + 4u * e + ly * (unsigned int)(W - TWP));
                const unsigned int q0 = 256u * k2 + 64u * wave;
                if (q0 < (unsigned int)NQ)
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src) + off),
                                                     (__attribute__((address_space(3))) void*)&tile[4u * q0], 16, 0, 0);
            }
        } else {
            for (int k2 = 0; k2 < 5; k2++) {
                const unsigned int e = min(threadIdx.x + 256u * k2, (unsigned int)(TWP * TW - 1)), ly = e / TWP, lx = e - ly * TWP;
                const unsigned int off = 4u * ((unsigned int)((Y0 + ly) * W + X0) + lx);
                const unsigned int e0 = 256u * k2 + 64u * wave;
                if (e0 < (unsigned int)(TWP * TW))
                    __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(reinterpret_cast<const char*>(src) + off),
                                                     (__attribute__((address_space(3))) void*)&tile[e0], 4, 0, 0);
            }
        }
        __syncthreads();
    }
    for (int i = threadIdx.x; i < TWP * TW; i += 256) out[blockIdx.x * TWP * TW + i] = tile[i];
}
int main() {
    const int W = 640, H = 480, NB = 64;
    std::vector<int> img(W * H);
    for (int i = 0; i < W * H; i++) img[i] = i * 2654435761u >> 7;
    int *d_src, *d_out;
    CK(hipMalloc(&d_src, W * H * 4)); CK(hipMalloc(&d_out, NB * TWP * TW * 4));
    CK(hipMemcpy(d_src, img.data(), W * H * 4, hipMemcpyHostToDevice));
    std::vector<int> out(NB * TWP * TW);
    for (int form = 0; form < 2; form++) for (int a = 0; a < 4; a++) {
        CK(hipMemset(d_out, 0, NB * TWP * TW * 4));
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(256), 0, 0, d_src, d_out, W, a, 1);
        else hipLaunchKernelGGL(k<1>, dim3(NB), dim3(256), 0, 0, d_src, d_out, W, a, 1);
        CK(hipDeviceSynchronize());
        CK(hipMemcpy(out.data(), d_out, out.size() * 4, hipMemcpyDeviceToHost));
        long bad = 0;
        for (int b = 0; b < NB; b++) { const int X0 = 64 * (b % 8) + a, Y0 = 3 + 32 * (b / 8);
            for (int ly = 0; ly < TW; ly++) for (int lx = 0; lx < TWP; lx++) if (out[b * TWP * TW + ly * TWP + lx] != img[(Y0 + ly) * W + X0 + lx]) bad++; }
        hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
        CK(hipEventRecord(e0));
        if (form == 0) hipLaunchKernelGGL(k<0>, dim3(NB), dim3(256), 0, 0, d_src, d_out, W, a, 1000);
        else hipLaunchKernelGGL(k<1>, dim3(NB), dim3(256), 0, 0, d_src, d_out, W, a, 1000);
        CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        printf("form %s source misaligned by %d labels (%2d bytes): %ld of %d labels wrong; 1000 tile fetches %.1f us (%.0f ns per tile)\n",
               form == 0 ? "dwordx4" : "dword  ", a, 4 * a, bad, NB * TWP * TW, ms * 1e3, ms * 1e3);
    }
    return 0;
}
