// Diagnostic: what fp32 MFMA rate does this box sustain with W waves per SIMD, no memory traffic,
// optionally with a workgroup barrier and a few LDS reads every 32 MFMAs (the conv kernel's skeleton)?
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_peak.hip -o /tmp/mfma_peak && /tmp/mfma_peak
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int NACC, int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a0, float b0) {
    __shared__ float lds[4096];
    f32x16 acc[NACC];
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) acc[i][r] = 0.f;
    float a = a0 + threadIdx.x * 1e-3f, b = b0 + threadIdx.x * 2e-3f;
    lds[threadIdx.x] = a; lds[threadIdx.x + 256] = b;
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int s = 0; s < 32 / NACC; ++s) {
            if (MODE >= 1) { a = lds[(threadIdx.x + s * 64) & 4095]; b = lds[(threadIdx.x + s * 64 + 256) & 4095]; }
#pragma unroll
            for (int i = 0; i < NACC; ++i) acc[i] = __builtin_amdgcn_mfma_f32_32x32x2f32(a, b, acc[i], 0, 0, 0);
        }
        if (MODE >= 2) __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int i = 0; i < NACC; ++i) for (int r = 0; r < 16; ++r) s += acc[i][r];
    out[blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); out[1] = (float)(r1 - r0); }
}

template <int NACC, int MODE>
void run(const char* name, int wgs_per_cu, size_t lds_pad) {
    float* out; hipMalloc(&out, 4 * 256 * 4096);
    const int iters = 4000, grid = 256 * wgs_per_cu;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    for (int rep = 0; rep < 2; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((k<NACC, MODE>), dim3(grid), dim3(256), lds_pad, 0, out, iters, 1.0f, 0.5f);
        hipEventRecord(e1); hipEventSynchronize(e1);
    }
    float ms; hipEventElapsedTime(&ms, e0, e1);
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    double flops = (double)grid * 4 * iters * 32 * 4096.0;
    printf("%-28s wgs/cu=%d  %7.3f ms  %7.1f TFLOP/s  in-kernel clock %.2f GHz\n", name, wgs_per_cu, ms, flops / ms / 1e9,
           h[0] / h[1] * 0.1);
    hipFree(out);
}

int main() {
    for (int w = 1; w <= 4; ++w) run<4, 0>("regs only, 4 acc", w, 0);
    for (int w = 1; w <= 3; ++w) run<4, 1>("lds reads, 4 acc", w, 0);
    for (int w = 1; w <= 3; ++w) run<4, 2>("lds reads + barrier/32", w, 0);
    run<1, 0>("regs only, 1 acc", 1, 0);
    run<2, 0>("regs only, 2 acc", 1, 0);
    return 0;
}
