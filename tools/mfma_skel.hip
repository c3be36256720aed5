// Diagnostic: the fp32-MFMA rate this box sustains for the SKELETON of an LDS-fed implicit-GEMM loop
// (LDS operand reads + MFMAs + one workgroup barrier per K tile; no global traffic), by wave tile
// shape, LDS read width, K-tile depth and workgroups per CU.  It bounds what any kernel with that
// skeleton can reach, at the clock the chip holds under that load.
//   hipcc --offload-arch=gfx950 -O3 tools/mfma_skel.hip -o /tmp/mfma_skel && /tmp/mfma_skel
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

// TM x TN accumulators per wave; W128: operands read as ds_read_b128 (4 k-steps per read) else b32
// KS: k-steps (MFMA K=2) per barrier
template <int TM, int TN, bool W128, int KS>
__global__ __launch_bounds__(256) void skel(float* out, int iters) {
    extern __shared__ __attribute__((aligned(16))) float lds[];
    f32x16 acc[TM][TN];
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;
    const int lane = threadIdx.x & 63;
    for (int i = threadIdx.x; i < 8192; i += 256) lds[i] = 1e-3f * (i & 255);
    __syncthreads();
    unsigned long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int it = 0; it < iters; ++it) {
        if (W128) {
            const float* base = lds + (lane & 31) * 20 + (lane >> 5) * 4 + (it & 1) * 2048;
#pragma unroll
            for (int q = 0; q < KS / 4; ++q) {
                v4f av[TM], bv[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[tm] = *reinterpret_cast<const v4f*>(base + tm * 640 + q * 8);
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[tn] = *reinterpret_cast<const v4f*>(base + 2048 + tn * 640 + q * 8);
#pragma unroll
                for (int s = 0; s < 4; ++s)
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn)
                            acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm][s], bv[tn][s], acc[tm][tn], 0, 0, 0);
            }
        } else {
            const float* base = lds + lane + (it & 1) * 2048;
#pragma unroll
            for (int ks = 0; ks < KS; ++ks) {
                float av[TM], bv[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[tm] = base[ks * 128 + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[tn] = base[2048 + ks * 128 + tn * 32];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        __syncthreads();
    }
    unsigned long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float s = 0;
    for (int a = 0; a < TM; ++a) for (int b = 0; b < TN; ++b) for (int r = 0; r < 16; ++r) s += acc[a][b][r];
    out[2 + blockIdx.x * 256 + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) { out[0] = (float)(t1 - t0); out[1] = (float)(r1 - r0); }
}

template <int TM, int TN, bool W128, int KS>
void run(const char* name, int wgs_per_cu) {
    float* out; hipMalloc(&out, 4 * (2 + 256 * 4096));
    const int grid = 256 * wgs_per_cu;
    const int iters = 160000 / (KS * TM * TN);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    float ms = 0;
    for (int rep = 0; rep < 3; ++rep) {
        hipEventRecord(e0);
        hipLaunchKernelGGL((skel<TM, TN, W128, KS>), dim3(grid), dim3(256), 32768, 0, out, iters);
        hipEventRecord(e1); hipEventSynchronize(e1);
        hipEventElapsedTime(&ms, e0, e1);
    }
    float h[2]; hipMemcpy(h, out, 8, hipMemcpyDeviceToHost);
    double flops = (double)grid * 4 * (double)iters * KS * TM * TN * 4096.0;
    printf("%-34s wgs/cu=%d  %8.3f ms  %7.1f TFLOP/s  clock %.2f GHz\n", name, wgs_per_cu, ms, flops / ms / 1e9,
           h[0] / h[1] * 0.1);
    hipFree(out);
}

int main() {
    for (int w : {1, 2, 4, 5}) run<1, 1, false, 16>("1x1 b32  16 ksteps/barrier", w);
    for (int w : {1, 2, 4, 5}) run<1, 1, true, 16>("1x1 b128 16 ksteps/barrier", w);
    for (int w : {1, 2, 4}) run<2, 2, false, 8>("2x2 b32   8 ksteps/barrier", w);
    for (int w : {1, 2, 4}) run<2, 2, true, 8>("2x2 b128  8 ksteps/barrier", w);
    for (int w : {1, 2, 4}) run<2, 2, true, 16>("2x2 b128 16 ksteps/barrier", w);
    for (int w : {1, 2}) run<2, 4, true, 8>("2x4 b128  8 ksteps/barrier", w);
    for (int w : {1, 2}) run<2, 4, true, 16>("2x4 b128 16 ksteps/barrier", w);
    return 0;
}
