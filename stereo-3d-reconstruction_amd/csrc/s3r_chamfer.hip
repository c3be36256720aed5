// Chamfer distance forward for Stereo2Point (the op the reference builds as extensions/chamfer_dist,
// /root/reference/README.md:64-65; its source is not in the mount).
//   dist1[b,i] = min_j |p[b,i] - q[b,j]|^2,  idx1[b,i] = argmin_j (first minimum)
//   dist2[b,j] = min_i |p[b,i] - q[b,j]|^2,  idx2[b,j] = argmin_i
// K = 3, so this is fp32 VALU work, not matrix-core work: one thread owns one query point, the
// other cloud is staged through LDS in 1024-point xyz tiles that every lane reads at the same
// address (LDS broadcast, one ds_read_b128 per candidate per wave).  The distance is evaluated as
// ((dx*dx + dy*dy) + dz*dz) with contraction disabled so results are bit-identical to the oracle.
#include "s3r_kernels.h"

namespace s3r {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int CH_TILE = 1024;

__global__ __launch_bounds__(256) void chamfer_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                      float* __restrict__ dist, int* __restrict__ idx, int N, int M) {
    __shared__ v4f qs[CH_TILE];
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;
    const float* __restrict__ pb = p + (size_t)b * N * 3;
    const float* __restrict__ qb = q + (size_t)b * M * 3;
    float px = 0.f, py = 0.f, pz = 0.f;
    if (i < N) { px = pb[i * 3 + 0]; py = pb[i * 3 + 1]; pz = pb[i * 3 + 2]; }
    float best = __builtin_inff();
    int besti = 0;
    for (int j0 = 0; j0 < M; j0 += CH_TILE) {
        const int cnt = min(CH_TILE, M - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) {
            const float* s = qb + (size_t)(j0 + t) * 3;
            v4f v = {s[0], s[1], s[2], 0.f};
            qs[t] = v;
        }
        __syncthreads();
#pragma unroll 8
        for (int t = 0; t < cnt; ++t) {
#pragma clang fp contract(off)   // (dx*dx + dy*dy) + dz*dz with three roundings each, as the oracle computes it
            const v4f v = qs[t];
            const float dx = px - v[0], dy = py - v[1], dz = pz - v[2];
            const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
            const float d = (xx + yy) + zz;
            if (d < best) { best = d; besti = j0 + t; }
        }
    }
    if (i < N) {
        dist[(size_t)b * N + i] = best;
        idx[(size_t)b * N + i] = besti;
    }
}

hipError_t launch_chamfer(const float* p, const float* q, float* d1, float* d2, int* i1, int* i2, int B, int N,
                          int M, hipStream_t s) {
    hipLaunchKernelGGL(chamfer_kernel, dim3((N + 255) / 256, B), dim3(256), 0, s, p, q, d1, i1, N, M);
    hipLaunchKernelGGL(chamfer_kernel, dim3((M + 255) / 256, B), dim3(256), 0, s, q, p, d2, i2, M, N);
    return hipGetLastError();
}

}  // namespace s3r
