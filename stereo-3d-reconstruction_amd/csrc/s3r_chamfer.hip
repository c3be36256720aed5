// Chamfer distance forward for Stereo2Point (the op the reference builds as extensions/chamfer_dist,
// /root/reference/README.md:64-65; its source is not in the mount).
//   dist1[b,i] = min_j |p[b,i] - q[b,j]|^2,  idx1[b,i] = argmin_j (first minimum)
//   dist2[b,j] = min_i |p[b,i] - q[b,j]|^2,  idx2[b,j] = argmin_i
// K = 3, so this is fp32 VALU work, not matrix-core work.  A workgroup owns 64*QPT query points; its four
// waves each scan a QUARTER of the other cloud (staged through LDS in xyz tiles that every lane reads at the
// same address: LDS broadcast, one ds_read_b128 per candidate per wave) for the same queries, QPT independent
// compare chains per thread (the candidate fetch and the loop overhead are paid once per QPT pairs), and the
// four partial minima are merged through LDS in slice order, so ties keep the first (lowest) index exactly as
// a sequential scan does.  The distance is ((dx*dx + dy*dy) + dz*dz) with contraction disabled: bit-identical
// to the oracle.
#include "s3r_kernels.h"

namespace s3r {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int CH_TILE = 1024;   // candidates staged per pass (4 slices of 256)
constexpr int QPT = 2;          // query points per thread

__global__ __launch_bounds__(256) void chamfer_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                      float* __restrict__ dist, int* __restrict__ idx, int N, int M) {
    __shared__ v4f qs[CH_TILE];
    __shared__ float rd[3][64 * QPT];
    __shared__ int ri[3][64 * QPT];
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63, slice = threadIdx.x >> 6;
    const int i0 = (blockIdx.x * 64 + lane) * QPT;
    const float* __restrict__ pb = p + (size_t)b * N * 3;
    const float* __restrict__ qb = q + (size_t)b * M * 3;
    float px[QPT], py[QPT], pz[QPT], best[QPT];
    int besti[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = min(i0 + k, N - 1);
        px[k] = pb[i * 3 + 0]; py[k] = pb[i * 3 + 1]; pz[k] = pb[i * 3 + 2];
        best[k] = __builtin_inff();
        besti[k] = 0;
    }
    for (int j0 = 0; j0 < M; j0 += CH_TILE) {
        const int cnt = min(CH_TILE, M - j0);
        __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) {
            const float* s = qb + (size_t)(j0 + t) * 3;
            v4f v = {s[0], s[1], s[2], 0.f};
            qs[t] = v;
        }
        __syncthreads();
        // this wave's quarter of the staged candidates, in index order
        const int per = (cnt + 3) >> 2;
        const int t_begin = min(cnt, slice * per), t_end = min(cnt, t_begin + per);
#pragma unroll 4
        for (int t = t_begin; t < t_end; ++t) {
#pragma clang fp contract(off)   // (dx*dx + dy*dy) + dz*dz with three roundings each, as the oracle computes it
            const v4f v = qs[t];
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const float dx = px[k] - v[0], dy = py[k] - v[1], dz = pz[k] - v[2];
                const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
                const float d = (xx + yy) + zz;
                if (d < best[k]) { best[k] = d; besti[k] = j0 + t; }
            }
        }
    }
    // merge the four slices' minima: strictly-smaller wins, so equal distances keep the lowest candidate index
    // (within a staging pass the slices are in index order; across passes an earlier pass always has lower indices
    // but may sit in ANY slice, hence the explicit index comparison on ties)
    if (slice > 0) {
#pragma unroll
        for (int k = 0; k < QPT; ++k) { rd[slice - 1][lane * QPT + k] = best[k]; ri[slice - 1][lane * QPT + k] = besti[k]; }
    }
    __syncthreads();
    if (slice == 0) {
#pragma unroll
        for (int k = 0; k < QPT; ++k) {
            float d = best[k];
            int bi = besti[k];
#pragma unroll
            for (int s2 = 0; s2 < 3; ++s2) {
                const float od = rd[s2][lane * QPT + k];
                const int oi = ri[s2][lane * QPT + k];
                if (od < d || (od == d && oi < bi)) { d = od; bi = oi; }
            }
            if (i0 + k < N) {
                dist[(size_t)b * N + i0 + k] = d;
                idx[(size_t)b * N + i0 + k] = bi;
            }
        }
    }
}

hipError_t launch_chamfer(const float* p, const float* q, float* d1, float* d2, int* i1, int* i2, int B, int N,
                          int M, hipStream_t s) {
    constexpr int PER_WG = 64 * QPT;
    hipLaunchKernelGGL(chamfer_kernel, dim3((N + PER_WG - 1) / PER_WG, B), dim3(256), 0, s, p, q, d1, i1, N, M);
    hipLaunchKernelGGL(chamfer_kernel, dim3((M + PER_WG - 1) / PER_WG, B), dim3(256), 0, s, q, p, d2, i2, M, N);
    return hipGetLastError();
}

}  // namespace s3r
