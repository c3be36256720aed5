// Chamfer distance forward for Stereo2Point (the op the reference builds as extensions/chamfer_dist,
// /root/reference/README.md:64-65; its source is not in the mount).
//   dist1[b,i] = min_j |p[b,i] - q[b,j]|^2,  idx1[b,i] = argmin_j (first minimum)
//   dist2[b,j] = min_i |p[b,i] - q[b,j]|^2,  idx2[b,j] = argmin_i
// K = 3, so this is fp32 VALU work, not matrix-core work.  A workgroup owns 64*QPT query points; the other cloud is
// staged through LDS as (x, y, z, -) quads, 2048 points (32 KiB) per pass — the whole cloud at the BASELINE size, one
// barrier pair per launch — and every lane reads a candidate at the same address (LDS broadcast, one ds_read_b128
// per candidate per wave).  The four waves each scan a QUARTER of the staged candidates for the same queries (QPT
// independent chains per thread), and the four partial minima are merged through LDS with an explicit index
// comparison on ties, so the result keeps the first (lowest) index exactly as a sequential scan does.  The distance
// is ((dx*dx + dy*dy) + dz*dz) with contraction disabled: bit-identical to the oracle.  Both directions run in ONE
// launch (blockIdx.z): 1024 workgroups at the BASELINE size, 4 waves per SIMD (r01: two launches of 512, staging
// passes of 1024 with a barrier pair each, 25 "TFLOP/s" at 8 flops per pair).
//
// Instruction budget (the roof of this op is VALU issue: 78.6 T lane-instructions/s): a pair costs 3 sub + 3 mul +
// 2 add = 8 instructions for its distance — the floor of the oracle's formula; tracking (min, argmin) per pair would
// cost 3 more (v_cmp + 2 v_cndmask).  The candidates are taken in BLOCKS of 8: the block's minimum is 3 v_min3_f32 +
// 1 v_min (0.5 per pair) and the argmin is only looked for when the block minimum beats the running best (strictly:
// an equal distance later in the scan never replaces an earlier index) — a branch a wave takes ~ln(M) times per
// query, skipped by s_cbranch_execz otherwise: ~8.6 instructions per pair.  The file is built with
// -fno-slp-vectorize: plain -O3 packs the chains into v_pk_*_f32, which issue several times slower on gfx950.
// (Fetching the candidates by SCALAR loads instead — the range is wave-uniform — measured the same 66 us for 1, 2 and
// 4 queries per thread: every wave then waits out a scalar-cache miss per block, in lock-step with its neighbours.)
#include "s3r_kernels.h"
#include <cstdlib>

namespace s3r {

typedef float v4f __attribute__((ext_vector_type(4)));
constexpr int CH_TILE = 2048;   // candidates staged per pass (4 slices of 512): 32 KiB
constexpr int QPT = 2;          // query points per thread
constexpr int CBLK = 8;         // candidates per block (one min tree, one rare argmin search)

__device__ __forceinline__ float dist3(float px, float py, float pz, const v4f& v) {
#pragma clang fp contract(off)   // (dx*dx + dy*dy) + dz*dz with three roundings each, as the oracle computes it
    const float dx = px - v[0], dy = py - v[1], dz = pz - v[2];
    const float xx = dx * dx, yy = dy * dy, zz = dz * dz;
    return (xx + yy) + zz;
}

// direction 0: queries p (N points) against candidates q (M points) -> d1 / i1; direction 1: the roles swapped
__global__ __launch_bounds__(256) void chamfer_kernel(const float* __restrict__ p, const float* __restrict__ q,
                                                      float* __restrict__ d1, float* __restrict__ d2, int* __restrict__ i1,
                                                      int* __restrict__ i2, int N, int M) {
    __shared__ v4f qs[CH_TILE];
    __shared__ float rd[3][64 * QPT];
    __shared__ int ri[3][64 * QPT];
    const int dir = blockIdx.z;
    const int nq = dir ? M : N, nc = dir ? N : M;                 // queries / candidates of this direction
    if (blockIdx.x * 64 * QPT >= nq) return;                       // (the grid covers the larger cloud; block-uniform)
    const int b = blockIdx.y;
    const int lane = threadIdx.x & 63;
    const int slice = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int i0 = (blockIdx.x * 64 + lane) * QPT;
    const float* __restrict__ pb = (dir ? q : p) + (size_t)b * nq * 3;
    const float* __restrict__ qb = (dir ? p : q) + (size_t)b * nc * 3;
    float* __restrict__ dist = dir ? d2 : d1;
    int* __restrict__ idx = dir ? i2 : i1;
    float px[QPT], py[QPT], pz[QPT], best[QPT];
    int besti[QPT];
#pragma unroll
    for (int k = 0; k < QPT; ++k) {
        const int i = min(i0 + k, nq - 1);
        px[k] = pb[i * 3 + 0]; py[k] = pb[i * 3 + 1]; pz[k] = pb[i * 3 + 2];
        best[k] = __builtin_inff();
        besti[k] = 0;
    }
    for (int j0 = 0; j0 < nc; j0 += CH_TILE) {
        const int cnt = min(CH_TILE, nc - j0);
        if (j0) __syncthreads();
        for (int t = threadIdx.x; t < cnt; t += 256) {
            const float* s = qb + (size_t)(j0 + t) * 3;
            v4f v = {s[0], s[1], s[2], 0.f};
            qs[t] = v;
        }
        __syncthreads();
        // this wave's quarter of the staged candidates, in index order
        const int per = (cnt + 3) >> 2;
        const int t_begin = min(cnt, slice * per), t_end = min(cnt, t_begin + per);
        int t = t_begin;
        for (; t + CBLK <= t_end; t += CBLK) {
            float d[QPT][CBLK];
#pragma unroll
            for (int e = 0; e < CBLK; ++e) {
                const v4f v = qs[t + e];
#pragma unroll
                for (int k = 0; k < QPT; ++k) d[k][e] = dist3(px[k], py[k], pz[k], v);
            }
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const float m = __builtin_fminf(__builtin_fminf(__builtin_fminf(__builtin_fminf(d[k][0], d[k][1]), d[k][2]),
                                                                __builtin_fminf(__builtin_fminf(d[k][3], d[k][4]), d[k][5])),
                                                __builtin_fminf(d[k][6], d[k][7]));
                if (m < best[k]) {                 // rare: ~ln(M) times per query
                    int e_min = CBLK - 1;
#pragma unroll
                    for (int e = CBLK - 2; e >= 0; --e) e_min = d[k][e] == m ? e : e_min;    // the FIRST candidate at the minimum
                    best[k] = m;
                    besti[k] = j0 + t + e_min;
                }
            }
        }
        for (; t < t_end; ++t) {                   // the quarter's last < 8 candidates
            const v4f v = qs[t];
#pragma unroll
            for (int k = 0; k < QPT; ++k) {
                const float dd = dist3(px[k], py[k], pz[k], v);
                if (dd < best[k]) { best[k] = dd; besti[k] = j0 + t; }
            }
        }
    }
    // merge the four slices' minima: strictly-smaller wins, equal distances keep the lowest candidate index
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
            if (i0 + k < nq) {
                dist[(size_t)b * nq + i0 + k] = d;
                idx[(size_t)b * nq + i0 + k] = bi;
            }
        }
    }
}

hipError_t launch_chamfer(const float* p, const float* q, float* d1, float* d2, int* i1, int* i2, int B, int N,
                          int M, hipStream_t s) {
    constexpr int PER_WG = 64 * QPT;
    const int big = N > M ? N : M;
    hipLaunchKernelGGL(chamfer_kernel, dim3((big + PER_WG - 1) / PER_WG, B, 2), dim3(256), 0, s, p, q, d1, d2, i1, i2, N, M);
    return hipGetLastError();
}

}  // namespace s3r
