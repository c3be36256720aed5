// Bandwidth-bound kernels of the bf16 channels-last path (BASELINE.json configs[2]): the Cin=3 stem
// (fp32 NCHW renders -> bf16 NHWC features), the fused bidirectional cost volume on channels-last bf16
// features, and the 1x1x1 occupancy head (bf16 NDHWC -> fp32 (B,32,32,32) probabilities).
#include "s3r_kernels.h"

namespace s3r {

typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));
typedef unsigned v4u_u __attribute__((ext_vector_type(4), aligned(4)));

__device__ __forceinline__ unsigned pack2(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};
    return __builtin_bit_cast(unsigned, v);
}
__device__ __forceinline__ float bf_lo(unsigned u) { return __builtin_bit_cast(float, u << 16); }
__device__ __forceinline__ float bf_hi(unsigned u) { return __builtin_bit_cast(float, u & 0xffff0000u); }

// ------------------------------------------------------------------------------------------------
// Stem: Conv2d(3 -> 32, k3, s2, p1) + affine + ReLU, fp32 NCHW in, bf16 NHWC (halo-padded) out.
// One thread per output pixel, 32 couts in registers = 64 bytes of the channels-last output.  Stored as they
// stand, a wave's store instruction would put 16 bytes into each of 64 pixels (one instruction touching 32
// lines: store-issue bound, 2.7 TB/s in r01); instead the workgroup's 256 x 64 B go through LDS and every store
// instruction writes 16 CONSECUTIVE pixels = 1 KiB contiguous (whole 128-byte lines).
//   images [0, nsplit) come from x, images [nsplit, N) from x2 (left / right renders: no concatenation copy)
__global__ __launch_bounds__(256) void stem_bf16_kernel(const float* __restrict__ x, const float* __restrict__ x2, int nsplit,
                                                        const float* __restrict__ wt,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        unsigned short* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo,
                                                        int y_bs, int y_hs, int y_org) {
    __shared__ __attribute__((aligned(16))) v4u st[256 * 4];      // [pixel][4 x 16 B], part q at slot q ^ ((pixel >> 1) & 3)
    __shared__ int yo[256];                                        // output element offset of each pixel, -1 = none
    const int HWo = Ho * Wo;
    const int tid = threadIdx.x;
    const long long gid = (long long)blockIdx.x * 256 + tid;
    const bool live = gid < (long long)N * HWo;
    const long long g = live ? gid : (long long)N * HWo - 1;
    const int n = (int)(g / HWo);
    const int sp = (int)(g - (long long)n * HWo);
    const int oh = sp / Wo, ow = sp - oh * Wo;
    const int ih0 = oh * 2 - 1, iw0 = ow * 2 - 1;
    const float* __restrict__ xn = n < nsplit ? x + (size_t)n * 3 * Hi * Wi : x2 + (size_t)(n - nsplit) * 3 * Hi * Wi;
    // accumulators in pairs: the channel loop compiles to v_pk_fma_f32 (two exact fp32 FMAs per lane per instruction,
    // the weight pair straight from SGPRs) — half the VALU instructions of the scalar form, same bits
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc2[c] = (f32x2){0.f, 0.f};
    // (channel, row) loops rolled: see stem_kernel — unrolled, the 864 scalar weight loads overflow the SGPR file
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci)
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = ih0 + kh;
            const bool vh = (unsigned)ih < (unsigned)Hi;
            const float* __restrict__ xrow = xn + ((size_t)ci * Hi + ih) * Wi;
            const float* __restrict__ wrow = wt + (ci * 3 + kh) * 96;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = iw0 + kw;
                const bool v = vh && ((unsigned)iw < (unsigned)Wi);
                const float xv = v ? xrow[iw] : 0.f;
                const f32x2 xv2 = {xv, xv};
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    acc2[c] = __builtin_elementwise_fma(xv2, *reinterpret_cast<const f32x2*>(wrow + kw * 32 + 2 * c), acc2[c]);
            }
        }
    float acc[32];
#pragma unroll
    for (int c = 0; c < 16; ++c) { acc[2 * c] = acc2[c].x; acc[2 * c + 1] = acc2[c].y; }
    yo[tid] = live ? (int)((size_t)n * y_bs + y_org + ((size_t)oh * y_hs + (size_t)ow * 32)) : -1;
    const int sw = (tid >> 1) & 3;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        v4u t;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int c = q * 8 + k * 2;
            t[k] = pack2(fmaxf(fmaf(acc[c], scale[c], shift[c]), 0.f), fmaxf(fmaf(acc[c + 1], scale[c + 1], shift[c + 1]), 0.f));
        }
        st[tid * 4 + (q ^ sw)] = t;
    }
    __syncthreads();
    // lane l of wave w stores part l & 3 of pixel 64 w + 16 j + (l >> 2): 16 consecutive pixels per instruction
    const int lane = tid & 63, wbase = tid & ~63;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int pix = wbase + 16 * j + (lane >> 2), q = lane & 3;
        const int off = yo[pix];
        const v4u t = st[pix * 4 + (q ^ ((pix >> 1) & 3))];
        if (off >= 0) *reinterpret_cast<v4u*>(y + (size_t)off + q * 8) = t;       // 64-byte pixels: 16-byte aligned
    }
}

hipError_t launch_stem_bf16(const float* x, const float* x2, int nsplit, const float* wt, const float* scale,
                            const float* shift, void* y, int N, int Hi, int Wi, int Ho, int Wo, int y_bs, int y_hs, int y_org,
                            hipStream_t s) {
    const long long total = (long long)N * Ho * Wo;
    if (!x2) { x2 = x; nsplit = N; }
    hipLaunchKernelGGL(stem_bf16_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, x2, nsplit, wt, scale,
                       shift, reinterpret_cast<unsigned short*>(y), N, Hi, Wi, Ho, Wo, y_bs, y_hs, y_org);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Cost volume on channels-last bf16: fl, fr (B,H,W,C) -> vol (B, D+2h, H+2h, W+2h, 2C), interior only:
//   vol[b,d,h,w, c]     = L[b,h,w,c] - R[b,h,w-d,c]   (0 where w-d < 0)
//   vol[b,d,h,w, C + c] = R[b,h,w,c] - L[b,h,w+d,c]   (0 where w+d >= W)
// One workgroup per (b, h) feature-row pair: both rows (W x C bf16 each) go to LDS once; a thread owns one
// 16-byte group (8 channels) of one output position, keeps its reference operand in registers and walks the D
// disparities: one ds_read_b128 of the shifted operand + one 16-byte store per step, no address arithmetic but an
// add (r01's one-thread-per-output form spent its time on five integer divisions per 16 bytes: VALU-bound at
// 3.3 TB/s).  A wave's store covers 8 consecutive positions = 1 KiB contiguous; an interior row of W positions
// is W whole 128-byte lines (2C = 64), so interior-only writes leave no partial line.  The differences are formed
// in fp32 and rounded to bf16 once.
__global__ __launch_bounds__(256) void cost_volume_bf16_kernel(const unsigned short* __restrict__ fl,
                                                               const unsigned short* __restrict__ fr,
                                                               unsigned short* __restrict__ vol, int C, int D,
                                                               int H, int W, int halo) {
    extern __shared__ __attribute__((aligned(16))) v4u cvh[];      // [2][W][C/8]
    const int CG = C >> 3;                            // 16-byte groups per position per view
    const int G = 2 * CG;                             // ... per output position
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    v4u* sl = cvh;
    v4u* sr = cvh + W * CG;
    const v4u* __restrict__ pl = reinterpret_cast<const v4u*>(fl + ((size_t)(b * H + hh) * W) * C);
    const v4u* __restrict__ pr = reinterpret_cast<const v4u*>(fr + ((size_t)(b * H + hh) * W) * C);
    for (int i = threadIdx.x; i < W * CG; i += 256) {
        sl[i] = pl[i];
        sr[i] = pr[i];
    }
    __syncthreads();
    const int Wp = W + 2 * halo, Hp = H + 2 * halo, Dp = D + 2 * halo;
    const size_t dstride = (size_t)Hp * Wp * 2 * C;
    for (int i = threadIdx.x; i < W * G; i += 256) {              // one pass at the network's shape (28 x 8 = 224)
        const int w = i / G, g = i - w * G;
        const bool right_ref = g >= CG;
        const int cg = right_ref ? g - CG : g;
        const v4u a = (right_ref ? sr : sl)[w * CG + cg];
        const v4u* m = (right_ref ? sl : sr) + cg;
        const int step = right_ref ? CG : -CG;
        unsigned short* po = vol + ((((size_t)b * Dp + halo) * Hp + hh + halo) * Wp + w + halo) * (2 * C) + g * 8;
        int ws = w, mi = w * CG;
        const int dws = right_ref ? 1 : -1;
#pragma unroll 4
        for (int d = 0; d < D; ++d) {
            v4u out = {0u, 0u, 0u, 0u};
            if (ws >= 0 && ws < W) {
                const v4u sft = m[mi];
#pragma unroll
                for (int k = 0; k < 4; ++k) out[k] = pack2(bf_lo(a[k]) - bf_lo(sft[k]), bf_hi(a[k]) - bf_hi(sft[k]));
            }
            *reinterpret_cast<v4u*>(po) = out;
            po += dstride;
            ws += dws;
            mi += step;
        }
    }
}

hipError_t launch_cost_volume_bf16(const void* fl, const void* fr, void* vol, int B, int C, int D, int H, int W, int halo,
                                   hipStream_t s) {
    const size_t lds = (size_t)2 * W * (C >> 3) * 16;
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(cost_volume_bf16_kernel, dim3((unsigned)(B * H)), dim3(256), lds, s,
                       reinterpret_cast<const unsigned short*>(fl), reinterpret_cast<const unsigned short*>(fr),
                       reinterpret_cast<unsigned short*>(vol), C, D, H, W, halo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Occupancy head: Conv3d(C -> 1, k=1) + bias + activation, bf16 (B,S,C) channels-last in, fp32 (B,S) out.
// Eight lanes share a voxel (16 bytes = 8 channels each for C = 64), reduce with wavefront shuffles.
__global__ __launch_bounds__(256) void head_bf16_kernel(const unsigned short* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        float* __restrict__ y, int C, long long voxels, int act) {
    const int lanes_per = C >> 3;                      // lanes per voxel (C % 8 == 0, C <= 512)
    const int per_wave = 64 / lanes_per;
    const long long wave_id = ((long long)blockIdx.x * 256 + threadIdx.x) >> 6;
    const int lane = threadIdx.x & 63;
    const int sub = lane % lanes_per;
    const long long v = wave_id * per_wave + lane / lanes_per;
    float s = 0.f;
    if (v < voxels) {
        const v4u a = *reinterpret_cast<const v4u*>(x + (size_t)v * C + sub * 8);
        const float* __restrict__ wk = w + sub * 8;
#pragma unroll
        for (int k = 0; k < 4; ++k) s = fmaf(bf_hi(a[k]), wk[2 * k + 1], fmaf(bf_lo(a[k]), wk[2 * k], s));
    }
    for (int o = lanes_per >> 1; o > 0; o >>= 1) s += __shfl_xor(s, o, 64);
    if (sub == 0 && v < voxels) {
        float t = fmaf(s, scale ? scale[0] : 1.f, shift ? shift[0] : 0.f);
        if (act == ACT_RELU) t = fmaxf(t, 0.f);
        else if (act == ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
        y[v] = t;
    }
}

hipError_t launch_head_bf16(const void* x, const float* w, const float* scale, const float* shift, float* y, int C,
                            int64_t voxels, int act, hipStream_t s) {
    const int lanes_per = C >> 3;
    if (C % 8 != 0 || lanes_per > 64 || (lanes_per & (lanes_per - 1)) != 0) return hipErrorInvalidValue;
    const long long waves = (voxels + 64 / lanes_per - 1) / (64 / lanes_per);
    hipLaunchKernelGGL(head_bf16_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s,
                       reinterpret_cast<const unsigned short*>(x), w, scale, shift, y, C, (long long)voxels, act);
    return hipGetLastError();
}

}  // namespace s3r
