// Bandwidth-bound kernels of the path: the Cin=3 stem convolution, the fused bidirectional
// shift-and-diff cost volume, the 1x1x1 occupancy head (+sigmoid), the point-head linear layers,
// and the on-device thresholded IoU.  All are priced against the HBM roof.
#include "s3r_kernels.h"

namespace s3r {

typedef float v4f __attribute__((ext_vector_type(4)));

__device__ __forceinline__ float apply_act(float v, int act) {
    if (act == ACT_RELU) return fmaxf(v, 0.f);
    if (act == ACT_SIGMOID) return 1.f / (1.f + __expf(-v));
    return v;
}

// ------------------------------------------------------------------------------------------------
// Stem: Conv2d(3 -> 32, k3, s2, p1) + affine + ReLU.  K = 27 is too shallow for the matrix cores
// and the layer is write-bound (32 output planes per 3 input planes), so: one thread per output
// pixel, all 32 couts in registers, weights read through the scalar cache (wave-uniform), every
// store a 256-byte coalesced row segment.
//   wt: packed [27][32] (k-major, cout fastest)
__global__ __launch_bounds__(256) void stem_kernel(const float* __restrict__ x, const float* __restrict__ wt,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo,
                                                   int y_cs, int y_hs, int y_org) {
    const int HWo = Ho * Wo;
    const long long total = (long long)N * HWo;
    const long long gid = (long long)blockIdx.x * 256 + threadIdx.x;
    if (gid >= total) return;
    const int n = (int)(gid / HWo);
    const int sp = (int)(gid - (long long)n * HWo);
    const int oh = sp / Wo, ow = sp - oh * Wo;
    const int ih0 = oh * 2 - 1, iw0 = ow * 2 - 1;
    const float* __restrict__ xn = x + (size_t)n * 3 * Hi * Wi;

    float acc[32];
#pragma unroll
    for (int c = 0; c < 32; ++c) acc[c] = 0.f;
#pragma unroll
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = ih0 + kh;
            const bool vh = (unsigned)ih < (unsigned)Hi;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = iw0 + kw;
                const bool v = vh && ((unsigned)iw < (unsigned)Wi);
                const float xv = v ? xn[((size_t)ci * Hi + ih) * Wi + iw] : 0.f;
                const float* __restrict__ wk = wt + ((ci * 3 + kh) * 3 + kw) * 32;
#pragma unroll
                for (int c = 0; c < 32; ++c) acc[c] = fmaf(xv, wk[c], acc[c]);
            }
        }
    }
    float* __restrict__ yn = y + (size_t)n * 32 * y_cs + y_org + oh * y_hs + ow;   // (halo-padded) NCHW
#pragma unroll
    for (int c = 0; c < 32; ++c) yn[(size_t)c * y_cs] = fmaxf(fmaf(acc[c], scale[c], shift[c]), 0.f);
}

__global__ void pack_stem_kernel(const float* __restrict__ w, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 27*32
    if (i < 27 * 32) {
        const int c = i & 31, k = i >> 5;
        wt[i] = w[c * 27 + k];
    }
}

hipError_t launch_stem(const float* x, const float* wt, const float* scale, const float* shift, float* y, int N,
                       int Hi, int Wi, int Ho, int Wo, int y_cs, int y_hs, int y_org, hipStream_t s) {
    const long long total = (long long)N * Ho * Wo;
    hipLaunchKernelGGL(stem_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, wt, scale, shift, y,
                       N, Hi, Wi, Ho, Wo, y_cs, y_hs, y_org);
    return hipGetLastError();
}

hipError_t launch_pack_stem(const float* w, float* wt, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_kernel, dim3(4), dim3(256), 0, s, w, wt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Cost volume: vol[b, c,   d, h, w] = L[b,c,h,w] - R[b,c,h,w-d]   (0 where w-d < 0)
//              vol[b, C+c, d, h, w] = R[b,c,h,w] - L[b,c,h,w+d]   (0 where w+d >= W)
// One workgroup per (b, c): both HxW planes are read from HBM exactly once into LDS, then the two
// D*H*W output slabs are streamed out with 16-byte stores (4 consecutive w of one row), rows walked in
// (d,h) order so consecutive lanes write consecutive HBM lines.  The volume may carry a zero halo
// (`halo` elements on the d, h and w axes) for the 3D conv that consumes it; only the interior is
// written.  Algorithmic bytes = 4*(2*H*W + 2*D*H*W) per (b,c); the kernel moves exactly that.
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                          float* __restrict__ vol, int C, int D, int H, int W,
                                                          int halo) {
    extern __shared__ __attribute__((aligned(16))) float cv_smem[];
    const int HW = H * W;
    float* sl = cv_smem;
    float* sr = cv_smem + HW;
    const int bc = blockIdx.x;
    const int b = bc / C, c = bc - b * C;
    const float* __restrict__ pl = fl + (size_t)bc * HW;
    const float* __restrict__ pr = fr + (size_t)bc * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        sl[i] = pl[i];
        sr[i] = pr[i];
    }
    __syncthreads();
    const int DHW = D * HW;
    const int Wp = W + 2 * halo, Hp = H + 2 * halo, Dp = D + 2 * halo;
    const int hs = Wp, ds = Hp * Wp;
    const size_t cs = (size_t)Dp * ds;
    const int org = halo * (ds + hs + 1);
    float* __restrict__ ol = vol + ((size_t)b * 2 * C + c) * cs + org;
    float* __restrict__ orr = vol + ((size_t)b * 2 * C + C + c) * cs + org;
    if ((W & 3) == 0) {
        const int nq = DHW >> 2;
        for (int q = threadIdx.x; q < nq; q += 256) {
            const int e = q << 2;
            const int d = e / HW;
            const int hw = e - d * HW;
            const int hh = hw / W;
            const int w0 = hw - hh * W;
            v4f a, r;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const int w = w0 + k;
                a[k] = (w >= d) ? sl[hw + k] - sr[hw + k - d] : 0.f;
                r[k] = (w + d < W) ? sr[hw + k] - sl[hw + k + d] : 0.f;
            }
            const int o = d * ds + hh * hs + w0;
            *reinterpret_cast<v4f_u*>(ol + o) = a;
            *reinterpret_cast<v4f_u*>(orr + o) = r;
        }
    } else {
        for (int e = threadIdx.x; e < DHW; e += 256) {
            const int d = e / HW;
            const int hw = e - d * HW;
            const int hh = hw / W;
            const int w = hw - hh * W;
            const int o = d * ds + hh * hs + w;
            ol[o] = (w >= d) ? sl[hw] - sr[hw - d] : 0.f;
            orr[o] = (w + d < W) ? sr[hw] - sl[hw + d] : 0.f;
        }
    }
}

hipError_t launch_cost_volume(const float* fl, const float* fr, float* vol, int B, int C, int D, int H, int W,
                              int halo, hipStream_t s) {
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    hipLaunchKernelGGL(cost_volume_kernel, dim3(B * C), dim3(256), lds, s, fl, fr, vol, C, D, H, W, halo);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// pad_copy: (planes, D, H, W) -> interior of a zero-halo (planes, D+2hd, H+2hh, W+2hw) buffer.  Used
// when a caller hands an unpadded tensor to a layer whose gather wants the halo.
__global__ __launch_bounds__(256) void pad_copy_kernel(const float* __restrict__ x, float* __restrict__ y,
                                                       long long total, int D, int H, int W, int hd, int hh, int hw) {
    const int Hp = H + 2 * hh, Wp = W + 2 * hw;
    const long long Dp = D + 2 * hd;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int w = (int)(i % W);
        long long r = i / W;
        const int h = (int)(r % H);
        r /= H;
        const int d = (int)(r % D);
        const long long pl = r / D;
        y[((pl * Dp + d + hd) * Hp + h + hh) * Wp + w + hw] = x[i];
    }
}

hipError_t launch_pad_copy(const float* x, float* y, int64_t planes, int D, int H, int W, int hd, int hh, int hw,
                           hipStream_t s) {
    const long long total = (long long)planes * D * H * W;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(pad_copy_kernel, dim3((unsigned)(blocks < 8192 ? blocks : 8192)), dim3(256), 0, s, x, y, total,
                       D, H, W, hd, hh, hw);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Occupancy head: Conv3d(C -> 1, k=1) + bias + activation over S voxels per sample.
// One thread per 4 consecutive voxels; channel loop reads are 16-byte coalesced per channel plane.
__global__ __launch_bounds__(256) void head_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                   const float* __restrict__ scale, const float* __restrict__ shift,
                                                   float* __restrict__ y, int C, long long S, int act,
                                                   long long total4) {
    const long long q = (long long)blockIdx.x * 256 + threadIdx.x;
    if (q >= total4) return;
    const long long S4 = S >> 2;
    const long long b = q / S4;
    const long long sp = (q - b * S4) << 2;
    const float* __restrict__ xb = x + (size_t)b * C * S + sp;
    v4f acc = {0.f, 0.f, 0.f, 0.f};
    const float sc = scale ? scale[0] : 1.f, sf = shift ? shift[0] : 0.f;
    for (int c = 0; c < C; ++c) {
        const v4f v = *reinterpret_cast<const v4f*>(xb + (size_t)c * S);
        const float wc = w[c];
        acc[0] = fmaf(v[0], wc, acc[0]);
        acc[1] = fmaf(v[1], wc, acc[1]);
        acc[2] = fmaf(v[2], wc, acc[2]);
        acc[3] = fmaf(v[3], wc, acc[3]);
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) acc[k] = apply_act(fmaf(acc[k], sc, sf), act);
    *reinterpret_cast<v4f*>(y + (size_t)b * S + sp) = acc;
}

hipError_t launch_head(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int C,
                       int64_t S, int act, hipStream_t s) {
    const long long total4 = (long long)B * (S >> 2);
    hipLaunchKernelGGL(head_kernel, dim3((unsigned)((total4 + 255) / 256)), dim3(256), 0, s, x, w, scale, shift, y, C,
                       (long long)S, act, total4);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Point-head linear layer  y[b][o] = act(sum_i x[b][i] * w[o][i] + bias[o]),  weight-streaming bound
// (p1 alone is 134 MB of fp32 weights for 2 GFLOP at B=32).  Split-K over blockIdx.y so that >=256
// workgroups stream disjoint weight slabs; partial sums are combined with fp32 atomics into a
// zeroed accumulator and finished (bias + activation) by a second tiny kernel.
constexpr int LIN_TO = 32, LIN_TB = 32, LIN_TK = 64;

__global__ __launch_bounds__(256) void linear_partial_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                             float* __restrict__ acc_out, int B, int Cin, int Cout,
                                                             int kper) {
    __shared__ float xs[LIN_TB][LIN_TK + 1];
    __shared__ float ws[LIN_TO][LIN_TK + 1];
    const int o0 = blockIdx.x * LIN_TO, b0 = blockIdx.z * LIN_TB;
    const int k_begin = blockIdx.y * kper;
    const int k_end = min(Cin, k_begin + kper);
    const int tid = threadIdx.x;
    const int tb = tid >> 3, to = tid & 7;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    for (int k0 = k_begin; k0 < k_end; k0 += LIN_TK) {
#pragma unroll
        for (int i = 0; i < (LIN_TB * LIN_TK) / 256; ++i) {
            const int e = tid + i * 256;
            const int r = e / LIN_TK, cidx = e % LIN_TK;
            const int k = k0 + cidx;
            xs[r][cidx] = (b0 + r < B && k < k_end) ? x[(size_t)(b0 + r) * Cin + k] : 0.f;
            ws[r][cidx] = (o0 + r < Cout && k < k_end) ? w[(size_t)(o0 + r) * Cin + k] : 0.f;
        }
        __syncthreads();
#pragma unroll 16
        for (int i = 0; i < LIN_TK; ++i) {
            const float xv = xs[tb][i];
#pragma unroll
            for (int q = 0; q < 4; ++q) acc[q] = fmaf(xv, ws[to + 8 * q][i], acc[q]);
        }
        __syncthreads();
    }
    const int b = b0 + tb;
    if (b < B) {
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const int o = o0 + to + 8 * q;
            if (o < Cout) atomicAdd(acc_out + (size_t)b * Cout + o, acc[q]);
        }
    }
}

__global__ void linear_finish_kernel(float* __restrict__ y, const float* __restrict__ scale,
                                     const float* __restrict__ bias, int Cout, long long total, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i < total) {
        const int o = (int)(i % Cout);
        y[i] = apply_act(fmaf(y[i], scale ? scale[o] : 1.f, bias ? bias[o] : 0.f), act);
    }
}

hipError_t launch_linear(const float* x, const float* w, const float* scale, const float* bias, float* y, int B,
                         int Cin, int Cout, int act, hipStream_t s) {
    hipError_t e = hipMemsetAsync(y, 0, (size_t)B * Cout * sizeof(float), s);
    if (e != hipSuccess) return e;
    const int otiles = (Cout + LIN_TO - 1) / LIN_TO, btiles = (B + LIN_TB - 1) / LIN_TB;
    int ksplit = 1;
    while (otiles * btiles * ksplit < 512 && Cin / (ksplit * 2) >= 4 * LIN_TK) ksplit *= 2;
    int kper = (Cin + ksplit - 1) / ksplit;
    kper = (kper + LIN_TK - 1) / LIN_TK * LIN_TK;
    ksplit = (Cin + kper - 1) / kper;
    hipLaunchKernelGGL(linear_partial_kernel, dim3(otiles, ksplit, btiles), dim3(256), 0, s, x, w, y, B, Cin, Cout,
                       kper);
    const long long total = (long long)B * Cout;
    hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, y, scale, bias,
                       Cout, total, act);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Thresholded voxel IoU per sample (eval collation gathers B scalars instead of B*32^3 floats).
// Wavefront shuffles (64 lanes) reduce the two counts; one workgroup per sample.
__device__ __forceinline__ unsigned wave_sum(unsigned v) {
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) v += __shfl_xor(v, o, 64);
    return v;
}

__global__ __launch_bounds__(256) void iou_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                  float th, float* __restrict__ iou, long long S) {
    __shared__ unsigned part[2][4];
    const float* __restrict__ p = pred + (size_t)blockIdx.x * S;
    const float* __restrict__ g = gt + (size_t)blockIdx.x * S;
    unsigned inter = 0, uni = 0;
    for (long long i = threadIdx.x; i < S; i += 256) {
        const bool a = p[i] > th, b = g[i] > th;
        inter += (a && b);
        uni += (a || b);
    }
    inter = wave_sum(inter);
    uni = wave_sum(uni);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { part[0][wave] = inter; part[1][wave] = uni; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned i4 = part[0][0] + part[0][1] + part[0][2] + part[0][3];
        const unsigned u4 = part[1][0] + part[1][1] + part[1][2] + part[1][3];
        iou[blockIdx.x] = u4 ? (float)i4 / (float)u4 : 1.f;
    }
}

hipError_t launch_iou(const float* pred, const float* gt, float th, float* iou, int B, int64_t S, hipStream_t s) {
    hipLaunchKernelGGL(iou_kernel, dim3(B), dim3(256), 0, s, pred, gt, th, iou, (long long)S);
    return hipGetLastError();
}

}  // namespace s3r
