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
//   images [0, nsplit) come from x, images [nsplit, N) from x2 (left / right renders: no concatenation copy)
//   TI = float: fp32 renders in [0,1]; TI = unsigned char: 8-bit renders, scaled by 1/255 on the way in (render_f32)
template <typename TI>
__global__ __launch_bounds__(256) void stem_kernel(const TI* __restrict__ x, const TI* __restrict__ x2, int nsplit,
                                                   const float* __restrict__ wt,
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
    const TI* __restrict__ xn = n < nsplit ? x + (size_t)n * 3 * Hi * Wi : x2 + (size_t)(n - nsplit) * 3 * Hi * Wi;

    // accumulators in pairs: the channel loop compiles to v_pk_fma_f32 (two exact fp32 FMAs per lane per instruction,
    // the weight pair straight from SGPRs) — half the VALU instructions of the scalar form, same bits
    typedef float f32x2 __attribute__((ext_vector_type(2)));
    f32x2 acc2[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) acc2[c] = (f32x2){0.f, 0.f};
    // The (channel, row) loops stay ROLLED: fully unrolled, all 864 weights become live scalar loads at once, do not
    // fit the SGPR file, and were spilled through v_writelane / v_readlane (1536 extra instructions per thread).
    // Rolled, one iteration holds the 96 weights of its three taps.
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const int ih = ih0 + kh;
            const bool vh = (unsigned)ih < (unsigned)Hi;
            const TI* __restrict__ xrow = xn + ((size_t)ci * Hi + ih) * Wi;
            const float* __restrict__ wrow = wt + (ci * 3 + kh) * 96;
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const int iw = iw0 + kw;
                const bool v = vh && ((unsigned)iw < (unsigned)Wi);
                const float xv = v ? render_f32(xrow[iw]) : 0.f;
                const f32x2 xv2 = {xv, xv};
#pragma unroll
                for (int c = 0; c < 16; ++c)
                    acc2[c] = __builtin_elementwise_fma(xv2, *reinterpret_cast<const f32x2*>(wrow + kw * 32 + 2 * c), acc2[c]);
            }
        }
    }
    float acc[32];
#pragma unroll
    for (int c = 0; c < 16; ++c) { acc[2 * c] = acc2[c].x; acc[2 * c + 1] = acc2[c].y; }
    float* __restrict__ yn = y + (size_t)n * 32 * y_cs + y_org + oh * y_hs + ow;   // (halo-padded) NCHW
#pragma unroll
    for (int c = 0; c < 32; ++c) yn[(size_t)c * y_cs] = fmaxf(fmaf(acc[c], scale[c], shift[c]), 0.f);
}

// The stem written straight into the layout its consumer's one-axis Winograd kernel reads (S3R_LAYOUT_WINO_H: V_i[n][c][q][wp],
// i = 0 .. 5, from the padded rows 4 q .. 4 q + 5 of the halo-1 activation through wino_rows_to_classes): the plain activation
// (106 MB at 64 renders) is neither written nor re-read by a transform kernel.  One workgroup per (image, row group q): the 13
// input rows 8 q - 3 .. 8 q + 9 of the three channels go to LDS once (zeros outside the image, even and odd columns apart so
// that a wave's reads are stride-1), then a thread per (half of the couts, padded column) computes the six rows of its column
// with the plain kernel's summation order per value — the bits wino_input_kernel makes of the plain stem's output — and
// transforms them.  Every output row is computed 1.5 times (groups overlap by two rows).  Single v_fma_f32 here, not the plain
// kernel's packed pairs (v_pk_fma_f32 issues about three times slower per FMA on gfx950: MI355X_MICROARCH.md, row 'F32').
// 59 us at 64 renders (stores alone 26, staging 9, the FMAs 40, overlapped) against 37 + 45 for plain stem + transform.
constexpr int STEM_W_ROWS = 13;
template <typename TI>
__global__ __launch_bounds__(256) void stem_wino_kernel(const TI* __restrict__ x, const TI* __restrict__ x2, int nsplit,
                                                        const float* __restrict__ wt, const float* __restrict__ scale,
                                                        const float* __restrict__ shift, float* __restrict__ V, int Hi, int Wi,
                                                        int Ho, int Wo, long long cls_stride) {
    extern __shared__ __attribute__((aligned(16))) float stem_sx[];      // [3][13][2][LH]: even columns, then odd, of iw + 3
    const int Wp = Wo + 2, HQ = Ho / 4;
    const int LW = 2 * Wp + 4, LH = LW / 2;              // columns iw = -3 .. 2 Wp (a thread reads 2 wp - 3 .. 2 wp - 1)
    const int n = blockIdx.x / HQ, q = blockIdx.x - n * HQ;
    const int tid = threadIdx.x;
    const TI* __restrict__ xn = n < nsplit ? x + (size_t)n * 3 * Hi * Wi : x2 + (size_t)(n - nsplit) * 3 * Hi * Wi;
    // staging: 16-byte pieces of the 39 rows, every load of a thread issued before the first LDS store (a load-then-store loop
    // with a runtime trip count compiles to one memory round trip per iteration: 36 in series, 30 us of this kernel)
    constexpr int E = 16 / (int)sizeof(TI);              // samples per piece
    constexpr int NV = sizeof(TI) == 4 ? 9 : 3;          // pieces per thread (Wi <= 236 / 304: the launcher checks)
    typedef TI piece_t __attribute__((ext_vector_type(E)));
    const int ppr = Wi / E;                              // pieces per row
    piece_t pv[NV];
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = tid + u * 256;
        const int r = i / ppr, c = i - r * ppr;
        const int ci = r / STEM_W_ROWS, ih = 8 * q - 3 + (r - ci * STEM_W_ROWS);
        const bool v = r < 3 * STEM_W_ROWS && (unsigned)ih < (unsigned)Hi;
        pv[u] = *reinterpret_cast<const piece_t*>(xn + ((size_t)(v ? ci : 0) * Hi + (v ? ih : 0)) * Wi + (v ? c : 0) * E);
        if (!v) pv[u] = piece_t{};
    }
    for (int i = tid; i < 3 * STEM_W_ROWS * (LW - Wi); i += 256) {       // the columns outside the image: iw = -3 .. -1, Wi ..
        const int r = i / (LW - Wi), jj = i - r * (LW - Wi);
        const int j = jj < 3 ? jj : Wi + jj;
        stem_sx[r * LW + (j & 1) * LH + (j >> 1)] = 0.f;
    }
#pragma unroll
    for (int u = 0; u < NV; ++u) {
        const int i = tid + u * 256;
        const int r = i / ppr, c = i - r * ppr;
        if (r < 3 * STEM_W_ROWS) {
#pragma unroll
            for (int e = 0; e < E; ++e) {
                const int j = c * E + e + 3;
                stem_sx[r * LW + (j & 1) * LH + (j >> 1)] = render_f32(pv[u][e]);
            }
        }
    }
    __syncthreads();
    const int half = __builtin_amdgcn_readfirstlane(tid >> 7);           // waves 0, 1: couts 0 .. 15; waves 2, 3: 16 .. 31
    const int wl = tid & 127;
    const int wp = wl < Wp ? wl : Wp - 1;                // (lanes past the row compute its last column and store nothing)
    const bool col_in = (unsigned)(wp - 1) < (unsigned)Wo;
    float sc[16], sf[16];
#pragma unroll
    for (int c = 0; c < 16; ++c) { sc[c] = scale[half * 16 + c]; sf[c] = shift[half * 16 + c]; }
    // The (channel, row) loops stay ROLLED, as in the plain kernel: one iteration holds the 48 weights of its three taps — and
    // spends them on all six rows (a value's own summation order is untouched by what is interleaved with it)
    float rows[6][16];
#pragma unroll
    for (int k = 0; k < 6; ++k)
#pragma unroll
        for (int c = 0; c < 16; ++c) rows[k][c] = 0.f;
#pragma unroll 1
    for (int ci = 0; ci < 3; ++ci) {
#pragma unroll 1
        for (int kh = 0; kh < 3; ++kh) {
            const float* __restrict__ wrow = wt + (ci * 3 + kh) * 96 + half * 16;
            float xk[6][3];
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                const float* __restrict__ xr = stem_sx + (ci * STEM_W_ROWS + 2 * k + kh) * LW + wp;      // input row 2 oh - 1 + kh
                xk[k][0] = xr[0]; xk[k][1] = xr[LH]; xk[k][2] = xr[1];   // columns 2 ow - 1, 2 ow, 2 ow + 1
            }
#pragma unroll
            for (int kw = 0; kw < 3; ++kw)
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const float wv = wrow[kw * 32 + c];
#pragma unroll
                    for (int k = 0; k < 6; ++k) rows[k][c] = fmaf(xk[k][kw], wv, rows[k][c]);
                }
        }
    }
#pragma unroll
    for (int k = 0; k < 6; ++k) {
        const int oh = 4 * q + k - 1;                    // padded row 4 q + k
        const bool in = col_in && (unsigned)oh < (unsigned)Ho;
#pragma unroll
        for (int c = 0; c < 16; ++c) {
            const float a = fmaxf(fmaf(rows[k][c], sc[c], sf[c]), 0.f);
            rows[k][c] = in ? a : 0.f;
        }
    }
    if (wl >= Wp) return;
    float* __restrict__ o = V + (((size_t)n * 32 + half * 16) * HQ + q) * Wp + wp;
#pragma unroll
    for (int c = 0; c < 16; ++c) {
        float r[6], v[6];
#pragma unroll
        for (int k = 0; k < 6; ++k) r[k] = rows[k][c];
        wino_rows_to_classes<4>(r, v);
#pragma unroll
        for (int k = 0; k < 6; ++k) o[(size_t)k * cls_stride + (size_t)c * HQ * Wp] = v[k];
    }
}

hipError_t launch_stem_wino(const void* x, const void* x2, int u8, int nsplit, const float* wt, const float* scale, const float* shift,
                            float* V, int N, int Hi, int Wi, int Ho, int Wo, hipStream_t s) {
    const int Wp = Wo + 2, HQ = Ho / 4;
    const size_t lds = (size_t)3 * STEM_W_ROWS * (2 * Wp + 4) * sizeof(float);
    const int pieces = 3 * STEM_W_ROWS * (Wi / (u8 ? 16 : 4));
    if (Ho % 4 != 0 || Wp > 128 || Wi != 2 * Wo || Hi != 2 * Ho || lds > 64 * 1024 || Wi % 16 != 0 || pieces > 256 * (u8 ? 3 : 9))
        return hipErrorInvalidValue;
    if (!x2) { x2 = x; nsplit = N; }
    const long long cls_stride = (long long)N * 32 * HQ * Wp;
    const dim3 grid((unsigned)(N * HQ));
    if (u8)
        hipLaunchKernelGGL(stem_wino_kernel<unsigned char>, grid, dim3(256), lds, s, static_cast<const unsigned char*>(x),
                           static_cast<const unsigned char*>(x2), nsplit, wt, scale, shift, V, Hi, Wi, Ho, Wo, cls_stride);
    else
        hipLaunchKernelGGL(stem_wino_kernel<float>, grid, dim3(256), lds, s, static_cast<const float*>(x),
                           static_cast<const float*>(x2), nsplit, wt, scale, shift, V, Hi, Wi, Ho, Wo, cls_stride);
    return hipGetLastError();
}

__global__ void pack_stem_kernel(const float* __restrict__ w, float* __restrict__ wt) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;   // over 27*32
    if (i < 27 * 32) {
        const int c = i & 31, k = i >> 5;
        wt[i] = w[c * 27 + k];
    }
}

hipError_t launch_stem(const void* x, const void* x2, int u8, int nsplit, const float* wt, const float* scale,
                       const float* shift, float* y, int N, int Hi, int Wi, int Ho, int Wo, int y_cs, int y_hs, int y_org,
                       hipStream_t s) {
    const long long total = (long long)N * Ho * Wo;
    if (!x2) { x2 = x; nsplit = N; }
    const dim3 grid((unsigned)((total + 255) / 256));
    if (u8)
        hipLaunchKernelGGL(stem_kernel<unsigned char>, grid, dim3(256), 0, s, static_cast<const unsigned char*>(x),
                           static_cast<const unsigned char*>(x2), nsplit, wt, scale, shift, y, N, Hi, Wi, Ho, Wo, y_cs, y_hs, y_org);
    else
        hipLaunchKernelGGL(stem_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(x),
                           static_cast<const float*>(x2), nsplit, wt, scale, shift, y, N, Hi, Wi, Ho, Wo, y_cs, y_hs, y_org);
    return hipGetLastError();
}

hipError_t launch_pack_stem(const float* w, float* wt, hipStream_t s) {
    hipLaunchKernelGGL(pack_stem_kernel, dim3(4), dim3(256), 0, s, w, wt);
    return hipGetLastError();
}

// ------------------------------------------------------------------------------------------------
// Cost volume: vol[b, c,   d, h, w] = L[b,c,h,w] - R[b,c,h,w-d]   (0 where w-d < 0)
//              vol[b, C+c, d, h, w] = R[b,c,h,w] - L[b,c,h,w+d]   (0 where w+d >= W)
// One workgroup per (b, c): both HxW planes are read from HBM exactly once into LDS, then the two output slabs
// are streamed out with 16-byte stores.  The volume may carry a zero halo (`halo` elements on the d, h and w
// axes) for the 3D conv that consumes it.  With a halo the interior rows are W-float segments inside (W+2h)-float
// rows: written alone they leave a partial 128-byte line at almost every row end (2.7 TB/s, r01).  So the kernel
// writes WHOLE padded planes d = halo .. halo+D-1 — halo rows and columns included, as the zeros they are — which
// makes each slab ONE contiguous run of D*Hp*Wp floats: every store instruction covers whole lines.  (The halo
// PLANES d < halo, d >= halo+D are never written: they keep the zeros of the buffer's one-time memset.)
// Algorithmic bytes = 4*(2*H*W + 2*D*H*W) per (b,c); the kernel moves 4*(2*H*W + 2*D*Hp*Wp).
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned 16-byte access

__global__ __launch_bounds__(256) void cost_volume_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                          float* __restrict__ vol, int C, int D, int H, int W,
                                                          int halo, FastDiv dPlane, FastDiv dRow) {
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
    const int Wp = W + 2 * halo, Hp = H + 2 * halo, Dp = D + 2 * halo;
    const int ds = Hp * Wp;
    const size_t cs = (size_t)Dp * ds;
    const int run = D * ds;                              // floats of one slab's contiguous run (planes halo .. halo+D-1)
    float* __restrict__ ol = vol + ((size_t)b * 2 * C + c) * cs + (size_t)halo * ds;
    float* __restrict__ orr = vol + ((size_t)b * 2 * C + C + c) * cs + (size_t)halo * ds;
    const int nq = run >> 2;
    for (int q = threadIdx.x; q < nq; q += 256) {
        const int e = q << 2;
        int d = dPlane.div(e);                           // e / (Hp*Wp)
        const int r = e - d * ds;
        int hp = dRow.div(r);                            // r / Wp
        int wp = r - hp * Wp;
        v4f a, rr;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int h = hp - halo, w = wp - halo;
            const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            const int hw = h * W + w;
            a[k] = (in && w >= d) ? sl[hw] - sr[hw - d] : 0.f;
            rr[k] = (in && w + d < W) ? sr[hw] - sl[hw + d] : 0.f;
            if (++wp == Wp) { wp = 0; if (++hp == Hp) { hp = 0; ++d; } }
        }
        *reinterpret_cast<v4f_u*>(ol + e) = a;
        *reinterpret_cast<v4f_u*>(orr + e) = rr;
    }
    for (int e = (nq << 2) + threadIdx.x; e < run; e += 256) {      // run % 4 != 0 (never at the network's shapes)
        const int d = dPlane.div(e);
        const int r = e - d * ds;
        const int hp = dRow.div(r);
        const int h = hp - halo, w = r - hp * Wp - halo;
        const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
        const int hw = h * W + w;
        ol[e] = (in && w >= d) ? sl[hw] - sr[hw - d] : 0.f;
        orr[e] = (in && w + d < W) ? sr[hw] - sl[hw + d] : 0.f;
    }
}

// The cost volume written straight into the layout its consumer's Winograd kernel reads (s3r_conv_wino.hip, S3R_LAYOUT_WINO_H):
// V_i[b][ch][1 + d][q][wp], i = 0 .. R+1, from the padded rows R q .. R q + R + 1 of plane d of the halo-1 volume through
// wino_rows_to_classes (the very function wino_input_kernel applies to the padded volume: bit-identical) — the volume itself (221 MB at B = 32) is never written
// or re-read.  One workgroup per (b, c) as above; per class and slab the planes d = 0 .. D-1 are one contiguous run.
template <int R>
__global__ __launch_bounds__(256) void cost_volume_wino_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                               float* __restrict__ V, int C, int D, int H, int W,
                                                               long long cls_stride, FastDiv dPlane, FastDiv dRow) {
    extern __shared__ __attribute__((aligned(16))) float cvw_smem[];
    const int HW = H * W;
    float* sl = cvw_smem;
    float* sr = cvw_smem + HW;
    const int bc = blockIdx.x;
    const int b = bc / C, c = bc - b * C;
    const float* __restrict__ pl = fl + (size_t)bc * HW;
    const float* __restrict__ pr = fr + (size_t)bc * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        sl[i] = pl[i];
        sr[i] = pr[i];
    }
    __syncthreads();
    const int Wp = W + 2, Hq = H / R, Dp = D + 2;
    const int ds = Hq * Wp;                              // one plane of a class
    const size_t cs = (size_t)Dp * ds;
    const int run = D * ds;
    float* __restrict__ ol = V + ((size_t)b * 2 * C + c) * cs + ds;            // plane dp = 1
    float* __restrict__ orr = V + ((size_t)b * 2 * C + C + c) * cs + ds;
    for (int e = threadIdx.x; e < run; e += 256) {
        const int d = dPlane.div(e);                     // e / (Hq*Wp)
        const int rem = e - d * ds;
        const int q = dRow.div(rem);                     // rem / Wp
        const int wp = rem - q * Wp;
        float a[R + 2], r[R + 2], va[R + 2], vr[R + 2];
#pragma unroll
        for (int k = 0; k < R + 2; ++k) {                // the padded volume at (d, R q + k, wp), both slabs
            const int h = R * q + k - 1, w = wp - 1;
            const bool in = (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
            const int hw = h * W + w;
            a[k] = (in && w >= d) ? sl[hw] - sr[hw - d] : 0.f;
            r[k] = (in && w + d < W) ? sr[hw] - sl[hw + d] : 0.f;
        }
        wino_rows_to_classes<R>(a, va);
        wino_rows_to_classes<R>(r, vr);
#pragma unroll
        for (int k = 0; k < R + 2; ++k) {
            ol[e + k * cls_stride] = va[k];
            orr[e + k * cls_stride] = vr[k];
        }
    }
}

hipError_t launch_cost_volume_wino(const float* fl, const float* fr, float* V, int B, int C, int D, int H, int W, int R, hipStream_t s) {
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    const int Wp = W + 2, Hq = H / R;
    const long long cls_stride = (long long)B * 2 * C * (D + 2) * Hq * Wp;
    if (R != 4) return hipErrorInvalidValue;              // F(4,3) groups
    hipLaunchKernelGGL(cost_volume_wino_kernel<4>, dim3(B * C), dim3(256), lds, s, fl, fr, V, C, D, H, W, cls_stride,
                       FastDiv((unsigned)(Hq * Wp)), FastDiv((unsigned)Wp));
    return hipGetLastError();
}

// ... and as the 36 TWO-AXIS plane sets (F(4,3) along D and H: s3r_conv_wino.hip, wino2_input_kernel, S3R_LAYOUT_WINO_DH):
// V[6 a + b][bb][ch][sd][sh][wp] from the 6 x 6 window of padded depths 4 sd .. 4 sd + 5 and padded rows 4 sh .. 4 sh + 5 at column
// wp of the halo-1 volume — rows first, then depths, through the same wino_rows_to_classes: bit-identical to the transform
// kernel applied to the padded volume.  One workgroup per (b, c); a thread per (sd, sh, wp) and slab half.  Small batches
// (gridDim.y = 2 or 2 SD): a workgroup per (b, c, slab half[, depth group]) — 32 workgroups of 12 serial passes are 46 us at B = 1.
__global__ __launch_bounds__(256) void cost_volume_wino2_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                                float* __restrict__ V, int C, int D, int H, int W,
                                                                long long cls_stride, FastDiv dPlane, FastDiv dRow) {
    extern __shared__ __attribute__((aligned(16))) float cvw2_smem[];
    const int HW = H * W;
    float* sl = cvw2_smem;
    float* sr = cvw2_smem + HW;
    const int bc = blockIdx.x;
    const int b = bc / C, c = bc - b * C;
    const float* __restrict__ pl = fl + (size_t)bc * HW;
    const float* __restrict__ pr = fr + (size_t)bc * HW;
    for (int i = threadIdx.x; i < HW; i += 256) {
        sl[i] = pl[i];
        sr[i] = pr[i];
    }
    __syncthreads();
    const int Wp = W + 2, SH = H / 4, SD = (D + 3) / 4;
    const int ds = SH * Wp;                              // one depth group of a class
    const int run = SD * ds;
    const int ny = (int)gridDim.y, pph = ny > 1 ? ny / 2 : 1;           // parts per slab half
    const int half0 = ny > 1 ? (int)blockIdx.y / pph : 0, half1 = ny > 1 ? half0 + 1 : 2;
    const int part = ny > 1 ? (int)blockIdx.y - half0 * pph : 0;
    const int e0 = (part * SD / pph) * ds, e1 = ((part + 1) * SD / pph) * ds;
    for (int half = half0; half < half1; ++half) {       // the L - R(shifted) slab, then the R - L(shifted) slab
        float* __restrict__ o = V + ((size_t)b * 2 * C + half * C + c) * run;
        for (int e = e0 + threadIdx.x; e < e1; e += 256) {
            const int sd = dPlane.div(e);                // e / (SH*Wp)
            const int rem = e - sd * ds;
            const int sh = dRow.div(rem);                // rem / Wp
            const int wp = rem - sh * Wp;
            const int w = wp - 1;
            float t[6][6];                               // [depth][row class]
#pragma unroll
            for (int a = 0; a < 6; ++a) {
                const int d = 4 * sd + a - 1;            // padded depth 4 sd + a
                float r[6], v[6];
#pragma unroll
                for (int k = 0; k < 6; ++k) {
                    const int h = 4 * sh + k - 1;
                    const bool in = (unsigned)d < (unsigned)D && (unsigned)h < (unsigned)H && (unsigned)w < (unsigned)W;
                    const int hw = h * W + w;
                    r[k] = half == 0 ? ((in && w >= d) ? sl[hw] - sr[hw - d] : 0.f) : ((in && w + d < W) ? sr[hw] - sl[hw + d] : 0.f);
                }
                wino_rows_to_classes<4>(r, v);
#pragma unroll
                for (int k = 0; k < 6; ++k) t[a][k] = v[k];
            }
#pragma unroll
            for (int k = 0; k < 6; ++k) {
                float r[6], v[6];
#pragma unroll
                for (int a = 0; a < 6; ++a) r[a] = t[a][k];
                wino_rows_to_classes<4>(r, v);
#pragma unroll
                for (int a = 0; a < 6; ++a) o[e + (size_t)(a * 6 + k) * cls_stride] = v[a];
            }
        }
    }
}

hipError_t launch_cost_volume_wino2(const float* fl, const float* fr, float* V, int B, int C, int D, int H, int W, hipStream_t s) {
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    const int Wp = W + 2, SH = H / 4, SD = (D + 3) / 4;
    const long long cls_stride = (long long)B * 2 * C * SD * SH * Wp;
    const int ny = B * C <= 128 ? 2 * SD : B * C <= 512 ? 2 : 1;      // (same bits: the split only re-distributes elements)
    hipLaunchKernelGGL(cost_volume_wino2_kernel, dim3(B * C, ny), dim3(256), lds, s, fl, fr, V, C, D, H, W, cls_stride,
                       FastDiv((unsigned)(SH * Wp)), FastDiv((unsigned)Wp));
    return hipGetLastError();
}

hipError_t launch_cost_volume(const float* fl, const float* fr, float* vol, int B, int C, int D, int H, int W,
                              int halo, hipStream_t s) {
    const size_t lds = (size_t)2 * H * W * sizeof(float);
    const int Wp = W + 2 * halo, Hp = H + 2 * halo;
    hipLaunchKernelGGL(cost_volume_kernel, dim3(B * C), dim3(256), lds, s, fl, fr, vol, C, D, H, W, halo,
                       FastDiv((unsigned)(Hp * Wp)), FastDiv((unsigned)Wp));
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
// Point-head linear layer  y[b][o] = act(scale[o] * sum_i x[b][i] * w[o][i] + bias[o]).
// HBM-bound weight streaming (p1 alone is 134 MB of fp32 weights for 2 GFLOP at B=32), so the design goal is "every
// weight byte crosses HBM once, in whole 128-byte lines, with enough of them in flight":
//   * a workgroup owns 128 output rows x 32 batch rows x one K slice and walks the slice in blocks of 32 k = one
//     128-byte line per row; per block the 128 weight lines (16 KiB) and the 32 activation lines (4 KiB, shared by the
//     four waves: x is re-read from L2 once per 128 outputs instead of once per 32) arrive by LDS-DMA, 16 bytes per
//     lane, EIGHT WHOLE LINES per wave-instruction (r01 loaded fragment-shaped: 16 bytes of each of 32 rows per
//     instruction, 1.5 TB/s), into a 4-deep LDS ring: three blocks = 60 KiB per workgroup stay in flight behind a
//     counted vmcnt and a raw s_barrier (a barrier the compiler knows about drains vmcnt);
//   * LDS rows are 128 bytes; the 16-byte chunk c of row r sits at slot c ^ ((r >> 1) & 7) (applied on the DMA's
//     per-lane SOURCE address: LDS-DMA destinations are lane-linear), which makes the fragment reads — 32 consecutive
//     rows at one chunk per lane half — conflict-free;
//   * wave w multiplies its 32 output rows: v_mfma_f32_32x32x2_f32, A = x (lane (i,h): batch row i), B = w (lane (j,h):
//     output row j), both as ds_read_b128 of chunk 2t + h: 16 MFMAs (1024 cycles) per 4 KiB of weights per wave keeps
//     the matrix pipe at a 9.8 TB/s equivalent, so the kernel stays memory-bound;
//   * split-K over workgroups for parallelism; partial slabs [kz][b][o] are reduced in kz order by
//     linear_finish_kernel (deterministic: no atomics).
typedef float f32x16_l __attribute__((ext_vector_type(16)));
#define S3R_LDS_PTR_PW(p) ((__attribute__((address_space(3))) void*)(p))

constexpr int LIN_NS = 4;                     // ring slots
constexpr int LIN_STAGE = (128 + 32) * 128;   // bytes per slot: 128 weight rows + 32 activation rows of 128 B

template <bool NT>
__global__ __launch_bounds__(256, 2) void linear_stream_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                               float* __restrict__ part, int B, int Cin, int Cout, int kper,
                                                               int ksplit) {
    extern __shared__ __attribute__((aligned(16))) char lin_smem[];       // [LIN_NS][160 rows][128 B]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 128;
    const int kz = blockIdx.y;
    const int b0 = blockIdx.z * 32;
    const int kbeg = kz * kper;
    const int nblk = (min(Cin, kbeg + kper) - kbeg) >> 5;

    // ---- loop-invariant DMA source offsets (bytes): a piece = 8 rows x 128 B; lane l fills slot l & 7 of row l >> 3
    //      with source chunk (l & 7) ^ ((row >> 1) & 7); rows past the tensor re-read its last row (never stored)
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)0x7fffffff, 0x00020000);
    int wvoff[4], xvoff;
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = (wave * 4 + q) * 8 + (lane >> 3);                  // row of the workgroup's 128-row weight tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        wvoff[q] = (int)(((long long)min(o0 + r, Cout - 1) * Cin + kbeg) * 4 + c * 16);      // (< 2^31: checked by the launcher)
    }
    {
        const int r = wave * 8 + (lane >> 3);                            // row of the 32-row activation tile
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        xvoff = (int)(((long long)min(b0 + r, B - 1) * Cin + kbeg) * 4 + c * 16);
    }
    auto issue = [&](int blk) {
        char* st = lin_smem + (blk & (LIN_NS - 1)) * LIN_STAGE;
        const int so = blk * 128;                                        // 32 floats further along every row
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, S3R_LDS_PTR_PW(st + (wave * 4 + q) * 1024), 16, wvoff[q], so, 0,
                                                     NT ? 2 : 0);
        __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, S3R_LDS_PTR_PW(st + 128 * 128 + wave * 1024), 16, xvoff, so, 0, 0);
    };

    f32x16_l acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    // fragment offsets inside a slot: row * 128 + ((2t + h) ^ ((row >> 1) & 7)) * 16; t toggles bits 5..6
    const int wrow = wave * 32 + j;
    const int w_off = wrow * 128 + ((h ^ ((wrow >> 1) & 7)) << 4);
    const int x_off = 128 * 128 + j * 128 + ((h ^ ((j >> 1) & 7)) << 4);

    for (int s0 = 0; s0 < LIN_NS - 1; ++s0)
        if (s0 < nblk) issue(s0);
    for (int blk = 0; blk < nblk; ++blk) {
        // block `blk` has landed when at most the DMAs of the (up to two) later blocks in flight are outstanding
        const int later = min(nblk - 1 - blk, LIN_NS - 2);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(10)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        asm volatile("s_barrier" ::: "memory");           // every wave's part of it has; slot (blk - 1) & 3 is free
        if (blk + LIN_NS - 1 < nblk) issue(blk + LIN_NS - 1);
        const char* st = lin_smem + (blk & (LIN_NS - 1)) * LIN_STAGE;
        v4f xv[4], wv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xv[t] = *reinterpret_cast<const v4f*>(st + (x_off ^ (t << 5)));
            wv[t] = *reinterpret_cast<const v4f*>(st + (w_off ^ (t << 5)));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][e], wv[t][e], acc, 0, 0, 0);
    }
    const int o = o0 + wave * 32 + j;
    if (o < Cout) {
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (b < B) part[((size_t)kz * B + b) * Cout + o] = acc[r];
        }
    }
}

// generic (any Cin) partial kernel: one thread per (b, o) of a K slice; used when Cin % 32 != 0
__global__ __launch_bounds__(256) void linear_naive_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                           float* __restrict__ part, int B, int Cin, int Cout) {
    const long long i = (long long)blockIdx.x * 256 + threadIdx.x;
    if (i >= (long long)B * Cout) return;
    const int b = (int)(i / Cout), o = (int)(i % Cout);
    float s = 0.f;
    for (int k = 0; k < Cin; ++k) s = fmaf(x[(size_t)b * Cin + k], w[(size_t)o * Cin + k], s);
    part[i] = s;
}

// The same for deep splits (ksplit % 4 == 0, >= 16): a workgroup owns 64 outputs and its four waves a quarter of the slabs
// each (wave w: kz in [w q, (w+1) q), q = ksplit / 4, summed in kz order), combined as ((s0 + s1) + s2) + s3 through LDS —
// a fixed order, and a quarter of the dependent L2 round trips per thread (p1: 64 slabs, 6.5 us with one thread per output).
__global__ __launch_bounds__(256) void linear_finish4_kernel(const float* __restrict__ part, float* __restrict__ y,
                                                             const float* __restrict__ scale, const float* __restrict__ bias,
                                                             int Cout, long long total, int ksplit, int act) {
    __shared__ float red[3][64];
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    const long long i = (long long)blockIdx.x * 64 + lane;
    const int q = ksplit >> 2;
    float s = 0.f;
    if (i < total) {
        const float* __restrict__ src = part + (size_t)w * q * total + i;
        s = src[0];
        int z = 1;
        for (; z + 8 <= q; z += 8) {
            float t[8];
#pragma unroll
            for (int u = 0; u < 8; ++u) t[u] = src[(size_t)(z + u) * total];
#pragma unroll
            for (int u = 0; u < 8; ++u) s += t[u];
        }
        for (; z < q; ++z) s += src[(size_t)z * total];
    }
    if (w) red[w - 1][lane] = s;
    __syncthreads();
    if (w == 0 && i < total) {
        s = ((s + red[0][lane]) + red[1][lane]) + red[2][lane];
        const int o = (int)(i % Cout);
        y[i] = apply_act(fmaf(s, scale ? scale[o] : 1.f, bias ? bias[o] : 0.f), act);
    }
}

// one thread per output; the ksplit partial slabs are summed in kz order (a fixed order: deterministic) with eight loads
// in flight per thread (a one-load-at-a-time loop is a chain of ksplit L2 round trips: 64 of them for p1)
__global__ __launch_bounds__(256) void linear_finish_kernel(const float* __restrict__ part, float* __restrict__ y,
                                                            const float* __restrict__ scale, const float* __restrict__ bias,
                                                            int Cout, long long total, int ksplit, int act) {
    const long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int o = (int)(i % Cout);
    float s = part[i];
    int z = 1;
    for (; z + 8 <= ksplit; z += 8) {
        float t[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) t[u] = part[(size_t)(z + u) * total + i];
#pragma unroll
        for (int u = 0; u < 8; ++u) s += t[u];
    }
    for (; z < ksplit; ++z) s += part[(size_t)z * total + i];
    y[i] = apply_act(fmaf(s, scale ? scale[o] : 1.f, bias ? bias[o] : 0.f), act);
}

// Small-K layers (p2, p3: Cin = 1024) in ONE launch.  Their activations are tiny (131 KB), so a workgroup can afford to
// own just 32 output rows (Cout / 32 workgroups: 32 for p2, 192 for p3) and split K over its FOUR WAVES instead of over
// workgroups: wave w streams k in [w Cin/4, (w+1) Cin/4) — its own 32 x 128-byte weight lines and 32 x 128-byte
// activation lines per block of 32 k, by LDS-DMA into a wave-private 3-slot ring behind a counted vmcnt (no barrier in
// the loop: a wave reads only what it fetched itself) — and the four 32 x 32 partial tiles are added through LDS as
// ((a0 + a1) + a2) + a3 with scale / bias / activation applied on the way out.  No slab round trip through HBM, no
// finish launch, and a summation order that depends on nothing but Cin.  (Split over workgroups, p2 and p3 took
// 6.4 + 4.5 and 10 + 4.6 us: launch-latency bound.)
constexpr int WGK_W = 4;                       // waves along K
constexpr int WGK_NS = 3;                      // ring slots per wave
constexpr int WGK_SLOT = 64 * 128;             // 32 weight rows + 32 activation rows of 128 B

__global__ __launch_bounds__(64 * WGK_W) void linear_wgk_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                                const float* __restrict__ scale, const float* __restrict__ bias,
                                                                float* __restrict__ y, int B, int Cin, int Cout, int act) {
    extern __shared__ __attribute__((aligned(16))) char wgk_smem[];       // [WGK_W][WGK_NS][64 rows][128 B]
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    const int o0 = blockIdx.x * 32, b0 = blockIdx.y * 32;
    const int kq = Cin / WGK_W, kbeg = wave * kq, nblk = kq >> 5;
    char* ring = wgk_smem + wave * (WGK_NS * WGK_SLOT);

    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(w), 0, (int)0x7fffffff, 0x00020000);
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(x), 0, (int)0x7fffffff, 0x00020000);
    // a piece = 8 rows x 128 B; lane l fills slot l & 7 of row l >> 3 with source chunk (l & 7) ^ ((row >> 1) & 7)
    int wvoff[4], xvoff[4];
#pragma unroll
    for (int q = 0; q < 4; ++q) {
        const int r = q * 8 + (lane >> 3);
        const int c = (lane & 7) ^ ((r >> 1) & 7);
        wvoff[q] = (int)(((long long)min(o0 + r, Cout - 1) * Cin + kbeg) * 4 + c * 16);
        xvoff[q] = (int)(((long long)min(b0 + r, B - 1) * Cin + kbeg) * 4 + c * 16);
    }
    auto issue = [&](int blk) {
        char* st = ring + (blk % WGK_NS) * WGK_SLOT;
        const int so = blk * 128;
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(wrsrc, S3R_LDS_PTR_PW(st + q * 1024), 16, wvoff[q], so, 0, 2);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(xrsrc, S3R_LDS_PTR_PW(st + 4096 + q * 1024), 16, xvoff[q], so, 0, 0);
    };
    f32x16_l acc;
#pragma unroll
    for (int r = 0; r < 16; ++r) acc[r] = 0.f;
    const int w_off = j * 128 + ((h ^ ((j >> 1) & 7)) << 4);
    const int x_off = 4096 + w_off;
    for (int s0 = 0; s0 < WGK_NS - 1; ++s0)
        if (s0 < nblk) issue(s0);
    for (int blk = 0; blk < nblk; ++blk) {
        // slot (blk + 2) % 3 was read in iteration blk - 1 (its ds_reads retired before those MFMAs issued): refill it,
        // then wait until only the DMAs of the blocks after `blk` are outstanding (8 per block)
        if (blk + WGK_NS - 1 < nblk) issue(blk + WGK_NS - 1);
        const int later = min(nblk - 1 - blk, WGK_NS - 1);
        if (later >= 2) asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        else if (later == 1) asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        const char* st = ring + (blk % WGK_NS) * WGK_SLOT;
        v4f xv[4], wv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            xv[t] = *reinterpret_cast<const v4f*>(st + (x_off ^ (t << 5)));
            wv[t] = *reinterpret_cast<const v4f*>(st + (w_off ^ (t << 5)));
        }
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc = __builtin_amdgcn_mfma_f32_32x32x2f32(xv[t][e], wv[t][e], acc, 0, 0, 0);
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");       // (the reads above are done before this slot is refilled)
    }
    // ---- the four K quarters: ((a0 + a1) + a2) + a3 through LDS (each wave's ring is idle now; its first 4 KiB hold its tile)
    float* mine = reinterpret_cast<float*>(ring);
#pragma unroll
    for (int r = 0; r < 16; ++r) mine[r * 64 + lane] = acc[r];
    __syncthreads();
    if (wave == 0) {
        const int o = o0 + j;
        const float sc = (scale && o < Cout) ? scale[o] : 1.f, bi = (bias && o < Cout) ? bias[o] : 0.f;
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            float sum = acc[r];
#pragma unroll
            for (int q = 1; q < WGK_W; ++q)
                sum += reinterpret_cast<const float*>(wgk_smem + q * (WGK_NS * WGK_SLOT))[r * 64 + lane];
            const int b = b0 + (r & 3) + 8 * (r >> 2) + 4 * h;
            if (b < B && o < Cout) y[(size_t)b * Cout + o] = apply_act(fmaf(sum, sc, bi), act);
        }
    }
}

// which layers the one-launch form serves: K small enough that a wave's quarter is a handful of blocks, tensors within
// 32-bit byte offsets
static bool linear_wgk_ok(int B, int Cin, int Cout) {
    // (>= 128 workgroups: p3's 192 take 14.5 us against 17.7 split over workgroups + finish; p2's 32 would take 12.6
    //  against 11.5 — too few waves in flight for a latency-bound stream of 4 MB: tools/point_bench.py, cold caches)
    return Cin % (32 * WGK_W) == 0 && Cin <= 4096 && (long long)Cin * Cout * 4 < (1ll << 31) &&
           (long long)B * Cin * 4 < (1ll << 31) && (long)((Cout + 31) / 32) * ((B + 31) / 32) >= 128;
}

// split of the K axis: about two workgroups per CU (512) so that every CU streams, K slices of whole 32-k blocks, >= 128
static void linear_split(int B, int Cin, int Cout, int* ksplit, int* kper) {
    if (Cin % 32 != 0 || (long long)Cin * Cout * 4 >= (1ll << 31) || (long long)B * Cin * 4 >= (1ll << 31)) {
        *ksplit = 1; *kper = Cin; return;            // (the streaming kernel addresses its tensors with 32-bit byte offsets)
    }
    const int otiles = (Cout + 127) / 128, btiles = (B + 31) / 32;
    int ks = 1;
    while (otiles * btiles * ks < 512 && Cin / (2 * ks) >= 128 && (Cin / (2 * ks)) % 32 == 0) ks *= 2;
    *ksplit = ks;
    *kper = Cin / ks;
}

static bool linear_streams(int B, int Cin, int Cout) {
    return Cin % 32 == 0 && (long long)Cin * Cout * 4 < (1ll << 31) && (long long)B * Cin * 4 < (1ll << 31);
}

int64_t linear_scratch_elems(int B, int Cin, int Cout) {
    if (linear_wgk_ok(B, Cin, Cout)) return 1;            // (no slabs; a non-empty scratch keeps the ABI's contract simple)
    int ks, kper;
    linear_split(B, Cin, Cout, &ks, &kper);
    return (int64_t)ks * B * Cout;
}

hipError_t launch_linear(const float* x, const float* w, const float* scale, const float* bias, float* y, int B,
                         int Cin, int Cout, int act, float* scratch, hipStream_t s) {
    if (linear_wgk_ok(B, Cin, Cout)) {
        const size_t lds = (size_t)WGK_W * WGK_NS * WGK_SLOT;
        static LdsAttr attr;
        hipError_t e = attr.ensure(reinterpret_cast<const void*>(&linear_wgk_kernel), (int)lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL(linear_wgk_kernel, dim3((Cout + 31) / 32, (B + 31) / 32), dim3(64 * WGK_W), lds, s, x, w, scale, bias,
                           y, B, Cin, Cout, act);
        return hipGetLastError();
    }
    int ks, kper;
    linear_split(B, Cin, Cout, &ks, &kper);
    const long long total = (long long)B * Cout;
    if (!linear_streams(B, Cin, Cout)) {
        hipLaunchKernelGGL(linear_naive_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, x, w, scratch, B,
                           Cin, Cout);
    } else {
        // weights are read once per forward, with the non-temporal hint (MI355X_MICROARCH.md nt-weights; measured faster than the
        // default policy inside the forward, r02)
        const size_t lds = (size_t)LIN_NS * LIN_STAGE;
        static LdsAttr attr_nt;
        hipError_t e = attr_nt.ensure(reinterpret_cast<const void*>(&linear_stream_kernel<true>), (int)lds);
        if (e != hipSuccess) return e;
        const dim3 grid((Cout + 127) / 128, ks, (B + 31) / 32);
        hipLaunchKernelGGL(linear_stream_kernel<true>, grid, dim3(256), lds, s, x, w, scratch, B, Cin, Cout, kper, ks);
    }
    if (ks >= 16 && ks % 4 == 0)
        hipLaunchKernelGGL(linear_finish4_kernel, dim3((unsigned)((total + 63) / 64)), dim3(256), 0, s, scratch, y, scale, bias,
                           Cout, total, ks, act);
    else
        hipLaunchKernelGGL(linear_finish_kernel, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, s, scratch, y, scale,
                           bias, Cout, total, ks, act);
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

// ------------------------------------------------------------------------------------------------
// Disparity read-out: winner-take-all over the SAME shift-and-diff costs the cost volume
// holds, without materialising the volume:
//   disp_l[b,h,w] = first argmin_{d in [0, min(D-1, w)]}     sum_c |L[b,c,h,w] - R[b,c,h,w-d]|
//   disp_r[b,h,w] = first argmin_{d in [0, min(D-1, W-1-w)]} sum_c |R[b,c,h,w] - L[b,c,h,w+d]|
// The channel sum runs c = 0..C-1 sequentially in fp32 (no FMA: |a-b| then add), which is what the oracle
// does, so ties and near-ties resolve identically: the result is bit-exact.  One workgroup per (b, h) row, both
// feature rows staged in LDS ([C][W] each); HBM-bound on 2*C*H*W*4 bytes per sample (25 KB at 28x28x32).
__global__ __launch_bounds__(256) void disparity_wta_kernel(const float* __restrict__ fl, const float* __restrict__ fr,
                                                            float* __restrict__ dl, float* __restrict__ dr, int C, int D,
                                                            int H, int W) {
    extern __shared__ __attribute__((aligned(16))) float dw_smem[];
    float* sl = dw_smem;
    float* sr = dw_smem + C * W;
    const int b = blockIdx.x / H, hh = blockIdx.x - b * H;
    const size_t plane = (size_t)H * W;
    const float* __restrict__ pl = fl + (size_t)b * C * plane + (size_t)hh * W;
    const float* __restrict__ pr = fr + (size_t)b * C * plane + (size_t)hh * W;
    for (int i = threadIdx.x; i < C * W; i += 256) {
        const int c = i / W, w = i - c * W;
        sl[i] = pl[(size_t)c * plane + w];
        sr[i] = pr[(size_t)c * plane + w];
    }
    __syncthreads();
    for (int o = threadIdx.x; o < 2 * W; o += 256) {
        const bool right = o >= W;
        const int w = right ? o - W : o;
        const float* a = right ? sr : sl;          // reference view
        const float* m = right ? sl : sr;          // matched view, shifted by -d (left-referenced) / +d
        const int dmax = right ? (W - 1 - w) : w;
        const int nd = (dmax < D - 1 ? dmax : D - 1) + 1;
        const int step = right ? 1 : -1;
        float best = __builtin_inff();
        int arg = 0;
        for (int d = 0; d < nd; ++d) {
            float cost = 0.f;
            const int wm = w + step * d;
            for (int c = 0; c < C; ++c) cost = cost + fabsf(a[c * W + w] - m[c * W + wm]);
            if (cost < best) { best = cost; arg = d; }
        }
        (right ? dr : dl)[(size_t)b * plane + (size_t)hh * W + w] = (float)arg;
    }
}

hipError_t launch_disparity_wta(const float* fl, const float* fr, float* dl, float* dr, int B, int C, int D, int H, int W,
                                hipStream_t s) {
    const size_t lds = (size_t)2 * C * W * sizeof(float);
    if (lds > 64 * 1024) return hipErrorInvalidValue;
    hipLaunchKernelGGL(disparity_wta_kernel, dim3(B * H), dim3(256), lds, s, fl, fr, dl, dr, C, D, H, W);
    return hipGetLastError();
}

// End-point error per sample: mean |pred - gt| over the pixels whose ground truth is valid (finite and >= 0: EXR
// disparity maps mark background with inf / negative values), plus the valid count so a caller can pool samples
// exactly.  fp64 accumulation in a fixed order (thread-strided partials, wave butterfly, 4 wave partials).
__global__ __launch_bounds__(256) void disparity_epe_kernel(const float* __restrict__ pred, const float* __restrict__ gt,
                                                            float* __restrict__ epe, int* __restrict__ count, long long S) {
    __shared__ double psum[4];
    __shared__ unsigned pcnt[4];
    const float* __restrict__ p = pred + (size_t)blockIdx.x * S;
    const float* __restrict__ g = gt + (size_t)blockIdx.x * S;
    double sum = 0.0;
    unsigned n = 0;
    for (long long i = threadIdx.x; i < S; i += 256) {
        const float t = g[i];
        const bool ok = (t >= 0.f) && (t < __builtin_inff());       // false for NaN as well
        if (ok) { sum += (double)fabsf(p[i] - t); ++n; }
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) sum += __shfl_xor(sum, o, 64);
    n = wave_sum(n);
    const int wave = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { psum[wave] = sum; pcnt[wave] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const double s4 = ((psum[0] + psum[1]) + psum[2]) + psum[3];
        const unsigned n4 = pcnt[0] + pcnt[1] + pcnt[2] + pcnt[3];
        epe[blockIdx.x] = n4 ? (float)(s4 / (double)n4) : 0.f;
        count[blockIdx.x] = (int)n4;
    }
}

hipError_t launch_disparity_epe(const float* pred, const float* gt, float* epe, int* count, int B, int64_t S,
                                hipStream_t s) {
    hipLaunchKernelGGL(disparity_epe_kernel, dim3(B), dim3(256), 0, s, pred, gt, epe, count, (long long)S);
    return hipGetLastError();
}

}  // namespace s3r
