// fp32 stride-1 convolutions and ConvTranspose3d(k4 s2 p1) with FEWER multiplications: the Winograd forms.  One class GEMM
// (wino_body) serves all of them; what differs is how many classes a layer has, what is left of its kernel inside a class, and
// who applies the output transform.  Map of this file:
//   1. one axis, F(4,3) along H (six classes, half the multiplications): input transform, weight transform         (e2, e4)
//   2. transposed convolutions, F(2,2) along D and H inside each output-parity class (nine classes, 9/16)       (d1, d2, d3)
//   3. wino_body and its launch forms: serial / class-parallel / dual (one axis), semi-fused (two axes); wino_plan
//   4. two axes, class-parallel or semi-fused: F(4,3) x F(4,3) over D, H (Conv3d k3: v1, v3, v5) or over H, W (Conv2d k3: e6, e7),
//      F(2,4) x F(2,4) (Conv3d k4 valid: v6); their input transforms, weight transforms and finish kernels
//
// One axis, F(4, 3) (Lavin & Gray): four output rows (4q .. 4q + 3) of a column need the padded input rows r0..r5 = 4q .. 4q + 5
// and the three kernel rows g0, g1, g2:
//     v0 = 4 r0 - 5 r2 + r4             u0 = g0 / 4                          y0 = m0 + m1 + m2 + m3 + m4
//     v1 = -4 r1 - 4 r2 + r3 + r4       u1 = -(g0 + g1 + g2) / 6             y1 = m1 - m2 + 2 m3 - 2 m4
//     v2 = 4 r1 - 4 r2 - r3 + r4        u2 = -(g0 - g1 + g2) / 6             y2 = m1 + m2 + 4 m3 + 4 m4
//     v3 = -2 r1 - r2 + 2 r3 + r4       u3 = g0 / 24 + g1 / 12 + g2 / 6      y3 = m1 - m2 + 8 m3 - 8 m4 + m5
//     v4 = 2 r1 - r2 - 2 r3 + r4        u4 = g0 / 24 - g1 / 12 + g2 / 6
//     v5 = 4 r1 - 5 r3 + r5             u5 = g2                              m_i = sum over (cin, kd, kw) of u_i * v_i
// so the layer becomes SIX convolutions with a (kd x 1 x kw) kernel — 9 taps instead of 27 in 3D, 3 instead of 9 in 2D, each over
// its own transformed input plane set V_i and its own transformed weights U_i — whose results are combined in registers: HALF
// the matrix work of the direct form for the same outputs (edges that are not a multiple of 4 compute a partial last group:
// 16 rows for 14, 8 for 7).  Along H alone K stays Cin x 9 (or x 3), the W axis stays contiguous (16-byte gathers where
// W % 4 == 0, whole-row stores), six classes' accumulators fit a workgroup's registers, and the kernel is the implicit GEMM of
// s3r_conv_glds.hip with a class loop around its K loop.  Every further axis multiplies the classes (36 for two) and shortens
// each class's K: a workgroup cannot hold them, so the two-axis forms (section 4) send class sums through slabs in HBM — which
// pays where the output is small (an edge <= 28) and not for e2 / e4.  In fp32 over 64-256 channels the forms are as accurate as
// the direct sum (rel-L2 against an fp64 convolution: direct 3e-7 .. 1e-6, one axis <= 2.3e-6, two axes <= 4.7e-6; north_star
// allows 1e-4) but they are DIFFERENT summations: results are not bit-identical to the direct kernels' nor to each other's,
// which is why the choice between them is part of the layer descriptor (s3r_algo, include/s3r.h).
//
//   wino_input_kernel   x (padded NC(D)HW, halo 1) -> V[6][B][C][Dp][ceil(H/4)][Wp]   (HBM-bound: reads x once, writes 1.5 x)
//   pack_wino_kernel    w[Cout][Cin][kd][3][kw]    -> Up[6][(chunk*T' + tap')*32 + c][CoutPad],  T' = kd*kw, 32-channel chunks
//   wino_kernel / wino_dual_kernel / wino_finish_kernel: the class kernels, below
#include "s3r_kernels.h"
#include <cstdlib>

namespace s3r {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wv2f __attribute__((ext_vector_type(2)));
template <int N> struct WVec;
template <> struct WVec<1> { typedef float type; };
template <> struct WVec<2> { typedef wv2f type; };
typedef float wv4f __attribute__((ext_vector_type(4)));
template <> struct WVec<4> { typedef wv4f type; };
template <int N>
__device__ __forceinline__ float wvget(const typename WVec<N>::type& v, int i) {
    if constexpr (N == 1) return v; else return v[i];
}

#define S3R_LDS_PTR_W(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef S3R_WNB
#define S3R_WNB 2      // LDS stages.  With 16-channel K tiles: 2 stages -0.3 %, 3 = 4.  32-channel tiles halve the barriers per MFMA
#endif                  // (32 MFMAs per wave between two, like the direct path's 64 x 256 tile) and, at two stages (48 KiB, three
                        // workgroups per CU), are 1.3 % of the step faster than 16 x 3: v1 0.807 -> 0.769 ms, d3 0.824 -> 0.803
#ifndef S3R_WBK
#define S3R_WBK 32
#endif
constexpr int WBM = 64, WBK = S3R_WBK, WNB = S3R_WNB;      // K tile: one tap x WBK channels
constexpr int WNPA = WBK * WBM / 1024;                                // 1 KiB weight pieces per wave and K tile
int wino_bk() { return WBK; }

#ifdef S3R_ABLATE
// S3R_ABL=7: per-workgroup timeline stamps (s_memrealtime, 100 MHz) of the class kernels, tools/timeline.py --wino:
// [cu key, start, loop start, loop end, end, stores issued]
__device__ unsigned long long s3r_wino_timeline[6 * 65536];
#endif

template <int BYTES>
__device__ __forceinline__ void wdma(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voffset, int soffset) {
    static_assert(BYTES == 16 || BYTES == 4, "LDS-DMA width");
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR_W(lds_dst), 16, voffset, soffset, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR_W(lds_dst), 4, voffset, soffset, 0, 0);
}

// ---- input transform: one thread per (plane row of V, column); rows 0 .. R+1 of the group in the padded input plane
template <int R>
__global__ __launch_bounds__(256) void wino_input_kernel(const float* __restrict__ x, float* __restrict__ V, long long planes,
                                                         int Hp, int Wp, int Hq, long long cls_stride) {
    const long long total = planes * Hq * Wp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / Wp;                     // (plane, q)
        const int w = (int)(i - row * Wp);
        const long long pl = row / Hq;
        const int q = (int)(row - pl * Hq);
        const float* __restrict__ src = x + (pl * Hp + R * q) * Wp + w;
        float r[R + 2], v[R + 2];
#pragma unroll
        for (int k = 0; k < R + 2; ++k) r[k] = (R * q + k < Hp) ? src[(long long)k * Wp] : 0.f;   // (rows below the halo are zero)
        wino_rows_to_classes<R>(r, v);
#pragma unroll
        for (int k = 0; k < R + 2; ++k) V[i + k * cls_stride] = v[k];
    }
}

hipError_t launch_wino_input(const float* x, float* V, long long planes, int Hp, int Wp, int Hq, int R, hipStream_t s) {
    const long long total = planes * Hq * Wp;
    const long long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 65536 ? blocks : 65536));
    if (R != 4) return hipErrorInvalidValue;
    hipLaunchKernelGGL(wino_input_kernel<4>, grid, dim3(256), 0, s, x, V, planes, Hp, Wp, Hq, total);
    return hipGetLastError();
}

// ---- weights: w[Cout][Cin][kd][3][kw] (torch layout, 2D: kd = 1) -> R + 2 class slabs in the conv kernel's packed K order
__global__ void pack_wino_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad, int kd,
                                 int kw, int R) {
    const int T = kd * kw;
    const size_t per_cls = (size_t)T * Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)(R + 2) * per_cls; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_cls);
        size_t r = i % per_cls;
        const int co = (int)(r % CoutPad);
        r /= CoutPad;
        const int c = (int)(r % WBK);
        r /= WBK;
        const int tap = (int)(r % T);
        const int cc = (int)(r / T);
        const int cin = cc * WBK + c;
        float v = 0.f;
        if (co < Cout) {
            const int td = tap / kw, tw = tap - td * kw;
            const float* g = w + (((size_t)co * Cin + cin) * kd + td) * 3 * kw + tw;       // g[kh * kw]
            const float g0 = g[0], g1 = g[kw], g2 = g[2 * kw];
            const float s02 = g0 + g2, a = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), b12 = g1 * (1.f / 12.f);
            v = cls == 0 ? g0 * 0.25f : cls == 1 ? (s02 + g1) * (-1.f / 6.f) : cls == 2 ? (s02 - g1) * (-1.f / 6.f)
              : cls == 3 ? a + b12 : cls == 4 ? a - b12 : g2;
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_wino(const float* w, float* wp, int Cin, int Cout, int CoutPad, int kd, int kw, int R, hipStream_t s) {
    hipLaunchKernelGGL(pack_wino_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad, kd, kw, R);
    return hipGetLastError();
}

// ================================================================================================
// ConvTranspose3d(k = 4, s = 2, p = 1) with fewer multiplications: Winograd F(2, 2) along D AND H inside every output-parity
// class.  A class (rd, rh, rw) is a 2 x 2 x 2-tap convolution over the input grid (s3r_conv_glds.hip).  Along one axis, two of its
// outputs that are neighbours — input positions 2q and 2q + 1 — read the padded input rows x0, x1, x2 = R, R + 1, R + 2 (R = 2q + r)
// through the axis' two taps g0, g1:
//     y(2q)     = x0 g0 + x1 g1 = m0 + m1        m0 = (x0 - x1) g0
//     y(2q + 1) = x1 g0 + x2 g1 = m1 - m2        m1 = x1 (g0 + g1)        m2 = (x1 - x2) g1
// — three products instead of four.  r03 did this along H (3/4 of the matrix work); nested along D and H it is NINE products
// M_ab (a, b in 0..2: the D and the H class) for 2 x 2 outputs instead of sixteen: 9/16 of the direct form's multiplications,
// each product a 2-tap (column) convolution over Cin.  Class (a, b) reads {a = 1: plain, else depth differences} x {b = 1: plain,
// else row differences} of the padded input at depth Z + (a ? 1 : 0), row R + (b ? 1 : 0):
//     Dh[z][r] = x[z][r] - x[z][r + 1],   Dd[z][r] = x[z][r] - x[z + 1][r],   Ddh[z][r] = Dh[z][r] - Dh[z + 1][r].
// The row differences Dh are always a tensor of their own (wino_diff_kernel: the input of a transposed convolution is small).  The
// DEPTH differences come in two forms with the same bits: MATERIALISED (p.xd_mode = 1: the same kernel writes Dd and Ddh too, the
// class kernel reads one tile per K step) where the four tensors stay in the Infinity Cache (d1, d2 at B = 32), or formed IN the
// kernel (p.xd_mode = 0: a depth-difference class stages the tiles of depths z and z + 1 side by side in LDS and subtracts the two
// fragments in front of the MFMA, one v_sub per matrix instruction) where writing and re-reading two more tensors would cross HBM
// (d3 at B = 32: 0.72 -> 0.69 ms; the 1.67 x operand traffic costs d1 / d2, which are L2-delivery bound, 27 / 4 %).  The transformed weights are the sums of the 2 x 2 (depth tap, row tap) weights over {a: td = 0 | both | 1} x
// {b: th = 0 | both | 1}, per (parity class, column tap).  Positions are (b, s, q, pw) = (sample, depth pair, row pair, column) over
// the input grid; the output transform Y[u][v] = sum A[u][a] A[v][b] M_ab, A = [[1, 1, 0], [0, 1, -1]], runs on the accumulator
// registers; HEAD = the fused 1 x 1 x 1 head (d3 -> d4).
// D = [Dh | Dd | Ddh] (THREE) or [Dh].  One thread per element; the index arithmetic is 32-bit mulhi division by launch
// invariants (the first version's 64-bit % and / per element ran at 2.7 TB/s).
template <bool THREE>
__global__ __launch_bounds__(256) void wino_diff_kernel(const float* __restrict__ x, float* __restrict__ D, unsigned total,
                                                        int Dp, int Hp, int Wp, FastDiv dPer, FastDiv dHW, FastDiv dW) {
    const unsigned hw = (unsigned)Hp * Wp, per = (unsigned)Dp * hw;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned pl = (unsigned)dPer.div((int)i);
        const unsigned e = i - pl * per;
        const unsigned z = (unsigned)dHW.div((int)e);
        const unsigned r = (unsigned)dW.div((int)(e - z * hw));
        const bool rn = r + 1 < (unsigned)Hp, zn = z + 1 < (unsigned)Dp;
        const float x00 = x[i];
        const float x01 = rn ? x[i + Wp] : 0.f;
        const float dh0 = rn ? x00 - x01 : 0.f;
        D[i] = dh0;
        if constexpr (THREE) {
            const float x10 = zn ? x[i + hw] : 0.f;
            const float x11 = (rn && zn) ? x[i + hw + Wp] : 0.f;
            const float dh1 = rn ? x10 - x11 : 0.f;          // Dh at depth z + 1
            D[i + (size_t)total] = zn ? x00 - x10 : 0.f;
            D[i + 2 * (size_t)total] = zn ? dh0 - dh1 : 0.f;
        }
    }
}

// the same two columns per thread (rows are an even number of floats — edge + 2 — and 8-byte aligned: half the memory instructions)
template <bool THREE>
__global__ __launch_bounds__(256) void wino_diff2_kernel(const float* __restrict__ x, float* __restrict__ D, unsigned total,
                                                         int Dp, int Hp, int Wp, FastDiv dPer, FastDiv dHW, FastDiv dW) {
    const unsigned hw = (unsigned)Hp * Wp, per = (unsigned)Dp * hw;
    for (unsigned i = (blockIdx.x * 256u + threadIdx.x) * 2u; i < total; i += gridDim.x * 512u) {
        const unsigned pl = (unsigned)dPer.div((int)i);
        const unsigned e = i - pl * per;
        const unsigned z = (unsigned)dHW.div((int)e);
        const unsigned r = (unsigned)dW.div((int)(e - z * hw));
        const bool rn = r + 1 < (unsigned)Hp, zn = z + 1 < (unsigned)Dp;
        const wv2f zero = {0.f, 0.f};
        const wv2f x00 = *reinterpret_cast<const wv2f*>(x + i);
        const wv2f x01 = rn ? *reinterpret_cast<const wv2f*>(x + i + Wp) : zero;
        const wv2f dh0 = rn ? x00 - x01 : zero;
        *reinterpret_cast<wv2f*>(D + i) = dh0;
        if constexpr (THREE) {
            const wv2f x10 = zn ? *reinterpret_cast<const wv2f*>(x + i + hw) : zero;
            const wv2f x11 = (rn && zn) ? *reinterpret_cast<const wv2f*>(x + i + hw + Wp) : zero;
            const wv2f dh1 = rn ? x10 - x11 : zero;          // Dh at depth z + 1
            *reinterpret_cast<wv2f*>(D + i + (size_t)total) = zn ? x00 - x10 : zero;
            *reinterpret_cast<wv2f*>(D + i + 2 * (size_t)total) = zn ? dh0 - dh1 : zero;
        }
    }
}

hipError_t launch_wino_diff(const float* x, float* D, long long planes, int Dp, int Hp, int Wp, int three, hipStream_t s) {
    const long long total = planes * Dp * Hp * Wp;
    if (total >= (1ll << 31)) return hipErrorInvalidValue;
    if ((Wp & 1) == 0 && ((reinterpret_cast<size_t>(x) | reinterpret_cast<size_t>(D)) & 7) == 0 && (total & 1) == 0) {
        const long long blocks2 = (total / 2 + 255) / 256;
        const dim3 grid2((unsigned)(blocks2 < 16384 ? blocks2 : 16384));
        const FastDiv dPer((unsigned)(Dp * Hp * Wp)), dHW((unsigned)(Hp * Wp)), dW((unsigned)Wp);
        if (three) hipLaunchKernelGGL(wino_diff2_kernel<true>, grid2, dim3(256), 0, s, x, D, (unsigned)total, Dp, Hp, Wp, dPer, dHW, dW);
        else hipLaunchKernelGGL(wino_diff2_kernel<false>, grid2, dim3(256), 0, s, x, D, (unsigned)total, Dp, Hp, Wp, dPer, dHW, dW);
        return hipGetLastError();
    }
    const long long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
    const FastDiv dPer((unsigned)(Dp * Hp * Wp)), dHW((unsigned)(Hp * Wp)), dW((unsigned)Wp);
    if (three) hipLaunchKernelGGL(wino_diff_kernel<true>, grid, dim3(256), 0, s, x, D, (unsigned)total, Dp, Hp, Wp, dPer, dHW, dW);
    else hipLaunchKernelGGL(wino_diff_kernel<false>, grid, dim3(256), 0, s, x, D, (unsigned)total, Dp, Hp, Wp, dPer, dHW, dW);
    return hipGetLastError();
}

// w[Cin][Cout][4][4][4] -> Up[pc = 8][cls = 9][(chunk*2 + tw)*32 + c][CoutPad]
__global__ void pack_wino_deconv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad) {
    const size_t per_f = (size_t)2 * Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 72 * per_f; i += (size_t)gridDim.x * blockDim.x) {
        const int pf = (int)(i / per_f);
        const int pc = pf / 9, f = pf - pc * 9;
        const int fa = f / 3, fb = f - fa * 3;
        size_t r = i % per_f;
        const int co = (int)(r % CoutPad);
        r /= CoutPad;
        const int c = (int)(r % WBK);
        r /= WBK;
        const int tw = (int)(r & 1);
        const int cc = (int)(r >> 1);
        const int cin = cc * WBK + c;
        float v = 0.f;
        if (co < Cout) {
            const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
            const int kw = 3 - rw - 2 * tw;
            const float* g = w + ((size_t)cin * Cout + co) * 64 + kw;
            // taps t = 0, 1 along an axis of parity r: kernel index 3 - r, 1 - r
            auto hsum = [&](int td) {
                const float* gd = g + (3 - rd - 2 * td) * 16;
                const float g0 = gd[(3 - rh) * 4], g1 = gd[(1 - rh) * 4];
                return fb == 0 ? g0 : fb == 2 ? g1 : g0 + g1;
            };
            v = fa == 0 ? hsum(0) : fa == 2 ? hsum(1) : hsum(0) + hsum(1);
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_wino_deconv(const float* w, float* wp, int Cin, int Cout, int CoutPad, hipStream_t s) {
    hipLaunchKernelGGL(pack_wino_deconv_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    return hipGetLastError();
}

// ================================================================================================
// The class kernels.  ONE body serves the Winograd forms of this file,
//     KIND 1: convolution, F(4,3) along H (6 classes, 4 output rows per group)
//     KIND 2: ConvTranspose3d(k4 s2 p1), F(2,2) along D and H inside every output-parity class (9 classes, 2 x 2 outputs)
// in two launch forms:
//     serial (CP = false): a workgroup owns a tile of 64 couts x BN positions (position = one group of R output rows of one
//         column) and walks ALL classes back to back, one accumulator set per class; the output transform, the folded BN +
//         ReLU (and d3's fused 1 x 1 x 1 head) run on the registers.  4 waves as WM x WN = 2 x 2 (32 couts x 32 positions per
//         wave, BN = 64: one MFMA tile per class — 109 registers for six classes, four workgroups per CU; r03's 64 x 128 tiles
//         held two and measured 3-13 % slower on every layer).  The fused head adds the two waves' 32-cout sums through LDS;
//     class-parallel (CP = true): a workgroup owns ONE class of a tile, runs that class's K loop — the same MFMA sequence in the
//         same order as the serial form, so the same bits — and writes the raw class sums to a slab part[class][cout][n];
//         wino_finish_kernel then applies the output transform and the epilogue through the SAME device functions
//         (wino_out / wino_act below) in the same operation order: bit-identical to the serial form.  It multiplies the
//         workgroup count by the class count (x 6 / x 9) and divides a workgroup's serial K walk by it: the form for
//         grids that leave the chip short of workgroups (small batches; v5; d1) and for the REMAINDER of a launch whose
//         workgroup count is a little over a multiple of the chip's slots (wino_dual_kernel: bulk serial + remainder
//         class-parallel in one launch — the direct path's plan_tail_cut idea with the class axis as the finer unit).
// Because the two forms agree bit for bit, which one runs may depend on the batch size (a sample's bits never do).
template <int KIND> struct WinoKind;
template <> struct WinoKind<1> { static constexpr int NCLS = 6, R = 4; };
template <> struct WinoKind<2> { static constexpr int NCLS = 9, R = 4; };     // R: outputs per position (2 depths x 2 rows)

// output transform of one (cout, position): m = the class sums, y = the R output rows.  Adds and explicit fmaf only (nothing
// the compiler could contract differently in the two kernels that share it).
template <int KIND>
__device__ __forceinline__ void wino_out(const float (&m)[WinoKind<KIND>::NCLS], float (&y)[WinoKind<KIND>::R]) {
    if constexpr (KIND == 1) {
        const float s12 = m[1] + m[2], d12 = m[1] - m[2];
        const float s34 = m[3] + m[4], d34 = m[3] - m[4];
        y[0] = (m[0] + s12) + s34;
        y[1] = fmaf(2.f, d34, d12);
        y[2] = fmaf(4.f, s34, s12);
        y[3] = fmaf(8.f, d34, d12) + m[5];
    } else {             // m[3 a + b]; rows first, then depths: y[2 u + v]
        const float t00 = m[0] + m[1], t01 = m[1] - m[2];
        const float t10 = m[3] + m[4], t11 = m[4] - m[5];
        const float t20 = m[6] + m[7], t21 = m[7] - m[8];
        y[0] = t00 + t10;
        y[1] = t01 + t11;
        y[2] = t10 - t20;
        y[3] = t11 - t21;
    }
}
// element offset of output i of a position from its first output (conv: rows 4q + i; transposed: depth pair u = i >> 1, row pair
// v = i & 1, two output planes / rows apart)
template <int KIND>
__device__ __forceinline__ int wino_out_off(int i, int y_ds, int y_hs) {
    if constexpr (KIND == 1) return i * y_hs;
    else return (i >> 1) * 2 * y_ds + (i & 1) * 2 * y_hs;
}
__device__ __forceinline__ float wino_act(float y, float sc, float sf, float lo) { return fmaxf(fmaf(y, sc, sf), lo); }
__device__ __forceinline__ float wino_head_act(float t, int act) {
    if (act == ACT_RELU) return fmaxf(t, 0.f);
    if (act == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-t));
    return t;
}

// p (convolution): the CLASS convolution — p.x = V, x_cs / x_ds / x_hs its strides (x_hs = one V row per group), p.x_cls the class
//   stride, Nh = groups per plane, T = kd * kw, x_org = 0; p.y the layer's padded output, p.Hout its true height (the last group's
//   missing rows are not stored).
// p (transposed): the layer's own parameters (make_params) with Nd = depth pairs, Nh = row pairs (n / 2), p.x = the padded input,
//   p.xd = its row differences Dh (x's shape and strides; p.xd_mode = 1: followed by Dd and Ddh), p.w = the 72 (parity class, class) slabs.
// p.part: class-parallel slabs [class][Cout][npad], npad = the position range's tile count x BN.
// XM (transposed form): the depth differences are materialised (p.xd = [Dh | Dd | Ddh]) instead of formed here
// SEMI (two-axis convolution, below): the workgroup owns ONE depth class a of its tile and walks that class's six row classes
// serially (plane sets / weight slabs 6 a .. 6 a + 5); the row transform runs on the registers and the four row outputs go,
// raw, to the slab part[a][row][cout][n] for wino2s_finish_kernel to transform along D
template <int VEC, int KIND, int WN, bool CP, bool HEAD, bool XM = false, bool SEMI = false>
__device__ __forceinline__ void wino_body(const ConvParams& p, const int bid_in, const int nwg_in, const int n_begin,
                                          const int n_end, float* __restrict__ wsmem) {
    constexpr bool DECONV = KIND == 2;
    constexpr int NCLS = WinoKind<KIND>::NCLS, R = WinoKind<KIND>::R;
    constexpr int WM = 4 / WN, TM = 2 / WM, BN = 32 * WN;
    constexpr int NACC = CP ? 1 : NCLS;
    static_assert(KIND == 1 || KIND == 2, "F(4,3) convolution or F(2,2) transposed convolution");
    static_assert(!SEMI || (KIND == 1 && !CP && !HEAD), "the semi-fused form: serial F(4,3) row classes of one depth class");
    static_assert(WN == 4 || WN == 2, "4 waves as 1 x 4 or 2 x 2");
    static_assert(!HEAD || (DECONV && !CP && WN == 2), "the fused head: serial transposed form");
    float* As = wsmem;                                   // [WNB][WBK][64]
    constexpr int BSTG = (DECONV ? 2 : 1) * WBK * BN;    // B floats per stage (transposed: depths z and z + 1 side by side)
    float* Bs = wsmem + WNB * WBK * WBM;                 // [WNB][BSTG]
    constexpr int PB = 64 * VEC;                         // floats per B piece
    constexpr int NPIECE_B = WBK * BN / PB;
    static_assert(NPIECE_B % 4 == 0, "B pieces divide over the 4 waves");
    constexpr int NPB = NPIECE_B / 4;
    constexpr bool B_WIDE = BN >= PB;
    constexpr int PPR = B_WIDE ? BN / PB : 1;
    constexpr int RPP = B_WIDE ? 1 : PB / BN;
    constexpr int LPR_B = BN / VEC;
    constexpr int NPD = WNPA + NPB;                      // DMAs per wave per K tile (WNPA 1 KiB weight pieces + NPB gathers)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;
#ifdef S3R_ABLATE
    unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
    if (p.debug == 7) tl0 = __builtin_amdgcn_s_memrealtime();
#endif

    const int n_tiles = (n_end - n_begin + BN - 1) / BN;
    int bid = bid_in, cls0 = 0, pc = 0;                  // cls0: the one class of a class-parallel workgroup; pc: output parity class
    if constexpr (CP || SEMI) {
        // an XCD walks a contiguous run of items, class-major: neighbouring workgroups share a class's weight slab (and the
        // transformed rows their tiles have in common) in that XCD's L2
        const int nwg = nwg_in;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        const int item = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
        const int ntile = p.m_tiles * n_tiles;
        const int c = item / ntile;
        bid = item - c * ntile;
        if constexpr (DECONV) { pc = c / NCLS; cls0 = c - pc * NCLS; } else cls0 = SEMI ? c * NCLS : c;
    } else if constexpr (DECONV) {
        // an XCD walks its run of tiles with the 8 parity classes of a tile back to back (they read the same input tile)
        const int nwg = nwg_in >> 3;
        const int item = (bid & 7) * nwg + (bid >> 3);
        bid = item >> 3;
        pc = item & 7;
    } else {
        const int nwg = nwg_in;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
    const int m_tile = bid % p.m_tiles, n_tile = bid / p.m_tiles;
    const int m0 = m_tile * WBM, n0 = n_begin + n_tile * BN;
    const int S = p.Nd * p.Nh * p.Nw;
    const int T = DECONV ? 2 : p.T;                      // taps per class: (depth tap,) column tap
    const int chunks = p.Cin / WBK;
    const int nkt = T * chunks;                          // K tiles per class
    const int total = NACC * nkt;

    int bvoff;
    {
        int col, lrow;
        if (B_WIDE) { col = (wave % PPR) * PB + lane * VEC; lrow = 0; }
        else        { col = (lane % LPR_B) * VEC;           lrow = lane / LPR_B; }
        int n = n0 + col;
        if (n >= n_end) n = n_end - VEC;                 // tail tile: fetch a valid group, never stored
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        if constexpr (DECONV)        // padded indices: depth 2 pd + rd (+ 1), row 2q + rh (+ 1), column pw + rw + tw
            bvoff = (b * p.Cin * p.x_cs + (2 * pd + rd) * p.x_ds + (2 * q + rh) * p.x_hs + pw + rw + lrow * p.x_cs) * 4;
        else
            bvoff = (b * p.Cin * p.x_cs + p.x_org + pd * p.x_ds + q * p.x_hs + pw + lrow * p.x_cs) * 4;
    }
    const int avoff = ((lane >> 4) * p.CoutPad + (lane & 15) * 4) * 4;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc =                 // (transposed form: the row differences)
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(DECONV ? p.xd : p.x), 0, (int)p.x_bytes, 0x00020000);
    const size_t x_el = (size_t)p.x_bytes / 4;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0,
        (int)((unsigned)(DECONV ? 72 : (p.ncls ? p.ncls : NCLS)) * (unsigned)T * (unsigned)p.Cin * (unsigned)p.CoutPad * 4u),
        0x00020000);
    const int b_row0 = B_WIDE ? wave / PPR : wave * RPP;
    constexpr int B_ROW_STEP = B_WIDE ? 4 / PPR : 4 * RPP;
    const int b_lds0 = B_WIDE ? b_row0 * BN + (wave % PPR) * PB : wave * PB;
    constexpr int B_LDS_STEP = B_WIDE ? B_ROW_STEP * BN : 4 * PB;
    const int cs4 = p.x_cs * 4;

    // cursor of the NEXT K tile to fetch (scalar): class, chunk, tap; c_kt = its row block in the packed class slabs
    int c_cls = cls0, c_cc = 0, c_td = 0, c_tw = 0, c_tap = 0;
    int c_kt = (DECONV ? pc * NCLS + cls0 : cls0) * nkt;
    int c_a = cls0 / 3, c_b = cls0 - 3 * (cls0 / 3);     // (transposed form) the class's depth / row part
#ifdef S3R_ABLATE   // diagnostic builds only (tools/README.md): what the operand DMAs of the K loop cost the matrix pipe
    bool abl_loop = false;                               // set once the prologue's tiles are out
#endif
    auto issue = [&](int buf) __attribute__((always_inline)) {
#ifdef S3R_ABLATE
        // (6: the activation tile of a (class, chunk) fetched for its first column tap only — the traffic a tap-shared B tile would have)
        const bool skip_a = abl_loop && (p.debug == 3 || p.debug == 5);
        const bool skip_b = abl_loop && (p.debug == 3 || p.debug == 4 || (p.debug == 6 && (DECONV ? c_tap != 0 : c_tw != 0)));
#else
        constexpr bool skip_a = false, skip_b = false;
#endif
        if (!skip_a) {
#pragma unroll
            for (int q = 0; q < WNPA; ++q)
                wdma<16>(wrsrc, As + buf * WBK * WBM + (wave + 4 * q) * 256, avoff,
                         (c_kt * WBK * p.CoutPad + m0) * 4 + (wave + 4 * q) * 4 * p.CoutPad * 4);
        }
        float* sb = Bs + buf * BSTG + b_lds0;
        if constexpr (DECONV) {
            // along an axis, F-class 0: differences at index R, 1: plain at R + 1, 2: differences at R + 1
            const int b_base = ((c_cc * WBK + b_row0) * p.x_cs + (c_a ? p.x_ds : 0) + (c_b ? p.x_hs : 0) + c_tap) * 4;
            if (skip_b) {
            } else if constexpr (XM) {
                // materialised differences: one tile of x / Dh / Dd / Ddh (a descriptor per K tile: the tensors together may pass
                // 2 GiB, and a 4-way choice between ready-made descriptors makes the compiler build a lookup table in scratch)
                const int t_idx = (c_a != 1 ? 2 : 0) + (c_b != 1 ? 1 : 0);          // 0: x, 1: Dh, 2: Dd, 3: Ddh
                const float* tb = t_idx == 0 ? p.x : p.xd + (size_t)(t_idx - 1) * x_el;
                const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tb), 0, (int)p.x_bytes, 0x00020000);
#pragma unroll
                for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(trsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
            } else if (c_b == 1) {
                // row part b: plain / row differences (two descriptors); depth part a: a difference class fetches depth z + 1 as well
#pragma unroll
                for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(xrsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
                if (c_a != 1) {
#pragma unroll
                    for (int q = 0; q < NPB; ++q)
                        wdma<4 * VEC>(xrsrc, sb + WBK * BN + q * B_LDS_STEP, bvoff, b_base + p.x_ds * 4 + q * B_ROW_STEP * cs4);
                }
            } else {
#pragma unroll
                for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(drsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
                if (c_a != 1) {
#pragma unroll
                    for (int q = 0; q < NPB; ++q)
                        wdma<4 * VEC>(drsrc, sb + WBK * BN + q * B_LDS_STEP, bvoff, b_base + p.x_ds * 4 + q * B_ROW_STEP * cs4);
                }
            }
            ++c_kt;
            if (++c_tap == 2) {
                c_tap = 0;
                if (++c_cc == chunks) {
                    c_cc = 0; ++c_cls;
                    if (++c_b == 3) { c_b = 0; ++c_a; }
                }
            }
        } else {
            const int b_base = (c_cls * p.x_cls + (c_cc * WBK + b_row0) * p.x_cs + c_td * p.x_ds + c_tw) * 4;
            if (!skip_b) {
#pragma unroll
                for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(xrsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
            }
            ++c_kt;
            if (++c_tw == p.kw) { c_tw = 0; ++c_td; }
            if (++c_tap == T) {
                c_tap = 0; c_td = 0; c_tw = 0;
                if (++c_cc == chunks) { c_cc = 0; ++c_cls; }
            }
        }
    };

    wf32x16 acc[NACC][TM];
#pragma unroll
    for (int c = 0; c < NACC; ++c)
#pragma unroll
        for (int t = 0; t < TM; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

    // tiles 0 .. WNB-2 go out; tile 0 has landed once at most WNB-2 tiles' DMAs are outstanding
#pragma unroll
    for (int i = 0; i < WNB - 1; ++i)
        if (i < total) issue(i);
    if (total >= WNB - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WNB - 2) * NPD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
#ifdef S3R_ABLATE
    abl_loop = true;
    if (p.debug == 7) tl1 = __builtin_amdgcn_s_memrealtime();
#endif

    const int a_off = h * WBM + wm * TM * 32 + j * TM;
    const int b_off = h * BN + wn * 32 + j;
    typedef typename WVec<TM>::type AV;
    int cur = 0, g = 0;
    // diff (transposed form, depth part a != 1): the B fragment is the difference of the two depth tiles of the stage
    auto run_class = [&](wf32x16 (&ac)[TM], const bool diff) __attribute__((always_inline)) {
        for (int kt = 0; kt < nkt; ++kt, ++g) {
            const bool more = g + WNB - 1 < total;
            if (more) issue(cur == 0 ? WNB - 1 : cur - 1);        // into the stage tile g - 1 was read from
            const float* a = As + cur * WBK * WBM + a_off;
            const float* b = Bs + cur * BSTG + b_off;
            if (DECONV && diff) {
#pragma unroll
                for (int ks = 0; ks < WBK / 2; ++ks) {
                    const AV av = *reinterpret_cast<const AV*>(a + ks * 2 * WBM);
                    const float bv = b[ks * 2 * BN] - b[WBK * BN + ks * 2 * BN];
#pragma unroll
                    for (int t = 0; t < TM; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wvget<TM>(av, t), bv, ac[t], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int ks = 0; ks < WBK / 2; ++ks) {
                    const AV av = *reinterpret_cast<const AV*>(a + ks * 2 * WBM);
                    const float bv = b[ks * 2 * BN];
#pragma unroll
                    for (int t = 0; t < TM; ++t) ac[t] = __builtin_amdgcn_mfma_f32_32x32x2f32(wvget<TM>(av, t), bv, ac[t], 0, 0, 0);
                }
            }
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WNB - 2) * NPD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur = cur + 1 == WNB ? 0 : cur + 1;
        }
    };
    static_assert(WNB == 2 || !DECONV, "the transposed form issues a class-dependent number of DMAs per K tile: vmcnt(0) only");
    constexpr bool otf = DECONV && !XM;                  // depth differences formed here
    if constexpr (CP) run_class(acc[0], otf && cls0 / 3 != 1);
    else {
#pragma unroll
        for (int c = 0; c < NACC; ++c) run_class(acc[c], otf && c / 3 != 1);
    }

#ifdef S3R_ABLATE
    if (p.debug == 7) tl2 = __builtin_amdgcn_s_memrealtime();
    auto tl_finish = [&]() __attribute__((always_inline)) {
#if defined(__HIP_DEVICE_COMPILE__)      // (the "=s" constraint below does not exist for the host pass)
        if (p.debug == 7 && tid == 0 && blockIdx.x < 65536) {
            const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
            unsigned xcc;
            asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
            unsigned long long* t = s3r_wino_timeline + 6 * (size_t)blockIdx.x;
            t[5] = t_issued;
            t[0] = ((unsigned long long)(xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
            t[1] = tl0; t[2] = tl1; t[3] = tl2; t[4] = __builtin_amdgcn_s_memrealtime();
        }
#endif
    };
#define S3R_TL_FINISH() tl_finish()
#if defined(__HIP_DEVICE_COMPILE__)      // (the "v" constraint does not exist for the host pass)
    if (p.debug == 1) {       // timing-only: no epilogue (accumulators kept live)
#pragma unroll
        for (int c = 0; c < NACC; ++c)
#pragma unroll
            for (int t = 0; t < TM; ++t) asm volatile("" ::"v"(acc[c][t]));
        S3R_TL_FINISH();
        return;
    }
#endif
#else
#define S3R_TL_FINISH() (void)0
#endif
    // rows of this lane: cout m0 + mbase + dm, dm = ((r & 3) + 8 (r >> 2)) * TM + tm
    const int mbase = wm * TM * 32 + 4 * h * TM;
    const int mlimit = p.Cout - (m0 + mbase);
    if constexpr (CP) {
        // ---- class-parallel: the raw class sums to the slab, blocked [cout][64-position tile][class][64]: all classes of a tile's
        // positions are one contiguous run (what the finish kernel reads per position); every lane of the tile has a slot
        static_assert(BN == 64 || !CP, "class-parallel slabs are blocked in 64-position tiles");
        const int cg = DECONV ? pc * NCLS + cls0 : cls0;
        const int nblk = DECONV ? 8 * NCLS : (p.ncls ? p.ncls : NCLS);       // classes per block
        float* __restrict__ slab = p.part + (((size_t)(m0 + mbase) * n_tiles + n_tile) * nblk + cg) * 64 + wn * 32 + j;
        const size_t mstride = (size_t)n_tiles * nblk * 64;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = ((r & 3) + 8 * (r >> 2)) * TM + tm;
                if (dm >= mlimit) continue;
                slab[(size_t)dm * mstride] = acc[0][tm][r];
            }
        S3R_TL_FINISH();
        return;
    } else if constexpr (SEMI) {
        // ---- semi-fused two-axis form: the four row outputs of this depth class, raw, to part[(a * 4 + row)][cout][n - n_begin]
        // blocked by (cout, tile): the 24 partial rows of a tile's 64 positions are one 6 KiB run, so the finish kernel — which
        // needs all 24 of a position — reads a few contiguous blocks instead of 24 streams 11 MB apart
        const int a = cls0 / NCLS;
        static_assert(BN == 64 || !SEMI, "semi-fused slabs are blocked in 64-position tiles");
        float* __restrict__ slab = p.part + (((size_t)(m0 + mbase) * n_tiles + n_tile) * (6 * R) + a * R) * 64 + wn * 32 + j;
        const size_t mstride = (size_t)n_tiles * (6 * R) * 64;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = ((r & 3) + 8 * (r >> 2)) * TM + tm;
                if (dm >= mlimit) continue;
                float m[NCLS], y[R];
#pragma unroll
                for (int c = 0; c < NCLS; ++c) m[c] = acc[c][tm][r];
                wino_out<KIND>(m, y);
#pragma unroll
                for (int i = 0; i < R; ++i) slab[(size_t)dm * mstride + i * 64] = y[i];
            }
        S3R_TL_FINISH();
        return;
    } else {
        // ---- epilogue: per-cout constants through LDS (every wave is past the last barrier: the ring is idle)
        float* ep_sc = wsmem;
        float* ep_sf = wsmem + WBM;
        float* ep_hw = wsmem + 2 * WBM;
        if (tid < WBM) {
            const int m = m0 + tid;
            ep_sc[tid] = (p.scale && m < p.Cout) ? p.scale[m] : 1.f;
            ep_sf[tid] = (p.shift && m < p.Cout) ? p.shift[m] : 0.f;
            if constexpr (HEAD) ep_hw[tid] = (p.head_w && m < p.Cout) ? p.head_w[m] : 0.f;
        }
        __syncthreads();
        const int n = n0 + wn * 32 + j;
        const bool ok = n < n_end;
        int e0, row1 = R;                                    // first output element of the group; rows of it that exist
        {
            const int nn = ok ? n : 0;
            const int b = p.dS.div(nn);
            int rem = nn - b * S;
            const int pd = p.dHW.div(rem);
            rem -= pd * p.Nh * p.Nw;
            const int q = p.dW.div(rem);
            const int pw = rem - q * p.Nw;
            if constexpr (DECONV)    // outputs (2 (2 pd + u) + rd, 2 (2 q + v) + rh, 2 pw + rw), u, v = 0, 1
                e0 = b * p.y_bs + p.y_org + (2 * pd * p.y_ds + 2 * q * p.y_hs + pw) * 2 + rd * p.y_ds + rh * p.y_hs + rw;
            else {
                e0 = b * p.y_bs + p.y_org + pd * p.y_ds + R * q * p.y_hs + pw;
                row1 = p.Hout - R * q;
            }
        }
        const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
        if constexpr (HEAD) {
            // the workgroup's 64 couts are the whole channel axis: this wave holds 32 of them (16 in this lane, 16 in lane j + 32),
            // the wave beside it (wm ^ 1, same wn) the other 32: in-lane chains, the lane halves' sum, then the two waves' sum
            // through LDS (wino_finish_kernel<2, true> walks the couts in this order)
            float t0 = 0.f, t1 = 0.f, t2 = 0.f, t3 = 0.f;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm], hw = ep_hw[mbase + dm];
                const float m[9] = {acc[0][0][r], acc[NACC > 1 ? 1 : 0][0][r], acc[NACC > 2 ? 2 : 0][0][r],
                                    acc[NACC > 3 ? 3 : 0][0][r], acc[NACC > 4 ? 4 : 0][0][r], acc[NACC > 5 ? 5 : 0][0][r],
                                    acc[NACC > 6 ? 6 : 0][0][r], acc[NACC > 7 ? 7 : 0][0][r], acc[NACC > 8 ? 8 : 0][0][r]};
                float y[4];
                wino_out<2>(m, y);
                t0 = fmaf(wino_act(y[0], sc, sf, lo), hw, t0);
                t1 = fmaf(wino_act(y[1], sc, sf, lo), hw, t1);
                t2 = fmaf(wino_act(y[2], sc, sf, lo), hw, t2);
                t3 = fmaf(wino_act(y[3], sc, sf, lo), hw, t3);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            t0 += __shfl_xor(t0, 32, 64);
            t1 += __shfl_xor(t1, 32, 64);
            t2 += __shfl_xor(t2, 32, 64);
            t3 += __shfl_xor(t3, 32, 64);
            float* ex = wsmem + 4 * WBM + (wn * 32 + j) * 4;          // behind the epilogue constants
            if (wm == 1 && h == 0) { ex[0] = t0; ex[1] = t1; ex[2] = t2; ex[3] = t3; }
            __syncthreads();
            if (wm == 0 && h == 0 && ok) {
                const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
                p.y[e0] = wino_head_act(fmaf(t0 + ex[0], hsc, hsf), p.head_act);
                p.y[e0 + wino_out_off<KIND>(1, p.y_ds, p.y_hs)] = wino_head_act(fmaf(t1 + ex[1], hsc, hsf), p.head_act);
                p.y[e0 + wino_out_off<KIND>(2, p.y_ds, p.y_hs)] = wino_head_act(fmaf(t2 + ex[2], hsc, hsf), p.head_act);
                p.y[e0 + wino_out_off<KIND>(3, p.y_ds, p.y_hs)] = wino_head_act(fmaf(t3 + ex[3], hsc, hsf), p.head_act);
            }
        } else {
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
            const int yvo0 = (e0 + (m0 + mbase) * p.y_cs) * 4;
            const int row_bytes = p.y_cs * 4;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = ((r & 3) + 8 * (r >> 2)) * TM + tm;
                    if (dm >= mlimit) continue;
                    const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm];
                    float m[NCLS], y[R];
#pragma unroll
                    for (int c = 0; c < NCLS; ++c) m[c] = acc[c][tm][r];
                    wino_out<KIND>(m, y);
                    const int so = dm * row_bytes;
                    if (ok) {
#pragma unroll
                        for (int i = 0; i < R; ++i)
                            if (i < row1) {
                                const float v = wino_act(y[i], sc, sf, lo);
                                __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrsrc,
                                                                      yvo0 + 4 * wino_out_off<KIND>(i, p.y_ds, p.y_hs), so, 0);
                            }
                    }
                    if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
                }
        }
        S3R_TL_FINISH();
    }
}
#undef S3R_TL_FINISH

// registers: six classes of one 32 x 32 tile = 96 accumulators (109 in all: four workgroups per CU), nine = 144 (three per CU);
// the class-parallel form 16
template <int VEC, int KIND, int WN, bool CP, bool HEAD, bool XM = false, bool SEMI = false>
__global__ __launch_bounds__(256, (CP || KIND == 1 ? 4 : 3)) void wino_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsmem[];
    wino_body<VEC, KIND, WN, CP, HEAD, XM, SEMI>(p, blockIdx.x, gridDim.x, p.n_begin, p.n_end, wsmem);
}

// bulk (serial form, positions [n_begin, n_cut)) + remainder (class-parallel form, positions [n_cut, n_end)) in ONE launch: the
// remainder's short workgroups fill the slots the bulk's last round leaves
template <int VEC, int KIND, bool HEAD = false, bool XM = false>
__global__ __launch_bounds__(256, (KIND == 1 ? 4 : 3)) void wino_dual_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsmem[];
    if ((int)blockIdx.x < p.big_wgs)
        wino_body<VEC, KIND, 2, false, HEAD, XM>(p, blockIdx.x, p.big_wgs, p.n_begin, p.n_cut, wsmem);
    else
        wino_body<VEC, KIND, 2, true, false, XM>(p, (int)blockIdx.x - p.big_wgs, (int)gridDim.x - p.big_wgs, p.n_cut, p.n_end, wsmem);
}

// ---- class-parallel finish: y = act(transform(class sums) * scale + shift) over positions [n_begin, n_end), through the same
// wino_out / wino_act as the serial epilogue.  One thread per (parity class,) cout and position; HEAD: one thread per position
// walks the couts in the serial epilogue's order (the two lane halves' 32-cout chains, then their sum).
template <int KIND, bool HEAD>
__global__ __launch_bounds__(256) void wino_finish_kernel(const ConvParams p, const int n_begin, const int n_end, const int npad) {
    constexpr bool DECONV = KIND == 2;
    constexpr int NCLS = WinoKind<KIND>::NCLS, R = WinoKind<KIND>::R;
    const int S = p.Nd * p.Nh * p.Nw;
    const int nn = n_end - n_begin;
    const int pc = DECONV ? (int)blockIdx.y : 0;
    const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    // slabs blocked [cout][64-position tile][class][64] (wino_body, CP): class c of (cout mm, position nl) at blk(mm, nl)[c * 64]
    constexpr int NBLK = DECONV ? 8 * NCLS : NCLS;
    const int ntl = npad >> 6;
    const float* __restrict__ base = p.part + (size_t)pc * NCLS * 64;       // (pc = 0 for convolutions)
    auto blk = [&](int mm, int nl) { return base + ((size_t)mm * ntl + (nl >> 6)) * (NBLK * 64) + (nl & 63); };
    const long long total = HEAD ? nn : (long long)p.Cout * nn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int mrow = HEAD ? 0 : (int)(i / nn);
        const int nl = (int)(i - (long long)mrow * nn);
        const int n = n_begin + nl;
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        int e0, row1 = R;
        if constexpr (DECONV)
            e0 = b * p.y_bs + p.y_org + (2 * pd * p.y_ds + 2 * q * p.y_hs + pw) * 2 + rd * p.y_ds + rh * p.y_hs + rw;
        else {
            e0 = b * p.y_bs + p.y_org + pd * p.y_ds + R * q * p.y_hs + pw;
            row1 = p.Hout - R * q;
        }
        if constexpr (HEAD) {
            // the serial epilogue's order: per wave (couts 32 wm ..), per lane half (4 hh + ..), a chain over the 16 registers;
            // then the lane halves' sum, then the two waves' sum
            float tw[2][R];
#pragma unroll
            for (int wm = 0; wm < 2; ++wm) {
                float th[2][R];
#pragma unroll
                for (int hh = 0; hh < 2; ++hh) {
#pragma unroll
                    for (int k = 0; k < R; ++k) th[hh][k] = 0.f;
                    for (int r = 0; r < 16; ++r) {
                        const int mm = 32 * wm + 4 * hh + (r & 3) + 8 * (r >> 2);
                        if (mm >= p.Cout) continue;                    // (padded couts carry head weight 0: t + 0 = t)
                        const float sc = p.scale ? p.scale[mm] : 1.f, sf = p.shift ? p.shift[mm] : 0.f, hw = p.head_w[mm];
                        float m[NCLS], y[R];
#pragma unroll
                        for (int c = 0; c < NCLS; ++c) m[c] = blk(mm, nl)[c * 64];
                        wino_out<KIND>(m, y);
#pragma unroll
                        for (int k = 0; k < R; ++k) th[hh][k] = fmaf(wino_act(y[k], sc, sf, lo), hw, th[hh][k]);
                    }
                }
#pragma unroll
                for (int k = 0; k < R; ++k) tw[wm][k] = th[0][k] + th[1][k];
            }
            const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
#pragma unroll
            for (int k = 0; k < R; ++k)
                p.y[e0 + wino_out_off<KIND>(k, p.y_ds, p.y_hs)] = wino_head_act(fmaf(tw[0][k] + tw[1][k], hsc, hsf), p.head_act);
        } else {
            const float sc = p.scale ? p.scale[mrow] : 1.f, sf = p.shift ? p.shift[mrow] : 0.f;
            float m[NCLS], y[R];
#pragma unroll
            for (int c = 0; c < NCLS; ++c) m[c] = blk(mrow, nl)[c * 64];
            wino_out<KIND>(m, y);
            float* __restrict__ yo = p.y + (size_t)mrow * p.y_cs + e0;
#pragma unroll
            for (int k = 0; k < R; ++k)
                if (k < row1) yo[wino_out_off<KIND>(k, p.y_ds, p.y_hs)] = wino_act(y[k], sc, sf, lo);
        }
    }
}

template <int KIND>
static hipError_t launch_wino_finish(const ConvParams& p, int n_begin, int n_end, int npad, hipStream_t stream) {
    const long long total = p.head_w ? (long long)(n_end - n_begin) : (long long)p.Cout * (n_end - n_begin);
    const long long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 8192 ? blocks : 8192), KIND == 2 ? 8 : 1);
    if constexpr (KIND == 2) {
        if (p.head_w) hipLaunchKernelGGL((wino_finish_kernel<KIND, true>), grid, dim3(256), 0, stream, p, n_begin, n_end, npad);
        else hipLaunchKernelGGL((wino_finish_kernel<KIND, false>), grid, dim3(256), 0, stream, p, n_begin, n_end, npad);
    } else {
        hipLaunchKernelGGL((wino_finish_kernel<KIND, false>), grid, dim3(256), 0, stream, p, n_begin, n_end, npad);
    }
    return hipGetLastError();
}

// ---- launch planning ---------------------------------------------------------------------------
constexpr int WCN = 64;                                  // positions per tile of the serial / class-parallel forms (head: WBN)
#ifdef S3R_ABLATE   // diagnostic builds: S3R_ABL=3 no operand DMA inside the K loop, 4 weights only, 5 activations only, 6 activations on a chunk's first column tap only
static int wabl_mode() { static const int m = getenv("S3R_ABL") ? atoi(getenv("S3R_ABL")) : 0; return m; }
#define S3R_WABL(p) (p).debug = wabl_mode()
#else
#define S3R_WABL(p) (void)0
#endif

int64_t wino_slab_elems(int kind, int cout, int ntotal, const WinoLaunch& L) {
    if (L.mode == WINO_SERIAL) return 0;
    const int ncls = kind == 2 ? 72 : 6;
    const int n0 = L.mode == WINO_DUAL ? L.n_cut : 0;
    const int64_t npad = (int64_t)((ntotal - n0 + WCN - 1) / WCN) * WCN;
    return (int64_t)ncls * cout * npad;
}

// Which form a layer's launch takes (all forms produce the same bits, so this may follow the batch).  forced >= 0 (tuning,
// tests): 0 serial, 1 class-parallel, 2 dual.  Otherwise a cost model in units of one K step of a 64 x 64 tile on one CU
// (13.4 ns at the fp32 MFMA peak), fitted to tools/layer_bench.py --algo 2 sweeps at B = 1 .. 32:
//   kcls = K per class (Cin x taps), u = serial workgroups per CU;
//   serial          ceil(u) workgroups of ncls * kcls steps each, at 0.9 of the pipe (a workgroup left alone on its CU: 0.7)
//   class-parallel  ceil(u * ncls) workgroups of kcls steps at 0.9, + 100 steps each for their slab's round trip, + a finish launch
//   dual            k serial workgroups per CU (k = floor(u), or one fewer when the last would run alone) + 0.6 of the remainder
//                   class-parallel
WinoLaunch wino_plan(int kind, int cout, int kcls, int ntotal, bool head, int forced) {
    WinoLaunch L;
    L.mode = WINO_SERIAL; L.n_cut = 0;
    const int m_tiles = (cout + WBM - 1) / WBM;
    const int pcs = kind == 2 ? 8 : 1;
    const int ncls = kind == 2 ? 9 : 6;
    const long W = (long)m_tiles * ((ntotal + WCN - 1) / WCN) * pcs;                 // serial workgroups (64-position tiles)
    if (forced < 0) {
        static const int env_mode = getenv("S3R_WINO_FORM") ? atoi(getenv("S3R_WINO_FORM")) : -1;      // A/B switch, read once
        forced = env_mode;
    }
    const int occ = kind == 2 ? 3 : 4;                   // serial workgroups a CU holds at once (registers)
    const int CUS = cu_count();
    const double u = (double)W / (double)CUS;            // serial workgroups per CU
    int kb = (int)u;                                     // bulk rounds of a dual launch: whole workgroups per CU
    if (forced >= 0) {
        L.mode = forced <= WINO_DUAL ? forced : WINO_SERIAL;
        if ((double)kb >= u) kb -= 1;                    // (a forced dual launch always leaves a remainder)
    } else {
        (void)head;
        constexpr double EFF = 0.9, EFF_ALONE = 0.7, SLAB = 100.0, LAUNCH = 220.0, TAIL = 0.6;
        const double unit = (double)ncls * kcls;
        // w workgroups per CU, `occ` at a time: the last one alone on its CU runs at EFF_ALONE
        auto serial_cost = [&](int w) {
            const int rem = w % occ;
            return (w - (rem == 1 ? 1 : 0)) * unit / EFF + (rem == 1 ? unit / EFF_ALONE : 0.0);
        };
        const double serial = serial_cost((int)__builtin_ceil(u));
        const double cp = __builtin_ceil(u * ncls) * (kcls / EFF + SLAB) + LAUNCH;
        double best = serial;
        if (cp < best) { best = cp; L.mode = WINO_CP; }
        // (the remainder's short workgroups run beside the bulk's last round: TAIL of their own time shows)
        // k = floor(u), or one fewer when that would leave a CU's last bulk workgroup alone
        for (int k : {(int)u, (int)u % occ == 1 ? (int)u - 1 : 0}) {
            if (k < 1 || (double)k >= u) continue;
            const double dual = serial_cost(k) + TAIL * __builtin_ceil((u - k) * ncls) * (kcls / EFF + SLAB) + LAUNCH;
            if (dual < 0.97 * best) { best = dual; L.mode = WINO_DUAL; kb = k; }
        }
    }
    if (L.mode == WINO_DUAL) {
        const long n_main = ((long)kb * CUS) / (m_tiles * pcs);                        // whole N tiles in the bulk
        L.n_cut = (int)(n_main * WCN);
        if (L.n_cut <= 0 || L.n_cut >= ntotal) { L.mode = L.n_cut <= 0 ? WINO_CP : WINO_SERIAL; L.n_cut = 0; }
    }
    return L;
}

template <int VEC, int KIND>
static hipError_t launch_wino_forms(ConvParams p, const WinoLaunch& L, hipStream_t stream) {
    const int ntotal = p.Ntotal;
    auto lds_of = [](int bn) { return (size_t)WNB * WBK * (WBM + (KIND == 2 ? 2 : 1) * bn) * sizeof(float); };
    const int pcs = KIND == 2 ? 8 : 1;
    constexpr int NCLS = WinoKind<KIND>::NCLS;
    p.n_begin = 0; p.n_end = ntotal;
    if (L.mode == WINO_SERIAL) {
        const dim3 grid(p.m_tiles * ((ntotal + WCN - 1) / WCN) * pcs);
        if constexpr (KIND == 2) {
            if (p.head_w) {
                if (p.xd_mode) hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, false, true, true>), grid, dim3(256), lds_of(WCN), stream, p);
                else hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, false, true, false>), grid, dim3(256), lds_of(WCN), stream, p);
            } else {
                if (p.xd_mode) hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, false, false, true>), grid, dim3(256), lds_of(WCN), stream, p);
                else hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, false, false, false>), grid, dim3(256), lds_of(WCN), stream, p);
            }
            return hipGetLastError();
        }
        hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, false, false>), grid, dim3(256), lds_of(WCN), stream, p);
        return hipGetLastError();
    }
    if (!p.part) return hipErrorInvalidValue;
    const int n0 = L.mode == WINO_DUAL ? L.n_cut : 0;
    const int cp_tiles = (ntotal - n0 + WCN - 1) / WCN;
    const int cp_wgs = p.m_tiles * cp_tiles * NCLS * pcs;
    if (L.mode == WINO_DUAL) {
        p.n_cut = n0;
        p.big_wgs = p.m_tiles * (n0 / WCN) * pcs;
        const dim3 grid(p.big_wgs + cp_wgs);
        if constexpr (KIND == 2) {
            if (p.head_w) {
                if (p.xd_mode) hipLaunchKernelGGL((wino_dual_kernel<VEC, KIND, true, true>), grid, dim3(256), lds_of(WCN), stream, p);
                else hipLaunchKernelGGL((wino_dual_kernel<VEC, KIND, true, false>), grid, dim3(256), lds_of(WCN), stream, p);
            } else {
                if (p.xd_mode) hipLaunchKernelGGL((wino_dual_kernel<VEC, KIND, false, true>), grid, dim3(256), lds_of(WCN), stream, p);
                else hipLaunchKernelGGL((wino_dual_kernel<VEC, KIND, false, false>), grid, dim3(256), lds_of(WCN), stream, p);
            }
        } else {
            hipLaunchKernelGGL((wino_dual_kernel<VEC, KIND>), grid, dim3(256), lds_of(WCN), stream, p);
        }
    } else if (KIND == 2 && p.xd_mode) {
        hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, true, false, KIND == 2>), dim3(cp_wgs), dim3(256), lds_of(WCN), stream, p);
    } else {
        hipLaunchKernelGGL((wino_kernel<VEC, KIND, 2, true, false>), dim3(cp_wgs), dim3(256), lds_of(WCN), stream, p);
    }
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    // (aux pass: reads the class slabs of the class-parallel range, writes that range's outputs)
    AuxScope aux(stream, 4.0 * (double)(ntotal - n0) * p.Cout * (pcs * NCLS + (p.head_w ? 0.0 : pcs * WinoKind<KIND>::R)));
    return launch_wino_finish<KIND>(p, n0, ntotal, cp_tiles * WCN, stream);
}

// p: see wino_body.  Returns the kernel launches made through *launches (may be null).
hipError_t launch_conv_wino(ConvParams p, const WinoLaunch& L, hipStream_t stream, int* launches) {
    if (p.Cin % WBK != 0 || p.stride != 1 || p.transposed || p.ksplit != 1 || p.head_w || p.act == ACT_SIGMOID)
        return hipErrorInvalidValue;
    p.kh = 1;
    S3R_WABL(p);
    p.m_tiles = (p.Cout + WBM - 1) / WBM;
    if (launches) *launches = L.mode == WINO_SERIAL ? 1 : 2;
    return p.Nw % 4 == 0 ? launch_wino_forms<4, 1>(p, L, stream) : launch_wino_forms<1, 1>(p, L, stream);
}

hipError_t launch_deconv_wino(ConvParams p, const WinoLaunch& L, hipStream_t stream, int* launches) {
    if (p.Cin % WBK != 0 || !p.transposed || p.ksplit != 1 || !p.xd || p.act == ACT_SIGMOID || (p.head_w && p.Cout > WBM))
        return hipErrorInvalidValue;
    S3R_WABL(p);
    p.m_tiles = (p.Cout + WBM - 1) / WBM;
    if (launches) *launches = L.mode == WINO_SERIAL ? 1 : 2;
    if (p.Nw % 4 != 0) return hipErrorInvalidValue;       // (16-byte gathers only: the library's policy asks for an edge % 4 == 0)
    return launch_wino_forms<4, 2>(p, L, stream);
}

// ================================================================================================
// Two-axis Winograd for the small 3D layers, CLASS-PARALLEL ONLY.  A class-parallel workgroup holds ONE class's accumulators,
// so nothing limits the number of classes: nesting the transform along D and H turns v5 (3 x 3 x 3 over 7^3) into 36 class
// convolutions with a 1 x 1 x 3 kernel for 4 x 4 outputs — F(4,3) x F(4,3): 108 multiply-adds per 16 outputs where the direct
// form spends 432 and the one-axis form 216 — and v6 (4 x 4 x 4, valid, 7^3 -> 4^3) into 25 classes with a 1 x 1 x 4 kernel for
// 2 x 2 outputs: F(2,4) x F(2,4), 100 per 4 outputs instead of 256.  The class GEMM is the one-axis kernel's class-parallel form
// unchanged (wino_kernel<VEC, 1, 2, true>: a class is a plane set + a weight slab, K = Cin x kw); what is new is the input
// transform (a window of n x n rows per group), the weight transform (G x G) and the finish kernel (A^T along H, then D).
// Slabs cost ncls / (m m) x the output's bytes, which is why only layers with small outputs take it (v5: 11 MB -> 32 MB).
//   F(2,4), points 0, 1, -1, 2, inf (tools/wino_matrices.py derives and checks both sets in exact arithmetic):
//     B^T rows (2,-1,-2,1,0) (0,-2,-1,1,0) (0,2,-3,1,0) (0,-1,0,1,0) (0,2,-1,-2,1)
//     G rows (1/2,0,0,0) (-1/2,-1/2,-1/2,-1/2) (-1/6,1/6,-1/6,1/6) (1/6,1/3,2/3,4/3) (0,0,0,1)     A^T rows (1,1,1,1,0) (0,1,-1,2,1)
template <int AX> struct WAxis;
template <> struct WAxis<0> { static constexpr int N = 6, M = 4, R = 3; };      // F(4,3)
template <> struct WAxis<1> { static constexpr int N = 5, M = 2, R = 4; };      // F(2,4)
int wino2_classes(int ax) { return ax == 1 ? 25 : 36; }      // ax 2: the 2D layers' (H, W) nesting, F(4,3) x F(4,3) again
int wino2_outputs(int ax) { return ax == 1 ? 2 : 4; }

template <int AX>
__device__ __forceinline__ void wax_bt(const float (&r)[WAxis<AX>::N], float (&v)[WAxis<AX>::N]) {
    if constexpr (AX == 0) wino_rows_to_classes<4>(r, v);
    else {
        v[0] = fmaf(2.f, r[0] - r[2], r[3] - r[1]);
        v[1] = fmaf(-2.f, r[1], r[3] - r[2]);
        v[2] = fmaf(2.f, r[1], fmaf(-3.f, r[2], r[3]));
        v[3] = r[3] - r[1];
        v[4] = fmaf(2.f, r[1] - r[3], r[4] - r[2]);
    }
}
template <int AX>
__device__ __forceinline__ void wax_at(const float (&m)[WAxis<AX>::N], float (&y)[WAxis<AX>::M]) {
    if constexpr (AX == 0) wino_out<1>(m, y);
    else {
        y[0] = ((m[0] + m[1]) + m[2]) + m[3];
        y[1] = fmaf(2.f, m[3], m[1] - m[2]) + m[4];
    }
}
// transformed weight of class c from the axis' R kernel values
template <int AX>
__device__ __forceinline__ float wax_g(int c, const float (&g)[WAxis<AX>::R]) {
    if constexpr (AX == 0) {
        const float s02 = g[0] + g[2], a = g[0] * (1.f / 24.f) + g[2] * (1.f / 6.f), b12 = g[1] * (1.f / 12.f);
        return c == 0 ? g[0] * 0.25f : c == 1 ? (s02 + g[1]) * (-1.f / 6.f) : c == 2 ? (s02 - g[1]) * (-1.f / 6.f)
             : c == 3 ? a + b12 : c == 4 ? a - b12 : g[2];
    } else {
        const float e = g[0] + g[2], o = g[1] + g[3];
        return c == 0 ? g[0] * 0.5f : c == 1 ? (e + o) * -0.5f : c == 2 ? (o - e) * (1.f / 6.f)
             : c == 3 ? fmaf(g[3], 4.f / 3.f, fmaf(g[2], 2.f / 3.f, fmaf(g[1], 1.f / 3.f, g[0] * (1.f / 6.f)))) : g[3];
    }
}

// one thread per (plane, depth group, row group, column): an N x N window of the input, B^T along H then along D
template <int AX>
__global__ __launch_bounds__(256) void wino2_input_kernel(const float* __restrict__ x, float* __restrict__ V, unsigned total,
                                                          int Dp, int Hp, int Wp, int SD, int SH, FastDiv dW, FastDiv dSH, FastDiv dSD) {
    constexpr int N = WAxis<AX>::N, M = WAxis<AX>::M;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned row = (unsigned)dW.div((int)i);                   // ((plane, sd), sh)
        const int w = (int)(i - row * (unsigned)Wp);
        const unsigned pg = (unsigned)dSH.div((int)row);                 // (plane, sd)
        const int sh = (int)(row - pg * (unsigned)SH);
        const unsigned pl = (unsigned)dSD.div((int)pg);
        const int sd = (int)(pg - pl * (unsigned)SD);
        const float* __restrict__ src = x + ((size_t)pl * Dp + M * sd) * Hp * Wp + (size_t)M * sh * Wp + w;
        float t[N][N];                                                   // [depth][row class]
#pragma unroll
        for (int a = 0; a < N; ++a) {
            float r[N], v[N];
#pragma unroll
            for (int b = 0; b < N; ++b)
                r[b] = (M * sd + a < Dp && M * sh + b < Hp) ? src[((size_t)a * Hp + b) * Wp] : 0.f;
            wax_bt<AX>(r, v);
#pragma unroll
            for (int b = 0; b < N; ++b) t[a][b] = v[b];
        }
#pragma unroll
        for (int b = 0; b < N; ++b) {
            float r[N], v[N];
#pragma unroll
            for (int a = 0; a < N; ++a) r[a] = t[a][b];
            wax_bt<AX>(r, v);
#pragma unroll
            for (int a = 0; a < N; ++a) V[(size_t)(a * N + b) * total + i] = v[a];
        }
    }
}

hipError_t launch_wino2_input(const float* x, float* V, int ax, long long planes, int Dp, int Hp, int Wp, int SD, int SH, hipStream_t s) {
    const long long total = planes * SD * SH * Wp;
    if (total >= (1ll << 31) || (ax != 0 && ax != 1)) return hipErrorInvalidValue;
    const long long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
    const FastDiv dW((unsigned)Wp), dSH((unsigned)SH), dSD((unsigned)SD);
    if (ax == 0) hipLaunchKernelGGL(wino2_input_kernel<0>, grid, dim3(256), 0, s, x, V, (unsigned)total, Dp, Hp, Wp, SD, SH, dW, dSH, dSD);
    else hipLaunchKernelGGL(wino2_input_kernel<1>, grid, dim3(256), 0, s, x, V, (unsigned)total, Dp, Hp, Wp, SD, SH, dW, dSH, dSD);
    return hipGetLastError();
}

// w[Cout][Cin][k][k][k] (k = R) -> Up[cls = a N + b][(chunk*k + tw)*32 + c][CoutPad]: G along H, then along D
template <int AX>
__global__ void pack_wino2_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad) {
    constexpr int N = WAxis<AX>::N, R = WAxis<AX>::R;
    const size_t per_cls = (size_t)R * Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < (size_t)N * N * per_cls; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_cls);
        const int ca = cls / N, cb = cls - ca * N;
        size_t r = i % per_cls;
        const int co = (int)(r % CoutPad);
        r /= CoutPad;
        const int c = (int)(r % WBK);
        r /= WBK;
        const int tw = (int)(r % R);
        const int cc = (int)(r / R);
        const int cin = cc * WBK + c;
        float v = 0.f;
        if (co < Cout) {
            const float* g = w + ((size_t)co * Cin + cin) * R * R * R + tw;       // g[(kd * R + kh) * R]
            float gd[R];
#pragma unroll
            for (int kd = 0; kd < R; ++kd) {
                float gh[R];
#pragma unroll
                for (int kh = 0; kh < R; ++kh) gh[kh] = g[(kd * R + kh) * R];
                gd[kd] = wax_g<AX>(cb, gh);
            }
            v = wax_g<AX>(ca, gd);
        }
        wp[i] = v;
    }
}

__global__ void pack_wino2p_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad);    // (ax 2, below)
hipError_t launch_pack_wino2(const float* w, float* wp, int ax, int Cin, int Cout, int CoutPad, hipStream_t s) {
    if (ax == 0) hipLaunchKernelGGL(pack_wino2_kernel<0>, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    else if (ax == 1) hipLaunchKernelGGL(pack_wino2_kernel<1>, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    else if (ax == 2) hipLaunchKernelGGL(pack_wino2p_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    else return hipErrorInvalidValue;
    return hipGetLastError();
}

// Finish kernels.  The output tensors carry their consumer's halo, so rows are (W + 2 halo) floats long: a kernel that stores
// interiors only leaves EVERY 128-byte line partly written, and partial lines cost the memory side a read-modify-write (measured on
// v1's finish: its stores alone ran at 2.5 TB/s, its loads at 6).  So a workgroup builds whole padded slices in LDS — halo zeros
// included — and writes them out as contiguous 16-byte stores: the halo is rewritten with the zeros it already holds.
__device__ __forceinline__ void wino2_copy_out(const float* __restrict__ lds, float* __restrict__ dst, int count, int tid) {
    if ((count & 3) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
        for (int i = tid * 4; i < count; i += 1024) *reinterpret_cast<wv4f*>(dst + i) = *reinterpret_cast<const wv4f*>(lds + i);
    } else {
        for (int i = tid; i < count; i += 256) dst[i] = lds[i];
    }
}
__device__ __forceinline__ void wino2_zero_out(float* __restrict__ dst, int count, int tid) {
    if ((count & 3) == 0 && (reinterpret_cast<size_t>(dst) & 15) == 0) {
        for (int i = tid * 4; i < count; i += 1024) *reinterpret_cast<wv4f*>(dst + i) = wv4f{0.f, 0.f, 0.f, 0.f};
    } else {
        for (int i = tid; i < count; i += 256) dst[i] = 0.f;
    }
}

// 3D, semi-fused form: one workgroup per (cout, SUB consecutive (sample, depth group) pairs — as many as give its 256 threads a
// group each): slabs [a][row][Cout][npad] (the row transform already applied by the class kernel) -> 4 output slices per pair:
// A^T along D, folded BN + activation.  y: halo-padded, y_hs = W + 2 halo, y_ds = a slice
__global__ __launch_bounds__(256) void wino2s_finish_kernel(const ConvParams p, const int npad, const int halo, const int SUB,
                                                            const FastDiv dNd, const FastDiv dMS) {
    constexpr int N = 6, M = 4;
    extern __shared__ __attribute__((aligned(16))) float fsm[];          // SUB x M slices of Hp x Wp
    const int tid = threadIdx.x;
    const int G = p.Nh * p.Nw, Wp = p.y_hs, slice = p.y_ds, MS = M * slice;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const int per_c = p.B * p.Nd;
    const int nrg = (per_c + SUB - 1) / SUB;
    const bool vec = (slice & 3) == 0 && (reinterpret_cast<size_t>(p.y) & 15) == 0;
    for (int item = blockIdx.x; item < p.Cout * nrg; item += gridDim.x) {
        const int mrow = item / nrg;
        const int r0 = (item - mrow * nrg) * SUB;                        // first (sample, depth group) pair
        const int nsub = per_c - r0 < SUB ? per_c - r0 : SUB;
        for (int i = tid; i < nsub * MS; i += 256) fsm[i] = 0.f;
        __syncthreads();
        const float sc = p.scale ? p.scale[mrow] : 1.f, sf = p.shift ? p.shift[mrow] : 0.f;
        for (int t = tid; t < nsub * G; t += 256) {
            const int sub = p.dHW.div(t);
            const int g = t - sub * G;
            const int sh = p.dW.div(g);
            const int pw = g - sh * p.Nw;
            const int r = r0 + sub;
            const int sd = r - dNd.div(r) * p.Nd;
            const int n = r0 * G + t;                                  // (slabs blocked by (cout, 64-position tile): wino_body, SEMI)
            const float* __restrict__ src = p.part + ((size_t)mrow * (npad >> 6) + (n >> 6)) * (N * M * 64) + (n & 63);
            float* __restrict__ o = fsm + sub * MS + (halo + M * sh) * Wp + halo + pw;
#pragma unroll
            for (int v = 0; v < M; ++v) {
                float m[N], y[M];
#pragma unroll
                for (int a = 0; a < N; ++a) m[a] = src[(a * M + v) * 64];
                wax_at<0>(m, y);
#pragma unroll
                for (int u = 0; u < M; ++u)
                    if (M * sd + u < p.Dout && M * sh + v < p.Hout) o[u * slice + v * Wp] = wino_act(y[u], sc, sf, lo);
            }
        }
        __syncthreads();
        float* __restrict__ yc = p.y + (size_t)mrow * p.y_cs;
        // every pair's slices are one contiguous run of the output (the depth halo slices of a sample's first / last pair with them)
        for (int e = vec ? tid * 4 : tid; e < nsub * MS; e += vec ? 1024 : 256) {
            const int sub = dMS.div(e);
            const int off = e - sub * MS;
            const int r = r0 + sub;
            const int b = dNd.div(r);
            const int sd = r - b * p.Nd;
            const int nsl = p.Dout - M * sd < M ? p.Dout - M * sd : M;
            if (off >= nsl * slice) continue;
            float* __restrict__ dst = yc + (size_t)b * p.y_bs + (size_t)(halo + M * sd) * slice + off;
            if (vec) *reinterpret_cast<wv4f*>(dst) = *reinterpret_cast<const wv4f*>(fsm + e);
            else *dst = fsm[e];
        }
        if (halo > 0)
            for (int sub = 0; sub < nsub; ++sub) {
                const int r = r0 + sub;
                const int b = dNd.div(r);
                const int sd = r - b * p.Nd;
                if (sd == 0) wino2_zero_out(yc + (size_t)b * p.y_bs, halo * slice, tid);
                if (sd == p.Nd - 1) wino2_zero_out(yc + (size_t)b * p.y_bs + (size_t)(halo + p.Dout) * slice, halo * slice, tid);
            }
        __syncthreads();
    }
}

// The class-parallel form of the 3D layers keeps the flat finish kernel — one thread per (cout, position), interiors only: its
// outputs are small (v3: 45 MB, v5 / v6: 11 / 4 MB) and the staged kernel's zero / barrier / copy chain costs more than the
// partial lines do (v3 42 -> 44 us, v5 11 -> 18, v6 10 -> 19).
// slabs [cls][Cout][npad] -> y: A^T along H, then along D, folded BN + activation; outputs beyond the true edge are not stored
template <int AX>
__global__ __launch_bounds__(256) void wino2_finish_flat_kernel(const ConvParams p, const int npad) {
    constexpr int N = WAxis<AX>::N, M = WAxis<AX>::M;
    const int S = p.Nd * p.Nh * p.Nw;
    const int nn = p.Ntotal;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const long long total = (long long)p.Cout * nn;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int mrow = (int)(i / nn);
        const int n = (int)(i - (long long)mrow * nn);
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int sd = p.dHW.div(rem);
        rem -= sd * p.Nh * p.Nw;
        const int sh = p.dW.div(rem);
        const int pw = rem - sh * p.Nw;
        const float* __restrict__ src = p.part + ((size_t)mrow * (npad >> 6) + (n >> 6)) * (N * N * 64) + (n & 63);     // (blocked slabs)
        float t[N][M];                                                   // [depth class][output row]
#pragma unroll
        for (int a = 0; a < N; ++a) {
            float m[N], y[M];
#pragma unroll
            for (int c = 0; c < N; ++c) m[c] = src[(a * N + c) * 64];
            wax_at<AX>(m, y);
#pragma unroll
            for (int v = 0; v < M; ++v) t[a][v] = y[v];
        }
        const float sc = p.scale ? p.scale[mrow] : 1.f, sf = p.shift ? p.shift[mrow] : 0.f;
        float* __restrict__ yo = p.y + (size_t)b * p.y_bs + (size_t)mrow * p.y_cs + p.y_org + (size_t)M * sd * p.y_ds + M * sh * p.y_hs + pw;
#pragma unroll
        for (int v = 0; v < M; ++v) {
            float m[N], y[M];
#pragma unroll
            for (int a = 0; a < N; ++a) m[a] = t[a][v];
            wax_at<AX>(m, y);
#pragma unroll
            for (int u = 0; u < M; ++u)
                if (M * sd + u < p.Dout && M * sh + v < p.Hout) yo[(size_t)u * p.y_ds + v * p.y_hs] = wino_act(y[u], sc, sf, lo);
        }
    }
}

// ---- the same nesting for a 2D layer, over H and W (ax = 2): 36 classes of a 1 x 1 kernel — K = Cin, 36 multiply-adds per 16
// outputs where the direct form spends 144 and the one-axis form 72.  With no tap left to shift along W the positions of the whole
// sub-batch are laid out FLAT: V[cls = a 6 + b][Cin][npad], n = (sample, row group, column group), a = the column class, b = the
// row class — the class GEMM sees one "sample" npad columns wide, so every gather is 16-byte even over a 7 x 7 grid of groups.
// one thread per (cin, sample, row group, column group), positions innermost — a wave's 36 stores are 256 contiguous bytes each
// (the windows it reads lie in a few 3.6 KB planes: L1 / L2 hits): a 6 x 6 window, B^T along H then along W
__global__ __launch_bounds__(256) void wino2p_input_kernel(const float* __restrict__ x, float* __restrict__ V, unsigned total, int Cin,
                                                           int Hp, int Wp, int SH, int SW, unsigned npad, unsigned nn, FastDiv dSW,
                                                           FastDiv dSH, FastDiv dN) {
    constexpr int N = 6, M = 4;
    const size_t cstride = (size_t)Cin * npad;
    for (unsigned i = blockIdx.x * 256u + threadIdx.x; i < total; i += gridDim.x * 256u) {
        const unsigned c = (unsigned)dN.div((int)i);                     // i = c nn + n, n = (sample, sh, sw)
        const unsigned n = i - c * nn;
        const unsigned row = (unsigned)dSW.div((int)n);                  // (sample, sh)
        const int sw = (int)(n - row * (unsigned)SW);
        const unsigned b = (unsigned)dSH.div((int)row);
        const int sh = (int)(row - b * (unsigned)SH);
        const float* __restrict__ src = x + (((size_t)b * Cin + c) * Hp + M * sh) * Wp + M * sw;
        float t[N][N];                                                   // [row class][column]
#pragma unroll
        for (int jc = 0; jc < N; ++jc) {
            float r[N], v[N];
#pragma unroll
            for (int ir = 0; ir < N; ++ir) r[ir] = (M * sh + ir < Hp && M * sw + jc < Wp) ? src[(size_t)ir * Wp + jc] : 0.f;
            wax_bt<0>(r, v);
#pragma unroll
            for (int ir = 0; ir < N; ++ir) t[ir][jc] = v[ir];
        }
        float* __restrict__ dst = V + (size_t)c * npad + n;
#pragma unroll
        for (int bb = 0; bb < N; ++bb) {
            float v[N];
            wax_bt<0>(t[bb], v);
#pragma unroll
            for (int a = 0; a < N; ++a) dst[(size_t)(a * N + bb) * cstride] = v[a];
        }
    }
}

// the same transform with the planes staged in LDS: a workgroup owns one channel of NS samples (NS x SH x SW <= 256 groups), loads
// their padded planes with 16-byte accesses and reads the 6 x 6 windows from LDS (the flat kernel's 36 loads per thread are 16 bytes
// apart between lanes: 3.2 TB/s); same operations per value, same bits
__global__ __launch_bounds__(256) void wino2p_input_lds_kernel(const float* __restrict__ x, float* __restrict__ V, int B, int Cin, int Hp,
                                                               int Wp, int SH, int SW, int NS, unsigned npad) {
    extern __shared__ __attribute__((aligned(16))) float w2p_sx[];
    constexpr int N = 6, M = 4;
    const int c = blockIdx.x % Cin, b0 = (blockIdx.x / Cin) * NS;
    const int ns = B - b0 < NS ? B - b0 : NS;
    const int plane = Hp * Wp, G = SH * SW;
    for (int sidx = 0; sidx < ns; ++sidx) {
        const float* __restrict__ src = x + ((size_t)(b0 + sidx) * Cin + c) * plane;
        float* dstp = w2p_sx + sidx * plane;
        if ((plane & 3) == 0) {
            for (int i = threadIdx.x; i < plane / 4; i += 256)
                reinterpret_cast<float4*>(dstp)[i] = reinterpret_cast<const float4*>(src)[i];
        } else {
            for (int i = threadIdx.x; i < plane; i += 256) dstp[i] = src[i];
        }
    }
    __syncthreads();
    const int t = threadIdx.x;
    if (t >= ns * G) return;
    const int sidx = t / G, g = t - sidx * G;
    const int sh = g / SW, sw = g - sh * SW;
    const float* __restrict__ src = w2p_sx + sidx * plane + (M * sh) * Wp + M * sw;
    const size_t cstride = (size_t)Cin * npad;
    float tt[N][N];                                                      // [row class][column]
#pragma unroll
    for (int jc = 0; jc < N; ++jc) {
        float r[N], v[N];
#pragma unroll
        for (int ir = 0; ir < N; ++ir) r[ir] = (M * sh + ir < Hp && M * sw + jc < Wp) ? src[ir * Wp + jc] : 0.f;
        wax_bt<0>(r, v);
#pragma unroll
        for (int ir = 0; ir < N; ++ir) tt[ir][jc] = v[ir];
    }
    float* __restrict__ dst = V + (size_t)c * npad + (size_t)(b0 + sidx) * G + g;
#pragma unroll
    for (int bb = 0; bb < N; ++bb) {
        float v[N];
        wax_bt<0>(tt[bb], v);
#pragma unroll
        for (int a2 = 0; a2 < N; ++a2) dst[(size_t)(a2 * N + bb) * cstride] = v[a2];
    }
}

hipError_t launch_wino2p_input(const float* x, float* V, int B, int Cin, int Hp, int Wp, int SH, int SW, long long npad, hipStream_t s) {
    const long long nn = (long long)B * SH * SW, total = nn * Cin;
    if (total >= (1ll << 31) || npad >= (1ll << 31)) return hipErrorInvalidValue;
    const int G = SH * SW;
    if (G <= 256) {
        int NS = 256 / G;
        if (NS > B) NS = B;
        while (NS > 1 && (size_t)NS * Hp * Wp * 4 > 48 * 1024) --NS;
        const size_t lds = (size_t)NS * Hp * Wp * 4;
        if (lds <= 48 * 1024) {
            const long long blocks = (long long)Cin * ((B + NS - 1) / NS);
            if (blocks < (1ll << 31)) {
                hipLaunchKernelGGL(wino2p_input_lds_kernel, dim3((unsigned)blocks), dim3(256), lds, s, x, V, B, Cin, Hp, Wp, SH, SW, NS,
                                   (unsigned)npad);
                return hipGetLastError();
            }
        }
    }
    const long long blocks = (total + 255) / 256;
    const dim3 grid((unsigned)(blocks < 16384 ? blocks : 16384));
    hipLaunchKernelGGL(wino2p_input_kernel, grid, dim3(256), 0, s, x, V, (unsigned)total, Cin, Hp, Wp, SH, SW, (unsigned)npad, (unsigned)nn,
                       FastDiv((unsigned)SW), FastDiv((unsigned)SH), FastDiv((unsigned)nn));
    return hipGetLastError();
}

// w[Cout][Cin][3][3] -> Up[cls = a 6 + b][cin][CoutPad]: G along W (a), then along H (b)
__global__ void pack_wino2p_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad) {
    const size_t per_cls = (size_t)Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 36 * per_cls; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_cls);
        const int ca = cls / 6, cb = cls - ca * 6;
        const size_t r = i % per_cls;
        const int co = (int)(r % CoutPad);
        const int cin = (int)(r / CoutPad);
        float v = 0.f;
        if (co < Cout) {
            const float* g = w + ((size_t)co * Cin + cin) * 9;
            float gh[3];
#pragma unroll
            for (int kh = 0; kh < 3; ++kh) {
                const float gw[3] = {g[kh * 3], g[kh * 3 + 1], g[kh * 3 + 2]};
                gh[kh] = wax_g<0>(ca, gw);
            }
            v = wax_g<0>(cb, gh);
        }
        wp[i] = v;
    }
}

// 2D: one workgroup per (cout, PL consecutive samples): slabs -> PL whole padded planes.  SEMI: [a][row][Cout][npad] (A^T along H
// already applied by the class kernel), otherwise [a 6 + b][Cout][npad]: A^T along H, then along W.  y_hs = W + 2 halo, y_cs = a plane.
// TOV (S3R_LAYOUT_WINO_HW output; halo 1, edge a multiple of 4): the planes never leave LDS — the workgroup applies the NEXT layer's
// input transform to them (wino2p_input_kernel's arithmetic on the values it would have read back) and writes that layer's 36
// plane sets V[cls][cout][position]: the positions of its PL samples are one contiguous run per class
template <bool SEMI, bool TOV>
__global__ __launch_bounds__(256) void wino2p_finish_kernel(const ConvParams p, const int npad, const int halo, const int PL) {
    constexpr int N = 6, M = 4;
    extern __shared__ __attribute__((aligned(16))) float fsm[];          // PL planes of Hp x Wp
    const int tid = threadIdx.x;
    const int G = p.Nh * p.Nw, Wp = TOV ? p.Hout + 2 : p.y_hs, plane = TOV ? Wp * Wp : p.y_cs;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const size_t cstride = (size_t)p.Cout * npad;
    const int nbg = (p.B + PL - 1) / PL;
    for (int item = blockIdx.x; item < p.Cout * nbg; item += gridDim.x) {
        const int mrow = item / nbg;
        const int b0 = (item - mrow * nbg) * PL;
        const int npl = p.B - b0 < PL ? p.B - b0 : PL;
        for (int i = tid; i < npl * plane; i += 256) fsm[i] = 0.f;
        __syncthreads();
        const float sc = p.scale ? p.scale[mrow] : 1.f, sf = p.shift ? p.shift[mrow] : 0.f;
        for (int t = tid; t < npl * G; t += 256) {
            const int pl = p.dS.div(t);
            const int s = t - pl * G;
            const int sh = p.dW.div(s);
            const int sw = s - sh * p.Nw;
            const int n = b0 * G + t;
            const float* __restrict__ src = p.part + ((size_t)mrow * (npad >> 6) + (n >> 6)) * ((SEMI ? N * M : N * N) * 64) + (n & 63);
            float tt[N][M];                                              // [column class][output row]
#pragma unroll
            for (int a = 0; a < N; ++a) {
                if constexpr (SEMI) {                                    // (slabs blocked by (cout, 64-position tile))
#pragma unroll
                    for (int v = 0; v < M; ++v) tt[a][v] = src[(a * M + v) * 64];
                } else {
                    float m[N], y[M];
#pragma unroll
                    for (int c = 0; c < N; ++c) m[c] = src[(a * N + c) * 64];
                    wax_at<0>(m, y);
#pragma unroll
                    for (int v = 0; v < M; ++v) tt[a][v] = y[v];
                }
            }
            float* __restrict__ o = fsm + pl * plane + (halo + M * sh) * Wp + halo + M * sw;
#pragma unroll
            for (int v = 0; v < M; ++v) {
                float m[N], y[M];
#pragma unroll
                for (int a = 0; a < N; ++a) m[a] = tt[a][v];
                wax_at<0>(m, y);
#pragma unroll
                for (int u = 0; u < M; ++u)
                    if (M * sh + v < p.Hout && M * sw + u < p.Hout) o[v * Wp + u] = wino_act(y[u], sc, sf, lo);
            }
        }
        __syncthreads();
        if constexpr (TOV) {
            for (int t = tid; t < npl * G; t += 256) {
                const int pl = p.dS.div(t);
                const int s = t - pl * G;
                const int sh = p.dW.div(s);
                const int sw = s - sh * p.Nw;
                const float* __restrict__ win = fsm + pl * plane + M * sh * Wp + M * sw;     // 6 x 6 window of the padded plane
                float tt[N][N];                                          // [row class][column]
#pragma unroll
                for (int jc = 0; jc < N; ++jc) {
                    float r[N], v[N];
#pragma unroll
                    for (int ir = 0; ir < N; ++ir) r[ir] = win[ir * Wp + jc];
                    wax_bt<0>(r, v);
#pragma unroll
                    for (int ir = 0; ir < N; ++ir) tt[ir][jc] = v[ir];
                }
                float* __restrict__ dst = p.y + (size_t)mrow * npad + (size_t)b0 * G + t;
#pragma unroll
                for (int bb = 0; bb < N; ++bb) {
                    float v[N];
                    wax_bt<0>(tt[bb], v);
#pragma unroll
                    for (int a = 0; a < N; ++a) dst[(size_t)(a * N + bb) * cstride] = v[a];
                }
            }
        } else {
            for (int pl = 0; pl < npl; ++pl)
                wino2_copy_out(fsm + pl * plane, p.y + (size_t)(b0 + pl) * p.y_bs + (size_t)mrow * p.y_cs, plane, tid);
        }
        __syncthreads();
    }
}

// which form a two-axis launch takes (same bits): class-parallel (one workgroup per class: 36 / 25 per tile) or — F(4,3) x F(4,3)
// only — semi-fused (one workgroup per depth class walks its six row classes: a third less slab traffic, six times the work
// per workgroup: for grids that fill the chip several times over).  forced: -1 the plan, 0 class-parallel, 1 semi-fused
int wino2_form(int ax, int cout, int ntotal, int forced) {
    if (ax == 1) return 0;
    if (forced >= 0) return forced ? 1 : 0;
    static const int env_form = getenv("S3R_WINO2_FORM") ? atoi(getenv("S3R_WINO2_FORM")) : -1;      // A/B switch, read once
    if (env_form >= 0) return env_form ? 1 : 0;
    const long semi_wgs = (long)((cout + WBM - 1) / WBM) * ((ntotal + WCN - 1) / WCN) * 6;
    // 3D: >= two rounds of the chip's four slots per CU.  2D: a class is Cin / 32 K tiles only, too short a workgroup on its own —
    // semi-fused from half a round on
    return semi_wgs >= (ax == 2 ? 2 : 8) * (long)cu_count() ? 1 : 0;
}
int64_t wino2_npad(int64_t ntotal) { return (ntotal + WCN - 1) / WCN * WCN; }
int64_t wino2_slab_elems(int ax, int cout, int ntotal, int form) {
    const int64_t npad = (int64_t)((ntotal + WCN - 1) / WCN) * WCN;
    return (form ? 6 * 4 : wino2_classes(ax)) * (int64_t)cout * npad;
}

// p: the CLASS convolution (wino_body): x = V, x_cs / x_ds / x_hs / x_cls its strides, Nd / Nh = depth / row groups, kd = kh = 1,
// T = kw, ncls = the class count; part = slabs; y / Dout / Hout the layer's output
hipError_t launch_conv_wino2(ConvParams p, int ax, int form, bool to_v, hipStream_t stream, int* launches) {
    if (p.Cin % WBK != 0 || p.stride != 1 || p.transposed || p.ksplit != 1 || p.head_w || p.act == ACT_SIGMOID || !p.part ||
        ax < 0 || ax > 2 || p.ncls != wino2_classes(ax))
        return hipErrorInvalidValue;
    p.kh = 1;
    S3R_WABL(p);
    p.m_tiles = (p.Cout + WBM - 1) / WBM;
    p.n_begin = 0; p.n_end = p.Ntotal;
    const int n_tiles = (p.Ntotal + WCN - 1) / WCN;
    const size_t lds = (size_t)WNB * WBK * (WBM + WCN) * sizeof(float);
    // finish kernels: whole padded slices / planes through LDS (the output's halo comes back from its strides)
    const int halo = to_v ? 1 : (p.y_hs - p.Hout) / 2;
    if (to_v && (ax != 2 || p.Hout % 4 != 0)) return hipErrorInvalidValue;
    if (!to_v && (halo < 0 || p.y_hs != p.Hout + 2 * halo)) return hipErrorInvalidValue;
    if (ax == 2) {
        // p describes the layer for the finish kernel (Nh x Nw groups per sample); the class GEMM runs over the flat positions
        const int plane = to_v ? (p.Hout + 2) * (p.Hout + 2) : p.y_hs * p.y_hs, G = p.Nh * p.Nw;
        if ((!to_v && p.y_cs != plane) || (size_t)plane * 4 > 64 * 1024) return hipErrorInvalidValue;
        int PL = 256 / G < 1 ? 1 : 256 / G;
        if (PL > p.B) PL = p.B;
        while (PL > 1 && (size_t)PL * plane * 4 > 64 * 1024) --PL;
        const int npad = n_tiles * WCN;
        ConvParams q = p;
        q.B = 1; q.Nd = 1; q.Nh = 1; q.Nw = npad;
        q.dS = q.dHW = q.dW = FastDiv((unsigned)npad);
        q.kd = q.kh = q.kw = 1; q.T = 1;
        q.x_org = 0; q.x_ds = 0; q.x_hs = 0; q.x_cs = npad; q.x_cls = p.Cin * npad;
        q.x_bytes = (unsigned)(4ull * 36 * p.Cin * npad);
        q.Ntotal = npad; q.n_end = npad;
        const dim3 grid(p.m_tiles * n_tiles * (form == 1 ? 6 : 36));
        if (form == 1) hipLaunchKernelGGL((wino_kernel<4, 1, 2, false, false, false, true>), grid, dim3(256), lds, stream, q);
        else hipLaunchKernelGGL((wino_kernel<4, 1, 2, true, false>), grid, dim3(256), lds, stream, q);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        const long long items = (long long)p.Cout * ((p.B + PL - 1) / PL);
        const dim3 fgrid((unsigned)(items < 16384 ? items : 16384));
        const size_t flds = (size_t)PL * plane * 4;
        // (aux pass: reads the slabs, writes the output planes — or, to_v, the consumer's 36 plane sets)
        AuxScope aux(stream, 4.0 * (double)p.Cout * ((double)(form == 1 ? 24 : 36) * npad + (to_v ? 36.0 * npad : (double)p.B * plane)));
        if (to_v) {
            if (form == 1) hipLaunchKernelGGL((wino2p_finish_kernel<true, true>), fgrid, dim3(256), flds, stream, p, npad, halo, PL);
            else hipLaunchKernelGGL((wino2p_finish_kernel<false, true>), fgrid, dim3(256), flds, stream, p, npad, halo, PL);
        } else if (form == 1) hipLaunchKernelGGL((wino2p_finish_kernel<true, false>), fgrid, dim3(256), flds, stream, p, npad, halo, PL);
        else hipLaunchKernelGGL((wino2p_finish_kernel<false, false>), fgrid, dim3(256), flds, stream, p, npad, halo, PL);
        if (launches) *launches = 2;
        return hipGetLastError();
    }
    if (p.y_ds != p.y_hs * p.y_hs || p.y_cs != (p.Dout + 2 * halo) * p.y_ds) return hipErrorInvalidValue;
    if (form == 1) {
        // (only this form's finish kernel stages output slices in LDS: the class-parallel form's flat kernel has no edge limit)
        const int M = wino2_outputs(ax), G = p.Nh * p.Nw;
        if (ax != 0 || (size_t)M * p.y_ds * 4 > 64 * 1024) return hipErrorInvalidValue;
        int SUB = 256 / G < 1 ? 1 : 256 / G;               // (sample, depth group) pairs per finish workgroup
        if (SUB > p.B * p.Nd) SUB = p.B * p.Nd;
        while (SUB > 1 && (size_t)SUB * M * p.y_ds * 4 > 64 * 1024) --SUB;
        const size_t flds = (size_t)SUB * M * p.y_ds * 4;
        const FastDiv dNd((unsigned)p.Nd), dMS((unsigned)(M * p.y_ds));
        const long long items = (long long)p.Cout * ((p.B * p.Nd + SUB - 1) / SUB);
        const dim3 fgrid((unsigned)(items < 32768 ? items : 32768));
        const dim3 grid(p.m_tiles * n_tiles * 6);
        if (p.Nw % 4 == 0) hipLaunchKernelGGL((wino_kernel<4, 1, 2, false, false, false, true>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((wino_kernel<1, 1, 2, false, false, false, true>), grid, dim3(256), lds, stream, p);
        hipError_t e = hipGetLastError();
        if (e != hipSuccess) return e;
        AuxScope aux(stream, 4.0 * (double)p.Cout * (24.0 * n_tiles * WCN + (double)p.B * p.y_cs));      // (slabs in, padded output out)
        hipLaunchKernelGGL(wino2s_finish_kernel, fgrid, dim3(256), flds, stream, p, n_tiles * WCN, halo, SUB, dNd, dMS);
        if (launches) *launches = 2;
        return hipGetLastError();
    }
    const dim3 grid(p.m_tiles * n_tiles * p.ncls);
    if (p.Nw % 4 == 0) hipLaunchKernelGGL((wino_kernel<4, 1, 2, true, false>), grid, dim3(256), lds, stream, p);
    else hipLaunchKernelGGL((wino_kernel<1, 1, 2, true, false>), grid, dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return e;
    const long long total = (long long)p.Cout * p.Ntotal;
    const long long blocks = (total + 255) / 256;
    const dim3 flat((unsigned)(blocks < 8192 ? blocks : 8192));
    AuxScope aux(stream, 4.0 * (double)p.Cout * ((double)p.ncls * n_tiles * WCN + (double)p.B * p.Dout * p.Hout * p.Nw));
    if (ax == 0) hipLaunchKernelGGL(wino2_finish_flat_kernel<0>, flat, dim3(256), 0, stream, p, n_tiles * WCN);
    else hipLaunchKernelGGL(wino2_finish_flat_kernel<1>, flat, dim3(256), 0, stream, p, n_tiles * WCN);
    if (launches) *launches = 2;
    return hipGetLastError();
}

#ifdef S3R_ABLATE
}  // namespace s3r
extern "C" int s3r_debug_read_timeline_wino(unsigned long long* out, int nblocks) {
    if (nblocks > 65536) nblocks = 65536;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(s3r::s3r_wino_timeline), sizeof(unsigned long long) * 6 * (size_t)nblocks) == hipSuccess ? nblocks : -1;
}
extern "C" int s3r_debug_clear_timeline_wino() {
    void* ptr = nullptr;
    if (hipGetSymbolAddress(&ptr, HIP_SYMBOL(s3r::s3r_wino_timeline)) != hipSuccess) return -1;
    return hipMemset(ptr, 0, sizeof(unsigned long long) * 6 * 65536) == hipSuccess ? 0 : -1;
}
namespace s3r {
#endif

}  // namespace s3r
