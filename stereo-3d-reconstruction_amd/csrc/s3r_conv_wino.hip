// fp32 3 x 3 [x 3] stride-1 convolutions with FEWER multiplications: Winograd F(2, 3) along the H axis only.
//
// Two output rows (2q, 2q + 1) of a column need input rows r0..r3 = 2q .. 2q + 3 of the padded input and the three
// kernel rows g0, g1, g2:
//     v0 = r0 - r2      u0 = g0                 m_i = sum over (cin, kd, kw) of u_i * v_i        (4 products, not 6)
//     v1 = r1 + r2      u1 = (g0 + g1 + g2) / 2
//     v2 = r2 - r1      u2 = (g0 - g1 + g2) / 2      y(2q)     = (m0 + m1) + m2
//     v3 = r1 - r3      u3 = g2                      y(2q + 1) = (m1 - m2) - m3
// so the layer becomes FOUR convolutions with a (kd x 1 x kw) kernel — 9 taps instead of 27 in 3D, 3 instead of 9 in 2D,
// each over its own transformed input plane set V_i and its own transformed weights U_i — whose results are combined in
// registers: 2/3 of the matrix work of the direct form for the same outputs.  Why only one axis: every further axis
// doubles the transform-domain accumulators per output again (4 per 2 outputs here; 64 per 8 for F(2,3)^3 — a whole CU's
// register file for a 32 x 32 tile) and shortens each GEMM's K to Cin; along H alone K stays Cin x 9 (or x 3), the
// W axis stays contiguous (16-byte gathers where W % 4 == 0, whole-row stores), and the kernel below is the implicit
// GEMM of s3r_conv_glds.hip with a class loop around its K loop.  fp32 F(2, 3) is as accurate as the direct fp32 sum
// here (3-5e-7 relative to fp64 over 64-256 input channels; north_star allows 1e-4), but it is a DIFFERENT summation:
// results are not bit-identical to the direct kernels' (the library's policy and the S3R_WINO switch: s3r_api.hip).
//
//   wino_input_kernel   x (padded NC(D)HW, halo 1) -> V[4][B][C][Dp][H/2][Wp]      (HBM-bound: reads x once, writes 2 x)
//   pack_wino_kernel    w[Cout][Cin][kd][3][kw]    -> Up[4][(chunk*T' + tap')*32 + c][CoutPad],  T' = kd*kw, 32-channel chunks
//   conv_wino_kernel    64 couts x 128 positions per workgroup (positions = (b, d, row pair q, w)), 4 waves of 64 x 32,
//                       4 classes x 2 MFMA tiles of accumulators per wave, K tiles of one tap x 32 channels in a 2-stage LDS
//                       ring (the next tile's DMAs in flight under the current tile's 32 MFMAs per wave)
#include "s3r_kernels.h"
#include <cstdlib>

namespace s3r {

typedef float wf32x16 __attribute__((ext_vector_type(16)));
typedef float wv2f __attribute__((ext_vector_type(2)));

#define S3R_LDS_PTR_W(p) ((__attribute__((address_space(3))) void*)(p))

#ifndef S3R_WNB
#define S3R_WNB 2      // LDS stages.  With 16-channel K tiles: 2 stages -0.3 %, 3 = 4.  32-channel tiles halve the barriers per MFMA
#endif                  // (32 MFMAs per wave between two, like the direct path's 64 x 256 tile) and, at two stages (48 KiB, three
                        // workgroups per CU), are 1.3 % of the step faster than 16 x 3: v1 0.807 -> 0.769 ms, d3 0.824 -> 0.803
#ifndef S3R_WBK
#define S3R_WBK 32
#endif
constexpr int WBM = 64, WBN = 128, WBK = S3R_WBK, WNB = S3R_WNB;      // K tile: one tap x WBK channels
constexpr int WNPA = WBK * WBM / 1024;                                // 1 KiB weight pieces per wave and K tile
int wino_bk() { return WBK; }

template <int BYTES>
__device__ __forceinline__ void wdma(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voffset, int soffset) {
    static_assert(BYTES == 16 || BYTES == 4, "LDS-DMA width");
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR_W(lds_dst), 16, voffset, soffset, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR_W(lds_dst), 4, voffset, soffset, 0, 0);
}

// ---- F(R, 3) along H, R = 2 or 4 outputs per group from R + 2 padded input rows through R + 2 products.
//   R = 2 (above).   R = 4 (Lavin & Gray's F(4, 3); HALF the direct form's multiplications; in fp32 over 64-128 channels 4.9e-7
//   relative to an fp64 convolution against the direct sum's 3.6e-7):
//     v0 = 4 r0 - 5 r2 + r4             u0 = g0 / 4                          y0 = m0 + m1 + m2 + m3 + m4
//     v1 = -4 r1 - 4 r2 + r3 + r4       u1 = -(g0 + g1 + g2) / 6             y1 = m1 - m2 + 2 m3 - 2 m4
//     v2 = 4 r1 - 4 r2 - r3 + r4        u2 = -(g0 - g1 + g2) / 6             y2 = m1 + m2 + 4 m3 + 4 m4
//     v3 = -2 r1 - r2 + 2 r3 + r4       u3 = g0 / 24 + g1 / 12 + g2 / 6      y3 = m1 - m2 + 8 m3 - 8 m4 + m5
//     v4 = 2 r1 - r2 - 2 r3 + r4        u4 = g0 / 24 - g1 / 12 + g2 / 6
//     v5 = 4 r1 - 5 r3 + r5             u5 = g2
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
    if (R == 4) hipLaunchKernelGGL(wino_input_kernel<4>, grid, dim3(256), 0, s, x, V, planes, Hp, Wp, Hq, total);
    else hipLaunchKernelGGL(wino_input_kernel<2>, grid, dim3(256), 0, s, x, V, planes, Hp, Wp, Hq, total);
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
            if (R == 2) {
                v = cls == 0 ? g0 : cls == 1 ? ((g0 + g1) + g2) * 0.5f : cls == 2 ? ((g0 - g1) + g2) * 0.5f : g2;
            } else {
                const float s02 = g0 + g2, a = g0 * (1.f / 24.f) + g2 * (1.f / 6.f), b12 = g1 * (1.f / 12.f);
                v = cls == 0 ? g0 * 0.25f : cls == 1 ? (s02 + g1) * (-1.f / 6.f) : cls == 2 ? (s02 - g1) * (-1.f / 6.f)
                  : cls == 3 ? a + b12 : cls == 4 ? a - b12 : g2;
            }
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_wino(const float* w, float* wp, int Cin, int Cout, int CoutPad, int kd, int kw, int R, hipStream_t s) {
    hipLaunchKernelGGL(pack_wino_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad, kd, kw, R);
    return hipGetLastError();
}

// ---- the four class convolutions and their combination.  p describes the CLASS convolution: p.x = V, x_cs / x_ds / x_hs
// its strides (x_hs = one V row per row pair), p.x_cls the class stride, Nh = row pairs per plane, kh = 1, T = kd * kw,
// x_org = 0; p.y the layer's padded output, p.Hout its true height (odd: the last pair's second row is not stored).
template <int VEC, int R>
__global__ __launch_bounds__(256, 2) void conv_wino_kernel(const ConvParams p) {
    constexpr int NCLS = R + 2;
    extern __shared__ __attribute__((aligned(16))) float wsmem[];
    float* As = wsmem;                                   // [WNB][WBK][64]
    float* Bs = wsmem + WNB * WBK * WBM;                 // [WNB][WBK][128]
    constexpr int PB = 64 * VEC;                         // floats per B piece
    constexpr int NPIECE_B = WBK * WBN / PB;
    constexpr int NPB = NPIECE_B / 4;
    constexpr bool B_WIDE = WBN >= PB;
    constexpr int PPR = B_WIDE ? WBN / PB : 1;
    constexpr int RPP = B_WIDE ? 1 : PB / WBN;
    constexpr int LPR_B = WBN / VEC;
    constexpr int NPD = WNPA + NPB;                      // DMAs per wave per K tile (WNPA 1 KiB weight pieces + NPB gathers)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    int bid = blockIdx.x;
    {   // XCD-aware tile order (as conv_glds_kernel)
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m_tile = bid % p.m_tiles, n_tile = bid / p.m_tiles;
    const int m0 = m_tile * WBM, n0 = n_tile * WBN;
    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.T;
    const int chunks = p.Cin / WBK;
    const int nkt = T * chunks;                          // K tiles per class
    const int total = NCLS * nkt;

    int bvoff;
    {
        int col, lrow;
        if (B_WIDE) { col = (wave % PPR) * PB + lane * VEC; lrow = 0; }
        else        { col = (lane % LPR_B) * VEC;           lrow = lane / LPR_B; }
        int n = n0 + col;
        if (n >= p.Ntotal) n = p.Ntotal - VEC;
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        bvoff = (b * p.Cin * p.x_cs + p.x_org + pd * p.x_ds + q * p.x_hs + pw + lrow * p.x_cs) * 4;
    }
    const int avoff = ((lane >> 4) * p.CoutPad + (lane & 15) * 4) * 4;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)((unsigned)NCLS * (unsigned)T * (unsigned)p.Cin * (unsigned)p.CoutPad * 4u), 0x00020000);
    const int b_row0 = B_WIDE ? wave / PPR : wave * RPP;
    constexpr int B_ROW_STEP = B_WIDE ? 4 / PPR : 4 * RPP;
    const int b_lds0 = B_WIDE ? b_row0 * WBN + (wave % PPR) * PB : wave * PB;
    constexpr int B_LDS_STEP = B_WIDE ? B_ROW_STEP * WBN : 4 * PB;
    const int cs4 = p.x_cs * 4;

    int c_cls = 0, c_cc = 0, c_td = 0, c_tw = 0, c_tap = 0, c_kt = 0;      // cursor of the NEXT K tile to fetch (scalar)
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < WNPA; ++q)
            wdma<16>(wrsrc, As + buf * WBK * WBM + (wave + 4 * q) * 256, avoff,
                     (c_kt * WBK * p.CoutPad + m0) * 4 + (wave + 4 * q) * 4 * p.CoutPad * 4);
        float* sb = Bs + buf * WBK * WBN + b_lds0;
        const int b_base = (c_cls * p.x_cls + (c_cc * WBK + b_row0) * p.x_cs + c_td * p.x_ds + c_tw) * 4;
#pragma unroll
        for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(xrsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
        ++c_kt;
        if (++c_tw == p.kw) { c_tw = 0; ++c_td; }
        if (++c_tap == T) {
            c_tap = 0; c_td = 0; c_tw = 0;
            if (++c_cc == chunks) { c_cc = 0; ++c_cls; }
        }
    };

    wf32x16 acc[NCLS][2];
#pragma unroll
    for (int c = 0; c < NCLS; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
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

    const int a_off = h * WBM + j * 2;
    const int b_off = h * WBN + wave * 32 + j;
    int cur = 0, g = 0;
    auto run_class = [&](wf32x16 (&ac)[2]) {
        for (int kt = 0; kt < nkt; ++kt, ++g) {
            const bool more = g + WNB - 1 < total;
            if (more) issue(cur == 0 ? WNB - 1 : cur - 1);        // into the stage tile g - 1 was read from
            const float* a = As + cur * WBK * WBM + a_off;
            const float* b = Bs + cur * WBK * WBN + b_off;
#pragma unroll
            for (int ks = 0; ks < WBK / 2; ++ks) {
                const wv2f av = *reinterpret_cast<const wv2f*>(a + ks * 2 * WBM);
                const float bv = b[ks * 2 * WBN];
                ac[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv, ac[0], 0, 0, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv, ac[1], 0, 0, 0);
            }
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WNB - 2) * NPD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur = cur + 1 == WNB ? 0 : cur + 1;
        }
    };
#pragma unroll
    for (int c = 0; c < NCLS; ++c) run_class(acc[c]);

    // ---- epilogue: per-cout constants through LDS (every wave is past the last barrier: the ring is idle)
    float* ep_sc = wsmem;
    float* ep_sf = wsmem + WBM;
    if (tid < WBM) {
        const int m = m0 + tid;
        ep_sc[tid] = (p.scale && m < p.Cout) ? p.scale[m] : 1.f;
        ep_sf[tid] = (p.shift && m < p.Cout) ? p.shift[m] : 0.f;
    }
    __syncthreads();
    const int n = n0 + wave * 32 + j;
    const bool ok = n < p.Ntotal;
    int e0, row1;                                        // first output element of the group, rows that exist
    {
        const int nn = ok ? n : 0;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        e0 = b * p.y_bs + p.y_org + pd * p.y_ds + R * q * p.y_hs + pw;
        row1 = p.Hout - R * q;                           // rows of this group that exist (>= R: all)
    }
    const int mbase = 8 * h;                             // rows of this lane: mbase + ((r & 3) + 8 (r >> 2)) * 2 + tm
    const int mlimit = p.Cout - (m0 + mbase);
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    const int yvo0 = (e0 + (m0 + mbase) * p.y_cs) * 4;
    const int yrow = p.y_hs * 4;
    const int row_bytes = p.y_cs * 4;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
#pragma unroll
    for (int tm = 0; tm < 2; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int dm = ((r & 3) + 8 * (r >> 2)) * 2 + tm;
            if (dm >= mlimit) continue;
            const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm];
            float y[R];
            if constexpr (R == 2) {
                y[0] = (acc[0][tm][r] + acc[1][tm][r]) + acc[2][tm][r];
                y[1] = (acc[1][tm][r] - acc[2][tm][r]) - acc[3][tm][r];
            } else {
                const float s12 = acc[1][tm][r] + acc[2][tm][r], d12 = acc[1][tm][r] - acc[2][tm][r];
                const float s34 = acc[3][tm][r] + acc[4][tm][r], d34 = acc[3][tm][r] - acc[4][tm][r];
                y[0] = (acc[0][tm][r] + s12) + s34;
                y[1] = fmaf(2.f, d34, d12);
                y[2] = fmaf(4.f, s34, s12);
                y[3] = fmaf(8.f, d34, d12) + acc[5][tm][r];
            }
            const int so = dm * row_bytes;
            if (ok) {
#pragma unroll
                for (int i = 0; i < R; ++i)
                    if (i < row1) {
                        const float v = fmaxf(fmaf(y[i], sc, sf), lo);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yrsrc, yvo0 + i * yrow, so, 0);
                    }
            }
            if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
        }
}

// p: see the kernel; p.kh = the outputs per group R (2 or 4) on entry (the class convolution's own kh is 1)
hipError_t launch_conv_wino(ConvParams p, hipStream_t stream) {
    const int R = p.kh;
    if (p.Cin % WBK != 0 || (R != 2 && R != 4) || p.stride != 1 || p.transposed || p.ksplit != 1 || p.head_w || p.act == ACT_SIGMOID)
        return hipErrorInvalidValue;
    p.kh = 1;
    p.m_tiles = (p.Cout + WBM - 1) / WBM;
    p.n_tiles = (p.Ntotal + WBN - 1) / WBN;
    const size_t lds = (size_t)WNB * WBK * (WBM + WBN) * sizeof(float);
    const dim3 grid(p.m_tiles * p.n_tiles);
    const bool v4 = p.Nw % 4 == 0;
    if (R == 4) {
        if (v4) hipLaunchKernelGGL((conv_wino_kernel<4, 4>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((conv_wino_kernel<1, 4>), grid, dim3(256), lds, stream, p);
    } else {
        if (v4) hipLaunchKernelGGL((conv_wino_kernel<4, 2>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((conv_wino_kernel<1, 2>), grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

// ================================================================================================
// ConvTranspose3d(k = 4, s = 2, p = 1) with fewer multiplications: Winograd F(2, 2) along H inside every output-parity
// class.  A class (rd, rh, rw) is a 2 x 2 x 2-tap convolution over the input grid (s3r_conv_glds.hip); two of its outputs
// that are neighbours along H — input rows ph = 2q and 2q + 1, i.e. output rows 4q + rh and 4q + 2 + rh — read the
// padded input rows x0, x1, x2 = R, R + 1, R + 2 (R = 2q + rh) through the two H taps g0, g1:
//     y(2q)     = x0 g0 + x1 g1 = m1 + m2        m1 = (x0 - x1) g0
//     y(2q + 1) = x1 g0 + x2 g1 = m2 - m3        m2 = x1 (g0 + g1)        m3 = (x1 - x2) g1
// — three products instead of four: 3/4 of the matrix work.  The transformed input is the padded input itself plus ONE
// tensor of row differences D[r] = x[r] - x[r + 1] (wino_rowdiff_kernel: the input of a transposed convolution is small),
// the transformed weights are g0, g0 + g1, g1 per (class, depth tap, column tap).  The kernel is conv_wino_kernel's shape:
// positions (b, pd, q, pw) over the input grid, 64 couts x 128 positions per workgroup, three F-classes x 2 MFMA tiles of
// accumulators per wave, the eight parity classes of a tile back to back on one XCD; HEAD = the fused 1 x 1 x 1 head (d3 -> d4).
__global__ __launch_bounds__(256) void wino_rowdiff_kernel(const float* __restrict__ x, float* __restrict__ D, long long rows,
                                                           int Hp, int Wp) {
    const long long total = rows * Wp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long row = i / Wp;
        const int r = (int)(row % Hp);
        D[i] = (r + 1 < Hp) ? x[i] - x[i + Wp] : 0.f;
    }
}

hipError_t launch_wino_rowdiff(const float* x, float* D, long long planes, int Hp, int Wp, hipStream_t s) {
    const long long total = planes * Hp * Wp;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(wino_rowdiff_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, x, D, planes * Hp, Hp, Wp);
    return hipGetLastError();
}

// w[Cin][Cout][4][4][4] -> Up[pc = 8][f = 3][(chunk*4 + (td*2 + tw))*32 + c][CoutPad]
__global__ void pack_wino_deconv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad) {
    const size_t per_f = (size_t)4 * Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < 24 * per_f; i += (size_t)gridDim.x * blockDim.x) {
        const int pf = (int)(i / per_f);
        const int pc = pf / 3, f = pf - pc * 3;
        size_t r = i % per_f;
        const int co = (int)(r % CoutPad);
        r /= CoutPad;
        const int c = (int)(r % WBK);
        r /= WBK;
        const int tap = (int)(r & 3);
        const int cc = (int)(r >> 2);
        const int cin = cc * WBK + c;
        float v = 0.f;
        if (co < Cout) {
            const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
            const int td = tap >> 1, tw = tap & 1;
            const int kd = 3 - rd - 2 * td, kw = 3 - rw - 2 * tw;
            const float* g = w + ((size_t)cin * Cout + co) * 64 + kd * 16 + kw;
            const float g0 = g[(3 - rh) * 4], g1 = g[(1 - rh) * 4];          // H taps th = 0, 1: kernel rows 3 - rh, 1 - rh
            v = f == 0 ? g0 : f == 1 ? g0 + g1 : g1;
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_wino_deconv(const float* w, float* wp, int Cin, int Cout, int CoutPad, hipStream_t s) {
    hipLaunchKernelGGL(pack_wino_deconv_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    return hipGetLastError();
}

// p: the layer's transposed-convolution parameters (make_params) with Nh = row pairs (n / 2), p.x = the padded input,
// p.part = its row differences (same shape and strides), p.w = the 24 (class, F) slabs.
template <int VEC, bool HEAD>
__global__ __launch_bounds__(256, 2) void deconv_wino_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float wsmem[];
    float* As = wsmem;
    float* Bs = wsmem + WNB * WBK * WBM;
    constexpr int PB = 64 * VEC;
    constexpr int NPIECE_B = WBK * WBN / PB;
    constexpr int NPB = NPIECE_B / 4;
    constexpr bool B_WIDE = WBN >= PB;
    constexpr int PPR = B_WIDE ? WBN / PB : 1;
    constexpr int RPP = B_WIDE ? 1 : PB / WBN;
    constexpr int LPR_B = WBN / VEC;
    constexpr int NPD = WNPA + NPB;

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int j = lane & 31, h = lane >> 5;

    int bid, pc;
    {   // an XCD walks its run of tiles with the 8 parity classes of a tile back to back (they read the same input tile)
        const int nwg = gridDim.x >> 3;
        const int item = ((int)blockIdx.x & 7) * nwg + ((int)blockIdx.x >> 3);
        bid = item >> 3;
        pc = item & 7;
    }
    const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
    const int m_tile = bid % p.m_tiles, n_tile = bid / p.m_tiles;
    const int m0 = m_tile * WBM, n0 = n_tile * WBN;
    const int S = p.Nd * p.Nh * p.Nw;
    const int chunks = p.Cin / WBK;
    const int nkt = 4 * chunks;                          // K tiles per F-class: (depth tap, column tap) x chunks
    const int total = 3 * nkt;

    int bvoff;
    {
        int col, lrow;
        if (B_WIDE) { col = (wave % PPR) * PB + lane * VEC; lrow = 0; }
        else        { col = (lane % LPR_B) * VEC;           lrow = lane / LPR_B; }
        int n = n0 + col;
        if (n >= p.Ntotal) n = p.Ntotal - VEC;
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        // padded indices: depth pd + rd + td, row 2q + rh (+ 1), column pw + rw + tw
        bvoff = (b * p.Cin * p.x_cs + (pd + rd) * p.x_ds + (2 * q + rh) * p.x_hs + pw + rw + lrow * p.x_cs) * 4;
    }
    const int avoff = ((lane >> 4) * p.CoutPad + (lane & 15) * 4) * 4;
    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t drsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.part), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)(24u * 4u * (unsigned)p.Cin * (unsigned)p.CoutPad * 4u), 0x00020000);
    const int b_row0 = B_WIDE ? wave / PPR : wave * RPP;
    constexpr int B_ROW_STEP = B_WIDE ? 4 / PPR : 4 * RPP;
    const int b_lds0 = B_WIDE ? b_row0 * WBN + (wave % PPR) * PB : wave * PB;
    constexpr int B_LDS_STEP = B_WIDE ? B_ROW_STEP * WBN : 4 * PB;
    const int cs4 = p.x_cs * 4;

    int c_f = 0, c_cc = 0, c_tap = 0, c_kt = pc * total;               // cursor of the NEXT K tile to fetch (scalar)
    auto issue = [&](int buf) {
#pragma unroll
        for (int q = 0; q < WNPA; ++q)
            wdma<16>(wrsrc, As + buf * WBK * WBM + (wave + 4 * q) * 256, avoff,
                     (c_kt * WBK * p.CoutPad + m0) * 4 + (wave + 4 * q) * 4 * p.CoutPad * 4);
        float* sb = Bs + buf * WBK * WBN + b_lds0;
        // F-class 0: D at row R, 1: X at row R + 1, 2: D at row R + 1
        const int b_base = ((c_cc * WBK + b_row0) * p.x_cs + (c_tap >> 1) * p.x_ds + (c_f ? p.x_hs : 0) + (c_tap & 1)) * 4;
        if (c_f == 1) {
#pragma unroll
            for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(xrsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
        } else {
#pragma unroll
            for (int q = 0; q < NPB; ++q) wdma<4 * VEC>(drsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
        }
        ++c_kt;
        if (++c_tap == 4) {
            c_tap = 0;
            if (++c_cc == chunks) { c_cc = 0; ++c_f; }
        }
    };

    wf32x16 acc[3][2];
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int t = 0; t < 2; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[c][t][r] = 0.f;

#pragma unroll
    for (int i = 0; i < WNB - 1; ++i)
        if (i < total) issue(i);
    if (total >= WNB - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WNB - 2) * NPD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");

    const int a_off = h * WBM + j * 2;
    const int b_off = h * WBN + wave * 32 + j;
    int cur = 0, g = 0;
    auto run_class = [&](wf32x16 (&ac)[2]) {
        for (int kt = 0; kt < nkt; ++kt, ++g) {
            const bool more = g + WNB - 1 < total;
            if (more) issue(cur == 0 ? WNB - 1 : cur - 1);
            const float* a = As + cur * WBK * WBM + a_off;
            const float* b = Bs + cur * WBK * WBN + b_off;
#pragma unroll
            for (int ks = 0; ks < WBK / 2; ++ks) {
                const wv2f av = *reinterpret_cast<const wv2f*>(a + ks * 2 * WBM);
                const float bv = b[ks * 2 * WBN];
                ac[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[0], bv, ac[0], 0, 0, 0);
                ac[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[1], bv, ac[1], 0, 0, 0);
            }
            if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((WNB - 2) * NPD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur = cur + 1 == WNB ? 0 : cur + 1;
        }
    };
    run_class(acc[0]);
    run_class(acc[1]);
    run_class(acc[2]);

    float* ep_sc = wsmem;
    float* ep_sf = wsmem + WBM;
    float* ep_hw = wsmem + 2 * WBM;
    if (tid < WBM) {
        const int m = m0 + tid;
        ep_sc[tid] = (p.scale && m < p.Cout) ? p.scale[m] : 1.f;
        ep_sf[tid] = (p.shift && m < p.Cout) ? p.shift[m] : 0.f;
        ep_hw[tid] = (HEAD && p.head_w && m < p.Cout) ? p.head_w[m] : 0.f;
    }
    __syncthreads();
    const int n = n0 + wave * 32 + j;
    const bool ok = n < p.Ntotal;
    int e0;
    {
        const int nn = ok ? n : 0;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int q = p.dW.div(rem);
        const int pw = rem - q * p.Nw;
        // output (2 pd + rd, 2 ph + rh, 2 pw + rw), ph = 2q (row 0) and 2q + 1 (row 1: 2 y_hs further)
        e0 = b * p.y_bs + p.y_org + (pd * p.y_ds + 2 * q * p.y_hs + pw) * 2 + rd * p.y_ds + rh * p.y_hs + rw;
    }
    const int mbase = 8 * h;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    if constexpr (HEAD) {
        // the workgroup's 64 couts are the whole channel axis: 32 of them in this lane, the other 32 in lane j + 32
        float t0 = 0.f, t1 = 0.f;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = ((r & 3) + 8 * (r >> 2)) * 2 + tm;
                const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm], hw = ep_hw[mbase + dm];
                const float y0 = acc[0][tm][r] + acc[1][tm][r];
                const float y1 = acc[1][tm][r] - acc[2][tm][r];
                t0 = fmaf(fmaxf(fmaf(y0, sc, sf), lo), hw, t0);
                t1 = fmaf(fmaxf(fmaf(y1, sc, sf), lo), hw, t1);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        t0 += __shfl_xor(t0, 32, 64);
        t1 += __shfl_xor(t1, 32, 64);
        const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
        t0 = fmaf(t0, hsc, hsf);
        t1 = fmaf(t1, hsc, hsf);
        if (p.head_act == ACT_RELU) { t0 = fmaxf(t0, 0.f); t1 = fmaxf(t1, 0.f); }
        else if (p.head_act == ACT_SIGMOID) { t0 = __builtin_amdgcn_rcpf(1.f + __expf(-t0)); t1 = __builtin_amdgcn_rcpf(1.f + __expf(-t1)); }
        if (h == 0 && ok) {
            p.y[e0] = t0;
            p.y[e0 + 2 * p.y_hs] = t1;
        }
        return;
    } else {
        const int mlimit = p.Cout - (m0 + mbase);
        const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
        const int yvo0 = (e0 + (m0 + mbase) * p.y_cs) * 4;
        const int yvo1 = yvo0 + 2 * p.y_hs * 4;
        const int row_bytes = p.y_cs * 4;
#pragma unroll
        for (int tm = 0; tm < 2; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = ((r & 3) + 8 * (r >> 2)) * 2 + tm;
                if (dm >= mlimit) continue;
                const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm];
                const float y0 = acc[0][tm][r] + acc[1][tm][r];
                const float y1 = acc[1][tm][r] - acc[2][tm][r];
                const float v0 = fmaxf(fmaf(y0, sc, sf), lo), v1 = fmaxf(fmaf(y1, sc, sf), lo);
                const int so = dm * row_bytes;
                if (ok) {
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v0), yrsrc, yvo0, so, 0);
                    __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v1), yrsrc, yvo1, so, 0);
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
    }
}

hipError_t launch_deconv_wino(ConvParams p, hipStream_t stream) {
    if (p.Cin % WBK != 0 || !p.transposed || p.ksplit != 1 || !p.part || p.act == ACT_SIGMOID || (p.head_w && p.Cout > WBM))
        return hipErrorInvalidValue;
    p.m_tiles = (p.Cout + WBM - 1) / WBM;
    p.n_tiles = (p.Ntotal + WBN - 1) / WBN;
    const size_t lds = (size_t)WNB * WBK * (WBM + WBN) * sizeof(float);
    const dim3 grid(p.m_tiles * p.n_tiles * 8);
    const bool v4 = p.Nw % 4 == 0;
    if (p.head_w) {
        if (v4) hipLaunchKernelGGL((deconv_wino_kernel<4, true>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((deconv_wino_kernel<1, true>), grid, dim3(256), lds, stream, p);
    } else {
        if (v4) hipLaunchKernelGGL((deconv_wino_kernel<4, false>), grid, dim3(256), lds, stream, p);
        else hipLaunchKernelGGL((deconv_wino_kernel<1, false>), grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

}  // namespace s3r
