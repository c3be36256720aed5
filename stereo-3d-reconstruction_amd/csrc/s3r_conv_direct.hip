// Barrier-free direct convolution on the fp32 matrix cores: every wave owns a (32*TM) x (32*TN)
// output tile and loads its MFMA operand fragments STRAIGHT from global memory (L1/L2) into
// registers — no LDS, no workgroup barrier, no coupling between the waves that share a SIMD.
//
// Why: with LDS staging, the four waves of a workgroup sit on four different SIMDs and meet at a
// barrier every K tile; each SIMD arbitrates its matrix pipe among waves of three workgroups
// independently, so a wave that loses arbitration on one SIMD stalls its three siblings (and the pipe
// time they would have used) on the others.  rocprofv3 showed the pipe 71 % busy with waves 70 % of
// their time waiting to issue.  Here a wave only ever waits for its own loads, which are issued one
// full K tile (16 channels of one tap = 8 k-steps, >= 2048 matrix-pipe cycles) ahead.
//
// v_mfma_f32_32x32x2_f32 needs ONE dword per lane per operand per k-step:
//   A: lane (i = l&31, h = l>>5) holds W[cout = m0+i][k = 2*ks + h]
//   B: lane (j = l&31, h)        holds X[k = 2*ks + h][pos = n0+j]
// so the B fragment IS a coalesced gather (32 consecutive positions of channel c0+2ks+h: two 128-byte
// segments per wave-load), addressed through a buffer descriptor with the channel step in an SGPR.
// Weights are pre-packed so a lane's 8 k-steps of one K tile are 32 contiguous bytes:
//   wd[cls][cc][tap][h][cout][ks]      (cc = 16-channel chunk, k = 2*ks + h inside the chunk)
// i.e. two dwordx4 loads per lane per 32x(16) weight tile, 1 KiB contiguous per half-wave.
// K order is chunk-major, tap-minor: the 9/27/8 taps of one 16-channel chunk re-read the same input
// patch back to back, so the gather mostly hits the CU's L1.
//
// Invalid taps (zero padding / outside the transposed-conv support) load through a voffset of 2^31,
// which the raw-buffer range check rejects (returns 0) — no branch in the K loop.
#include "s3r_kernels.h"

namespace s3r {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256) void conv_direct_kernel(const ConvParams p) {
    constexpr int BM = 32 * WM * TM;
    constexpr int BN = 32 * WN * TN;
    static_assert(WM * WN == 4, "4 waves per workgroup");

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m_tile = bid % p.m_tiles;
    const int n_tile = bid / p.m_tiles;
    const int m0 = m_tile * BM + wm * TM * 32;     // this wave's first cout
    const int n0 = n_tile * BN + wn * TN * 32;     // this wave's first position
    const int cls = blockIdx.y;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;

    const int HWi = p.Hi * p.Wi;
    const int DHWi = p.Di * HWi;
    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.kd * p.kh * p.kw;
    const int nchunk = p.Cin >> 4;
    const int nkt = T * nchunk;

    // ---- per-lane gather state, one per N sub-tile ----
    int base[TN];           // byte offset of (b, cin = h, id0, ih0, iw0); garbage when no tap is valid
    unsigned mask[TN];      // bits 0-3 d taps, 4-7 h taps, 8-11 w taps
    const int sd = p.transposed ? (rd ? 1 : -1) : 1;
    const int sh = p.transposed ? (rh ? 1 : -1) : 1;
    const int sw = p.transposed ? (rw ? 1 : -1) : 1;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + tn * 32 + j;
        const bool nvalid = n < p.Ntotal;
        const int nn = nvalid ? n : 0;
        const int b = nn / S;
        int rem = nn - b * S;
        const int pd = rem / (p.Nh * p.Nw);
        rem -= pd * p.Nh * p.Nw;
        const int ph = rem / p.Nw;
        const int pw = rem - ph * p.Nw;
        int id0, ih0, iw0;
        if (!p.transposed) {
            id0 = pd * p.stride - p.pad_d; ih0 = ph * p.stride - p.pad_h; iw0 = pw * p.stride - p.pad_w;
        } else {
            id0 = pd; ih0 = ph; iw0 = pw;
        }
        unsigned m = 0;
        for (int t = 0; t < p.kd; ++t) m |= ((unsigned)(id0 + sd * t) < (unsigned)p.Di) ? (1u << t) : 0u;
        for (int t = 0; t < p.kh; ++t) m |= ((unsigned)(ih0 + sh * t) < (unsigned)p.Hi) ? (16u << t) : 0u;
        for (int t = 0; t < p.kw; ++t) m |= ((unsigned)(iw0 + sw * t) < (unsigned)p.Wi) ? (256u << t) : 0u;
        mask[tn] = nvalid ? m : 0u;
        base[tn] = ((b * p.Cin + h) * DHWi + id0 * HWi + ih0 * p.Wi + iw0) * 4;
    }

    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((unsigned)p.B * (unsigned)p.Cin * (unsigned)DHWi * 4u), 0x00020000);
    const int step2 = 2 * DHWi * 4;        // bytes between k-steps (two channels)

    // weights: lane's 8 floats of K tile (cc, tap) for sub-tile tm start at
    //   wd + ((((cc*T + tap)*2 + h) * CoutPad) + m0 + tm*32 + i) * 8
    const float* __restrict__ wlane =
        p.w + (size_t)cls * T * p.Cin * p.CoutPad + ((size_t)h * p.CoutPad + m0 + j) * 8;
    const size_t wtile = (size_t)2 * p.CoutPad * 8;      // floats per (cc, tap) K tile

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    // K-tile cursor of the next tile to load: chunk-major, tap-minor
    int c_cc = 0, c_td = 0, c_th = 0, c_tw = 0, c_tap = 0;

    float bq[2][TN][8];
    v4f aq[2][TM][2];

#define S3R_DLOAD(S)                                                                                       \
    {                                                                                                      \
        const int cce = c_cc < nchunk ? c_cc : nchunk - 1;   /* past the end: harmless re-load */          \
        const float* __restrict__ wp = wlane + (size_t)(cce * T + c_tap) * wtile;                          \
        _Pragma("unroll") for (int tm = 0; tm < TM; ++tm) {                                                \
            aq[S][tm][0] = *reinterpret_cast<const v4f*>(wp + tm * 256);                                   \
            aq[S][tm][1] = *reinterpret_cast<const v4f*>(wp + tm * 256 + 4);                               \
        }                                                                                                  \
        const int delta = ((sd * c_td * p.Hi + sh * c_th) * p.Wi + sw * c_tw) * 4;                         \
        const int soff = cce * 16 * DHWi * 4;                                                              \
        const unsigned tapbits = (1u << c_td) | (16u << c_th) | (256u << c_tw);                            \
        _Pragma("unroll") for (int tn = 0; tn < TN; ++tn) {                                                \
            const int vo = ((mask[tn] & tapbits) == tapbits) ? base[tn] + delta : (int)0x80000000;         \
            _Pragma("unroll") for (int ks = 0; ks < 8; ++ks) bq[S][tn][ks] = __builtin_bit_cast(           \
                float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, vo, soff + ks * step2, 0));             \
        }                                                                                                  \
        if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }                        \
        if (++c_tap == T) { c_tap = 0; c_td = 0; c_th = 0; c_tw = 0; ++c_cc; }                             \
    }
#define S3R_DCOMPUTE(S)                                                                                    \
    {                                                                                                      \
        _Pragma("unroll") for (int ks = 0; ks < 8; ++ks)                                                   \
            _Pragma("unroll") for (int tm = 0; tm < TM; ++tm)                                              \
                _Pragma("unroll") for (int tn = 0; tn < TN; ++tn)                                          \
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(aq[S][tm][ks >> 2][ks & 3],        \
                                                                       bq[S][tn][ks], acc[tm][tn], 0, 0, 0); \
    }

    // The steady-state loop has NO conditional loads or computes: hipcc merges s_waitcnt counts
    // conservatively across control-flow joins, and one skipped load collapses the counted vmcnt(N)
    // that keeps the next tile in flight into vmcnt(0).  An odd tile count is peeled up front and the
    // one load past the end re-reads the last tile (clamped cursor) instead of being skipped.
    if (nkt & 1) {
        S3R_DLOAD(0);
        S3R_DCOMPUTE(0);
    }
    const int pairs = nkt >> 1;
    if (pairs > 0) {
        S3R_DLOAD(0);
        for (int it = 0; it < pairs; ++it) {
            S3R_DLOAD(1);
            S3R_DCOMPUTE(0);
            S3R_DLOAD(0);
            S3R_DCOMPUTE(1);
        }
    }
#undef S3R_DLOAD
#undef S3R_DCOMPUTE

    // ---- epilogue (same map as the LDS kernel) ----
    const int So = p.Do * p.Ho * p.Wo;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + tn * 32 + j;
        if (n >= p.Ntotal) continue;
        const int b = n / S;
        int rem = n - b * S;
        int sp;
        if (!p.transposed) {
            sp = rem;
        } else {
            const int pd = rem / (p.Nh * p.Nw);
            rem -= pd * p.Nh * p.Nw;
            const int ph = rem / p.Nw;
            const int pw = rem - ph * p.Nw;
            sp = ((2 * pd + rd) * p.Ho + 2 * ph + rh) * p.Wo + 2 * pw + rw;
        }
        float* __restrict__ yb = p.y + (size_t)b * p.Cout * So + sp;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                if (m < p.Cout) {
                    float v = acc[tm][tn][r];
                    const float sc = p.scale ? p.scale[m] : 1.f;
                    const float sf = p.shift ? p.shift[m] : 0.f;
                    v = fmaf(v, sc, sf);
                    if (p.act == ACT_RELU) v = fmaxf(v, 0.f);
                    else if (p.act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                    yb[(size_t)m * So] = v;
                }
            }
        }
    }
}

// tile configurations of the direct kernel (same ids as the LDS kernel where the shape coincides)
//   id  WM WN TM TN   BM x BN    wave tile
//    0   2  2  2  2  128 x 128   64 x 64
//    1   1  4  2  2   64 x 256   64 x 64
//    2   1  4  1  2   32 x 256   32 x 64
//    3   2  2  1  1   64 x  64   32 x 32
//    4   1  4  2  1   64 x 128   64 x 32
//    5   2  2  2  1  128 x  64   64 x 32
//    6   1  4  2  4   64 x 512   64 x 128
//    7   2  2  2  4  128 x 256   64 x 128
static const int kDirectDims[][2] = {{128, 128}, {64, 256}, {32, 256}, {64, 64}, {64, 128}, {128, 64}, {64, 512}, {128, 256}};

template <int WM, int WN, int TM, int TN>
static hipError_t launch_dcfg(ConvParams p, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    p.m_tiles = (p.Cout + BM - 1) / BM;
    p.n_tiles = (p.Ntotal + BN - 1) / BN;
    dim3 grid(p.m_tiles * p.n_tiles, p.transposed ? 8 : 1, 1);
    hipLaunchKernelGGL((conv_direct_kernel<WM, WN, TM, TN>), grid, dim3(256), 0, stream, p);
    return hipGetLastError();
}

int conv_direct_pick_tile(const ConvParams& p) {
    const int classes = p.transposed ? 8 : 1;
    auto wgs = [&](int cfg) {
        const long bm = kDirectDims[cfg][0], bn = kDirectDims[cfg][1];
        return ((p.Cout + bm - 1) / bm) * ((p.Ntotal + bn - 1) / bn) * classes;
    };
    if (p.Cout <= 32) return 2;
    if (p.Cout <= 64) {
        if (wgs(1) >= 512) return 1;
        if (wgs(4) >= 512) return 4;
        return 3;
    }
    if (wgs(0) >= 512) return 0;
    if (wgs(5) >= 512) return 5;
    return 3;
}

hipError_t launch_conv_direct(const ConvParams& p, int cfg, hipStream_t stream) {
    if (cfg == 15) cfg = conv_direct_pick_tile(p);
    switch (cfg) {
        case 0: return launch_dcfg<2, 2, 2, 2>(p, stream);
        case 1: return launch_dcfg<1, 4, 2, 2>(p, stream);
        case 2: return launch_dcfg<1, 4, 1, 2>(p, stream);
        case 3: return launch_dcfg<2, 2, 1, 1>(p, stream);
        case 4: return launch_dcfg<1, 4, 2, 1>(p, stream);
        case 5: return launch_dcfg<2, 2, 2, 1>(p, stream);
        case 6: return launch_dcfg<1, 4, 2, 4>(p, stream);
        case 7: return launch_dcfg<2, 2, 2, 4>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing for the direct kernel:  wd[cls][cc][tap][h][cout][ks],  cin = cc*16 + 2*ks + h
__global__ void pack_direct_kernel(const float* __restrict__ w, float* __restrict__ wd, int Cin, int Cout,
                                   int CoutPad, int T, int transposed) {
    const size_t per_cls = (size_t)T * Cin * CoutPad;
    const size_t total = (transposed ? 8 : 1) * per_cls;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = i / per_cls;
        size_t r = i % per_cls;
        const int ks = r & 7; r >>= 3;
        const int co = r % CoutPad; r /= CoutPad;
        const int hh = r & 1; r >>= 1;
        const int tap = r % T;
        const int cc = r / T;
        const int cin = cc * 16 + 2 * ks + hh;
        float v = 0.f;
        if (co < Cout) {
            if (!transposed) {
                v = w[((size_t)co * Cin + cin) * T + tap];
            } else {
                const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
                const int td = (tap >> 2) & 1, th = (tap >> 1) & 1, tw = tap & 1;
                const int kd = rd ? 2 - 2 * td : 1 + 2 * td;
                const int kh = rh ? 2 - 2 * th : 1 + 2 * th;
                const int kw = rw ? 2 - 2 * tw : 1 + 2 * tw;
                v = w[((size_t)cin * Cout + co) * 64 + (kd * 4 + kh) * 4 + kw];
            }
        }
        wd[i] = v;
    }
}

hipError_t launch_pack_direct(const float* w, float* wd, int Cin, int Cout, int CoutPad, int T, int transposed,
                              hipStream_t s) {
    hipLaunchKernelGGL(pack_direct_kernel, dim3(1024), dim3(256), 0, s, w, wd, Cin, Cout, CoutPad, T, transposed);
    return hipGetLastError();
}

}  // namespace s3r
