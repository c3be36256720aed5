// Direct (im2col-free) 2D/3D convolution and transposed convolution as an implicit GEMM on the
// gfx950 fp32 matrix cores, with the folded-BN affine + activation fused into the epilogue.
//
//   D[cout][pos] = sum_{tap, cin} Wp[tap][cin][cout] * X[b(pos)][cin][in(pos) + tap]
//
//   GEMM M = Cout, N = B*Do*Ho*Wo output positions, K = taps*Cin (tap-major, so one 16-deep K tile
//   is ONE tap x 16 consecutive input channels: the validity predicate and the address delta are
//   wave-uniform per tile and the gather is 16 coalesced dword loads per lane, nothing else).
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain, 64 FLOP/clk/SIMD = the chip's
// fp32 roof of 157 TFLOP/s; there is no TF32-style fast path on gfx950).  Operand maps (64-lane wave):
//   A: lane l holds A[i = l&31][k = l>>5]        (weights,  i = cout)
//   B: lane l holds B[k = l>>5][j = l&31]        (gathered input, j = position)
//   D: reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// so every accumulator register is 32 consecutive output positions of one cout: NCHW stores are
// 128-byte coalesced segments with no transpose.
//
// Workgroup = 4 waves, tile BM x BN = (32*WM*TM) x (32*WN*TN); both operands are staged through a
// double-buffered LDS tile ([16][BM] weights, [16][BN] gathered input) with the global loads of tile
// k+1 in flight under the MFMAs of tile k (register-staged, one barrier per K tile).  Because the
// fp32 MFMA takes 64 cycles, one ds_read_b32 per operand per MFMA is <5% of LDS bandwidth: the
// kernel is bounded by matrix-core issue, which is what the roofline in bench.py prices it against.
//
// ConvTranspose3d(k=4,s=2,p=1) is run as 8 output-parity classes (blockIdx.y); each class is a
// 2x2x2-tap gather over the input grid with its own packed weight slab, so there are no atomics
// and no zero-stuffed taps (output-stationary gather formulation).
#include "s3r_kernels.h"

namespace s3r {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));

template <int WM, int WN, int TM, int TN, int VAR, int BK>
__global__ __launch_bounds__(256) void conv_mfma_kernel(const ConvParams p) {
    constexpr int BM = 32 * WM * TM;
    constexpr int BN = 32 * WN * TN;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    static_assert(BN == 64 || BN == 128 || BN == 256, "gather mapping assumes BN in {64,128,256}");
    static_assert(BK == 16 || BK == 32, "K tile depth");
    constexpr int KG = 256 / BN;              // k-groups among the 256 threads
    constexpr int LPT = BK / KG;              // gathered dwords per thread per K tile
    constexpr int A_F4 = BK * BM / 4;         // float4s in one weight tile
    constexpr int A_PT = (A_F4 + 255) / 256;  // float4s per thread

    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* As = smem;                  // [2][BK][BM]
    float* Bs = smem + 2 * BK * BM;    // [2][BK][BN]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;

    // ---- XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch), so give each
    // XCD a contiguous run of tiles: neighbouring N tiles share their input halo in that XCD's L2.
    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m_tile = bid % p.m_tiles;
    const int n_tile = bid / p.m_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * BN;
    const int cls = blockIdx.y;                       // parity class (transposed only)
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;

    const int HWi = p.Hi * p.Wi;
    const int DHWi = p.Di * HWi;
    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.kd * p.kh * p.kw;
    const int nchunk = p.Cin / BK;
    const int nkt = T * nchunk;
    const float* __restrict__ wbase = p.w + (size_t)cls * T * p.Cin * p.CoutPad;

    // ---- per-thread gather position (fixed for the whole K loop) ----
    const int gn = tid % BN;            // position inside the tile
    const int kg = tid / BN;            // which LPT-slice of the 16 channels this thread fetches
    int base;                           // element offset of (b, cin=0, id0, ih0, iw0); may be negative
    unsigned md = 0, mh = 0, mw = 0;    // per-axis tap validity bit masks
    int sd = 1, sh = 1, sw = 1;         // tap step sign per axis
    {
        const int n = n0 + gn;
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
            id0 = pd * p.stride - p.pad_d;
            ih0 = ph * p.stride - p.pad_h;
            iw0 = pw * p.stride - p.pad_w;
        } else {
            id0 = pd; ih0 = ph; iw0 = pw;
            sd = rd ? 1 : -1; sh = rh ? 1 : -1; sw = rw ? 1 : -1;
        }
        for (int t = 0; t < p.kd; ++t) md |= ((unsigned)(id0 + sd * t) < (unsigned)p.Di) ? (1u << t) : 0u;
        for (int t = 0; t < p.kh; ++t) mh |= ((unsigned)(ih0 + sh * t) < (unsigned)p.Hi) ? (1u << t) : 0u;
        for (int t = 0; t < p.kw; ++t) mw |= ((unsigned)(iw0 + sw * t) < (unsigned)p.Wi) ? (1u << t) : 0u;
        if (!nvalid) md = 0;
        base = (b * p.Cin + kg * LPT) * DHWi + id0 * HWi + ih0 * p.Wi + iw0;
    }

    // ---- K-tile cursor of the NEXT tile to fetch (wave-uniform scalars) ----
    int c_td = 0, c_th = 0, c_tw = 0, c_tap = 0, c_cc = 0;

    // input gather goes through a buffer descriptor: per-lane byte offset in a VGPR, the per-channel
    // step in an SGPR (soffset) -> no per-load vector address arithmetic at all.
    const __amdgpu_buffer_rsrc_t xrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.x), 0, (int)((unsigned)p.B * (unsigned)p.Cin * (unsigned)DHWi * 4u), 0x00020000);
    const int chan_step = DHWi * 4;     // bytes between consecutive input channels

    float breg[LPT];
    v4f areg0 = {0.f, 0.f, 0.f, 0.f}, areg1 = {0.f, 0.f, 0.f, 0.f};
    static_assert(A_PT <= 2, "at most two float4 of weights per thread per K tile");
    const int aq0 = tid, aq1 = tid + 256;
    const int ak0 = aq0 / (BM / 4), am0 = (aq0 % (BM / 4)) * 4;
    const int ak1 = aq1 / (BM / 4), am1 = (aq1 % (BM / 4)) * 4;

#define S3R_FETCH()                                                                                        \
    {                                                                                                      \
        const float* __restrict__ wrow = wbase + ((size_t)(c_tap * p.Cin + c_cc * BK)) * p.CoutPad + m0;   \
        if (A_F4 >= 256 || aq0 < A_F4)                                                                     \
            areg0 = *reinterpret_cast<const v4f*>(wrow + (size_t)ak0 * p.CoutPad + am0);                   \
        if (A_PT == 2) areg1 = *reinterpret_cast<const v4f*>(wrow + (size_t)ak1 * p.CoutPad + am1);        \
        const bool v = ((md >> c_td) & (mh >> c_th) & (mw >> c_tw) & 1u) != 0;                             \
        const int off = (VAR == 3) ? (gn & 1023)                                                            \
                      : base + (sd * c_td * p.Hi + sh * c_th) * p.Wi + sw * c_tw + c_cc * BK * DHWi;       \
        if (v || VAR == 3) {                                                                               \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i) breg[i] = __builtin_bit_cast(                  \
                float, __builtin_amdgcn_raw_buffer_load_b32(xrsrc, off * 4, i * chan_step, 0));            \
        } else {                                                                                           \
            _Pragma("unroll") for (int i = 0; i < LPT; ++i) breg[i] = 0.f;                                 \
        }                                                                                                  \
        if (VAR >= 2) { /* chunk-major, tap-minor: the taps of one 16-channel chunk run back to back */     \
            if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }                    \
            if (++c_tap == T) { c_tap = 0; c_td = 0; c_th = 0; c_tw = 0; ++c_cc; }                         \
        } else if (++c_cc == nchunk) {                                                                     \
            c_cc = 0; ++c_tap;                                                                             \
            if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }                    \
        }                                                                                                  \
    }
#define S3R_STAGE(buf)                                                                                     \
    {                                                                                                      \
        float* sa = As + (buf) * BK * BM;                                                                  \
        float* sb = Bs + (buf) * BK * BN;                                                                  \
        if (A_F4 >= 256 || aq0 < A_F4) *reinterpret_cast<v4f*>(sa + aq0 * 4) = areg0;                      \
        if (A_PT == 2) *reinterpret_cast<v4f*>(sa + aq1 * 4) = areg1;                                      \
        _Pragma("unroll") for (int i = 0; i < LPT; ++i) sb[(kg * LPT + i) * BN + gn] = breg[i];            \
    }

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    S3R_FETCH();
    S3R_STAGE(0);
    __syncthreads();

    const int a_off = wm * TM * 32 + j;
    const int b_off = wn * TN * 32 + j;

    if (VAR == 0) {
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < nkt);
            if (more) S3R_FETCH();
            const float* a = As + cur * BK * BM + h * BM + a_off;
            const float* b = Bs + cur * BK * BN + h * BN + b_off;
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                float av[TM], bv[TN];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[tm] = a[ks * 2 * BM + tm * 32];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) bv[tn] = b[ks * 2 * BN + tn * 32];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
            }
            if (more) S3R_STAGE(cur ^ 1);
            __syncthreads();
        }
    } else {
        // VAR 1: operand fragments of k-step ks+1 are read from LDS BEFORE the MFMAs of k-step ks are
        // issued (an in-order wave otherwise reaches its next ds_read only after its 4th MFMA has been
        // accepted by the pipe, and the read latency shows as an idle matrix pipe), and the next tile's
        // registers are written to the other LDS buffer in the MIDDLE of the tile, under MFMAs in flight.
        for (int kt = 0; kt < nkt; ++kt) {
            const int cur = kt & 1;
            const bool more = (kt + 1 < nkt);
            if (more) S3R_FETCH();
            const float* a = As + cur * BK * BM + h * BM + a_off;
            const float* b = Bs + cur * BK * BN + h * BN + b_off;
            float av[2][TM], bv[2][TN];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[0][tm] = a[tm * 32];
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) bv[0][tn] = b[tn * 32];
#pragma unroll
            for (int ks = 0; ks < BK / 2; ++ks) {
                const int c = ks & 1, n = c ^ 1;
                if (ks + 1 < BK / 2) {
#pragma unroll
                    for (int tm = 0; tm < TM; ++tm) av[n][tm] = a[(ks + 1) * 2 * BM + tm * 32];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) bv[n][tn] = b[(ks + 1) * 2 * BN + tn * 32];
                }
                __builtin_amdgcn_sched_barrier(0);     // keep the prefetch reads ABOVE this step's MFMAs
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(av[c][tm], bv[c][tn], acc[tm][tn], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
                if (ks == BK / 4 && more) S3R_STAGE(cur ^ 1);
            }
            __syncthreads();
        }
    }

    // ---- epilogue: y = act(acc * scale[cout] + shift[cout]), NC(D)HW, 32 consecutive positions per store
    const int So = p.Do * p.Ho * p.Wo;
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = n0 + wn * TN * 32 + tn * 32 + j;
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
                const int m = m0 + wm * TM * 32 + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
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

// ------------------------------------------------------------------------------------------------
// tile configurations
//   id  WM WN TM TN   BM x BN
//    0   2  2  2  2  128 x 128   wide layers, large N
//    1   1  4  2  2   64 x 256   Cout == 64
//    2   1  4  1  2   32 x 256   Cout <= 32
//    3   2  2  1  1   64 x  64   small N: more workgroups
//    4   1  4  2  1   64 x 128
//    5   2  2  2  1  128 x  64
static const int kTileDims[][2] = {{128, 128}, {64, 256}, {32, 256}, {64, 64}, {64, 128}, {128, 64},
                                  {64, 64}, {32, 128}, {32, 128}, {64, 128}, {128, 64}};

void conv_tile_dims(int cfg, int* bm, int* bn) {
    *bm = kTileDims[cfg][0];
    *bn = kTileDims[cfg][1];
}

int conv_pick_tile(const ConvParams& p) {
    // Measured on MI355X (tools/layer_bench.py, B=32, every layer of arch_spec, all tile shapes): the
    // 64x64 tile (one MFMA tile per wave, 8 workgroups per CU at BK=16 / 5 at BK=32) wins or ties on
    // every layer: the finer grain shortens the last partial round of workgroups, and many resident
    // waves per SIMD cover each other's barrier waits.  BK=32 halves the barriers per MFMA and is
    // worth +2 % overall (+20 % on the layers with fewer workgroups than CU slots: v4, v5, v6, d1).
    const bool k32 = (p.Cin % 32) == 0;
    if (p.Cout <= 32) return k32 ? 8 : 7;
    return k32 ? 6 : 3;
}

template <int WM, int WN, int TM, int TN, int VAR, int BK = 16>
static hipError_t launch_cfg(ConvParams p, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    if (p.Cin % BK != 0) return hipErrorInvalidValue;
    p.m_tiles = (p.Cout + BM - 1) / BM;
    p.n_tiles = (p.Ntotal + BN - 1) / BN;
    const size_t lds = (size_t)2 * BK * (BM + BN) * sizeof(float);
    dim3 grid(p.m_tiles * p.n_tiles, p.transposed ? 8 : 1, 1);
    hipLaunchKernelGGL((conv_mfma_kernel<WM, WN, TM, TN, VAR, BK>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

template <int VAR>
static hipError_t launch_var(const ConvParams& p, int cfg, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_cfg<2, 2, 2, 2, VAR>(p, stream);
        case 1: return launch_cfg<1, 4, 2, 2, VAR>(p, stream);
        case 2: return launch_cfg<1, 4, 1, 2, VAR>(p, stream);
        case 3: return launch_cfg<2, 2, 1, 1, VAR>(p, stream);
        case 4: return launch_cfg<1, 4, 2, 1, VAR>(p, stream);
        case 5: return launch_cfg<2, 2, 2, 1, VAR>(p, stream);
        case 6: return launch_cfg<2, 2, 1, 1, VAR, 32>(p, stream);   // 64 x 64, BK 32
        case 7: return launch_cfg<1, 4, 1, 1, VAR>(p, stream);       // 32 x 128
        case 8: return launch_cfg<1, 4, 1, 1, VAR, 32>(p, stream);   // 32 x 128, BK 32
        case 9: return launch_cfg<1, 4, 2, 1, VAR, 32>(p, stream);   // 64 x 128, BK 32
        case 10: return launch_cfg<4, 1, 1, 2, VAR>(p, stream);      // 128 x 64 (4 waves along M)
        default: return hipErrorInvalidValue;
    }
}

// code = tile_cfg + 16 * variant; tile_cfg 15 = heuristic
hipError_t launch_conv_mfma(const ConvParams& p, int code, hipStream_t stream) {
    int cfg = code & 15;
    const int var = code >> 4;
    if (cfg == 15) cfg = conv_pick_tile(p);
    switch (var) {
        case 1: return launch_var<1>(p, cfg, stream);   // tap-major K order
        case 0:
        case 2: return launch_var<2>(p, cfg, stream);   // chunk-major K order (default)
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing (device-side, run once per parameter update): K-major rows of CoutPad couts
//   conv   : w[Cout][Cin][T]           -> wp[(tap*Cin + cin)*CoutPad + cout]
//   deconv : w[Cin][Cout][4][4][4]     -> wp[cls][(tap*Cin + cin)*CoutPad + cout],  tap in 2x2x2
//            kernel index along an axis with output parity r and tap t:  r==0 ? 1+2t : 2-2t
__global__ void pack_conv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cout, int Cin, int T,
                                 int CoutPad) {
    const size_t total = (size_t)T * Cin * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int co = i % CoutPad;
        const size_t row = i / CoutPad;
        const int cin = row % Cin;
        const int tap = row / Cin;
        wp[i] = co < Cout ? w[((size_t)co * Cin + cin) * T + tap] : 0.f;
    }
}

__global__ void pack_deconv_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout,
                                   int CoutPad) {
    const size_t per_cls = (size_t)8 * Cin * CoutPad;
    const size_t total = 8 * per_cls;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = i / per_cls;
        const size_t r = i % per_cls;
        const int co = r % CoutPad;
        const size_t row = r / CoutPad;
        const int cin = row % Cin;
        const int tap = row / Cin;
        const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
        const int td = (tap >> 2) & 1, th = (tap >> 1) & 1, tw = tap & 1;
        const int kd = rd ? 2 - 2 * td : 1 + 2 * td;
        const int kh = rh ? 2 - 2 * th : 1 + 2 * th;
        const int kw = rw ? 2 - 2 * tw : 1 + 2 * tw;
        wp[i] = co < Cout ? w[((size_t)cin * Cout + co) * 64 + (kd * 4 + kh) * 4 + kw] : 0.f;
    }
}

hipError_t launch_pack_conv(const float* w, float* wp, int Cout, int Cin, int T, int CoutPad, hipStream_t s) {
    hipLaunchKernelGGL(pack_conv_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cout, Cin, T, CoutPad);
    return hipGetLastError();
}

hipError_t launch_pack_deconv_k4s2(const float* w, float* wp, int Cin, int Cout, int CoutPad, hipStream_t s) {
    hipLaunchKernelGGL(pack_deconv_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad);
    return hipGetLastError();
}

}  // namespace s3r
