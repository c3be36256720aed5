// bf16 MFMA path (BASELINE.json configs[2]): 2D/3D convolution and transposed convolution as an implicit
// GEMM on v_mfma_f32_32x32x16_bf16, bf16 activations in CHANNELS-LAST layout with a zero halo, fp32
// accumulation, folded-BN affine + activation + bf16 rounding fused into the epilogue.
//
//   D[pos][cout] = sum_{chunk, tap, c} X[b(pos)][in(pos) + tap][chunk*32 + c] * Wp[(chunk*T + tap)][cout][c]
//
//   GEMM M = B*Nd*Nh*Nw positions (A operand: gathered activations), N = Cout (B operand: weights),
//   K = Cin*T in K tiles of ONE tap x 32 channels = 64 contiguous bytes per position.
//
// Why channels-last here when the fp32 path is NCHW: the bf16 MFMA takes 8 consecutive k per lane
// (A: lane (r,h) holds A[row r][k = 8h..8h+7]), i.e. 16 contiguous BYTES of one position's channel
// vector — one ds_read_b128 — whereas the fp32 MFMA takes one k per lane and wants position-contiguous
// rows.  The output falls out channels-last as well: D has the cout on the LANE and the position in the
// registers, so a register row is 32 lanes x consecutive couts = one contiguous run per position.
//
// LDS images (per K tile): As[pos][4 slots x 16 B], Bs[cout row][4 slots x 16 B], slot = kgroup ^
// ((row >> 2) & 3).  With 64-byte rows a plain image makes every ds_read_b128 lane group hit the same
// 16-byte column of four rows; the XOR spreads the four rows of each bank-row residue over the four
// slots (conflict-free for the b128 lane groups {0-3,12-15,20-27}, ...).  Both operands arrive by
// LDS-DMA (16 B per lane, lane-linear destination), so the swizzle is applied on the SOURCE side: a lane
// fetches channel group slot ^ f(row) of its position; weights are stored pre-swizzled by the pack kernel.
//
// Workgroup: 4 waves along M, each 32*TM positions x 64 couts (TN = 2 MFMA tiles, couts interleaved
// 2c+tn so that a lane packs its two bf16 results into ONE dword store: 128 contiguous bytes per 32
// lanes).  Per-position input / output offsets are decoded once per workgroup into LDS.
#include "s3r_kernels.h"
#include <cstdlib>

namespace s3r {

#ifdef S3R_ABLATE   // diagnostic builds only: S3R_ABL=1 no epilogue stores, 2 one K tile only, 3 no DMA in the loop
static int abl_mode_h() { static const int m = getenv("S3R_ABL") ? atoi(getenv("S3R_ABL")) : 0; return m; }
#define S3R_ABLH(p, m) ((p).debug == (m))
#else
#define S3R_ABLH(p, m) false
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));
typedef __bf16 bf16x2 __attribute__((ext_vector_type(2)));

#define S3R_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

__device__ __forceinline__ unsigned pack_bf16(float lo, float hi) {
    const bf16x2 v = {(__bf16)lo, (__bf16)hi};      // one v_cvt_pk_bf16_f32 (round to nearest even)
    return __builtin_bit_cast(unsigned, v);
}

__device__ __forceinline__ void dma16(__amdgpu_buffer_rsrc_t rsrc, char* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR(lds_dst), 16, voffset, soffset, 0, 0);
}

constexpr int HKC = 32;    // channels per K tile
constexpr int HBN = 64;    // couts per workgroup

// Epilogue shared by both bf16 kernels.  Lane c owns couts n0 + 2c (tn 0) and n0 + 2c + 1 (tn 1); register r of
// MFMA tile tm is position row wave*32*TM + tm*32 + (r&3) + 8*(r>>2) + 4h of the tile; yoff[row] is that
// position's output element offset (-1: past the end).
template <int TM>
__device__ __forceinline__ void epilogue_h(const ConvParamsH& p, f32x16 (&acc)[TM][2], const int* yoff, int wave, int c,
                                           int h, int m0, int n0, int cls, int kz, int bm) {
    const int co = n0 + 2 * c;
    if (p.ksplit > 1) {
        // split-K: fp32 partial sums [cls][kz][position][CoutPad]; conv_finish_bf16 reduces in kz order
        const int mpad = p.m_tiles * bm;
        float* __restrict__ slab = p.part + ((size_t)(cls * p.ksplit + kz) * mpad + m0) * p.CoutPad + co;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int row = wave * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                float2 t = {acc[tm][0][r], acc[tm][1][r]};
                *reinterpret_cast<float2*>(slab + (size_t)row * p.CoutPad) = t;
            }
        return;
    }
    const bool c0 = co < p.Cout, c1 = co + 1 < p.Cout;
    const float sc0 = (c0 && p.scale) ? p.scale[co] : 1.f, sf0 = (c0 && p.shift) ? p.shift[co] : 0.f;
    const float sc1 = (c1 && p.scale) ? p.scale[co + 1] : 1.f, sf1 = (c1 && p.shift) ? p.shift[co + 1] : 0.f;
    // ReLU / identity as ONE v_max against a wave-uniform floor; sigmoid (rare) behind a wave-uniform flag
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const bool sig = p.act == ACT_SIGMOID;
    if (p.head_w) {
        // fused pointwise head (conv -> 1x1x1 conv to ONE channel + activation; the workgroup's 64-cout tile
        // is the whole channel axis): per position, 2 FMAs in the lane, a 32-lane butterfly over the couts, one
        // fp32 store.  The conv's own bf16 output — the largest activation of the network — is never written.
        const float hw0 = c0 ? p.head_w[co] : 0.f, hw1 = c1 ? p.head_w[co + 1] : 0.f;
        const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
        float* __restrict__ y = reinterpret_cast<float*>(p.y);
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                float v0 = fmaxf(fmaf(acc[tm][0][r], sc0, sf0), lo), v1 = fmaxf(fmaf(acc[tm][1][r], sc1, sf1), lo);
                float t = fmaf(v1, hw1, v0 * hw0);
#pragma unroll
                for (int o = 16; o > 0; o >>= 1) t += __shfl_xor(t, o, 64);      // over the 32 lanes of this half
                const int row = wave * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
                const int ye = yoff[row];
                if (c == 0 && ye >= 0) {
                    t = fmaf(t, hsc, hsf);
                    if (p.head_act == ACT_RELU) t = fmaxf(t, 0.f);
                    else if (p.head_act == ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
                    y[ye] = t;
                }
            }
        return;
    }
    unsigned short* __restrict__ y = reinterpret_cast<unsigned short*>(p.y);
#pragma unroll
    for (int tm = 0; tm < TM; ++tm)
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int row = wave * 32 * TM + tm * 32 + (r & 3) + 8 * (r >> 2) + 4 * h;
            const int ye = yoff[row];
            if (ye < 0 || !c0) continue;
            float v0 = fmaf(acc[tm][0][r], sc0, sf0), v1 = fmaf(acc[tm][1][r], sc1, sf1);
            if (S3R_ABLH(p, 1) && v0 != 12345.f) continue;
            if (sig) { v0 = __builtin_amdgcn_rcpf(1.f + __expf(-v0)); v1 = __builtin_amdgcn_rcpf(1.f + __expf(-v1)); }
            else { v0 = fmaxf(v0, lo); v1 = fmaxf(v1, lo); }
            const unsigned pk = pack_bf16(v0, v1);
            if (c1) *reinterpret_cast<unsigned*>(y + (size_t)ye + co) = pk;
            else y[(size_t)ye + co] = (unsigned short)(pk & 0xffffu);
        }
}

constexpr int min_waves_h(int tm) { return tm >= 4 ? 2 : 4; }

template <int TM>
__global__ __launch_bounds__(256, min_waves_h(TM)) void conv_bf16_kernel(const ConvParamsH p) {
    constexpr int BM = 128 * TM;
    constexpr int NPA = BM / 64;           // A pieces (16 positions x 64 B) per wave per K tile
    constexpr int A_BYTES = BM * 64, B_BYTES = HBN * 64;

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                               // [2][BM][64 B]
    char* Bs = smem + 2 * A_BYTES;                 // [2][64][64 B]
    int* xoff = reinterpret_cast<int*>(smem + 2 * A_BYTES + 2 * B_BYTES);   // [BM] input byte offsets
    int* yoff = xoff + BM;                                                   // [BM] output element offsets, -1 = none

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n_tile = bid % p.n_tiles;            // cout tile (fastest: neighbours share the gathered input)
    const int m_tile = bid / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * HBN;
    const int cls = blockIdx.y;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int kz = blockIdx.z;

    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.T;
    const int chunks = (p.Cin / HKC) / p.ksplit;
    const int nkt = S3R_ABLH(p, 2) ? 1 : T * chunks;

    // ---- decode this tile's positions once: input corner (bytes) and output offset (elements)
    for (int t = tid; t < BM; t += 256) {
        const int n = m0 + t;
        const bool ok = n < p.Ntotal;
        const int nn = ok ? n : p.Ntotal - 1;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        int xe = b * p.x_bs + p.x_org + (pd * p.x_ds + ph * p.x_hs + pw * p.x_ws) * p.stride;
        const int ostep = p.transposed ? 2 : 1;
        int ye = b * p.y_bs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw * p.y_ws) * ostep;
        if (p.transposed) {
            xe += (rd - 1) * p.x_ds + (rh - 1) * p.x_hs + (rw - 1) * p.x_ws;
            ye += rd * p.y_ds + rh * p.y_hs + rw * p.y_ws;
        }
        xoff[t] = xe * 2;
        yoff[t] = ok ? ye : -1;
    }
    __syncthreads();

    // ---- loop-invariant DMA offsets
    int avoff[NPA];
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int pl = (wave + 4 * q) * 16 + (lane >> 2);          // position inside the tile
        const int kg = (lane & 3) ^ ((pl >> 2) & 3);               // swizzle on the source side
        avoff[q] = xoff[pl] + kg * 16;
    }
    const int bvoff = lane * 16;                                    // weights are stored pre-swizzled

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // packed weights: [cls][kt][cout tile][64 rows][64 B]
    const size_t w_cls = (size_t)cls * T * (p.Cin / HKC) * p.n_tiles * B_BYTES;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + w_cls), 0,
        (int)((unsigned)T * (unsigned)(p.Cin / HKC) * (unsigned)p.n_tiles * (unsigned)B_BYTES), 0x00020000);

    int c_td = 0, c_th = 0, c_tw = 0, c_tap = 0, c_cc = kz * chunks, c_kt = kz * nkt;

    auto issue = [&](int buf) {
        char* sb = Bs + buf * B_BYTES + wave * 1024;
        dma16(wrsrc, sb, bvoff, (c_kt * p.n_tiles + n_tile) * B_BYTES + wave * 1024);
        char* sa = As + buf * A_BYTES + wave * 1024;
        const int a_base = (c_cc * HKC + (c_td * p.x_ds + c_th * p.x_hs + c_tw * p.x_ws)) * 2;
#pragma unroll
        for (int q = 0; q < NPA; ++q)
            dma16(xrsrc, sa + q * 4096, avoff[q], a_base);
        ++c_kt;
        if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }
        if (++c_tap == T) { c_tap = 0; c_td = 0; c_th = 0; c_tw = 0; ++c_cc; }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // fragment byte offsets inside a K tile image: row*64 + ((h + 2q) ^ f(row))*16; q toggles bit 5
    int a_off[TM], b_off[2];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) {
        const int row = wave * 32 * TM + tm * 32 + c;
        a_off[tm] = row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
    }
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int row = tn * 32 + c;
        b_off[tn] = row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
    }

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt && !S3R_ABLH(p, 3)) issue(cur ^ 1);
        const char* a = As + cur * A_BYTES;
        const char* b = Bs + cur * B_BYTES;
#pragma unroll
        for (int q = 0; q < 2; ++q) {
            bf16x8 av[TM], bv[2];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) av[tm] = *reinterpret_cast<const bf16x8*>(a + (a_off[tm] ^ (q << 5)));
#pragma unroll
            for (int tn = 0; tn < 2; ++tn) bv[tn] = *reinterpret_cast<const bf16x8*>(b + (b_off[tn] ^ (q << 5)));
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    epilogue_h<TM>(p, acc, yoff, wave, c, h, m0, n0, cls, kz, BM);
}

// ------------------------------------------------------------------------------------------------
// Row-reuse variant (the default): the kw taps of one (chunk, td, th) group read the SAME gathered input
// rows shifted by one position, so the A operand is fetched ONCE per group instead of once per tap.
//
// LDS A image = the input positions the tile needs for a fixed (td, th), in input order: for every run of
// tile positions inside one output row, a segment of stride*(run-1)+kw consecutive input positions
// (64 B = 32 channels each); MFMA row r reads LDS row lrow[r] + tw for tap tw.  No position is computed that
// is not stored (the halo columns sit in LDS but no MFMA row maps to them), the gather is kw (x stride)
// times smaller, and it is CONTIGUOUS in HBM (whole runs of positions) instead of 64-byte pieces.
// The weights of the group's kw taps (kw x 4 KiB, consecutive in the packed image) ride along, so there
// is one barrier per GROUP: kw*2*TM*2 MFMAs per wave between barriers.
constexpr int NPA_MAX = 10;    // 16-position A pieces per wave per group, upper bound (registers)

template <int TM>
__global__ __launch_bounds__(256, 2) void conv_bf16r_kernel(const ConvParamsH p, int r_max) {
    constexpr int BM = 128 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kw = p.kw;
    const int a_bytes = r_max * 64, stage_bytes = a_bytes + kw * 4096;
    int* yoff = reinterpret_cast<int*>(smem + 2 * stage_bytes);     // [BM] output element offsets, -1 = none
    int* lrow = yoff + BM;                                            // [BM] LDS row of each tile position (tap tw = 0)

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c = lane & 31, h = lane >> 5;

    const int nwg = gridDim.x;
    int bid = blockIdx.x;
    {
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n_tile = bid % p.n_tiles;
    const int m_tile = bid / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * HBN;
    const int cls = blockIdx.y;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int kz = blockIdx.z;

    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.T;
    const int chunks = (p.Cin / HKC) / p.ksplit;
    const int ngroups = S3R_ABLH(p, 2) ? 1 : chunks * p.kd * p.kh;
    const int cls_x = p.transposed ? (rd - 1) * p.x_ds + (rh - 1) * p.x_hs + (rw - 1) * p.x_ws : 0;

    // ---- geometry of the tile's first position (wave-uniform): its output row and column
    int rowid0, pw0;
    {
        const int b = p.dS.div(m0);
        int rem = m0 - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        pw0 = rem - ph * p.Nw;
        rowid0 = (b * p.Nd + pd) * p.Nh + ph;
    }
    const int seg0 = p.stride * (p.Nw - pw0 - 1) + kw;       // LDS rows of the first (partial) run
    const int segw = p.stride * (p.Nw - 1) + kw;             // LDS rows of a full output row
    const int last_row = p.B * p.Nd * p.Nh - 1;

    for (int t = tid; t < BM; t += 256) {
        const int n = m0 + t;
        const bool ok = n < p.Ntotal;
        const int nn = ok ? n : p.Ntotal - 1;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        const int ostep = p.transposed ? 2 : 1;
        int ye = b * p.y_bs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw * p.y_ws) * ostep;
        if (p.transposed) ye += rd * p.y_ds + rh * p.y_hs + rw * p.y_ws;
        yoff[t] = ok ? ye : -1;
        const int k = (b * p.Nd + pd) * p.Nh + ph - rowid0;
        lrow[t] = k == 0 ? p.stride * (pw - pw0) : seg0 + (k - 1) * segw + p.stride * pw;
    }

    // ---- loop-invariant DMA source offsets: LDS row q of the A image <- input position src(q)
    const int npa = r_max >> 6;                               // pieces per wave (r_max % 64 == 0)
    int avoff[NPA_MAX];
#pragma unroll
    for (int q = 0; q < NPA_MAX; ++q) {
        const int lr = ((wave + 4 * q) << 4) + (lane >> 2);   // LDS row this lane fills in its q-th piece
        int k, off;
        if (lr < seg0) { k = 0; off = lr + p.stride * pw0; }
        else { k = 1 + (lr - seg0) / segw; off = (lr - seg0) - (k - 1) * segw; }      // (segw depends on kw: plain divide)
        int rowid = rowid0 + k;
        if (rowid > last_row) rowid = last_row;              // past the tensor: any valid address, never read
        const int b = p.dDH.div(rowid);
        int rem = rowid - b * (p.Nd * p.Nh);
        const int pd = p.dH.div(rem);
        const int ph = rem - pd * p.Nh;
        const int e = b * p.x_bs + p.x_org + (pd * p.x_ds + ph * p.x_hs) * p.stride + off * p.x_ws + cls_x;
        const int kg = (lane & 3) ^ ((lr >> 2) & 3);          // swizzle on the source side
        avoff[q] = e * 2 + kg * 16;
    }
    const int bvoff = lane * 16;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const size_t w_cls = (size_t)cls * T * (p.Cin / HKC) * p.n_tiles * 4096;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + w_cls), 0,
        (int)((unsigned)T * (unsigned)(p.Cin / HKC) * (unsigned)p.n_tiles * 4096u), 0x00020000);

    int c_td = 0, c_th = 0, c_cc = kz * chunks, c_kt = kz * chunks * T;      // cursor of the NEXT group to fetch

    auto issue = [&](int buf) {
        char* st = smem + buf * stage_bytes;
        for (int tw = 0; tw < kw; ++tw)                                      // the group's kw weight tiles
            dma16(wrsrc, st + a_bytes + tw * 4096 + wave * 1024, bvoff,
                  ((c_kt + tw) * p.n_tiles + n_tile) * 4096 + wave * 1024);
        const int a_base = (c_cc * HKC + c_td * p.x_ds + c_th * p.x_hs) * 2;
#pragma unroll
        for (int q = 0; q < NPA_MAX; ++q)
            if (q < npa) dma16(xrsrc, st + ((wave + 4 * q) << 10), avoff[q], a_base);
        c_kt += kw;
        if (++c_th == p.kh) { c_th = 0; if (++c_td == p.kd) { c_td = 0; ++c_cc; } }
    };

    f32x16 acc[TM][2];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < 2; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // also publishes yoff / lrow

    int lr[TM];
#pragma unroll
    for (int tm = 0; tm < TM; ++tm) lr[tm] = lrow[wave * 32 * TM + tm * 32 + c];
    int b_off[2];
#pragma unroll
    for (int tn = 0; tn < 2; ++tn) {
        const int row = tn * 32 + c;
        b_off[tn] = row * 64 + ((h ^ ((row >> 2) & 3)) << 4);
    }

    for (int g = 0; g < ngroups; ++g) {
        const int cur = g & 1;
        if (g + 1 < ngroups && !S3R_ABLH(p, 3)) issue(cur ^ 1);
        const char* a = smem + cur * stage_bytes;
        const char* b = a + a_bytes;
        for (int tw = 0; tw < kw; ++tw) {
            int a_off[TM];
#pragma unroll
            for (int tm = 0; tm < TM; ++tm) {
                const int row = lr[tm] + tw;
                a_off[tm] = (row << 6) + (((h ^ (row >> 2)) & 3) << 4);
            }
#pragma unroll
            for (int q = 0; q < 2; ++q) {
                bf16x8 av[TM], bv[2];
#pragma unroll
                for (int tm = 0; tm < TM; ++tm) av[tm] = *reinterpret_cast<const bf16x8*>(a + (a_off[tm] ^ (q << 5)));
#pragma unroll
                for (int tn = 0; tn < 2; ++tn)
                    bv[tn] = *reinterpret_cast<const bf16x8*>(b + tw * 4096 + (b_off[tn] ^ (q << 5)));
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int tn = 0; tn < 2; ++tn)
                        acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(av[tm], bv[tn], acc[tm][tn], 0, 0, 0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    epilogue_h<TM>(p, acc, yoff, wave, c, h, m0, n0, cls, kz, BM);
}

// split-K finish: y[pos][cout] = bf16(act(scale * sum_kz slab + shift)); one thread per (position, cout pair)
__global__ __launch_bounds__(256) void conv_finish_bf16_kernel(const ConvParamsH p, int mpad) {
    const int S = p.Nd * p.Nh * p.Nw;
    const int cls = blockIdx.y;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int half = p.CoutPad >> 1;
    const long long total = (long long)p.Ntotal * half;
    const int ostep = p.transposed ? 2 : 1;
    unsigned short* __restrict__ y = reinterpret_cast<unsigned short*>(p.y);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int n = (int)(i / half);
        const int co = (int)(i - (long long)n * half) * 2;
        if (co >= p.Cout) continue;
        const float* __restrict__ src = p.part + ((size_t)cls * p.ksplit * mpad + n) * p.CoutPad + co;
        float2 s = *reinterpret_cast<const float2*>(src);
        for (int z = 1; z < p.ksplit; ++z) {
            const float2 t = *reinterpret_cast<const float2*>(src + (size_t)z * mpad * p.CoutPad);
            s.x += t.x; s.y += t.y;
        }
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        int ye = b * p.y_bs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw * p.y_ws) * ostep;
        if (p.transposed) ye += rd * p.y_ds + rh * p.y_hs + rw * p.y_ws;
        const bool c1 = co + 1 < p.Cout;
        float v0 = fmaf(s.x, p.scale ? p.scale[co] : 1.f, p.shift ? p.shift[co] : 0.f);
        float v1 = c1 ? fmaf(s.y, p.scale ? p.scale[co + 1] : 1.f, p.shift ? p.shift[co + 1] : 0.f) : 0.f;
        if (p.act == ACT_RELU) { v0 = fmaxf(v0, 0.f); v1 = fmaxf(v1, 0.f); }
        else if (p.act == ACT_SIGMOID) { v0 = 1.f / (1.f + __expf(-v0)); v1 = 1.f / (1.f + __expf(-v1)); }
        const unsigned pk = pack_bf16(v0, v1);
        if (c1) *reinterpret_cast<unsigned*>(y + (size_t)ye + co) = pk;
        else y[(size_t)ye + co] = (unsigned short)(pk & 0xffffu);
    }
}

static int rowreuse_rows(const ConvParamsH& p, int bm);

// Tile / gather choice (tools/layer_bench.py --dtype bf16, B = 256, MI355X):
//   * stride-1 layers with >= 3 taps along w and a deep K (v1, v3, v5, e7, v6) gain 10-25 % from the row-reuse
//     gather (codes 9 / 10 = 128 / 256 positions);
//   * the shallow-K 2D layers and the big-output layers (e2-e5, d3) sit near their HBM floor: they want the
//     small-LDS per-tap kernel with 128-position tiles (more workgroups per CU to overlap loads and stores);
//   * stride-2 layers would need a 2x larger LDS image for the reuse: per-tap kernel.
int conv_bf16_pick_tm(const ConvParamsH& p) {
    const long classes = (p.transposed ? 8 : 1) * (long)p.ksplit;
    const long n_tiles = p.CoutPad / HBN;
    auto wgs = [&](int tm) { return ((p.Ntotal + 128 * tm - 1) / (128 * tm)) * n_tiles * classes; };
    // (transposed layers reuse 2 taps per group: worth it from Cin*T >= 2048 — d1 +7 %, d2 +9 %, d3 -5 %)
    const bool deep = p.transposed ? (long)p.Cin * p.T >= 2048 : (p.kw >= 3 && (long)p.Cin * p.T >= 64 * 27);
    const bool reuse = p.stride == 1 && deep && rowreuse_rows(p, 128) <= 64 * NPA_MAX;
    if (reuse) return (wgs(2) >= 1024 && rowreuse_rows(p, 256) <= 64 * NPA_MAX) ? 10 : 9;
    return 1;
}

int conv_bf16_pick_ksplit(const ConvParamsH& p) {
    // per-sample geometry at a nominal batch of 32 (batch-invariant, as in the fp32 path)
    const int chunks = p.Cin / HKC;
    const long S = (long)p.Nd * p.Nh * p.Nw;
    const long wg_nom = ((32 * S + 127) / 128) * (p.CoutPad / HBN) * (p.transposed ? 8 : 1);
    int ks = 1;
    while (wg_nom * ks < 1024 && chunks % (2 * ks) == 0 && (chunks / (2 * ks)) * p.T >= 64) ks *= 2;
    return ks;
}

int64_t conv_bf16_scratch_elems(const ConvParamsH& p, int tm) {
    if (p.ksplit <= 1) return 0;
    const int bm = 128 * (tm >= 9 ? tm - 8 : tm);
    const int64_t mpad = (int64_t)((p.Ntotal + bm - 1) / bm) * bm;
    return (int64_t)(p.transposed ? 8 : 1) * p.ksplit * mpad * p.CoutPad;
}

// LDS rows (64 B each) the row-reuse kernel's A image needs for a BM-position tile, rounded to whole pieces per wave
static int rowreuse_rows(const ConvParamsH& p, int bm) {
    const int nrows = (bm + p.Nw - 2) / p.Nw + 1;
    const int r = p.stride * bm + nrows * p.kw;
    return (r + 63) / 64 * 64;
}

template <int TM>
static hipError_t launch_tm_rowreuse(ConvParamsH p, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    p.m_tiles = (p.Ntotal + BM - 1) / BM;
    p.n_tiles = p.CoutPad / HBN;
    const int r_max = rowreuse_rows(p, BM);
    if (r_max > 64 * NPA_MAX) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * (r_max * 64 + p.kw * 4096) + 2 * BM * sizeof(int);
    static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16r_kernel<TM>),
                                                       hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024);
    if (attr != hipSuccess) return attr;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(p.m_tiles * p.n_tiles, p.transposed ? 8 : 1, p.ksplit);
    hipLaunchKernelGGL((conv_bf16r_kernel<TM>), grid, dim3(256), lds, stream, p, r_max);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
        const long long total = (long long)p.Ntotal * (p.CoutPad >> 1);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(conv_finish_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), p.transposed ? 8 : 1),
                           dim3(256), 0, stream, p, p.m_tiles * BM);
        e = hipGetLastError();
    }
    return e;
}

template <int TM>
static hipError_t launch_tm(ConvParamsH p, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    p.m_tiles = (p.Ntotal + BM - 1) / BM;
    p.n_tiles = p.CoutPad / HBN;
    const size_t lds = (size_t)2 * BM * 64 + 2 * HBN * 64 + 2 * BM * sizeof(int);
    if (lds > 48 * 1024) {
        static const hipError_t attr = hipFuncSetAttribute(reinterpret_cast<const void*>(&conv_bf16_kernel<TM>),
                                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
        if (attr != hipSuccess) return attr;
    }
    dim3 grid(p.m_tiles * p.n_tiles, p.transposed ? 8 : 1, p.ksplit);
    hipLaunchKernelGGL((conv_bf16_kernel<TM>), grid, dim3(256), lds, stream, p);
    hipError_t e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
        const long long total = (long long)p.Ntotal * (p.CoutPad >> 1);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(conv_finish_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), p.transposed ? 8 : 1),
                           dim3(256), 0, stream, p, p.m_tiles * BM);
        e = hipGetLastError();
    }
    return e;
}

hipError_t launch_conv_bf16(const ConvParamsH& pin, int tm, hipStream_t stream) {
    ConvParamsH p = pin;
#ifdef S3R_ABLATE
    p.debug = abl_mode_h();
#endif
    if (p.Cin % HKC != 0 || p.CoutPad % HBN != 0 || p.ksplit < 1 || (p.Cin / HKC) % p.ksplit != 0 ||
        (p.ksplit > 1 && !p.part))
        return hipErrorInvalidValue;
    // tm = 1, 2, 4: per-tap gather (conv_bf16_kernel); tm = 9, 10: row-reuse gather (conv_bf16r_kernel) with TM 1, 2
    switch (tm) {
        case 1: return launch_tm<1>(p, stream);
        case 2: return launch_tm<2>(p, stream);
        case 4: return launch_tm<4>(p, stream);
        case 9: return launch_tm_rowreuse<1>(p, stream);
        case 10: return launch_tm_rowreuse<2>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

// ------------------------------------------------------------------------------------------------
// weight packing for the bf16 kernel (fp32 torch layout -> bf16, K-tile major, pre-swizzled):
//   wp[cls][kt = chunk*T + tap][cout tile][row = tn*32 + c][slot][8]   (64 B per row)
//     cout = tile*64 + 2c + tn,   slot holds channel group kg = slot ^ ((row >> 2) & 3),  cin = chunk*32 + kg*8 + e
__global__ void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout,
                                 int CoutPad, int T, int transposed) {
    const size_t per_cls = (size_t)T * Cin * CoutPad;
    const size_t total = (transposed ? 8 : 1) * per_cls;
    const int n_tiles = CoutPad / 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_cls);
        size_t r = i % per_cls;
        const int e = (int)(r & 7); r >>= 3;
        const int slot = (int)(r & 3); r >>= 2;
        const int row = (int)(r & 63); r >>= 6;
        const int tile = (int)(r % n_tiles); r /= n_tiles;
        const int tap = (int)(r % T);
        const int cc = (int)(r / T);
        const int tn = row >> 5, c = row & 31;
        const int co = tile * 64 + 2 * c + tn;
        const int kg = slot ^ ((row >> 2) & 3);
        const int cin = cc * 32 + kg * 8 + e;
        float v = 0.f;
        if (co < Cout) {
            if (!transposed) {
                v = w[((size_t)co * Cin + cin) * T + tap];
            } else {
                const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
                const int td = (tap >> 2) & 1, th = (tap >> 1) & 1, tw = tap & 1;
                const int kd = 3 - rd - 2 * td, kh = 3 - rh - 2 * th, kw = 3 - rw - 2 * tw;
                v = w[((size_t)cin * Cout + co) * 64 + (kd * 4 + kh) * 4 + kw];
            }
        }
        wp[i] = __builtin_bit_cast(unsigned short, (__bf16)v);
    }
}

hipError_t launch_pack_bf16(const float* w, void* wp, int Cin, int Cout, int CoutPad, int T, int transposed,
                            hipStream_t s) {
    hipLaunchKernelGGL(pack_bf16_kernel, dim3(1024), dim3(256), 0, s, w, reinterpret_cast<unsigned short*>(wp), Cin, Cout,
                       CoutPad, T, transposed);
    return hipGetLastError();
}

}  // namespace s3r
