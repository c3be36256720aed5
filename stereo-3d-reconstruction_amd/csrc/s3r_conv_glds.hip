// Direct (im2col-free) 2D/3D convolution and transposed convolution as an implicit GEMM on the
// gfx950 fp32 matrix cores, operands staged HBM/L2 -> LDS by LDS-DMA (buffer_load ... lds), with the
// folded-BN affine + activation fused into the epilogue.
//
//   D[cout][pos] = sum_{chunk, tap, c} Wp[(chunk*T + tap)*16 + c][cout] * X[b(pos)][chunk*16 + c][in(pos) + tap]
//
//   GEMM M = Cout, N = B*Nd*Nh*Nw positions, K = Cin*T walked chunk-major / tap-minor in K tiles of
//   BK = 16 = ONE tap x 16 consecutive input channels, so the T taps of a channel chunk re-read the
//   same input patch back to back (L1/L2 hits) and everything that changes from one K tile to the
//   next is a scalar byte offset.
//
// Activations live in HBM with a one-element ZERO HALO on every spatial axis ("padded NC(D)HW",
// DESIGN.md §3): the gather needs no validity predicate at all, so a lane's VEC consecutive output
// positions of one row are VEC consecutive input dwords for every tap (stride-1 layers) and are
// fetched by ONE buffer_load_dwordx{VEC} ... lds.  Per K tile a wave issues NPB+NPA such DMAs and
// nothing else memory-side: no staging VGPRs, no ds_write, no address VALU (per-lane voffsets are
// loop-invariant; the per-tile part sits in the SGPR soffset).
//
// Matrix instruction: v_mfma_f32_32x32x2_f32 (exact fp32 FMA chain; 64 FLOP/clk/SIMD = the chip's
// 157 TFLOP/s fp32 roof — gfx950 has no TF32-style shortcut).  Operand maps (64-lane wave):
//   A: lane l holds A[i = l&31][k = l>>5]     B: lane l holds B[k = l>>5][j = l&31]
//   D: reg r of lane l is D[(r&3) + 8*(r>>2) + 4*(l>>5)][l&31]
// A wave owns (32*TM) couts x (32*TN) positions as TM x TN MFMA tiles INTERLEAVED at element
// granularity: MFMA tile (tm,tn) row i / column j is cout m0 + i*TM + tm / position n0 + j*TN + tn.
// With the LDS images As[k][cout], Bs[k][pos] (exactly what the DMA writes: lane-linear rows), lane
// (i,h) fetches its TM A values of a k-step with ONE ds_read_b{32*TM} and its TN B values with ONE
// ds_read_b{32*TN} (conflict-free: consecutive lanes, consecutive addresses), and in the epilogue
// every accumulator register row is TN consecutive positions per lane -> dwordx{TN} stores.
//
// Pipeline: two LDS buffers; the DMAs of K tile t+1 are issued before the MFMAs of tile t and waited
// for (vmcnt(0)) just before the one barrier that ends tile t.
//
// ConvTranspose3d(k=4,s=2,p=1) runs as 8 output-parity classes (blockIdx.y); each class is a
// 2x2x2-tap gather over the input grid with its own packed weight slab: output-stationary, no atomics,
// no zero-stuffed taps.
#include "s3r_kernels.h"
#include <cstdlib>
#include <type_traits>

namespace s3r {

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef float v2f __attribute__((ext_vector_type(2)));
typedef float v4f_u __attribute__((ext_vector_type(4), aligned(4)));   // dword-aligned vector access
typedef float v2f_u __attribute__((ext_vector_type(2), aligned(4)));
typedef unsigned v2u __attribute__((ext_vector_type(2)));
typedef unsigned v4u __attribute__((ext_vector_type(4)));

#define S3R_LDS_PTR(p) ((__attribute__((address_space(3))) void*)(p))

template <int N> struct FVec;
template <> struct FVec<1> { typedef float type; };
template <> struct FVec<2> { typedef v2f type; };
template <> struct FVec<4> { typedef v4f type; };

template <int N>
__device__ __forceinline__ float vget(const typename FVec<N>::type& v, int i) {
    if constexpr (N == 1) return v; else return v[i];
}

template <int BYTES>
__device__ __forceinline__ void dma_to_lds(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voffset, int soffset) {
    // the size operand must be a literal
    // the size operand must be a literal; LDS-DMA exists for 1, 2, 4, 12 and 16 bytes per lane (no 8)
    static_assert(BYTES == 16 || BYTES == 4, "LDS-DMA width");
    if constexpr (BYTES == 16) __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR(lds_dst), 16, voffset, soffset, 0, 0);
    else __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR(lds_dst), 4, voffset, soffset, 0, 0);
}

constexpr int GBK = 16;   // K tile depth: one tap x 16 input channels

// activation with the kind resolved at compile time: the epilogue is instantiated once per kind and entered
// through ONE wave-uniform switch, instead of branching on p.act for every output element (the branchy form
// was 3800 instructions and 20-25 % of a workgroup's lifetime: tools/timeline.py)
template <int ACT>
__device__ __forceinline__ float act_fn(float t) {
    if constexpr (ACT == ACT_RELU) return fmaxf(t, 0.f);
    else if constexpr (ACT == ACT_SIGMOID) return __builtin_amdgcn_rcpf(1.f + __expf(-t));
    else return t;
}

#ifdef S3R_ABLATE
// S3R_ABL=7: per-workgroup timeline stamps (s_memrealtime, 100 MHz): [cu key, start, loop start, loop end, end]
__device__ unsigned long long s3r_timeline[6 * 65536];
#endif
#ifdef S3R_ABLATE   // diagnostic builds only: S3R_ABL=1 no epilogue stores, 2 one K tile only, 3 no DMA in the loop
static int abl_mode() { static const int m = getenv("S3R_ABL") ? atoi(getenv("S3R_ABL")) : 0; return m; }
#define S3R_ABL(p, m) ((p).debug == (m))
#else
#define S3R_ABL(p, m) false
#endif

// waves per SIMD the register allocator must leave room for: 16*TM*TN accumulator registers + ~48
constexpr int min_waves(int tm, int tn) { return tm * tn >= 8 ? 2 : (tm * tn >= 4 ? 4 : 5); }

// HEAD: the fused pointwise-head epilogue (its own instantiation: the two epilogues never share registers)
// NBUF: LDS stages of the K loop.  2 = the DMAs of tile t+1 in flight during tile t (every full launch: several
// workgroups per CU cover each other's waits); 6 = five tiles in flight, for launches that leave a CU with one small
// workgroup or none (the remainder of a cut launch, tiny batches): alone, a workgroup pays the whole DMA round trip per
// K tile with two stages (0.5 us against 0.2 us of MFMAs on the 64 x 64 tile).  The K order is the same either way.
// The kernel's body as a device function of (workgroup id, workgroup count, position range): conv_glds_kernel below runs
// it over its whole grid; conv_glds_dual_kernel runs TWO tile shapes in one launch (a layer's bulk and its re-tiled
// remainder).
template <int WM, int WN, int TM, int TN, int VEC, bool HEAD, int NBUF>
__device__ __forceinline__ void conv_glds_body(const ConvParams& p, const int bid_in, const int nwg_in, const int n_begin,
                                               const int n_end, const int kz, float* smem) {
    constexpr int BM = 32 * WM * TM;
    constexpr int BN = 32 * WN * TN;
    constexpr int BK = GBK;
    static_assert(WM * WN == 4, "4 waves per workgroup");
    // ---- DMA piece geometry (one wave-instruction = 64 lanes x 4*VEC bytes (B) or x 16 bytes (A))
    constexpr int PB = 64 * VEC;                              // floats per B piece
    constexpr int NPIECE_B = BK * BN / PB;
    static_assert(NPIECE_B % 4 == 0, "B pieces divide over the 4 waves");
    constexpr int NPB = NPIECE_B / 4;                         // B pieces per wave per K tile
    constexpr bool B_WIDE = BN >= PB;                         // a piece is (part of) one k row
    constexpr int PPR = B_WIDE ? BN / PB : 1;                 // pieces per row
    constexpr int RPP = B_WIDE ? 1 : PB / BN;                 // rows per piece
    static_assert(PPR == 1 || PPR == 2 || PPR == 4, "pieces per row");
    constexpr int LPR_B = BN / VEC;                           // lanes per row (when RPP > 1)
    constexpr int NPIECE_A = BK * BM / 256;                   // A pieces of 256 floats (x4 per lane)
    constexpr int NPA = (NPIECE_A + 3) / 4;
    constexpr int RPP_A = 256 / BM;                           // rows per A piece (BM <= 256)
    static_assert(BM <= 256 && 256 % BM == 0, "BM");
    constexpr int LPR_A = BM / 4;

    float* As = smem;                     // [NBUF][BK][BM]
    float* Bs = smem + NBUF * BK * BM;    // [NBUF][BK][BN]
    static_assert(NBUF == 2 || NPIECE_A % 4 == 0, "a deeper ring counts DMAs per wave: every wave must issue as many");
    constexpr int NPD = NPA + NPB;        // DMAs per wave per K tile

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / WN, wn = wave % WN;
    const int j = lane & 31, h = lane >> 5;
#ifdef S3R_ABLATE
    unsigned long long tl0 = 0, tl1 = 0, tl2 = 0;
    if (p.debug == 7) tl0 = __builtin_amdgcn_s_memrealtime();
#endif

    // ---- XCD-aware tile order: blocks b and b+8 share an XCD (round-robin dispatch); give each XCD a
    // contiguous run of tiles so neighbouring N tiles find their shared input rows in that XCD's L2.
    // Transposed convolutions: the grid is 8 x as long and an XCD walks its tiles with the 8 output-parity classes of a
    // tile back to back (they gather from the same input tile: L2 hits instead of eight passes over the input).
    int bid = bid_in, cls = 0;                        // cls: output parity class (transposed only)
    if (p.transposed) {
        const int nwg = nwg_in >> 3;
        const int item = (bid & 7) * nwg + (bid >> 3);
        bid = item >> 3;
        cls = item & 7;
    } else {
        const int nwg = nwg_in;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int m_tiles = (p.Cout + BM - 1) / BM;       // (of THIS tile shape: a dual launch runs two)
    const int m_tile = bid % m_tiles;
    const int n_tile = bid / m_tiles;
    const int m0 = m_tile * BM, n0 = n_begin + n_tile * BN;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;

    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.T;
    // split-K: blockIdx.z owns a contiguous range of 16-channel chunks (all taps of each)
    const int chunks = (p.Cin / BK) / p.ksplit;
    const int nkt = S3R_ABL(p, 2) ? 1 : T * chunks;

    // ---- per-lane loop-invariant DMA offsets (bytes)
    int bvoff;     // B: this lane's VEC positions (+ its row inside a multi-row piece)
    {
        int col, lrow;
        if (B_WIDE) { col = (wave % PPR) * PB + lane * VEC; lrow = 0; }
        else        { col = (lane % LPR_B) * VEC;           lrow = lane / LPR_B; }
        int n = n0 + col;
        if (n >= n_end) n = n_end - VEC;                // tail tile: fetch a valid group, never stored
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        int e = b * p.Cin * p.x_cs + p.x_org + (pd * p.x_ds + ph * p.x_hs + pw) * p.stride + lrow * p.x_cs;
        if (p.transposed) e += (rd - 1) * p.x_ds + (rh - 1) * p.x_hs + (rw - 1);
        bvoff = e * 4;
    }
    const int avoff = ((lane / LPR_A) * p.CoutPad + (lane % LPR_A) * 4) * 4;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const size_t w_cls = (size_t)cls * T * p.Cin * p.CoutPad;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w + w_cls), 0, (int)((unsigned)T * (unsigned)p.Cin * (unsigned)p.CoutPad * 4u), 0x00020000);

    // wave-uniform piece placement
    const int b_row0 = B_WIDE ? wave / PPR : wave * RPP;               // first k row this wave fetches
    constexpr int B_ROW_STEP = B_WIDE ? 4 / PPR : 4 * RPP;             // k rows between its pieces
    const int b_lds0 = B_WIDE ? b_row0 * BN + (wave % PPR) * PB : wave * PB;
    constexpr int B_LDS_STEP = B_WIDE ? B_ROW_STEP * BN : 4 * PB;
    const int cs4 = p.x_cs * 4;

    // K-tile cursor of the NEXT tile to fetch (all scalar)
    int c_td = 0, c_th = 0, c_tw = 0, c_tap = 0, c_cc = kz * chunks, c_kt = kz * nkt;

    auto issue = [&](int buf) {
        // ---- weights: rows [c_kt*16, +16) of the packed slab, couts [m0, m0+BM)
        float* sa = As + buf * BK * BM;
        const int a_base = (c_kt * BK * p.CoutPad + m0) * 4;
#pragma unroll
        for (int q = 0; q < NPA; ++q) {
            const int piece = wave + 4 * q;
            if (NPIECE_A % 4 == 0 || piece < NPIECE_A)
                dma_to_lds<16>(wrsrc, sa + piece * 256, avoff, a_base + piece * RPP_A * p.CoutPad * 4);
        }
        // ---- gathered input: 16 channels of chunk c_cc at tap (c_td, c_th, c_tw)
        float* sb = Bs + buf * BK * BN + b_lds0;
        const int b_base = ((c_cc * BK + b_row0) * p.x_cs + (c_td * p.x_ds + c_th * p.x_hs + c_tw) * p.dil) * 4;
#pragma unroll
        for (int q = 0; q < NPB; ++q)
            dma_to_lds<4 * VEC>(xrsrc, sb + q * B_LDS_STEP, bvoff, b_base + q * B_ROW_STEP * cs4);
        // ---- advance the cursor: chunk-major, tap-minor
        ++c_kt;
        if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }
        if (++c_tap == T) { c_tap = 0; c_td = 0; c_th = 0; c_tw = 0; ++c_cc; }
    };

    f32x16 acc[TM][TN];
#pragma unroll
    for (int a = 0; a < TM; ++a)
#pragma unroll
        for (int b = 0; b < TN; ++b)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[a][b][r] = 0.f;

    if constexpr (NBUF == 2) {
        issue(0);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    } else {
        // tiles 0 .. NBUF-2 go out; tile 0 has landed once at most NBUF-2 tiles' DMAs are outstanding (vmcnt retires
        // in issue order)
#pragma unroll
        for (int i = 0; i < NBUF - 1; ++i)
            if (i < nkt) issue(i);
        if (nkt >= NBUF - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * NPD) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    }

#ifdef S3R_ABLATE
    if (p.debug == 7) tl1 = __builtin_amdgcn_s_memrealtime();
#endif
    const int a_off = h * BM + wm * TM * 32 + j * TM;
    const int b_off = h * BN + wn * TN * 32 + j * TN;
    typedef typename FVec<TM>::type AV;
    typedef typename FVec<TN>::type BV;

    int cur = 0;                                   // stage of tile kt (NBUF > 2; kt & 1 otherwise)
    for (int kt = 0; kt < nkt; ++kt) {
        if constexpr (NBUF == 2) {
            cur = kt & 1;
            if (kt + 1 < nkt && !S3R_ABL(p, 3)) issue(cur ^ 1);
        } else {
            // tile kt+NBUF-1 into the stage tile kt-1 was read from (every wave is past the barrier that ended it)
            if (kt + NBUF - 1 < nkt) issue(cur == 0 ? NBUF - 1 : cur - 1);
        }
        const float* a = As + cur * BK * BM + a_off;
        const float* b = Bs + cur * BK * BN + b_off;
#pragma unroll
        for (int ks = 0; ks < BK / 2; ++ks) {
            const AV av = *reinterpret_cast<const AV*>(a + ks * 2 * BM);
            const BV bv = *reinterpret_cast<const BV*>(b + ks * 2 * BN);
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int tn = 0; tn < TN; ++tn)
                    acc[tm][tn] = __builtin_amdgcn_mfma_f32_32x32x2f32(vget<TM>(av, tm), vget<TN>(bv, tn),
                                                                       acc[tm][tn], 0, 0, 0);
        }
        if constexpr (NBUF == 2) {
            if (!S3R_ABL(p, 5)) asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // this wave's DMAs of tile kt+1 have landed
            if (S3R_ABL(p, 5)) __builtin_amdgcn_s_barrier(); else
            __syncthreads();                                    // ... everyone's have, and buffer `cur` is free
        } else {
            // tile kt+1 has landed when only the NBUF-2 tiles after it are outstanding; in the drain (nothing issued
            // this iteration) wait for everything
            if (kt + NBUF - 1 < nkt) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((NBUF - 2) * NPD) : "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            cur = cur + 1 == NBUF ? 0 : cur + 1;
        }
    }

#ifdef S3R_ABLATE
    if (p.debug == 7) tl2 = __builtin_amdgcn_s_memrealtime();
    struct TlGuard {
        const ConvParams& p; unsigned long long a, b, c; int tid;
        __device__ ~TlGuard() {
            if (p.debug == 7 && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 65536) {
                const unsigned long long t_issued = __builtin_amdgcn_s_memrealtime();
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                unsigned long long* t = s3r_timeline + 6 * (size_t)blockIdx.x;
                t[5] = t_issued;
                t[0] = ((unsigned long long)(xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
                t[1] = a; t[2] = b; t[3] = c; t[4] = __builtin_amdgcn_s_memrealtime();
            }
        }
    } tl_guard{p, tl0, tl1, tl2, tid};
#endif
    // ---- split-K: raw partial sums to the scratch slab [cls][kz][cout][n] (n = GEMM position index,
    // padded to whole N tiles so no lane needs a bounds check); s3r::launch_conv_finish reduces the slabs
    // in kz order and applies the epilogue.
    if (!HEAD && p.ksplit > 1) {
        const int npad = p.n_tiles * BN;
        float* __restrict__ slab = p.part + ((size_t)(cls * p.ksplit + kz) * p.Cout) * npad + n0 + wn * TN * 32 + j * TN;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm) {
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int m = m0 + wm * TM * 32 + ((r & 3) + 8 * (r >> 2) + 4 * h) * TM + tm;
                if (m >= p.Cout) continue;
                typename FVec<TN>::type t;
                if constexpr (TN == 1) t = acc[tm][0][r];
                else {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) t[tn] = acc[tm][tn][r];
                }
                *reinterpret_cast<typename FVec<TN>::type*>(slab + (size_t)m * npad) = t;
            }
        }
        return;
    }

    // ---- per-cout epilogue constants through LDS.  vmcnt retires in issue order and counts stores, so a
    // scale/shift load issued between two stores would wait for every earlier store to LAND (measured with
    // tools/timeline.py: the epilogue took 100 us of a 470 us workgroup on v1 that way).  LDS reads wait on lgkmcnt.
    float* ep_sc = smem;            // [BM]
    float* ep_sf = smem + BM;       // [BM]
    float* ep_hw = smem + 2 * BM;   // [BM] fused-head weights (0 for padded couts)
    if (tid < BM) {
        const int m = m0 + tid;
        const int mc = p.shuf_s ? p.dSC.div(m) : m;                // (depth-to-space rows: s^nd taps share their cout's constants)
        ep_sc[tid] = (p.scale && m < p.Cout) ? p.scale[mc] : 1.f;
        ep_sf[tid] = (p.shift && m < p.Cout) ? p.shift[mc] : 0.f;
        ep_hw[tid] = (p.head_w && m < p.Cout) ? p.head_w[m] : 0.f;
    }
    __syncthreads();

    // ---- epilogue: y = act(acc * scale[cout] + shift[cout]) into the (halo-padded) NC(D)HW output;
    // a lane's TN positions are consecutive in one output row when Nw % TN == 0 (convolutions).
    const int nl = n0 + wn * TN * 32 + j * TN;          // this lane's first position
    const int ostep = p.transposed ? 2 : (p.y_step > 1 ? p.y_step : 1);
    const bool vec_ok = !p.transposed && p.y_step <= 1 && (p.Nw % TN == 0);
    int yoff[TN];
    bool yok[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) {
        const int n = nl + tn;
        yok[tn] = n < n_end;
        const int nn = yok[tn] ? n : 0;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        int e = b * p.y_bs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw) * ostep;
        if (p.transposed) e += rd * p.y_ds + rh * p.y_hs + rw;
        yoff[tn] = e;
    }
    // ---- fused pointwise head (conv -> 1x1x1 conv to ONE channel + activation, e.g. d3 -> d4 + sigmoid): the
    // workgroup's M tile holds every cout of its positions inside one wave (WM == 1), so the channel
    // reduction is 16*TM in-lane FMAs + one exchange between the two lane halves; the conv's own output
    // (the largest activation of the network) is never written to or re-read from HBM.
    if constexpr (HEAD) {
        if constexpr (WM == 1) {
            {   // (the planner fuses only ReLU / identity convs: one copy of this loop)
                const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
                float part[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) part[tn] = 0.f;
#pragma unroll
                for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                    for (int r = 0; r < 16; ++r) {
                        const int m = ((r & 3) + 8 * (r >> 2) + 4 * h) * TM + tm;
                        const float hw = ep_hw[m];                            // 0 for padded couts
#pragma unroll
                        for (int tn = 0; tn < TN; ++tn) {
                            const float t = fmaf(acc[tm][tn][r], ep_sc[m], ep_sf[m]);
                            part[tn] = fmaf(fmaxf(t, lo), hw, part[tn]);
                        }
                        if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);   // keep the LDS reads of later rows from
                    }                                                          // being hoisted (and spilled) up front
                const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    float t = part[tn] + __shfl_xor(part[tn], 32, 64);       // the other 32 couts live in lane j+32
                    t = fmaf(t, hsc, hsf);
                    if (p.head_act == ACT_RELU) t = fmaxf(t, 0.f);
                    else if (p.head_act == ACT_SIGMOID) t = __builtin_amdgcn_rcpf(1.f + __expf(-t));
                    if (h == 0 && yok[tn]) p.y[yoff[tn]] = t;
                }
            }
        }
        return;
    }

    // ---- stores.  The per-element work is branch-free: ReLU / identity is one v_max against a wave-uniform
    // floor (0 or -inf); the (rare) sigmoid layers take a separate copy of the loop chosen by ONE wave-uniform
    // branch.  The branchy form (3-way p.act test + IEEE divide per element) was 3800 instructions and 20-25 %
    // of a workgroup's lifetime (tools/timeline.py).
#if defined(S3R_ABLATE) && defined(__HIP_DEVICE_COMPILE__)
    if (S3R_ABL(p, 1)) {       // timing-only build: no stores, accumulators kept live
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int tn = 0; tn < TN; ++tn) asm volatile("" ::"v"(acc[tm][tn]));
        return;
    }
#endif
    if constexpr (!HEAD) {
        if (p.shuf_s) {
            // ---- depth-to-space store (ConvTranspose with k == stride as ONE GEMM over cout x taps rows, s3r_general.hip): row
            // m = cout * s^nd + (rd, rh, rw) of position q lands at (s qd + rd, s qh + rh, s qw + rw) of channel cout.  The rw
            // rows of a (cout, rd, rh) are consecutive r of ONE lane, the lanes of a wave consecutive qw: a wave's stores of s
            // consecutive rows fill whole lines (the per-class launches of the residue-class form wrote every s-th dword of a line
            // per launch: 2 TB/s effective)
            const int mb = m0 + wm * TM * 32 + 4 * h * TM;
            const __amdgpu_buffer_rsrc_t yr = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
            const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
            const bool sig = p.act == ACT_SIGMOID;
            const int s = p.shuf_s;
#pragma unroll
            for (int tm = 0; tm < TM; ++tm)
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const int dm = ((r & 3) + 8 * (r >> 2)) * TM + tm;
                    const int m = mb + dm;
                    if (m >= p.Cout) continue;
                    const int co = p.dSC.div(m);
                    int tap = m - co * (p.shuf_nd == 3 ? s * s * s : s * s);
                    const int t1 = p.dS1.div(tap);
                    const int rw = tap - t1 * s;
                    const int rd = p.shuf_nd == 3 ? p.dS1.div(t1) : 0;
                    const int rh = t1 - rd * s;
                    const int ro = (co * p.y_cs + rd * p.y_ds + rh * p.y_hs + rw) * 4;
                    const float sc = ep_sc[m - m0], sf = ep_sf[m - m0];
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn) {
                        const float t = fmaf(acc[tm][tn][r], sc, sf);
                        const float v = sig ? act_fn<ACT_SIGMOID>(t) : fmaxf(t, fmaf(t, p.slope, lo));
                        if (yok[tn]) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v), yr, yoff[tn] * 4 + ro, 0, 0);
                    }
                }
            return;
        }
    }
    const bool lane_vec = TN > 1 && vec_ok && yok[TN - 1];
    const int mbase = wm * TM * 32 + 4 * h * TM;
    const int mlimit = p.Cout - (m0 + mbase);                  // rows dm >= mlimit are padding
    // buffer stores: per-lane 32-bit byte offset in a VGPR (computed once), the row's cout offset in the SGPR
    // soffset -> no per-row address VALU and no 64-bit row pointers to keep (or spill) across the loop
    const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
    int yvo[TN];
#pragma unroll
    for (int tn = 0; tn < TN; ++tn) yvo[tn] = (yoff[tn] + (m0 + mbase) * p.y_cs) * 4;
    const int row_bytes = p.y_cs * 4;
    auto rows = [&](auto mode_tag) {
        constexpr int MODE = decltype(mode_tag)::value;      // 0: ReLU / identity, 1: sigmoid, 2: LeakyReLU (max(t, slope t), slope in (0, 1])
        constexpr bool SIG = MODE == 1;
        const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
        const float slope = p.slope;
#pragma unroll
        for (int tm = 0; tm < TM; ++tm)
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = ((r & 3) + 8 * (r >> 2)) * TM + tm;              // compile-time row offset
                if (dm >= mlimit) continue;
                const float sc = ep_sc[mbase + dm], sf = ep_sf[mbase + dm];
                float v[TN];
#pragma unroll
                for (int tn = 0; tn < TN; ++tn) {
                    const float t = fmaf(acc[tm][tn][r], sc, sf);
                    v[tn] = SIG ? act_fn<ACT_SIGMOID>(t) : MODE == 2 ? fmaxf(t, t * slope) : fmaxf(t, lo);
                }
                const int so = dm * row_bytes;
                if (lane_vec) {                    // dword-aligned (not 16-B aligned) vector store: legal on gfx950
                    if constexpr (TN == 2) {
                        const v2f t = {v[0], v[1]};
                        __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(v2u, t), yrsrc, yvo[0], so, 0);
                    } else if constexpr (TN == 4) {
                        const v4f t = {v[0], v[1], v[2], v[3]};
                        __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(v4u, t), yrsrc, yvo[0], so, 0);
                    }
                } else {
#pragma unroll
                    for (int tn = 0; tn < TN; ++tn)
                        if (yok[tn]) __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, v[tn]), yrsrc, yvo[tn], so, 0);
                }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
    };
    if (p.act == ACT_SIGMOID) rows(std::integral_constant<int, 1>{});
    else if (p.act == ACT_RELU && p.slope != 0.f) rows(std::integral_constant<int, 2>{});
    else rows(std::integral_constant<int, 0>{});
}

template <int WM, int WN, int TM, int TN, int VEC, bool HEAD = false, int NBUF = 2>
__global__ __launch_bounds__(256, min_waves(TM, TN) - (HEAD ? 1 : 0)) void conv_glds_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    conv_glds_body<WM, WN, TM, TN, VEC, HEAD, NBUF>(p, blockIdx.x, gridDim.x, p.n_begin, p.n_end, blockIdx.z, smem);
}

// A layer's bulk and its remainder in ONE launch (see plan_tail_cut): workgroups [0, p.big_wgs) run the layer's own tile
// over positions [n_begin, n_cut), the rest run the 64 x 64 tile over [n_cut, n_end).  As a launch of its own the remainder
// — a few dozen small workgroups — had the chip to itself for 68 / 36 / 18 us after e7 / e6 / e4 at B = 32 (a workgroup
// alone on a CU pays every DMA round trip in full); inside the bulk's launch its workgroups take the fourth slot of a
// CU beside three bulk workgroups and their MFMAs fill the pipe time the others leave.  Same K order per output whatever
// the tile, so nothing changes bitwise (tests/test_quantization_gpu.py).
template <int WM, int WN, int TM, int TN, int VEC>
__global__ __launch_bounds__(256, min_waves(TM, TN)) void conv_glds_dual_kernel(const ConvParams p) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    if ((int)blockIdx.x < p.big_wgs)
        conv_glds_body<WM, WN, TM, TN, VEC, false, 2>(p, blockIdx.x, p.big_wgs, p.n_begin, p.n_cut, 0, smem);
    else
        conv_glds_body<2, 2, 1, 1, VEC, false, 2>(p, (int)blockIdx.x - p.big_wgs, (int)gridDim.x - p.big_wgs, p.n_cut, p.n_end, 0,
                                                  smem);
}

// The residue classes of a general ConvTranspose in ONE launch (r06): class c owns workgroups [wg_begin_c, wg_begin_{c+1}) and runs the
// same body on a parameter block patched with its own grid, taps, origins and weight slab (all wave-uniform: SGPRs).  As one launch
// per class a k4 s2 2D layer was four grids of under a round each, every one paying its own ramp and tail; same K order per output,
// so nothing changes bitwise.
template <int WM, int WN, int TM, int TN>
__global__ __launch_bounds__(256, min_waves(TM, TN)) void conv_glds_tcls_kernel(const ConvParams p, const TClsTable tab) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    int c = 0;
    for (int i = 1; i < tab.n; ++i)
        if ((int)blockIdx.x >= tab.e[i].wg_begin) c = i;
    const TClsEntry e = tab.e[c];
    const int wg_end = c + 1 < tab.n ? tab.e[c + 1].wg_begin : (int)gridDim.x;
    ConvParams q = p;
    q.Nd = e.Nd; q.Nh = e.Nh; q.Nw = e.Nw;
    q.kd = e.kd; q.kh = e.kh; q.kw = e.kw; q.T = e.T;
    q.x_org = e.x_org; q.y_org = e.y_org;
    q.Ntotal = e.Ntotal;
    q.dS = e.dS; q.dHW = e.dHW; q.dW = e.dW;
    q.w = p.w + e.w_off;
    conv_glds_body<WM, WN, TM, TN, 1, false, 2>(q, (int)blockIdx.x - e.wg_begin, wg_end - e.wg_begin, 0, e.Ntotal, 0, smem);
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_tcls_cfg(const ConvParams& base, TClsTable& tab, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    const int m_tiles = (base.Cout + BM - 1) / BM;
    int wgs = 0;
    for (int i = 0; i < tab.n; ++i) {
        tab.e[i].wg_begin = wgs;
        wgs += m_tiles * ((tab.e[i].Ntotal + BN - 1) / BN);
    }
    const size_t lds = (size_t)2 * GBK * (BM + BN) * sizeof(float);
    if (lds > 48 * 1024) {
        static LdsAttr lds_attr;
        const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_glds_tcls_kernel<WM, WN, TM, TN>), (int)lds);
        if (attr != hipSuccess) return attr;
    }
    hipLaunchKernelGGL((conv_glds_tcls_kernel<WM, WN, TM, TN>), dim3(wgs), dim3(256), lds, stream, base, tab);
    return hipGetLastError();
}

hipError_t launch_conv_tcls(const ConvParams& base, TClsTable tab, hipStream_t stream) {
    if (tab.n < 1 || tab.n > kTClsMax || base.Cin % GBK != 0 || base.ksplit != 1 || base.transposed || base.head_w) return hipErrorInvalidValue;
    long w128 = 0;                               // workgroups of the 64 x 128 tile
    for (int i = 0; i < tab.n; ++i) w128 += (long)((base.Cout + 63) / 64) * ((tab.e[i].Ntotal + 127) / 128);
    if (w128 >= 1024) return launch_tcls_cfg<1, 4, 2, 1>(base, tab, stream);      // cfg 7
    return launch_tcls_cfg<2, 2, 1, 1>(base, tab, stream);                       // cfg 3: 64 x 64
}

// ------------------------------------------------------------------------------------------------
// split-K finish: y = act(scale * sum_kz slab[cls][kz][m][n] + shift), summed in kz order (deterministic),
// scattered to the (halo-padded) output position of n.  One thread per 4 consecutive n of one cout.
__global__ __launch_bounds__(256) void conv_finish_kernel(const ConvParams p, int npad) {
    const int S = p.Nd * p.Nh * p.Nw;
    const int cls = blockIdx.y;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int nq = npad >> 2;
    const long long total = (long long)p.Cout * nq;
    const int ostep = p.transposed ? 2 : (p.y_step > 1 ? p.y_step : 1);
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / nq);
        const int n = (int)(i - (long long)m * nq) * 4;
        if (n >= p.Ntotal) continue;
        const float* __restrict__ src = p.part + ((size_t)cls * p.ksplit * p.Cout + m) * npad + n;
        const size_t zs = (size_t)p.Cout * npad;
        // (the slabs are added in kz order either way; four loads in flight instead of a chain of dependent round trips)
        v4f sum;
        int z;
        if (p.ksplit >= 4) {
            const v4f t0 = *reinterpret_cast<const v4f*>(src);
            const v4f t1 = *reinterpret_cast<const v4f*>(src + zs);
            const v4f t2 = *reinterpret_cast<const v4f*>(src + 2 * zs);
            const v4f t3 = *reinterpret_cast<const v4f*>(src + 3 * zs);
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] = ((t0[k] + t1[k]) + t2[k]) + t3[k];
            z = 4;
        } else {
            sum = *reinterpret_cast<const v4f*>(src);
            z = 1;
        }
        for (; z + 4 <= p.ksplit; z += 4) {
            const v4f t0 = *reinterpret_cast<const v4f*>(src + (size_t)z * zs);
            const v4f t1 = *reinterpret_cast<const v4f*>(src + (size_t)(z + 1) * zs);
            const v4f t2 = *reinterpret_cast<const v4f*>(src + (size_t)(z + 2) * zs);
            const v4f t3 = *reinterpret_cast<const v4f*>(src + (size_t)(z + 3) * zs);
#pragma unroll
            for (int k = 0; k < 4; ++k) sum[k] = (((sum[k] + t0[k]) + t1[k]) + t2[k]) + t3[k];
        }
        for (; z < p.ksplit; ++z) {
            const v4f t = *reinterpret_cast<const v4f*>(src + (size_t)z * zs);
            sum[0] += t[0]; sum[1] += t[1]; sum[2] += t[2]; sum[3] += t[3];
        }
        const float sc = p.scale ? p.scale[m] : 1.f, sf = p.shift ? p.shift[m] : 0.f;
        float* __restrict__ yrow = p.y + (size_t)m * p.y_cs;
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const int nn = n + k;
            if (nn >= p.Ntotal) break;
            const int b = p.dS.div(nn);
            int rem = nn - b * S;
            const int pd = p.dHW.div(rem);
            rem -= pd * p.Nh * p.Nw;
            const int ph = p.dW.div(rem);
            const int pw = rem - ph * p.Nw;
            int e = b * p.Cout * p.y_cs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw) * ostep;
            if (p.transposed) e += rd * p.y_ds + rh * p.y_hs + rw;
            float t = fmaf(sum[k], sc, sf);
            if (p.act == ACT_RELU) t = fmaxf(t, fmaf(t, p.slope, 0.f));
            else if (p.act == ACT_SIGMOID) t = 1.f / (1.f + __expf(-t));
            yrow[e] = t;
        }
    }
}

static thread_local int g_launch_count = 0;          // kernels launched by the last launch_conv_mfma (split-K finish included)
int conv_last_launch_count() { return g_launch_count; }

hipError_t launch_conv_finish(const ConvParams& p, int npad, hipStream_t stream) {
    g_launch_count += 1;
    const long long total = (long long)p.Cout * (npad >> 2);
    const long long blocks = (total + 255) / 256;
    // (aux pass: reads the ksplit partial slabs, writes the outputs)
    AuxScope aux(stream, 4.0 * (double)p.Cout * (p.transposed ? 8 : 1) * ((double)p.ksplit * npad + (double)p.Ntotal));
    hipLaunchKernelGGL(conv_finish_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), p.transposed ? 8 : 1), dim3(256),
                       0, stream, p, npad);
    return hipGetLastError();
}

// floats of split-K scratch a launch with tile `cfg` needs
int64_t conv_scratch_elems(const ConvParams& p, int cfg) {
    if (p.ksplit <= 1) return 0;
    int bm, bn;
    conv_tile_dims(cfg, &bm, &bn);
    const int64_t npad = (int64_t)((p.Ntotal + bn - 1) / bn) * bn;
    return (int64_t)(p.transposed ? 8 : 1) * p.ksplit * p.Cout * npad;
}

// ------------------------------------------------------------------------------------------------
// tile configurations:  id -> (WM, WN, TM, TN); the gather width VEC is chosen per layer
//   id  WM WN TM TN   BM x BN
//    0   2  2  2  2  128 x 128   Cout >= 128
//    1   1  4  2  2   64 x 256   Cout == 64
//    2   1  4  1  2   32 x 256   Cout <= 32
//    3   2  2  1  1   64 x  64   small N (more workgroups)
//    4   2  2  2  4  128 x 256   Cout >= 128, large N
//    5   1  4  2  4   64 x 512   Cout == 64, large N
//    6   2  2  2  1  128 x  64   small N, Cout >= 128
//    7   1  4  2  1   64 x 128
static const int kTileDims[][2] = {{128, 128}, {64, 256}, {32, 256}, {64, 64}, {128, 256}, {64, 512}, {128, 64}, {64, 128}};
constexpr int kNumTiles = 8;

void conv_tile_dims(int cfg, int* bm, int* bn) {
    *bm = kTileDims[cfg][0];
    *bn = kTileDims[cfg][1];
}
int conv_num_tiles() { return kNumTiles; }

// ---- measured configurations -------------------------------------------------------------------
// (tile, split-K) per layer shape, picked by Stereo2Voxel.autotune() on MI355X at the BASELINE batch
// (32 pairs; the per-layer tables it printed are in profiles/r01_bench_stderr.txt).  Keyed by PER-SAMPLE geometry only, so the split-K factor —
// which changes a sample's summation order — never depends on the batch a sample is computed in.
// Shapes not listed fall back to the rules below.
struct Tuned { int cin, cout, T, stride, S, transposed, tile, ksplit; };
static const Tuned kTuned[] = {
    //  cin cout   T  s       S  tr  tile ks
    {32,  64,  9, 1, 12544, 0, 7, 1},   // e2  112^2   (r03 re-tune, three A/B pairs of bench.py: e2 1 -> 7, e4 2 -> 4, d1 2 -> 7: -0.5 % per step)
    {64,  64,  9, 2,  3136, 0, 1, 1},   // e3  -> 56^2  (with the remainder inside the bulk's launch, r03: e3 3 -> 1, v2 / v3 3 -> 7: -0.7 %)
    {64, 128,  9, 1,  3136, 0, 4, 1},   // e4
    {128, 128, 9, 2,   784, 0, 7, 1},   // e5  -> 28^2  (r05 re-sweep: 64 x 128, 784 workgroups, 0.137 -> 0.130 alone)
    {128, 256, 9, 1,   784, 0, 1, 1},   // e6  (64x256 bulk + 64x64 remainder, see plan_tail_cut)
    {256, 256, 9, 1,   784, 0, 1, 1},   // e7
    {256, 32,  1, 1,   784, 0, 2, 1},   // e8
    {64,  64, 27, 1, 21952, 0, 1, 1},   // v1  28^3
    {64, 128, 27, 2,  2744, 0, 7, 1},   // v2  -> 14^3
    {128, 128, 27, 1, 2744, 0, 7, 1},   // v3
    {128, 256, 27, 2,  343, 0, 3, 4},   // v4  -> 7^3
    {256, 256, 27, 1,  343, 0, 3, 4},   // v5
    {256, 512, 64, 1,   64, 0, 7, 8},   // v6  k4 valid -> 4^3
    {512, 256,  8, 1,   64, 1, 7, 1},   // d1  4^3 -> 8^3
    {256, 128,  8, 1,  512, 1, 1, 1},   // d2
    {128,  64,  8, 1, 4096, 1, 1, 1},   // d3
};

static const Tuned* find_tuned(const ConvParams& p) {
    const int S = p.Nd * p.Nh * p.Nw;
    for (const Tuned& t : kTuned)
        if (t.cin == p.Cin && t.cout == p.Cout && t.T == p.T && t.stride == p.stride && t.S == S &&
            t.transposed == p.transposed)
            return &t;
    return nullptr;
}

// Split-K factor.  Decided from the layer's PER-SAMPLE geometry at a nominal batch of 32 samples, never
// from the actual batch (see kTuned).  Splits deep-K layers whose 64x64-tile grid would leave the 256
// CUs short of workgroups (tools/layer_bench.py: v6 79 -> 118 TFLOP/s at 8 splits, v5 110 -> 121 at 4;
// layers with fewer than ~100 K tiles per split lose more to the extra prologues than they gain).
int conv_pick_ksplit(const ConvParams& p, int /*tile_cfg*/) {
    if (const Tuned* t = find_tuned(p)) return t->ksplit;
    const int chunks = p.Cin / GBK;
    const long S = (long)p.Nd * p.Nh * p.Nw;
    const long wg_nom = ((p.Cout + 63) / 64) * ((32 * S + 63) / 64) * (p.transposed ? 8 : 1);
    int ks = 1;
    while (wg_nom * ks < 2048 && chunks % (2 * ks) == 0 && (chunks / (2 * ks)) * p.T >= 100) ks *= 2;
    return ks;
}

// Tile shape by workgroup count (same tool): narrow-M tiles win on every layer of this network — finer
// work units balance the 256 CUs better than 128-tall tiles save in operand traffic — and the position
// extent shrinks (256 -> 128 -> 64) as the layer offers fewer workgroups.
int conv_pick_tile(const ConvParams& p) {
    const int classes = (p.transposed ? 8 : 1) * (p.ksplit > 1 ? p.ksplit : 1);
    auto wgs = [&](int cfg) {
        const long bm = kTileDims[cfg][0], bn = kTileDims[cfg][1];
        return ((p.Cout + bm - 1) / bm) * ((p.Ntotal + bn - 1) / bn) * classes;
    };
    if (const Tuned* t = find_tuned(p))
        if (wgs(t->tile) >= 512) return t->tile;      // tuned at batch 32; tiny batches use the rules
    // (r06 shape sweep: a 32 x 256 tile over a few thousand positions is a few dozen workgroups — conv2d 64 -> 32, k7 s2 over 16^2 at
    // B = 256 ran 64 workgroups at 0.10 of its roof; below half a round of the chip the 64 x 64 tile's 4x the workgroups win although
    // half of its cout rows are padding.  Not earlier: e8 of this network (196 workgroups at B = 32, HBM-bound) measured 22 us with the
    // 32 x 256 tile and 25 us with 64 x 64.  Tiles never change a bit: the K order is fixed)
    if (p.Cout <= 32) return wgs(2) >= 128 ? 2 : 3;
    if (p.Cout <= 64) return wgs(1) >= 2000 ? 1 : (wgs(7) >= 2000 ? 7 : 3);
    if (wgs(2) >= 1500) return 2;
    return wgs(7) >= 1500 ? 7 : 3;
}

// widest gather the layer geometry allows: VEC consecutive positions of a row must be VEC consecutive
// input dwords (stride 1) and must not straddle rows (Nw % VEC == 0)
int conv_pick_vec(const ConvParams& p) {
    if (p.stride != 1) return 1;
    return (p.Nw % 4 == 0) ? 4 : 1;   // (there is no 8-byte LDS-DMA)
}

template <int WM, int WN, int TM, int TN, int VEC, bool HEAD, int NBUF = 2>
static hipError_t launch_cfg(ConvParams p, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    if constexpr (GBK * BN % (256 * VEC) != 0 || BN > 256 * VEC) {
        return hipErrorInvalidValue;   // tile too narrow / too wide for this gather width
    } else {
        p.m_tiles = (p.Cout + BM - 1) / BM;
        p.n_tiles = (p.n_end - p.n_begin + BN - 1) / BN;
        const size_t lds = (size_t)NBUF * GBK * (BM + BN) * sizeof(float);
        if (lds > 48 * 1024) {   // above the default dynamic-LDS limit: raise it once per instantiation and device
            static LdsAttr lds_attr;
            const hipError_t attr = lds_attr.ensure(
                reinterpret_cast<const void*>(&conv_glds_kernel<WM, WN, TM, TN, VEC, HEAD, NBUF>), (int)lds);
            if (attr != hipSuccess) return attr;
        }
        dim3 grid(p.m_tiles * p.n_tiles * (p.transposed ? 8 : 1), 1, p.ksplit);      // (class inside blockIdx.x)
        hipLaunchKernelGGL((conv_glds_kernel<WM, WN, TM, TN, VEC, HEAD, NBUF>), grid, dim3(256), lds, stream, p);
        hipError_t e = hipGetLastError();
        if (e == hipSuccess && p.ksplit > 1) e = launch_conv_finish(p, p.n_tiles * BN, stream);
        return e;
    }
}

template <int WM, int WN, int TM, int TN, int VEC>
static hipError_t launch_dual_cfg(ConvParams p, int n_cut, hipStream_t stream) {
    constexpr int BM = 32 * WM * TM, BN = 32 * WN * TN;
    if constexpr (GBK * BN % (256 * VEC) != 0 || BN > 256 * VEC || GBK * 64 % (256 * VEC) != 0) {
        return hipErrorInvalidValue;
    } else {
        p.n_cut = n_cut;
        p.big_wgs = ((p.Cout + BM - 1) / BM) * ((n_cut - p.n_begin + BN - 1) / BN);
        const int small = ((p.Cout + 63) / 64) * ((p.n_end - n_cut + 63) / 64);
        p.m_tiles = 0; p.n_tiles = 0;                  // (unused: the body derives its own; no split-K here)
        const size_t lds = (size_t)2 * GBK * (BM + BN) * sizeof(float);      // >= the 64 x 64 tile's
        if (lds > 48 * 1024) {
            static LdsAttr lds_attr;
            const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_glds_dual_kernel<WM, WN, TM, TN, VEC>), (int)lds);
            if (attr != hipSuccess) return attr;
        }
        hipLaunchKernelGGL((conv_glds_dual_kernel<WM, WN, TM, TN, VEC>), dim3(p.big_wgs + small), dim3(256), lds, stream, p);
        return hipGetLastError();
    }
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_dual_vec(const ConvParams& p, int n_cut, int vec, hipStream_t stream) {
    switch (vec) {
        case 4: return launch_dual_cfg<WM, WN, TM, TN, 4>(p, n_cut, stream);
        case 1: return launch_dual_cfg<WM, WN, TM, TN, 1>(p, n_cut, stream);
        default: return hipErrorInvalidValue;
    }
}

// bulk tile `cfg` over [0, n_cut) + 64 x 64 tiles over the rest, one launch
static hipError_t launch_dual(const ConvParams& p, int cfg, int n_cut, int vec, hipStream_t stream) {
    switch (cfg) {
        case 0: return launch_dual_vec<2, 2, 2, 2>(p, n_cut, vec, stream);
        case 1: return launch_dual_vec<1, 4, 2, 2>(p, n_cut, vec, stream);
        case 4: return launch_dual_vec<2, 2, 2, 4>(p, n_cut, vec, stream);
        case 7: return launch_dual_vec<1, 4, 2, 1>(p, n_cut, vec, stream);
        default: return hipErrorNotSupported;          // (the caller then issues the two launches)
    }
}

// the 64 x 64 tile on a launch that leaves CUs with one workgroup or none: six LDS stages (see the kernel's NBUF note)
static bool sparse_launch(const ConvParams& p, int bm, int bn) {
    static const bool off = getenv("S3R_DEEP_RING") && atoi(getenv("S3R_DEEP_RING")) == 0;      // A/B switch
    const long w = (long)((p.Cout + bm - 1) / bm) * ((p.n_end - p.n_begin + bn - 1) / bn) * (p.transposed ? 8 : 1) * p.ksplit;
    return !off && w <= 384;
}

template <int WM, int WN, int TM, int TN>
static hipError_t launch_vec(const ConvParams& p, int vec, hipStream_t stream) {
    if constexpr (WM == 1 && TM * TN <= 4) {      // the head epilogue exists for the one-wave-tall tiles it can use
        if (p.head_w) {
            switch (vec) {
                case 4: return launch_cfg<WM, WN, TM, TN, 4, true>(p, stream);
                case 1: return launch_cfg<WM, WN, TM, TN, 1, true>(p, stream);
                default: return hipErrorInvalidValue;
            }
        }
    } else if (p.head_w) {
        return hipErrorInvalidValue;
    }
    if constexpr (WM == 2 && WN == 2 && TM == 1 && TN == 1) {
        if (sparse_launch(p, 64, 64)) {
            switch (vec) {
                case 4: return launch_cfg<WM, WN, TM, TN, 4, false, 6>(p, stream);
                case 1: return launch_cfg<WM, WN, TM, TN, 1, false, 6>(p, stream);
                default: return hipErrorInvalidValue;
            }
        }
    }
    switch (vec) {
        case 4: return launch_cfg<WM, WN, TM, TN, 4, false>(p, stream);
        case 1: return launch_cfg<WM, WN, TM, TN, 1, false>(p, stream);
        default: return hipErrorInvalidValue;
    }
}

static hipError_t launch_tile(const ConvParams& p, int cfg, int vec, hipStream_t stream) {
    g_launch_count += 1;
    switch (cfg) {
        case 0: return launch_vec<2, 2, 2, 2>(p, vec, stream);
        case 1: return launch_vec<1, 4, 2, 2>(p, vec, stream);
        case 2: return launch_vec<1, 4, 1, 2>(p, vec, stream);
        case 3: return launch_vec<2, 2, 1, 1>(p, vec, stream);
        case 4: return launch_vec<2, 2, 2, 4>(p, vec, stream);
        case 5: return launch_vec<1, 4, 2, 4>(p, vec, stream);
        case 6: return launch_vec<2, 2, 2, 1>(p, vec, stream);
        case 7: return launch_vec<1, 4, 2, 1>(p, vec, stream);
        default: return hipErrorInvalidValue;
    }
}

// Workgroup-count quantisation.  The four waves of a workgroup sit on the four SIMDs of a CU, so a CU that
// hosts n workgroups takes n workgroup-times: a launch of W equal workgroups costs ceil(W / 256) of them
// (measured: tools/quant_exp.py — e7 with 128x128 tiles runs 135 TFLOP/s at W = 1018 and 104 at W = 1030).
// When W is a little over a multiple of 256 the layer is cut in two along the position axis: the bulk —
// a whole number of 256-workgroup rounds — keeps its tile, and the remainder is re-tiled 64 x 64 so that it
// spreads over all CUs in a fraction of a round.  Any tile shape produces the same bits (the K order is
// fixed), so the cut never changes a result.  Both parts go out as ONE launch where the dual kernel is built for the
// bulk's tile (conv_glds_dual_kernel), as two otherwise.
static bool plan_tail_cut(const ConvParams& p, int cfg, int* n_cut) {
    if (p.ksplit != 1 || p.Cout <= 32 || cfg == 3) return false;
    const int classes = p.transposed ? 8 : 1;
    const long bm = kTileDims[cfg][0], bn = kTileDims[cfg][1];
    const long m_tiles = (p.Cout + bm - 1) / bm, n_tiles = (p.Ntotal + bn - 1) / bn;
    const long W = m_tiles * n_tiles * classes;
    const long CUS = cu_count();
    const long rounds = W / CUS, rem = W % CUS;
    // where both parts share ONE launch (launch_dual) the remainder's small workgroups run beside the bulk's last round
    // instead of after it: its cost is its share of a round, not a round of its own, and the cut pays for longer launches
    const bool dual = !p.transposed && (cfg == 0 || cfg == 1 || cfg == 4 || cfg == 7) && getenv("S3R_NO_DUAL") == nullptr;
    if (rounds < 1 || rounds >= (dual ? 64 : 16) || rem == 0) return false;
    const long n_main = (rounds * CUS) / (m_tiles * classes);          // whole N tiles in the bulk
    if (n_main < 1 || n_main >= n_tiles) return false;
    const long pos_tail = p.Ntotal - n_main * bn;
    const long W_tail = ((pos_tail + 63) / 64) * ((p.Cout + 63) / 64) * classes;
    const double before = (double)((W + CUS - 1) / CUS);
    const double bulk = (double)((n_main * m_tiles * classes + CUS - 1) / CUS);
    const double small = (64.0 * 64.0) / (double)(bm * bn);
    const double after = dual ? bulk + (double)W_tail / (double)CUS * small / 0.85
                              : bulk + (double)((W_tail + CUS - 1) / CUS) * small / 0.8;
    if (after > (dual ? 0.99 : 0.97) * before) return false;
    *n_cut = (int)(n_main * bn);
    return true;
}

// code = tile_cfg (15 = heuristic) + 16 * forced_vec (0 = widest legal)
hipError_t launch_conv_mfma(const ConvParams& pin, int code, hipStream_t stream) {
    ConvParams p = pin;
    g_launch_count = 0;
    int cfg = code & 15;
    int vec = code >> 4;
    if (cfg == 15) cfg = conv_pick_tile(p);
    if (cfg >= kNumTiles) return hipErrorInvalidValue;
    const int vmax = conv_pick_vec(p);
    if (vec != 1 && vec != 4) vec = vmax;
    if (vec > vmax) vec = vmax;
    if (p.Cin % GBK != 0 || p.Ntotal % vec != 0) return hipErrorInvalidValue;
    if (p.ksplit < 1 || (p.Cin / GBK) % p.ksplit != 0 || (p.ksplit > 1 && !p.part)) return hipErrorInvalidValue;
    if (p.head_w) {   // fused head: one wave must hold all couts of its positions (WM == 1, BM >= Cout), no split-K
        int bm, bn;
        conv_tile_dims(cfg, &bm, &bn);
        const bool wm1 = cfg == 1 || cfg == 2 || cfg == 7;
        if (!wm1 || bm < p.Cout || p.ksplit != 1 || p.act == ACT_SIGMOID) return hipErrorInvalidValue;
    }
#ifdef S3R_ABLATE
    p.debug = abl_mode();
#endif
    p.n_begin = 0;
    p.n_end = p.Ntotal;
    int n_cut = 0;
    static const bool no_cut = getenv("S3R_NO_TAIL_CUT") != nullptr;      // tuning / A-B switch
    if (!p.head_w && !no_cut && plan_tail_cut(p, cfg, &n_cut)) {
        static const bool no_dual = getenv("S3R_NO_DUAL") != nullptr;         // A/B switch: bulk and remainder as two launches
        if (!no_dual && !p.transposed && p.Cout > 32) {
            g_launch_count += 1;
            const hipError_t ed = launch_dual(p, cfg, n_cut, vec, stream);
            if (ed != hipErrorNotSupported) return ed;
            g_launch_count -= 1;
        }
        p.n_end = n_cut;
        hipError_t e = launch_tile(p, cfg, vec, stream);
        if (e != hipSuccess) return e;
        p.n_begin = n_cut;
        p.n_end = p.Ntotal;
        return launch_tile(p, 3, vec, stream);
    }
    return launch_tile(p, cfg, vec, stream);
}

// ------------------------------------------------------------------------------------------------
// weight packing (device-side, once per parameter update): K rows of CoutPad couts in the kernel's
// K order (chunk-major, tap-minor):   row = (chunk*T + tap)*16 + c,  cin = chunk*16 + c
//   conv   : w[Cout][Cin][T]        -> wp[row][cout]
//   deconv : w[Cin][Cout][4][4][4]  -> wp[cls][row][cout], tap = (td,th,tw) in 2x2x2; along an axis with
//            output parity r, tap t reads input offset t-1+r and kernel index 3 - r - 2t
__global__ void pack_glds_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int CoutPad,
                                 int T, int transposed) {
    const size_t per_cls = (size_t)T * Cin * CoutPad;
    const size_t total = (transposed ? 8 : 1) * per_cls;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        const int cls = (int)(i / per_cls);
        size_t r = i % per_cls;
        const int co = (int)(r % CoutPad);
        r /= CoutPad;
        const int c = (int)(r & 15);
        r >>= 4;
        const int tap = (int)(r % T);
        const int cc = (int)(r / T);
        const int cin = cc * 16 + c;
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
        wp[i] = v;
    }
}

hipError_t launch_pack_conv(const float* w, float* wp, int Cin, int Cout, int CoutPad, int T, int transposed,
                            hipStream_t s) {
    hipLaunchKernelGGL(pack_glds_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, CoutPad, T, transposed);
    return hipGetLastError();
}

#ifdef S3R_ABLATE
}  // namespace s3r
extern "C" int s3r_debug_read_timeline(unsigned long long* out, int nblocks) {
    if (nblocks > 65536) nblocks = 65536;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(s3r::s3r_timeline), sizeof(unsigned long long) * 6 * (size_t)nblocks) == hipSuccess ? nblocks : -1;
}
namespace s3r {
#endif
}  // namespace s3r
