// bf16 MFMA path (BASELINE.json configs[2]): 2D/3D convolution and transposed convolution as an implicit
// GEMM on the bf16 matrix cores, bf16 activations in CHANNELS-LAST layout with a zero halo, fp32
// accumulation, folded-BN affine + activation + bf16 rounding fused into the epilogue.
//
//   D[pos][cout] = sum_{chunk, tap, c} X[b(pos)][in(pos) + tap][chunk*32 + c] * Wp[(chunk*T + tap)][cout][c]
//
//   GEMM M = B*Nd*Nh*Nw positions (A operand: gathered activations), N = Cout (B operand: weights),
//   K = Cin*T in K tiles of ONE tap x 32 channels = 64 contiguous bytes per position.
//
// Why channels-last here when the fp32 path is NCHW: the bf16 MFMA takes 8 consecutive k per lane, i.e. 16
// contiguous BYTES of one position's channel vector — one ds_read_b128 — whereas the fp32 MFMA takes one k
// per lane and wants position-contiguous rows.  The output falls out channels-last as well.
//
// Two matrix instructions, one code path (template parameter SH; MI355X_MICROARCH.md "DVFS give-back" item 7 and
// cdna_hip_programming.md rule 28: the chip can hold a higher clock on one shape than on the other, so both are
// built at the same per-wave output tile and the faster BY WALL ON RANDOM DATA is kept):
//   SH = 32: v_mfma_f32_32x32x16_bf16  lane = (row li = lane & 31, k group lk = lane >> 5), 16 acc registers
//   SH = 16: v_mfma_f32_16x16x32_bf16  lane = (row li = lane & 15, k group lk = lane >> 4),  4 acc registers
// A wave owns 32*TM positions x 64*NH couts as MFMA tiles of MT x MT (MT = SH).  The MFMA is issued with the
// WEIGHTS as its A operand, so D = [cout][position]: a lane owns ONE position per position tile, and the pack
// kernel permutes the weight rows so that the couts a lane holds are CONSECUTIVE (16 per 32 couts for SH = 32, 16 per
// 64 couts for SH = 16): 32 contiguous bytes of the channels-last output per group, one output offset per position,
// and the fused head's reduction over the couts stays inside the lane (+ one or two cross-lane exchanges).
//
// LDS images (per K tile): As[pos][slots x 16 B], Bs[cout row][slots x 16 B], slot = kgroup ^ swz(row).  A plain
// image makes every ds_read_b128 lane group hit the same 16-byte column of several rows; the XOR spreads them
// (swz below: conflict-free for BOTH instruction shapes' lane -> (row, k group) maps).  Both operands arrive by
// LDS-DMA (16 B per lane, lane-linear destination), so the swizzle is applied on the SOURCE side: a lane
// fetches channel group slot ^ swz(row) of its position; weights are stored pre-swizzled by the pack kernel.
//
// Workgroup: 4 waves along M.  Per-position input / output offsets are decoded once per workgroup into LDS.
#include "s3r_kernels.h"
#include <cstdio>
#include <cstdlib>
#include <type_traits>

#ifndef S3R_PRIO_EDGE
#define S3R_PRIO_EDGE 3               // wave priority of the prologue / epilogue phases (the K loops run at 0)
#endif
#ifndef S3R_BF16_MFMA_DEFAULT
#define S3R_BF16_MFMA_DEFAULT 32      // the matrix instruction used when S3R_BF16_MFMA is not set (see conv_bf16_shape)
#endif

namespace s3r {

#ifdef S3R_ABLATE
// S3R_ABL=7: per-workgroup timeline stamps of the plane kernel (s_memrealtime, 100 MHz):
// [cu key, entry, tables done, first image landed, loop end, epilogue issued, stores landed]
__device__ unsigned long long s3r_timeline_h[8 * 65536];
#endif
#ifdef S3R_ABLATE   // diagnostic builds only: S3R_ABL=1 no epilogue stores, 2 one K tile only, 3 no DMA in the loop
static int abl_mode_h() { static const int m = getenv("S3R_ABL") ? atoi(getenv("S3R_ABL")) : 0; return m; }
#define S3R_ABLH(p, m) ((p).debug == (m))
#else
#define S3R_ABLH(p, m) false
#endif

typedef float f32x16 __attribute__((ext_vector_type(16)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
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

// ---- the two matrix instructions behind one interface
template <int SH> struct Mf;
template <> struct Mf<32> {
    static constexpr int MT = 32;      // rows / columns of one MFMA tile
    static constexpr int KS = 16;      // k per instruction
    static constexpr int NK = 2;       // k groups of 8 per instruction = lane groups
    typedef f32x16 acc_t;
    static constexpr int NACC = 16;
};
template <> struct Mf<16> {
    static constexpr int MT = 16;
    static constexpr int KS = 32;
    static constexpr int NK = 4;
    typedef f32x4 acc_t;
    static constexpr int NACC = 4;
};
template <int SH>
__device__ __forceinline__ typename Mf<SH>::acc_t mma(const bf16x8& w, const bf16x8& a, const typename Mf<SH>::acc_t& c) {
    if constexpr (SH == 32) return __builtin_amdgcn_mfma_f32_32x32x16_bf16(w, a, c, 0, 0, 0);
    else return __builtin_amdgcn_mfma_f32_16x16x32_bf16(w, a, c, 0, 0, 0);
}
// The 16 CONSECUTIVE couts a lane holds in one group: element r (0..15) of group gi of position tile `acc[..]`.
//   SH = 32: acc[gi = 32-cout tile][r]                 couts  gi*32 + 16*lk + r      (2 groups per 64 couts)
//   SH = 16: acc[r >> 2 = 16-cout tile][r & 3]         couts          16*lk + r      (1 group  per 64 couts)
template <int SH> constexpr int groups_per_64() { return SH == 32 ? 2 : 1; }
template <int SH> __device__ __forceinline__ int group_cout(int gi, int lk) { return SH == 32 ? gi * 32 + 16 * lk : 16 * lk; }
template <int SH>
__device__ __forceinline__ float acc_at(const typename Mf<SH>::acc_t (&t)[64 / Mf<SH>::MT], int gi, int r) {
    if constexpr (SH == 32) return t[gi][r]; else return t[r >> 2][r & 3];
}

// which instruction the library uses: S3R_BF16_MFMA = 16 | 32 (read per call: the A/B tools flip it in-process;
// the weights must be packed under the same value — the Python module keys its pack cache on it)
int conv_bf16_shape() {
    const char* e = getenv("S3R_BF16_MFMA");
    if (e && atoi(e) == 16) return 16;
    if (e && atoi(e) == 32) return 32;
    return S3R_BF16_MFMA_DEFAULT;
}

constexpr int HKC = 32;    // channels per K tile
constexpr int HBN = 64;    // couts per workgroup

// element offset of output position (b, pd, ph, pw) [of parity class (rd, rh, rw) for a transposed conv] in p.y
__device__ __forceinline__ int out_offset(const ConvParamsH& p, int b, int pd, int ph, int pw, int rd, int rh, int rw) {
    const int ostep = p.transposed ? 2 : 1;
    int ye = b * p.y_bs + p.y_org + (pd * p.y_ds + ph * p.y_hs + pw * p.y_ws) * ostep;
    if (p.transposed) ye += rd * p.y_ds + rh * p.y_hs + rw * p.y_ws;
    return ye;
}

// Epilogue shared by the bf16 kernels.  D is [cout][position]: lane (li, lk) owns position row
// wave*32*TM + pt*MT + li of the tile for each position tile pt, and per 64-cout block the 16-cout groups of
// group_cout() (the pack kernel permutes the weight rows to make them consecutive).  A lane therefore stores 32
// contiguous bytes per group, needs one output offset per position, and the fused head's reduction over the couts
// stays inside the lane (then one exchange per lane-group bit).
// ep = LDS [3][64]: folded-BN scale, shift and head weight of the workgroup's 64 couts; yoff[row] = the
// position's output element offset (-1: past the end).
constexpr int EP_BYTES = 3 * 64 * 4;
constexpr int ST_ROW = 144;     // staging row: 128 payload bytes + 16 (rows 9 sixteen-byte slots apart: conflict-free b128 writes)

// The per-cout constants are LOADED first thing in the kernel (their latency hides behind the first operand
// fetch) and only written to LDS after the main loop: nothing before the epilogue waits for them.
struct EpRegs { float sc, sf, hw; };

__device__ __forceinline__ EpRegs load_ep(const ConvParamsH& p, int tid, int n0, int nco = 64) {
    EpRegs e = {1.f, 0.f, 0.f};
    if (tid < nco) {
        const int co = n0 + tid;
        const bool ok = co < p.Cout;
        e.sc = (ok && p.scale) ? p.scale[co] : 1.f;
        e.sf = (ok && p.shift) ? p.shift[co] : 0.f;
        e.hw = (ok && p.head_w) ? p.head_w[co] : 0.f;
    }
    return e;
}

__device__ __forceinline__ void store_ep(const EpRegs& e, float* ep, int tid, int nco = 64) {
    if (tid < nco) {                               // [64-cout half][scale | shift | head weight][64]
        float* q = ep + (tid >> 6) * 192 + (tid & 63);
        q[0] = e.sc;
        q[64] = e.sf;
        q[128] = e.hw;
    }
    __syncthreads();
}

// acc: the wave's accumulators of ONE 64-cout block: [position tile][cout tile of MT].
// NOSIG: the caller guarantees p.act != sigmoid (the per-element test of the wave-uniform flag compiles to a branch per
// value: 64 of them per unit)
template <int SH, int TM, bool HEAD, bool NOSIG = false>
__device__ __forceinline__ void epilogue_h(const ConvParamsH& p, typename Mf<SH>::acc_t (&acc)[32 * TM / Mf<SH>::MT][64 / Mf<SH>::MT],
                                           const int* yoff, const float* ep, char* stage, int wave, int li, int lk, int m0,
                                           int n0, int cls, int kz, int bm, int ybase = 0) {
    constexpr int MT = Mf<SH>::MT;
    constexpr int NPT = 32 * TM / MT;             // position tiles per wave
    constexpr int PPU = 32 / MT;                  // position tiles per 32-row staging unit (1 or 2)
    constexpr int GL = groups_per_64<SH>();       // 16-cout groups a lane holds per 64 couts
    // `stage`: this wave's 32 x ST_ROW bytes of the (now idle) operand buffers.  A lane holds 32 B (bf16) /
    // 64 B (fp32 slab) pieces of ONE position; stored as they stand, a wave instruction would touch 16-32 rows with
    // 16-byte pieces.  Staged through LDS, lanes 8r..8r+7 write the 128 contiguous bytes of row r: every store
    // instruction covers eight whole 128-byte lines.
    const int lane = threadIdx.x & 63;
    const int srow = lane >> 3, spiece = lane & 7;
    if (p.ksplit > 1) {
        // split-K: fp32 partial sums [cls][kz][position][CoutPad]; conv_finish_bf16 reduces in kz order.
        const int mpad = p.m_tiles * bm;
        if constexpr (SH == 32) {
            // one staging unit = 32 positions x 32 couts fp32 (128-byte rows): lane (li, lk) holds couts ch*32 + 16 lk + r
            float* __restrict__ slab = p.part + ((size_t)(cls * p.ksplit + kz) * mpad + m0) * p.CoutPad + n0 + 4 * spiece;
#pragma unroll
            for (int u = 0; u < TM; ++u)
#pragma unroll
                for (int ch = 0; ch < 2; ++ch) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const f32x4 t = {acc[u][ch][4 * j], acc[u][ch][4 * j + 1], acc[u][ch][4 * j + 2], acc[u][ch][4 * j + 3]};
                        *reinterpret_cast<f32x4*>(stage + li * ST_ROW + lk * 64 + j * 16) = t;
                    }
                    if (S3R_ABLH(p, 1)) continue;
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int r = i * 8 + srow;
                        const f32x4 t = *reinterpret_cast<const f32x4*>(stage + r * ST_ROW + spiece * 16);
                        *reinterpret_cast<f32x4*>(slab + (size_t)(wave * 32 * TM + u * 32 + r) * p.CoutPad + ch * 32) = t;
                    }
                }
        } else {
            // one staging unit = ONE position tile: 16 positions x 64 couts fp32 (256-byte rows + 16: 4352 B of the wave's
            // 4608).  EVERY lane writes (its 16 couts 16 lk + 4 ct + r of position li) — a unit in which only half the
            // lanes stage data would put the writes in a divergent branch, and nothing orders the other lanes' reads
            // behind it.  Read back as 4 rows x 256 contiguous bytes per store instruction.
            constexpr int SR = 272;
            float* __restrict__ slab = p.part + ((size_t)(cls * p.ksplit + kz) * mpad + m0) * p.CoutPad + n0 + 4 * (lane & 15);
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
#pragma unroll
                for (int j = 0; j < 4; ++j) *reinterpret_cast<f32x4*>(stage + li * SR + lk * 64 + j * 16) = acc[pt][j];
                if (S3R_ABLH(p, 1)) continue;
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const int r = i * 4 + (lane >> 4);
                    const f32x4 t = *reinterpret_cast<const f32x4*>(stage + r * SR + (lane & 15) * 16);
                    *reinterpret_cast<f32x4*>(slab + (size_t)(wave * 32 * TM + pt * MT + r) * p.CoutPad) = t;
                }
            }
        }
        return;
    }
    // ReLU / identity as ONE v_max against a wave-uniform floor; sigmoid (rare) behind a wave-uniform flag
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const bool sig = !NOSIG && p.act == ACT_SIGMOID;
    if constexpr (HEAD) {
        int ye[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) ye[pt] = yoff[wave * 32 * TM + pt * MT + li] + ybase;   // (-1 + 0: still 'none')
        // fused pointwise head (conv -> 1x1x1 conv to ONE channel + activation; the workgroup's 64-cout tile
        // is the whole channel axis): 16 * GL FMAs in the lane, an exchange per lane-group bit, one fp32 store per
        // position.  The conv's own bf16 output — the largest activation of the network — is never written.
        float t[NPT];
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) t[pt] = 0.f;
#pragma unroll
        for (int gi = 0; gi < GL; ++gi) {
            const float* e = ep + group_cout<SH>(gi, lk);
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const f32x4 sc = *reinterpret_cast<const f32x4*>(e + 4 * q);
                const f32x4 sf = *reinterpret_cast<const f32x4*>(e + 64 + 4 * q);
                const f32x4 hw = *reinterpret_cast<const f32x4*>(e + 128 + 4 * q);
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) {
                    t[pt] = fmaf(fmaxf(fmaf(acc_at<SH>(acc[pt], gi, 4 * q + 0), sc.x, sf.x), lo), hw.x, t[pt]);
                    t[pt] = fmaf(fmaxf(fmaf(acc_at<SH>(acc[pt], gi, 4 * q + 1), sc.y, sf.y), lo), hw.y, t[pt]);
                    t[pt] = fmaf(fmaxf(fmaf(acc_at<SH>(acc[pt], gi, 4 * q + 2), sc.z, sf.z), lo), hw.z, t[pt]);
                    t[pt] = fmaf(fmaxf(fmaf(acc_at<SH>(acc[pt], gi, 4 * q + 3), sc.w, sf.w), lo), hw.w, t[pt]);
                }
            }
        }
        const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
        float* __restrict__ y = reinterpret_cast<float*>(p.y);
#pragma unroll
        for (int pt = 0; pt < NPT; ++pt) {
            float v = t[pt] + __shfl_xor(t[pt], 32, 64);          // fixed order: (lk) + (lk ^ 2 | lk ^ 1 for SH = 32) ...
            if constexpr (SH == 16) v += __shfl_xor(v, 16, 64);
            if (lk == 0 && ye[pt] >= 0) {
                v = fmaf(v, hsc, hsf);
                if (p.head_act == ACT_RELU) v = fmaxf(v, 0.f);
                else if (p.head_act == ACT_SIGMOID) v = 1.f / (1.f + __expf(-v));
                y[ye[pt]] = v;
            }
        }
        return;
    }
    unsigned short* __restrict__ y = reinterpret_cast<unsigned short*>(p.y);
    const bool wide = (p.Cout & 7) == 0;          // 16-byte stores need the channel axis in whole groups of 8
#pragma unroll
    for (int u = 0; u < TM; ++u) {                // one staging unit = 32 positions x 64 couts bf16 (128-byte rows)
#pragma unroll
        for (int q = 0; q < PPU; ++q)
#pragma unroll
            for (int gi = 0; gi < GL; ++gi)
#pragma unroll
                for (int g = 0; g < 2; ++g) {
                    const int cl = group_cout<SH>(gi, lk) + 8 * g;      // (tile-local) cout of element 8g of the group
                    const f32x4 sc0 = *reinterpret_cast<const f32x4*>(ep + cl);
                    const f32x4 sc1 = *reinterpret_cast<const f32x4*>(ep + cl + 4);
                    const f32x4 sf0 = *reinterpret_cast<const f32x4*>(ep + 64 + cl);
                    const f32x4 sf1 = *reinterpret_cast<const f32x4*>(ep + 64 + cl + 4);
                    const float sc[8] = {sc0.x, sc0.y, sc0.z, sc0.w, sc1.x, sc1.y, sc1.z, sc1.w};
                    const float sf[8] = {sf0.x, sf0.y, sf0.z, sf0.w, sf1.x, sf1.y, sf1.z, sf1.w};
                    float v[8];
#pragma unroll
                    for (int r = 0; r < 8; ++r) {
                        v[r] = fmaf(acc_at<SH>(acc[u * PPU + q], gi, 8 * g + r), sc[r], sf[r]);
                        v[r] = sig ? __builtin_amdgcn_rcpf(1.f + __expf(-v[r])) : fmaxf(v[r], lo);
                    }
                    const uint4 pk = {pack_bf16(v[0], v[1]), pack_bf16(v[2], v[3]), pack_bf16(v[4], v[5]),
                                      pack_bf16(v[6], v[7])};
                    *reinterpret_cast<uint4*>(stage + (q * MT + li) * ST_ROW + cl * 2) = pk;
                }
        if (S3R_ABLH(p, 1)) continue;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int r = i * 8 + srow;
            const int ye = yoff[wave * 32 * TM + u * 32 + r] + ybase;     // (ybase: a persistent kernel's per-step offset)
            const int co = n0 + 8 * spiece;
            const uint4 pk = *reinterpret_cast<const uint4*>(stage + r * ST_ROW + spiece * 16);
            if (ye < 0 || co >= p.Cout) continue;
            unsigned short* dst = y + (size_t)ye + co;
            if (wide) {
                *reinterpret_cast<uint4*>(dst) = pk;
            } else {
                const unsigned w[4] = {pk.x, pk.y, pk.z, pk.w};
#pragma unroll
                for (int e = 0; e < 8; ++e)
                    if (co + e < p.Cout) dst[e] = (unsigned short)((w[e >> 1] >> (16 * (e & 1))) & 0xffffu);
            }
        }
    }
}

// (the fused-head epilogue keeps 3 constants per cout live: its own instantiation, one workgroup fewer per CU,
// so that the plain kernel stays free of scratch; KC = 64 doubles the LDS per workgroup)
constexpr int min_waves_h(int tm, int kc, bool head) {
    return kc == 64 ? (tm == 1 ? 3 : 1) : (tm >= 4 ? 2 : (head ? 3 : 4));
}

// LDS rows are KC channels = KC*2 bytes = KC/8 sixteen-byte slots; slot = kgroup ^ swz(row) keeps every
// ds_read_b128 lane group ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ... of each 32) on 16 distinct 16-byte slots of
// the 256-byte bank row, for BOTH lane maps (SH = 32: 32 consecutive rows, one k group per lane half; SH = 16: 16
// consecutive rows x 4 k groups):
//   128-byte rows (2 rows per bank row): (row >> 1) & 7;
//   64-byte rows (4 rows per bank row): (-(row >> 2)) & 3 — for SH = 16 a lane group mixes k groups g and g + 1 on rows
//   {0-3, 12-15} and {4-11}: with t = (row >> 2) & 3 the four slots (g ^ f(0), g ^ f(3), (g+1) ^ f(1), (g+1) ^ f(2)) are
//   distinct for f = (0, 3, 2, 1) (the plain f(t) = t collides); any bijection serves SH = 32.
template <int KC>
__device__ __host__ __forceinline__ int swz(int row) { return KC == 64 ? (row >> 1) & 7 : (0 - (row >> 2)) & 3; }

// Per-tap kernel.  K tile = ONE tap x KC channels.  KC = 64 (Cin % 64 == 0) makes every gathered piece a WHOLE
// 128-byte line of the channels-last input (KC = 32 fetches half of each line, and the other half again one
// chunk later: the operand path of these layers is L2 -> LDS bound) and halves the barriers per FLOP.
// The packed weights stay in the 32-channel layout for both: with KC = 64 a lane's 16 bytes come from chunk
// 2j or 2j+1 by a per-lane source offset (LDS-DMA destinations are lane-linear, sources are free).
// NH = 64-cout halves per workgroup: 2 (a 128 x 128 tile; Cout % 128 == 0) gathers the activations once for twice
// the couts — 1.5x the FLOPs per byte brought into LDS, for the layers this kernel keeps (stride 2), which are
// bound by exactly that.
// tap visiting order of stride-2 3x3x3 layers (tap id = (td*3 + th)*3 + tw, 5 bits each, 9 per word): parity classes,
// each along a Gray path —  0 2 8 6 24 26 20 18 | 1 7 25 19 | 3 5 23 21 | 9 11 17 15 | 4 22 | 10 16 | 12 14 | 13
constexpr unsigned long long S2_3D_A = 0x19535832040ull, S2_3D_B = 0xb4d6e51cf27ull, S2_3D_C = 0xd7320ab11f1ull;

template <int SH, int TM, int KC, bool HEAD, int NH>
__global__ __launch_bounds__(256, NH == 2 ? (KC == 64 ? 2 : 3) : min_waves_h(TM, KC, HEAD)) void conv_bf16_kernel(const ConvParamsH p) {
    static_assert(NH == 1 || !HEAD, "the fused head needs the whole channel axis in one 64-cout tile");
    typedef Mf<SH> M;
    constexpr int MT = M::MT;
    constexpr int NPT = 32 * TM / MT;              // position tiles per wave
    constexpr int NCT = 64 / MT;                   // cout tiles per 64-cout block
    constexpr int NKS = KC / M::KS;                // MFMA k steps per K tile
    constexpr int BM = 128 * TM;
    constexpr int BNW = HBN * NH;                  // couts per workgroup
    constexpr int ROWB = KC * 2;                   // bytes per LDS row
    constexpr int LPR = ROWB / 16;                 // lanes (16-byte slots) per row
    constexpr int RPP = 64 / LPR;                  // rows per 1 KiB LDS-DMA piece
    constexpr int A_BYTES = BM * ROWB, B_BYTES = BNW * ROWB;
    constexpr int NPA = A_BYTES / 4096, NPB = B_BYTES / 4096;     // pieces per wave per K tile

    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* As = smem;                               // [2][BM][ROWB]
    char* Bs = smem + 2 * A_BYTES;                 // [2][BNW][ROWB]
    int* xoff = reinterpret_cast<int*>(smem + 2 * A_BYTES + 2 * B_BYTES);   // [BM] input byte offsets
    int* yoff = xoff + BM;                                                   // [BM] output element offsets, -1 = none
    float* ep = reinterpret_cast<float*>(yoff + BM);                         // [NH][3][64] scale / shift / head weight

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane % MT, lk = lane / MT;

    int bid = blockIdx.x, cls = 0;                 // (transposed: the 8 classes of a tile back to back on one XCD, as in
    if (p.transposed) {                            // conv_bf16p_kernel)
        const int nwg = gridDim.x >> 3;
        const int item = (bid & 7) * nwg + (bid >> 3);
        bid = item >> 3;
        cls = item & 7;
    } else {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n_tiles_w = p.n_tiles / NH;          // workgroup cout tiles
    const int n_tile = (bid % n_tiles_w) * NH;     // first 64-cout tile (fastest: neighbours share the gathered input)
    const int m_tile = bid / n_tiles_w;
    const int m0 = m_tile * BM, n0 = n_tile * HBN;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int kz = blockIdx.z;

    const int S = p.Nd * p.Nh * p.Nw;
    const int T = p.T;
    const int chunks = (p.Cin / KC) / p.ksplit;
    const int nkt = S3R_ABLH(p, 2) ? 1 : T * chunks;
#ifdef S3R_ABLATE   // S3R_ABL=7: [cu key, entry, tables done, first K tile landed, loop end, epilogue issued, stores landed, ticks spent in the
    unsigned long long tl[7] = {0, 0, 0, 0, 0, 0, 0};      // loop's `s_waitcnt vmcnt(0)` (operands of the NEXT K tile not landed yet)]
    if (p.debug == 7) tl[0] = __builtin_amdgcn_s_memrealtime();
#endif

    const EpRegs epr = load_ep(p, tid, n0, BNW);
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
        if (p.transposed) xe += (rd - 1) * p.x_ds + (rh - 1) * p.x_hs + (rw - 1) * p.x_ws;
        const int ye = out_offset(p, b, pd, ph, pw, rd, rh, rw);
        xoff[t] = xe * 2;
        yoff[t] = ok ? ye : -1;
    }
    __syncthreads();

    // ---- loop-invariant DMA source offsets (the swizzle is applied on the source side)
    const int w_tile = p.n_tiles * 4096;           // bytes between consecutive 32-channel weight tiles (kt32 + 1)
    int avoff[NPA], bvoff[NPB];
#pragma unroll
    for (int q = 0; q < NPA; ++q) {
        const int pl = (wave + 4 * q) * RPP + lane / LPR;          // position inside the tile
        const int kg = (lane % LPR) ^ swz<KC>(pl);
        avoff[q] = xoff[pl] + kg * 16;
    }
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        if constexpr (KC == 32) {
            bvoff[q] = (wave + 4 * q) * 1024 + lane * 16;          // stored pre-swizzled for 64-byte rows
        } else {
            const int r = (wave + 4 * q) * RPP + lane / LPR;       // cout row (consecutive 64-cout tiles are 4 KiB apart)
            const int kg = (lane % LPR) ^ swz<64>(r);              // 16-byte channel group 0..7 of the 64
            bvoff[q] = (kg >> 2) * (T * w_tile) + r * 64 + (((kg & 3) ^ swz<32>(r)) << 4);
        }
    }

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    // packed weights: [cls][kt32 = chunk32*T + tap][cout tile][64 rows][64 B]
    const size_t w_cls = (size_t)cls * T * (p.Cin / HKC) * w_tile;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + w_cls), 0,
        (int)((unsigned)T * (unsigned)(p.Cin / HKC) * (unsigned)w_tile), 0x00020000);

    int c_td = 0, c_th = 0, c_tw = 0, c_tap = 0, c_cc = kz * chunks;
    // Stride-2 3x3[x3] layers visit their taps by PARITY CLASS.  At stride 2 a tap index 0 and a tap index 2 along an
    // axis read the same input rows (planes, columns) one output position apart, tap index 1 reads the others: the
    // taps fall into classes that gather the same pixels — the 4 (8 in 3D) "corner" taps, 2 + 2 (4 + 4 + 4) edge groups,
    // ..., the centre alone.  In kh-major order the second gather of a pixel comes up to 6 K tiles after the first — of
    // every workgroup on the XCD, ~10 MB of other gathers against a 4 MB L2 — and e3 / v2 fetched 1.5x / 1.6x their
    // input.  Walking each class along a Gray path (consecutive taps differ along one axis) puts every re-read one
    // K tile behind the read it repeats.  The packed weights stay in tap order (c_tap indexes them); the order is a
    // function of the layer's geometry only, so a sample's K order is the same in every batch.
    const bool s2_order = p.stride == 2 && !p.transposed && p.kh == 3 && p.kw == 3 && (p.kd == 1 || p.kd == 3);
    int c_v = 0;                                   // visit index 0 .. T-1 (s2_order)
    auto visit = [&](int v) {                      // -> c_td, c_th, c_tw, c_tap of visit v
        // (orders packed in immediates: a table in memory put a load on the path of every K tile's DMA issue)
        if (p.kd == 1) {
            c_tap = (int)((0x453716820ull >> (4 * v)) & 15ull);                       // 0 2 8 6 | 1 7 | 3 5 | 4
        } else {
            const int g = v / 9, i = v - g * 9;                                       // 27 ids, 5 bits each, 9 per word
            const unsigned long long w = g == 0 ? S2_3D_A : (g == 1 ? S2_3D_B : S2_3D_C);
            c_tap = (int)((w >> (5 * i)) & 31ull);
        }
        c_td = c_tap / 9;
        const int r = c_tap - c_td * 9;
        c_th = r / 3;
        c_tw = r - c_th * 3;
    };
    if (s2_order) visit(0);

    auto issue = [&](int buf) {
        const int b_base = ((c_cc * (KC / 32) * T + c_tap) * p.n_tiles + n_tile) * 4096;
#pragma unroll
        for (int q = 0; q < NPB; ++q)
            dma16(wrsrc, Bs + buf * B_BYTES + (wave + 4 * q) * 1024, bvoff[q], b_base);
        const int a_base = (c_cc * KC + (c_td * p.x_ds + c_th * p.x_hs + c_tw * p.x_ws)) * 2;
#pragma unroll
        for (int q = 0; q < NPA; ++q)
            dma16(xrsrc, As + buf * A_BYTES + (wave + 4 * q) * 1024, avoff[q], a_base);
        if (s2_order) {
            if (++c_v == T) { c_v = 0; ++c_cc; }
            visit(c_v);
        } else {
            if (++c_tw == p.kw) { c_tw = 0; if (++c_th == p.kh) { c_th = 0; ++c_td; } }
            if (++c_tap == T) { c_tap = 0; c_td = 0; c_th = 0; c_tw = 0; ++c_cc; }
        }
    };

    typename M::acc_t acc[NH][NPT][NCT];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
#pragma unroll
        for (int a = 0; a < NPT; ++a)
#pragma unroll
            for (int b = 0; b < NCT; ++b)
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc[nh][a][b][r] = 0.f;

#ifdef S3R_ABLATE
    if (p.debug == 7) tl[1] = __builtin_amdgcn_s_memrealtime();
#endif
    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#ifdef S3R_ABLATE
    if (p.debug == 7) tl[2] = __builtin_amdgcn_s_memrealtime();
#endif

    // fragment byte offsets inside a K tile image: row*ROWB + ((NK*q + lk) ^ swz(row))*16; q toggles the bits above lk's
    int a_off[NPT], b_off[NCT * NH];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int row = wave * 32 * TM + pt * MT + li;
        a_off[pt] = row * ROWB + ((lk ^ swz<KC>(row)) << 4);
    }
#pragma unroll
    for (int ct = 0; ct < NCT * NH; ++ct) {
        const int row = ct * MT + li;
        b_off[ct] = row * ROWB + ((lk ^ swz<KC>(row)) << 4);
    }

    for (int kt = 0; kt < nkt; ++kt) {
        const int cur = kt & 1;
        if (kt + 1 < nkt && !S3R_ABLH(p, 3)) issue(cur ^ 1);
        const char* a = As + cur * A_BYTES;
        const char* b = Bs + cur * B_BYTES;
#pragma unroll
        for (int q = 0; q < NKS; ++q) {
            bf16x8 av[NPT], bv[NCT * NH];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) av[pt] = *reinterpret_cast<const bf16x8*>(a + (a_off[pt] ^ (q * M::NK * 16)));
#pragma unroll
            for (int ct = 0; ct < NCT * NH; ++ct) bv[ct] = *reinterpret_cast<const bf16x8*>(b + (b_off[ct] ^ (q * M::NK * 16)));
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int ct = 0; ct < NCT * NH; ++ct)
                    acc[ct / NCT][pt][ct % NCT] = mma<SH>(bv[ct], av[pt], acc[ct / NCT][pt][ct % NCT]);
        }
#ifdef S3R_ABLATE
        unsigned long long w0 = 0;
        if (p.debug == 7) w0 = __builtin_amdgcn_s_memrealtime();
#endif
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
#ifdef S3R_ABLATE
        if (p.debug == 7) tl[6] += __builtin_amdgcn_s_memrealtime() - w0;
#endif
        __syncthreads();
    }
#ifdef S3R_ABLATE
    if (p.debug == 7) tl[3] = __builtin_amdgcn_s_memrealtime();
#endif

    store_ep(epr, ep, tid, BNW);
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
        epilogue_h<SH, TM, HEAD>(p, acc[nh], yoff, ep + nh * 192, smem + wave * (32 * ST_ROW), wave, li, lk, m0, n0 + nh * HBN,
                                 cls, kz, BM);
#ifdef S3R_ABLATE
    if (p.debug == 7 && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 65536) {
        tl[4] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tl[5] = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* t = s3r_timeline_h + 8 * (size_t)blockIdx.x;
        t[0] = ((unsigned long long)(xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
        for (int i = 0; i < 6; ++i) t[1 + i] = tl[i];
        t[7] = tl[6];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Row-reuse variant: the kw taps of one (chunk, td, th) group read the SAME gathered input
// rows shifted by one position, so the A operand is fetched ONCE per group instead of once per tap.
//
// LDS A image = the input positions the tile needs for a fixed (td, th), in input order: for every run of
// tile positions inside one output row, a segment of stride*(run-1)+kw consecutive input positions
// (64 B = 32 channels each); MFMA row r reads LDS row lrow[r] + tw for tap tw.  No position is computed that
// is not stored (the halo columns sit in LDS but no MFMA row maps to them), the gather is kw (x stride)
// times smaller, and it is CONTIGUOUS in HBM (whole runs of positions) instead of 64-byte pieces.
// The weights of the group's kw taps (kw x 4 KiB, consecutive in the packed image) ride along, so there
// is one barrier per GROUP.
constexpr int NPA_MAX = 10;    // 16-position A pieces per wave per group, upper bound (registers)

template <int SH, int TM>
__global__ __launch_bounds__(256, 2) void conv_bf16r_kernel(const ConvParamsH p, int r_max) {
    typedef Mf<SH> M;
    constexpr int MT = M::MT;
    constexpr int NPT = 32 * TM / MT, NCT = 64 / MT, NKS = HKC / M::KS;
    constexpr int BM = 128 * TM;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int kw = p.kw;
    const int a_bytes = r_max * 64, stage_bytes = a_bytes + kw * 4096;
    int* yoff = reinterpret_cast<int*>(smem + 2 * stage_bytes);     // [BM] output element offsets, -1 = none
    int* lrow = yoff + BM;                                            // [BM] LDS row of each tile position (tap tw = 0)
    float* ep = reinterpret_cast<float*>(lrow + BM);                  // [3][64] scale / shift / head weight

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane % MT, lk = lane / MT;

    int bid = blockIdx.x, cls = 0;                 // (transposed: the 8 classes of a tile back to back on one XCD, as in
    if (p.transposed) {                            // conv_bf16p_kernel)
        const int nwg = gridDim.x >> 3;
        const int item = (bid & 7) * nwg + (bid >> 3);
        bid = item >> 3;
        cls = item & 7;
    } else {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n_tile = bid % p.n_tiles;
    const int m_tile = bid / p.n_tiles;
    const int m0 = m_tile * BM, n0 = n_tile * HBN;
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

    const EpRegs epr = load_ep(p, tid, n0);
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
        const int ye = out_offset(p, b, pd, ph, pw, rd, rh, rw);
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
        const int kg = (lane & 3) ^ swz<32>(lr);              // swizzle on the source side
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

    typename M::acc_t acc[NPT][NCT];
#pragma unroll
    for (int a = 0; a < NPT; ++a)
#pragma unroll
        for (int b = 0; b < NCT; ++b)
#pragma unroll
            for (int r = 0; r < M::NACC; ++r) acc[a][b][r] = 0.f;

    issue(0);
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();                                          // also publishes yoff / lrow

    int lr[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) lr[pt] = lrow[wave * 32 * TM + pt * MT + li];
    int b_off[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int row = ct * MT + li;
        b_off[ct] = row * 64 + ((lk ^ swz<32>(row)) << 4);
    }

    for (int g = 0; g < ngroups; ++g) {
        const int cur = g & 1;
        if (g + 1 < ngroups && !S3R_ABLH(p, 3)) issue(cur ^ 1);
        const char* a = smem + cur * stage_bytes;
        const char* b = a + a_bytes;
        for (int tw = 0; tw < kw; ++tw) {
            int a_off[NPT];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int row = lr[pt] + tw;
                a_off[pt] = (row << 6) + ((lk ^ swz<32>(row)) << 4);
            }
#pragma unroll
            for (int q = 0; q < NKS; ++q) {
                bf16x8 av[NPT], bv[NCT];
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt) av[pt] = *reinterpret_cast<const bf16x8*>(a + (a_off[pt] ^ (q * M::NK * 16)));
#pragma unroll
                for (int ct = 0; ct < NCT; ++ct)
                    bv[ct] = *reinterpret_cast<const bf16x8*>(b + tw * 4096 + (b_off[ct] ^ (q * M::NK * 16)));
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct)
                        acc[pt][ct] = mma<SH>(bv[ct], av[pt], acc[pt][ct]);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
    }

    store_ep(epr, ep, tid);
    if (p.head_w) epilogue_h<SH, TM, true>(p, acc, yoff, ep, smem + wave * (32 * ST_ROW), wave, li, lk, m0, n0, cls, kz, BM);
    else epilogue_h<SH, TM, false>(p, acc, yoff, ep, smem + wave * (32 * ST_ROW), wave, li, lk, m0, n0, cls, kz, BM);
}

// ------------------------------------------------------------------------------------------------
// Plane-reuse variant (stride-1 layers): ALL kh*kw taps of one (KC-channel chunk, td) group
// read the same gathered input PLANE, shifted by th*in_p + tw positions, so the A operand is fetched once per
// kh*kw taps (9 for a 3x3[x3] conv, 4 for a transposed-conv class) instead of once per tap / per kw taps.
//
// LDS A image = for every (batch, depth) plane the tile touches, the contiguous run of padded-input positions
// u = ph*in_p + pw .. that its output positions need, plus
// HALO = (kh-1)*in_p + kw-1 trailing positions; MFMA row r reads LDS row lrow[r] + th*in_p + tw.  The image is
// SINGLE-buffered (two or three workgroups per CU overlap one's reload with the others' MFMAs); the weights
// stream through a 3-slot ring with one barrier per tap.
constexpr int NPA_PL = 15;     // 8-row A pieces per wave, upper bound (registers), 128-byte rows
constexpr int NPA_PL32 = 12;   // 16-row pieces, 64-byte rows
// weight ring slots of the plane kernel: a tap's weights are requested pl_nb - 1 taps ahead.  Three slots (two taps of lead) is the
// measured optimum: r06 built the ring for any depth and five slots (four taps of lead; + 8 KiB per workgroup, which takes e4 / v3 /
// v5 / d3 from three workgroups per CU to two) measured e4 + 19 %, d3 + 17 %, v3 + 11 %, v5 + 13 % SLOWER at B = 256, e7 / d1 / d2
// - 2 ... - 3 % (gpurun_out/r6e/ab, A/B on one device): occupancy, not the weights' latency, is what these loops live on.
#ifndef S3R_PL_NB32
#define S3R_PL_NB32 3
#endif
constexpr int pl_nb(int kc) { return kc == 32 ? S3R_PL_NB32 : 3; }
template <int N> __device__ __forceinline__ void wait_vm() { asm volatile("s_waitcnt vmcnt(%0)" :: "n"(N) : "memory"); }
__device__ __forceinline__ void wait_vm_n(int n) {      // n is wave-uniform and small: one immediate form per value
    switch (n) {
        case 0: wait_vm<0>(); break;
        case 1: wait_vm<1>(); break;
        case 2: wait_vm<2>(); break;
        case 3: wait_vm<3>(); break;
        case 4: wait_vm<4>(); break;
        case 5: wait_vm<5>(); break;
        case 6: wait_vm<6>(); break;
        case 7: wait_vm<7>(); break;
        default: wait_vm<8>(); break;
    }
}

// KC = 64: 128-byte image rows (whole lines), two workgroups per CU; KC = 32: 64-byte rows, half the LDS, three
// workgroups per CU (and the only form for Cin = 32).
// NH = 64-cout halves per workgroup (2: a 256 x 128 tile, the image serves twice the couts; two workgroups per CU).
// HEAD: the fused pointwise-head epilogue is its own instantiation (d3 only).  Compiled into the same kernel behind a
// run-time test (r01), the two epilogues shared one register allocation at the 168-VGPR cap of three waves per SIMD
// and every layer's kernel spilled 40 bytes per lane to scratch; now the plain kernel needs 113 VGPRs and none.  The
// head instantiation still spills its 40 bytes at three waves per SIMD — measured faster (d3 1.32 vs 1.45 ms) than
// giving it 176 registers at two.
template <int SH, int TM, int KC, int NH, bool HEAD = false>
__global__ __launch_bounds__(256, (KC == 64 || NH == 2) ? 2 : 3) void conv_bf16p_kernel(const ConvParamsH p, int r_max) {
    static_assert(!HEAD || NH == 1, "the fused head needs the whole channel axis in one 64-cout tile");
    typedef Mf<SH> M;
    constexpr int MT = M::MT;
    constexpr int NPT = 32 * TM / MT, NCT = 64 / MT, NKS = KC / M::KS;
    constexpr int BM = 128 * TM;
    constexpr int BNW = HBN * NH;                  // couts per workgroup
    constexpr int ROWB = KC * 2;                   // bytes per LDS row
    constexpr int LPR = ROWB / 16;                 // lanes (16-byte slots) per row
    constexpr int RPP = 64 / LPR;                  // rows per 1 KiB LDS-DMA piece
    constexpr int B_BYTES = BNW * ROWB;            // one tap's weight tile
    constexpr int NPB = B_BYTES / 4096;            // its pieces per wave
    constexpr int NPA_CAP = KC == 64 ? NPA_PL : NPA_PL32;
    constexpr int PL_NB = pl_nb(KC);               // weight ring slots; a tap's weights are requested PL_D taps ahead
    constexpr int PL_D = PL_NB - 1;
    static_assert(NPB * (PL_D - 1) <= 8, "wait_vm_n covers 0 .. 8");
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int a_bytes = r_max * ROWB;
    char* Bs = smem + a_bytes;                                        // [PL_NB][64][ROWB]
    int* yoff = reinterpret_cast<int*>(Bs + PL_NB * B_BYTES);         // [BM] output element offsets, -1 = none
    int* lrow = yoff + BM;                                            // [BM] LDS row of each tile position (tap 0,0)
    float* ep = reinterpret_cast<float*>(lrow + BM);                  // [3][64] scale / shift / head weight
    int* asrc = reinterpret_cast<int*>(Bs + (PL_NB - 1) * B_BYTES);  // [r_max] source byte offset of each image row
                                                                      // (prologue only: aliases the last ring slot)
    // The prologue (tables, first image) and the epilogue are short, latency-bound instruction streams that run beside
    // two other workgroups' K loops on the same SIMDs; as the YOUNGEST wave of its SIMD this one loses every issue
    // arbitration (priority, then age) and a timeline showed 3-4 us of tables and 4.4-5.8 us of epilogue per workgroup
    // (tools/timeline_bf16.py).  Raised priority outside the K loop lets those phases through; the loop runs at 0.
    __builtin_amdgcn_s_setprio(S3R_PRIO_EDGE);
    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane % MT, lk = lane / MT;
#ifdef S3R_ABLATE
    unsigned long long tl[6] = {0, 0, 0, 0, 0, 0};
    if (p.debug == 7) tl[0] = __builtin_amdgcn_s_memrealtime();
#endif

    // Workgroup order.  Convolutions: each XCD (blocks b, b+8, ... share one) walks a contiguous run of tiles.
    // Transposed convolutions: the grid is 8 x as long and an XCD walks its run of tiles with the 8 output-parity classes
    // of a tile BACK TO BACK — they read the same input tile, so seven of the eight reads are hits in that XCD's L2.
    // With the class on blockIdx.y instead (all tiles of class 0, then of class 1, ...) every class streamed the whole
    // input from HBM again once it outgrew the caches: d3 at B = 256 fetched 1.67 GB per launch for a 382 MB input.
    int bid = blockIdx.x, cls = 0;
    if (p.transposed) {
        const int nwg = gridDim.x >> 3;
        const int item = (bid & 7) * nwg + (bid >> 3);
        bid = item >> 3;
        cls = item & 7;
    } else {
        const int nwg = gridDim.x;
        const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, slot = bid >> 3;
        bid = (xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q) + slot;
    }
    const int n_tiles_w = p.n_tiles / NH;
    const int n_tile = (bid % n_tiles_w) * NH;     // first 64-cout tile of this workgroup
    const int m_tile = bid / n_tiles_w;
    const int m0 = m_tile * BM, n0 = n_tile * HBN;
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int kz = blockIdx.z;

    const int S = p.Nd * p.Nh * p.Nw, P = p.Nh * p.Nw;
    const int T = p.T, kh = p.kh, kw = p.kw;
    const int taps_g = kh * kw;
    const int chunks = (p.Cin / KC) / p.ksplit;
    const int gpc = p.kd;                                    // groups (image loads) per chunk
    const int ngroups = S3R_ABLH(p, 2) ? 1 : chunks * gpc;
    const int total = ngroups * taps_g;
    const int cls_x = p.transposed ? (rd - 1) * p.x_ds + (rh - 1) * p.x_hs + (rw - 1) * p.x_ws : 0;
    const int in_p = p.x_hs / p.x_ws;                        // padded input row length, in positions
    const int halo = (kh - 1) * in_p + kw - 1;
    const int umax = (p.Nh - 1) * in_p + p.Nw - 1;
    const int LP = umax + 1 + halo;                          // image rows of a whole plane

    // ---- geometry of the tile's first and last position (wave-uniform)
    const int m1 = (m0 + BM < p.Ntotal ? m0 + BM : p.Ntotal) - 1;
    int pl0, u0, pl1, u1;
    {
        pl0 = p.dHW.div(m0);
        int nl = m0 - pl0 * P;
        int ph = p.dW.div(nl);
        u0 = ph * in_p + (nl - ph * p.Nw);
        pl1 = p.dHW.div(m1);
        nl = m1 - pl1 * P;
        ph = p.dW.div(nl);
        u1 = ph * in_p + (nl - ph * p.Nw);
    }
    const int nseg = pl1 - pl0 + 1;
    const int len0 = (nseg > 1 ? umax : u1) - u0 + 1 + halo; // image rows of the first plane's segment

    const EpRegs epr = load_ep(p, tid, n0, BNW);

    // ---- the first two taps' weights depend on nothing the prologue computes: fetch them before it
    const int w_tile = p.n_tiles * 4096;
    int bvoff[NPB];
#pragma unroll
    for (int q = 0; q < NPB; ++q) {
        if constexpr (KC == 32) {
            bvoff[q] = (wave + 4 * q) * 1024 + lane * 16;     // stored pre-swizzled for 64-byte rows
        } else {
            const int r = (wave + 4 * q) * RPP + lane / LPR;
            const int kg = (lane % LPR) ^ swz<64>(r);
            bvoff[q] = (kg >> 2) * (T * w_tile) + r * 64 + (((kg & 3) ^ swz<32>(r)) << 4);
        }
    }
    const size_t w_cls = (size_t)cls * T * (p.Cin / HKC) * w_tile;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<char*>(reinterpret_cast<const char*>(p.w) + w_cls), 0,
        (int)((unsigned)T * (unsigned)(p.Cin / HKC) * (unsigned)w_tile), 0x00020000);
    int b_cc = kz * chunks, b_tap = 0, b_slot = 0;            // cursor of the NEXT weight tile to fetch
    auto issue_b = [&]() {
        const int b_base = ((b_cc * (KC / 32) * T + b_tap) * p.n_tiles + n_tile) * 4096;
#pragma unroll
        for (int q = 0; q < NPB; ++q) dma16(wrsrc, Bs + b_slot * B_BYTES + ((wave + 4 * q) << 10), bvoff[q], b_base);
        if (++b_tap == T) { b_tap = 0; ++b_cc; }
        if (++b_slot == PL_NB) b_slot = 0;
    };
#pragma unroll
    for (int i = 0; i < PL_D; ++i)
        if (i < total) issue_b();

    for (int t = tid; t < BM; t += 256) {
        const int n = m0 + t;
        const bool ok = n < p.Ntotal;
        const int nn = ok ? n : p.Ntotal - 1;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * P;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        const int ye = out_offset(p, b, pd, ph, pw, rd, rh, rw);
        yoff[t] = ok ? ye : -1;
        const int sgm = b * p.Nd + pd - pl0, u = ph * in_p + pw;
        lrow[t] = sgm == 0 ? u - u0 : len0 + (sgm - 1) * LP + u;
    }
    // source of every image row: (plane, position inside the padded plane)
    for (int j = tid; j < r_max; j += 256) {
        int sgm, off;
        if (j < len0) { sgm = 0; off = j + u0; }
        else { sgm = 1 + (j - len0) / LP; off = (j - len0) - (sgm - 1) * LP; }
        int pl = pl0 + sgm;
        if (sgm >= nseg) { pl = pl1; off = 0; }              // past the image: any valid address, never read
        const int b = p.dS.div(pl * P);
        const int pd = pl - b * p.Nd;
        asrc[j] = (b * p.x_bs + p.x_org + pd * p.x_ds + off * p.x_ws + cls_x) * 2;
    }
    // publish the LDS tables.  NOT __syncthreads(): the compiler drains vmcnt before a barrier it knows about, which would
    // wait out the two weight tiles and the per-cout constants requested above (an L2 / HBM round trip in the prologue
    // of every workgroup) — they only have to land by the first tap / the epilogue
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");

    // ---- loop-invariant DMA source offsets
    // A pieces (RPP rows each) are dealt round-robin to the four waves; r_max is a whole number of pieces
    int avoff[NPA_CAP];
#pragma unroll
    for (int q = 0; q < NPA_CAP; ++q) {
        const int j = (wave + 4 * q) * RPP + lane / LPR;
        avoff[q] = (j < r_max ? asrc[j] : 0) + (((lane % LPR) ^ swz<KC>(j)) << 4);
    }
    int lr[NPT];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) lr[pt] = lrow[wave * 32 * TM + pt * MT + li];
    int b_off[NCT * NH];
#pragma unroll
    for (int ct = 0; ct < NCT * NH; ++ct) {
        const int row = ct * MT + li;
        b_off[ct] = row * ROWB + ((lk ^ swz<KC>(row)) << 4);
    }

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    int a_cc = kz * chunks, a_td = 0;                         // cursor of the NEXT image to fetch
    auto issue_a = [&]() {
        const int a_base = (a_cc * KC + a_td * p.x_ds) * 2;
#pragma unroll
        for (int q = 0; q < NPA_CAP; ++q)
            if ((wave + 4 * q) * RPP < r_max) dma16(xrsrc, smem + ((wave + 4 * q) << 10), avoff[q], a_base);
        if (++a_td == gpc) { a_td = 0; ++a_cc; }
    };
    typename M::acc_t acc[NH][NPT][NCT];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
#pragma unroll
        for (int a = 0; a < NPT; ++a)
#pragma unroll
            for (int b = 0; b < NCT; ++b)
#pragma unroll
                for (int r = 0; r < M::NACC; ++r) acc[nh][a][b][r] = 0.f;

#ifdef S3R_ABLATE
    if (p.debug == 7) tl[1] = __builtin_amdgcn_s_memrealtime();
#endif
    issue_a();                                                // (asrc aliases the LAST ring slot, first filled
                                                              //  behind the loop's first barrier)

    __builtin_amdgcn_s_setprio(0);
    int tt = 0, c_slot = 0;
    for (int g = 0; g < ngroups; ++g) {
        int tapoff = 0, c_tw = 0;
        const int ntaps = taps_g;
        for (int t = 0; t < ntaps; ++t, ++tt) {
            // this tap's weights (and, at t == 0, the image) have landed; the next tap's may still be in flight.
            // (s_barrier as inline asm: the compiler drains vmcnt before every barrier it knows about, which
            // would cut the weight prefetch back to one tap)
            // (the taps tt + 1 .. tt + PL_D - 1 requested behind it may still be in flight; at t == 0 the image, requested last, must
            // have landed too)
            {
                const int ahead = total - 1 - tt < PL_D - 1 ? total - 1 - tt : PL_D - 1;
                if (t == 0) wait_vm<0>();
                else wait_vm_n(NPB * ahead);
            }
            asm volatile("s_barrier" ::: "memory");           // ... for every wave; ring slot (tt + PL_D) % PL_NB is free
#ifdef S3R_ABLATE
            if (p.debug == 7 && tt == 0) tl[2] = __builtin_amdgcn_s_memrealtime();
#endif
            if (tt + PL_D < total && !S3R_ABLH(p, 3)) issue_b();
            const char* b = Bs + c_slot * B_BYTES;
            int a_off[NPT];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) {
                const int row = lr[pt] + tapoff;
                a_off[pt] = row * ROWB + ((lk ^ swz<KC>(row)) << 4);
            }
            // fragments of k-step q+1 are requested before the MFMAs of k-step q
            bf16x8 av[2][NPT], bv[2][NCT * NH];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt) av[0][pt] = *reinterpret_cast<const bf16x8*>(smem + a_off[pt]);
#pragma unroll
            for (int ct = 0; ct < NCT * NH; ++ct) bv[0][ct] = *reinterpret_cast<const bf16x8*>(b + b_off[ct]);
#pragma unroll
            for (int q = 0; q < NKS; ++q) {
                if (q < NKS - 1) {
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt)
                        av[(q + 1) & 1][pt] = *reinterpret_cast<const bf16x8*>(smem + (a_off[pt] ^ ((q + 1) * M::NK * 16)));
#pragma unroll
                    for (int ct = 0; ct < NCT * NH; ++ct)
                        bv[(q + 1) & 1][ct] = *reinterpret_cast<const bf16x8*>(b + (b_off[ct] ^ ((q + 1) * M::NK * 16)));
                }
#pragma unroll
                for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                    for (int ct = 0; ct < NCT * NH; ++ct)
                        acc[ct / NCT][pt][ct % NCT] = mma<SH>(bv[q & 1][ct], av[q & 1][pt], acc[ct / NCT][pt][ct % NCT]);
            }
            // pin the issue order (the scheduler otherwise reuses the fragment registers and serialises
            // read -> wait -> MFMA): k-step 0's reads, then per k-step one read of the NEXT step behind each MFMA
            if constexpr (NKS > 1) {
                __builtin_amdgcn_sched_group_barrier(0x100, NPT + NCT * NH, 0);
#pragma unroll
                for (int q = 0; q < NKS - 1; ++q)
#pragma unroll
                    for (int i = 0; i < NCT * NH * NPT; ++i) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        if (i < NPT + NCT * NH) __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                    }
                __builtin_amdgcn_sched_group_barrier(0x008, NCT * NH * NPT, 0);
            }
            if (++c_slot == PL_NB) c_slot = 0;
            ++tapoff;
            if (++c_tw == kw) { c_tw = 0; tapoff += in_p - kw; }
        }
        asm volatile("s_barrier" ::: "memory");               // every wave is done with this image
        if (g + 1 < ngroups && !S3R_ABLH(p, 3)) issue_a();
    }

#ifdef S3R_ABLATE
    if (p.debug == 7) tl[3] = __builtin_amdgcn_s_memrealtime();
#endif
    __builtin_amdgcn_s_setprio(S3R_PRIO_EDGE);
    store_ep(epr, ep, tid, BNW);
#pragma unroll
    for (int nh = 0; nh < NH; ++nh)
        epilogue_h<SH, TM, HEAD>(p, acc[nh], yoff, ep + nh * 192, smem + wave * (32 * ST_ROW), wave, li, lk, m0, n0 + nh * HBN,
                                 cls, kz, BM);
#ifdef S3R_ABLATE
    if (p.debug == 7 && tid == 0 && blockIdx.y == 0 && blockIdx.z == 0 && blockIdx.x < 65536) {
        tl[4] = __builtin_amdgcn_s_memrealtime();
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        tl[5] = __builtin_amdgcn_s_memrealtime();
        const unsigned hw = __builtin_amdgcn_s_getreg((15 << 11) | (0 << 6) | 4);
        unsigned xcc;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long* t = s3r_timeline_h + 8 * (size_t)blockIdx.x;
        t[0] = ((unsigned long long)(xcc & 15u) << 8) | ((hw >> 8) & 0xffu);
        for (int i = 0; i < 6; ++i) t[1 + i] = tl[i];
    }
#endif
}

// ------------------------------------------------------------------------------------------------
// Row-persistent variant for the shallow front layer (e2: Conv2d 32 -> 64, k3 s1 p1 over 112 x 112; K = 288).  In the
// plane kernel above such a layer is all fixed cost: 11.6 us of tables, first image, epilogue and store tail per
// workgroup around 1.2 us of MFMAs, its input fetched 1.9 times over (every tile reloads its halo rows), its output
// stored in 32-position pieces (tools/timeline_bf16.py).  Here a workgroup of SEVEN waves owns a strip
// of output rows of one image and slides down it FOUR output rows (448 positions = 7 waves x 2 tiles of 32) per step:
//   * the layer's weights (9 taps x 4 KiB per 64 couts) are fetched ONCE per workgroup and stay in LDS;
//   * input rows live in a ring of ten 8-KiB slots (a padded row is 114 x 64 B; slot = padded row mod 10); a step
//     reads six of them and, at its start, requests the four NEW rows of the next step — one contiguous 29 KB run of the
//     channels-last input, each row as eight whole 1-KiB LDS-DMA pieces — so every input row crosses HBM -> LDS once;
//   * ONE workgroup barrier per step (rows landed for everyone / everyone done with the rows about to be replaced),
//     none inside it: 72 MFMAs per wave straight through, operands from LDS only;
//   * the position -> (row, column) tables are computed once per workgroup: a step moves two scalar offsets;
//   * the four output rows of a step are 448 x 128 B = 56 KB contiguous in the channels-last output.
// The K order (tap-major, two 16-channel k-steps per tap) is the plane kernel's, so the two produce the same bits and
// the library is free to pick by batch size (the rows kernel needs >= 256 strips to fill the chip).
constexpr int RW_WAVES = 7;
constexpr int RW_R = 4;                          // output rows per step
constexpr int RW_NS = 10;                        // ring slots (input rows)
constexpr int RW_SLOT = 128 * 64;                // bytes per slot: 128 positions x 32 channels (114 used)
constexpr int RW_LDS = RW_NS * RW_SLOT + 9 * 4096 + RW_WAVES * 32 * ST_ROW + 448 * 4 + EP_BYTES;

template <int SH>
__global__ __launch_bounds__(64 * RW_WAVES) void conv_bf16w_kernel(const ConvParamsH p, int strips, int units) {
    typedef Mf<SH> M;
    constexpr int MT = M::MT;
    constexpr int NPT = 64 / MT, NCT = 64 / MT, NKS = HKC / M::KS;
    constexpr int W = 112, WP = 114;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* ring = smem;                                               // [RW_NS][128][64 B]
    char* wts = smem + RW_NS * RW_SLOT;                              // [9 taps][64 rows][64 B] (pre-swizzled by the pack kernel)
    char* stage = wts + 9 * 4096 + 0;                                // [RW_WAVES][32][ST_ROW]
    int* yoff = reinterpret_cast<int*>(stage + RW_WAVES * 32 * ST_ROW);   // [448] output element offset inside a step
    float* ep = reinterpret_cast<float*>(yoff + 448);

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int li = lane % MT, lk = lane / MT;
    const int n_tile = blockIdx.y, n0 = n_tile * HBN;

    const __amdgpu_buffer_rsrc_t xrsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<void*>(p.x), 0, (int)p.x_bytes, 0x00020000);
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<void*>(p.w), 0, (int)(9u * (unsigned)p.n_tiles * 4096u), 0x00020000);

    // ---- once per workgroup: the weights, the per-cout constants, the position tables
    for (int pc = wave; pc < 36; pc += RW_WAVES) {                   // 36 pieces of 1 KiB: tap = pc >> 2
        const int tap = pc >> 2;
        dma16(wrsrc, wts + pc * 1024, lane * 16, ((tap * p.n_tiles + n_tile) * 4 + (pc & 3)) * 1024);
    }
    const EpRegs epr = load_ep(p, tid, n0);
    for (int t = tid; t < RW_R * W; t += 64 * RW_WAVES) {
        const int dr = t / W, c = t - dr * W;
        yoff[t] = dr * p.y_hs + c * p.y_ws;
    }
    store_ep(epr, ep, tid);                                          // (__syncthreads inside: also publishes yoff)

    // per lane: the two position tiles' (row in step, column), and per tw the byte offset inside a slot
    int dr[NPT], colpart[NPT][3];
#pragma unroll
    for (int pt = 0; pt < NPT; ++pt) {
        const int n = wave * 64 + pt * MT + li;
        dr[pt] = n / W;
        const int c = n - dr[pt] * W;
#pragma unroll
        for (int tw = 0; tw < 3; ++tw) colpart[pt][tw] = (c + tw) * 64 + ((lk ^ swz<32>(c + tw)) << 4);
    }
    int b_off[NCT];
#pragma unroll
    for (int ct = 0; ct < NCT; ++ct) {
        const int row = ct * MT + li;
        b_off[ct] = row * 64 + ((lk ^ swz<32>(row)) << 4);
    }
    // DMA source offset of this lane inside a 16-position piece: position l >> 2, channel group (l & 3) ^ swz(position).
    // A slot starts at a multiple of 128 LDS rows and a piece at a multiple of 16, so the swizzle of an LDS row depends
    // on the lane only: ONE loop-invariant offset, the piece's 1 KiB goes into the scalar offset.
    const int pvoff = (lane >> 2) * 64 + (((lane & 3) ^ swz<32>(lane >> 2)) << 4);
    // rows [r0, r0 + nr) of the padded image at byte offset `img` -> their ring slots; 8 pieces per row
    auto issue_rows = [&](int img, int r0, int nr) {
        for (int pc = wave; pc < nr * 8; pc += RW_WAVES) {
            const int rr = pc >> 3, i = pc & 7;
            const int prow = r0 + rr;
            const int slot = prow % RW_NS;
            dma16(xrsrc, ring + slot * RW_SLOT + i * 1024, pvoff, img + prow * (WP * 64) + i * 1024);
        }
    };

    const int rows_per_strip = W / strips;                           // (launcher: a multiple of RW_R)
    const int steps = rows_per_strip / RW_R;
    for (int unit = blockIdx.x; unit < units; unit += gridDim.x) {
        const int b = unit / strips, st = unit - b * strips;
        const int img = (b * p.x_bs) * 2;                            // byte offset of the padded image (x_org = 0: halo 1, pad 1)
        const int row0 = st * rows_per_strip;                        // first output row = first padded input row of the strip
        __syncthreads();                                             // every wave is done with the previous unit's rows
        issue_rows(img, row0, RW_R + 2);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        for (int s = 0; s < steps; ++s) {
            const int p0 = row0 + s * RW_R;                          // padded input row of tap row 0 of the step's first output row
            __builtin_amdgcn_s_barrier();                            // this step's rows have landed for every wave; the rows of
            asm volatile("" ::: "memory");                           // step s - 1 that the next request replaces are free
            if (s + 1 < steps) issue_rows(img, p0 + RW_R + 2, RW_R);
            int slotoff[NPT][3];
#pragma unroll
            for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                for (int th = 0; th < 3; ++th) slotoff[pt][th] = ((p0 + dr[pt] + th) % RW_NS) * RW_SLOT;
            typename M::acc_t acc[NPT][NCT];
#pragma unroll
            for (int a = 0; a < NPT; ++a)
#pragma unroll
                for (int c = 0; c < NCT; ++c)
#pragma unroll
                    for (int r = 0; r < M::NACC; ++r) acc[a][c][r] = 0.f;
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) {
                const int th = tap / 3, tw = tap - th * 3;
                const char* b = wts + tap * 4096;
#pragma unroll
                for (int q = 0; q < NKS; ++q) {
                    bf16x8 av[NPT], bv[NCT];
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt)
                        av[pt] = *reinterpret_cast<const bf16x8*>(ring + ((slotoff[pt][th] + colpart[pt][tw]) ^ (q * M::NK * 16)));
#pragma unroll
                    for (int ct = 0; ct < NCT; ++ct) bv[ct] = *reinterpret_cast<const bf16x8*>(b + (b_off[ct] ^ (q * M::NK * 16)));
#pragma unroll
                    for (int pt = 0; pt < NPT; ++pt)
#pragma unroll
                        for (int ct = 0; ct < NCT; ++ct) acc[pt][ct] = mma<SH>(bv[ct], av[pt], acc[pt][ct]);
                }
            }
            const int ybase = b * p.y_bs + p.y_org + (row0 + s * RW_R) * p.y_hs;
            // the rows of step s + 1 (requested at the top of this step, a whole MFMA phase ago) and the previous step's stores are
            // all that is outstanding here: wait for them BEFORE this step's stores go out, so that the wait does not have to know
            // how many stores the epilogue issues (it used to be vmcnt(8) behind it: a count shared silently by two functions)
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            epilogue_h<SH, 2, false, true>(p, acc, yoff, ep, stage + wave * (32 * ST_ROW), wave, li, lk, 0, n0, 0, 0, 0, ybase);
        }
    }
}

// split-K finish: y[pos][cout] = bf16(act(scale * sum_kz slab + shift)); one thread per (position, cout pair)
__global__ __launch_bounds__(256) void conv_finish_bf16_kernel(const ConvParamsH p, int mpad) {
    const int S = p.Nd * p.Nh * p.Nw;
    const int cls = blockIdx.y;                       // (this kernel's own grid: y = output parity class)
    const int rd = (cls >> 2) & 1, rh = (cls >> 1) & 1, rw = cls & 1;
    const int half = p.CoutPad >> 1;
    const long long total = (long long)p.Ntotal * half;
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
        const int ye = out_offset(p, b, pd, ph, pw, rd, rh, rw);
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
static int plane_rows(const ConvParamsH& p, int bm, int kc);
static int rows_strips(const ConvParamsH& p);

// Tile / gather choice (tools/layer_bench.py --dtype bf16, B = 256, MI355X):
//   * stride-1 layers with 4+ taps per plane and K >= 256 whose 256-position plane image leaves room for THREE
//     workgroups per CU (e2, e4, e6, e7, v1, v3, v5, d1, d2, d3) run the plane-reuse gather with 32-channel K tiles (code 22): 5-25 %
//     faster than the per-tap / row-reuse kernels, and 2-7 % faster than its own 64-channel form (code 6, two
//     workgroups per CU): overlapping one workgroup's image reload and stores with the others' MFMAs is worth
//     more than whole-line gathers once the image is fetched only once per kh*kw taps;
//   * everything else (stride 2, shallow K, v6's 16-position planes) runs the per-tap kernel with 128-position
//     tiles — more workgroups per CU to overlap loads and stores, 64-channel K tiles wherever Cin % 64 == 0 — and
//     128-cout workgroup tiles (code 3) where Cout % 128 == 0: these layers are bound by L2 -> LDS delivery;
//   * the row-reuse gather (codes 9 / 10) remains for deep stride-1 layers with an odd number of 64-cout tiles.
int conv_bf16_pick_tm(const ConvParamsH& p) {
    const long classes = (p.transposed ? 8 : 1) * (long)p.ksplit;
    const long n_tiles = p.CoutPad / HBN;
    auto wgs = [&](int tm) { return ((p.Ntotal + 128 * tm - 1) / (128 * tm)) * n_tiles * classes; };
    // The K summation order of a layer must not depend on the batch (a sample's result is batch-invariant): the
    // plane / row-reuse kernels and the 32-channel per-tap kernel accumulate chunk32-major, the 64-channel per-tap
    // kernel chunk64-major.  So the FAMILY is chosen from per-sample geometry, only the tile from the batch.
    const int pr = plane_rows(p, 256, 32);
    const bool plane_family = pr > 0 && pr <= 576 && (p.Cin / 32) % p.ksplit == 0 && p.kh * p.kw >= 4 &&
                              (long)p.Cin * p.T >= 256;      // (e2, K = 288: -12 % once its planes tile exactly)
    // (transposed classes have only 4 taps per image: its 256 x 128-cout form, code 23, amortises the image over twice
    //  the couts — d1 -10 %, d2 -6 %; the 9-tap layers are level or slower with it)
    if (plane_family && rows_strips(p)) return 40;           // (same K order as the plane kernel: a per-batch choice)
    if (plane_family && wgs(2) >= 512) return (p.transposed && n_tiles % 2 == 0 && wgs(2) / 2 >= 512) ? 23 : 22;
    // row-reuse (32-channel K order, like the plane kernel): small batches of deep plane-family layers (v5 at B = 32)
    const bool deep = !p.transposed && p.kw >= 3 && (long)p.Cin * p.T >= 64 * 27;
    const bool reuse = p.stride == 1 && deep && rowreuse_rows(p, 128) <= 64 * NPA_MAX;
    if (plane_family) return reuse ? 9 : 17;
    // the per-tap family (64-channel K tiles where Cin allows): 128 x 128-cout tiles when the channel axis has an even
    // number of 64-cout tiles and the grid stays full (e5 -9 %, v2 -14 %, v4 -11 %, v6 -21 % vs the row-reuse gather)
    if (n_tiles % 2 == 0 && !p.head_w) return wgs(1) / 2 >= 1024 ? 3 : 1;
    // pointwise layers (e8: 1 x 1, 256 -> 32): 32-channel K tiles — with one tap the channels are walked in the same order
    // either way (bit-identical), and the half-size tiles leave room for more workgroups per CU: 0.045 -> 0.040 ms
    if (p.T == 1 && !p.head_w) return 17;
    if (reuse) return (wgs(2) >= 1024 && p.Nw >= 7 && rowreuse_rows(p, 256) <= 64 * NPA_MAX) ? 10 : 9;
    return 1;
}

int conv_bf16_pick_ksplit(const ConvParamsH& p) {
    // per-sample geometry at a nominal batch (batch-invariant, as in the fp32 path); 128 here: this path's
    // named configuration is batch 256, where splitting v5 / v6 / d1 further costs 15-30 % of their time
    const int chunks = p.Cin / HKC;
    const long S = (long)p.Nd * p.Nh * p.Nw;
    const long wg_nom = ((128 * S + 127) / 128) * (p.CoutPad / HBN) * (p.transposed ? 8 : 1);
    int ks = 1;
    while (wg_nom * ks < 1024 && chunks % (2 * ks) == 0 && (chunks / (2 * ks)) * p.T >= 64) ks *= 2;
    return ks;
}

int64_t conv_bf16_scratch_elems(const ConvParamsH& p, int tm) {
    if (p.ksplit <= 1) return 0;
    const int t0 = tm == 23 ? 2 : tm >= 21 ? tm - 20 : (tm >= 16 ? tm - 16 : (tm >= 9 ? tm - 8 : (tm >= 5 ? tm - 4 : tm)));
    const int bm = 128 * (t0 == 3 ? 1 : t0);
    const int64_t mpad = (int64_t)((p.Ntotal + bm - 1) / bm) * bm;
    return (int64_t)(p.transposed ? 8 : 1) * p.ksplit * mpad * p.CoutPad;
}

// LDS rows (64 B each) the row-reuse kernel's A image needs for a BM-position tile, rounded to whole pieces per wave
static int rowreuse_rows(const ConvParamsH& p, int bm) {
    const int nrows = (bm + p.Nw - 2) / p.Nw + 1;
    const int r = p.stride * bm + nrows * p.kw;
    return (r + 63) / 64 * 64;
}

template <int SH, int TM>
static hipError_t launch_tm_rowreuse(ConvParamsH p, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    p.m_tiles = (p.Ntotal + BM - 1) / BM;
    p.n_tiles = p.CoutPad / HBN;
    const int r_max = rowreuse_rows(p, BM);
    if (r_max > 64 * NPA_MAX) return hipErrorInvalidValue;
    const size_t lds = (size_t)2 * (r_max * 64 + p.kw * 4096) + 2 * BM * sizeof(int) + EP_BYTES;
    static LdsAttr lds_attr;                       // (per device: see LdsAttr)
    const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_bf16r_kernel<SH, TM>), 160 * 1024);
    if (attr != hipSuccess) return attr;
    if (lds > 160 * 1024) return hipErrorInvalidValue;
    dim3 grid(p.m_tiles * p.n_tiles * (p.transposed ? 8 : 1), 1, p.ksplit);
    hipLaunchKernelGGL((conv_bf16r_kernel<SH, TM>), grid, dim3(256), lds, stream, p, r_max);
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

// LDS rows (128 B each) of the plane-reuse kernel's A image for a BM-position tile: an upper bound over all
// tiles (see conv_bf16p_kernel), in whole 1 KiB LDS-DMA pieces (8 / 16 rows); 0 = layer not eligible
static int plane_rows(const ConvParamsH& p, int bm, int kc) {
    if (p.stride != 1 || p.Cin % kc != 0 || p.x_hs % p.x_ws != 0) return 0;
    const int in_p = p.x_hs / p.x_ws, P = p.Nh * p.Nw;
    const int halo = (p.kh - 1) * in_p + p.kw - 1;
    // tiles start at multiples of bm: when rows / planes divide bm (or bm divides the plane) they are never straddled
    // (e2: 112^2 = 49 x 256; d3: 16^2 = 256; d2: 4 planes of 64, d1: 16 planes of 16 per tile)
    const int rows_touched = (bm % p.Nw == 0) ? bm / p.Nw : (bm + p.Nw - 2) / p.Nw + 1;
    const int nseg = (P % bm == 0) ? 1 : (bm % P == 0) ? bm / P : (bm + P - 2) / P + 1;
    const int r = bm + (in_p - p.Nw) * rows_touched + nseg * halo;
    const int unit = kc == 64 ? 8 : 16;            // rows per 1 KiB LDS-DMA piece
    return (r + unit - 1) / unit * unit;
}

template <int SH, int TM, int KC, int NH = 1>
static hipError_t launch_tm_plane(ConvParamsH p, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    p.m_tiles = (p.Ntotal + BM - 1) / BM;
    p.n_tiles = p.CoutPad / HBN;
    const int r_max = plane_rows(p, BM, KC);
    if (r_max == 0 || r_max > (KC == 64 ? 32 * NPA_PL : 64 * NPA_PL32) || (p.Cin / KC) % p.ksplit != 0)
        return hipErrorInvalidValue;
    if (NH == 2 && (p.n_tiles % 2 != 0 || p.head_w)) return hipErrorInvalidValue;
    const size_t lds = (size_t)r_max * KC * 2 + pl_nb(KC) * HBN * NH * KC * 2 + 2 * BM * sizeof(int) + EP_BYTES * NH;
    if (lds > 160 * 1024 || r_max * 4 > HBN * NH * KC * 2) return hipErrorInvalidValue;
    static LdsAttr lds_attr;
    dim3 grid(p.m_tiles * (p.n_tiles / NH) * (p.transposed ? 8 : 1), 1, p.ksplit);      // (class inside blockIdx.x: see the kernel)
    hipError_t e;
    if constexpr (NH == 1) {
        if (p.head_w && p.ksplit == 1) {
            static LdsAttr lds_attr_h;
            const hipError_t ah = lds_attr_h.ensure(reinterpret_cast<const void*>(&conv_bf16p_kernel<SH, TM, KC, NH, true>), 160 * 1024);
            if (ah != hipSuccess) return ah;
            hipLaunchKernelGGL((conv_bf16p_kernel<SH, TM, KC, NH, true>), grid, dim3(256), lds, stream, p, r_max);
            return hipGetLastError();
        }
    }
    if (p.head_w) return hipErrorInvalidValue;
    const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_bf16p_kernel<SH, TM, KC, NH, false>), 160 * 1024);
    if (attr != hipSuccess) return attr;
    hipLaunchKernelGGL((conv_bf16p_kernel<SH, TM, KC, NH, false>), grid, dim3(256), lds, stream, p, r_max);
    e = hipGetLastError();
    if (e == hipSuccess && p.ksplit > 1) {
        const long long total = (long long)p.Ntotal * (p.CoutPad >> 1);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(conv_finish_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), p.transposed ? 8 : 1),
                           dim3(256), 0, stream, p, p.m_tiles * BM);
        e = hipGetLastError();
    }
    return e;
}

template <int SH, int TM, int KC, bool HEAD, int NH>
static hipError_t launch_tm_k(const ConvParamsH& p, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    constexpr size_t lds = (size_t)2 * BM * KC * 2 + 2 * HBN * NH * KC * 2 + 2 * BM * sizeof(int) + EP_BYTES * NH;
    static_assert(lds <= 160 * 1024, "tile does not fit the LDS");
    if (lds > 48 * 1024) {
        static LdsAttr lds_attr;
        const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_bf16_kernel<SH, TM, KC, HEAD, NH>), (int)lds);
        if (attr != hipSuccess) return attr;
    }
    dim3 grid(p.m_tiles * (p.n_tiles / NH) * (p.transposed ? 8 : 1), 1, p.ksplit);
    hipLaunchKernelGGL((conv_bf16_kernel<SH, TM, KC, HEAD, NH>), grid, dim3(256), lds, stream, p);
    return hipGetLastError();
}

// kc32: the caller forces 32-channel K tiles (tile code + 16); otherwise 64 wherever the layer allows it.
// NH = 2: 128-cout workgroup tiles (Cout % 128 == 0, no fused head).
template <int SH, int TM, int NH>
static hipError_t launch_tm(ConvParamsH p, bool kc32, hipStream_t stream) {
    constexpr int BM = 128 * TM;
    p.m_tiles = (p.Ntotal + BM - 1) / BM;
    p.n_tiles = p.CoutPad / HBN;
    const bool head = p.head_w && p.ksplit == 1;
    if (NH == 2 && (p.n_tiles % 2 != 0 || head)) return hipErrorInvalidValue;
    const bool kc64 = !kc32 && p.Cin % 64 == 0 && (p.Cin / 64) % p.ksplit == 0;
    hipError_t e;
    if constexpr (NH == 2) {
        e = kc64 ? launch_tm_k<SH, TM, 64, false, 2>(p, stream) : launch_tm_k<SH, TM, 32, false, 2>(p, stream);
    } else {
        e = kc64 ? (head ? launch_tm_k<SH, TM, 64, true, 1>(p, stream) : launch_tm_k<SH, TM, 64, false, 1>(p, stream))
                 : (head ? launch_tm_k<SH, TM, 32, true, 1>(p, stream) : launch_tm_k<SH, TM, 32, false, 1>(p, stream));
    }
    if (e == hipSuccess && p.ksplit > 1) {
        const long long total = (long long)p.Ntotal * (p.CoutPad >> 1);
        const long long blocks = (total + 255) / 256;
        hipLaunchKernelGGL(conv_finish_bf16_kernel, dim3((unsigned)(blocks < 4096 ? blocks : 4096), p.transposed ? 8 : 1),
                           dim3(256), 0, stream, p, p.m_tiles * BM);
        e = hipGetLastError();
    }
    return e;
}

// can the row-persistent kernel serve this layer, and in how many row strips per image?  (0 = no.)  Geometry: e2's.  The
// kernel is bit-identical to the plane family, so — unlike the choice of family — this may depend on the batch: it needs
// >= 512 strips of >= 28 rows to fill 256 CUs with two work units each.
static int rows_strips(const ConvParamsH& p) {
    if (p.transposed || p.stride != 1 || p.kd != 1 || p.kh != 3 || p.kw != 3 || p.Cin != 32 || p.Nd != 1 ||
        p.Nh != 112 || p.Nw != 112 || p.x_ws != 32 || p.x_hs != 114 * 32 || p.x_org != 0 || p.ksplit != 1 || p.head_w ||
        (p.Cout & 7) != 0 || p.act == ACT_SIGMOID)
        return 0;
    for (int strips = 1; strips <= 4; strips *= 2)
        if ((long)p.B * strips >= 512) return strips;
    return 0;
}

template <int SH>
static hipError_t launch_rows(ConvParamsH p, hipStream_t stream) {
    const int strips = rows_strips(p);
    if (!strips) return hipErrorInvalidValue;
    p.n_tiles = p.CoutPad / HBN;
    p.m_tiles = 0;
    static LdsAttr lds_attr;
    const hipError_t attr = lds_attr.ensure(reinterpret_cast<const void*>(&conv_bf16w_kernel<SH>), RW_LDS);
    if (attr != hipSuccess) return attr;
    static int cus[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!cus[dev] && (hipDeviceGetAttribute(&cus[dev], hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus[dev] < 1))
        cus[dev] = 256;
    const int units = p.B * strips;
    const int per_tile = cus[dev] / p.n_tiles > 0 ? cus[dev] / p.n_tiles : 1;       // one workgroup per CU in all
    dim3 grid(units < per_tile ? units : per_tile, p.n_tiles, 1);
    hipLaunchKernelGGL((conv_bf16w_kernel<SH>), grid, dim3(64 * RW_WAVES), RW_LDS, stream, p, strips, units);
    return hipGetLastError();
}

template <int SH>
static hipError_t launch_shape(const ConvParamsH& p, int tm, hipStream_t stream) {
    // tm = 1, 2, 4: per-tap gather (conv_bf16_kernel; + 16: 32-channel K tiles even where 64 are possible);
    // tm = 9, 10: row-reuse gather (conv_bf16r_kernel) with TM 1, 2
    switch (tm) {
        case 1: case 17: return launch_tm<SH, 1, 1>(p, tm > 16, stream);
        case 2: case 18: return launch_tm<SH, 2, 1>(p, tm > 16, stream);
        case 4: case 20: return launch_tm<SH, 4, 1>(p, tm > 16, stream);
        case 3: case 19: return launch_tm<SH, 1, 2>(p, tm > 16, stream);      // 128 positions x 128 couts
        case 5: return launch_tm_plane<SH, 1, 64>(p, stream);
        case 6: return launch_tm_plane<SH, 2, 64>(p, stream);
        case 21: return launch_tm_plane<SH, 1, 32>(p, stream);
        case 22: return launch_tm_plane<SH, 2, 32>(p, stream);
        case 23: return launch_tm_plane<SH, 2, 32, 2>(p, stream);            // 256 positions x 128 couts
        case 9: return launch_tm_rowreuse<SH, 1>(p, stream);
        case 10: return launch_tm_rowreuse<SH, 2>(p, stream);
        case 40: return launch_rows<SH>(p, stream);                          // row-persistent (e2 at large batches)
        default: return hipErrorInvalidValue;
    }
}

hipError_t launch_conv_bf16(const ConvParamsH& pin, int tm, hipStream_t stream) {
    ConvParamsH p = pin;
#ifdef S3R_ABLATE
    p.debug = abl_mode_h();
#endif
    if (p.Cin % HKC != 0 || p.CoutPad % HBN != 0 || p.ksplit < 1 || (p.Cin / HKC) % p.ksplit != 0 ||
        (p.ksplit > 1 && !p.part))
        return hipErrorInvalidValue;
    return conv_bf16_shape() == 16 ? launch_shape<16>(p, tm, stream) : launch_shape<32>(p, tm, stream);
}

// ------------------------------------------------------------------------------------------------
// weight packing for the bf16 kernels (fp32 torch layout -> bf16, K-tile major, pre-swizzled):
//   wp[cls][kt = chunk*T + tap][cout tile][row 0..63][slot][8]   (64 B per row)
//     slot holds channel group kg = slot ^ swz<32>(row),  cin = chunk*32 + kg*8 + e
//     row -> cout of the 64-cout tile, so that the couts a lane ends up holding are consecutive (epilogue_h):
//       SH = 32: row = tn*32 + c; MFMA output row c sits in lane half h = (c>>2)&1, register r = (c&3) + 4*(c>>3):
//                cout = tn*32 + 16*h + r
//       SH = 16: row = ct*16 + i; MFMA output row i sits in lane group g = i>>2, register r = i&3:
//                cout = 16*g + 4*ct + r
__global__ void pack_bf16_kernel(const float* __restrict__ w, unsigned short* __restrict__ wp, int Cin, int Cout,
                                 int CoutPad, int T, int transposed, int sh) {
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
        int co;
        if (sh == 32) {
            const int tn = row >> 5, c = row & 31;
            co = tile * 64 + tn * 32 + 16 * ((c >> 2) & 1) + (c & 3) + 4 * (c >> 3);
        } else {
            const int ct = row >> 4, ii = row & 15;
            co = tile * 64 + 16 * (ii >> 2) + 4 * ct + (ii & 3);
        }
        const int kg = slot ^ swz<32>(row);
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
                       CoutPad, T, transposed, conv_bf16_shape());
    return hipGetLastError();
}

}  // namespace s3r

#ifdef S3R_ABLATE
extern "C" int s3r_debug_read_timeline_h(unsigned long long* out, int nblocks) {
    if (nblocks > 65536) nblocks = 65536;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(s3r::s3r_timeline_h), sizeof(unsigned long long) * 8 * (size_t)nblocks) == hipSuccess ? nblocks : -1;
}
#endif
