// Bandwidth-bound kernels of the bf16 channels-last path (BASELINE.json configs[2]): the Cin=3 stem
// (fp32 NCHW renders -> bf16 NHWC features), the fused bidirectional cost volume on channels-last bf16
// features, and the 1x1x1 occupancy head (bf16 NDHWC -> fp32 (B,32,32,32) probabilities).
#include "s3r_kernels.h"
#include <cstdlib>

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
//   TI = float | unsigned char (8-bit renders, scaled by 1/255 on the way in: render_f32)
template <typename TI>
__global__ __launch_bounds__(256) void stem_bf16_kernel(const TI* __restrict__ x, const TI* __restrict__ x2, int nsplit,
                                                        const float* __restrict__ wt,
                                                        const float* __restrict__ scale, const float* __restrict__ shift,
                                                        unsigned short* __restrict__ y, int N, int Hi, int Wi, int Ho, int Wo,
                                                        int y_bs, int y_hs, int y_org) {
    __shared__ __attribute__((aligned(16))) v4u st[256 * 4];      // [pixel][4 x 16 B], part q at slot q ^ ((pixel >> 1) & 3)
    __shared__ long long yo[256];                                  // output element offset of each pixel, -1 = none
    const int HWo = Ho * Wo;
    const int tid = threadIdx.x;
    const long long gid = (long long)blockIdx.x * 256 + tid;
    const bool live = gid < (long long)N * HWo;
    const long long g = live ? gid : (long long)N * HWo - 1;
    const int n = (int)(g / HWo);
    const int sp = (int)(g - (long long)n * HWo);
    const int oh = sp / Wo, ow = sp - oh * Wo;
    const int ih0 = oh * 2 - 1, iw0 = ow * 2 - 1;
    const TI* __restrict__ xn = n < nsplit ? x + (size_t)n * 3 * Hi * Wi : x2 + (size_t)(n - nsplit) * 3 * Hi * Wi;
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
    float acc[32];
#pragma unroll
    for (int c = 0; c < 16; ++c) { acc[2 * c] = acc2[c].x; acc[2 * c + 1] = acc2[c].y; }
    // (64-bit: n * y_bs passes 2^31 elements beyond ~5160 images, and this form has no image-count limit of its own)
    yo[tid] = live ? (long long)((size_t)n * y_bs + y_org + ((size_t)oh * y_hs + (size_t)ow * 32)) : -1ll;
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
        const long long off = yo[pix];
        const v4u t = st[pix * 4 + (q ^ ((pix >> 1) & 3))];
        if (off >= 0) *reinterpret_cast<v4u*>(y + (size_t)off + q * 8) = t;       // 64-byte pixels: 16-byte aligned
    }
}

// ------------------------------------------------------------------------------------------------
// The same stem on the matrix cores, rows staged through LDS.  What the VALU form above is bound by is not its 432
// packed FMAs per pixel but its memory pattern: 27 stride-2 dword gathers per pixel-thread (a diagnostic build with the
// stores removed still took 142 us at B = 256 for 308 MB of renders: 2.2 TB/s) and a store path that waits behind them
// (0.21 ms in all, 3.4 TB/s).  Here a workgroup of 7 waves owns 4 output rows of one image:
//   * the 9 input rows x 3 channels it needs arrive by LDS-DMA, one whole 896-byte row per wave-instruction (56 lanes
//     x 16 B, contiguous in HBM and in LDS) into 1-KiB LDS slots whose last 128 bytes stay zero — which is also the
//     zero a tap at column -1 must read (it is the tail of the slot before); the row above the image is zero-filled;
//     two such slabs: the rows of the next pass are in flight while this one is computed (counted vmcnt, raw barriers);
//   * a wave computes 32 pixels x 32 couts as D[cout][pixel] with v_mfma_f32_32x32x2_f32 over K = 28: tap k = ci*9 +
//     kh*3 + kw for k < 27 (weights x folded-BN scale) and a constant-one column whose weight is the folded-BN shift —
//     exact fp32 FMA chains, no VALU arithmetic at all before the ReLU; lane (j = l & 31, h = l >> 5) feeds pixel j /
//     cout j at k = 2 s + h, one ds_read_b32 per MFMA;
//   * the 32 x 64-byte result goes through a wave-private LDS slab so that every store instruction writes 16 whole
//     pixels = 1 KiB contiguous.
// 14 MFMAs of 64 cycles per tile: 73 us of matrix time at B = 256 under 145 us of HBM time; measured 156 us (4.6 TB/s,
// VALU form 210).  tools/timeline_stem.py (diagnostic build) on what is left: a pass takes 6.1 us per workgroup —
// issuing the next rows 0.7, waiting for this pass's 0.25, barrier 0.3, 28 operand reads 0.4, the two MFMA chains
// 1.4, ReLU + LDS slab + stores 0.85 per tile, closing barrier 0.3 — with 2 workgroups (14 waves) per CU the matrix
// pipe is busy 48 % of the time and no unit more than that: what is left is the serial chain inside a pass.
// ReLU as ONE instruction the compiler knows (so the MFMA -> VALU read hazard is its to cover): fmaxf on an MFMA result
// gets a canonicalising v_max_f32 x, x in front of it; as signed integers every negative float (and -0) is below zero and
// every positive float is itself, so v_max_i32(bits, 0) is the same function
__device__ __forceinline__ float relu1(float v) {
    const int b = __builtin_bit_cast(int, v);
    return __builtin_bit_cast(float, b > 0 ? b : 0);
}

constexpr int STEM_ROWS = 4;                 // output rows per workgroup pass
constexpr int STEM_SLOTS = 3 * (2 * STEM_ROWS + 1);
constexpr int STEM_WAVES = 7;                // 4 rows x 112 pixels = 14 tiles of 32: two per wave
constexpr int STEM_SLABS = 2;                // row slabs: 2 = the next pass's rows in flight during this one (0.156 ms at B = 256);
                                             // 1 = this pass's own rows, waited for in full (0.172: registers, not LDS, hold it to 2 workgroups/CU too)
constexpr int STEM_LDS_BYTES = 1024 + STEM_SLABS * STEM_SLOTS * 1024 + STEM_WAVES * 2048;  // guard + row slabs + store slabs

#ifdef S3R_ABLATE   // diagnostic builds: S3R_ABL = 32 + bits: 1 no global stores, 2 no row DMA (after the first pass), 4 one MFMA per tile
#define S3R_STEM_DBG_PARAM , int dbg
#define S3R_STEM_DBG(bit) (dbg >= 32 && (dbg & (bit)))
#else
#define S3R_STEM_DBG_PARAM
#define S3R_STEM_DBG(m) false
#endif
#ifdef S3R_ABLATE
// per workgroup (first 4096), wave 0, pass 5: [pass start, rows issued, rows landed (before barrier), after barrier,
// tile 1 done, tile 2 done, after the closing barrier, operands read (landed), MFMAs of both tiles issued+done, ...]
__device__ unsigned long long s3r_stem_timeline[4096 * 16];
#define S3R_STEM_STAMP(k) do { if (dbg >= 32 && stamp_pass == 5 && lane == 0 && wave == 0 && blockIdx.x < 4096) \
        s3r_stem_timeline[blockIdx.x * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define S3R_STEM_STAMP(k) do {} while (0)
#endif
// U8: the renders are 8-bit (N,3,224,224) uint8 — what the reference's PNG decode yields and a quarter of the bytes: a row
// is 224 bytes = 14 lanes x 16 B per DMA, slots are 256 B apart (32 zero bytes behind each row), and an operand is read as
// one byte and scaled by 1/255 in registers (render_f32: the same single rounding as the host's float32(u) / 255, so
// the result is bit-identical to the fp32 entry on renders converted on the host).
template <bool U8>
__global__ __launch_bounds__(64 * STEM_WAVES) void stem_bf16_mfma_kernel(
        const void* __restrict__ xv, const void* __restrict__ x2v, int nsplit, const float* __restrict__ wt,
        const float* __restrict__ scale, const float* __restrict__ shift, unsigned short* __restrict__ y, int N,
        int y_bs, int y_hs, int y_org S3R_STEM_DBG_PARAM) {
    typedef float f32x16 __attribute__((ext_vector_type(16)));
    typedef unsigned v2u __attribute__((ext_vector_type(2)));
    constexpr int Hi = 224, Wi = 224, Wo = 112, BPI = 112 / STEM_ROWS;       // the network's stem geometry (launcher checks)
    constexpr int ES = U8 ? 1 : 4;                                            // bytes per render sample
    constexpr int ROWB = Wi * ES;                                             // 896 | 224
    constexpr int SLOT = U8 ? 256 : 1024;                                     // bytes between row slots (tail: zeros)
    const unsigned char* x = static_cast<const unsigned char*>(xv);
    const unsigned char* x2 = static_cast<const unsigned char*>(x2v);
    extern __shared__ __attribute__((aligned(16))) unsigned char slds[];
    unsigned char* slab = slds + 1024;                                        // slot sl at slab + SLOT sl; 1 KiB of zeros before slot 0
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int j = lane & 31, h = lane >> 5;
    v2u* sw = reinterpret_cast<v2u*>(slab + STEM_SLABS * STEM_SLOTS * SLOT + wave * 2048);  // [pixel 32][8 x 8 B], 16-B slot g at g ^ ((pixel >> 1) & 3)
    const v4u* sr = reinterpret_cast<const v4u*>(sw);

    // ---- once: zero guard + slot tails, weight fragments (A operand: cout j, k = 2 s + h), tap offsets (B operand)
    constexpr int TAILW = (SLOT - ROWB) / 4;                                  // zero words behind each row: 32 | 8
    for (int i = threadIdx.x; i < 256 + STEM_SLABS * STEM_SLOTS * TAILW; i += 64 * STEM_WAVES) {
        if (i < 256) reinterpret_cast<unsigned*>(slds)[i] = 0u;
        else { const int sl = (i - 256) / TAILW, w = (i - 256) % TAILW; reinterpret_cast<unsigned*>(slab + sl * SLOT + ROWB)[w] = 0u; }
    }
    float wa[14];
    int off[14];
    {
        const float sc = scale ? scale[j] : 1.f;
#pragma unroll
        for (int s = 0; s < 14; ++s) {
            const int k = 2 * s + h;
            const int t = k < 27 ? k : 0;
            const int ci = t / 9, kh = (t - ci * 9) / 3, kw = t - ci * 9 - kh * 3;
            off[s] = (ci * (2 * STEM_ROWS + 1) + kh) * SLOT + kw * ES;
            wa[s] = k < 27 ? wt[t * 32 + j] * sc : (shift ? shift[j] : 0.f);
        }
    }
    const int sj = (j >> 1) & 3;

    // ---- rows of pass `blk` into slab `sb`: slot sl = ci*9 + rr holds input row 8 q - 1 + rr of channel ci.  Returns the
    // number of DMA instructions this wave issued (wave-uniform)
    auto issue = [&](int blk, unsigned char* sb) -> int {
        const int n = blk / BPI, q = blk - n * BPI;
        const unsigned char* xn = n < nsplit ? x + (size_t)n * 3 * Hi * Wi * ES : x2 + (size_t)(n - nsplit) * 3 * Hi * Wi * ES;
        const __amdgpu_buffer_rsrc_t rs = __builtin_amdgcn_make_buffer_rsrc(const_cast<unsigned char*>(xn), 0, 3 * Hi * Wi * ES, 0x00020000);
        int cnt = 0;
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const int sl = wave * 4 + u;
            if (sl >= STEM_SLOTS) break;
            const int ci = sl / (2 * STEM_ROWS + 1), rr = sl - ci * (2 * STEM_ROWS + 1);
            const int irow = 2 * STEM_ROWS * q - 1 + rr;
            if (irow < 0) {                                                    // the row above the image
                if (lane < ROWB / 16) *reinterpret_cast<v4u*>(sb + sl * SLOT + lane * 16) = (v4u){0u, 0u, 0u, 0u};
            } else if (!(S3R_STEM_DBG(2) && blk >= (int)gridDim.x)) {
                if (lane < ROWB / 16)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (__attribute__((address_space(3))) void*)(sb + sl * SLOT), 16,
                                                             lane * 16, (ci * Hi + irow) * ROWB, 0, 0);
                ++cnt;
            }
        }
        return cnt;
    };

    const int nblk = N * BPI;
    int blk = blockIdx.x;
    unsigned char* cur = slab;
    unsigned char* oth = slab + (STEM_SLABS - 1) * STEM_SLOTS * SLOT;
    if (STEM_SLABS == 2 && blk < nblk) issue(blk, cur);
#ifdef S3R_ABLATE
    int stamp_pass = -1;
#endif
    for (; blk < nblk; blk += gridDim.x) {
#ifdef S3R_ABLATE
        ++stamp_pass;
#endif
        S3R_STEM_STAMP(0);
        // ---- the next pass's rows go into the other slab while this one is computed; then wait for THIS pass's rows.
        // vmcnt retires in issue order (loads and stores alike), and what was issued after this pass's DMAs is the
        // previous pass's 4 stores of this wave and the DMAs just issued: they may stay in flight — waiting for the
        // stores' acknowledgements here cost 5 us per pass
        const int nxt = blk + gridDim.x;
        int nn = 0;
        if (STEM_SLABS == 1) issue(blk, cur);                   // (one slab: this pass's own rows, waited for in full)
        else nn = (nxt < nblk ? issue(nxt, oth) : 0) + ((blk != (int)blockIdx.x && !S3R_STEM_DBG(1)) ? 4 : 0);
        S3R_STEM_STAMP(1);
        switch (nn) {
            case 0: asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory"); break;
            case 1: asm volatile("s_waitcnt vmcnt(1) lgkmcnt(0)" ::: "memory"); break;
            case 2: asm volatile("s_waitcnt vmcnt(2) lgkmcnt(0)" ::: "memory"); break;
            case 3: asm volatile("s_waitcnt vmcnt(3) lgkmcnt(0)" ::: "memory"); break;
            case 4: asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory"); break;
            case 5: asm volatile("s_waitcnt vmcnt(5) lgkmcnt(0)" ::: "memory"); break;
            case 6: asm volatile("s_waitcnt vmcnt(6) lgkmcnt(0)" ::: "memory"); break;
            case 7: asm volatile("s_waitcnt vmcnt(7) lgkmcnt(0)" ::: "memory"); break;
            default: asm volatile("s_waitcnt vmcnt(8) lgkmcnt(0)" ::: "memory"); break;
        }
        S3R_STEM_STAMP(2);
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        S3R_STEM_STAMP(3);
        const int n = blk / BPI, q = blk - n * BPI;
        // ---- two tiles per wave.  All 28 operand reads go out before the first MFMA (left to itself the compiler
        // pairs every ds_read with the MFMA that consumes it: 14 exposed LDS latencies per tile, 1.7 us instead of 0.6)
        float bv[2][14];
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int px = (wave + k * STEM_WAVES) * 32 + j;
            const int orow = px / Wo, ow = px - orow * Wo;
            const unsigned char* bp = cur + (2 * orow) * SLOT + (2 * ow - 1) * ES;
            if constexpr (U8) {
                unsigned char raw[14];
#pragma unroll
                for (int s = 0; s < 14; ++s) raw[s] = bp[off[s]];
#pragma unroll
                for (int s = 0; s < 14; ++s) bv[k][s] = render_f32(raw[s]);
            } else {
#pragma unroll
                for (int s = 0; s < 14; ++s) bv[k][s] = *reinterpret_cast<const float*>(bp + off[s]);
            }
        }
        __builtin_amdgcn_sched_barrier(0);
#ifdef S3R_ABLATE
        if (dbg >= 32) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); S3R_STEM_STAMP(7); }
#endif
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            const int tt = wave + k * STEM_WAVES;
            f32x16 acc;
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[r] = 0.f;
#pragma unroll
            for (int s = 0; s < 14; ++s) {
                float b = bv[k][s];
                if (s == 13) b = h ? 1.f : b;                                  // k = 27: the constant-one column
                if (S3R_STEM_DBG(4) && s > 0) { acc[s] += b; continue; }
                acc = __builtin_amdgcn_mfma_f32_32x32x2f32(wa[s], b, acc, 0, 0, 0);
            }
#ifdef S3R_ABLATE
            if (dbg >= 32) { asm volatile("s_nop 0" :: "v"(acc[0])); S3R_STEM_STAMP(8 + 4 * k); }
#endif
            // ReLU, bf16: lane (j, h) owns bytes [16 g + 8 h, +8) of pixel j, g = 0..3  (couts 8 g + 4 h + 0..3)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                v2u t;
                t[0] = pack2(relu1(acc[4 * g]), relu1(acc[4 * g + 1]));
                t[1] = pack2(relu1(acc[4 * g + 2]), relu1(acc[4 * g + 3]));
                sw[j * 8 + ((g ^ sj) << 1) + h] = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
#ifdef S3R_ABLATE
            if (dbg >= 32) { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); S3R_STEM_STAMP(9 + 4 * k); }
#endif
#pragma unroll
            for (int i = 0; i < 2; ++i) {
                const int pix = 16 * i + (lane >> 2), qq = lane & 3;
                const int ppx = tt * 32 + pix;
                const int prow = ppx / Wo, pw = ppx - prow * Wo;
                const v4u t = sr[pix * 4 + (qq ^ ((pix >> 1) & 3))];
                const size_t o = (size_t)n * y_bs + y_org + (size_t)(STEM_ROWS * q + prow) * y_hs + pw * 32 + qq * 8;
                if (!S3R_STEM_DBG(1) || t[0] == 0x12345678u) *reinterpret_cast<v4u*>(y + o) = t;
            }
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
            __builtin_amdgcn_wave_barrier();
            S3R_STEM_STAMP(4 + k);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                                          // this slab is free for the pass after next
        asm volatile("" ::: "memory");
        S3R_STEM_STAMP(6);
        unsigned char* t = cur; cur = oth; oth = t;
    }
}

static bool stem_mfma_enabled() {
    static const bool on = !(getenv("S3R_STEM_MFMA") && atoi(getenv("S3R_STEM_MFMA")) == 0);   // A/B switch
    return on;
}

template <bool U8>
static hipError_t launch_stem_bf16_mfma(const void* x, const void* x2, int nsplit, const float* wt, const float* scale,
                                        const float* shift, void* y, int N, int y_bs, int y_hs, int y_org, hipStream_t s) {
    // persistent workgroups, every one of them resident at once (as many as the occupancy query says fit)
    static LdsAttr attr;
    hipError_t e = attr.ensure(reinterpret_cast<const void*>(&stem_bf16_mfma_kernel<U8>), STEM_LDS_BYTES);
    if (e != hipSuccess) return e;
    static int resident[16] = {0};
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= 16) return hipErrorInvalidDevice;
    if (!resident[dev]) {
        int per_cu = 0, cus = 0;
        if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, stem_bf16_mfma_kernel<U8>, 64 * STEM_WAVES, STEM_LDS_BYTES) !=
                hipSuccess || per_cu < 1)
            per_cu = 2;
        if (hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || cus < 1) cus = 256;
        resident[dev] = per_cu * cus;
    }
    const int blocks = N * (112 / STEM_ROWS);
    int wgs = blocks < resident[dev] ? blocks : resident[dev];
    hipLaunchKernelGGL(stem_bf16_mfma_kernel<U8>, dim3((unsigned)wgs), dim3(64 * STEM_WAVES), STEM_LDS_BYTES, s, x, x2, nsplit,
                       wt, scale, shift, reinterpret_cast<unsigned short*>(y), N, y_bs, y_hs, y_org
#ifdef S3R_ABLATE
                       , getenv("S3R_ABL") ? atoi(getenv("S3R_ABL")) : 0
#endif
                       );
    return hipGetLastError();
}

hipError_t launch_stem_bf16(const void* x, const void* x2, int u8, int nsplit, const float* wt, const float* scale,
                            const float* shift, void* y, int N, int Hi, int Wi, int Ho, int Wo, int y_bs, int y_hs, int y_org,
                            hipStream_t s) {
    const long long total = (long long)N * Ho * Wo;
    if (!x2) { x2 = x; nsplit = N; }
    if (stem_mfma_enabled() && Hi == 224 && Wi == 224 && Ho == 112 && Wo == 112 && N > 0 && N <= (1 << 16))
        return u8 ? launch_stem_bf16_mfma<true>(x, x2, nsplit, wt, scale, shift, y, N, y_bs, y_hs, y_org, s)
                  : launch_stem_bf16_mfma<false>(x, x2, nsplit, wt, scale, shift, y, N, y_bs, y_hs, y_org, s);
    const dim3 grid((unsigned)((total + 255) / 256));
    if (u8)
        hipLaunchKernelGGL(stem_bf16_kernel<unsigned char>, grid, dim3(256), 0, s, static_cast<const unsigned char*>(x),
                           static_cast<const unsigned char*>(x2), nsplit, wt, scale, shift, reinterpret_cast<unsigned short*>(y),
                           N, Hi, Wi, Ho, Wo, y_bs, y_hs, y_org);
    else
        hipLaunchKernelGGL(stem_bf16_kernel<float>, grid, dim3(256), 0, s, static_cast<const float*>(x),
                           static_cast<const float*>(x2), nsplit, wt, scale, shift, reinterpret_cast<unsigned short*>(y),
                           N, Hi, Wi, Ho, Wo, y_bs, y_hs, y_org);
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

// ------------------------------------------------------------------------------------------------
// Hand-off from the bf16 path to an fp32 consumer (the point head's linear layers, the disparity read-out): channels-last
// bf16 (N, S, C) -> fp32 NC(S) (N, C, S), exact (a bf16 is the top half of an fp32).  32 x 32 tiles through LDS: reads are
// 64-byte channel runs, writes 128-byte position runs.
__global__ __launch_bounds__(256) void cl_bf16_to_f32_kernel(const unsigned short* __restrict__ x, float* __restrict__ y,
                                                             int S, int C) {
    __shared__ float tile[32][33];
    const int n = blockIdx.z, s0 = blockIdx.x * 32, c0 = blockIdx.y * 32;
    const int tx = threadIdx.x & 31, ty = threadIdx.x >> 5;
    const unsigned short* __restrict__ xn = x + (size_t)n * S * C;
    float* __restrict__ yn = y + (size_t)n * C * S;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int s = s0 + ty + 8 * i, c = c0 + tx;
        tile[ty + 8 * i][tx] = (s < S && c < C) ? __builtin_bit_cast(float, (unsigned)xn[(size_t)s * C + c] << 16) : 0.f;
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int c = c0 + ty + 8 * i, sp = s0 + tx;
        if (sp < S && c < C) yn[(size_t)c * S + sp] = tile[tx][ty + 8 * i];
    }
}

hipError_t launch_cl_bf16_to_f32(const void* x, float* y, int N, int64_t S, int C, hipStream_t s) {
    if (N <= 0 || S <= 0 || C <= 0 || N > 65535 || (C + 31) / 32 > 65535) return hipErrorInvalidValue;
    const dim3 grid((unsigned)((S + 31) / 32), (unsigned)((C + 31) / 32), (unsigned)N);
    hipLaunchKernelGGL(cl_bf16_to_f32_kernel, grid, dim3(256), 0, s, reinterpret_cast<const unsigned short*>(x), y, (int)S, C);
    return hipGetLastError();
}

}  // namespace s3r

#ifdef S3R_ABLATE
extern "C" int s3r_debug_read_stem_timeline(unsigned long long* out, int nblocks) {
    if (nblocks > 4096) nblocks = 4096;
    return hipMemcpyFromSymbol(out, HIP_SYMBOL(s3r::s3r_stem_timeline), sizeof(unsigned long long) * 16 * (size_t)nblocks) == hipSuccess ? nblocks : -1;
}
#endif
