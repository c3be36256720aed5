// ConvTranspose3d(k4 s2 p1), fp32, Winograd F(2,2) on ALL THREE axes inside every output-parity class: 27 products for 2 x 2 x 2
// outputs where the two-axis form of s3r_conv_wino.hip spends 36 and the direct form 64 (27 / 64 = 0.42 of the multiplications).
//
// Along one axis (s3r_conv_wino.hip, section 2): outputs 2q and 2q + 1 of parity r read the padded inputs X0, X1, X2 = x[2q + r ..
// 2q + r + 2] through the parity's two taps g0, g1:   m0 = (X0 - X1) g0,  m1 = X1 (g0 + g1),  m2 = (X1 - X2) g1,
// y(2q) = m0 + m1,  y(2q + 1) = m1 - m2.  Nested over D, H, W that is 27 classes (a, b, c), each a 1 x 1 x 1 convolution (K = Cin)
// of its own input combination with its own weight sum.
//
// What makes it fit a CU:
//   * the D and H combinations are the two-axis form's tensors (x, Dh, Dd, Ddh: wino_diff_kernel), the W combination is formed on
//     the registers: a position is a PAIR of columns (2q, 2q + 1), a lane reads the pair P = (X0, X1) with one 8-byte LDS read and
//     X2 with a second one, and  b0 = X0 - X1, b1 = X1, b2 = X1 - X2  feed the three W classes' matrix instructions — so the three
//     classes of an (a, b) share ONE input tile (16 channels x the tile's whole padded rows), fetched once;
//   * the 27 class sums are never live together: the three W classes of an (a, b) accumulate side by side (3 tiles), their W
//     transform is folded into the H transform's running sums z[v][w] as each b ends (4 tiles), and z into the depth transform's
//     Y[v][w] as each a ends (4 tiles): depth output u = 0 is complete — and stored — after a = 1, u = 1 after a = 2.  11 tiles of
//     32 x 32 accumulators per wave (176 registers): two workgroups per CU.
// Other summation order than the other forms: its own bits (same accuracy class: +-1 transforms only), so it is an ALGORITHM in
// the sense of include/s3r.h (s3r_algo), chosen by the descriptor, never by the batch.
//
//   pack_dwino3_kernel   w[Cin][Cout][4][4][4] -> Up[pc = 8][ab = 9][m tile][chunk16][c = 3][k = 16][64 couts]
//   dwino3_kernel        grid = m tiles x position tiles x 8 parity classes; 256 threads as 2 x 2 waves; tile 64 couts x 64 positions
//                        (position = (sample, depth pair, row pair, column pair)); stage = one (a, b, 16-channel chunk): 12 KiB of
//                        weights + the input tile, LDS-DMA, three stages in flight
//                        SPLIT (class-parallel over the depth class): x 3 the workgroups, each walks ONE a and writes its H-complete
//                        sums z[v][w], raw, to the slab part[cout][tile][(pc, a, v, w)][64]; dwino3_finish_kernel forms
//                        y(0) = z0 + z1, y(1) = z1 - z2 and the epilogue with the serial form's operations in the serial form's
//                        order: the same bits, a third of the serial chain (64 workgroups of 72 stages at B = 1 otherwise)
#include "s3r_kernels.h"

namespace s3r {

typedef float d3f16 __attribute__((ext_vector_type(16)));
typedef float d3f2 __attribute__((ext_vector_type(2)));

#define S3R_LDS_PTR_D(p) ((__attribute__((address_space(3))) void*)(p))
__device__ __forceinline__ void d3dma16(__amdgpu_buffer_rsrc_t rsrc, float* lds_dst, int voffset, int soffset) {
    __builtin_amdgcn_raw_ptr_buffer_load_lds(rsrc, S3R_LDS_PTR_D(lds_dst), 16, voffset, soffset, 0, 0);
}

// AGPR residency for the running transform sums (z, Y: 8 tiles of 16).  They are touched by the folds at the end of an (a, b) only,
// never by a matrix instruction, so they live in the accumulation half of the unified register file, which the VGPR-form MFMAs of
// the K loop leave empty: 128 AGPRs + ~120 VGPRs at two waves per SIMD instead of 256 VGPRs + 72 spilled dwords (r05: the spills'
// reloads sat in one phase's DMA issue path behind s_waitcnt vmcnt(0)).  Same operations in the same order: same bits.
__device__ __forceinline__ float d3_put(float v) { float a; asm("v_accvgpr_write_b32 %0, %1" : "=a"(a) : "v"(v)); return a; }
__device__ __forceinline__ float d3_get(float a) { float v; asm("v_accvgpr_read_b32 %0, %1" : "=v"(v) : "a"(a)); return v; }

constexpr int D3_CK = 16;          // channels per stage
constexpr int D3_NST = 3;          // LDS stages
constexpr int D3_A = 3 * D3_CK * 64;                     // weight floats per stage

// geometry of the input tile for NWP column pairs per row (edge = 2 NWP): ROWS padded rows per tile, PPR 16-byte pieces per row
template <int NWP> struct D3Geo {
    static constexpr int ROWS = 64 / NWP;
    static constexpr int PPR = (2 * NWP + 1 + 3) / 4;                        // dwords 0 .. 2 NWP of the (rw-shifted) padded row
    static constexpr int RS = 4 * PPR;                                       // row stride in LDS (dwords)
    static constexpr int SLOTS = D3_CK * ROWS * PPR;                         // pieces per stage
    static constexpr int NBI = ((SLOTS + 255) / 256);                        // B DMA instructions per WAVE and stage
    static constexpr int B = NBI * 256 * 4;                                  // floats reserved per stage (whole instructions)
    static constexpr int STAGE = D3_A + B;
};

template <int NWP> int dwino3_lds_bytes() { return (D3_NST * D3Geo<NWP>::STAGE + 192 + 256) * 4; }

// p: make_params of the transposed layer; Nd = Nh = Nw = n / 2 (pairs per axis), p.x = padded input, p.xd = [Dh | Dd | Ddh],
// p.w = the 8 x 27 class slabs, p.m_tiles = ceil(Cout / 64).  HEAD: the fused 1 x 1 x 1 head (Cout <= 64): y is the head's output.
template <int NWP, bool HEAD, bool SPLIT>
__global__ __launch_bounds__(256, SPLIT ? 3 : 2) void dwino3_kernel(const ConvParams p) {
    typedef D3Geo<NWP> G;
    extern __shared__ __attribute__((aligned(16))) float smem[];
    float* const ring = smem;                                    // [D3_NST][A | B]
    float* const ep = smem + D3_NST * G::STAGE;                  // [64 scale | 64 shift | 64 head weight] [256 head exchange]

    const int tid = threadIdx.x;
    const int lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave >> 1, wn = wave & 1;
    const int j = lane & 31, h = lane >> 5;

    // workgroup -> (m tile, position tile, parity class): an XCD walks a contiguous run of (tile, class) items, the 8 classes of a
    // tile back to back (they read the same input rows)
    const int nwg = (int)gridDim.x >> 3;
    int item = ((int)blockIdx.x & 7) * nwg + ((int)blockIdx.x >> 3);
    int a0 = 0;                                                  // SPLIT: the one depth class of this workgroup
    if constexpr (SPLIT) { a0 = item % 3; item /= 3; }
    const int pc = item & 7, tile = item >> 3;
    const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
    const int m_tile = tile % p.m_tiles, n_tile = tile / p.m_tiles;
    const int m0 = m_tile * 64, n0 = n_tile * 64;
    const int S = p.Nd * p.Nh * p.Nw;                            // pair positions per sample
    const int chunks = p.Cin / D3_CK;
    const int total = (SPLIT ? 3 : 9) * chunks;

    // ---- per-lane DMA offsets of the input tile: slot s = (i * 4 + wave) * 64 + lane -> (channel k, tile row rho, piece pi)
    int bvo[G::NBI];
#pragma unroll
    for (int i = 0; i < G::NBI; ++i) {
        int s = (i * 4 + wave) * 64 + lane;
        if (s >= G::SLOTS) s -= G::SLOTS;                        // (a short slot list is padded with repeats: whole instructions)
        const int k = s / (G::ROWS * G::PPR), r = s - k * (G::ROWS * G::PPR);
        const int rho = r / G::PPR, pi = r - rho * G::PPR;
        int n = n0 + rho * NWP;
        if (n >= p.Ntotal) n = p.Ntotal - NWP;                   // tail tile: a valid row, its positions are never stored
        const int b = p.dS.div(n);
        int rem = n - b * S;
        const int pd = p.dHW.div(rem);
        const int ph = p.dW.div(rem - pd * p.Nh * p.Nw);
        bvo[i] = (b * p.Cin * p.x_cs + (2 * pd + rd) * p.x_ds + (2 * ph + rh) * p.x_hs + rw + 4 * pi + k * p.x_cs) * 4;
    }
    const int avo = lane * 16;
    const size_t x_el = (size_t)p.x_bytes / 4;
    const __amdgpu_buffer_rsrc_t wrsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<float*>(p.w), 0, (int)(8u * 27u * (unsigned)p.Cin * (unsigned)p.m_tiles * 64u * 4u), 0x00020000);

    // cursor of the NEXT stage to fetch
    int c_ab = 3 * a0, c_cc = 0;
    auto issue = [&](int buf) __attribute__((always_inline)) {
        float* sa = ring + buf * G::STAGE;
        const int wbase = (((pc * 9 + c_ab) * p.m_tiles + m_tile) * chunks + c_cc) * D3_A * 4;
#pragma unroll
        for (int i = 0; i < 3; ++i) d3dma16(wrsrc, sa + (i * 4 + wave) * 256, avo, wbase + (i * 4 + wave) * 1024);
        // source tensor of (a, b): plain / differences along D (a != 1) and H (b != 1), at depth Z + (a != 0), row R + (b != 0)
        const int a = c_ab / 3, b = c_ab - 3 * a;
        const int t_idx = (a != 1 ? 2 : 0) + (b != 1 ? 1 : 0);                  // 0: x, 1: Dh, 2: Dd, 3: Ddh
        const float* tb = t_idx == 0 ? p.x : p.xd + (size_t)(t_idx - 1) * x_el;
        const __amdgpu_buffer_rsrc_t trsrc = __builtin_amdgcn_make_buffer_rsrc(const_cast<float*>(tb), 0, (int)p.x_bytes, 0x00020000);
        const int soff = (c_cc * D3_CK * p.x_cs + (a ? p.x_ds : 0) + (b ? p.x_hs : 0)) * 4;
        float* sb = sa + D3_A;
#pragma unroll
        for (int i = 0; i < G::NBI; ++i) d3dma16(trsrc, sb + (i * 4 + wave) * 256, bvo[i], soff);
        if (++c_cc == chunks) { c_cc = 0; ++c_ab; }
    };
    constexpr int NPD = 3 + G::NBI;                              // DMAs per wave and stage

    // per-cout constants (and the head's weights) once; the ring's first two stages meanwhile
    if (tid < 64) {
        const int m = m0 + tid;
        ep[tid] = (p.scale && m < p.Cout) ? p.scale[m] : 1.f;
        ep[64 + tid] = (p.shift && m < p.Cout) ? p.shift[m] : 0.f;
        ep[128 + tid] = (HEAD && p.head_w && m < p.Cout) ? p.head_w[m] : 0.f;
    }
#pragma unroll
    for (int i = 0; i < D3_NST - 1; ++i)
        if (i < total) issue(i);
    if (total >= D3_NST - 1) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D3_NST - 2) * NPD) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    // ---- this lane's position (output addressing) and fragment addresses
    const int jt = wn * 32 + j;                                  // position within the tile
    const int rho = jt / NWP, q = jt - rho * NWP;
    const int a_off = h * 64 + wm * 32 + j;                      // A[c][k][m]: k = 2 ks + h
    const int b_off = h * G::ROWS * G::RS + rho * G::RS + 2 * q; // B[k][rho][dword]: the pair at 2q, X2 at 2q + 2
    const int n = n0 + jt;
    const bool ok = n < p.Ntotal;
    int e0;                                                      // element of output (u, v, w) = (0, 0, 0) of this position
    {
        const int nn = ok ? n : 0;
        const int b = p.dS.div(nn);
        int rem = nn - b * S;
        const int pd = p.dHW.div(rem);
        rem -= pd * p.Nh * p.Nw;
        const int ph = p.dW.div(rem);
        const int pw = rem - ph * p.Nw;
        e0 = b * p.y_bs + p.y_org + (2 * pd * p.y_ds + 2 * ph * p.y_hs + 2 * pw) * 2 + rd * p.y_ds + rh * p.y_hs + rw;
    }
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();

    d3f16 acc[3];
    float z[2][2][16], Y[2][2][16];                              // AGPR-resident (d3_put / d3_get)

    // outputs of depth u: Y[v][w] -> (2 (2 pd + u) + rd, 2 (2 ph + v) + rh, 2 (2 pw + w) + rw)
    auto store_u = [&](const int u) __attribute__((always_inline)) {
        const int mbase = wm * 32 + 4 * h;
        if constexpr (HEAD) {
            float t[2][2] = {{0.f, 0.f}, {0.f, 0.f}};
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = mbase + (r & 3) + 8 * (r >> 2);
                const float sc = ep[dm], sf = ep[64 + dm], hw = ep[128 + dm];
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int w = 0; w < 2; ++w) t[v][w] = fmaf(fmaxf(fmaf(d3_get(Y[v][w][r]), sc, sf), lo), hw, t[v][w]);
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
            float* ex = ep + 192 + jt * 4;
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w) t[v][w] += __shfl_xor(t[v][w], 32, 64);
            __syncthreads();                                     // (the exchange slots of the previous depth have been read)
            if (wm == 1 && h == 0) { ex[0] = t[0][0]; ex[1] = t[0][1]; ex[2] = t[1][0]; ex[3] = t[1][1]; }
            __syncthreads();
            if (wm == 0 && h == 0 && ok) {
                const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        float o = fmaf(t[v][w] + ex[v * 2 + w], hsc, hsf);
                        if (p.head_act == ACT_RELU) o = fmaxf(o, 0.f);
                        else if (p.head_act == ACT_SIGMOID) o = __builtin_amdgcn_rcpf(1.f + __expf(-o));
                        p.y[e0 + u * 2 * p.y_ds + v * 2 * p.y_hs + w * 2] = o;
                    }
            }
        } else {
            const __amdgpu_buffer_rsrc_t yrsrc = __builtin_amdgcn_make_buffer_rsrc(p.y, 0, (int)p.y_bytes, 0x00020000);
            const int mlimit = p.Cout - (m0 + mbase);
            const int yvo = (e0 + u * 2 * p.y_ds + (m0 + mbase) * p.y_cs) * 4;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
                if (dm >= mlimit || !ok) continue;
                const float sc = ep[mbase + dm], sf = ep[64 + mbase + dm];
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int w = 0; w < 2; ++w) {
                        const float o = fmaxf(fmaf(d3_get(Y[v][w][r]), sc, sf), lo);
                        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(unsigned, o), yrsrc, yvo + (v * 2 * p.y_hs + w * 2) * 4,
                                                              dm * p.y_cs * 4, 0);
                    }
                if ((r & 3) == 3) __builtin_amdgcn_sched_barrier(0);
            }
        }
    };

    int cur = 0, g = 0;
#pragma unroll
    for (int a = 0; a < (SPLIT ? 1 : 3); ++a) {
#pragma unroll
        for (int b = 0; b < 3; ++b) {
            // ---- the three W classes of (a, b): K = Cin in 16-channel stages (accumulators zeroed HERE, not behind the fold: they
            // are dead — and their registers free — while a depth's outputs are stored)
#pragma unroll
            for (int c = 0; c < 3; ++c)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[c][r] = 0.f;
            for (int cc = 0; cc < chunks; ++cc, ++g) {
                const bool more = g + D3_NST - 1 < total;
                if (more) issue(cur == 0 ? D3_NST - 1 : cur - 1);    // into the stage tile g - 1 was read from
                const float* A = ring + cur * G::STAGE + a_off;
                const float* B = ring + cur * G::STAGE + D3_A + b_off;
                // software pipeline over the k-steps: the fragments of step ks + 2 are requested in front of the matrix
                // instructions of step ks (two waves per SIMD do not hide an LDS round trip per step on their own)
                struct Frag { d3f2 P; float X2, a0, a1, a2; };
                auto fetch = [&](int ks) __attribute__((always_inline)) {
                    Frag f;
                    f.P = *reinterpret_cast<const d3f2*>(B + ks * 2 * G::ROWS * G::RS);
                    f.X2 = B[ks * 2 * G::ROWS * G::RS + 2];
                    f.a0 = A[ks * 128]; f.a1 = A[D3_CK * 64 + ks * 128]; f.a2 = A[2 * D3_CK * 64 + ks * 128];
                    return f;
                };
                Frag f0 = fetch(0), f1 = fetch(1);
#pragma unroll
                for (int ks = 0; ks < D3_CK / 2; ++ks) {
                    Frag f2 = f1;
                    if (ks + 2 < D3_CK / 2) f2 = fetch(ks + 2);
                    const float b0 = f0.P[0] - f0.P[1], b1 = f0.P[1], b2 = f0.P[1] - f0.X2;
                    acc[0] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a0, b0, acc[0], 0, 0, 0);
                    acc[1] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a1, b1, acc[1], 0, 0, 0);
                    acc[2] = __builtin_amdgcn_mfma_f32_32x32x2f32(f0.a2, b2, acc[2], 0, 0, 0);
                    __builtin_amdgcn_sched_barrier(0);
                    f0 = f1; f1 = f2;
                }
                if (more) asm volatile("s_waitcnt vmcnt(%0)" :: "n"((D3_NST - 2) * NPD) : "memory");
                else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
                __builtin_amdgcn_s_barrier();
                asm volatile("" ::: "memory");
                cur = cur + 1 == D3_NST ? 0 : cur + 1;
            }
            // ---- W transform of (a, b), folded into the H transform's sums z[v][w]
            if (b == 0) {
#pragma unroll
                for (int r = 0; r < 16; ++r) { z[0][0][r] = d3_put(acc[0][r] + acc[1][r]); z[0][1][r] = d3_put(acc[1][r] - acc[2][r]); }
            } else if (b == 1) {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    const float t0 = acc[0][r] + acc[1][r], t1 = acc[1][r] - acc[2][r];
                    z[0][0][r] = d3_put(d3_get(z[0][0][r]) + t0); z[0][1][r] = d3_put(d3_get(z[0][1][r]) + t1);
                    z[1][0][r] = d3_put(t0); z[1][1][r] = d3_put(t1);
                }
            } else {
#pragma unroll
                for (int r = 0; r < 16; ++r) {
                    z[1][0][r] = d3_put(d3_get(z[1][0][r]) - (acc[0][r] + acc[1][r]));
                    z[1][1][r] = d3_put(d3_get(z[1][1][r]) - (acc[1][r] - acc[2][r]));
                }
            }
        }
        if constexpr (SPLIT) {
            // ---- class-parallel: the H-complete sums of depth class a0, raw, to the slab (blocked like the other forms' slabs)
            const int n_tiles = (p.Ntotal + 63) / 64;
            const int mbase = wm * 32 + 4 * h;
            float* __restrict__ slab = p.part + (((size_t)(m0 + mbase) * n_tiles + n_tile) * 96 + (pc * 3 + a0) * 4) * 64 + jt;
            const size_t mstride = (size_t)n_tiles * 96 * 64;
#pragma unroll
            for (int r = 0; r < 16; ++r) {
                const int dm = (r & 3) + 8 * (r >> 2);
#pragma unroll
                for (int v = 0; v < 2; ++v)
#pragma unroll
                    for (int w = 0; w < 2; ++w) slab[(size_t)dm * mstride + (v * 2 + w) * 64] = d3_get(z[v][w][r]);
            }
            (void)store_u;
        } else
        // ---- H-complete sums of depth class a, folded into the depth transform
        if (a == 0) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[v][w][r] = z[v][w][r];
        } else if (a == 1) {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[v][w][r] = d3_put(d3_get(Y[v][w][r]) + d3_get(z[v][w][r]));
            store_u(0);
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[v][w][r] = z[v][w][r];   // depth output 1 starts from the same class: y(1) = z1 - z2
        } else {
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w)
#pragma unroll
                    for (int r = 0; r < 16; ++r) Y[v][w][r] = d3_put(d3_get(Y[v][w][r]) - d3_get(z[v][w][r]));
            store_u(1);
        }
    }
}

// w[Cin][Cout][4][4][4] -> Up[pc][ab][m tile][chunk16][c][k][64]: the class's weight sum over {td in Ta} x {th in Tb} x {tw in Tc},
// T0 = {0}, T1 = {0, 1}, T2 = {1}; tap t of parity r along an axis is kernel index 3 - r - 2 t
__global__ void pack_dwino3_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int Cout, int m_tiles) {
    const int chunks = Cin / D3_CK;
    const size_t total = (size_t)8 * 27 * Cin * m_tiles * 64;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int col = (int)(r % 64); r /= 64;
        const int k = (int)(r % D3_CK); r /= D3_CK;
        const int c = (int)(r % 3); r /= 3;
        const int cc = (int)(r % chunks); r /= chunks;
        const int mt = (int)(r % m_tiles); r /= m_tiles;
        const int ab = (int)(r % 9);
        const int pc = (int)(r / 9);
        const int co = mt * 64 + col, cin = cc * D3_CK + k;
        float v = 0.f;
        if (co < Cout) {
            const int a = ab / 3, b = ab - 3 * a;
            const int rd = (pc >> 2) & 1, rh = (pc >> 1) & 1, rw = pc & 1;
            const float* g = w + ((size_t)cin * Cout + co) * 64;
            auto wsum = [&](int td, int th) {
                const float* gg = g + (3 - rd - 2 * td) * 16 + (3 - rh - 2 * th) * 4;
                const float g0 = gg[3 - rw], g1 = gg[1 - rw];
                return c == 0 ? g0 : c == 2 ? g1 : g0 + g1;
            };
            auto hsum = [&](int td) { return b == 0 ? wsum(td, 0) : b == 2 ? wsum(td, 1) : wsum(td, 0) + wsum(td, 1); };
            v = a == 0 ? hsum(0) : a == 2 ? hsum(1) : hsum(0) + hsum(1);
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_dwino3(const float* w, float* wp, int Cin, int Cout, hipStream_t s) {
    hipLaunchKernelGGL(pack_dwino3_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, Cout, (Cout + 63) / 64);
    return hipGetLastError();
}

int64_t dwino3_w_elems(int Cin, int Cout) { return (int64_t)8 * 27 * Cin * ((Cout + 63) / 64) * 64; }
bool dwino3_edge_ok(int n) { return n == 8 || n == 16 || n == 32; }

// ---- class-parallel finish: the depth transform of the three slabs of each (parity class, v, w), then the serial epilogue's
// operations in its order.  Plain: one thread per (cout, pair position), the 8 parity classes in turn.  HEAD: a wave per (parity
// class, 16 positions), lane = (position, chain): the four 16-cout chains of a position (wave row wm, lane half h of the class
// kernel; r = 0 .. 15) run in four lanes, their halves' sums and then the two wave rows' sum by two shuffles — the serial kernel's
// shfl_xor 32 and LDS exchange
__device__ __forceinline__ int d3_out_elem(const ConvParams& p, int n, int S) {
    const int b = p.dS.div(n);
    int rem = n - b * S;
    const int pd = p.dHW.div(rem);
    rem -= pd * p.Nh * p.Nw;
    const int ph = p.dW.div(rem);
    const int pw = rem - ph * p.Nw;
    return b * p.y_bs + p.y_org + (2 * pd * p.y_ds + 2 * ph * p.y_hs + 2 * pw) * 2;
}

__global__ __launch_bounds__(256) void dwino3_finish_kernel(const ConvParams p, const int n_tiles) {
    const int S = p.Nd * p.Nh * p.Nw;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const size_t mstride = (size_t)n_tiles * 96 * 64;
    const long long total = (long long)p.Cout * p.Ntotal;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const int m = (int)(i / p.Ntotal), n = (int)(i - (long long)m * p.Ntotal);
        const int e00 = d3_out_elem(p, n, S);
        const float sc = p.scale ? p.scale[m] : 1.f, sf = p.shift ? p.shift[m] : 0.f;
        const float* __restrict__ blk = p.part + (size_t)m * mstride + ((size_t)(n >> 6) * 96) * 64 + (n & 63);
#pragma unroll 1
        for (int pc = 0; pc < 8; ++pc) {
            const float* __restrict__ zz = blk + (size_t)(pc * 12) * 64;
            float* __restrict__ yo = p.y + (size_t)m * p.y_cs + e00 + ((pc >> 2) & 1) * p.y_ds + ((pc >> 1) & 1) * p.y_hs + (pc & 1);
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const float z0 = zz[(0 * 4 + v * 2 + w) * 64], z1 = zz[(1 * 4 + v * 2 + w) * 64], z2 = zz[(2 * 4 + v * 2 + w) * 64];
                    yo[v * 2 * p.y_hs + w * 2] = fmaxf(fmaf(z0 + z1, sc, sf), lo);
                    yo[2 * p.y_ds + v * 2 * p.y_hs + w * 2] = fmaxf(fmaf(z1 - z2, sc, sf), lo);
                }
        }
    }
}

__global__ __launch_bounds__(256) void dwino3_finish_head_kernel(const ConvParams p, const int n_tiles) {
    const int S = p.Nd * p.Nh * p.Nw;
    const float lo = p.act == ACT_RELU ? 0.f : -__builtin_inff();
    const size_t mstride = (size_t)n_tiles * 96 * 64;
    const int lane = threadIdx.x & 63;
    const int nb16 = (p.Ntotal + 15) / 16;
    const long long waves = (long long)nb16 * 8;
    for (long long q = (long long)blockIdx.x * 4 + (threadIdx.x >> 6); q < waves; q += (long long)gridDim.x * 4) {
        const int pc = (int)(q & 7), n = (int)(q >> 3) * 16 + (lane & 15);
        const int g = lane >> 4;                                 // chain: wave row wm = g >> 1, lane half h = g & 1
        const bool ok = n < p.Ntotal;
        const int nn = ok ? n : p.Ntotal - 1;
        const float* __restrict__ blk = p.part + ((size_t)(nn >> 6) * 96 + pc * 12) * 64 + (nn & 63);
        float t[2][2][2];                                        // [u][v][w]
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w) t[u][v][w] = 0.f;
#pragma unroll 4
        for (int r = 0; r < 16; ++r) {
            const int dm = (g >> 1) * 32 + 4 * (g & 1) + (r & 3) + 8 * (r >> 2);
            const float sc = (p.scale && dm < p.Cout) ? p.scale[dm] : 1.f, sf = (p.shift && dm < p.Cout) ? p.shift[dm] : 0.f;
            const float hw = (p.head_w && dm < p.Cout) ? p.head_w[dm] : 0.f;
            const float* __restrict__ zz = blk + (size_t)dm * mstride;
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    const float z0 = zz[(0 * 4 + v * 2 + w) * 64], z1 = zz[(1 * 4 + v * 2 + w) * 64], z2 = zz[(2 * 4 + v * 2 + w) * 64];
                    t[0][v][w] = fmaf(fmaxf(fmaf(z0 + z1, sc, sf), lo), hw, t[0][v][w]);
                    t[1][v][w] = fmaf(fmaxf(fmaf(z1 - z2, sc, sf), lo), hw, t[1][v][w]);
                }
        }
        const float hsc = p.head_scale ? p.head_scale[0] : 1.f, hsf = p.head_shift ? p.head_shift[0] : 0.f;
        const int e0 = d3_out_elem(p, nn, S) + ((pc >> 2) & 1) * p.y_ds + ((pc >> 1) & 1) * p.y_hs + (pc & 1);
#pragma unroll
        for (int u = 0; u < 2; ++u)
#pragma unroll
            for (int v = 0; v < 2; ++v)
#pragma unroll
                for (int w = 0; w < 2; ++w) {
                    float s = t[u][v][w];
                    s += __shfl_xor(s, 16, 64);                  // the lane halves' sum (h = 0 + h = 1)
                    s += __shfl_xor(s, 32, 64);                  // wave row 0's + wave row 1's
                    float o = fmaf(s, hsc, hsf);
                    if (p.head_act == ACT_RELU) o = fmaxf(o, 0.f);
                    else if (p.head_act == ACT_SIGMOID) o = __builtin_amdgcn_rcpf(1.f + __expf(-o));
                    if (g == 0 && ok) p.y[e0 + u * 2 * p.y_ds + v * 2 * p.y_hs + w * 2] = o;
                }
    }
}

int64_t dwino3_slab_elems(int cout, int ntotal) { return (int64_t)((cout + 63) / 64) * 64 * ((ntotal + 63) / 64) * 96 * 64; }

// class-parallel when the serial form's grid leaves half the CUs without a workgroup (d3: B <= 2); forced: 0 serial, 1
// class-parallel, < 0 the rule (S3R_WINO_FORM, read once, overrides it like the other forms' plans)
bool dwino3_split(int cout, int ntotal, int forced) {
    if (forced < 0) {
        static const int env_mode = getenv("S3R_WINO_FORM") ? atoi(getenv("S3R_WINO_FORM")) : -1;      // A/B switch, read once
        forced = env_mode == 0 || env_mode == 1 ? env_mode : -1;
    }
    if (forced >= 0) return forced == 1;
    const long W = (long)((cout + 63) / 64) * ((ntotal + 63) / 64) * 8;
    return 2 * W <= cu_count();          // (d3: B <= 2 — 0.101 -> 0.053, 0.104 -> 0.070 ms; B = 4: 0.110 -> 0.120)
}

template <int NWP>
static hipError_t launch_dwino3_t(const ConvParams& p, bool split, hipStream_t stream) {
    const int n_tiles = (p.Ntotal + 63) / 64;
    const dim3 grid(p.m_tiles * n_tiles * 8 * (split ? 3 : 1));
    const int lds = dwino3_lds_bytes<NWP>();
    static LdsAttr attr_h, attr_p, attr_s;
    if (split) {
        hipError_t e = attr_s.ensure(reinterpret_cast<const void*>(&dwino3_kernel<NWP, false, true>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwino3_kernel<NWP, false, true>), grid, dim3(256), lds, stream, p);
        if ((e = hipGetLastError()) != hipSuccess) return e;
        // (aux pass: reads the 96 slab rows of every position, writes the layer's — or the head's — output)
        AuxScope aux(stream, 4.0 * (double)p.Ntotal * (96.0 * p.Cout + 64.0 * (p.head_w ? 1 : p.Cout)));
        const long long blocks = p.head_w ? ((long long)((p.Ntotal + 15) / 16) * 8 + 3) / 4 : ((long long)p.Cout * p.Ntotal + 255) / 256;
        const dim3 fgrid((unsigned)(blocks < 32768 ? blocks : 32768));
        if (p.head_w) hipLaunchKernelGGL(dwino3_finish_head_kernel, fgrid, dim3(256), 0, stream, p, n_tiles);
        else hipLaunchKernelGGL(dwino3_finish_kernel, fgrid, dim3(256), 0, stream, p, n_tiles);
    } else if (p.head_w) {
        hipError_t e = attr_h.ensure(reinterpret_cast<const void*>(&dwino3_kernel<NWP, true, false>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwino3_kernel<NWP, true, false>), grid, dim3(256), lds, stream, p);
    } else {
        hipError_t e = attr_p.ensure(reinterpret_cast<const void*>(&dwino3_kernel<NWP, false, false>), lds);
        if (e != hipSuccess) return e;
        hipLaunchKernelGGL((dwino3_kernel<NWP, false, false>), grid, dim3(256), lds, stream, p);
    }
    return hipGetLastError();
}

// p: the transposed layer's parameters with Nd = Nh = Nw = n / 2 and Ntotal = B (n / 2)^3 pair positions; p.xd = [Dh | Dd | Ddh]
hipError_t launch_deconv_wino3(ConvParams p, bool split, hipStream_t stream) {
    if (p.Cin % D3_CK != 0 || !p.transposed || !p.xd || p.act == ACT_SIGMOID || (p.head_w && p.Cout > 64) || p.Nd != p.Nw || p.Nh != p.Nw ||
        (split && !p.part))
        return hipErrorInvalidValue;
    p.m_tiles = (p.Cout + 63) / 64;
    switch (p.Nw) {
        case 4: return launch_dwino3_t<4>(p, split, stream);
        case 8: return launch_dwino3_t<8>(p, split, stream);
        case 16: return launch_dwino3_t<16>(p, split, stream);
        default: return hipErrorInvalidValue;
    }
}

}  // namespace s3r
