// Internal (non-ABI) declarations shared by the HIP translation units of libs3r_hip.so.
// The public C-ABI is include/s3r.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <atomic>

namespace s3r {

// hipFuncSetAttribute(MaxDynamicSharedMemorySize) applies to the CURRENT device only: remember per (kernel
// instantiation, device), so a process that drives several devices raises the limit on each of them (a process-wide
// "done once" flag left every device but the first at the 48 KiB default: a silent launch failure there).
struct LdsAttr {
    std::atomic<unsigned long long> done{0};          // bit d: raised on device d (devices >= 64: set every time)
    hipError_t ensure(const void* fn, int bytes) {
        int dev = 0;
        hipError_t e = hipGetDevice(&dev);
        if (e != hipSuccess) return e;
        if (dev >= 0 && dev < 64 && ((done.load(std::memory_order_relaxed) >> dev) & 1ull)) return hipSuccess;
        e = hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
        if (e == hipSuccess && dev >= 0 && dev < 64) done.fetch_or(1ull << dev, std::memory_order_relaxed);
        return e;
    }
};

// Compute units of the current device (256 on an unpartitioned MI355X; fewer in the CPX / partitioned modes): what the launch
// planners (plan_tail_cut, wino_plan, wino2_form) count workgroup rounds against.  Read once per device; 256 where no device
// answers (the host-only size queries of a CPU-only box, so that they plan what a full MI355X would).
inline int cu_count() {
    static std::atomic<int> cached[64];
    int dev = 0;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0) { (void)hipGetLastError(); return 256; }
    int n = dev < 64 ? cached[dev].load(std::memory_order_relaxed) : 0;      // (a device beyond the table is asked every time: never another's answer)
    if (n > 0) return n;
    n = 0;
    if (hipDeviceGetAttribute(&n, hipDeviceAttributeMultiprocessorCount, dev) != hipSuccess || n <= 0) { (void)hipGetLastError(); n = 256; }
    if (dev < 64) cached[dev].store(n, std::memory_order_relaxed);
    return n;
}

// Profiler hook for the launchers (s3r_prof.hip): brackets ONE pass that does no matrix work — an input transform, a difference
// tensor, a finish kernel — with its algorithmic bytes, so that a profile shows the HBM-bound passes of a Winograd layer beside
// its class GEMM (record family "aux", the tag of the layer being run).  Costs nothing unless s3r_profile_enable is on.
struct AuxScope {
    void* impl;
    AuxScope(hipStream_t s, double bytes);
    ~AuxScope();
    AuxScope(const AuxScope&) = delete;
    AuxScope& operator=(const AuxScope&) = delete;
};

// Division of a non-negative int (< 2^31) by a launch-invariant divisor, as mulhi + add + shift (Granlund &
// Montgomery): the position decodes of the conv kernels did ~10 integer divisions per thread, ~40 instructions each.
struct FastDiv {
    unsigned mul, shr;
    FastDiv() : mul(1), shr(0) {}
    explicit FastDiv(unsigned d) {
        shr = 0;
        while ((1ull << shr) < d) ++shr;
        mul = (unsigned)((((1ull << shr) - d) << 32) / d + 1);
    }
#if defined(__HIPCC__)
    __device__ __forceinline__ int div(int n) const { return (int)((__umulhi((unsigned)n, mul) + (unsigned)n) >> shr); }
#endif
};

#if defined(__HIPCC__)
// A render sample as the fp32 value the network computes on.  8-bit renders (what the reference's PNG decode yields,
// /root/reference/requirements.txt:5, README.md:73-74) are scaled by 1/255 HERE, with the one rounding of the host
// conversion `float32(u) / float32(255)` it replaces: q = u * rcp, then one Newton correction through the exact
// residual u - 255 q, is the correctly rounded quotient for every u in 0..255 (checked exhaustively against exact
// rational arithmetic in tests/test_data_cpu.py::test_u8_scale_formula_is_the_correctly_rounded_quotient).
__device__ __forceinline__ float render_f32(float v) { return v; }
__device__ __forceinline__ float render_f32(unsigned char u) {
    const float f = (float)u;
    const float rcp = 1.f / 255.f;                 // (compile-time constant: the nearest fp32 to 1/255)
    const float q = __fmul_rn(f, rcp);
    return fmaf(fmaf(-q, 255.f, f), rcp, q);
}
#endif

// Kernel-side view of one convolution launch.  All tensors are fp32 NC(D)HW; activations may carry a
// zero halo of `halo` elements on every spatial axis (padded edge = edge + 2*halo), described here
// purely by element strides and origin offsets.
struct ConvParams {
    const float* x;       // input  (B, Cin,  Dp, Hp, Wp)   (padded)
    const float* w;       // packed weights [cls][(chunk*T + tap)*16 + c][CoutPad]
    const float* scale;   // per-cout epilogue scale  (folded BN gamma/sqrt(var+eps)), may be null -> 1
    const float* shift;   // per-cout epilogue shift  (folded bias/BN beta/mean),      may be null -> 0
    float* y;             // output (B, Cout, Dp', Hp', Wp') (padded)
    int B, Cin, Cout, CoutPad;
    int Nd, Nh, Nw;       // per-sample position grid walked by the GEMM N index
                          //   conv: output grid; transposed conv: input grid (one parity class per blockIdx.y)
    FastDiv dS, dHW, dW;  // fast division by Nd*Nh*Nw, Nh*Nw, Nw
    int kd, kh, kw, T;    // taps per axis and in total (transposed k4s2p1: 2,2,2 per parity class)
    int stride;           // input step per position (transposed: 1)
    int dil;              // dilation: taps are dil input elements apart (the direct kernel only; >= 1)
    int x_cs, x_ds, x_hs; // input element strides: channel, depth, row   (batch stride = Cin * x_cs)
    int x_org;            // element offset of tap (0,0,0) of position (0,0,0): (halo_in - pad) per axis
    int y_cs, y_ds, y_hs; // output element strides
    int y_bs;             // output batch stride (Cout * y_cs; the fused head's single-channel output: y_cs)
    int y_org;            // element offset of output (0,0,0): halo_out per axis
    int y_step;           // output elements between consecutive positions along every axis (0 / 1: dense; s: one residue class of a
                          // general ConvTranspose of stride s, run as a stride-1 convolution — s3r_general.hip; transposed = 1 implies 2)
    int shuf_s, shuf_nd;  // > 0: the GEMM's M rows are (cout, tap) of a ConvTranspose with k == stride (row m = cout * s^nd + tap): row m of
    FastDiv dSC, dS1;     // position q goes to output s q + tap of channel cout — a depth-to-space store (dSC: / s^nd, dS1: / s); y_step = s
    unsigned x_bytes;     // size of the input buffer (buffer descriptor range)
    unsigned y_bytes;     // size of the output buffer
    int transposed;       // 0: convolution, 1: ConvTranspose3d(k=4,s=2,p=1) split into 8 parity classes
    int act;              // 0 none, 1 relu, 2 sigmoid
    float slope;          // direct kernel + split-K finish: act = relu with slope a in (0, 1] is LeakyReLU(a) = max(t, a t), fused
                          // (0: plain ReLU, bit for bit the r05 epilogue: the floor is fma(t, slope, lo) = +0)
    int Ntotal;           // B*Nd*Nh*Nw
    int n_begin, n_end;   // position range [n_begin, n_end) this launch covers (a layer may be cut in two launches)
    int n_tiles, m_tiles;
    int n_cut, big_wgs;   // dual launch (bulk + re-tiled remainder in one grid): positions [n_begin, n_cut) by workgroups
                          // [0, big_wgs) with the layer's tile, [n_cut, n_end) by the rest with 64 x 64 tiles
    int ksplit;           // split-K factor (divides Cin/16); > 1: partial slabs to `part`, then conv_finish
    float* part;          // split-K scratch: [cls][ksplit][Cout][n_tiles*BN]
    int debug;            // diagnostic builds (-DS3R_ABLATE) only
    // fused pointwise head (conv -> 1x1x1 conv to one channel + activation): when head_w != null, y / y_* describe
    // the HEAD's output and the conv's own output is never materialised
    const float* head_w;       // [Cout]
    const float* head_scale;   // [1] or null
    const float* head_shift;   // [1] or null
    int head_act;
    // Winograd along H (s3r_conv_wino.hip): x = the transformed input plane sets, x_cls elements apart; Nh = row groups per
    // plane; Hout = the output's true height.  Two-axis form (D and H): Nd = depth groups, Dout = the true depth, ncls = the
    // number of (depth class, row class) plane sets / weight slabs (0: the one-axis form's own count)
    int x_cls, Hout, Dout, ncls;
    // transposed Winograd form: the row differences Dh of the padded input (x's shape and strides); xd_mode = 1: followed by the
    // depth differences Dd and the mixed differences Ddh (materialised), 0: those are formed inside the class kernel
    const float* xd;
    int xd_mode;
};

// The residue classes of a general ConvTranspose (s3r_general.hip) as ONE launch of the direct kernel: workgroup ranges of a class
// table, each entry overriding what differs between the classes — position grid, taps, input / output origin, weight slab
struct TClsEntry {
    int Nd, Nh, Nw, kd, kh, kw, T, x_org, y_org, Ntotal, wg_begin, w_off;
    FastDiv dS, dHW, dW;
};
constexpr int kTClsMax = 27;                 // stride 3 in 3D; more classes (stride 4 in 3D: 64) run one launch per class
struct TClsTable { int n; TClsEntry e[kTClsMax]; };
hipError_t launch_conv_tcls(const ConvParams& base, TClsTable tab, hipStream_t stream);     // fills wg_begin; base.w = the first slab

// Launch form of a Winograd layer (s3r_conv_wino.hip): serial (one workgroup walks all classes of its tile), class-parallel (one
// workgroup per (tile, class), class sums to slabs, a finish kernel transforms them — bit-identical to the serial form), or dual
// (positions [0, n_cut) serial and [n_cut, N) class-parallel in one launch).
enum { WINO_SERIAL = 0, WINO_CP = 1, WINO_DUAL = 2 };
struct WinoLaunch { int mode, n_cut; };

// bf16 channels-last path: activations (B, [D,] H, W, C) bf16 with a zero halo, fp32 accumulate
struct ConvParamsH {
    const void* x;        // bf16 (B, Dp, Hp, Wp, Cin)
    const void* w;        // bf16 packed weights, see pack_bf16_kernel
    const float* scale;
    const float* shift;
    void* y;              // bf16 (B, Dp', Hp', Wp', Cout)
    float* part;          // split-K scratch [cls][ksplit][m_tiles*BM][CoutPad] fp32
    int B, Cin, Cout, CoutPad;            // CoutPad % 64 == 0
    int Nd, Nh, Nw;
    FastDiv dS, dHW, dW, dDH, dH;         // fast division by Nd*Nh*Nw, Nh*Nw, Nw, Nd*Nh, Nh
    int kd, kh, kw, T;
    int stride;
    int x_bs, x_ds, x_hs, x_ws;           // input element strides: batch, depth, row, position (= Cin)
    int x_org;
    int y_bs, y_ds, y_hs, y_ws;           // output element strides (y_ws = Cout)
    int y_org;
    unsigned x_bytes;
    int transposed, act, Ntotal;
    int m_tiles, n_tiles;                 // position tiles, 64-cout tiles
    int ksplit;
    int debug;                            // diagnostic builds (-DS3R_ABLATE) only
    // fused pointwise head: when head_w != null, y / y_* describe the head's fp32 single-channel output
    const float* head_w;
    const float* head_scale;
    const float* head_shift;
    int head_act;
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };      // (the epilogues'; LeakyReLU / ELU / Tanh run as launch_act behind ACT_NONE)

// launchers (defined in the .hip files); every launcher enqueues on `stream` and returns hipGetLastError()
hipError_t launch_conv_mfma(const ConvParams& p, int tile_code, hipStream_t stream);   // code = cfg + 16*vec
int conv_pick_tile(const ConvParams& p);
int conv_pick_vec(const ConvParams& p);
int conv_pick_ksplit(const ConvParams& p, int tile_cfg);
int64_t conv_scratch_elems(const ConvParams& p, int tile_cfg);
int conv_num_tiles();
int conv_last_launch_count();     // kernel launches the calling thread's last launch_conv_mfma made
void conv_tile_dims(int tile_cfg, int* bm, int* bn);
hipError_t launch_pack_conv(const float* w, float* wp, int Cin, int Cout, int CoutPad, int T, int transposed,
                            hipStream_t s);
// Winograd F(2,3) along H for 3 x 3 [x 3] stride-1 pad-1 convolutions (s3r_conv_wino.hip)
hipError_t launch_wino_input(const float* x, float* V, long long planes, int Hp, int Wp, int Hq, int R, hipStream_t s);
hipError_t launch_pack_wino(const float* w, float* wp, int Cin, int Cout, int CoutPad, int kd, int kw, int R, hipStream_t s);
// kind: 1 conv F(4,3), 2 transposed F(2,2) x F(2,2); kcls = K per class (Cin x taps per class); ntotal = positions (groups of R output
// rows) of the launch
WinoLaunch wino_plan(int kind, int cout, int kcls, int ntotal, bool head, int forced);
int64_t wino_slab_elems(int kind, int cout, int ntotal, const WinoLaunch& L);     // floats of class-parallel slabs (p.part)
hipError_t launch_conv_wino(ConvParams p, const WinoLaunch& L, hipStream_t stream, int* launches);
// Two-axis (D and H) class-parallel Winograd convolution of 3D layers with a small edge: ax = 0: F(4,3) on both axes (k3 p1, 36
// classes per 4 x 4 outputs), 1: F(2,4) (k4 p0, 25 classes per 2 x 2 outputs).  x: (B, C, Dp, Hp, Wp) as the layer reads it;
// V: [ncls][B][C][SD][SH][Wp]; slabs in p.part
int wino2_classes(int ax);          // 36 / 25 (ax 0: 3D k3 p1 over D, H; 1: 3D k4 p0; 2: 2D k3 p1 over H, W)
int wino2_outputs(int ax);          // outputs per group and axis: 4 / 2
hipError_t launch_wino2_input(const float* x, float* V, int ax, long long planes, int Dp, int Hp, int Wp, int SD, int SH, hipStream_t s);
hipError_t launch_wino2p_input(const float* x, float* V, int B, int Cin, int Hp, int Wp, int SH, int SW, long long npad, hipStream_t s);
hipError_t launch_pack_wino2(const float* w, float* wp, int ax, int Cin, int Cout, int CoutPad, hipStream_t s);
int wino2_form(int ax, int cout, int ntotal, int forced);                      // 0 class-parallel, 1 semi-fused (same bits)
int64_t wino2_slab_elems(int ax, int cout, int ntotal, int form);
int64_t wino2_npad(int64_t ntotal);                                           // positions rounded up to whole GEMM tiles
hipError_t launch_conv_wino2(ConvParams p, int ax, int form, bool to_v, hipStream_t stream, int* launches);   // to_v: write the next layer's plane sets
// the cost volume written as the 36 two-axis plane sets of its halo-1 padded form: V[36][B][2C][(D)/4][H/4][W+2]
hipError_t launch_cost_volume_wino2(const float* fl, const float* fr, float* V, int B, int C, int D, int H, int W, hipStream_t s);
int wino_bk();                      // channels per K tile of the Winograd kernels (Cin must be a multiple)
// Winograd F(2,2) along D and H inside the parity classes of ConvTranspose3d(k4 s2 p1); D = [Dh] or (three) [Dh | Dd | Ddh]
hipError_t launch_wino_diff(const float* x, float* D, long long planes, int Dp, int Hp, int Wp, int three, hipStream_t s);
hipError_t launch_pack_wino_deconv(const float* w, float* wp, int Cin, int Cout, int CoutPad, hipStream_t s);
hipError_t launch_deconv_wino(ConvParams p, const WinoLaunch& L, hipStream_t stream, int* launches);
// Winograd F(2,2) along D, H AND W inside the parity classes (s3r_deconv_wino3.hip: 27 / 64 of the direct multiplications): positions
// are (sample, depth pair, row pair, column pair), p.Nd = p.Nh = p.Nw = n / 2, p.xd = [Dh | Dd | Ddh]; split: the class-parallel form
// (p.part = dwino3_slab_elems floats of slabs), same bits as the serial one
hipError_t launch_pack_dwino3(const float* w, float* wp, int Cin, int Cout, hipStream_t s);
int64_t dwino3_w_elems(int Cin, int Cout);
bool dwino3_edge_ok(int n);
int64_t dwino3_slab_elems(int cout, int ntotal);
bool dwino3_split(int cout, int ntotal, int forced);
hipError_t launch_deconv_wino3(ConvParams p, bool split, hipStream_t stream);
// y (N,32,Ho+2h,Wo+2h) <- stem conv of x (N,3,Hi,Wi); y_hs / y_cs / y_org describe the padded output.
// Images [0, nsplit) are read from x, images [nsplit, N) from x2 (the left / right renders of a stereo batch live in
// two tensors: no concatenation copy); nsplit = N, x2 = null: one tensor.  u8 != 0: x / x2 are 8-bit renders (N,3,Hi,Wi)
// uint8, scaled by 1/255 inside the kernel (render_f32); otherwise fp32.
hipError_t launch_stem_wino(const void* x, const void* x2, int u8, int nsplit, const float* wt, const float* scale, const float* shift,
                            float* V, int N, int Hi, int Wi, int Ho, int Wo, hipStream_t s);      // (the consumer's S3R_LAYOUT_WINO_H planes)
hipError_t launch_stem(const void* x, const void* x2, int u8, int nsplit, const float* w, const float* scale,
                       const float* shift, float* y, int N, int Hi, int Wi, int Ho, int Wo, int y_cs, int y_hs, int y_org,
                       hipStream_t s);
hipError_t launch_cost_volume(const float* fl, const float* fr, float* vol, int B, int C, int D, int H, int W,
                              int halo, hipStream_t s);
// the same volume written as the Winograd F(R,3)-along-H plane sets of its halo-1 padded form: V[R+2][B][2C][D+2][H/R][W+2]
hipError_t launch_cost_volume_wino(const float* fl, const float* fr, float* V, int B, int C, int D, int H, int W, int R, hipStream_t s);
// parameter-general layers (s3r_general.hip): staged input (channel padding, padding halo, zero-stuffing), general weight packing,
// LeakyReLU / ELU / Tanh as a pass of their own
hipError_t launch_stage(const float* x, float* y, int B, int Cin, int CinPad, int nd, int n, int in_halo, int sp, int pe, int step,
                        hipStream_t s);
hipError_t launch_pack_general(const float* w, float* wp, int Cin, int CinPad, int Cout, int CoutPad, int T, int flip, hipStream_t s);
// im2col staging of a convolution with very few input channels (an RGB first layer): y[b][(c, td, th, tw)][od][oh][ow] = x at the tap's
// input position (0 where it is padding, 0 for the rows that pad Cin k^nd up to KPad) — the layer is then a 1 x 1 GEMM over KPad
hipError_t launch_stage_im2col(const float* x, float* y, int B, int Cin, int KPad, int nd, int n, int in_halo, int n_out, int k, int stride,
                               int pad, int dil, hipStream_t s);
// one residue class of a general ConvTranspose (w[Cin][Cout][k^nd]) as a stride-1 correlation kernel of kd x kh x kw taps: tap j of an
// axis with residue r and kr taps reads kernel index r + stride * (kr - 1 - j); an axis with NO tap of that residue (kr = 0: stride > k)
// packs one zero tap (the class's outputs are act(shift))
hipError_t launch_pack_tclass(const float* w, float* wp, int Cin, int CinPad, int Cout, int CoutPad, int nd, int k, int stride,
                              int rd, int rh, int rw, hipStream_t s);
hipError_t launch_act(float* y, long long total, int act, float param, hipStream_t s);
hipError_t launch_pad_copy(const float* x, float* y, int64_t planes, int D, int H, int W, int hd, int hh, int hw,
                           hipStream_t s);
hipError_t launch_pack_stem(const float* w, float* wt, hipStream_t s);
hipError_t launch_head(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int C,
                       int64_t S, int act, hipStream_t s);
hipError_t launch_chamfer(const float* p, const float* q, float* d1, float* d2, int* i1, int* i2,
                          int B, int N, int M, hipStream_t s);
// scratch: linear_scratch_elems floats of split-K slabs
hipError_t launch_linear(const float* x, const float* w, const float* scale, const float* bias, float* y, int B,
                         int Cin, int Cout, int act, float* scratch, hipStream_t s);
int64_t linear_scratch_elems(int B, int Cin, int Cout);
// ---- bf16 channels-last path
hipError_t launch_conv_bf16(const ConvParamsH& p, int tm, hipStream_t stream);
int conv_bf16_pick_tm(const ConvParamsH& p);
int conv_bf16_shape();              // 16 | 32: the bf16 matrix instruction in use (S3R_BF16_MFMA, else the build's default)
int conv_bf16_pick_ksplit(const ConvParamsH& p);
int64_t conv_bf16_scratch_elems(const ConvParamsH& p, int tm);
hipError_t launch_pack_bf16(const float* w, void* wp, int Cin, int Cout, int CoutPad, int T, int transposed, hipStream_t s);
hipError_t launch_stem_bf16(const void* x, const void* x2, int u8, int nsplit, const float* wt, const float* scale,
                            const float* shift, void* y, int N, int Hi, int Wi, int Ho, int Wo, int y_bs, int y_hs, int y_org,
                            hipStream_t s);
hipError_t launch_cost_volume_bf16(const void* fl, const void* fr, void* vol, int B, int C, int D, int H, int W, int halo,
                                   hipStream_t s);
hipError_t launch_head_bf16(const void* x, const float* w, const float* scale, const float* shift, float* y, int C,
                            int64_t voxels, int act, hipStream_t s);
hipError_t launch_cl_bf16_to_f32(const void* x, float* y, int N, int64_t S, int C, hipStream_t s);
hipError_t launch_iou(const float* pred, const float* gt, float th, float* iou, int B, int64_t S, hipStream_t s);
hipError_t launch_disparity_wta(const float* fl, const float* fr, float* dl, float* dr, int B, int C, int D, int H, int W,
                                hipStream_t s);
hipError_t launch_disparity_epe(const float* pred, const float* gt, float* epe, int* count, int B, int64_t S,
                                hipStream_t s);

#if defined(__HIPCC__)
// Winograd F(4, 3) input transform along one axis (s3r_conv_wino.hip): the six class values of six consecutive padded rows.
// Shared by the transform kernel and the cost-volume kernel that writes the transformed planes directly: same code, same bits.
template <int R>
__device__ __forceinline__ void wino_rows_to_classes(const float (&r)[R + 2], float (&v)[R + 2]) {
    static_assert(R == 4, "F(4,3)");
    v[0] = fmaf(4.f, r[0], fmaf(-5.f, r[2], r[4]));
    v[1] = fmaf(-4.f, r[1] + r[2], r[3] + r[4]);
    v[2] = fmaf(4.f, r[1] - r[2], r[4] - r[3]);
    v[3] = fmaf(2.f, r[3] - r[1], r[4] - r[2]);
    v[4] = fmaf(2.f, r[1] - r[3], r[4] - r[2]);
    v[5] = fmaf(4.f, r[1], fmaf(-5.f, r[3], r[5]));
}

#endif

}  // namespace s3r
