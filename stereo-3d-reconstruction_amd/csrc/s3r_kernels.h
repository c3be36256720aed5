// Internal (non-ABI) declarations shared by the HIP translation units of libs3r_hip.so.
// The public C-ABI is include/s3r.h.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace s3r {

// Kernel-side view of one convolution launch.  All tensors are fp32, contiguous NC(D)HW.
struct ConvParams {
    const float* x;       // (B, Cin, Di, Hi, Wi)
    const float* w;       // packed weights, see pack kernels: [cls][tap][cin][CoutPad]
    const float* scale;   // per-cout epilogue scale  (folded BN gamma/sqrt(var+eps)), may be null -> 1
    const float* shift;   // per-cout epilogue shift  (folded bias/BN beta/mean),      may be null -> 0
    float* y;             // (B, Cout, Do, Ho, Wo)
    int B, Cin, Cout, CoutPad;
    int Di, Hi, Wi;
    int Do, Ho, Wo;
    int Nd, Nh, Nw;       // per-sample position grid walked by the GEMM N index
                          //   conv: output grid; transposed conv: input grid (one parity class per blockIdx.y)
    int kd, kh, kw;       // taps per axis (transposed k4s2p1: 2,2,2 per parity class)
    int stride;
    int pad_d, pad_h, pad_w;
    int transposed;       // 0: convolution, 1: ConvTranspose3d(k=4,s=2,p=1) split into 8 parity classes
    int act;              // 0 none, 1 relu, 2 sigmoid
    int Ntotal;           // B*Nd*Nh*Nw
    int n_tiles, m_tiles;
};

enum { ACT_NONE = 0, ACT_RELU = 1, ACT_SIGMOID = 2 };

// launchers (defined in the .hip files); every launcher enqueues on `stream` and returns hipGetLastError()
hipError_t launch_conv_mfma(const ConvParams& p, int tile_cfg, hipStream_t stream);
int conv_pick_tile(const ConvParams& p);                     // heuristic tile choice
void conv_tile_dims(int tile_cfg, int* bm, int* bn);
hipError_t launch_conv_direct(const ConvParams& p, int tile_cfg, hipStream_t stream);
hipError_t launch_pack_direct(const float* w, float* wd, int Cin, int Cout, int CoutPad, int T, int transposed,
                              hipStream_t s);
hipError_t launch_pack_conv(const float* w, float* wp, int Cout, int Cin, int T, int CoutPad, hipStream_t s);
hipError_t launch_pack_deconv_k4s2(const float* w, float* wp, int Cin, int Cout, int CoutPad, hipStream_t s);
hipError_t launch_stem(const float* x, const float* w, const float* scale, const float* shift, float* y,
                       int N, int Hi, int Wi, int Ho, int Wo, hipStream_t s);
hipError_t launch_cost_volume(const float* fl, const float* fr, float* vol, int B, int C, int D, int H, int W,
                              hipStream_t s);
hipError_t launch_pack_stem(const float* w, float* wt, hipStream_t s);
hipError_t launch_head(const float* x, const float* w, const float* scale, const float* shift, float* y, int B, int C,
                       int64_t S, int act, hipStream_t s);
hipError_t launch_chamfer(const float* p, const float* q, float* d1, float* d2, int* i1, int* i2,
                          int B, int N, int M, hipStream_t s);
hipError_t launch_linear(const float* x, const float* w, const float* scale, const float* bias, float* y, int B,
                         int Cin, int Cout, int act, hipStream_t s);
hipError_t launch_iou(const float* pred, const float* gt, float th, float* iou, int B, int64_t S, hipStream_t s);

}  // namespace s3r
