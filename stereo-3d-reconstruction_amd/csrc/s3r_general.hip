// Parameter-general layers (ABI 8), correctness first: what lets ANY fp32 Conv / ConvTranspose (2D or 3D; k, stride, pad, dilation,
// output padding; any channel count; LeakyReLU / ELU / Tanh) run on the direct implicit-GEMM kernel of s3r_conv_glds.hip.
//
//   stage_kernel         x (B, Cin, n [+ 2 halo] ...) -> staged (B, CinPad, U + 2 pe, ...), zero everywhere else: channels padded to a
//                        multiple of 16, the zero padding pe materialised as a halo.
//   ConvTranspose(k, s, p, op), dilation 1 (r06): s^ndim RESIDUE CLASSES (ONE launch over a class table), each a stride-1 convolution of the direct kernel over
//                        the halo-padded (NOT stuffed) input: output o = s q + r - p of class r takes the taps t = r, r + s, ... < k from
//                        inputs q, q - 1, ...; the class kernel is that tap subset in correlation order (pack_tclass_kernel), its
//                        positions are the q with 0 <= o < n_out, its outputs go out s elements apart (ConvParams::y_step).  Executes
//                        the algorithmic multiplications (x CinPad / Cin, + the taps that meet the halo) where r05's zero-stuffed form
//                        executed s^ndim times as many.
//   ConvTranspose with dilation > 1: still the zero-stuffed form — the input stuffed at the stride (sample i at pe + i * stride),
//                        a stride-1 convolution with the flipped kernel, dilation d and padding d (k - 1) - p.
//   pack_general_kernel  torch weights -> the direct kernel's packed K order with CinPad rows (zeros beyond Cin), flipped for
//                        transposed layers
//   act_kernel           LeakyReLU / ELU / Tanh in place (the MFMA epilogues know none / ReLU / sigmoid); act(0) = 0 keeps halos zero
#include "s3r_kernels.h"

namespace s3r {

__global__ __launch_bounds__(256) void stage_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int Cin, int CinPad,
                                                    int nd, int n, int x_hs, int x_ds, int x_cs, int x_org, int sp, int pe, int step) {
    // one thread per INPUT element (b, c, d, h, w): scattered to its staged position; the rest of y was zeroed by the caller
    const long long S = nd == 3 ? (long long)n * n * n : (long long)n * n;
    const long long y_cs = nd == 3 ? (long long)sp * sp * sp : (long long)sp * sp;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const long long bc = i / S;
        long long r = i - bc * S;
        const int w = (int)(r % n); r /= n;
        const int hh = (int)(r % n); r /= n;
        const int dd = (int)r;                                   // (0 in 2D)
        const long long b = bc / Cin;
        const int c = (int)(bc - b * Cin);
        const float v = x[bc * x_cs + x_org + (long long)dd * x_ds + (long long)hh * x_hs + w];
        long long o = (b * CinPad + c) * y_cs + (long long)(pe + hh * step) * sp + (pe + w * step);
        if (nd == 3) o += (long long)(pe + dd * step) * sp * sp;
        y[o] = v;
    }
}

hipError_t launch_stage(const float* x, float* y, int B, int Cin, int CinPad, int nd, int n, int in_halo, int sp, int pe, int step,
                        hipStream_t s) {
    const long long S = nd == 3 ? (long long)n * n * n : (long long)n * n;
    const long long total = (long long)B * Cin * S;
    const long long y_elems = (long long)B * CinPad * (nd == 3 ? (long long)sp * sp * sp : (long long)sp * sp);
    hipError_t e = hipMemsetAsync(y, 0, (size_t)y_elems * sizeof(float), s);
    if (e != hipSuccess) return e;
    const int np = n + 2 * in_halo;
    const int x_hs = np, x_ds = nd == 3 ? np * np : 0, x_cs = nd == 3 ? np * np * np : np * np;
    const int x_org = in_halo * (x_ds + x_hs + 1);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(stage_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, x, y, total, Cin, CinPad, nd, n,
                       x_hs, x_ds, x_cs, x_org, sp, pe, step);
    return hipGetLastError();
}

// im2col staging (r06): a convolution with cin <= 8 (an RGB stem that is not this network's: Conv2d 3 -> 64, k7 s2 p3) wastes 13 of
// every 16 channel rows of the direct kernel's K tiles when its channels are padded to 16 (5.3 x the multiplications).  Unfolded, the
// layer is a 1 x 1 GEMM over K = cin k^nd rows (147 -> 160: 1.09 x) at the price of a staged tensor k^nd / stride^nd times the input.
__global__ __launch_bounds__(256) void stage_im2col_kernel(const float* __restrict__ x, float* __restrict__ y, long long total, int Cin, int KPad,
                                                           int nd, int n, int x_hs, int x_ds, int x_cs, int x_org, int no, int k, int stride,
                                                           int pad, int dil) {
    const long long OS = nd == 3 ? (long long)no * no * no : (long long)no * no;
    const int T = nd == 3 ? k * k * k : k * k;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        long long r = i;
        const int ow = (int)(r % no); r /= no;
        const int oh = (int)(r % no); r /= no;
        const int od = nd == 3 ? (int)(r % no) : 0;
        if (nd == 3) r /= no;
        const int kidx = (int)(r % KPad);
        const long long b = r / KPad;
        float v = 0.f;
        if (kidx < Cin * T) {
            const int c = kidx / T, tap = kidx - c * T;
            const int tw = tap % k, th = (tap / k) % k, td = nd == 3 ? tap / (k * k) : 0;
            const int iw = ow * stride - pad + tw * dil, ih = oh * stride - pad + th * dil, id = nd == 3 ? od * stride - pad + td * dil : 0;
            if (iw >= 0 && iw < n && ih >= 0 && ih < n && id >= 0 && id < n)
                v = x[(b * Cin + c) * x_cs + x_org + (long long)id * x_ds + (long long)ih * x_hs + iw];
        }
        y[i] = v;
    }
    (void)OS;
}

hipError_t launch_stage_im2col(const float* x, float* y, int B, int Cin, int KPad, int nd, int n, int in_halo, int n_out, int k, int stride,
                               int pad, int dil, hipStream_t s) {
    const long long total = (long long)B * KPad * (nd == 3 ? (long long)n_out * n_out * n_out : (long long)n_out * n_out);
    const int np = n + 2 * in_halo;
    const int x_hs = np, x_ds = nd == 3 ? np * np : 0, x_cs = nd == 3 ? np * np * np : np * np;
    const int x_org = in_halo * (x_ds + x_hs + 1);
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(stage_im2col_kernel, dim3((unsigned)(blocks < 65536 ? blocks : 65536)), dim3(256), 0, s, x, y, total, Cin, KPad, nd, n,
                       x_hs, x_ds, x_cs, x_org, n_out, k, stride, pad, dil);
    return hipGetLastError();
}

// wp[(chunk * T + tap) * 16 + c][CoutPad]; conv: w[Cout][Cin][taps]; transposed: w[Cin][Cout][taps] read at the FLIPPED tap
__global__ void pack_general_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int CinPad, int Cout, int CoutPad, int T,
                                    int flip) {
    const size_t total = (size_t)T * CinPad * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int co = (int)(r % CoutPad); r /= CoutPad;
        const int c = (int)(r & 15); r >>= 4;
        const int tap = (int)(r % T);
        const int cin = (int)(r / T) * 16 + c;
        float v = 0.f;
        if (co < Cout && cin < Cin)
            v = flip ? w[((size_t)cin * Cout + co) * T + (T - 1 - tap)] : w[((size_t)co * Cin + cin) * T + tap];
        wp[i] = v;
    }
}

hipError_t launch_pack_general(const float* w, float* wp, int Cin, int CinPad, int Cout, int CoutPad, int T, int flip, hipStream_t s) {
    hipLaunchKernelGGL(pack_general_kernel, dim3(1024), dim3(256), 0, s, w, wp, Cin, CinPad, Cout, CoutPad, T, flip);
    return hipGetLastError();
}

// taps of residue r along one axis: kernel indices r, r + s, ... < k
__host__ __device__ inline int tclass_taps(int k, int s, int r) { return r < k ? (k - r + s - 1) / s : 0; }

__global__ void pack_tclass_kernel(const float* __restrict__ w, float* __restrict__ wp, int Cin, int CinPad, int Cout, int CoutPad, int nd,
                                   int k, int stride, int rd, int rh, int rw) {
    const int krd = nd == 3 ? tclass_taps(k, stride, rd) : 1, krh = tclass_taps(k, stride, rh), krw = tclass_taps(k, stride, rw);
    const int ed = krd > 0 ? krd : 1, eh = krh > 0 ? krh : 1, ew = krw > 0 ? krw : 1;      // (an empty axis: one zero tap)
    const int T = ed * eh * ew, KT = nd == 3 ? k * k * k : k * k;
    const size_t total = (size_t)T * CinPad * CoutPad;
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < total; i += (size_t)gridDim.x * blockDim.x) {
        size_t r = i;
        const int co = (int)(r % CoutPad); r /= CoutPad;
        const int c = (int)(r & 15); r >>= 4;
        const int tap = (int)(r % T);
        const int cin = (int)(r / T) * 16 + c;
        const int jw = tap % ew, jh = (tap / ew) % eh, jd = tap / (ew * eh);
        float v = 0.f;
        if (co < Cout && cin < Cin && krh > 0 && krw > 0 && (nd != 3 || krd > 0)) {
            const int tw = rw + stride * (krw - 1 - jw), th = rh + stride * (krh - 1 - jh);
            const int td = nd == 3 ? rd + stride * (krd - 1 - jd) : 0;
            v = w[((size_t)cin * Cout + co) * KT + ((size_t)td * k + th) * k + tw];
        }
        wp[i] = v;
    }
}

hipError_t launch_pack_tclass(const float* w, float* wp, int Cin, int CinPad, int Cout, int CoutPad, int nd, int k, int stride,
                              int rd, int rh, int rw, hipStream_t s) {
    hipLaunchKernelGGL(pack_tclass_kernel, dim3(256), dim3(256), 0, s, w, wp, Cin, CinPad, Cout, CoutPad, nd, k, stride, rd, rh, rw);
    return hipGetLastError();
}

__global__ __launch_bounds__(256) void act_kernel(float* __restrict__ y, long long total, int act, float param) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < total; i += (long long)gridDim.x * 256) {
        const float v = y[i];
        float o;
        if (act == 3) o = v > 0.f ? v : v * param;                       // LeakyReLU
        else if (act == 4) o = v > 0.f ? v : param * expm1f(v);          // ELU
        else o = tanhf(v);                                               // Tanh
        y[i] = o;
    }
}

hipError_t launch_act(float* y, long long total, int act, float param, hipStream_t s) {
    if (act < 3 || act > 5) return hipErrorInvalidValue;
    const long long blocks = (total + 255) / 256;
    hipLaunchKernelGGL(act_kernel, dim3((unsigned)(blocks < 16384 ? blocks : 16384)), dim3(256), 0, s, y, total, act, param);
    return hipGetLastError();
}

}  // namespace s3r
