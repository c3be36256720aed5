/* ORACLE — test infrastructure, NOT product code.  Parity unpinned (see s2v_oracle.py header).
 *
 * Plain-C, loop-nest restatements of the four primitive ops of the path, written from the
 * mathematical definitions (independent of both MKL-DNN and the HIP kernels' formulations).  They pin
 * the PyTorch oracle's operator semantics at small sizes (tests/test_oracle_cpu.py):
 *   s3r_ref_conv       Conv2d/Conv3d, zero padding, cubic kernels      (SURVEY.md §8a rows 1,3)
 *   s3r_ref_deconv     ConvTranspose3d in the SCATTER form out[i*s-p+k] += x[i]*w[k]  (row 3)
 *   s3r_ref_cost_volume bidirectional shift-and-diff                    (row 2; README.md:75-76)
 *   s3r_ref_chamfer    squared-L2 nearest neighbours, first minimum     (row 5; README.md:64-65)
 * Accumulation is in double so the result is a tight reference for fp32 kernels.
 */
#include <math.h>
#include <stdint.h>
#include <string.h>

/* x (B,Cin,Di,Hi,Wi) w (Cout,Cin,kd,k,k) -> y (B,Cout,Do,Ho,Wo); 2D: Di=Do=kd=1, pd=0 */
void s3r_ref_conv(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int Di,
                  int Hi, int Wi, int kd, int k, int s, int pd, int p) {
    const int Do = (Di + 2 * pd - kd) / s + 1, Ho = (Hi + 2 * p - k) / s + 1, Wo = (Wi + 2 * p - k) / s + 1;
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co)
            for (int od = 0; od < Do; ++od)
                for (int oh = 0; oh < Ho; ++oh)
                    for (int ow = 0; ow < Wo; ++ow) {
                        double acc = bias ? bias[co] : 0.0;
                        for (int ci = 0; ci < Cin; ++ci)
                            for (int a = 0; a < kd; ++a)
                                for (int c = 0; c < k; ++c)
                                    for (int e = 0; e < k; ++e) {
                                        const int id = od * s - pd + a, ih = oh * s - p + c, iw = ow * s - p + e;
                                        if (id < 0 || id >= Di || ih < 0 || ih >= Hi || iw < 0 || iw >= Wi) continue;
                                        acc += (double)x[(((size_t)b * Cin + ci) * Di + id) * Hi * Wi + ih * Wi + iw] *
                                               w[((((size_t)co * Cin + ci) * kd + a) * k + c) * k + e];
                                    }
                        y[(((size_t)b * Cout + co) * Do + od) * Ho * Wo + oh * Wo + ow] = (float)acc;
                    }
}

/* x (B,Cin,n,n,n) w (Cin,Cout,k,k,k) -> y (B,Cout,m,m,m), m=(n-1)s-2p+k; scatter form */
void s3r_ref_deconv(const float* x, const float* w, const float* bias, float* y, int B, int Cin, int Cout, int n,
                    int k, int s, int p) {
    const int m = (n - 1) * s - 2 * p + k;
    const size_t plane = (size_t)m * m * m;
    double* acc = (double*)__builtin_malloc(sizeof(double) * plane);
    for (int b = 0; b < B; ++b)
        for (int co = 0; co < Cout; ++co) {
            for (size_t i = 0; i < plane; ++i) acc[i] = bias ? bias[co] : 0.0;
            for (int ci = 0; ci < Cin; ++ci)
                for (int id = 0; id < n; ++id)
                    for (int ih = 0; ih < n; ++ih)
                        for (int iw = 0; iw < n; ++iw) {
                            const double xv = x[(((size_t)b * Cin + ci) * n + id) * n * n + ih * n + iw];
                            for (int a = 0; a < k; ++a)
                                for (int c = 0; c < k; ++c)
                                    for (int e = 0; e < k; ++e) {
                                        const int od = id * s - p + a, oh = ih * s - p + c, ow = iw * s - p + e;
                                        if (od < 0 || od >= m || oh < 0 || oh >= m || ow < 0 || ow >= m) continue;
                                        acc[((size_t)od * m + oh) * m + ow] +=
                                            xv * w[((((size_t)ci * Cout + co) * k + a) * k + c) * k + e];
                                    }
                        }
            for (size_t i = 0; i < plane; ++i) y[((size_t)b * Cout + co) * plane + i] = (float)acc[i];
        }
    __builtin_free(acc);
}

void s3r_ref_cost_volume(const float* fl, const float* fr, float* vol, int B, int C, int D, int H, int W) {
    for (int b = 0; b < B; ++b)
        for (int c = 0; c < C; ++c)
            for (int d = 0; d < D; ++d)
                for (int h = 0; h < H; ++h)
                    for (int w = 0; w < W; ++w) {
                        const size_t f = (((size_t)b * C + c) * H + h) * W;
                        const size_t ol = (((((size_t)b * 2 * C + c) * D + d) * H + h) * W) + w;
                        const size_t orr = (((((size_t)b * 2 * C + C + c) * D + d) * H + h) * W) + w;
                        vol[ol] = (w - d >= 0) ? fl[f + w] - fr[f + w - d] : 0.f;
                        vol[orr] = (w + d < W) ? fr[f + w] - fl[f + w + d] : 0.f;
                    }
}

/* p (B,N,3) q (B,M,3): d1/i1 over N, d2/i2 over M; float arithmetic ((dx*dx+dy*dy)+dz*dz), first minimum */
void s3r_ref_chamfer(const float* p, const float* q, float* d1, float* d2, int32_t* i1, int32_t* i2, int B, int N,
                     int M) {
    for (int b = 0; b < B; ++b) {
        const float* pb = p + (size_t)b * N * 3;
        const float* qb = q + (size_t)b * M * 3;
        for (int j = 0; j < M; ++j) { d2[(size_t)b * M + j] = INFINITY; i2[(size_t)b * M + j] = 0; }
        for (int i = 0; i < N; ++i) {
            float best = INFINITY;
            int bi = 0;
            for (int j = 0; j < M; ++j) {
                const float dx = pb[i * 3] - qb[j * 3], dy = pb[i * 3 + 1] - qb[j * 3 + 1], dz = pb[i * 3 + 2] - qb[j * 3 + 2];
                volatile float xx = dx * dx, yy = dy * dy, zz = dz * dz;   /* no FMA contraction */
                volatile float s1 = xx + yy;
                const float d = s1 + zz;
                if (d < best) { best = d; bi = j; }
                if (d < d2[(size_t)b * M + j]) { d2[(size_t)b * M + j] = d; i2[(size_t)b * M + j] = i; }
            }
            d1[(size_t)b * N + i] = best;
            i1[(size_t)b * N + i] = bi;
        }
    }
}
