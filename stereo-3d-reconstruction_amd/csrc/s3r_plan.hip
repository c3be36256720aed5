// Host-side planning of libs3r_hip.so: argument validation and layer geometry, which kernel / algorithm / launch form serves a
// descriptor (the library's policy), the sizes that follow (packed weights, scratch, transformed inputs), kernel-side parameter
// blocks, and the chain planner that lays the activation arena out.  Nothing here enqueues anything; the size queries of the
// C-ABI (include/s3r.h) live here because they ARE the plan.
#include "s3r_host.h"

#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace s3rh {

thread_local char g_err[512] = "";

int fail(int code, const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
    return code;
}

int hip_fail(hipError_t e, const char* what) {
    return fail(S3R_ERR_HIP, "%s: %s", what, hipGetErrorString(e));
}

int64_t ipow(int64_t b, int e) {
    int64_t r = 1;
    while (e-- > 0) r *= b;
    return r;
}

// outputs per Winograd group along H of an fp32 3 x 3 [x 3] stride-1 convolution: F(4,3), half the direct form's multiplications.
// Edges that are not a multiple of 4 compute a partial last group (v3: 16 rows for 14, v5: 8 for 7) and still beat F(2,3), which
// r03 used for them (v3 0.505 -> 0.431 ms, v5 0.304 -> 0.251 alone at B = 32): one group size, one class kernel.
int wino_r(const s3r_conv_desc*) { return 4; }


const char* last_error() { return g_err; }

// ---------------------------------------------------------------- layer geometry

int dil_of(const s3r_conv_desc* d) { return d->dilation > 0 ? d->dilation : 1; }      // (a zero-initialised ABI-7 descriptor: 1)

int out_size(const s3r_conv_desc* d) {
    if (d->op == S3R_OP_LINEAR) return 1;
    if (d->op == S3R_OP_DECONV) return (d->in_size - 1) * d->stride - 2 * d->pad + dil_of(d) * (d->k - 1) + d->out_pad + 1;
    const int span = d->in_size + 2 * d->pad - dil_of(d) * (d->k - 1) - 1;
    return span < 0 ? 0 : span / d->stride + 1;
}

// the transposed layer the tuned kernels serve (8 output-parity classes of 2 x 2 x 2 taps; Winograd forms)
static bool deconv_fast(const s3r_conv_desc* d) {
    return d->op == S3R_OP_DECONV && d->ndim == 3 && d->k == 4 && d->stride == 2 && d->pad == 1 && dil_of(d) == 1 && d->out_pad == 0;
}
static bool stem_shape(const s3r_conv_desc* d) {
    return d->op == S3R_OP_CONV && d->ndim == 2 && d->cin == 3 && d->cout == 32 && d->k == 3 && d->stride == 2 && d->pad == 1 &&
           d->act == S3R_ACT_RELU && dil_of(d) == 1;
}
static bool head_shape(const s3r_conv_desc* d) {
    return d->op == S3R_OP_CONV && d->cout == 1 && d->k == 1 && d->stride == 1 && d->pad == 0 && d->act <= S3R_ACT_SIGMOID &&
           (ipow(d->in_size, d->ndim) % 4) == 0;
}
// Parameter-general layers (ABI 8, s3r_general.hip): an fp32 convolution whose channel count is not a multiple of 16, or any
// transposed convolution but the tuned one, runs the direct kernel over a STAGED copy of its input (channels zero-padded, the
// padding as a halo, transposed: zero-stuffed at the stride).  Correct first; the network's own shapes never take this path.
bool staged_layer(const s3r_conv_desc* d) {
    if (d->dtype != S3R_F32 || (d->op != S3R_OP_CONV && d->op != S3R_OP_DECONV) || (d->ndim != 2 && d->ndim != 3)) return false;
    if (stem_shape(d) || head_shape(d)) return false;
    if (d->op == S3R_OP_DECONV) return !(deconv_fast(d) && d->cin % 16 == 0);
    return d->cin % 16 != 0;
}
// A staged ConvTranspose with dilation 1 runs as stride^ndim residue classes over the halo-padded (not stuffed) input
bool tclass_layer(const s3r_conv_desc* d) { return d->op == S3R_OP_DECONV && staged_layer(d) && dil_of(d) == 1; }
// ... and WITHOUT the staged copy when its input already carries the halo the classes read and whole 16-channel K tiles: what
// s3r_chain_forward plans for such a layer behind another one (want_halo), or a caller states with in_halo
bool tclass_direct(const s3r_conv_desc* d) {
    return tclass_layer(d) && d->cin % 16 == 0 && d->in_halo >= tclass_halo(d) && d->in_layout == S3R_LAYOUT_PLAIN;
}
TClassAxis tclass_axis(const s3r_conv_desc* d, int r) {
    TClassAxis a;
    const int k = d->k, s = d->stride, p = d->pad, n_out = out_size(d);
    a.kr = r < k ? (k - r + s - 1) / s : 0;
    a.ke = a.kr > 0 ? a.kr : 1;
    a.qmin = p > r ? (p - r + s - 1) / s : 0;                       // first q with s q + r - p >= 0
    const int top = n_out - 1 + p - r;                               // last q: s q + r - p <= n_out - 1
    a.nq = top < 0 ? 0 : top / s - a.qmin + 1;
    return a;
}
int tclass_halo(const s3r_conv_desc* d) {
    int h = 0;
    for (int r = 0; r < d->stride; ++r) {
        const TClassAxis a = tclass_axis(d, r);
        if (a.nq <= 0) continue;
        const int lo = (a.ke - 1) - a.qmin, hi = a.qmin + a.nq - 1 - (d->in_size - 1);      // inputs q - (ke - 1) .. q
        if (lo > h) h = lo;
        if (hi > h) h = hi;
    }
    return h;
}
bool tshuf_layer(const s3r_conv_desc* d) {
    return tclass_layer(d) && d->k == d->stride && d->stride >= 2 && d->pad == 0 && d->out_pad == 0;
}
int64_t tclass_w_elems(const s3r_conv_desc* d) {
    if (tshuf_layer(d)) return (int64_t)((d->cin + 15) / 16 * 16) * cout_pad(d->cout * (int)ipow(d->stride, d->ndim));      // [CinPad][cout x taps]
    int64_t per_axis = 0;
    for (int r = 0; r < d->stride; ++r) per_axis += tclass_axis(d, r).ke;
    return ipow(per_axis, d->ndim) * ((d->cin + 15) / 16 * 16) * cout_pad(d->cout);      // sum over classes of prod ke = (sum ke)^nd
}
// (decided from the layer's PER-SAMPLE geometry only: the packed weights have the unfolded layout, and they are packed once for every
// batch; a batch whose unfolded copy would pass 1 GiB goes through in sub-batches)
bool im2col_layer(const s3r_conv_desc* d) {
    if (d->op != S3R_OP_CONV || !staged_layer(d) || d->cin > 8 || d->k > 16 || d->k < 2 || d->in_size > 4096) return false;
    const int no = out_size(d);
    if (no <= 0) return false;
    const double per_sample = ((double)d->cin * (double)ipow(d->k, d->ndim) + 15.0) * (d->ndim == 3 ? (double)no * no * no : (double)no * no);
    return per_sample * 4 <= 64.0 * (1 << 20);      // (else: channels padded to 16, the halo-padded copy)
}
StagedGeo staged_geo(const s3r_conv_desc* d) {
    StagedGeo s;
    s.bmax = 0;
    s.cin_pad = (d->cin + 15) / 16 * 16;
    if (im2col_layer(d)) {                        // (B, KPad, n_out, ...): every tap of every channel is a K row
        s.cin_pad = (d->cin * (int)ipow(d->k, d->ndim) + 15) / 16 * 16;
        s.step = 1; s.pe = 0;
        s.sp = out_size(d);
        const int64_t per_sample = (int64_t)s.cin_pad * ipow(s.sp, d->ndim);
        s.bmax = (int)(((int64_t)1 << 28) / per_sample);          // 1 GiB of floats per pass (>= 16 samples by im2col_layer's bound)
        s.elems = (int64_t)(d->batch < s.bmax ? d->batch : s.bmax) * per_sample;
        return s;
    }
    if (tclass_layer(d)) {
        s.step = 1;
        s.pe = tclass_halo(d);
        s.sp = d->in_size + 2 * s.pe;
        s.elems = (int64_t)d->batch * s.cin_pad * ipow(s.sp, d->ndim);
        return s;
    }
    const bool tr = d->op == S3R_OP_DECONV;
    s.step = tr ? d->stride : 1;
    s.pe = tr ? dil_of(d) * (d->k - 1) - d->pad : d->pad;
    const int u = tr ? (d->in_size - 1) * d->stride + 1 + d->out_pad : d->in_size;
    s.sp = u + 2 * s.pe;
    s.elems = (int64_t)d->batch * s.cin_pad * ipow(s.sp, d->ndim);
    return s;
}

int geometry(const s3r_conv_desc* d, Geo* g) {
    if (!d) return fail(S3R_ERR_INVALID, "null descriptor");
    if (d->batch <= 0 || d->cin <= 0 || d->cout <= 0) return fail(S3R_ERR_INVALID, "batch/cin/cout must be positive");
    if (d->in_halo < 0 || d->out_halo < 0 || d->in_halo > 8 || d->out_halo > 8)
        return fail(S3R_ERR_INVALID, "halo must be in [0, 8]");
    if (d->dtype != S3R_F32 && d->dtype != S3R_BF16) return fail(S3R_ERR_INVALID, "unknown dtype %d", d->dtype);
    if ((d->in_layout != S3R_LAYOUT_PLAIN && d->in_layout != S3R_LAYOUT_WINO_H && d->in_layout != S3R_LAYOUT_WINO_DH &&
         d->in_layout != S3R_LAYOUT_WINO_HW) || (d->out_layout != S3R_LAYOUT_PLAIN && d->out_layout != S3R_LAYOUT_WINO_HW))
        return fail(S3R_ERR_INVALID, "unknown layout");
    if ((d->in_layout || d->out_layout) && d->op == S3R_OP_LINEAR)
        return fail(S3R_ERR_INVALID, "the transformed input layout exists on the convolution paths only");
    if (d->op == S3R_OP_LINEAR) {
        if (d->in_halo || d->out_halo) return fail(S3R_ERR_INVALID, "linear layers take no halo");
        if (d->act < S3R_ACT_NONE || d->act > S3R_ACT_TANH) return fail(S3R_ERR_INVALID, "unknown activation %d", d->act);
        g->nd = 0; g->in = g->out = g->in_p = g->out_p = 1; g->in_sp = 1; g->out_sp = 1;
        g->x_elems = (int64_t)d->batch * d->cin;
        g->y_elems = (int64_t)d->batch * d->cout;
        g->w_elems = (int64_t)d->cin * d->cout;
        g->x_store = g->x_elems; g->y_store = g->y_elems;
        g->flops = 2.0 * d->batch * (double)d->cin * d->cout;
        g->bytes = 4.0 * (g->x_elems + g->y_elems + g->w_elems);
        if (d->dtype != S3R_F32) return fail(S3R_ERR_INVALID, "linear layers exist on the fp32 path only");
        return S3R_OK;
    }
    if (d->op != S3R_OP_CONV && d->op != S3R_OP_DECONV) return fail(S3R_ERR_INVALID, "unknown op %d", d->op);
    if (d->ndim != 2 && d->ndim != 3) return fail(S3R_ERR_INVALID, "ndim must be 2 or 3");
    if (d->in_size <= 0 || d->k <= 0 || d->stride <= 0 || d->pad < 0) return fail(S3R_ERR_INVALID, "bad size/k/stride/pad");
    if (d->dilation < 0 || d->out_pad < 0) return fail(S3R_ERR_INVALID, "dilation / out_pad must not be negative");
    if (d->act < S3R_ACT_NONE || d->act > S3R_ACT_TANH) return fail(S3R_ERR_INVALID, "unknown activation %d", d->act);
    if (d->dtype == S3R_BF16 && (dil_of(d) != 1 || d->out_pad != 0 || d->act > S3R_ACT_SIGMOID || (d->op == S3R_OP_DECONV && !deconv_fast(d))))
        return fail(S3R_ERR_INVALID, "bf16 path: Conv with dilation 1 / ConvTranspose3d k4 s2 p1 and none / relu / sigmoid only (the "
                    "parameter-general layers are fp32)");
    if (d->op == S3R_OP_CONV && d->out_pad != 0) return fail(S3R_ERR_INVALID, "out_pad belongs to transposed convolutions");
    if (d->op == S3R_OP_DECONV && !deconv_fast(d)) {
        if (dil_of(d) * (d->k - 1) - d->pad < 0)
            return fail(S3R_ERR_INVALID, "ConvTranspose with pad %d > dilation * (k - 1) = %d crops more than the kernel reaches: not supported",
                        d->pad, dil_of(d) * (d->k - 1));
        if (d->out_pad >= (d->stride > dil_of(d) ? d->stride : dil_of(d)))
            return fail(S3R_ERR_INVALID, "out_pad %d must be smaller than max(stride, dilation)", d->out_pad);
    }
    // (a staged layer builds its own halo of any width — pe is not bounded by the caller-halo limit above; what bounds it is the size
    // of the staged copy, checked HERE so that planning and the forward agree: ADVICE r05)
    if (staged_layer(d) && !im2col_layer(d)) {
        const bool tr = d->op == S3R_OP_DECONV && !tclass_layer(d);      // (in double first: the int64 product of staged_geo may not exist)
        if (d->stride > 64 || d->k > 1024) return fail(S3R_ERR_INVALID, "stride / kernel size out of range");
        const double pe = tclass_layer(d) ? (double)tclass_halo(d) : tr ? (double)dil_of(d) * (d->k - 1) - d->pad : d->pad;
        const double sp = (tr ? ((double)d->in_size - 1) * d->stride + 1 + d->out_pad : d->in_size) + 2 * pe;
        const double est = (double)d->batch * ((d->cin + 15) / 16 * 16) * (d->ndim == 3 ? sp * sp * sp : sp * sp);
        if (est >= (double)kMaxElems || est * 4 >= (double)kMaxBytes)
            return fail(S3R_ERR_INVALID, "staged input too large for one call (>= 2^31 elements / 4 GiB): split the batch");
    }
    g->nd = d->ndim;
    g->in = d->in_size;
    g->out = out_size(d);
    if (g->out <= 0) return fail(S3R_ERR_INVALID, "empty output");
    g->in_p = g->in + 2 * d->in_halo;
    g->out_p = g->out + 2 * d->out_halo;
    g->in_sp = ipow(g->in, g->nd);
    g->out_sp = ipow(g->out, g->nd);
    g->x_elems = (int64_t)d->batch * d->cin * ipow(g->in_p, g->nd);
    g->y_elems = (int64_t)d->batch * d->cout * ipow(g->out_p, g->nd);
    if (d->in_layout == S3R_LAYOUT_WINO_H) {     // the four F(2,3)-along-H plane sets a 3 x 3 [x 3] stride-1 pad-1 convolution reads
        if (d->dtype != S3R_F32 || d->op != S3R_OP_CONV || d->stride != 1 || d->k != 3 || d->pad != 1 || (g->in & 1) || d->in_halo != 1 ||
            d->cin % 16 != 0 || d->cout <= 1)
            return fail(S3R_ERR_INVALID, "a Winograd-transformed input serves an fp32 Conv k=3 s=1 p=1 over an even edge, in_halo = 1");
        if (g->in % wino_r(d) != 0) return fail(S3R_ERR_INVALID, "a Winograd-transformed input needs an edge that is a multiple of %d", wino_r(d));
        g->x_elems = (wino_r(d) + 2) * (int64_t)d->batch * d->cin * (g->nd == 3 ? g->in_p : 1) * (g->in / wino_r(d)) * g->in_p;
    }
    if (d->in_layout == S3R_LAYOUT_WINO_DH) {    // the 36 two-axis plane sets a 3 x 3 x 3 stride-1 pad-1 convolution reads
        if (d->dtype != S3R_F32 || d->op != S3R_OP_CONV || d->ndim != 3 || d->stride != 1 || d->k != 3 || d->pad != 1 || (g->in & 3) ||
            d->in_halo != 1 || d->cin % 32 != 0 || d->cout <= 1)
            return fail(S3R_ERR_INVALID, "a two-axis Winograd-transformed input serves an fp32 Conv3d k=3 s=1 p=1 over an edge %% 4 == 0, in_halo = 1");
        g->x_elems = 36 * (int64_t)d->batch * d->cin * (g->in / 4) * (g->in / 4) * g->in_p;
    }
    if (d->in_layout == S3R_LAYOUT_WINO_HW) {    // the 36 two-axis plane sets of a 2D layer, positions flat
        if (d->dtype != S3R_F32 || d->op != S3R_OP_CONV || d->ndim != 2 || d->stride != 1 || d->k != 3 || d->pad != 1 || (g->in & 3) ||
            d->in_halo != 1 || d->cin % 32 != 0 || d->cout <= 1)
            return fail(S3R_ERR_INVALID, "a two-axis Winograd-transformed input (2D) serves an fp32 Conv2d k=3 s=1 p=1 over an edge %% 4 == 0, in_halo = 1");
        g->x_elems = 36 * (int64_t)d->cin * s3r::wino2_npad((int64_t)d->batch * (g->in / 4) * (g->in / 4));
    }
    if (d->out_layout == S3R_LAYOUT_WINO_HW) {   // ... written by the layer in front of it: the plane sets of THIS layer's halo-1 output
        if (d->dtype != S3R_F32 || d->op != S3R_OP_CONV || d->ndim != 2 || (g->out & 3) || d->cout % 32 != 0 || d->act == S3R_ACT_SIGMOID)
            return fail(S3R_ERR_INVALID, "the two-axis Winograd output layout is written by an fp32 Conv2d with an output edge %% 4 == 0 and cout %% 32 == 0");
        g->y_elems = 36 * (int64_t)d->cout * s3r::wino2_npad((int64_t)d->batch * (g->out / 4) * (g->out / 4));
    }
    g->w_elems = (int64_t)d->cin * d->cout * ipow(d->k, g->nd);
    if (d->op == S3R_OP_DECONV)
        g->flops = 2.0 * d->batch * (double)d->cin * g->in_sp * d->cout * ipow(d->k, g->nd);
    else
        g->flops = 2.0 * d->batch * (double)d->cout * g->out_sp * d->cin * ipow(d->k, g->nd);
    // element sizes: the bf16 path reads fp32 renders in its stem and writes fp32 probabilities from its head
    const bool bf = d->dtype == S3R_BF16;
    const bool stem = d->ndim == 2 && d->cin == 3;
    const bool head = d->cout == 1 && d->k == 1;
    const int xs = (bf && !stem) ? 2 : 4, ys = (bf && !head) ? 2 : 4, wsz = bf && !stem && !head ? 2 : 4;
    g->x_store = (g->x_elems * xs + 3) / 4;
    g->y_store = (g->y_elems * ys + 3) / 4;
    g->bytes = (double)d->batch * (xs * d->cin * (double)g->in_sp + ys * d->cout * (double)g->out_sp) + wsz * (double)g->w_elems;
    if (g->x_elems >= kMaxElems || g->y_elems >= kMaxElems || g->x_elems * 4 >= kMaxBytes || g->y_elems * 4 >= kMaxBytes)
        return fail(S3R_ERR_INVALID, "tensor too large for one call (>= 2^31 elements / 4 GiB): split the batch");
    return S3R_OK;
}


// which kernel serves a layer shape (halos are checked separately, by check_halos)
int route(const s3r_conv_desc* d, Route* r) {
    if (d->op == S3R_OP_LINEAR) { *r = R_LINEAR; return S3R_OK; }
    if (stem_shape(d)) { *r = R_STEM; return S3R_OK; }
    if (head_shape(d)) { *r = R_HEAD; return S3R_OK; }
    if (d->dtype == S3R_BF16 ? d->cin % 32 == 0 : (d->cin % 16 == 0 || staged_layer(d))) { *r = R_MFMA; return S3R_OK; }
    return fail(S3R_ERR_INVALID, "no kernel for this layer shape (cin=%d cout=%d k=%d s=%d p=%d ndim=%d): the MFMA path "
                "needs cin %% 16 == 0", d->cin, d->cout, d->k, d->stride, d->pad, d->ndim);
}

// input halo the layer's kernel needs (the MFMA gather reads its zero padding from memory)
int need_halo(const s3r_conv_desc* d, Route r) {
    if (r != R_MFMA || staged_layer(d)) return 0;          // (a staged layer builds its own padded copy)
    return d->op == S3R_OP_DECONV ? 1 : d->pad;
}

// the halo a chain gives the layer's input when it can choose (>= need_halo): a residue-class ConvTranspose with whole K tiles reads
// its border from the producer's zero halo instead of staging a padded copy
int want_halo(const s3r_conv_desc* d, Route r) {
    if (r == R_MFMA && tclass_layer(d) && d->cin % 16 == 0 && tclass_halo(d) <= 8) return tclass_halo(d);
    return need_halo(d, r);
}

int check_halos(const s3r_conv_desc* d, Route r) {
    if ((d->in_layout || d->out_layout) && r != R_MFMA)
        return fail(S3R_ERR_INVALID, "the transformed input layout is read by the MFMA convolution kernels only");
    if (d->in_halo < need_halo(d, r))
        return fail(S3R_ERR_INVALID, "this layer's kernel reads its zero padding from memory: the input must carry a "
                    "zero halo of >= %d (got in_halo=%d); s3r_chain_forward pads unpadded inputs itself",
                    need_halo(d, r), d->in_halo);
    if ((r == R_STEM || r == R_HEAD) && d->in_halo != 0) return fail(S3R_ERR_INVALID, "stem / head kernels take an unpadded input");
    if (r == R_HEAD && d->out_halo != 0) return fail(S3R_ERR_INVALID, "head kernel writes an unpadded output");
    return S3R_OK;
}

int cout_pad(int cout) { return (cout + 127) / 128 * 128; }

// Winograd along H (s3r_conv_wino.hip) for the fp32 3 x 3 [x 3] stride-1 pad-1 convolutions (F(4,3): 1/2 of the matrix work, F(2,3):
// 2/3) and the transposed convolutions (F(2,2) along D and H inside the parity classes: 9/16): another summation order than the direct
// kernels' — same fp32 accuracy, other bits.  Such a layer's packed weights hold BOTH forms (the direct slab, then the class
// slabs); which kernel a call runs is the descriptor's `algo` (include/s3r.h): AUTO resolves from the layer's per-sample
// geometry (and the descriptor's own tile / split-K / layout fields) alone — never from the scratch a caller offers or the
// batch — under the process-level policy S3R_WINO, read once:
//   unset / 1: every layer that has the form (e2, e4, e6, e7, v1, v3, v5, d1, d2, d3 of this network: each measured faster on it at
//   B = 32, and — with the class-parallel launch form on sparse grids — at every smaller batch);  0: never.
int wino_mode() {
    static const int mode = getenv("S3R_WINO") ? atoi(getenv("S3R_WINO")) : 1;      // process-level: read once
    return mode;
}
// structural: the layer has a Winograd form (decides the packed layout; independent of any switch)
bool wino_layer(const s3r_conv_desc* d) {
    return d->dtype != S3R_BF16 && d->op == S3R_OP_CONV && (d->ndim == 2 || d->ndim == 3) && d->k == 3 && d->stride == 1 &&
           d->pad == 1 && d->cin % s3r::wino_bk() == 0 && d->cout > 1 && d->in_size >= 4;
}
bool dwino_layer(const s3r_conv_desc* d) {
    return d->dtype != S3R_BF16 && deconv_fast(d) && d->cin % s3r::wino_bk() == 0 && d->in_size >= 4 && (d->in_size & 3) == 0;
}
// the three-axis form of a transposed layer (s3r_deconv_wino3.hip): whole padded rows per 64-position tile
bool dwino3_layer(const s3r_conv_desc* d) {
    return d->dtype != S3R_BF16 && deconv_fast(d) && d->cin % 16 == 0 && d->cout > 1 && s3r::dwino3_edge_ok(d->in_size);
}
bool dwino3_desc_ok(const s3r_conv_desc* d) {
    return dwino3_layer(d) && d->act < S3R_ACT_SIGMOID && d->in_halo == 1 && d->ksplit <= 1 && d->in_layout == S3R_LAYOUT_PLAIN &&
           d->out_layout == S3R_LAYOUT_PLAIN;
}
// its 8 x 27 class slabs sit behind the direct slab and the two-axis form's
// scratch of the three-axis form: [Dh | Dd | Ddh] then, class-parallel form only, the slabs
Dwino3Need dwino3_need(const s3r_conv_desc* d, int form) {
    Dwino3Need n;
    n.diff = (3 * (int64_t)d->batch * d->cin * ipow(d->in_size + 2, 3) + 255) / 256 * 256;
    const int ntotal = d->batch * (int)ipow(d->in_size / 2, 3);
    n.split = s3r::dwino3_split(d->cout, ntotal, form);
    // (sized for the class-parallel form whenever the LIBRARY picks the launch form: that pick counts workgroups against the current
    // device's compute units, and a size query must not depend on the device it is asked on — ADVICE r05)
    n.total = n.diff + ((n.split || form < 0) ? (s3r::dwino3_slab_elems(d->cout, ntotal) + 255) / 256 * 256 : 0);
    return n;
}
int64_t dwino3_w_offset(const s3r_conv_desc* d) {
    return 64 * (int64_t)d->cin * cout_pad(d->cout) + (dwino_layer(d) ? dwino_w_elems(d) : 0);
}
// the descriptor can run its layer's Winograd form
bool wino_desc_ok(const s3r_conv_desc* d) {
    if (!(wino_layer(d) || dwino_layer(d)) || d->act >= S3R_ACT_SIGMOID || d->in_halo != 1 || d->ksplit > 1 || dil_of(d) != 1) return false;
    if (d->out_layout != S3R_LAYOUT_PLAIN) return false;
    return d->in_layout == S3R_LAYOUT_PLAIN || (d->in_layout == S3R_LAYOUT_WINO_H && wino_layer(d));
}
// Two-axis class-parallel Winograd (s3r_conv_wino.hip): the stride-1 layers with a small edge — Conv3d k3 p1 as F(4,3) x F(4,3)
// over D and H (returns 0), Conv3d k4 p0 as F(2,4) x F(2,4) (returns 1), Conv2d k3 p1 as F(4,3) x F(4,3) over H and W (returns 2);
// -1: the layer has no such form
int wino2_ax(const s3r_conv_desc* d) {
    if (d->dtype == S3R_BF16 || d->op != S3R_OP_CONV || d->stride != 1 || d->cin % s3r::wino_bk() != 0 || d->cout <= 1) return -1;
    // (2D: the finish kernel stages a whole padded output plane in 64 KiB of LDS)
    if (d->ndim == 2) return d->k == 3 && d->pad == 1 && d->in_size >= 4 && d->in_size <= 124 ? 2 : -1;
    if (d->ndim != 3) return -1;
    if (d->k == 3 && d->pad == 1 && d->in_size >= 4) return 0;
    if (d->k == 4 && d->pad == 0 && d->in_size >= 5) return 1;
    return -1;
}
bool wino2_desc_ok(const s3r_conv_desc* d) {
    return wino2_ax(d) >= 0 && d->act < S3R_ACT_SIGMOID && dil_of(d) == 1 && d->in_halo == d->pad && d->ksplit <= 1 &&
           (d->in_layout == S3R_LAYOUT_PLAIN || (d->in_layout == S3R_LAYOUT_WINO_DH && wino2_ax(d) == 0 && d->in_size % 4 == 0) ||
            (d->in_layout == S3R_LAYOUT_WINO_HW && wino2_ax(d) == 2 && d->in_size % 4 == 0)) &&
           (d->out_layout == S3R_LAYOUT_PLAIN || (d->out_layout == S3R_LAYOUT_WINO_HW && wino2_ax(d) == 2 && d->in_size % 4 == 0));
}
// library policy: the two-axis form where the output is small enough for its class slabs (ncls / m^2 x the output) to be cheap
// or, in its semi-fused launch form (6 / 4 x the output), worth the halved matrix work — v1 (edge 28), v3 (14), v5, v6 (7) of this
// network: every 3D stride-1 layer; S3R_WINO2_MAX_EDGE (read once) moves the bound for experiments
int wino2_max_edge() {
    static const int e = getenv("S3R_WINO2_MAX_EDGE") ? atoi(getenv("S3R_WINO2_MAX_EDGE")) : 28;
    return e;
}
// LeakyReLU with a slope in [0, 1] is max(t, slope t): the direct kernel's epilogue and its split-K finish apply it themselves (r06: a
// DispNet-style network has it behind every convolution; as a pass of its own it cost every layer a round trip of its output)
bool leaky_fused(const s3r_conv_desc* d) {
    return d->dtype == S3R_F32 && d->op != S3R_OP_LINEAR && d->act == S3R_ACT_LEAKY_RELU && d->act_param >= 0.f && d->act_param <= 1.f;
}
bool needs_act_pass(const s3r_conv_desc* d) { return d->act > S3R_ACT_SIGMOID && !leaky_fused(d); }

// the algorithm a descriptor resolves to; *form = the forced launch form of the one-axis kernel, or -1
int resolve_algo(const s3r_conv_desc* d, int* alg, int* form) {
    *alg = ALG_DIRECT;
    *form = -1;
    if (d->algo != S3R_ALGO_AUTO && d->algo != S3R_ALGO_DIRECT && d->algo != S3R_ALGO_WINOGRAD)
        return fail(S3R_ERR_INVALID, "unknown algo %d", d->algo);
    if (staged_layer(d) || d->act > S3R_ACT_SIGMOID || dil_of(d) != 1) {      // parameter-general layers: the direct kernel
        if (d->algo == S3R_ALGO_WINOGRAD) return fail(S3R_ERR_INVALID, "algo = WINOGRAD: a parameter-general layer (staged input, dilation, "
                                                      "LeakyReLU / ELU / Tanh) has no Winograd form");
        if (d->in_layout != S3R_LAYOUT_PLAIN || d->out_layout != S3R_LAYOUT_PLAIN)
            return fail(S3R_ERR_INVALID, "a parameter-general layer reads and writes plain layouts");
        return S3R_OK;
    }
    if (d->algo == S3R_ALGO_WINOGRAD && d->tile >= 6 && d->tile <= 8) {        // the three-axis form of a transposed layer
        if (!dwino3_desc_ok(d))
            return fail(S3R_ERR_INVALID, "algo = WINOGRAD, tile = 6: the three-axis form serves an fp32 ConvTranspose3d k4 s2 p1 over an edge of "
                        "8, 16 or 32 with cin %% 16 == 0 (in_halo = 1, plain layouts, no split-K, no sigmoid)");
        *alg = ALG_WINO3;
        *form = d->tile == 6 ? -1 : d->tile == 7 ? 1 : 0;      // 6: the library's launch form; 7: class-parallel; 8: serial (same bits)
        return S3R_OK;
    }
    if (d->algo == S3R_ALGO_WINOGRAD) {
        const bool one = wino_desc_ok(d), two = wino2_desc_ok(d);
        // tile: -1 the library's pick between the forms the layer has; 0, 1, 2 a launch form of the one-axis kernel; 3 the two-axis
        // algorithm (4: its class-parallel form, 5: its semi-fused form)
        if ((d->tile >= 3 && !two) || (d->tile >= 0 && d->tile <= 2 && !one) || (!one && !two) || d->tile > 5 ||
            (d->tile == 5 && (wino2_ax(d) == 1 || (wino2_ax(d) == 0 && d->in_size > 60))))      // (semi-fused 3D: four padded slices in LDS)
            return fail(S3R_ERR_INVALID, "algo = WINOGRAD: this layer / descriptor has no such Winograd form (one-axis: fp32 Conv k3 s1 p1 "
                        "with cin %% %d == 0 and edge >= 4, or ConvTranspose3d k4 s2 p1 over an edge %% 4 == 0, in_halo = 1; two-axis "
                        "(tile = 3): Conv3d k3 s1 p1 / k4 s1 p0, in_halo = pad; plain layouts, no split-K, no sigmoid)", s3r::wino_bk());
        const bool two_io = d->in_layout == S3R_LAYOUT_WINO_DH || d->in_layout == S3R_LAYOUT_WINO_HW || d->out_layout == S3R_LAYOUT_WINO_HW;
        if (two_io && !(two && (d->tile < 0 || d->tile >= 3)))
            return fail(S3R_ERR_INVALID, "a two-axis transformed input / output runs the two-axis kernel only");
        if (d->tile >= 3 || !one || two_io ||
            (d->tile < 0 && two && d->in_layout == S3R_LAYOUT_PLAIN && d->in_size <= wino2_max_edge())) {
            *alg = ALG_WINO2;
            *form = d->tile >= 4 ? d->tile - 4 : -1;
        } else { *alg = ALG_WINO; *form = d->tile; }
        return S3R_OK;
    }
    if (d->in_layout == S3R_LAYOUT_WINO_DH || d->in_layout == S3R_LAYOUT_WINO_HW || d->out_layout == S3R_LAYOUT_WINO_HW) {
        // only the two-axis kernel reads / writes the 36 plane sets
        if (d->algo == S3R_ALGO_DIRECT || !wino2_desc_ok(d) || d->tile >= 0)
            return fail(S3R_ERR_INVALID, "a two-axis transformed input / output runs the two-axis kernel only: algo AUTO / WINOGRAD, no direct "
                        "tile / split-K override");
        *alg = ALG_WINO2;
        return S3R_OK;
    }
    if (d->in_layout == S3R_LAYOUT_WINO_H) {             // only the one-axis Winograd kernel reads the transformed planes
        if (d->algo == S3R_ALGO_DIRECT || !wino_desc_ok(d) || d->tile >= 0)
            return fail(S3R_ERR_INVALID, "a Winograd-transformed input runs the Winograd kernel only: algo AUTO / WINOGRAD, no direct tile / "
                        "split-K override, a plain output");
        *alg = ALG_WINO;
        return S3R_OK;
    }
    if (d->algo == S3R_ALGO_DIRECT || d->tile >= 0 || d->ksplit >= 1 || wino_mode() <= 0) return S3R_OK;
    if (wino2_desc_ok(d) && d->in_size <= wino2_max_edge()) *alg = ALG_WINO2;
    // transposed layers with an edge >= 16: the three-axis form (27 / 64 of the multiplications).  r05, d3 (16^3 -> 32^3, fused head)
    // inside the forward: 0.694 -> 0.622 ms at B = 32, -4 % at B = 8 / 16, equal at 4, +5 % at B = 1 / 2 (serial form only: 64
    // workgroups); at edge 8 (d2) its one round of long workgroups loses to the two-axis form's dual launch (0.438 vs 0.380 ms)
    else if (d->op == S3R_OP_DECONV && dwino3_desc_ok(d) && d->in_size >= 16) *alg = ALG_WINO3;
    else if (wino_desc_ok(d)) *alg = ALG_WINO;
    return S3R_OK;
}
bool resolves_to_wino(const s3r_conv_desc* d) {
    int a, f;
    return resolve_algo(d, &a, &f) == S3R_OK && a != ALG_DIRECT;
}
// ---- two-axis form: sizes
Wino2Geo wino2_geo(const s3r_conv_desc* d) {
    Wino2Geo w;
    w.ax = wino2_ax(d);
    w.m = s3r::wino2_outputs(w.ax);
    w.ncls = s3r::wino2_classes(w.ax);
    w.out = out_size(d);
    w.sg = (w.out + w.m - 1) / w.m;                       // groups per axis
    w.wp = d->in_size + 2 * d->in_halo;
    w.kw = w.ax == 2 ? 1 : d->k;                          // column taps left to the class kernel
    w.w_elems = (int64_t)w.ncls * w.kw * d->cin * cout_pad(d->cout);
    // ax 2: V is [36][Cin][positions of the sub-batch rounded up to a GEMM tile]: v_sample is the bound used to size a sub-batch
    w.v_sample = w.ax == 2 ? (int64_t)w.ncls * d->cin * (w.sg * w.sg + 64) : (int64_t)w.ncls * d->cin * w.sg * w.sg * w.wp;
    w.pos_sample = w.ax == 2 ? (int64_t)w.sg * w.sg : (int64_t)w.sg * w.sg * w.out;
    const int64_t mx = w.v_sample > 0 ? (((int64_t)1 << 31) - 1) / (4 * w.v_sample) : 0;
    w.bmax = (int)(mx < d->batch ? mx : d->batch);
    w.n = 0;
    return w;
}
int64_t wino_w_elems(const s3r_conv_desc* d) {       // the R + 2 class slabs behind the direct slab
    return (wino_r(d) + 2) * ipow(3, d->ndim - 1) * d->cin * (int64_t)cout_pad(d->cout);
}
int64_t wino_v_elems(const s3r_conv_desc* d) {       // the transformed plane sets: [R + 2][B][Cin][Dp][ceil(H / R)][Wp]
    const int R = wino_r(d);
    const int64_t dp = d->ndim == 3 ? d->in_size + 2 : 1, hq = (d->in_size + R - 1) / R, wp = d->in_size + 2;
    return (R + 2) * (int64_t)d->batch * d->cin * dp * hq * wp;
}
// samples per Winograd call (the transformed input of a call stays below 2 GiB)
int wino_bmax(const s3r_conv_desc* d) {
    const int64_t v_sample = wino_v_elems(d) / (d->batch > 0 ? d->batch : 1);
    const int64_t m = v_sample > 0 ? (((int64_t)1 << 31) - 1) / (4 * v_sample) : 0;
    return (int)(m < d->batch ? m : d->batch);
}
int64_t dwino_w_elems(const s3r_conv_desc* d) { return 72 * 2 * (int64_t)d->cin * cout_pad(d->cout); }      // (parity class, class) x 2 taps
// The depth differences of the transposed Winograd form are materialised (two more tensors behind the row differences) while
// the four tensors stay in the Infinity Cache, and formed inside the class kernel otherwise: same bits either way, so this
// may follow the batch (s3r_conv_wino.hip).
bool dwino_materialise(const s3r_conv_desc* d) {
    static const int forced = getenv("S3R_DWINO_MAT") ? atoi(getenv("S3R_DWINO_MAT")) : -1;      // A/B switch, read once
    if (forced >= 0) return forced != 0;
    return 4 * 4 * (int64_t)d->batch * d->cin * ipow(d->in_size + 2, 3) <= (int64_t)192 << 20;
}
int64_t dwino_d_elems(const s3r_conv_desc* d) {
    return (dwino_materialise(d) ? 3 : 1) * (int64_t)d->batch * d->cin * ipow(d->in_size + 2, 3);
}

// Scratch of a Winograd call: [transformed input V (a convolution fed with plain input) | the difference tensors (transposed) ] then
// the class-parallel slabs of the launch form the library plans for this batch (every form gives the same bits, so the form
// — unlike the algorithm — may follow the batch).
int wino_kind(const s3r_conv_desc* d) { return d->op == S3R_OP_DECONV ? 2 : 1; }
int wino_kcls(const s3r_conv_desc* d) {                        // K per class: Cin x (depth taps x column taps)
    return d->cin * (d->op == S3R_OP_DECONV ? 2 : (d->ndim == 3 ? 9 : 3));
}
int64_t wino_positions(const s3r_conv_desc* d, int nb) {       // GEMM positions (groups of R output rows) of nb samples
    const int n = d->in_size;
    if (d->op == S3R_OP_DECONV) return (int64_t)nb * (n / 2) * (n / 2) * n;
    const int R = wino_r(d);
    return (int64_t)nb * (d->ndim == 3 ? n : 1) * ((n + R - 1) / R) * n;
}
WinoNeed wino_need(const s3r_conv_desc* d, int form, bool head) {
    WinoNeed w = {0, 0, 0};
    if (d->batch <= 0) return w;
    const int kind = wino_kind(d);
    if (d->op == S3R_OP_DECONV) {
        w.v = dwino_d_elems(d);
        const int nt = (int)wino_positions(d, d->batch);
        // (form < 0, the library's pick: sized for the class-parallel form, the largest — the pick depends on the device's CU count)
        w.slab = s3r::wino_slab_elems(kind, d->cout, nt, s3r::wino_plan(kind, d->cout, wino_kcls(d), nt, head, form < 0 ? 1 : form));
    } else {
        const int bmax = wino_bmax(d);
        if (bmax <= 0) return w;
        if (d->in_layout != S3R_LAYOUT_WINO_H) w.v = wino_v_elems(d) / d->batch * bmax;
        for (int b0 = 0; b0 < d->batch; b0 += bmax) {          // (at most two different sub-batch sizes)
            const int nb = d->batch - b0 < bmax ? d->batch - b0 : bmax;
            if (b0 > 0 && nb == bmax) continue;
            const int nt = (int)wino_positions(d, nb);
            const int64_t sl = s3r::wino_slab_elems(kind, d->cout, nt, s3r::wino_plan(kind, d->cout, wino_kcls(d), nt, false, form < 0 ? 1 : form));
            if (sl > w.slab) w.slab = sl;
        }
    }
    w.v = (w.v + 255) / 256 * 256;
    w.total = w.v + w.slab;
    return w;
}
// MFMA FLOPs the Winograd form executes for the whole batch
double wino_exec_flops(const s3r_conv_desc* d, const Geo& g) {
    if (d->op == S3R_OP_DECONV) return g.flops * 0.5625;
    const int R = wino_r(d);
    const double taps = (d->ndim == 3 ? 3.0 : 1.0) * 3.0 * (R + 2);
    return 2.0 * (double)wino_positions(d, d->batch) * d->cout * d->cin * taps;
}

// the launch form of a two-axis call (0 class-parallel, 1 semi-fused): the library's plan unless forced; a 3D layer whose four padded
// output slices do not fit the semi-fused finish kernel's LDS stays class-parallel
int wino2_form_of(const s3r_conv_desc* d, int ntotal, int forced) {
    if (forced < 0 && wino2_ax(d) == 0 && d->in_size > 60) forced = 0;
    return s3r::wino2_form(wino2_ax(d), d->cout, ntotal, forced);
}
// scratch of a two-axis call: [V of one sub-batch (unless the producer wrote it) | slabs of the launch form planned for the batch]
WinoNeed wino2_need(const s3r_conv_desc* d, int form) {
    WinoNeed w = {0, 0, 0};
    const Wino2Geo g2 = wino2_geo(d);
    if (d->batch <= 0 || g2.bmax <= 0) return w;
    if (d->in_layout != S3R_LAYOUT_WINO_DH && d->in_layout != S3R_LAYOUT_WINO_HW)
        w.v = ((g2.ax == 2 ? g2.ncls * d->cin * s3r::wino2_npad(g2.pos_sample * g2.bmax) : g2.v_sample * g2.bmax) + 255) / 256 * 256;
    for (int b0 = 0; b0 < d->batch; b0 += g2.bmax) {              // (at most two different sub-batch sizes)
        const int nb = d->batch - b0 < g2.bmax ? d->batch - b0 : g2.bmax;
        if (b0 > 0 && nb == g2.bmax) continue;
        const int nt = (int)(g2.pos_sample * nb);
        // (form < 0: sized for the class-parallel form — 36 / 25 class slabs against the semi-fused form's 24)
        const int64_t sl = s3r::wino2_slab_elems(g2.ax, d->cout, nt, form < 0 ? 0 : wino2_form_of(d, nt, form));
        if (sl > w.slab) w.slab = sl;
    }
    w.total = w.v + w.slab;
    return w;
}
double wino2_exec_flops(const s3r_conv_desc* d) {
    const Wino2Geo g2 = wino2_geo(d);
    return 2.0 * (double)g2.pos_sample * d->batch * g2.ncls * d->cout * d->cin * g2.kw;
}
int cout_pad_h(int cout) { return (cout + 63) / 64 * 64; }

// bf16 channels-last twin of make_params: strides are in elements of (B, Dp, Hp, Wp, C)
s3r::ConvParamsH make_params_h(const s3r_conv_desc* d, const Geo& g) {
    s3r::ConvParamsH p;
    memset(&p, 0, sizeof(p));
    const bool is3 = d->ndim == 3;
    p.B = d->batch; p.Cin = d->cin; p.Cout = d->cout; p.CoutPad = cout_pad_h(d->cout);
    p.act = d->act;
    p.x_ws = d->cin; p.x_hs = g.in_p * d->cin; p.x_ds = is3 ? g.in_p * g.in_p * d->cin : 0;
    p.x_bs = (int)ipow(g.in_p, g.nd) * d->cin;
    p.y_ws = d->cout; p.y_hs = g.out_p * d->cout; p.y_ds = is3 ? g.out_p * g.out_p * d->cout : 0;
    p.y_bs = (int)ipow(g.out_p, g.nd) * d->cout;
    p.y_org = d->out_halo * (p.y_ds + p.y_hs + p.y_ws);
    p.x_bytes = (unsigned)(g.x_elems * 2);
    if (d->op == S3R_OP_DECONV) {
        p.transposed = 1;
        p.Nd = g.in; p.Nh = g.in; p.Nw = g.in;
        p.kd = p.kh = p.kw = 2; p.T = 8;
        p.stride = 1;
        p.x_org = d->in_halo * (p.x_ds + p.x_hs + p.x_ws);
    } else {
        p.transposed = 0;
        p.Nd = is3 ? g.out : 1; p.Nh = g.out; p.Nw = g.out;
        p.kd = is3 ? d->k : 1; p.kh = d->k; p.kw = d->k; p.T = p.kd * p.kh * p.kw;
        p.stride = d->stride;
        p.x_org = (d->in_halo - d->pad) * (p.x_ds + p.x_hs + p.x_ws);
    }
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.dDH = s3r::FastDiv((unsigned)(p.Nd * p.Nh));
    p.dH = s3r::FastDiv((unsigned)p.Nh);
    p.ksplit = 1;
    return p;
}

// (position-tile multiplier TM, split-K) of a bf16 MFMA layer

int resolve_launch_h(const s3r_conv_desc* d, s3r::ConvParamsH* p, LaunchH* L) {
    const int chunks = d->cin / 32;
    if (d->ksplit < 0 || (d->ksplit > 0 && chunks % d->ksplit != 0))
        return fail(S3R_ERR_INVALID, "ksplit=%d must divide cin/32=%d", d->ksplit, chunks);
    L->ksplit = d->ksplit > 0 ? d->ksplit : s3r::conv_bf16_pick_ksplit(*p);
    p->ksplit = L->ksplit;
    if (d->tile >= 0 && d->tile != 1 && d->tile != 2 && d->tile != 3 && d->tile != 19 && d->tile != 4 && d->tile != 5 && d->tile != 6 && d->tile != 9 &&
        d->tile != 10 && d->tile != 17 && d->tile != 18 && d->tile != 20 && d->tile != 21 && d->tile != 22 && d->tile != 23 &&
        d->tile != 40)
        return fail(S3R_ERR_INVALID, "bf16 path: tile must be -1 (auto), 1, 2, 4 (x128 positions, per-tap gather), 3 (128 x 128 couts), 5, 6 "
                    "(x128 positions = 1, 2, plane-reuse gather) or 9, 10 (row-reuse gather); per-tap / plane + 16 = "
                    "32-channel K tiles; 40 (row-persistent e2)");
    L->tm = d->tile >= 0 ? d->tile : s3r::conv_bf16_pick_tm(*p);
    return S3R_OK;
}

s3r::ConvParams make_params(const s3r_conv_desc* d, const Geo& g) {
    s3r::ConvParams p;
    memset(&p, 0, sizeof(p));
    const bool is3 = d->ndim == 3;
    p.B = d->batch; p.Cin = d->cin; p.Cout = d->cout; p.CoutPad = cout_pad(d->cout);
    p.act = d->act;
    // strides of the padded buffers; a 2D layer has no depth axis (x_ds = y_ds = 0, Nd = kd = 1)
    p.x_hs = g.in_p; p.x_ds = is3 ? g.in_p * g.in_p : 0; p.x_cs = (int)ipow(g.in_p, g.nd);
    p.y_hs = g.out_p; p.y_ds = is3 ? g.out_p * g.out_p : 0; p.y_cs = (int)ipow(g.out_p, g.nd);
    p.y_org = d->out_halo * (p.y_ds + p.y_hs + 1);
    p.y_bs = d->cout * p.y_cs;
    p.x_bytes = (unsigned)(g.x_elems * 4);
    p.y_bytes = (unsigned)(g.y_elems * 4);
    if (d->op == S3R_OP_DECONV) {
        p.transposed = 1;
        p.Nd = g.in; p.Nh = g.in; p.Nw = g.in;
        p.kd = p.kh = p.kw = 2; p.T = 8;
        p.stride = 1;
        p.x_org = d->in_halo * (p.x_ds + p.x_hs + 1);
    } else {
        p.transposed = 0;
        p.Nd = is3 ? g.out : 1; p.Nh = g.out; p.Nw = g.out;
        p.kd = is3 ? d->k : 1; p.kh = d->k; p.kw = d->k; p.T = p.kd * p.kh * p.kw;
        p.stride = d->stride;
        p.x_org = (d->in_halo - d->pad) * (p.x_ds + p.x_hs + 1);
    }
    p.dil = dil_of(d);
    if (leaky_fused(d)) { p.act = S3R_ACT_RELU; p.slope = d->act_param; }      // max(t, slope t) in the direct kernel's epilogue
    else if (p.act > S3R_ACT_SIGMOID) p.act = S3R_ACT_NONE;     // (ELU / Tanh, LeakyReLU with a slope outside [0, 1]: launch_act behind the convolution)
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    return p;
}

// a staged layer as the direct kernel sees it: a convolution (stride 1 if the layer is transposed) over the staged tensor, whose
// halo is exactly the effective padding
s3r::ConvParams make_params_staged(const s3r_conv_desc* d, const Geo& g) {
    s3r::ConvParams p = make_params(d, g);
    const StagedGeo sg = staged_geo(d);
    const bool is3 = d->ndim == 3;
    p.Cin = sg.cin_pad;
    p.transposed = 0;
    p.x_hs = sg.sp; p.x_ds = is3 ? sg.sp * sg.sp : 0; p.x_cs = (int)ipow(sg.sp, g.nd);
    p.x_org = 0;
    p.x_bytes = (unsigned)(sg.elems * 4);
    p.Nd = is3 ? g.out : 1; p.Nh = g.out; p.Nw = g.out;
    if (im2col_layer(d)) {                        // the unfolded tensor: a 1 x 1 GEMM, one position per output
        p.kd = p.kh = p.kw = 1; p.T = 1;
        p.stride = 1; p.dil = 1;
    } else {
        p.kd = is3 ? d->k : 1; p.kh = d->k; p.kw = d->k; p.T = p.kd * p.kh * p.kw;
        p.stride = d->op == S3R_OP_DECONV ? 1 : d->stride;
    }
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    return p;
}

s3r::ConvParams make_params_tshuf(const s3r_conv_desc* d, const Geo& g) {
    const bool is3 = d->ndim == 3;
    StagedGeo sg = staged_geo(d);                 // (k == stride, pad 0: the classes read no halo; staged only to pad the channels)
    if (tclass_direct(d)) {
        sg.pe = d->in_halo;
        sg.sp = d->in_size + 2 * d->in_halo;
        sg.elems = (int64_t)d->batch * sg.cin_pad * ipow(sg.sp, d->ndim);
    }
    const int s = d->stride, sc = (int)ipow(s, d->ndim);
    s3r::ConvParams p = make_params(d, g);
    p.Cin = sg.cin_pad;
    p.Cout = d->cout * sc;                        // GEMM rows: (cout, tap); y_bs / y_cs stay the real output's
    p.CoutPad = cout_pad(p.Cout);
    p.transposed = 0;
    p.x_hs = sg.sp; p.x_ds = is3 ? sg.sp * sg.sp : 0; p.x_cs = (int)ipow(sg.sp, g.nd);
    p.x_bytes = (unsigned)(sg.elems * 4);
    p.x_org = sg.pe * (p.x_ds + p.x_hs + 1);
    p.Nd = is3 ? d->in_size : 1; p.Nh = d->in_size; p.Nw = d->in_size;
    p.kd = p.kh = p.kw = 1; p.T = 1;
    p.stride = 1; p.dil = 1;
    p.y_step = s;
    p.shuf_s = s; p.shuf_nd = d->ndim;
    p.dSC = s3r::FastDiv((unsigned)sc);
    p.dS1 = s3r::FastDiv((unsigned)s);
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    return p;
}

bool make_params_tclass(const s3r_conv_desc* d, const Geo& g, int rd, int rh, int rw, s3r::ConvParams* q, int64_t* w_off, double* macs) {
    const bool is3 = d->ndim == 3;
    StagedGeo sg = staged_geo(d);
    if (tclass_direct(d)) {                       // the layer's own input, read in place: its halo is the classes' border
        sg.pe = d->in_halo;
        sg.sp = d->in_size + 2 * d->in_halo;
        sg.elems = (int64_t)d->batch * sg.cin_pad * ipow(sg.sp, d->ndim);
    }
    const int s = d->stride, h = sg.pe;
    const TClassAxis ad = is3 ? tclass_axis(d, rd) : TClassAxis{1, 1, 0, 1}, ah = tclass_axis(d, rh), aw = tclass_axis(d, rw);
    // slab offset: classes in (rd, rh, rw) order, each prod(ke) taps
    int64_t taps_before = 0;
    {
        int64_t sum_ke = 0;
        for (int r = 0; r < s; ++r) sum_ke += tclass_axis(d, r).ke;
        int64_t kd_before = 0, kh_before = 0, kw_before = 0;
        for (int r = 0; r < rd; ++r) kd_before += tclass_axis(d, r).ke;
        for (int r = 0; r < rh; ++r) kh_before += tclass_axis(d, r).ke;
        for (int r = 0; r < rw; ++r) kw_before += tclass_axis(d, r).ke;
        // classes before (rd, rh, rw): all (rd' < rd, *, *), then (rd, rh' < rh, *), then (rd, rh, rw' < rw)
        taps_before = (is3 ? kd_before * sum_ke * sum_ke : 0) + (int64_t)ad.ke * kh_before * sum_ke + (int64_t)ad.ke * ah.ke * kw_before;
    }
    *w_off = taps_before * sg.cin_pad * cout_pad(d->cout);
    if (ad.nq <= 0 || ah.nq <= 0 || aw.nq <= 0) return false;
    s3r::ConvParams p = make_params(d, g);
    p.Cin = sg.cin_pad;
    p.transposed = 0;
    p.x_hs = sg.sp; p.x_ds = is3 ? sg.sp * sg.sp : 0; p.x_cs = (int)ipow(sg.sp, g.nd);
    p.x_bytes = (unsigned)(sg.elems * 4);
    p.Nd = is3 ? ad.nq : 1; p.Nh = ah.nq; p.Nw = aw.nq;
    p.kd = is3 ? ad.ke : 1; p.kh = ah.ke; p.kw = aw.ke; p.T = p.kd * p.kh * p.kw;
    p.stride = 1; p.dil = 1;
    // position u of an axis is q = qmin + u: it reads staged indices h + q - (ke - 1) .. h + q and writes output s q + r - pad
    p.x_org = (is3 ? (h + ad.qmin - (ad.ke - 1)) * p.x_ds : 0) + (h + ah.qmin - (ah.ke - 1)) * p.x_hs + (h + aw.qmin - (aw.ke - 1));
    p.y_step = s;
    p.y_org += (is3 ? (s * ad.qmin + rd - d->pad) * p.y_ds : 0) + (s * ah.qmin + rh - d->pad) * p.y_hs + (s * aw.qmin + rw - d->pad);
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    *macs = (double)p.Ntotal * p.T * (double)sg.cin_pad * d->cout;
    *q = p;
    return true;
}

// (tile cfg, gather width, split-K) of an MFMA-route layer: the caller's forced values or the heuristics

int resolve_launch(const s3r_conv_desc* d, s3r::ConvParams* p, Launch* L) {
    const int chunks = d->cin / 16;
    if (d->ksplit < 0 || (d->ksplit > 0 && chunks % d->ksplit != 0))
        return fail(S3R_ERR_INVALID, "ksplit=%d must divide cin/16=%d", d->ksplit, chunks);
    L->ksplit = d->ksplit > 0 ? d->ksplit : s3r::conv_pick_ksplit(*p, 0);
    p->ksplit = L->ksplit;
    const int code = d->tile >= 0 ? d->tile : 15;
    L->cfg = code & 15;
    L->vec = code >> 4;
    if (L->cfg != 15 && L->cfg >= s3r::conv_num_tiles()) return fail(S3R_ERR_INVALID, "unknown tile configuration %d", L->cfg);
    if (L->cfg == 15) L->cfg = s3r::conv_pick_tile(*p);
    return S3R_OK;
}

// ---------------------------------------------------------------- chain planning
// A chain gives every intermediate activation its own region of the caller's workspace, with the zero
// halo the NEXT layer's gather wants; regions are written interior-only, so the halos stay zero from
// the one memset that initialises the workspace (ws_fresh).

int64_t align_up(int64_t v, int64_t a) { return (v + a - 1) / a * a; }

int plan_chain(const s3r_layer* layers, int n, Plan* pl) {
    if (!layers || n <= 0) return fail(S3R_ERR_INVALID, "empty chain");
    pl->d.resize(n); pl->r.resize(n); pl->g.resize(n); pl->off.assign(n, -1); pl->fuse_head.assign(n, 0);
    for (int i = 0; i < n; ++i) {
        pl->d[i] = layers[i].desc;
        if (i > 0) pl->d[i].in_layout = S3R_LAYOUT_PLAIN;            // intermediates: the library's plan, not the caller's
        if (i + 1 < n) pl->d[i].out_layout = S3R_LAYOUT_PLAIN;
        int rc = route(&pl->d[i], &pl->r[i]);
        if (rc) return rc;
    }
    const int need0 = need_halo(&pl->d[0], pl->r[0]);
    pl->pad_input = pl->d[0].in_halo < need0;
    const int user_in_halo = pl->d[0].in_halo;
    if (pl->pad_input) pl->d[0].in_halo = need0;
    for (int i = 0; i < n; ++i) {
        if (i > 0) pl->d[i].in_halo = pl->d[i - 1].out_halo;
        if (i + 1 < n) pl->d[i].out_halo = want_halo(&pl->d[i + 1], pl->r[i + 1]);
        if (i == 1 && pl->r[0] == R_STEM && pl->r[1] == R_MFMA && pl->d[1].dtype == S3R_F32 && pl->d[1].op == S3R_OP_CONV) {
            // a stem feeding the one-axis Winograd kernel writes that kernel's planes itself (the same bits: the stem's values
            // through wino_input_kernel's transform): no plain activation, no transform launch
            static const int fuse = getenv("S3R_STEM_WINO") ? atoi(getenv("S3R_STEM_WINO")) : 1;      // A/B switch, read once
            int alg, form;
            // (launch_stem_wino's own limits: 13 input rows of three channels staged in LDS by at most 9 pieces per thread)
            const bool fits = pl->d[0].in_size % 16 == 0 && pl->d[0].in_size <= 236;
            if (fuse && fits && resolve_algo(&pl->d[1], &alg, &form) == S3R_OK && alg == ALG_WINO &&
                pl->d[1].in_size % wino_r(&pl->d[1]) == 0 && wino_bmax(&pl->d[1]) >= pl->d[1].batch) {
                pl->stem_wino = true;
                pl->d[1].in_layout = S3R_LAYOUT_WINO_H;
            }
        }
        if (i > 0 && pl->r[i - 1] == R_MFMA && pl->r[i] == R_MFMA && pl->d[i].dtype == S3R_F32) {
            // two-axis Conv2d -> two-axis Conv2d over the same edge: the first one's finish kernel writes the second one's plane sets
            // (S3R_LAYOUT_WINO_HW: the bits wino2p_input_kernel makes of the plain activation)
            static const int fuse = getenv("S3R_WINO_HANDOFF") ? atoi(getenv("S3R_WINO_HANDOFF")) : 1;      // A/B switch, read once
            s3r_conv_desc& a = pl->d[i - 1];
            s3r_conv_desc& b = pl->d[i];
            int alg_a, alg_b, form;
            if (fuse && wino2_ax(&a) == 2 && wino2_ax(&b) == 2 && a.cout == b.cin && out_size(&a) == b.in_size && b.in_size % 4 == 0 &&
                resolve_algo(&a, &alg_a, &form) == S3R_OK && alg_a == ALG_WINO2 && resolve_algo(&b, &alg_b, &form) == S3R_OK &&
                alg_b == ALG_WINO2 && wino2_geo(&a).bmax >= a.batch && wino2_geo(&b).bmax >= b.batch) {
                a.out_layout = S3R_LAYOUT_WINO_HW;
                b.in_layout = S3R_LAYOUT_WINO_HW;
                const int rg = geometry(&a, &pl->g[i - 1]);              // (its output is now the plane sets)
                if (rg) return rg;
            }
        }
        int rc = geometry(&pl->d[i], &pl->g[i]);
        if (rc) return rc;
        if ((rc = check_halos(&pl->d[i], pl->r[i]))) return rc;
        if (pl->d[i].dtype != pl->d[0].dtype) return fail(S3R_ERR_INVALID, "all layers of a chain must share one dtype");
        if (i > 0) {   // shapes must chain
            const s3r_conv_desc& a = pl->d[i - 1];
            const int64_t prev_out = (int64_t)a.cout * pl->g[i - 1].out_sp, cur_in = (int64_t)pl->d[i].cin * pl->g[i].in_sp;      // (logical sizes)
            if (prev_out != cur_in || a.batch != pl->d[i].batch)
                return fail(S3R_ERR_INVALID, "layer %d input (%lld/sample) does not match layer %d output (%lld/sample)", i,
                            (long long)cur_in, i - 1, (long long)prev_out);
            if (pl->d[i].in_halo && (pl->g[i - 1].out != pl->g[i].in || a.cout != pl->d[i].cin))
                return fail(S3R_ERR_INVALID, "layer %d needs a halo but reshapes layer %d's output", i, i - 1);
        }
    }
    // conv -> pointwise head fusion (fp32 path): a <=64-cout MFMA conv that does not split K, followed by the
    // 1x1 single-channel head, runs the head inside its epilogue; its own output is never materialised
    for (int i = 0; i + 1 < n; ++i) {
        if (pl->r[i] != R_MFMA || pl->r[i + 1] != R_HEAD || pl->d[i].cout > 64) continue;
        if (pl->d[i + 1].cin != pl->d[i].cout || pl->d[i].out_halo != 0 || pl->d[i].act >= S3R_ACT_SIGMOID || staged_layer(&pl->d[i])) continue;
        if (pl->d[i].op == S3R_OP_CONV && resolves_to_wino(&pl->d[i])) continue;     // (the Winograd conv kernel has no fused-head epilogue)
        if (pl->d[i].dtype == S3R_BF16) {
            s3r::ConvParamsH ph = make_params_h(&pl->d[i], pl->g[i]);
            LaunchH Lh;
            if (resolve_launch_h(&pl->d[i], &ph, &Lh) != S3R_OK || Lh.ksplit != 1) continue;
        } else {
            int alg, form;
            if (resolve_algo(&pl->d[i], &alg, &form) != S3R_OK) continue;
            if (alg == ALG_DIRECT) {                     // (a Winograd form's `tile` is a launch-form code, not a tile of the direct kernel)
                s3r::ConvParams p = make_params(&pl->d[i], pl->g[i]);
                Launch L;
                if (resolve_launch(&pl->d[i], &p, &L) != S3R_OK || L.ksplit != 1) continue;
            }
        }
        pl->fuse_head[i] = 1;
    }
    int64_t off = 0;
    if (pl->pad_input && pl->d[0].in_layout != S3R_LAYOUT_PLAIN)
        return fail(S3R_ERR_INVALID, "a transformed chain input must come with its halo (in_halo = 1)");
    if (pl->pad_input) {
        if (user_in_halo != 0) return fail(S3R_ERR_INVALID, "chain input halo %d is smaller than the %d its first layer needs",
                                           user_in_halo, need0);
        pl->pad_off = 0;
        off = align_up(pl->g[0].x_store, 256);
    }
    for (int i = 0; i + 1 < n; ++i) {
        pl->off[i] = off;
        if (i == 0 && pl->stem_wino) off = align_up(off + pl->g[1].x_store, 256);      // layer 1's transformed planes
        else if (!pl->fuse_head[i]) off = align_up(off + pl->g[i].y_store, 256);   // a fused conv's output does not exist
    }
    for (int i = 0; i < n; ++i) {
        const int64_t sc = s3r_conv_scratch_elems(&pl->d[i]);
        if (sc < 0) return (int)sc;
        if (sc > pl->scratch_elems) pl->scratch_elems = sc;
    }
    pl->scratch_off = off;
    off = align_up(off + pl->scratch_elems, 256);
    pl->total = off;
    return S3R_OK;
}

}  // namespace s3rh

using namespace s3rh;

extern "C" {

int s3r_conv_out_size(const s3r_conv_desc* d) {
    if (!d) return fail(S3R_ERR_INVALID, "null descriptor");
    if (d->op == S3R_OP_LINEAR) return 1;
    if (d->op != S3R_OP_CONV && d->op != S3R_OP_DECONV) return fail(S3R_ERR_INVALID, "unknown op %d", d->op);
    if (d->in_size <= 0 || d->k <= 0 || d->stride <= 0 || d->pad < 0)
        return fail(S3R_ERR_INVALID, "bad size/k/stride/pad (in_size=%d k=%d stride=%d pad=%d)", d->in_size, d->k, d->stride, d->pad);
    const int n = out_size(d);
    if (n <= 0) return fail(S3R_ERR_INVALID, "empty output");
    return n;
}

int s3r_conv_packed_elems(const s3r_conv_desc* d, int64_t* elems) {
    Geo g; Route r;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if ((rc = route(d, &r))) return rc;
    if (!elems) return fail(S3R_ERR_INVALID, "null output");
    switch (r) {
        case R_STEM: *elems = 27 * 32; break;
        case R_HEAD: *elems = d->cin; break;
        case R_LINEAR: *elems = g.w_elems; break;
        case R_MFMA: {
            if (tclass_layer(d)) { *elems = tclass_w_elems(d); break; }
            if (im2col_layer(d)) { *elems = (int64_t)staged_geo(d).cin_pad * cout_pad(d->cout); break; }
            if (staged_layer(d)) { *elems = ipow(d->k, g.nd) * staged_geo(d).cin_pad * cout_pad(d->cout); break; }
            const int64_t taps = d->op == S3R_OP_DECONV ? 64 : ipow(d->k, g.nd);
            if (d->dtype == S3R_BF16) *elems = (taps * d->cin * cout_pad_h(d->cout) + 1) / 2;   // bf16, in float units
            else *elems = taps * d->cin * cout_pad(d->cout) + (wino_layer(d) ? wino_w_elems(d) : 0) + (dwino_layer(d) ? dwino_w_elems(d) : 0) +
                          (wino2_ax(d) >= 0 ? wino2_geo(d).w_elems : 0) + (dwino3_layer(d) ? s3r::dwino3_w_elems(d->cin, d->cout) : 0);
            break;
        }
    }
    return S3R_OK;
}

int64_t s3r_conv_scratch_elems(const s3r_conv_desc* d) {
    Geo g; Route r;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if ((rc = route(d, &r))) return rc;
    if (r == R_LINEAR) return s3r::linear_scratch_elems(d->batch, d->cin, d->cout);
    if (r != R_MFMA) return 0;
    if (d->dtype == S3R_BF16) {
        s3r::ConvParamsH ph = make_params_h(d, g);
        LaunchH Lh;
        if ((rc = resolve_launch_h(d, &ph, &Lh))) return rc;
        return s3r::conv_bf16_scratch_elems(ph, Lh.tm);
    }
    int alg, form;
    if ((rc = resolve_algo(d, &alg, &form))) return rc;
    if (tclass_direct(d)) return 0;                                                // (reads the producer's halo: nothing staged)
    if (staged_layer(d)) return (staged_geo(d).elems + 255) / 256 * 256;          // the staged copy (no split-K behind it)
    if (alg == ALG_WINO3) return dwino3_need(d, form).total;      // Dh, Dd, Ddh (+ the class-parallel form's slabs)
    if (alg == ALG_WINO) return wino_need(d, form, false).total;
    if (alg == ALG_WINO2) return wino2_need(d, form).total;
    s3r::ConvParams p = make_params(d, g);
    Launch L;
    if ((rc = resolve_launch(d, &p, &L))) return rc;
    return s3r::conv_scratch_elems(p, L.cfg);
}

}  // extern "C"

extern "C" {

int64_t s3r_chain_workspace_elems(const s3r_layer* layers, int n_layers) {
    Plan pl;
    int rc = plan_chain(layers, n_layers, &pl);
    if (rc) return rc;
    return pl.total;
}

// which transformed layout (if any) the producer of layer `d`'s input may write instead of the plain halo-padded tensor
static int wino_input_layout(const s3r_conv_desc* d, int64_t* elems) {
    *elems = 0;
    if (!d) return S3R_LAYOUT_PLAIN;
    s3r_conv_desc t = *d;
    t.in_layout = S3R_LAYOUT_PLAIN;
    int alg, form;
    if (resolve_algo(&t, &alg, &form) != S3R_OK || t.op != S3R_OP_CONV) return S3R_LAYOUT_PLAIN;
    if (alg == ALG_WINO && t.in_size % wino_r(&t) == 0 && wino_bmax(&t) >= t.batch) {
        *elems = wino_v_elems(&t);
        return S3R_LAYOUT_WINO_H;
    }
    if (alg == ALG_WINO2 && wino2_ax(&t) == 0 && t.in_size % 4 == 0) {
        const Wino2Geo g2 = wino2_geo(&t);
        if (g2.bmax >= t.batch) {
            *elems = g2.v_sample * t.batch;
            return S3R_LAYOUT_WINO_DH;
        }
    }
    if (alg == ALG_WINO2 && wino2_ax(&t) == 2 && t.in_size % 4 == 0) {
        const Wino2Geo g2 = wino2_geo(&t);
        if (g2.bmax >= t.batch) {
            *elems = g2.ncls * t.cin * s3r::wino2_npad(g2.pos_sample * t.batch);
            return S3R_LAYOUT_WINO_HW;
        }
    }
    return S3R_LAYOUT_PLAIN;
}

int64_t s3r_conv_wino_input_elems(const s3r_conv_desc* d) {
    if (!d) return fail(S3R_ERR_INVALID, "null descriptor");
    int64_t elems;
    (void)wino_input_layout(d, &elems);
    return elems;
}

int s3r_conv_wino_input_layout(const s3r_conv_desc* d) {
    if (!d) return fail(S3R_ERR_INVALID, "null descriptor");
    int64_t elems;
    return wino_input_layout(d, &elems);
}

}  // extern "C"
