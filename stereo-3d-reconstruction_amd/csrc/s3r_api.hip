// C-ABI of libs3r_hip.so (declared in include/s3r.h): the entry points that enqueue kernels — single layers, chains / stages,
// cost volume, linear, Chamfer, metrics — and the runners behind them (Winograd transform + class kernel + finish sequences, fused
// heads).  Planning (geometry, kernel policy, sizes, the arena) lives in s3r_plan.hip, the profiler in s3r_prof.hip.  Host code only.
#include "s3r_host.h"

#include <cstdlib>
#include <cstring>

using namespace s3rh;

namespace {

// input transform + class kernel + finish, in sub-batches that keep the transformed input inside 32-bit byte offsets
int wino2_run(const s3r_conv_desc* d, const Geo& g, s3r::ConvParams p, const float* x, const float* packed_w, float* y, float* scratch,
              int64_t scratch_elems, int form, hipStream_t s, int* launches) {
    const Wino2Geo g2 = wino2_geo(d);
    const WinoNeed need = wino2_need(d, form);
    const bool pre = d->in_layout == S3R_LAYOUT_WINO_DH || d->in_layout == S3R_LAYOUT_WINO_HW;      // the producer wrote the plane sets
    const bool to_v = d->out_layout == S3R_LAYOUT_WINO_HW;           // ... and this layer writes its consumer's
    if (g2.bmax <= 0 || ((pre || to_v) && g2.bmax < d->batch))
        return fail(S3R_ERR_INVALID, "two-axis Winograd form: the transformed input of this batch exceeds 2 GiB (at most %d samples per "
                    "call with a producer-written input: s3r_conv_wino_input_elems)", g2.bmax);
    if (!scratch || scratch_elems < need.total)
        return fail(S3R_ERR_WORKSPACE, "the two-axis Winograd form of this layer needs %lld floats of scratch (s3r_conv_scratch_elems), "
                    "got %lld", (long long)need.total, (long long)(scratch ? scratch_elems : 0));
    const int64_t direct_w = ipow(d->k, g.nd) * d->cin * cout_pad(d->cout);
    p.w = packed_w + direct_w + (wino_layer(d) ? wino_w_elems(d) : 0);      // behind the direct (and the one-axis) slabs
    p.x = pre ? x : scratch;
    p.part = scratch + need.v;
    p.Nd = g2.sg; p.Nh = g2.sg; p.Nw = g2.out;
    p.kd = 1; p.kh = 1; p.kw = g2.kw; p.T = g2.kw;
    p.x_hs = g2.wp; p.x_ds = g2.sg * g2.wp; p.x_cs = g2.sg * g2.sg * g2.wp;
    p.x_org = 0;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    if (g2.ax == 2) {                                     // (the finish kernel's view; launch_conv_wino2 derives the class GEMM's)
        p.Nd = 1; p.Nw = g2.sg;
        p.dS = s3r::FastDiv((unsigned)(g2.sg * g2.sg));
        p.dW = s3r::FastDiv((unsigned)g2.sg);
    }
    p.Hout = g2.out; p.Dout = g2.out;
    p.ncls = g2.ncls;
    p.ksplit = 1;
    const int64_t x_sample = g.x_elems / d->batch;
    *launches = 0;
    for (int b0 = 0; b0 < d->batch; b0 += g2.bmax) {
        const int nb = d->batch - b0 < g2.bmax ? d->batch - b0 : g2.bmax;
        hipError_t e = hipSuccess;
        if (!pre) {
            // (aux pass: reads the padded input once, writes its ncls plane sets)
            s3r::AuxScope aux(s, 4.0 * ((double)nb * x_sample + (double)(g2.ax == 2 ? g2.ncls * d->cin * s3r::wino2_npad(g2.pos_sample * nb) : g2.v_sample * nb)));
            if (g2.ax == 2)
                e = s3r::launch_wino2p_input(x + (int64_t)b0 * x_sample, scratch, nb, d->cin, g2.wp, g2.wp, g2.sg, g2.sg,
                                             s3r::wino2_npad(g2.pos_sample * nb), s);
            else
                e = s3r::launch_wino2_input(x + (int64_t)b0 * x_sample, scratch, g2.ax, (long long)nb * d->cin, g2.wp, g2.wp, g2.wp, g2.sg, g2.sg, s);
            if (e != hipSuccess) return hip_fail(e, "two-axis Winograd input transform launch");
            *launches += 1;
        }
        p.B = nb;
        p.x_cls = nb * d->cin * p.x_cs;
        p.x_bytes = (unsigned)(4 * (int64_t)nb * g2.v_sample);
        p.Ntotal = nb * p.Nd * p.Nh * p.Nw;
        p.y = y + (int64_t)b0 * p.y_bs;
        p.y_bytes = (unsigned)(4 * (int64_t)nb * p.y_bs);
        int nl = 0;
        e = s3r::launch_conv_wino2(p, g2.ax, wino2_form_of(d, p.Ntotal, form), to_v, s, &nl);
        if (e != hipSuccess) return hip_fail(e, "two-axis Winograd conv launch");
        *launches += nl;
    }
    return S3R_OK;
}

// fills the transposed-convolution parameters for the Winograd kernel and runs transform + kernel
int dwino_run(const s3r_conv_desc* d, s3r::ConvParams p, const float* x, const float* packed_w, float* scratch, int64_t scratch_elems,
              int form, hipStream_t s, int* launches, int* ran) {
    const WinoNeed need = wino_need(d, form, p.head_w != nullptr);
    if (!scratch || scratch_elems < need.total)
        return fail(S3R_ERR_WORKSPACE, "the Winograd form of this transposed convolution needs %lld floats of scratch "
                    "(s3r_conv_scratch_elems), got %lld", (long long)need.total, (long long)(scratch ? scratch_elems : 0));
    const int n = d->in_size;
    p.xd_mode = dwino_materialise(d) ? 1 : 0;
    hipError_t e;
    {
        const double x_el = (double)d->batch * d->cin * (double)ipow(n + 2, 3);
        s3r::AuxScope aux(s, 4.0 * x_el * (p.xd_mode ? 4.0 : 2.0));      // (reads x, writes Dh [, Dd, Ddh])
        e = s3r::launch_wino_diff(x, scratch, (long long)d->batch * d->cin, n + 2, n + 2, n + 2, p.xd_mode, s);
    }
    if (e != hipSuccess) return hip_fail(e, "Winograd difference-tensor launch");
    p.x = x;
    p.xd = scratch;
    p.part = scratch + need.v;
    p.w = packed_w + 64 * (int64_t)d->cin * cout_pad(d->cout);          // behind the direct slab (8 classes x 8 taps)
    p.Nd = n / 2; p.Nh = n / 2;
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.n_begin = 0; p.n_end = p.Ntotal;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    const s3r::WinoLaunch L = s3r::wino_plan(2, d->cout, wino_kcls(d), p.Ntotal, p.head_w != nullptr, form);
    int nl = 0;
    e = s3r::launch_deconv_wino(p, L, s, &nl);
    if (e != hipSuccess) return hip_fail(e, "Winograd transposed-conv launch");
    *launches = 1 + nl;
    *ran = 1 + L.mode;
    return S3R_OK;
}
// the three-axis form: difference tensors (all three materialised) + the class kernel (serial, or class-parallel over the depth
// class with a finish kernel: same bits)
int dwino3_run(const s3r_conv_desc* d, s3r::ConvParams p, const float* x, const float* packed_w, float* scratch, int64_t scratch_elems,
               hipStream_t s, int form, int* launches, int* ran) {
    const int n = d->in_size;
    const Dwino3Need nd = dwino3_need(d, form);
    const int64_t need = nd.total;
    if (!scratch || scratch_elems < need)
        return fail(S3R_ERR_WORKSPACE, "the three-axis Winograd form of this transposed convolution needs %lld floats of scratch "
                    "(s3r_conv_scratch_elems), got %lld", (long long)need, (long long)(scratch ? scratch_elems : 0));
    hipError_t e;
    {
        s3r::AuxScope aux(s, 4.0 * 4.0 * (double)d->batch * d->cin * (double)ipow(n + 2, 3));
        e = s3r::launch_wino_diff(x, scratch, (long long)d->batch * d->cin, n + 2, n + 2, n + 2, 1, s);
    }
    if (e != hipSuccess) return hip_fail(e, "Winograd difference-tensor launch");
    p.x = x;
    p.xd = scratch;
    p.w = packed_w + dwino3_w_offset(d);
    p.Nd = p.Nh = p.Nw = n / 2;
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    p.part = nd.split ? scratch + nd.diff : nullptr;
    e = s3r::launch_deconv_wino3(p, nd.split, s);
    if (e != hipSuccess) return hip_fail(e, "three-axis Winograd transposed-conv launch");
    *launches = nd.split ? 3 : 2;
    *ran = nd.split ? 6 : 5;
    return S3R_OK;
}

// MFMA conv with the following 1x1 single-channel head folded into its epilogue (plan_chain decides)
int conv_head_fused(const s3r_conv_desc* d, const Geo& g, const s3r_conv_desc* hd, const Geo& hg, const float* x,
                    const s3r_layer& L, const s3r_layer& H, float* out, float* scratch, int64_t scratch_elems, hipStream_t s) {
    s3r::ConvParams p = make_params(d, g);
    p.x = x; p.w = static_cast<const float*>(L.packed_w); p.scale = L.scale; p.shift = L.shift; p.y = out;
    // y_* now describe the head's (B, 1, n[, n], n) output
    const bool is3 = hd->ndim == 3;
    p.y_hs = hg.out_p; p.y_ds = is3 ? hg.out_p * hg.out_p : 0; p.y_cs = (int)ipow(hg.out_p, hg.nd);
    p.y_bs = p.y_cs;
    p.y_org = hd->out_halo * (p.y_ds + p.y_hs + 1);
    p.y_bytes = (unsigned)(hg.y_elems * 4);
    p.head_w = static_cast<const float*>(H.packed_w); p.head_scale = H.scale; p.head_shift = H.shift; p.head_act = hd->act;
    int alg, form;
    int rc = resolve_algo(d, &alg, &form);
    if (rc) return rc;
    if (alg == ALG_WINO3) {
        if (d->cout > 64) return fail(S3R_ERR_INVALID, "the fused head rides on the transposed Winograd kernel with <= 64 couts only");
        ProfScope ps(s, F_MFMA, d->tag, g.flops + hg.flops, g.bytes - 4.0 * d->batch * d->cout * (double)g.out_sp +
                     4.0 * d->batch * (double)hg.out_sp);
        ps.exec = g.flops * (27.0 / 64.0) + hg.flops;
        ps.algo = 5;
        return dwino3_run(d, p, x, static_cast<const float*>(L.packed_w), scratch, scratch_elems, s, form, &ps.launches, &ps.algo);
    }
    if (alg != ALG_DIRECT) {
        if (alg != ALG_WINO || d->op != S3R_OP_DECONV || d->cout > 64) return fail(S3R_ERR_INVALID, "the fused head rides on the transposed Winograd kernel with <= 64 couts only");
        ProfScope ps(s, F_MFMA, d->tag, g.flops + hg.flops, g.bytes - 4.0 * d->batch * d->cout * (double)g.out_sp +
                     4.0 * d->batch * (double)hg.out_sp);
        ps.exec = wino_exec_flops(d, g) + hg.flops;
        return dwino_run(d, p, x, static_cast<const float*>(L.packed_w), scratch, scratch_elems, form, s, &ps.launches, &ps.algo);
    }
    Launch Ln;
    rc = resolve_launch(d, &p, &Ln);
    if (rc) return rc;
    if (!(Ln.cfg == 1 || Ln.cfg == 2 || Ln.cfg == 7) || (Ln.cfg == 2 && d->cout > 32)) {
        // the heuristic's tile splits the couts over waves: take the widest one-wave-tall tile that fills the chip
        int bm, bn;
        s3r::conv_tile_dims(1, &bm, &bn);
        const long wg1 = (long)((p.Ntotal + bn - 1) / bn) * (p.transposed ? 8 : 1);
        Ln.cfg = d->cout <= 32 ? 2 : (wg1 >= 2000 ? 1 : 7);
    }
    ProfScope ps(s, F_MFMA, d->tag, g.flops + hg.flops, g.bytes - 4.0 * d->batch * d->cout * (double)g.out_sp +
                 4.0 * d->batch * (double)hg.out_sp);
    hipError_t e = s3r::launch_conv_mfma(p, Ln.cfg + 16 * Ln.vec, s);
    ps.launches = s3r::conv_last_launch_count();
    if (e != hipSuccess) return hip_fail(e, "fused conv+head launch");
    return S3R_OK;
}

// the same on the bf16 channels-last path: the conv's 64-cout tile is the whole channel axis of its positions
int conv_head_fused_h(const s3r_conv_desc* d, const Geo& g, const s3r_conv_desc* hd, const Geo& hg, const void* x,
                      const s3r_layer& L, const s3r_layer& H, float* out, hipStream_t s) {
    s3r::ConvParamsH p = make_params_h(d, g);
    p.x = x; p.w = L.packed_w; p.scale = L.scale; p.shift = L.shift; p.y = out;
    const bool is3 = hd->ndim == 3;
    p.y_ws = 1; p.y_hs = hg.out_p; p.y_ds = is3 ? hg.out_p * hg.out_p : 0; p.y_bs = (int)ipow(hg.out_p, hg.nd);
    p.y_org = hd->out_halo * (p.y_ds + p.y_hs + 1);
    p.head_w = static_cast<const float*>(H.packed_w); p.head_scale = H.scale; p.head_shift = H.shift; p.head_act = hd->act;
    LaunchH Ln;
    int rc = resolve_launch_h(d, &p, &Ln);
    if (rc) return rc;
    ProfScope ps(s, F_MFMA, d->tag, g.flops + hg.flops, g.bytes - 2.0 * d->batch * d->cout * (double)g.out_sp +
                 4.0 * d->batch * (double)hg.out_sp);
    hipError_t e = s3r::launch_conv_bf16(p, Ln.tm, s);
    if (e != hipSuccess) return hip_fail(e, "fused conv+head launch (bf16)");
    return S3R_OK;
}

}  // namespace

extern "C" {

int s3r_abi_version(void) { return S3R_ABI_VERSION; }

const char* s3r_last_error(void) { return last_error(); }

int s3r_conv_pack_weights(const s3r_conv_desc* d, const float* w, void* packedv, void* stream) {
    float* packed = static_cast<float*>(packedv);
    Geo g; Route r;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if ((rc = route(d, &r))) return rc;
    if (!w || !packed) return fail(S3R_ERR_INVALID, "null weight pointer");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_PACK, d->tag, 0.0, 8.0 * g.w_elems);
    hipError_t e = hipSuccess;
    switch (r) {
        case R_STEM: e = s3r::launch_pack_stem(w, packed, s); break;
        case R_HEAD: e = hipMemcpyAsync(packed, w, sizeof(float) * d->cin, hipMemcpyDeviceToDevice, s); break;
        case R_LINEAR: e = hipMemcpyAsync(packed, w, sizeof(float) * g.w_elems, hipMemcpyDeviceToDevice, s); break;
        case R_MFMA:
            if (d->dtype == S3R_BF16)
                e = s3r::launch_pack_bf16(w, packed, d->cin, d->cout, cout_pad_h(d->cout),
                                          d->op == S3R_OP_DECONV ? 8 : (int)ipow(d->k, g.nd), d->op == S3R_OP_DECONV, s);
            else if (tshuf_layer(d)) {      // w[Cin][Cout][k^nd] IS [Cin][cout x taps]: pad the rows to 16 and the columns to the tile
                const int rows = d->cout * (int)ipow(d->stride, g.nd);
                e = s3r::launch_pack_general(w, packed, d->cin, staged_geo(d).cin_pad, rows, cout_pad(rows), 1, 1, s);
            } else if (tclass_layer(d)) {
                for (int cls = 0; cls < (int)ipow(d->stride, g.nd) && e == hipSuccess; ++cls) {
                    const int rw = cls % d->stride, rh = (cls / d->stride) % d->stride, rd = g.nd == 3 ? cls / (d->stride * d->stride) : 0;
                    s3r::ConvParams q; int64_t w_off; double macs;
                    (void)make_params_tclass(d, g, rd, rh, rw, &q, &w_off, &macs);      // (the slab is packed whether or not this size uses it)
                    e = s3r::launch_pack_tclass(w, packed + w_off, d->cin, staged_geo(d).cin_pad, d->cout, cout_pad(d->cout), g.nd, d->k,
                                                d->stride, rd, rh, rw, s);
                }
            } else if (im2col_layer(d)) {      // w[Cout][Cin][k^nd] IS [Cout][cin x taps]: the K rows of the unfolded GEMM
                e = s3r::launch_pack_general(w, packed, d->cin * (int)ipow(d->k, g.nd), staged_geo(d).cin_pad, d->cout, cout_pad(d->cout), 1, 0, s);
            } else if (staged_layer(d)) {
                e = s3r::launch_pack_general(w, packed, d->cin, staged_geo(d).cin_pad, d->cout, cout_pad(d->cout), (int)ipow(d->k, g.nd),
                                             d->op == S3R_OP_DECONV, s);
            } else {
                e = s3r::launch_pack_conv(w, packed, d->cin, d->cout, cout_pad(d->cout),
                                          d->op == S3R_OP_DECONV ? 8 : (int)ipow(d->k, g.nd), d->op == S3R_OP_DECONV, s);
                if (e == hipSuccess && wino_layer(d))
                    e = s3r::launch_pack_wino(w, packed + ipow(d->k, g.nd) * d->cin * cout_pad(d->cout), d->cin, d->cout,
                                              cout_pad(d->cout), g.nd == 3 ? 3 : 1, 3, wino_r(d), s);
                if (e == hipSuccess && dwino_layer(d))
                    e = s3r::launch_pack_wino_deconv(w, packed + 64 * (int64_t)d->cin * cout_pad(d->cout), d->cin, d->cout, cout_pad(d->cout), s);
                if (e == hipSuccess && dwino3_layer(d))
                    e = s3r::launch_pack_dwino3(w, packed + dwino3_w_offset(d), d->cin, d->cout, s);
                if (e == hipSuccess && wino2_ax(d) >= 0)
                    e = s3r::launch_pack_wino2(w, packed + ipow(d->k, g.nd) * d->cin * cout_pad(d->cout) + (wino_layer(d) ? wino_w_elems(d) : 0),
                                               wino2_ax(d), d->cin, d->cout, cout_pad(d->cout), s);
            }
            break;
    }
    if (e != hipSuccess) return hip_fail(e, "pack weights");
    return S3R_OK;
}

}  // extern "C"

namespace {
// s3r_conv_forward with an optional SECOND input tensor for the stem: images [0, nsplit) are read from xv, images
// [nsplit, batch) from x2v (left / right renders of a stereo batch: no concatenation copy)
// x_u8 != 0 (stem only): xv / x2v are 8-bit renders, scaled by 1/255 inside the stem kernel
int conv_forward_impl(const s3r_conv_desc* d, const void* xv, const void* x2v, int nsplit, int x_u8, const void* packed_wv,
                      const float* scale, const float* shift, void* yv, float* scratch, int64_t scratch_elems, void* stream);
}

extern "C" {

int s3r_conv_forward(const s3r_conv_desc* d, const void* xv, const void* packed_wv, const float* scale,
                     const float* shift, void* yv, float* scratch, int64_t scratch_elems, void* stream) {
    return conv_forward_impl(d, xv, nullptr, 0, 0, packed_wv, scale, shift, yv, scratch, scratch_elems, stream);
}

}  // extern "C"

namespace {

int conv_forward_impl(const s3r_conv_desc* d, const void* xv, const void* x2v, int nsplit, int x_u8, const void* packed_wv,
                      const float* scale, const float* shift, void* yv, float* scratch, int64_t scratch_elems, void* stream) {
    const float* x = static_cast<const float*>(xv);
    const float* x2 = static_cast<const float*>(x2v);
    const float* packed_w = static_cast<const float*>(packed_wv);
    float* y = static_cast<float*>(yv);
    Geo g; Route r;
    int rc = geometry(d, &g);
    if (rc) return rc;
    if ((rc = route(d, &r))) return rc;
    if ((rc = check_halos(d, r))) return rc;
    if (!x || !packed_w || !y) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (x2 && r != R_STEM) return fail(S3R_ERR_INVALID, "only the stem convolution takes a second input tensor");
    if (x_u8 && r != R_STEM) return fail(S3R_ERR_INVALID, "only the stem convolution reads 8-bit renders");
    const double u8_saved = x_u8 ? 3.0 * d->batch * d->cin * (double)g.in_sp : 0.0;      // algorithmic bytes: 1 instead of 4 per sample
    if (x2 && (nsplit <= 0 || nsplit >= d->batch)) return fail(S3R_ERR_INVALID, "split %d outside (0, batch=%d)", nsplit, d->batch);
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = hipSuccess;
    if (d->dtype == S3R_BF16) {
        if (g.x_elems * 2 >= ((int64_t)1 << 31) || g.y_elems * 2 >= ((int64_t)1 << 31))
            return fail(S3R_ERR_INVALID, "bf16 path: tensors must be < 2 GiB per call: split the batch");
        switch (r) {
            case R_STEM: {
                if (!scale || !shift) return fail(S3R_ERR_INVALID, "stem needs scale and shift");
                ProfScope ps(s, F_STEM, d->tag, g.flops, g.bytes - u8_saved);
                e = s3r::launch_stem_bf16(xv, x2v, x_u8, nsplit, packed_w, scale, shift, yv, d->batch, g.in, g.in, g.out, g.out,
                                          g.out_p * g.out_p * 32, g.out_p * 32, d->out_halo * (g.out_p + 1) * 32, s);
                break;
            }
            case R_HEAD: {
                ProfScope ps(s, F_HEAD, d->tag, g.flops, g.bytes);
                e = s3r::launch_head_bf16(xv, packed_w, scale, shift, y, d->cin, (int64_t)d->batch * g.in_sp, d->act, s);
                break;
            }
            case R_MFMA: {
                s3r::ConvParamsH p = make_params_h(d, g);
                p.x = xv; p.w = packed_wv; p.scale = scale; p.shift = shift; p.y = yv;
                LaunchH L;
                if ((rc = resolve_launch_h(d, &p, &L))) return rc;
                if (L.ksplit > 1) {
                    const int64_t need = s3r::conv_bf16_scratch_elems(p, L.tm);
                    if (scratch && scratch_elems >= need) p.part = scratch;
                    else
                        return fail(S3R_ERR_WORKSPACE, "ksplit=%d needs %lld floats of scratch (s3r_conv_scratch_elems), got %lld", L.ksplit,
                                    (long long)need, (long long)(scratch ? scratch_elems : 0));
                }
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                e = s3r::launch_conv_bf16(p, L.tm, s);
                ps.launches = 1;
                break;
            }
            default: return fail(S3R_ERR_INVALID, "layer not available on the bf16 path");
        }
        if (e != hipSuccess) return hip_fail(e, "conv forward launch (bf16)");
        return S3R_OK;
    }
    switch (r) {
        case R_STEM: {
            if (!scale || !shift) return fail(S3R_ERR_INVALID, "stem needs scale and shift");
            ProfScope ps(s, F_STEM, d->tag, g.flops, g.bytes - u8_saved);
            e = s3r::launch_stem(xv, x2v, x_u8, nsplit, packed_w, scale, shift, y, d->batch, g.in, g.in, g.out, g.out, g.out_p * g.out_p,
                                 g.out_p, d->out_halo * (g.out_p + 1), s);
            break;
        }
        case R_HEAD: {
            ProfScope ps(s, F_HEAD, d->tag, g.flops, g.bytes);
            e = s3r::launch_head(x, packed_w, scale, shift, y, d->batch, d->cin, g.in_sp, d->act, s);
            break;
        }
        case R_LINEAR: {
            const int64_t need = s3r::linear_scratch_elems(d->batch, d->cin, d->cout);
            if (!scratch || scratch_elems < need)
                return fail(S3R_ERR_WORKSPACE, "linear layer needs %lld floats of scratch (s3r_conv_scratch_elems), got %lld",
                            (long long)need, (long long)(scratch ? scratch_elems : 0));
            ProfScope ps(s, F_LINEAR, d->tag, g.flops, g.bytes);
            // (the linear epilogue knows none / ReLU / sigmoid; LeakyReLU / ELU / Tanh are a pass behind it, as on the convolutions)
            e = s3r::launch_linear(x, packed_w, scale, shift, y, d->batch, d->cin, d->cout, d->act > S3R_ACT_SIGMOID ? S3R_ACT_NONE : d->act,
                                   scratch, s);
            if (e == hipSuccess && d->act > S3R_ACT_SIGMOID) e = s3r::launch_act(y, g.y_elems, d->act, d->act_param, s);
            break;
        }
        case R_MFMA: {
            int alg, form;
            if ((rc = resolve_algo(d, &alg, &form))) return rc;
            if (staged_layer(d)) {
                // ---- parameter-general layer behind a staged copy: staged copy -> direct kernel [-> activation pass]
                s3r::ConvParams q = make_params_staged(d, g);
                q.x = x; q.w = packed_w; q.scale = scale; q.shift = shift; q.y = y;
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.launches = 0;
                const bool in_place = tclass_direct(d);       // (a residue-class layer whose input already carries its halo)
                if (!in_place) {
                    const StagedGeo sg = staged_geo(d);
                    const int64_t need = (sg.elems + 255) / 256 * 256;
                    if (!scratch || scratch_elems < need)
                        return fail(S3R_ERR_WORKSPACE, "a parameter-general layer stages its input in %lld floats of scratch "
                                    "(s3r_conv_scratch_elems), got %lld", (long long)need, (long long)(scratch ? scratch_elems : 0));
                    if (im2col_layer(d)) {
                        // ---- cin <= 8: unfold (sub-batches of sg.bmax samples) -> a 1 x 1 GEMM per pass [-> activation pass]
                        Launch Li;
                        s3r_conv_desc di = *d;
                        di.ksplit = 1;
                        const int64_t x_sample = g.x_elems / d->batch, s_sample = sg.elems / (d->batch < sg.bmax ? d->batch : sg.bmax);
                        for (int b0 = 0; b0 < d->batch; b0 += sg.bmax) {
                            const int nb = d->batch - b0 < sg.bmax ? d->batch - b0 : sg.bmax;
                            {
                                s3r::AuxScope aux(s, 4.0 * ((double)nb * x_sample + (double)nb * s_sample));
                                e = s3r::launch_stage_im2col(x + (int64_t)b0 * x_sample, scratch, nb, d->cin, sg.cin_pad, g.nd, d->in_size,
                                                             d->in_halo, sg.sp, d->k, d->stride, d->pad, dil_of(d), s);
                            }
                            if (e != hipSuccess) return hip_fail(e, "unfolding launch");
                            s3r::ConvParams c = q;
                            c.x = scratch;
                            c.B = nb; c.Ntotal = nb * c.Nd * c.Nh * c.Nw;
                            c.x_bytes = (unsigned)(4 * nb * s_sample);
                            c.y = y + (int64_t)b0 * c.y_bs; c.y_bytes = (unsigned)(4 * (int64_t)nb * c.y_bs);
                            if ((rc = resolve_launch(&di, &c, &Li))) return rc;
                            e = s3r::launch_conv_mfma(c, Li.cfg + 16 * Li.vec, s);
                            if (e != hipSuccess) return hip_fail(e, "conv forward launch (unfolded)");
                            ps.launches += 1 + s3r::conv_last_launch_count();
                        }
                        ps.exec = 2.0 * d->batch * (double)d->cout * (double)g.out_sp * sg.cin_pad;
                        if (needs_act_pass(d)) {
                            s3r::AuxScope aux2(s, 8.0 * (double)g.y_elems);
                            e = s3r::launch_act(y, g.y_elems, d->act, d->act_param, s);
                            if (e != hipSuccess) return hip_fail(e, "activation launch");
                            ps.launches += 1;
                        }
                        return S3R_OK;
                    } else {
                        s3r::AuxScope aux(s, 4.0 * ((double)g.x_elems + (double)sg.elems));
                        e = s3r::launch_stage(x, scratch, d->batch, d->cin, sg.cin_pad, g.nd, d->in_size, d->in_halo, sg.sp, sg.pe, sg.step, s);
                        ps.launches += 2;
                        ps.exec = 2.0 * d->batch * (double)d->cout * (double)g.out_sp * sg.cin_pad * (double)ipow(d->k, g.nd);
                    }
                    if (e != hipSuccess) return hip_fail(e, "staging launch");
                    q.x = scratch;
                }
                Launch L;
                s3r_conv_desc dd = *d;
                dd.ksplit = 1;                                   // (no split-K slabs behind the staged copy)
                if (tshuf_layer(d)) {
                    // ---- ConvTranspose with k == stride: ONE 1 x 1 GEMM over (cout, tap) rows, depth-to-space store
                    s3r::ConvParams c = make_params_tshuf(d, g);
                    c.x = in_place ? x : scratch; c.w = packed_w; c.scale = scale; c.shift = shift; c.y = y;
                    dd.tile = -1;
                    dd.cout = c.Cout;
                    if ((rc = resolve_launch(&dd, &c, &L))) return rc;
                    e = s3r::launch_conv_mfma(c, L.cfg + 16 * L.vec, s);
                    if (e != hipSuccess) return hip_fail(e, "conv forward launch (depth-to-space)");
                    ps.launches += s3r::conv_last_launch_count();
                    ps.exec = 2.0 * (double)c.Ntotal * c.Cin * (double)c.Cout;
                } else if (tclass_layer(d)) {
                    // ---- ConvTranspose, dilation 1: the residue classes of the output, each a stride-1 convolution — ONE launch over a
                    // class table (up to 27 classes), one launch per class beyond that
                    double macs_all = 0.0;
                    dd.tile = -1;
                    const int ncls = (int)ipow(d->stride, g.nd);
                    s3r::TClsTable tab;
                    tab.n = 0;
                    s3r::ConvParams base;
                    int64_t w_first = 0;
                    for (int cls = 0; cls < ncls; ++cls) {
                        const int rw = cls % d->stride, rh = (cls / d->stride) % d->stride, rd = g.nd == 3 ? cls / (d->stride * d->stride) : 0;
                        s3r::ConvParams c; int64_t w_off; double macs;
                        if (!make_params_tclass(d, g, rd, rh, rw, &c, &w_off, &macs)) continue;
                        c.x = in_place ? x : scratch; c.w = packed_w + w_off; c.scale = scale; c.shift = shift; c.y = y;
                        macs_all += macs;
                        if (ncls <= s3r::kTClsMax) {
                            if (tab.n == 0) { base = c; w_first = w_off; }
                            s3r::TClsEntry& t = tab.e[tab.n++];
                            t.Nd = c.Nd; t.Nh = c.Nh; t.Nw = c.Nw; t.kd = c.kd; t.kh = c.kh; t.kw = c.kw; t.T = c.T;
                            t.x_org = c.x_org; t.y_org = c.y_org; t.Ntotal = c.Ntotal; t.wg_begin = 0; t.w_off = (int)(w_off - w_first);
                            t.dS = c.dS; t.dHW = c.dHW; t.dW = c.dW;
                            continue;
                        }
                        if ((rc = resolve_launch(&dd, &c, &L))) return rc;
                        e = s3r::launch_conv_mfma(c, L.cfg + 16 * L.vec, s);
                        if (e != hipSuccess) return hip_fail(e, "conv forward launch (transposed class)");
                        ps.launches += s3r::conv_last_launch_count();
                    }
                    if (tab.n > 0) {
                        e = s3r::launch_conv_tcls(base, tab, s);
                        if (e != hipSuccess) return hip_fail(e, "conv forward launch (transposed classes)");
                        ps.launches += 1;
                    }
                    ps.exec = 2.0 * macs_all;
                } else {
                    if ((rc = resolve_launch(&dd, &q, &L))) return rc;
                    e = s3r::launch_conv_mfma(q, L.cfg + 16 * L.vec, s);
                    if (e != hipSuccess) return hip_fail(e, "conv forward launch");
                    ps.launches += s3r::conv_last_launch_count();
                }
                if (needs_act_pass(d)) {
                    s3r::AuxScope aux(s, 8.0 * (double)g.y_elems);
                    e = s3r::launch_act(y, g.y_elems, d->act, d->act_param, s);
                    if (e != hipSuccess) return hip_fail(e, "activation launch");
                    ps.launches += 1;
                }
                return S3R_OK;
            }
            s3r::ConvParams p = make_params(d, g);
            p.x = x; p.w = packed_w; p.scale = scale; p.shift = shift; p.y = y;
            const bool wino = alg == ALG_WINO;
            if (alg == ALG_WINO2) {
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.exec = wino2_exec_flops(d);
                ps.algo = 4;
                return wino2_run(d, g, p, x, packed_w, y, scratch, scratch_elems, form, s, &ps.launches);
            }
            if (alg == ALG_WINO3) {
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.exec = g.flops * (27.0 / 64.0);
                ps.algo = 5;
                return dwino3_run(d, p, x, packed_w, scratch, scratch_elems, s, form, &ps.launches, &ps.algo);
            }
            if (wino && d->op == S3R_OP_DECONV) {
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.exec = wino_exec_flops(d, g);
                return dwino_run(d, p, x, packed_w, scratch, scratch_elems, form, s, &ps.launches, &ps.algo);
            }
            if (wino) {
                // The transformed input must stay inside 32-bit byte offsets: larger batches go through in sub-batches (a sample's
                // result does not depend on the batch it is computed in, so neither does it on this split)
                const bool pre = d->in_layout == S3R_LAYOUT_WINO_H;      // the producer wrote the transformed planes
                const int R = wino_r(d), kind = wino_kind(d);
                const int is3 = d->ndim == 3, n = d->in_size, wp = n + 2, h2 = (n + R - 1) / R, dp = is3 ? n + 2 : 1;
                const int bmax = wino_bmax(d);
                if (bmax <= 0 || (pre && bmax < d->batch))
                    return fail(S3R_ERR_INVALID, "a Winograd-transformed input takes at most %d samples per call here "
                                "(s3r_conv_wino_input_elems)", bmax);
                const WinoNeed need = wino_need(d, form, false);
                if (need.total > 0 && (!scratch || scratch_elems < need.total))
                    return fail(S3R_ERR_WORKSPACE, "the Winograd form of this layer needs %lld floats of scratch (s3r_conv_scratch_elems), "
                                "got %lld", (long long)need.total, (long long)(scratch ? scratch_elems : 0));
                const int64_t v_sample = wino_v_elems(d) / d->batch, x_sample = g.x_elems / d->batch;
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.launches = 0;
                ps.exec = wino_exec_flops(d, g);
                p.w = packed_w + ipow(3, g.nd) * d->cin * cout_pad(d->cout);        // the class slabs sit behind the direct slab
                p.x = pre ? x : scratch;
                p.part = scratch ? scratch + need.v : nullptr;
                p.Nh = h2; p.T = p.kd * p.kw;
                p.x_hs = wp; p.x_ds = is3 ? h2 * wp : 0; p.x_cs = dp * h2 * wp;
                p.x_org = 0;
                p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
                p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
                p.dW = s3r::FastDiv((unsigned)p.Nw);
                p.Hout = n;
                p.ksplit = 1;
                for (int b0 = 0; b0 < d->batch; b0 += bmax) {
                    const int nb = d->batch - b0 < bmax ? d->batch - b0 : bmax;
                    if (!pre) {
                        s3r::AuxScope aux(s, 4.0 * ((double)nb * x_sample + (double)nb * v_sample));
                        e = s3r::launch_wino_input(x + (int64_t)b0 * x_sample, scratch, (long long)nb * d->cin * dp, n + 2, wp, h2, R, s);
                        if (e != hipSuccess) return hip_fail(e, "Winograd input transform launch");
                        ps.launches += 1;
                    }
                    p.B = nb;
                    p.x_cls = nb * d->cin * p.x_cs;
                    p.x_bytes = (unsigned)(4 * (int64_t)nb * v_sample);
                    p.Ntotal = nb * p.Nd * p.Nh * p.Nw;
                    p.y = y + (int64_t)b0 * p.y_bs;
                    p.y_bytes = (unsigned)(4 * (int64_t)nb * p.y_bs);
                    const s3r::WinoLaunch WL = s3r::wino_plan(kind, d->cout, wino_kcls(d), p.Ntotal, false, form);
                    int nl = 0;
                    e = s3r::launch_conv_wino(p, WL, s, &nl);
                    if (e != hipSuccess) return hip_fail(e, "Winograd conv launch");
                    ps.launches += nl;
                    ps.algo = 1 + WL.mode;
                }
                return S3R_OK;
            }
            Launch L;
            if ((rc = resolve_launch(d, &p, &L))) return rc;
            if (L.ksplit > 1) {
                const int64_t need = s3r::conv_scratch_elems(p, L.cfg);
                if (scratch && scratch_elems >= need) p.part = scratch;
                else      // (never answered unsplit: another summation order is other bits)
                    return fail(S3R_ERR_WORKSPACE, "ksplit=%d needs %lld floats of scratch (s3r_conv_scratch_elems), got %lld", L.ksplit,
                                (long long)need, (long long)(scratch ? scratch_elems : 0));
            }
            ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
            e = s3r::launch_conv_mfma(p, L.cfg + 16 * L.vec, s);
            ps.launches = s3r::conv_last_launch_count();
            if (e == hipSuccess && needs_act_pass(d)) {       // (make_params ran the kernel with ACT_NONE; split-K as the descriptor says)
                s3r::AuxScope aux(s, 8.0 * (double)g.y_elems);
                e = s3r::launch_act(y, g.y_elems, d->act, d->act_param, s);
                ps.launches += 1;
            }
            break;
        }
    }
    if (e != hipSuccess) return hip_fail(e, "conv forward launch");
    return S3R_OK;
}

int chain_forward_impl(const s3r_layer* layers, int n_layers, const void* x, const void* x2, int nsplit, int x_u8, void* y,
                       float* ws, int64_t ws_elems, int ws_fresh, void* stream);

}  // namespace

extern "C" {

int s3r_chain_forward(const s3r_layer* layers, int n_layers, const void* x, void* y, float* ws, int64_t ws_elems,
                      int ws_fresh, void* stream) {
    return chain_forward_impl(layers, n_layers, x, nullptr, 0, 0, y, ws, ws_elems, ws_fresh, stream);
}

}  // extern "C"

namespace {

int chain_forward_impl(const s3r_layer* layers, int n_layers, const void* x, const void* x2, int nsplit, int x_u8, void* y,
                       float* ws, int64_t ws_elems, int ws_fresh, void* stream) {
    Plan pl;
    int rc = plan_chain(layers, n_layers, &pl);
    if (rc) return rc;
    if (!x || !y) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (pl.total > 0 && (!ws || ws_elems < pl.total))
        return fail(S3R_ERR_WORKSPACE, "chain needs a workspace of %lld floats, got %lld", (long long)pl.total,
                    (long long)ws_elems);
    hipStream_t s = (hipStream_t)stream;
    if (ws_fresh && pl.total > 0) {   // zero halos (and everything else) once; later calls write interiors only
        hipError_t e = hipMemsetAsync(ws, 0, (size_t)pl.total * sizeof(float), s);
        if (e != hipSuccess) return hip_fail(e, "workspace memset");
    }
    const void* cur = x;
    if ((x2 || x_u8) && (pl.pad_input || pl.r[0] != R_STEM))
        return fail(S3R_ERR_INVALID, "a second input tensor / 8-bit renders need a chain that starts with the stem convolution");
    // the stems fetch render rows by 16-byte vector loads / LDS-DMA from base + row * width: fp32 tensors from any allocator are
    // aligned, an 8-bit view at an odd offset (or a C caller's offset pointer) is not.  Checked HERE so that every entry that can
    // reach a stem (s3r_encoder_forward[_u8] and s3r_chain_forward over a tower prefix) shares the check
    if (pl.r[0] == R_STEM && (((uintptr_t)x & 15) || ((uintptr_t)x2 & 15)))
        return fail(S3R_ERR_INVALID, "render tensors must be 16-byte aligned (got %p, %p)", x, x2);
    if (pl.pad_input) {
        const s3r_conv_desc& d0 = pl.d[0];
        const int hl = d0.in_halo, is3 = d0.ndim == 3;
        ProfScope ps(s, F_PAD, d0.tag, 0.0, 8.0 * d0.batch * d0.cin * (double)pl.g[0].in_sp);
        hipError_t e;
        if (d0.dtype == S3R_BF16) {
            // channels-last bf16 (B, [D,] H, W, C): rows of W*C bf16 = W*C/2 floats, planes = samples
            const int wc = d0.in_size * d0.cin / 2, hwc = hl * d0.cin / 2;
            e = s3r::launch_pad_copy(static_cast<const float*>(x), ws + pl.pad_off, d0.batch, is3 ? d0.in_size : 1,
                                     d0.in_size, wc, is3 ? hl : 0, hl, hwc, s);
        } else {
            e = s3r::launch_pad_copy(static_cast<const float*>(x), ws + pl.pad_off, (int64_t)d0.batch * d0.cin,
                                     is3 ? d0.in_size : 1, d0.in_size, d0.in_size, is3 ? hl : 0, hl, hl, s);
        }
        if (e != hipSuccess) return hip_fail(e, "pad copy launch");
        cur = ws + pl.pad_off;
    }
    for (int i = 0; i < n_layers; ++i) {
        const s3r_layer& L = layers[i];
        if (pl.fuse_head[i]) {
            const s3r_layer& H = layers[i + 1];
            void* out = (i + 1 == n_layers - 1) ? y : static_cast<void*>(ws + pl.off[i + 1]);
            if (pl.d[i].dtype == S3R_BF16)
                rc = conv_head_fused_h(&pl.d[i], pl.g[i], &pl.d[i + 1], pl.g[i + 1], cur, L, H, static_cast<float*>(out), s);
            else
                rc = conv_head_fused(&pl.d[i], pl.g[i], &pl.d[i + 1], pl.g[i + 1], static_cast<const float*>(cur), L, H,
                                     static_cast<float*>(out), pl.scratch_elems ? ws + pl.scratch_off : nullptr, pl.scratch_elems, s);
            if (rc) return rc;
            cur = out;
            ++i;
            continue;
        }
        void* out = (i == n_layers - 1) ? y : static_cast<void*>(ws + pl.off[i]);
        if (i == 0 && pl.stem_wino) {
            const s3r_conv_desc& d0 = pl.d[0];
            const Geo& g0 = pl.g[0];
            if (!L.packed_w || !L.scale || !L.shift) return fail(S3R_ERR_INVALID, "stem needs packed weights, scale and shift");
            if (x2 && (nsplit <= 0 || nsplit >= d0.batch)) return fail(S3R_ERR_INVALID, "split %d outside (0, batch=%d)", nsplit, d0.batch);
            // bytes: the renders in (1 byte a sample when 8-bit), layer 1's six plane sets out
            ProfScope ps(s, F_STEM, d0.tag, g0.flops, (x_u8 ? 1.0 : 4.0) * d0.batch * d0.cin * (double)g0.in_sp + 4.0 * (double)pl.g[1].x_elems);
            hipError_t e = s3r::launch_stem_wino(cur, x2, x_u8, nsplit, static_cast<const float*>(L.packed_w), L.scale, L.shift,
                                                 static_cast<float*>(out), d0.batch, g0.in, g0.in, g0.out, g0.out, s);
            if (e != hipSuccess) return hip_fail(e, "stem (Winograd layout) launch");
            cur = out;
            continue;
        }
        rc = conv_forward_impl(&pl.d[i], cur, i == 0 ? x2 : nullptr, nsplit, i == 0 ? x_u8 : 0, L.packed_w, L.scale, L.shift, out,
                               pl.scratch_elems ? ws + pl.scratch_off : nullptr, pl.scratch_elems, stream);
        if (rc) return rc;
        cur = out;
    }
    return S3R_OK;
}

}  // namespace

extern "C" {

static int encoder_forward(const s3r_layer* layers, int n_layers, const void* images_left, const void* images_right, int u8,
                           void* features, float* ws, int64_t ws_elems, int ws_fresh, void* stream) {
    if (!layers || n_layers <= 0) return fail(S3R_ERR_INVALID, "empty encoder");
    const s3r_conv_desc& f = layers[0].desc;
    if (f.op != S3R_OP_CONV || f.ndim != 2 || f.cin != 3)
        return fail(S3R_ERR_INVALID, "encoder must start with a 2D convolution over 3-channel renders");
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].desc.op != S3R_OP_CONV || layers[i].desc.ndim != 2)
            return fail(S3R_ERR_INVALID, "encoder layer %d is not a 2D convolution", i);
    if (images_right && (f.batch < 2 || (f.batch & 1)))
        return fail(S3R_ERR_INVALID, "a (left, right) pair of tensors needs an even image count N = 2B (got %d)", f.batch);
    return chain_forward_impl(layers, n_layers, images_left, images_right, f.batch / 2, u8, features, ws, ws_elems, ws_fresh,
                              stream);
}

int s3r_encoder_forward(const s3r_layer* layers, int n_layers, const float* images_left, const float* images_right,
                        void* features, float* ws, int64_t ws_elems, int ws_fresh, void* stream) {
    return encoder_forward(layers, n_layers, images_left, images_right, 0, features, ws, ws_elems, ws_fresh, stream);
}

int s3r_encoder_forward_u8(const s3r_layer* layers, int n_layers, const uint8_t* images_left, const uint8_t* images_right,
                           void* features, float* ws, int64_t ws_elems, int ws_fresh, void* stream) {
    return encoder_forward(layers, n_layers, images_left, images_right, 1, features, ws, ws_elems, ws_fresh, stream);
}

int s3r_channels_last_to_f32(const void* x, float* y, int batch, int channels, int64_t positions, void* stream) {
    if (!x || !y) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || positions <= 0) return fail(S3R_ERR_INVALID, "dims must be positive");
    if (batch > 65535 || (int64_t)channels * positions >= kMaxElems)
        return fail(S3R_ERR_INVALID, "tensor too large for one call: split the batch");
    hipStream_t s = (hipStream_t)stream;
    const double n = (double)batch * channels * (double)positions;
    ProfScope ps(s, F_PAD, 1, 0.0, 6.0 * n);
    hipError_t e = s3r::launch_cl_bf16_to_f32(x, y, batch, positions, channels, s);
    if (e != hipSuccess) return hip_fail(e, "channels-last to fp32 launch");
    return S3R_OK;
}

int s3r_decoder_forward(const s3r_layer* layers, int n_layers, const void* volume, float* occupancy, float* ws,
                        int64_t ws_elems, int ws_fresh, void* stream) {
    if (!layers || n_layers <= 0) return fail(S3R_ERR_INVALID, "empty decoder");
    for (int i = 0; i < n_layers; ++i)
        if (layers[i].desc.op == S3R_OP_LINEAR || layers[i].desc.ndim != 3)
            return fail(S3R_ERR_INVALID, "decoder layer %d is not a 3D (transposed) convolution", i);
    return s3r_chain_forward(layers, n_layers, volume, occupancy, ws, ws_elems, ws_fresh, stream);
}

int s3r_cost_volume_forward(const float* fl, const float* fr, float* vol, int batch, int channels, int max_disp,
                            int height, int width, int out_halo, void* stream) {
    if (!fl || !fr || !vol) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || max_disp <= 0 || height <= 0 || width <= 0)
        return fail(S3R_ERR_INVALID, "cost volume dims must be positive");
    const int64_t hw = (int64_t)height * width;
    if (2 * hw * 4 > 64 * 1024) return fail(S3R_ERR_INVALID, "feature plane %dx%d does not fit the LDS staging", height, width);
    if (out_halo < 0 || out_halo > 8) return fail(S3R_ERR_INVALID, "halo must be in [0, 8]");
    const int64_t out = (int64_t)batch * 2 * channels * max_disp * hw;
    const int64_t out_p = (int64_t)batch * 2 * channels * (max_disp + 2 * out_halo) * (height + 2 * out_halo) *
                          (width + 2 * out_halo);
    if (out_p >= kMaxElems) return fail(S3R_ERR_INVALID, "cost volume too large for one call: split the batch");
    hipStream_t s = (hipStream_t)stream;
    const double bytes = 4.0 * (2.0 * batch * channels * hw + (double)out);
    ProfScope ps(s, F_COSTVOL, 0, (double)out, bytes);
    hipError_t e = s3r::launch_cost_volume(fl, fr, vol, batch, channels, max_disp, height, width, out_halo, s);
    if (e != hipSuccess) return hip_fail(e, "cost volume launch");
    return S3R_OK;
}

int s3r_cost_volume_forward_wino(const float* fl, const float* fr, float* planes, int batch, int channels, int max_disp,
                                 int height, int width, void* stream) {
    if (!fl || !fr || !planes) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || max_disp <= 0 || height < 4 || width <= 0 || (height & 3))
        return fail(S3R_ERR_INVALID, "bad cost-volume shape for the Winograd layout (height a multiple of 4: F(4,3) groups)");
    if (max_disp > width) return fail(S3R_ERR_INVALID, "max_disp %d exceeds the feature width %d", max_disp, width);
    if ((size_t)2 * height * width * sizeof(float) > 64 * 1024)
        return fail(S3R_ERR_INVALID, "feature plane %dx%d does not fit the kernel's 64 KiB of LDS", height, width);
    const int64_t elems = 6 * (int64_t)batch * 2 * channels * (max_disp + 2) * (height / 4) * (width + 2);
    if (elems * 4 >= ((int64_t)1 << 31)) return fail(S3R_ERR_INVALID, "Winograd planes of %lld floats exceed 2 GiB: split the batch", (long long)elems);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_COSTVOL, 0, 2.0 * batch * channels * (double)max_disp * height * width,
                 4.0 * batch * channels * (2.0 * height * width) + 4.0 * (double)elems);
    hipError_t e = s3r::launch_cost_volume_wino(fl, fr, planes, batch, channels, max_disp, height, width, 4, s);
    if (e != hipSuccess) return hip_fail(e, "cost volume (Winograd layout) launch");
    return S3R_OK;
}

int s3r_cost_volume_forward_wino2(const float* fl, const float* fr, float* planes, int batch, int channels, int max_disp,
                                  int height, int width, void* stream) {
    if (!fl || !fr || !planes) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || max_disp < 4 || height < 4 || width <= 0 || (height & 3) || (max_disp & 3))
        return fail(S3R_ERR_INVALID, "bad cost-volume shape for the two-axis Winograd layout (height and max_disp multiples of 4: F(4,3) groups)");
    if (max_disp > width) return fail(S3R_ERR_INVALID, "max_disp %d exceeds the feature width %d", max_disp, width);
    if ((size_t)2 * height * width * sizeof(float) > 64 * 1024)
        return fail(S3R_ERR_INVALID, "feature plane %dx%d does not fit the kernel's 64 KiB of LDS", height, width);
    const int64_t elems = 36 * (int64_t)batch * 2 * channels * (max_disp / 4) * (height / 4) * (width + 2);
    if (elems * 4 >= ((int64_t)1 << 31)) return fail(S3R_ERR_INVALID, "two-axis Winograd planes of %lld floats exceed 2 GiB: split the batch", (long long)elems);
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_COSTVOL, 0, 2.0 * batch * channels * (double)max_disp * height * width,
                 4.0 * batch * channels * (2.0 * height * width) + 4.0 * (double)elems);
    hipError_t e = s3r::launch_cost_volume_wino2(fl, fr, planes, batch, channels, max_disp, height, width, s);
    if (e != hipSuccess) return hip_fail(e, "cost volume (two-axis Winograd layout) launch");
    return S3R_OK;
}

int s3r_cost_volume_forward_bf16(const void* fl, const void* fr, void* vol, int batch, int channels, int max_disp,
                                 int height, int width, int out_halo, void* stream) {
    if (!fl || !fr || !vol) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || max_disp <= 0 || height <= 0 || width <= 0)
        return fail(S3R_ERR_INVALID, "cost volume dims must be positive");
    if (channels % 8 != 0) return fail(S3R_ERR_INVALID, "bf16 cost volume needs channels %% 8 == 0");
    if (out_halo < 0 || out_halo > 8) return fail(S3R_ERR_INVALID, "halo must be in [0, 8]");
    const int64_t hw = (int64_t)height * width;
    const int64_t out = (int64_t)batch * 2 * channels * max_disp * hw;
    const int64_t out_p = (int64_t)batch * 2 * channels * (max_disp + 2 * out_halo) * (height + 2 * out_halo) *
                          (width + 2 * out_halo);
    if (out_p >= kMaxElems) return fail(S3R_ERR_INVALID, "cost volume too large for one call: split the batch");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_COSTVOL, 0, (double)out, 2.0 * (2.0 * batch * channels * hw + (double)out));
    hipError_t e = s3r::launch_cost_volume_bf16(fl, fr, vol, batch, channels, max_disp, height, width, out_halo, s);
    if (e != hipSuccess) return hip_fail(e, "cost volume launch (bf16)");
    return S3R_OK;
}

int64_t s3r_linear_scratch_elems(int batch, int cin, int cout) {
    if (batch <= 0 || cin <= 0 || cout <= 0) return fail(S3R_ERR_INVALID, "linear dims must be positive");
    return s3r::linear_scratch_elems(batch, cin, cout);
}

int s3r_linear_forward(const float* x, const float* w, const float* bias, float* y, int batch, int cin, int cout,
                       int act, float* scratch, int64_t scratch_elems, void* stream) {
    if (!x || !w || !y) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || cin <= 0 || cout <= 0) return fail(S3R_ERR_INVALID, "linear dims must be positive");
    if (act < S3R_ACT_NONE || act > S3R_ACT_SIGMOID)
        return fail(S3R_ERR_INVALID, "s3r_linear_forward takes none / relu / sigmoid (act %d): LeakyReLU / ELU / Tanh carry a parameter — "
                    "run the layer as an S3R_OP_LINEAR descriptor through s3r_conv_forward / s3r_chain_forward", act);
    if (!scratch || scratch_elems < s3r::linear_scratch_elems(batch, cin, cout))
        return fail(S3R_ERR_WORKSPACE, "linear needs %lld floats of scratch (s3r_linear_scratch_elems)",
                    (long long)s3r::linear_scratch_elems(batch, cin, cout));
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_LINEAR, 0, 2.0 * batch * (double)cin * cout, 4.0 * ((double)cin * cout + (double)batch * (cin + cout)));
    hipError_t e = s3r::launch_linear(x, w, nullptr, bias, y, batch, cin, cout, act, scratch, s);
    if (e != hipSuccess) return hip_fail(e, "linear launch");
    return S3R_OK;
}

int s3r_chamfer_forward(const float* p, const float* q, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                        int batch, int n, int m, void* stream) {
    if (!p || !q || !dist1 || !dist2 || !idx1 || !idx2) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || n <= 0 || m <= 0) return fail(S3R_ERR_INVALID, "chamfer needs non-empty clouds (batch=%d n=%d m=%d)", batch, n, m);
    if (batch > 65535) return fail(S3R_ERR_INVALID, "batch > 65535: split the call");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_CHAMFER, 0, 2.0 * 8.0 * batch * (double)n * m, 4.0 * batch * (5.0 * n + 5.0 * m));
    hipError_t e = s3r::launch_chamfer(p, q, dist1, dist2, idx1, idx2, batch, n, m, s);
    if (e != hipSuccess) return hip_fail(e, "chamfer launch");
    return S3R_OK;
}

int s3r_voxel_iou(const float* pred, const float* gt, float threshold, float* iou, int batch, int64_t voxels,
                  void* stream) {
    if (!pred || !gt || !iou) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || voxels <= 0) return fail(S3R_ERR_INVALID, "iou dims must be positive");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_IOU, 0, 0.0, 8.0 * batch * (double)voxels);
    hipError_t e = s3r::launch_iou(pred, gt, threshold, iou, batch, voxels, s);
    if (e != hipSuccess) return hip_fail(e, "iou launch");
    return S3R_OK;
}

int s3r_disparity_wta(const float* feat_l, const float* feat_r, float* disp_l, float* disp_r, int batch, int channels,
                      int height, int width, int max_disp, void* stream) {
    if (!feat_l || !feat_r || !disp_l || !disp_r) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || channels <= 0 || height <= 0 || width <= 0 || max_disp <= 0)
        return fail(S3R_ERR_INVALID, "disparity dims must be positive");
    if ((size_t)2 * channels * width * sizeof(float) > 64 * 1024)
        return fail(S3R_ERR_INVALID, "disparity read-out: a feature row pair (2*C*W floats) must fit 64 KiB of LDS");
    hipStream_t s = (hipStream_t)stream;
    const double px = (double)batch * height * width;
    ProfScope ps(s, F_DISP, 0, 0.0, 4.0 * px * (2.0 * channels + 2.0));
    hipError_t e = s3r::launch_disparity_wta(feat_l, feat_r, disp_l, disp_r, batch, channels, max_disp, height, width, s);
    if (e != hipSuccess) return hip_fail(e, "disparity read-out launch");
    return S3R_OK;
}

int s3r_disparity_epe(const float* pred, const float* gt, float* epe, int32_t* count, int batch, int64_t pixels,
                      void* stream) {
    if (!pred || !gt || !epe || !count) return fail(S3R_ERR_INVALID, "null tensor pointer");
    if (batch <= 0 || pixels <= 0) return fail(S3R_ERR_INVALID, "epe dims must be positive");
    hipStream_t s = (hipStream_t)stream;
    ProfScope ps(s, F_DISP, 1, 0.0, 8.0 * batch * (double)pixels);
    hipError_t e = s3r::launch_disparity_epe(pred, gt, epe, count, batch, pixels, s);
    if (e != hipSuccess) return hip_fail(e, "epe launch");
    return S3R_OK;
}

}  // extern "C"
