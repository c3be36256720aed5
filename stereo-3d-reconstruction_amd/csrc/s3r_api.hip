// C-ABI of libs3r_hip.so (declared in include/s3r.h): argument validation, kernel dispatch by
// layer shape, the chain/stage runners and the event-based kernel profiler.  Host code only.
#include "../../include/s3r.h"
#include "s3r_kernels.h"

#include <cstdarg>
#include <cstdlib>
#include <cstdio>
#include <atomic>
#include <cstring>
#include <mutex>
#include <vector>

namespace {

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

constexpr int64_t kMaxElems = (int64_t)1 << 31;
constexpr int64_t kMaxBytes = (int64_t)1 << 32;

// ---------------------------------------------------------------- profiler
enum Family { F_MFMA = 0, F_STEM = 1, F_HEAD = 2, F_COSTVOL = 3, F_LINEAR = 4, F_CHAMFER = 5, F_IOU = 6, F_PACK = 7, F_PAD = 8, F_DISP = 9 };

struct Prof {
    std::mutex mu;
    std::atomic<bool> on{false};
    int cap = 0;
    unsigned gen = 0;             // bumped by every enable / disable: a scope opened under an older pool skips its stop
    int device = -1;              // the device the event pool was created on
    std::vector<hipEvent_t> ev;   // 2 per record
    std::vector<s3r_prof_record> rec;
} g_prof;

// A scope takes COPIES of its two event handles under the lock, so nothing of the pool is touched outside it; its
// stop record happens under the lock as well and is skipped when the pool was rebuilt meanwhile (s3r_profile_enable
// from another thread: the handles would be destroyed events).  Events live on the device that was current at
// s3r_profile_enable: launches on another device are not profiled (a record there would fail).
struct ProfScope {
    bool active = false;
    int slot = -1;
    unsigned gen = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int launches = 1;     // kernel launches inside the scope (a conv may be cut into bulk + remainder, + split-K finish)
    int algo = 0;         // what ran: 0 direct, 1 / 2 / 3 the Winograd serial / class-parallel / dual form
    double exec = -1.0;   // MFMA FLOPs executed (< 0: the algorithmic count)
    hipStream_t stream;
    ProfScope(hipStream_t s, int family, int tag, double flops, double bytes) : stream(s) {
        if (!g_prof.on.load(std::memory_order_relaxed)) return;
        int dev = -1;
        if (hipGetDevice(&dev) != hipSuccess) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (!g_prof.on.load(std::memory_order_relaxed) || dev != g_prof.device || (int)g_prof.rec.size() >= g_prof.cap) return;
        slot = (int)g_prof.rec.size();
        s3r_prof_record r;
        r.family = family; r.tag = tag; r.ms = 0.f; r.flops = flops; r.bytes = bytes; r.launches = 1;
        r.exec_flops = flops; r.algo = 0; r.reserved = 0;
        g_prof.rec.push_back(r);
        gen = g_prof.gen;
        e0 = g_prof.ev[2 * slot];
        e1 = g_prof.ev[2 * slot + 1];
        active = hipEventRecord(e0, stream) == hipSuccess;
        if (!active) g_prof.rec.pop_back();
    }
    ~ProfScope() {
        if (!active) return;
        std::lock_guard<std::mutex> lk(g_prof.mu);
        if (gen != g_prof.gen) return;                       // the pool this scope belongs to is gone
        (void)hipEventRecord(e1, stream);
        if (slot < (int)g_prof.rec.size()) {
            g_prof.rec[slot].launches = launches;
            g_prof.rec[slot].algo = algo;
            if (exec >= 0.0) g_prof.rec[slot].exec_flops = exec;
        }
    }
};

// ---------------------------------------------------------------- layer geometry
struct Geo {
    int nd;            // spatial dims
    int in, out;       // logical edge sizes
    int in_p, out_p;   // edge sizes of the halo-padded buffers
    int64_t in_sp, out_sp;         // logical voxels per channel
    int64_t x_elems, y_elems;      // elements of the (padded) buffers
    int64_t x_store, y_store;      // their storage in 4-byte units (bf16 buffers take half)
    int64_t w_elems;
    double flops, bytes;           // algorithmic (unpadded) work of the layer
};

int out_size(const s3r_conv_desc* d) {
    if (d->op == S3R_OP_LINEAR) return 1;
    if (d->op == S3R_OP_DECONV) return (d->in_size - 1) * d->stride - 2 * d->pad + d->k;
    return (d->in_size + 2 * d->pad - d->k) / d->stride + 1;
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
    if (d->op == S3R_OP_DECONV && !(d->ndim == 3 && d->k == 4 && d->stride == 2 && d->pad == 1))
        return fail(S3R_ERR_INVALID, "ConvTranspose is supported for ndim=3,k=4,s=2,p=1 only");
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

enum Route { R_STEM, R_HEAD, R_MFMA, R_LINEAR };

// which kernel serves a layer shape (halos are checked separately, by check_halos)
int route(const s3r_conv_desc* d, Route* r) {
    if (d->op == S3R_OP_LINEAR) { *r = R_LINEAR; return S3R_OK; }
    if (d->op == S3R_OP_CONV && d->ndim == 2 && d->cin == 3 && d->cout == 32 && d->k == 3 && d->stride == 2 &&
        d->pad == 1 && d->act == S3R_ACT_RELU) { *r = R_STEM; return S3R_OK; }
    if (d->op == S3R_OP_CONV && d->cout == 1 && d->k == 1 && d->stride == 1 && d->pad == 0 &&
        (ipow(d->in_size, d->ndim) % 4) == 0) { *r = R_HEAD; return S3R_OK; }
    if (d->dtype == S3R_BF16 ? d->cin % 32 == 0 : d->cin % 16 == 0) { *r = R_MFMA; return S3R_OK; }
    return fail(S3R_ERR_INVALID, "no kernel for this layer shape (cin=%d cout=%d k=%d s=%d p=%d ndim=%d): the MFMA path "
                "needs cin %% 16 == 0", d->cin, d->cout, d->k, d->stride, d->pad, d->ndim);
}

// input halo the layer's kernel needs (the MFMA gather reads its zero padding from memory)
int need_halo(const s3r_conv_desc* d, Route r) {
    if (r != R_MFMA) return 0;
    return d->op == S3R_OP_DECONV ? 1 : d->pad;
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
    return d->dtype != S3R_BF16 && d->op == S3R_OP_DECONV && d->ndim == 3 && d->k == 4 && d->stride == 2 && d->pad == 1 &&
           d->cin % s3r::wino_bk() == 0 && d->in_size >= 4 && (d->in_size & 3) == 0;
}
// the descriptor can run its layer's Winograd form
bool wino_desc_ok(const s3r_conv_desc* d) {
    if (!(wino_layer(d) || dwino_layer(d)) || d->act == S3R_ACT_SIGMOID || d->in_halo != 1 || d->ksplit > 1) return false;
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
    return wino2_ax(d) >= 0 && d->act != S3R_ACT_SIGMOID && d->in_halo == d->pad && d->ksplit <= 1 &&
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
enum { ALG_DIRECT = 0, ALG_WINO = 1, ALG_WINO2 = 2 };
// the algorithm a descriptor resolves to; *form = the forced launch form of the one-axis kernel, or -1
int resolve_algo(const s3r_conv_desc* d, int* alg, int* form) {
    *alg = ALG_DIRECT;
    *form = -1;
    if (d->algo != S3R_ALGO_AUTO && d->algo != S3R_ALGO_DIRECT && d->algo != S3R_ALGO_WINOGRAD)
        return fail(S3R_ERR_INVALID, "unknown algo %d", d->algo);
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
    else if (wino_desc_ok(d)) *alg = ALG_WINO;
    return S3R_OK;
}
bool resolves_to_wino(const s3r_conv_desc* d) {
    int a, f;
    return resolve_algo(d, &a, &f) == S3R_OK && a != ALG_DIRECT;
}
// ---- two-axis form: sizes
struct Wino2Geo { int ax, m, n, ncls, out, sg, wp, kw, bmax; int64_t w_elems, v_sample, pos_sample; };
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
struct WinoNeed { int64_t v, slab, total; };
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
        w.slab = s3r::wino_slab_elems(kind, d->cout, nt, s3r::wino_plan(kind, d->cout, wino_kcls(d), nt, head, form));
    } else {
        const int bmax = wino_bmax(d);
        if (bmax <= 0) return w;
        if (d->in_layout != S3R_LAYOUT_WINO_H) w.v = wino_v_elems(d) / d->batch * bmax;
        for (int b0 = 0; b0 < d->batch; b0 += bmax) {          // (at most two different sub-batch sizes)
            const int nb = d->batch - b0 < bmax ? d->batch - b0 : bmax;
            if (b0 > 0 && nb == bmax) continue;
            const int nt = (int)wino_positions(d, nb);
            const int64_t sl = s3r::wino_slab_elems(kind, d->cout, nt, s3r::wino_plan(kind, d->cout, wino_kcls(d), nt, false, form));
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
        const int64_t sl = s3r::wino2_slab_elems(g2.ax, d->cout, nt, wino2_form_of(d, nt, form));
        if (sl > w.slab) w.slab = sl;
    }
    w.total = w.v + w.slab;
    return w;
}
double wino2_exec_flops(const s3r_conv_desc* d) {
    const Wino2Geo g2 = wino2_geo(d);
    return 2.0 * (double)g2.pos_sample * d->batch * g2.ncls * d->cout * d->cin * g2.kw;
}
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
    hipError_t e = s3r::launch_wino_diff(x, scratch, (long long)d->batch * d->cin, n + 2, n + 2, n + 2, p.xd_mode, s);
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
struct LaunchH { int tm, ksplit; };

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
    p.Ntotal = p.B * p.Nd * p.Nh * p.Nw;
    p.dS = s3r::FastDiv((unsigned)(p.Nd * p.Nh * p.Nw));
    p.dHW = s3r::FastDiv((unsigned)(p.Nh * p.Nw));
    p.dW = s3r::FastDiv((unsigned)p.Nw);
    p.ksplit = 1;
    return p;
}

// (tile cfg, gather width, split-K) of an MFMA-route layer: the caller's forced values or the heuristics
struct Launch { int cfg, vec, ksplit; };

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
struct Plan {
    std::vector<s3r_conv_desc> d;     // descriptors with planned halos
    std::vector<Route> r;
    std::vector<Geo> g;
    std::vector<int64_t> off;         // workspace offset of layer i's OUTPUT (-1: the caller's y)
    std::vector<char> fuse_head;      // layer i is an MFMA conv whose epilogue also runs layer i+1 (1x1 head)
    bool stem_wino = false;           // the stem writes layer 1's Winograd-transformed planes (launch_stem_wino), not its activation
    bool pad_input = false;
    int64_t pad_off = 0;
    int64_t scratch_off = 0, scratch_elems = 0;   // split-K slabs, shared by all layers of the chain
    int64_t total = 0;
};

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
        if (i + 1 < n) pl->d[i].out_halo = need_halo(&pl->d[i + 1], pl->r[i + 1]);
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
        if (pl->d[i + 1].cin != pl->d[i].cout || pl->d[i].out_halo != 0 || pl->d[i].act == S3R_ACT_SIGMOID) continue;
        if (pl->d[i].op == S3R_OP_CONV && resolves_to_wino(&pl->d[i])) continue;     // (the Winograd conv kernel has no fused-head epilogue)
        if (pl->d[i].dtype == S3R_BF16) {
            s3r::ConvParamsH ph = make_params_h(&pl->d[i], pl->g[i]);
            LaunchH Lh;
            if (resolve_launch_h(&pl->d[i], &ph, &Lh) != S3R_OK || Lh.ksplit != 1) continue;
        } else {
            s3r::ConvParams p = make_params(&pl->d[i], pl->g[i]);
            Launch L;
            if (resolve_launch(&pl->d[i], &p, &L) != S3R_OK || L.ksplit != 1) continue;
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

const char* s3r_last_error(void) { return g_err; }

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
            const int64_t taps = d->op == S3R_OP_DECONV ? 64 : ipow(d->k, g.nd);
            if (d->dtype == S3R_BF16) *elems = (taps * d->cin * cout_pad_h(d->cout) + 1) / 2;   // bf16, in float units
            else *elems = taps * d->cin * cout_pad(d->cout) + (wino_layer(d) ? wino_w_elems(d) : 0) + (dwino_layer(d) ? dwino_w_elems(d) : 0) +
                          (wino2_ax(d) >= 0 ? wino2_geo(d).w_elems : 0);
            break;
        }
    }
    return S3R_OK;
}

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
            else {
                e = s3r::launch_pack_conv(w, packed, d->cin, d->cout, cout_pad(d->cout),
                                          d->op == S3R_OP_DECONV ? 8 : (int)ipow(d->k, g.nd), d->op == S3R_OP_DECONV, s);
                if (e == hipSuccess && wino_layer(d))
                    e = s3r::launch_pack_wino(w, packed + ipow(d->k, g.nd) * d->cin * cout_pad(d->cout), d->cin, d->cout,
                                              cout_pad(d->cout), g.nd == 3 ? 3 : 1, 3, wino_r(d), s);
                if (e == hipSuccess && dwino_layer(d))
                    e = s3r::launch_pack_wino_deconv(w, packed + 64 * (int64_t)d->cin * cout_pad(d->cout), d->cin, d->cout, cout_pad(d->cout), s);
                if (e == hipSuccess && wino2_ax(d) >= 0)
                    e = s3r::launch_pack_wino2(w, packed + ipow(d->k, g.nd) * d->cin * cout_pad(d->cout) + (wino_layer(d) ? wino_w_elems(d) : 0),
                                               wino2_ax(d), d->cin, d->cout, cout_pad(d->cout), s);
            }
            break;
    }
    if (e != hipSuccess) return hip_fail(e, "pack weights");
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
    if (alg == ALG_WINO) return wino_need(d, form, false).total;
    if (alg == ALG_WINO2) return wino2_need(d, form).total;
    s3r::ConvParams p = make_params(d, g);
    Launch L;
    if ((rc = resolve_launch(d, &p, &L))) return rc;
    return s3r::conv_scratch_elems(p, L.cfg);
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
            e = s3r::launch_linear(x, packed_w, scale, shift, y, d->batch, d->cin, d->cout, d->act, scratch, s);
            break;
        }
        case R_MFMA: {
            s3r::ConvParams p = make_params(d, g);
            p.x = x; p.w = packed_w; p.scale = scale; p.shift = shift; p.y = y;
            int alg, form;
            if ((rc = resolve_algo(d, &alg, &form))) return rc;
            const bool wino = alg == ALG_WINO;
            if (alg == ALG_WINO2) {
                ProfScope ps(s, F_MFMA, d->tag, g.flops, g.bytes);
                ps.exec = wino2_exec_flops(d);
                ps.algo = 4;
                return wino2_run(d, g, p, x, packed_w, y, scratch, scratch_elems, form, s, &ps.launches);
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

int64_t s3r_chain_workspace_elems(const s3r_layer* layers, int n_layers) {
    Plan pl;
    int rc = plan_chain(layers, n_layers, &pl);
    if (rc) return rc;
    return pl.total;
}

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
    // the stems fetch render rows by 16-byte LDS-DMA from base + row * width: fp32 tensors from any allocator are aligned, an
    // 8-bit view at an odd offset is not
    if (((uintptr_t)images_left & 15) || ((uintptr_t)images_right & 15))
        return fail(S3R_ERR_INVALID, "render tensors must be 16-byte aligned (got %p, %p)", images_left, images_right);
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

int s3r_profile_enable(int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.on = false;
    ++g_prof.gen;                                            // live scopes of the old pool skip their stop record
    for (hipEvent_t ev : g_prof.ev) (void)hipEventDestroy(ev);
    g_prof.ev.clear();
    g_prof.rec.clear();
    g_prof.cap = 0;
    g_prof.device = -1;
    if (max_records <= 0) return S3R_OK;
    if (hipGetDevice(&g_prof.device) != hipSuccess) g_prof.device = -1;
    g_prof.ev.assign((size_t)2 * max_records, nullptr);
    for (auto& ev : g_prof.ev) {
        hipError_t e = hipEventCreate(&ev);
        if (e != hipSuccess) {
            for (hipEvent_t x : g_prof.ev) if (x) (void)hipEventDestroy(x);
            g_prof.ev.clear();
            return hip_fail(e, "hipEventCreate");
        }
    }
    g_prof.rec.reserve(max_records);
    g_prof.cap = max_records;
    g_prof.on = true;
    return S3R_OK;
}

int s3r_profile_reset(void) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    g_prof.rec.clear();
    return S3R_OK;
}

int s3r_profile_read(s3r_prof_record* out, int max_records) {
    std::lock_guard<std::mutex> lk(g_prof.mu);
    const int n = (int)g_prof.rec.size() < max_records ? (int)g_prof.rec.size() : max_records;
    for (int i = 0; i < n; ++i) {
        hipError_t e = hipEventSynchronize(g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventSynchronize");
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, g_prof.ev[2 * i], g_prof.ev[2 * i + 1]);
        if (e != hipSuccess) return hip_fail(e, "hipEventElapsedTime");
        g_prof.rec[i].ms = ms;
        if (out) out[i] = g_prof.rec[i];
    }
    return n;
}

}  // extern "C"
