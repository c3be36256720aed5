// Host-side internals shared by the host translation units of libs3r_hip.so — s3r_prof.hip (event profiler), s3r_plan.hip (layer
// geometry, kernel policy, sizes, chain / arena planner) and s3r_api.hip (the C-ABI entry points and the runners that enqueue
// kernels).  Not part of the ABI (include/s3r.h) and not seen by the kernel sources (those share s3r_kernels.h).
#pragma once
#include "../../include/s3r.h"
#include "s3r_kernels.h"

#include <cstdint>
#include <vector>

namespace s3rh {

constexpr int64_t kMaxElems = (int64_t)1 << 31;
constexpr int64_t kMaxBytes = (int64_t)1 << 32;

// ---------------------------------------------------------------- errors (s3r_plan.hip): thread-local message behind s3r_last_error
const char* last_error();

// ---------------------------------------------------------------- profiler (s3r_prof.hip)
enum Family { F_MFMA = 0, F_STEM = 1, F_HEAD = 2, F_COSTVOL = 3, F_LINEAR = 4, F_CHAMFER = 5, F_IOU = 6, F_PACK = 7, F_PAD = 8, F_DISP = 9,
              F_AUX = 10 };    // F_AUX: a transform / difference / finish pass of a Winograd layer, nested inside its F_MFMA record

// One record: two events around everything enqueued while the scope lives.  A scope takes COPIES of its two event handles under
// the profiler's lock, so nothing of the pool is touched outside it; its stop record happens under the lock as well and is
// skipped when the pool was rebuilt meanwhile (s3r_profile_enable from another thread: the handles would be destroyed events).
// Events live on the device that was current at s3r_profile_enable: launches on another device are not profiled.
struct ProfScope {
    bool active = false;
    int slot = -1;
    unsigned gen = 0;
    hipEvent_t e0 = nullptr, e1 = nullptr;
    int launches = 1;     // kernel launches inside the scope (a conv may be cut into bulk + remainder, + split-K finish)
    int algo = 0;         // what ran: 0 direct, 1 / 2 / 3 the Winograd serial / class-parallel / dual form, 4 the two-axis algorithm
    double exec = -1.0;   // MFMA FLOPs executed (< 0: the algorithmic count)
    int family, prev_tag = 0;
    hipStream_t stream;
    ProfScope(hipStream_t s, int family, int tag, double flops, double bytes);
    ~ProfScope();
    ProfScope(const ProfScope&) = delete;
    ProfScope& operator=(const ProfScope&) = delete;
};

// ---------------------------------------------------------------- geometry, policy, sizes, planner (s3r_plan.hip)
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

enum Route { R_STEM, R_HEAD, R_MFMA, R_LINEAR };

enum { ALG_DIRECT = 0, ALG_WINO = 1, ALG_WINO2 = 2, ALG_WINO3 = 3 };      // (WINO3: the three-axis form of the transposed layers)

struct Wino2Geo { int ax, m, n, ncls, out, sg, wp, kw, bmax; int64_t w_elems, v_sample, pos_sample; };

struct WinoNeed { int64_t v, slab, total; };

struct LaunchH { int tm, ksplit; };

struct Launch { int cfg, vec, ksplit; };

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


int fail(int code, const char* fmt, ...);
int hip_fail(hipError_t e, const char* what);
int64_t ipow(int64_t b, int e);
int wino_r(const s3r_conv_desc*);
int out_size(const s3r_conv_desc* d);
int dil_of(const s3r_conv_desc* d);
bool staged_layer(const s3r_conv_desc* d);
struct StagedGeo { int cin_pad, step, pe, sp; int64_t elems; int bmax; };      // bmax > 0 (im2col): samples staged per pass
StagedGeo staged_geo(const s3r_conv_desc* d);
s3r::ConvParams make_params_staged(const s3r_conv_desc* d, const Geo& g);
// residue classes of a general ConvTranspose with dilation 1 (s3r_general.hip): class r of an axis = the outputs o with
// (o + pad) % stride == r; kr taps (0: none of that residue), ke >= 1 taps packed, positions q = qmin .. qmin + nq - 1 (nq <= 0: no output)
struct TClassAxis { int kr, ke, qmin, nq; };
bool im2col_layer(const s3r_conv_desc* d);      // a staged convolution with cin <= 8: unfolded to a 1 x 1 GEMM over cin k^nd rows
bool leaky_fused(const s3r_conv_desc* d);
bool needs_act_pass(const s3r_conv_desc* d);
bool tclass_layer(const s3r_conv_desc* d);
bool tclass_direct(const s3r_conv_desc* d);
// ... and a residue-class layer with k == stride, pad 0, out_pad 0 (every class is ONE tap on the same input element): one GEMM over
// cout x taps rows with a depth-to-space store instead of stride^ndim launches
bool tshuf_layer(const s3r_conv_desc* d);
s3r::ConvParams make_params_tshuf(const s3r_conv_desc* d, const Geo& g);
int want_halo(const s3r_conv_desc* d, Route r);
TClassAxis tclass_axis(const s3r_conv_desc* d, int r);
int tclass_halo(const s3r_conv_desc* d);
int64_t tclass_w_elems(const s3r_conv_desc* d);                      // the s^nd class slabs, in class order
// class (rd, rh, rw) as the direct kernel sees it: a stride-1 convolution over the staged tensor; false: the class has no output.
// *w_off = floats from the start of the packed image to the class's slab; *macs = its multiply-adds (CinPad channels)
bool make_params_tclass(const s3r_conv_desc* d, const Geo& g, int rd, int rh, int rw, s3r::ConvParams* q, int64_t* w_off, double* macs);
int geometry(const s3r_conv_desc* d, Geo* g);
int route(const s3r_conv_desc* d, Route* r);
int need_halo(const s3r_conv_desc* d, Route r);
int check_halos(const s3r_conv_desc* d, Route r);
int cout_pad(int cout);
int wino_mode();
bool wino_layer(const s3r_conv_desc* d);
bool dwino_layer(const s3r_conv_desc* d);
bool dwino3_layer(const s3r_conv_desc* d);
bool dwino3_desc_ok(const s3r_conv_desc* d);
int64_t dwino3_w_offset(const s3r_conv_desc* d);
struct Dwino3Need { int64_t diff, total; bool split; };
Dwino3Need dwino3_need(const s3r_conv_desc* d, int form);
bool wino_desc_ok(const s3r_conv_desc* d);
int wino2_ax(const s3r_conv_desc* d);
bool wino2_desc_ok(const s3r_conv_desc* d);
int wino2_max_edge();
int resolve_algo(const s3r_conv_desc* d, int* alg, int* form);
bool resolves_to_wino(const s3r_conv_desc* d);
Wino2Geo wino2_geo(const s3r_conv_desc* d);
int64_t wino_w_elems(const s3r_conv_desc* d);
int64_t wino_v_elems(const s3r_conv_desc* d);
int wino_bmax(const s3r_conv_desc* d);
int64_t dwino_w_elems(const s3r_conv_desc* d);
bool dwino_materialise(const s3r_conv_desc* d);
int64_t dwino_d_elems(const s3r_conv_desc* d);
int wino_kind(const s3r_conv_desc* d);
int wino_kcls(const s3r_conv_desc* d);
int64_t wino_positions(const s3r_conv_desc* d, int nb);
WinoNeed wino_need(const s3r_conv_desc* d, int form, bool head);
double wino_exec_flops(const s3r_conv_desc* d, const Geo& g);
int wino2_form_of(const s3r_conv_desc* d, int ntotal, int forced);
WinoNeed wino2_need(const s3r_conv_desc* d, int form);
double wino2_exec_flops(const s3r_conv_desc* d);
int cout_pad_h(int cout);
s3r::ConvParamsH make_params_h(const s3r_conv_desc* d, const Geo& g);
int resolve_launch_h(const s3r_conv_desc* d, s3r::ConvParamsH* p, LaunchH* L);
s3r::ConvParams make_params(const s3r_conv_desc* d, const Geo& g);
int resolve_launch(const s3r_conv_desc* d, s3r::ConvParams* p, Launch* L);
int64_t align_up(int64_t v, int64_t a);
int plan_chain(const s3r_layer* layers, int n, Plan* pl);

}  // namespace s3rh
