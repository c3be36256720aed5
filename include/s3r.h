/* s3r.h — C-ABI of libs3r_hip.so: the MI355X (gfx950) forward path of the Stereo2Voxel /
 * Stereo2Point network.  Plain pointers and sizes only; no torch / C++ types cross this boundary.
 *
 * What each entry point replaces on the reference side (the reference's model code lives on the
 * unmounted Stereo2Voxel / Stereo2Point branches, /root/reference/README.md:5, so the most specific
 * citation the mount supports is given):
 *
 *   s3r_conv_forward, s3r_encoder_forward   the torch conv2d+BatchNorm+ReLU calls inside the stereo
 *                                           feature encoder's nn.Module.forward        (README.md:5,73-74)
 *   s3r_encoder_forward_u8                  the same fed with the 8-bit renders the PNG decode yields: replaces the
 *                                           dataset transform's uint8 -> float32 / 255 as well (requirements.txt:5)
 *   s3r_cost_volume_forward                 the per-disparity shift/subtract Python loop that builds the
 *                                           disparity cost volume                       (README.md:75-76)
 *   s3r_decoder_forward (+conv/deconv/head) the torch conv3d / ConvTranspose3d + BN + ReLU + sigmoid
 *                                           calls of the voxel decoder                  (README.md:77)
 *   s3r_linear_forward                      the point decoder's nn.Linear layers        (README.md:36)
 *   s3r_chamfer_forward                     extensions/chamfer_dist (the reference's one native op,
 *                                           built by `python setup.py install`)         (README.md:64-65)
 *   s3r_voxel_iou                           the IoU metric of `runner.py --test`        (README.md:88-92)
 *   s3r_disparity_wta, s3r_disparity_epe    predicted left / right disparity and its end-point error
 *                                           against the disp_%02d_{l,r}.exr ground truth (README.md:75-76)
 *
 * Conventions
 *   - every tensor is fp32, contiguous, NCHW / NCDHW, resident in device memory owned by the caller;
 *     the library never allocates, frees or synchronises;
 *   - `stream` is a hipStream_t (NULL = the default stream); work is enqueued, not waited for;
 *   - return value: S3R_OK (0) or a negative s3r_status; s3r_last_error() gives a message for the
 *     calling thread; nothing throws across the ABI;
 *   - every tensor of one call must be < 2^31 elements and < 4 GiB (32-bit buffer offsets);
 *   - size queries (s3r_conv_scratch_elems, s3r_chain_workspace_elems, ...) do not depend on the device they are asked on: launch
 *     forms of one algorithm (bit-identical among themselves) are planned against the current device's compute-unit count and have
 *     different scratch footprints, so whenever the LIBRARY picks the form the query is sized for the largest one (r06).  A process
 *     with no device (host-only planning) plans launches for an unpartitioned MI355X (256 CUs).
 */
#ifndef S3R_H
#define S3R_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ABI 8 (r05) over ABI 7: s3r_conv_desc grew `dilation`, `out_pad`, `act_param` and three activations — the parameter-general
 * fp32 layers; `tile` = 6 under S3R_ALGO_WINOGRAD names the three-axis form of a transposed convolution (AUTO takes it from edge
 * 16 up: other bits than ABI 7 for such a layer); s3r_profile_detail and record family 10 (aux passes). */
#define S3R_ABI_VERSION 8

typedef enum s3r_status {
    S3R_OK = 0,
    S3R_ERR_INVALID = -1,     /* bad argument / unsupported shape */
    S3R_ERR_HIP = -2,         /* a HIP runtime call failed (message has hipGetErrorString) */
    S3R_ERR_WORKSPACE = -3    /* workspace too small */
} s3r_status;

typedef enum s3r_op {
    S3R_OP_CONV = 0,          /* Conv2d / Conv3d (ndim selects) */
    S3R_OP_DECONV = 1,        /* ConvTranspose2d / 3d.  3D k4 s2 p1 has the tuned kernels (parity classes, Winograd forms); every
                                 other (k, stride, pad, out_pad) with dilation 1 runs as stride^ndim residue classes, each a stride-1
                                 launch of the direct kernel over a halo-padded copy (the algorithmic multiplications); dilation > 1
                                 runs zero-stuffed (stride^ndim times as many) (fp32) */
    S3R_OP_LINEAR = 2         /* nn.Linear on the flattened input */
} s3r_op;

typedef enum s3r_dtype {
    S3R_F32 = 0,              /* fp32 activations, NCHW / NCDHW, v_mfma_f32_32x32x2_f32 (exact fp32) */
    S3R_BF16 = 1              /* bf16 activations, CHANNELS-LAST (B,[D,]H,W,C), v_mfma_f32_32x32x16_bf16, fp32
                                 accumulate; the stem still reads fp32 NCHW renders and the occupancy head still
                                 writes fp32 probabilities */
} s3r_dtype;

typedef enum s3r_act { S3R_ACT_NONE = 0, S3R_ACT_RELU = 1, S3R_ACT_SIGMOID = 2,
                       S3R_ACT_LEAKY_RELU = 3, S3R_ACT_ELU = 4, S3R_ACT_TANH = 5 /* ABI 8: fp32 convolution layers only */ } s3r_act;

/* Memory layout of an activation buffer (beyond NCHW vs channels-last, which the dtype fixes):
 *   S3R_LAYOUT_PLAIN   the (halo-padded) tensor as described under "Halos" below;
 *   S3R_LAYOUT_WINO_H  fp32 path, INPUT of a 3 x 3 [x 3] stride-1 pad-1 convolution only, in_halo 1, edge n a multiple of 4: the six
 *                      Winograd F(4,3)-along-H plane sets of the halo-padded tensor, (6, B, C, [n+2,] n/4, n+2) — set i, row q =
 *                      the i-th F(4,3) input-transform combination of the padded rows 4 q .. 4 q + 5 (csrc/s3r_kernels.h,
 *                      wino_rows_to_classes).  What s3r_cost_volume_forward_wino writes: the consumer then skips its input
 *                      transform.  The batch of such a call is bounded: s3r_conv_wino_input_elems returns 0 when the layer /
 *                      batch cannot take it.
 * (Value 1 was S3R_LAYOUT_S2D — parity-split intermediates for stride-2 consumers, bf16 and fp32 forms — through ABI 6: built,
 * bit-identical, measured slower / no gain inside the forward in rounds 2 and 3, never planned by default; removed in ABI 7.) */
/*   S3R_LAYOUT_WINO_DH the same for the TWO-AXIS kernel (Conv3d k3 s1 p1, edge a multiple of 4): 36 plane sets
 *                      (36, B, C, n/4, n/4, n+2), set 6 a + b = depth class a, row class b of the 6 x 6 window of padded depths
 *                      4 s .. 4 s + 5 and padded rows 4 q .. 4 q + 5 (rows first, then depths).  s3r_cost_volume_forward_wino2 writes it.
 *   S3R_LAYOUT_WINO_HW the two-axis kernel's input in 2D (Conv2d k3 s1 p1, edge n a multiple of 4): (36, C, P) with the positions
 *                      of the whole batch flat, P = B (n/4)^2 rounded up to a multiple of 64, position (b, q, s); set 6 a + b =
 *                      column class a, row class b of the 6 x 6 window of padded rows 4 q .. 4 q + 5 and padded columns
 *                      4 s .. 4 s + 5 (rows first, then columns).  It is also the one OUTPUT layout other than PLAIN: a
 *                      two-axis Conv2d whose consumer is another one writes these plane sets of its own (halo-1) output from
 *                      its finish kernel — s3r_chain_forward plans that hand-off itself (e6 -> e7 of this network). */
typedef enum s3r_layout { S3R_LAYOUT_PLAIN = 0, S3R_LAYOUT_WINO_H = 2, S3R_LAYOUT_WINO_DH = 3, S3R_LAYOUT_WINO_HW = 4 } s3r_layout;

/* Which convolution algorithm a layer's forward runs (ABI 7).  The fp32 3 x 3 [x 3] stride-1 pad-1 convolutions and the
 * transposed convolutions have two kernels — the direct implicit GEMM and a Winograd form with 1/2 .. 9/16 of the
 * multiplications (csrc/s3r_conv_wino.hip) — that agree to fp32 rounding, NOT bit for bit.  So the choice is part of the
 * descriptor and never depends on anything else a caller passes (workspace size, batch):
 *   S3R_ALGO_AUTO      the library's policy, a function of the layer's PER-SAMPLE geometry only: Winograd where the layer has
 *                      that form (fp32 Conv k3 s1 p1 with cin % 32 == 0, cout > 1, edge >= 4; ConvTranspose3d k4 s2 p1 over an
 *                      edge that is a multiple of 4, cin % 32 == 0; in_halo = 1, no sigmoid); direct otherwise, and direct whenever the descriptor
 *                      forces a direct-kernel tile / split-K (tile >= 0 or ksplit >= 1) or a non-plain layout.  The
 *                      process-level override S3R_WINO=0 (AUTO never picks Winograd) is read ONCE, when the library is loaded;
 *   S3R_ALGO_DIRECT    the direct kernel;
 *   S3R_ALGO_WINOGRAD  the Winograd kernel (S3R_ERR_INVALID if the layer has no such form or the descriptor cannot take it:
 *                      needs in_halo = 1, plain layouts — or S3R_LAYOUT_WINO_H input —, no split-K, no sigmoid).  `tile` >= 0 then
 *                      forces the launch FORM of the one-axis kernel (tuning / tests; every form gives the same bits, and the
 *                      library picks among them by batch): 0 serial, 1 class-parallel, 2 dual (bulk serial + remainder
 *                      class-parallel in one launch); `tile` = 6 (ABI 8): the THREE-AXIS form of a ConvTranspose3d k4 s2 p1 over an edge
 *                      of 8, 16 or 32 (F(2,2) along D, H and W: 27 / 64 of the multiplications; AUTO takes it from edge 16 up;
 *                      7 / 8 force its class-parallel / serial launch form: same bits, the library picks by batch); `tile` = 3: the TWO-AXIS algorithm (Conv3d k3 s1 p1 as F(4,3) x F(4,3) over D
 *                      and H, Conv2d k3 s1 p1 as F(4,3) x F(4,3) over H and W, Conv3d k4 s1 p0 as F(2,4) x F(2,4); in_halo = pad;
 *                      4 / 5 force its class-parallel / semi-fused launch form: same bits) — another algorithm, other bits than
 *                      the one-axis kernel; AUTO takes it for every stride-1 layer that has it and an edge <= 28 (e6, e7, v1, v3,
 *                      v5, v6 of this network).
 * A call whose scratch is smaller than s3r_conv_scratch_elems says for the RESOLVED algorithm fails with S3R_ERR_WORKSPACE; it
 * is never answered with the other kernel's bits. */
typedef enum s3r_algo { S3R_ALGO_AUTO = 0, S3R_ALGO_DIRECT = 1, S3R_ALGO_WINOGRAD = 2 } s3r_algo;

/* One layer's geometry.  Spatial sizes are cubic/square: `in_size` per axis, `ndim` axes.
 *
 * Halos.  The MFMA convolution kernels read their zero padding from memory: an activation may be
 * stored with a ZERO HALO of `halo` elements on every spatial axis, i.e. as a contiguous
 * (B, C, n+2*halo, ...) tensor whose border is zero and whose interior is the logical (B, C, n, ...)
 * tensor.  `in_halo` / `out_halo` describe the buffers `x` / `y` of s3r_conv_forward.  A layer with
 * padding p (or a ConvTranspose) served by the MFMA kernel needs in_halo >= p (>= 1); kernels write
 * interiors only, so a buffer zeroed once keeps its halo.  s3r_chain_forward plans the halos of all
 * intermediates itself and pads an unpadded chain input on the fly. */
typedef struct s3r_conv_desc {
    int32_t op;        /* s3r_op */
    int32_t ndim;      /* 2 or 3 (ignored for LINEAR) */
    int32_t batch;     /* B */
    int32_t cin, cout;
    int32_t in_size;   /* input edge (H=W[=D]) */
    int32_t k, stride, pad;
    int32_t act;       /* s3r_act */
    int32_t tag;       /* caller's label, echoed by the profiler */
    int32_t tile;      /* -1: library picks; >=0 (tuning): S3R_F32 direct kernel: MFMA tile cfg 0..7 + 16*gather_width (algo =
                          WINOGRAD: the launch form, see s3r_algo); S3R_BF16: 1,2,4 per-tap gather x128 positions (3: 128x128
                          couts; +16: 32-channel K tiles), 9,10 row-reuse gather, 5,6 / 21,22 plane-reuse gather (64- / 32-channel
                          K tiles), 23 plane-reuse 256x128 couts, 40 row-persistent (e2's geometry) */
    int32_t in_halo;   /* zero halo of the input buffer  (elements per spatial axis side) */
    int32_t out_halo;  /* zero halo of the output buffer */
    int32_t ksplit;    /* 0: library picks; >=1: force the split-K factor (must divide cin/16; bf16: cin/32) */
    int32_t dtype;     /* s3r_dtype: which path (layout + matrix instruction) the layer runs on */
    int32_t in_layout; /* s3r_layout of the input buffer: PLAIN, or the plane sets a producer wrote for this layer's Winograd kernel —
                          WINO_H (one-axis), WINO_DH (two-axis Conv3d), WINO_HW (two-axis Conv2d); in_halo must be 1 for those */
    int32_t out_layout;/* s3r_layout of the output buffer: PLAIN, or WINO_HW (a two-axis Conv2d writing its consumer's plane sets) */
    int32_t algo;      /* s3r_algo (ABI 7): AUTO = the library's geometry-only policy */
    /* ABI 8 — parameter-general layers (fp32 path).  The shapes this build's network has keep their tuned kernels; any other
     * (k, stride, pad, dilation) convolution with cin % 16 == 0 runs the direct kernel; everything else listed here goes through
     * the direct kernel too: cin % 16 != 0 behind a staged copy (channels zero-padded to 16; cin <= 8: unfolded so that every tap
     * of every channel is a K row, in sub-batches of <= 1 GiB), ConvTranspose2d / 3d with any k / stride / pad / output padding
     * (dilation 1: one stride-1 launch per output residue class, reading a halo of ceil(k / stride) — the producer's, when in_halo
     * provides it and cin % 16 == 0, else a halo-padded copy's; k == stride, pad 0: one GEMM with a depth-to-space store;
     * dilation > 1: the input zero-stuffed at the stride, the kernel flipped), LeakyReLU with a slope in [0, 1] inside the direct
     * kernel's epilogue, ELU / Tanh / other slopes as a pass of their own behind the layer.  0 / 0 / 0.f are NOT the neutral
     * values of `dilation`: a zero-initialised ABI-7 descriptor means dilation 1 and is read so. */
    int32_t dilation;  /* >= 1 (0 is read as 1) */
    int32_t out_pad;   /* ConvTranspose output_padding (< max(stride, dilation)) */
    float act_param;   /* S3R_ACT_LEAKY_RELU: negative slope; S3R_ACT_ELU: alpha (S3R_OP_LINEAR descriptors take the three too: a pass behind
                          the layer; the flat s3r_linear_forward entry, which has no parameter argument, takes none / relu / sigmoid) */
} s3r_conv_desc;

/* One layer of a stage: geometry + its packed weights + folded epilogue vectors (device pointers). */
typedef struct s3r_layer {
    s3r_conv_desc desc;
    const void* packed_w;    /* from s3r_conv_pack_weights (fp32 or bf16 image, per desc.dtype) */
    const float* scale;      /* [cout] gamma/sqrt(var+eps)            (NULL = 1) */
    const float* shift;      /* [cout] beta + (bias-mean)*scale       (NULL = 0) */
} s3r_layer;

int s3r_abi_version(void);
const char* s3r_last_error(void);

/* output edge of a layer: conv (n+2p-k)/s+1, deconv (n-1)s-2p+k, linear 1 */
int s3r_conv_out_size(const s3r_conv_desc* d);
/* size of the packed weight buffer for a layer IN 4-BYTE UNITS (>= the torch weight's numel on the fp32
 * path: couts are padded; about half of it on the bf16 path).  An fp32 3 x 3 [x 3] stride-1 pad-1 convolution packs every
 * form it has: the direct slab, the six Winograd F(4,3)-along-H class slabs (csrc/s3r_conv_wino.hip: half the
 * multiplications) and the 36 slabs of the two-axis form (2D over H, W; 3D over D, H; a 3D k4 valid layer its 25); which kernel a forward
 * runs is the descriptor's `algo` (s3r_algo above).  The transposed convolutions likewise: 72 F(2,2) x F(2,2) (parity class, class) slabs
 * behind the direct ones. */
int s3r_conv_packed_elems(const s3r_conv_desc* d, int64_t* elems);
/* repack a torch-layout weight (Conv: [cout][cin][k..]; ConvTranspose: [cin][cout][k..]; Linear:
 * [cout][cin]) into the kernel's K-major layout.  Device to device, on `stream`. */
int s3r_conv_pack_weights(const s3r_conv_desc* d, const float* w, void* packed, void* stream);
/* floats of scratch s3r_conv_forward needs for this layer under its resolved algorithm: split-K partial slabs (direct
 * kernel), the transformed input and the class-parallel slabs (Winograd kernel); 0 when it needs none */
int64_t s3r_conv_scratch_elems(const s3r_conv_desc* d);
/* y = act(conv(x) * scale + shift); dispatches to the stem / MFMA / head kernel by shape.  `scratch` must hold
 * s3r_conv_scratch_elems floats: a smaller one is S3R_ERR_WORKSPACE, never a silent switch to another kernel or another
 * split (ABI 6 ran such a layer unsplit / on the direct kernel: other bits for the same descriptor). */
int s3r_conv_forward(const s3r_conv_desc* d, const void* x, const void* packed_w, const float* scale,
                     const float* shift, void* y, float* scratch, int64_t scratch_elems, void* stream);

/* Run a chain of layers x -> y.  Every intermediate activation gets its own region of `ws`
 * (s3r_chain_workspace_elems floats; with 288 GB of HBM nothing is recycled), laid out with the halo
 * the next layer wants.  layers[0].desc.in_halo / layers[n-1].desc.out_halo describe x / y; the halos
 * of the intermediates are planned by the library (the descriptors' values are ignored for them).
 * `ws_fresh` != 0 makes the call zero the workspace first: pass 1 the first time a (chain, batch,
 * workspace) combination is used — or whenever anything else wrote to `ws` — and 0 afterwards. */
int64_t s3r_chain_workspace_elems(const s3r_layer* layers, int n_layers);
int s3r_chain_forward(const s3r_layer* layers, int n_layers, const void* x, void* y, float* ws, int64_t ws_elems,
                      int ws_fresh, void* stream);

/* Stage entry points (thin, shape-checked views of s3r_chain_forward):
 *   encoder: renders -> features (N,C,28,28), N = layers[0].desc.batch = 2B images: the B left renders
 *            (B,3,224,224) at `images_left`, the B right renders at `images_right` — two tensors, as the
 *            reference's forward receives them (README.md:73-74); the shared-weight tower runs once over all 2B
 *            images and the first kernel picks its source by image index, so nothing is concatenated.
 *            images_right = NULL: `images_left` holds all N images (any N).
 *   decoder: cost volume (B,2C,D,H,W) -> occupancy (B,32,32,32)  */
int s3r_encoder_forward(const s3r_layer* layers, int n_layers, const float* images_left, const float* images_right,
                        void* features, float* ws, int64_t ws_elems, int ws_fresh, void* stream);
int s3r_decoder_forward(const s3r_layer* layers, int n_layers, const void* volume, float* occupancy, float* ws,
                        int64_t ws_elems, int ws_fresh, void* stream);

/* The encoder on 8-BIT renders: (B,3,224,224) uint8 NCHW, as the reference's PNG decode yields them (OpenCV,
 * /root/reference/requirements.txt:5; README.md:73-74) — a quarter of the bytes across PCIe and into the first kernel.
 * The stem scales a sample by 1/255 as it reads it, with the single rounding of the host conversion
 * float32(u) / float32(255) it replaces: the features equal, bit for bit, those of s3r_encoder_forward on renders
 * converted that way on the host.  Everything else as s3r_encoder_forward (both precisions; images_right may be NULL). */
int s3r_encoder_forward_u8(const s3r_layer* layers, int n_layers, const uint8_t* images_left, const uint8_t* images_right,
                           void* features, float* ws, int64_t ws_elems, int ws_fresh, void* stream);

/* Hand-off from the bf16 path to an fp32 consumer (s3r_linear_forward on the latent, s3r_disparity_wta on the features):
 * x channels-last bf16 (batch, positions, channels) -> y fp32 (batch, channels, positions), exact. */
int s3r_channels_last_to_f32(const void* x, float* y, int batch, int channels, int64_t positions, void* stream);

/* vol[b,c,d,h,w] = L[b,c,h,w]-R[b,c,h,w-d] (w>=d), vol[b,C+c,d,h,w] = R[b,c,h,w]-L[b,c,h,w+d] (w+d<W), else 0.
 * `out_halo` > 0 writes the interior of a (B,2C,D+2h,H+2h,W+2h) buffer whose halo the caller zeroed. */
int s3r_cost_volume_forward(const float* feat_left, const float* feat_right, float* volume, int batch, int channels,
                            int max_disp, int height, int width, int out_halo, void* stream);

/* The volume written directly as the S3R_LAYOUT_WINO_H input of the 3D convolution that consumes it (halo 1):
 * (6, B, 2C, D+2, H/4, W+2) floats (F(4,3) groups), bit-identical to the input transform of the padded volume; the caller
 * zeroed the buffer once (the depth-halo planes are never written).  height a multiple of 4. */
int s3r_cost_volume_forward_wino(const float* feat_left, const float* feat_right, float* planes, int batch, int channels,
                                 int max_disp, int height, int width, void* stream);
/* floats of the S3R_LAYOUT_WINO_H input of layer `d` (in_halo = 1) if a forward of it would run the Winograd kernel under the
 * library's current policy (S3R_WINO) and the batch fits one call; 0 otherwise (hand the layer its plain input then). */
int64_t s3r_conv_wino_input_elems(const s3r_conv_desc* d);
/* ... and which layout that is: S3R_LAYOUT_WINO_H (the one-axis kernel), S3R_LAYOUT_WINO_DH (the two-axis kernel), or
 * S3R_LAYOUT_PLAIN (none: hand the layer its plain halo-padded input) */
int s3r_conv_wino_input_layout(const s3r_conv_desc* d);
/* the volume as the S3R_LAYOUT_WINO_DH input of the 3D convolution that consumes it: (36, B, 2C, D/4, H/4, W+2) floats,
 * bit-identical to the two-axis input transform of the padded volume; max_disp and height multiples of 4 */
int s3r_cost_volume_forward_wino2(const float* feat_left, const float* feat_right, float* planes, int batch, int channels,
                                  int max_disp, int height, int width, void* stream);

/* the same on channels-last bf16 features (B,H,W,C) -> volume (B,D+2h,H+2h,W+2h,2C); channels % 8 == 0 */
int s3r_cost_volume_forward_bf16(const void* feat_left, const void* feat_right, void* volume, int batch, int channels,
                                 int max_disp, int height, int width, int out_halo, void* stream);

/* y[b][o] = act(sum_i x[b][i] w[o][i] + bias[o]); w in torch Linear layout (no packing).  `scratch` holds the
 * split-K partial sums (s3r_linear_scratch_elems floats; reduced in a fixed order: deterministic). */
int64_t s3r_linear_scratch_elems(int batch, int cin, int cout);
int s3r_linear_forward(const float* x, const float* w, const float* bias, float* y, int batch, int cin, int cout,
                       int act, float* scratch, int64_t scratch_elems, void* stream);

/* squared-L2 nearest neighbours both ways; p (B,N,3), q (B,M,3) */
int s3r_chamfer_forward(const float* p, const float* q, float* dist1, float* dist2, int32_t* idx1, int32_t* idx2,
                        int batch, int n, int m, void* stream);

/* per-sample IoU of (pred > th) vs (gt > th) over `voxels` elements */
int s3r_voxel_iou(const float* pred, const float* gt, float threshold, float* iou, int batch, int64_t voxels,
                  void* stream);

/* Disparity read-out: winner-take-all over the shift-and-diff costs of the cost volume (same features, same
 * |L - R shifted| costs, volume never materialised).  feat_* (B,C,H,W) fp32; disp_* (B,H,W) fp32, integer-valued,
 * in feature-resolution pixels:
 *   disp_l[b,h,w] = first argmin_{d in [0, min(max_disp-1, w)]}     sum_c |L[b,c,h,w] - R[b,c,h,w-d]|
 *   disp_r[b,h,w] = first argmin_{d in [0, min(max_disp-1, W-1-w)]} sum_c |R[b,c,h,w] - L[b,c,h,w+d]|   */
int s3r_disparity_wta(const float* feat_l, const float* feat_r, float* disp_l, float* disp_r, int batch, int channels,
                      int height, int width, int max_disp, void* stream);

/* per-sample end-point error mean|pred - gt| over the pixels with a valid ground truth (finite, >= 0), and that
 * pixel count; epe = 0 where no pixel is valid */
int s3r_disparity_epe(const float* pred, const float* gt, float* epe, int32_t* count, int batch, int64_t pixels,
                      void* stream);

/* Kernel-level profiler: when enabled, every kernel the library launches is bracketed by HIP events
 * on the launch stream.  s3r_profile_read synchronises those events and returns, per launch, the
 * kernel family (0 mfma conv, 1 stem, 2 head, 3 cost volume, 4 linear, 5 chamfer, 6 iou, 7 pack, 8 pad copy, 9 disparity read-out / epe,
 * 10 aux: ONE transform / difference / finish / split-K-combine pass of a convolution layer — no matrix work, `bytes` = what it must
 * read and write — recorded INSIDE that layer's family-0 record, same tag: the layer's record includes its aux passes' time),
 * the caller's tag, milliseconds, and the algorithmic flops / bytes of that launch. */
typedef struct s3r_prof_record {
    int32_t family;
    int32_t tag;
    float ms;
    int32_t launches;   /* kernel launches bracketed by this record (a conv layer may be 1-3 launches) */
    double flops;       /* algorithmic (direct-form) FLOPs of the layer: SURVEY 8d's count */
    double bytes;
    double exec_flops;  /* FLOPs the kernel that ran EXECUTES on the matrix cores (= flops for the direct kernels; 1/2 .. 9/16 of it
                           for the Winograd forms) */
    int32_t algo;       /* what ran: 0 direct, 1 Winograd serial form, 2 class-parallel form, 3 dual form, 4 two-axis form, 5 three-axis
                           form (transposed layers), 6 its class-parallel launch form */
    int32_t reserved;
} s3r_prof_record;
int s3r_profile_enable(int max_records);   /* 0 disables and frees the event pool */
int s3r_profile_reset(void);
/* ABI 8.  level 1: the aux passes of the convolution layers (record family 10) get records of their own, nested inside their layer's;
 * 0 (default): layer records only — an event pair between two kernels of a layer costs queue time, which a profile taken for the
 * layers' durations must not carry (bench.py runs one pass of each kind) */
int s3r_profile_detail(int level);
int s3r_profile_read(s3r_prof_record* out, int max_records);   /* returns the number of records */

#ifdef __cplusplus
}
#endif
#endif /* S3R_H */
