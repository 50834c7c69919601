/*
 * yololite_hip.h -- C ABI of the MI355X-native (gfx950) YoloLite inference hot path.
 *
 * The reference (Lillthorin/YoloLite-Official-Repo) is pure Python and has no FFI / operator
 * registry; its hot path sits behind plain Python calls.  This header is the boundary a native
 * replacement introduces.  Every entry point cites the reference interface it replaces
 * (paths relative to the reference repository root).
 *
 * Conventions
 *   - extern "C", plain pointers and sizes; no torch / C++ types cross the boundary.
 *   - All *_dev pointers are device (HBM) pointers owned by the CALLER (e.g. torch tensors'
 *     data_ptr()); the context owns weights, activations and workspaces.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls are asynchronous
 *     on that stream and never synchronise implicitly, except yl_forward_timed.
 *   - Every function returns YL_OK (0) or a negative yl_status; nothing throws across the ABI.
 *   - One context per device; a context is not thread-safe (one stream at a time).
 *   - Deterministic: results are bitwise repeatable run to run (no floating-point atomics).
 *   - Arithmetic is fp32 throughout (the reference CPU path is fp32).
 */
#ifndef YOLOLITE_HIP_H
#define YOLOLITE_HIP_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YL_ABI_VERSION 5
#define YL_MAX_LEVELS 8

typedef struct yl_ctx yl_ctx;
typedef int32_t yl_status;

enum {
  YL_OK = 0,
  YL_ERR_INVALID = -1,      /* bad argument / inconsistent model description          */
  YL_ERR_HIP = -2,          /* a HIP runtime call failed (see yl_last_error)          */
  YL_ERR_NOMEM = -3,        /* device or host allocation failed                       */
  YL_ERR_STATE = -4,        /* call not valid for this context (e.g. no layers)       */
  YL_ERR_UNSUPPORTED = -5,  /* valid request the kernels do not implement             */
  YL_ERR_CAPACITY = -6      /* a caller-provided buffer is too small                  */
};

/* activations fused into conv epilogues (reference: nn.ReLU / nn.SiLU in model_v2.py:21,34,132,299;
 * timm ReLU / ReLU6 inside the backbones) */
enum {
  YL_ACT_NONE = 0, YL_ACT_RELU = 1, YL_ACT_RELU6 = 2, YL_ACT_SILU = 3,
  YL_ACT_GELU = 4,      /* erf form, x * 0.5 * (1 + erf(x / sqrt 2)): timm convnextv2 `GlobalResponseNormMlp` (ABI v5)   */
  YL_ACT_RELU_LAB = 5   /* ReLU followed by timm's LearnableAffineBlock (hgnetv2 `ConvBNAct(use_lab=True)`): two scalars,
                           lab_scale * relu(v) + lab_bias; valid as `act` of YL_OP_STEM / YL_OP_CONV / YL_OP_DW only (ABI v5) */
};

/* layer kinds of the forward program */
enum {
  YL_OP_STEM = 0,  /* dense kxk conv reading the NCHW fp32 network input (Cin<=4), writes NHWC   */
  YL_OP_CONV = 1,  /* dense kxk conv (groups=1) on NHWC, optional depthwise prologue, fused
                      bias/act/residual/nearest-upsample-add epilogue, optional head-layout store  */
  YL_OP_DW = 2,    /* stand-alone depthwise kxk conv on NHWC with bias/act                         */
  YL_OP_STEMBLOCK = 3, /* fused network entry: stem 3x3 s2 (w,b,act) -> dense 3x3 s2 pad 1 (w2,b2,act2, cout c2)
                         -> optional 1x1 (w3,b3,act3, cout c3), NCHW input to NHWC output; the stem's
                         full-resolution activation (the largest tensor of the network) never reaches HBM  */
  YL_OP_SE = 4,    /* squeeze-excite gate of timm's SqueezeExcite (efficientnetv2 `ir_..._se0.25` blocks behind
                      model_v2.py:94-100): in_slot [B,H,W,cin] -> out_slot [B,1,1,cin],
                      gate = sigmoid(w2 . act(w . mean_hw(x) + b) + b2); w = conv_reduce [cout][cin][1][1], b [cout],
                      w2 = conv_expand [cin][cout][1][1], b2 [cin] (cout = the reduced width, c2 must equal cin).
                      The spatial mean is a fixed-order two-pass sum (no floating-point atomics: bitwise repeatable).
                      The gate multiplies the INPUT of the conv that names it in `scale_slot`.                     */
  /* ---- ABI v5: the op kinds of timm's hgnetv2 (configs/models/edge_xl.yaml:4) and convnextv2 (configs/v2_models/yololite_l.yaml:4)
   * feature extractors behind model_v2.py:94-100,266-272.  Element-wise / reduction passes, HBM-bound. */
  YL_OP_POOL = 5,  /* max-pool k x k, stride, window origin (oy*stride - pad_t, ox*stride - pad_l) over the input EXTENDED WITH
                      ZEROS (timm hgnet StemV2: F.pad(x, (0,1,0,1)) then MaxPool2d(2, 1)); cin == cout                        */
  YL_OP_COPY = 6,  /* channel-slice copy: in_slot [B,H,W,cin] -> channels [out_ch_off, out_ch_off + cin) of out_slot
                      [B,H,W,C_total]; torch.cat(..., dim=1) = one copy per input (hgnet StemV2 / HighPerfGpuBlock)            */
  YL_OP_LN = 7,    /* LayerNorm over the channels of every pixel (timm LayerNorm2d / nn.LayerNorm on channels-last):
                      (x - mean) / sqrt(var + eps) * w + b, biased variance, w / b [cin]; cin == cout                         */
  YL_OP_GRN = 8,   /* gate of timm's GlobalResponseNorm: in_slot [B,H,W,cin] -> out_slot [B,1,1,cin],
                      g = sqrt(sum_hw x^2) (fixed-order two-pass sum), gate = 1 + w * g / (mean_c(g) + eps), w [cin].  With the
                      gate as `scale_slot` of the following 1x1 conv and GRN's bias folded into that conv's bias by the host
                      (W2 . beta), the conv computes fc2(x + (beta + w * (x * n))) without writing the normalised tensor      */
  YL_OP_NHWC4 = 9  /* the NCHW fp32 network input [B,3,S,S] -> NHWC [B,S,S,4] (channel 3 = 0) for stems the MFMA stem kernel
                      does not cover (convnext: 4x4 stride 4, 96 outputs -> a YL_OP_CONV with cin = 4 follows); in_slot ignored */
};

/*
 * One fused layer.  Weights are HOST pointers in PyTorch layout with BatchNorm already folded
 * (w' = w * gamma/sqrt(var+eps), b' = beta - mean*gamma/sqrt(var+eps)); yl_create repacks them into
 * the MFMA fragment order and uploads them.  Replaces the nn.Conv2d / nn.BatchNorm2d / activation /
 * F.interpolate(nearest)+add / view-cat-permute sequences of scripts/model/model_v2.py:15-53,
 * 179-192,337-350 and of the timm backbone (model_v2.py:94-100,266-272).
 */
typedef struct {
  int32_t op;                 /* YL_OP_*                                                         */
  int32_t in_slot;            /* input tensor slot (ignored for YL_OP_STEM: reads the net input)  */
  int32_t out_slot;           /* output tensor slot, or -1 when head_level >= 0                   */
  int32_t res_slot;           /* tensor added after bias+act (residual), -1 = none                */
  int32_t up_slot;            /* tensor nearest-upsampled to the output size and added, -1 = none */
  int32_t head_level;         /* >=0: store as detection level [B,A,S,S,5+C] (cout = A*(5+C))     */
  int32_t cin, cout;
  int32_t k, stride, pad_t, pad_l;
  int32_t act;                /* YL_ACT_* applied after bias                                      */
  int32_t in_shift;           /* YL_OP_CONV, k>1, no prologue: the conv reads its input nearest-upsampled by
                                 2^in_shift (F.interpolate(scale_factor=2) folded into the tap addressing)   */
  int32_t dw_k;               /* YL_OP_CONV: 0 = none, else depthwise kxk prologue on input.  YL_OP_STEMBLOCK (round 6): 3 = the
                               * second conv is DEPTHWISE 3x3 stride 1 pad 1 (w2 [c1][1][3][3], c2 == cout == 32) followed by
                               * the 1x1 w3 -- timm's EfficientNet-Lite entry conv_stem -> blocks.0.0 as one launch            */
  int32_t dw_stride, dw_pad_t, dw_pad_l, dw_act;
  const float* w;             /* CONV/STEM: [cout][cin][k][k]; DW: [cout][1][k][k]                */
  const float* b;             /* [cout] or NULL                                                   */
  const float* dw_w;          /* [cin][1][dw_k][dw_k] or NULL                                     */
  const float* dw_b;          /* [cin] or NULL                                                    */
  /* YL_OP_STEMBLOCK: second and optional third conv of the fused entry block.
   * YL_OP_CONV with c2 > 0: fused inverted-residual block -- in_slot has c2 channels, w2/b2/act2 is the 1x1
   * EXPANSION [cin][c2][1][1] applied first, then the depthwise prologue (dw_*, stride 1 or 2) on the cin expanded
   * channels, then the 1x1 projection (w,b,act,res_slot); the expanded tensor never reaches HBM.  With up_slot >= 0
   * (cin channels) the nearest-upsampled tensor is added to the EXPANSION's output before act2 -- the FPN pair
   * "lateral 1x1 (+bias) + upsample-add, then depthwise smooth block" (model_v2.py:359-361) as one launch.
   * YL_OP_CONV with c3 > 0 (k > 1, no depthwise prologue, cout <= 96, c3 <= 32): a 1x1 conv w3/b3/act3 [c3][cout][1][1]
   * chained behind this conv in the same launch; out_slot then has c3 channels and the cout-channel tensor never
   * reaches HBM.
   * 0 / NULL otherwise. */
  int32_t c2, act2;           /* STEMBLOCK: 3x3 stride-2 pad-1 conv [c2][cout][3][3]              */
  int32_t c3, act3;           /* 1x1 conv: [c3][c2][1][1]; c3 = 0 -> absent                       */
  const float* w2;
  const float* b2;
  const float* w3;
  const float* b3;
  int32_t scale_slot;         /* YL_OP_CONV, 1x1 stride 1, no depthwise prologue: slot [B,1,1,cin] (a YL_OP_SE output) whose
                                 values multiply the conv's input per image and input channel before the GEMM
                                 (x * gate, then conv_pwl: timm InvertedResidual.forward); -1 = none                 */
  int32_t reserved0;          /* 0                                                                                */
  /* ---- ABI v5 */
  float lab_scale, lab_bias;  /* act == YL_ACT_RELU_LAB: the two scalars of the LearnableAffineBlock                              */
  float eps;                  /* YL_OP_LN / YL_OP_GRN                                                                            */
  int32_t out_ch_off;         /* YL_OP_COPY: first channel of out_slot written; 0 otherwise                                      */
} yl_layer;

/*
 * Model + detection-head geometry.  Mirrors what build_model_from_meta() (tools/infer.py:34-77)
 * derives from a checkpoint's meta and what forward() returns (model_v2.py:352-383):
 * level l is a tensor [B, level_anchors[l], level_size[l], level_size[l], 5+num_classes],
 * last-dim order [tx,ty,tw,th,tobj,cls_0..cls_{C-1}] (model_v2.py:345-350).
 */
typedef struct {
  int32_t abi_version;        /* YL_ABI_VERSION                                                   */
  int32_t img_size;           /* square network input S (x is [B,3,S,S] NCHW fp32)                */
  int32_t in_channels;        /* 3                                                                */
  int32_t num_classes;        /* C                                                                */
  int32_t num_masks;          /* NM: mask coefficients appended to every level row (0 = detector);
                                 level rows are [tx,ty,tw,th,tobj, cls_0..C-1, mc_0..NM-1]                */
  int32_t proto_slot;         /* activation slot holding the mask prototypes [B,PH,PW,NM] (-1 = none)   */
  int32_t num_levels;         /* L <= YL_MAX_LEVELS                                               */
  int32_t level_size[YL_MAX_LEVELS];
  int32_t level_anchors[YL_MAX_LEVELS];
  int32_t num_slots;          /* activation tensors                                               */
  const int32_t* slot_h;      /* [num_slots]                                                      */
  const int32_t* slot_w;
  const int32_t* slot_c;
  int32_t num_layers;         /* 0 = post-processing-only context                                 */
  const yl_layer* layers;
} yl_model_desc;

/* post-processing pipelines */
enum {
  YL_POST_MAIN = 0,      /* tools/infer.py:460-493  torchvision-nms semantics, per-class cap        */
  YL_POST_FALLBACK = 1,  /* tools/infer.py:247-389  greedy nms (+1e-6), min-side>=2, global top-k   */
  YL_POST_EVAL = 2       /* scripts/helpers/helpers.py:87-153  torchvision-nms semantics, no cap    */
};
enum { YL_CENTER_V8 = 0, YL_CENTER_SIMPLE = 1 };              /* utils_ms.py:83-88   */
enum { YL_WH_SOFTPLUS = 0, YL_WH_V8 = 1, YL_WH_EXP = 2 };     /* utils_ms.py:91-99   */
enum { YL_NMS_TORCHVISION = 0, YL_NMS_GREEDY = 1 };           /* tools/infer.py:134-163 */

typedef struct {
  int32_t mode;               /* YL_POST_*                                                        */
  float conf_thr;             /* keep iff score > conf_thr (strict)                               */
  float iou_thr;
  int32_t per_class_cap;      /* keep[:cap] per class (300 in the reference's nms()); <=0 = none  */
  int32_t topk;               /* FALLBACK only: global top-k by score; <=0 = none                 */
  int32_t max_out;            /* rows per image available in dets_dev / keep_idx_dev              */
  int32_t center_mode;        /* YL_CENTER_*                                                      */
  int32_t wh_mode;            /* YL_WH_*                                                          */
  const float* backmap_dev;   /* optional [B][5] = padx,pady,scale,w0,h0 (tools/infer.py:508-516) */
  int32_t fallback_nms;       /* YL_POST_FALLBACK only: the primitive behind nms() (tools/infer.py:134-152).
                                 YL_NMS_TORCHVISION (0, default) = what nms() runs when torchvision imports (a normal
                                 reference install); YL_NMS_GREEDY = its pure-torch loop (IoU + 1e-6, keep <= thr),
                                 taken when the import fails -- the situation in which tools/infer.py itself reaches
                                 decode_anchorfree_like_train (utils_ms.py:4 imports torchvision too)            */
} yl_post_cfg;

/* ---- lifetime ---------------------------------------------------------------------------------
 * yl_create replaces build_model_from_meta + load_state_dict + .to(device).eval()
 * (tools/infer.py:34-102): it validates the layer program, packs and uploads the weights.        */
yl_status yl_create(const yl_model_desc* desc, int32_t device_id, yl_ctx** out);
/* A second context of the SAME model on the same device: shares the packed weights (immutable after yl_create, freed by
 * the last owner), copies the current options, and owns everything a call touches -- activation arenas, post-processing
 * workspace, internal streams, cached hipGraphs.  Contexts are independent (one call at a time per context, any number of
 * contexts at once): a serving host keeps ONE CONTEXT PER BATCH IN FLIGHT and issues yl_predict for batch i on context
 * i % K from its own stream, so that the launch chain of batch i+1 overlaps the latency-bound tail of batch i instead of
 * waiting behind the join of its chunks (K = 2, one internal stream each: +10 % images/s on edge_n 640x640 B=64 against
 * back-to-back calls on one context).  The reference has no counterpart (one model object, one call at
 * a time: tools/infer.py:435-516); this is the C form of serving.ServingPipeline.  ABI v4.                          */
yl_status yl_clone(const yl_ctx* src, yl_ctx** out);
/* Do kernels on HIP streams `a` and `b` of device `device_id` RUN CONCURRENTLY?  ROCm hands every stream one of
 * GPU_MAX_HW_QUEUES hardware queues when it is created, round-robin over the streams the PROCESS has created so far (idle and
 * destroyed ones included), and two streams on one queue serialise.  Measured on MI355X / ROCm 7.2 (round 6): the same
 * two-batches-in-flight loop runs at 46.3 k or at 39 k images/s, and back-to-back calls on a two-stream context at 42.5 k or
 * 27.9 k, depending only on how many streams had been created before -- so neither the library nor a host can pick "a second
 * stream" blindly.  The probe launches a CHAIN of eight dependent 25 us spin kernels (one wave each) on each stream, interleaved,
 * and reads the wall clock: *overlap = 1 when both chains finished in less than 1.5 chain times (a good pair: ~231 us, an
 * aliasing pair: ~465 us; lone kernels overlap on every pair and show nothing).  Both streams are synchronised by the call; not capturable.  The library uses
 * it for its own chunk streams (a candidate that does not overlap the caller's stream and its siblings is replaced); a serving
 * host uses it for its per-batch streams (serving.ServingPipeline does).  `a` / `b`: hipStream_t, NULL = the null stream.  */
yl_status yl_streams_overlap(int32_t device_id, void* a, void* b, int32_t* overlap);
void yl_destroy(yl_ctx* ctx);
const char* yl_strerror(yl_status s);
const char* yl_last_error(const yl_ctx* ctx);   /* detail of the last failure on this context     */
int32_t yl_abi_version(void);

/* ---- forward ----------------------------------------------------------------------------------
 * Replaces model(x) (model_v2.py:352-377 / :194-224): x_dev is [B,3,S,S] NCHW fp32; level_out_dev
 * holds num_levels device pointers, level l receiving the contiguous tensor [B,A_l,S_l,S_l,5+C].  */
yl_status yl_forward(yl_ctx* ctx, const float* x_dev, int32_t batch, float* const* level_out_dev,
                     void* stream);
/* Same, but records a HIP event pair around every layer on `stream`, synchronises, and returns the
 * per-layer durations (ms) in layer_ms[num_layers].  Measurement aid for bench.py (roofline).      */
yl_status yl_forward_timed(yl_ctx* ctx, const float* x_dev, int32_t batch, float* const* level_out_dev,
                           void* stream, float* layer_ms);
/* With option "time_split" 1, yl_predict runs as ONE chunk on the caller's stream (eagerly, or with option "graph" as two
 * replayed hipGraphs: conv layers | post-processing) with a HIP event
 * before the conv layers, between the last conv layer and the NMS, and after the NMS; this call waits for the last
 * event and returns the two intervals: infer_ms (backbone + neck + heads, decode fused into the head epilogues) and
 * post_ms (per-class NMS + back-map) -- the pre / infer / post split of the reference's timing harness
 * (export/infer_onnx.py:152-244, README.md:182-189).  Measurement aid: the split costs the chunk overlap. */
yl_status yl_last_timing(yl_ctx* ctx, float* infer_ms, float* post_ms);
/* Bytes of activation memory the context holds for the batch of its last forward / predict call (one arena per
 * batch chunk, tensors placed by liveness; see option "reuse_slots").  0 before the first call. */
int64_t yl_activation_bytes(const yl_ctx* ctx);
/* Copies activation slot `slot` (NHWC fp32, [B,h,w,c]) of the last forward to dst_dev (testing aid).  With option
 * "reuse_slots" on (default) only tensors that are outputs of the network (mask prototypes) are guaranteed to still
 * hold their values after the call; set the option to 0 to inspect intermediate tensors. */
yl_status yl_read_slot(yl_ctx* ctx, int32_t slot, int32_t batch, float* dst_dev, void* stream);
/* Options: "graph" (0/1: replay the whole call from a captured hipGraph), "streams" (1..4 internal
 * streams the batch is split over; default 2), "tile_m" (conv M-tile hint), "lanes" (0/1, default 0: neck/head
 * layers of the coarser levels run on a side stream next to the finest level's chain),
 * "nms_groups" (0..4, default 0 = auto: NMS workgroups per image, classes dealt to the groups by load; auto takes 4 at
 * evaluation thresholds (conf < 0.05: thousands of survivors per image) and 1 otherwise; main / eval modes),
 * "hybrid" (0/1, default 0: high-resolution layers run as full-batch launches, only the run of <= 1/16-resolution
 * layers is split into batch chunks over the internal streams), "batch_levels" (0/1, default 1: head trunks /
 * head outputs of all pyramid levels in one launch each),
 * "fuse_decode" (0/1, default 1: yl_predict decodes inside the head-output convs; the raw level tensors are
 * then NOT materialised; a model with mask coefficients gets ONLY the coefficient columns of its level rows written --
 * what yl_masks / yl_masks_image read -- by a second plain 1x1 launch per head ("fuse_head" 1, single-anchor levels,
 * 5+C <= 96; otherwise the whole raw rows as before)),
 * "fuse_head" (0/1, default 1: run-time launch fusion of layer pairs whose shapes are instantiated.  With "fuse_decode"
 * and "batch_levels", a head branch -- depthwise 3x3 -> 1x1 trunk of 96 or 64 channels feeding the 1x1 head output --
 * runs trunk, output conv and decode as ONE launch for all levels; a depthwise 3x3 -> 1x1 expand layer followed by the
 * 1x1 project conv that is its only consumer (48 -> 192 -> 48) runs as one launch in every call.  The intermediate
 * tensors are then never written (yl_read_slot of those slots returns stale data).  Same results bit for bit),
 * "time_split" (0/1, default 0: see yl_last_timing), "pre_norm" (0/1: see yl_preprocess),
 * "reuse_slots" (0/1, default 1: activation tensors share memory by liveness inside one arena per batch chunk;
 * 0 keeps every tensor of the forward pass),
 * "winograd" (0/1/2, default 1: dense 3x3 stride-1 pad-1 convolutions with >= 16 input and output channels as Winograd
 * F(2x2,3x3) -- 16 instead of 36 multiplications per 2x2 output tile; fp32.  1 = every such layer (the dense FPN smooth
 * blocks of the YOLOLiteMS neck, model_v2.py:125-127; the ConvBnAct / fused-MBConv expansions of the efficientnetv2
 * backbones), 2 = only the >= 64-channel layers on the LARGEST grid they occur on (the finest pyramid level's smooth
 * block), 0 = direct convolution everywhere.  Winograd rounds differently from the direct convolution (not bit-identical
 * to it); against the fp32 CPU oracle its score error is no larger than the direct path's -- measured over 4 weight seeds x
 * 32 images x 8400 candidates at 640x640 on yololite_m: max |score - oracle score| 1.4-1.7e-5 (0), 1.4-1.7e-5 (2),
 * 0.9-1.6e-5 (1); on the efficientnetv2 yololite_m: 1.0-1.3e-5 (0), 0.7-1.2e-5 (1) -- i.e. >= 5.9x inside the 1e-4 bar
 * (profiles/r04_winograd_margin*.json, tools/wino_margin.py)),
 * "split_k" (0/1, default 0: depthwise -> 1x1 layers with 49..64 outputs on grids of <= 20 x 20 pixels split their k-blocks
 * over the four waves of a workgroup (one tile per workgroup, partial sums joined in LDS in a fixed order).  A latency
 * option for small batches: edge_n batch-1 forward -9 %, batch-64 throughput -0.7 %.  Chosen per context, never by the
 * batch size, so results stay batch-invariant and bitwise repeatable within a setting; between the settings they differ
 * by fp32 rounding (another summation order of the same products).  The pip API (api.YoloLite) turns it on),
 * "mfma_bf16" (0/1, default 0: reduced-precision inference mode, never the parity path -- every conv rounds the lane's
 * operands (weights and activations) to bf16 in registers and issues v_mfma_f32_16x16x16_bf16 with fp32 accumulation;
 * tensors stay fp32 in HBM; raw logits within 3e-2 of the level maximum),
 * "mfma_f16" (0/1, default 0: the same with fp16 operands on v_mfma_f32_16x16x16_f16 -- 11 mantissa bits instead of 8:
 * raw logits within 4e-3 of the level maximum; exclusive with "mfma_bf16"),
 * "store_f16" (0/1, default 0, round 6: "mfma_f16" AND fp16 ACTIVATION TENSORS IN HBM -- the storage side of the reference's
 * fp16 autocast, scripts/helpers/evaluate.py:399,415.  Every activation tensor between two launches is fp16 (arenas are
 * planned at 2 bytes per element, loads widen, stores round to nearest even, arithmetic and accumulation fp32); the network
 * input, the detection level tensors, the mask prototypes and the squeeze-excite gates stay fp32, and so do the packed
 * weights (they are L2 / Infinity-Cache resident: 2.3 MB for edge_n).  MFMA tiles are the 16x16x16 fp16 ones of "mfma_f16".
 * Kernels that stage activations by raw LDS-DMA copies (window-in-LDS depthwise, Winograd) give way to their tap-load
 * predecessors in this mode; models with the hgnetv2 / convnextv2 element-wise ops are refused at the first forward
 * (YL_ERR_UNSUPPORTED); yl_read_slot hands out fp32 and refuses fp16 slots.  Raw logits within 8e-3 of the level maximum
 * (measured 3.7e-3 .. 5.9e-3 at 640x640); exclusive with the other two modes; never the parity path). */
yl_status yl_set_option(yl_ctx* ctx, const char* name, int32_t value);
/* Current value of an option (the library's default if it was never written; values are stored clamped to the
 * option's range, e.g. "streams" 1..4).  YL_ERR_INVALID for an unknown name.  Also "dev_select" (default 0): a word of
 * DEVELOPER kernel-selection switches for A/B runs and the bitwise kernel-equivalence tests -- per context, carried
 * with every launch; never needed in production (bit 0: stand-alone depthwise through the one-output-per-lane kernel,
 * 1: no weight-streaming 1x1 kernel, 2: no staged-patch 3x3 s2 kernel, 3: producer/consumer depthwise kernel on every
 * shape it supports (with "tile_m" 7), 4: no wave-autonomous depthwise kernel, 5-6: streamed depthwise->1x1 kernel
 * variant (0 default, 1 one n-group per item, 2 off), 7-8: streamed dense 3x3 waves per workgroup (0 auto, 1 four,
 * 2 eight, 3 off for 4-n-tile layers), 9: two m-tiles per wave in its 4-wave form, 10: depthwise -> 1x1 layers on
 * grids of <= 20 x 20 pixels with one wave per tile instead of the split-K form, 11: Winograd layers through the first
 * kernel form (every transform position in one wave), 12-13: item shape of the position-split Winograd kernel (0 auto),
 * 14: depthwise 3x3 -> wide 1x1 layers with the taps from L1/L2 instead of the window-in-LDS kernel, 15: that kernel on
 * every grid it supports, 16: the fused head launch with the taps from L1/L2 (yl_conv_dpp_kernel) instead of the
 * window-in-LDS form (yl_conv_dpw_kernel)).                                                                         */
yl_status yl_get_option(const yl_ctx* ctx, const char* name, int32_t* value);
/* Host-side query, no device needed: would yl_create accept a fused inverted-residual block (yl_layer with c2 > 0:
 * 1x1 expand c_in -> c_mid, depthwise dw_k x dw_k stride dw_stride, 1x1 project c_mid -> c_out) producing an
 * out_h x out_w tensor?  Returns 1 (workgroup-level-halo kernel yl_ir_kernel), 2 (per-wave kernel yl_uib_kernel:
 * stride 1, in/out grids multiples of 4) or 0 (not instantiated: emit the block as separate layers).  The host
 * "compiler" (program.py) asks this instead of mirroring the kernels' shape tables.                                */
int32_t yl_query_fused_block(int32_t c_in, int32_t c_mid, int32_t c_out, int32_t dw_k, int32_t dw_stride, int32_t out_h,
                             int32_t out_w);
/* Host-side query (ABI v5): can a depthwise dw_k x dw_k (stride dw_stride) conv on c_in channels be the PROLOGUE of the 1x1
 * conv c_in -> c_out that follows it (yl_layer.dw_k > 0; timm `conv_dw` -> `conv_pwl`, DWConvBlock model_v2.py:23-39), output
 * out_h x out_w?  2: the streamed-tap kernel takes it (yl_conv_dws_kernel: >= 192 depthwise channels, 7..22 output n-tiles,
 * 4x4-tileable output -- tf_efficientnet_lite stages 4-6), 1: the generic depthwise-prologue kernels do (taps + bias of all
 * channels within 32 KiB of LDS), 0: no -- emit the depthwise conv as its own YL_OP_DW layer.                          */
int32_t yl_query_dw_prologue(int32_t c_in, int32_t c_out, int32_t dw_k, int32_t dw_stride, int32_t out_h, int32_t out_w);

/* ---- pre-processing ----------------------------------------------------------------------------
 * Replaces letterbox() + cv2.cvtColor + /255 + (x-mean)/std + transpose of tools/infer.py:121-131,446-453
 * for a batch: `packed_u8_dev` holds the BGR uint8 HWC images back to back, `imgs_dev` one descriptor
 * per image (the caller computes the letterbox geometry exactly like the reference: scale=min(S/h,S/w),
 * nh,nw=int(round(.)), top=(S-nh)//2, left=(S-nw)//2).  x_dev receives [B,3,S,S] fp32.
 * Option "pre_norm" 1 switches the normalisation arithmetic to the evaluate path's (tools/evaluate.py:57-72 ->
 * scripts/data/augment.py:153-171: LongestMaxSize + PadIfNeeded give the SAME geometry -- scale S/max(h,w),
 * round-half-even sizes, floor-half padding, border 114 -- and A.Normalize computes (x - 255*mean) * (1/(255*std))
 * in float32 instead of (x/255 - mean)/std).                                                          */
typedef struct {
  int64_t offset;             /* byte offset of image b in packed_u8_dev                          */
  int32_t h0, w0;             /* source size                                                      */
  int32_t nh, nw;             /* resized size inside the letterbox                                */
  int32_t top, left;          /* padding                                                          */
} yl_pre_image;
yl_status yl_preprocess(yl_ctx* ctx, const uint8_t* packed_u8_dev, const yl_pre_image* imgs_dev, int32_t batch,
                        float* x_dev, void* stream);

/* ---- decode -----------------------------------------------------------------------------------
 * Replaces decode_preds_anchorfree (scripts/helpers/utils_ms.py:26-123): levels -> box [B,N,4]
 * xyxy pixels clamped to [0,S-1], obj [B,N,1] logits, cls [B,N,C] logits; N = sum A_l*S_l^2.        */
yl_status yl_decode(yl_ctx* ctx, const float* const* levels_dev, int32_t batch, int32_t center_mode,
                    int32_t wh_mode, float* box_dev, float* obj_dev, float* cls_dev, void* stream);

/* Forward + decode in ONE call: the exported "decoded" wire format of the reference (export/export_onnx.py:283-296:
 * boxes_xyxy [B,N,4], obj_logits [B,N,1], cls_logits [B,N,C]) straight from the network input; the raw level tensors
 * stay in the context's own buffers (not handed out).  Same arithmetic as yl_forward followed by yl_decode.       */
yl_status yl_forward_decoded(yl_ctx* ctx, const float* x_dev, int32_t batch, int32_t center_mode, int32_t wh_mode,
                             float* box_dev, float* obj_dev, float* cls_dev, void* stream);

/* ---- post-processing --------------------------------------------------------------------------
 * Replaces the score/threshold/per-class-NMS/(top-k)/(back-map) code of tools/infer.py:460-516
 * (MAIN), tools/infer.py:247-389 (FALLBACK) and helpers.py:87-136 (EVAL).
 * dets_dev  [B][max_out][6] = x1,y1,x2,y2,score,class   (class asc, score desc; FALLBACK after a
 *           fired top-k: score desc)
 * counts_dev[B] = number of detections produced (may exceed max_out: rows beyond it are dropped)
 * keep_idx_dev optional [B][max_out] candidate index n of every detection (NULL to skip).          */
yl_status yl_postprocess(yl_ctx* ctx, const float* const* levels_dev, int32_t batch,
                         const yl_post_cfg* cfg, float* dets_dev, int32_t* counts_dev,
                         int32_t* keep_idx_dev, void* stream);
/* forward + postprocess (YoloLite.predict hot loop).  With option "fuse_decode" (default) the raw head rows
 * never reach memory; set it to 0 to have the context's level buffers filled as by yl_forward.
 * keep_idx_dev: optional [B][max_out] candidate indices (needed by yl_masks), NULL to skip.          */
yl_status yl_predict(yl_ctx* ctx, const float* x_dev, int32_t batch, const yl_post_cfg* cfg,
                     float* dets_dev, int32_t* counts_dev, int32_t* keep_idx_dev, void* stream);

/* ---- multi-GPU exchange ----------------------------------------------------------------------
 * The reference is single-device; sharding the batch over GPUs needs exactly ONE exchange per step: an all-gather
 * of the packed per-rank result.  Every rank runs yl_predict on its shard with dets_dev / counts_dev pointing INTO
 * one flat buffer  local_dev = [ dets: b*max_out*6 floats | counts: b int32 ]  (row_floats = b*max_out*6 + b; equal
 * shards), then this call issues ncclAllGather(local_dev -> all_dev[world][row_floats]) on `stream` -- asynchronous,
 * no pack / unpack kernels.  `nccl_comm` is the caller's ncclComm_t (RCCL).  RCCL is resolved at run time from the
 * libraries already loaded in the process (YL_ERR_UNSUPPORTED if none is).  Python hosts use torch.distributed
 * instead (yololite_amd.dist.DetGatherer: same buffer layout). */
yl_status yl_allgather_dets(yl_ctx* ctx, void* nccl_comm, const float* local_dev, int64_t row_floats, float* all_dev,
                            void* stream);

/* ---- instance masks (BUILD-DEFINED: the reference repository contains no mask code; parity unpinned) --
 * For detection i of image b (candidate keep_idx[b][i] of the levels of the last yl_forward/yl_predict on
 * this context, box = its decoded box in network-input pixels):
 *     m(y,x) = sigmoid( sum_k mc_k * proto[b][y][x][k] )          proto = slot `proto_slot`, [PH,PW,NM]
 *     mask(y,x) = m(y,x) > thr  and  x1*PW/S <= x < x2*PW/S  and  y1*PH/S <= y < y2*PH/S
 * masks_dev: uint8 [B][max_out][PH][PW] (rows >= counts[b] are left untouched).                       */
yl_status yl_masks(yl_ctx* ctx, const float* const* levels_dev, int32_t batch, const int32_t* counts_dev,
                   const int32_t* keep_idx_dev, int32_t max_out, float thr, uint8_t* masks_dev, void* stream);
/* Masks at IMAGE resolution -- the form the pip API returns (README.md:38-42: results['masks']).  BUILD-DEFINED like
 * yl_masks.  For detection i of image b (row dets_dev[b][i] = x1,y1,x2,y2,.. as written by yl_predict / yl_postprocess,
 * i.e. ALREADY back-mapped when `backmap_dev` was given there; candidate keep_idx[b][i]) and every pixel (y,x) of the
 * output grid of image b (h_b x w_b = out_hw_dev[b]: the original image, or S x S without back-map):
 *     xs = (x + 0.5) * scale + padx,  ys = (y + 0.5) * scale + pady        letterbox coordinate of the pixel centre
 *                                                                          (scale 1, pad 0 without backmap_dev)
 *     u  = max(xs, 0) * (PW / S) - 0.5 clamped at 0,  v likewise           source index of a bilinear PW/S resize,
 *                                                                          align_corners = false, edges replicated
 *     m  = bilinear interpolation of sigmoid(mc . proto[b]) at (v, u)      4 prototype pixels
 *     mask(y,x) = m > thr  and  x1 <= x < x2  and  y1 <= y < y2
 * Output of image b starts at masks_dev + mask_off_dev[b] bytes: [min(counts[b], max_out)][h_b][row] with row = w_b
 * uint8 (packed = 0) or ceil(w_b / 32) uint32 words, bit k of word j = pixel 32 j + k (packed = 1).
 * max_h / max_w: the largest h_b / w_b of the batch.  The caller sizes masks_dev from the counts, or gives every
 * image a fixed capacity of max_out masks (mask_off_dev[b] = b * max_out * h * row: no host read of the counts, the
 * call stays asynchronous).  packed = 1 needs mask_off_dev[b] % 4 == 0; 16-byte aligned offsets give full-width
 * stores.  Limits: prototype width PW <= 1024 (img_size <= 4096), batch <= 8192 (YL_ERR_HIP / invalid value beyond).
 * Every byte of the [min(counts[b], max_out)][h_b][row] block of image b is written (zeros outside the boxes). */
yl_status yl_masks_image(yl_ctx* ctx, const float* const* levels_dev, int32_t batch, const float* dets_dev,
                         const int32_t* counts_dev, const int32_t* keep_idx_dev, int32_t max_out, float thr,
                         const float* backmap_dev, const int32_t* out_hw_dev, const int64_t* mask_off_dev, int32_t max_h,
                         int32_t max_w, int32_t packed, uint8_t* masks_dev, void* stream);
/* Replaces nms(boxes, scores, iou_th, max_det) (tools/infer.py:134-152): keep_dev[max_det] receives
 * the kept indices in score-descending order, count_dev[0] their number (<= max_det).              */
yl_status yl_nms(yl_ctx* ctx, const float* boxes_dev, const float* scores_dev, int32_t n, float iou_thr,
                 int32_t nms_impl, int32_t max_det, int32_t* keep_dev, int32_t* count_dev, void* stream);

/* ---- evaluate-path consumers ----------------------------------------------
 * The two O(detections x ground truths) matching loops of the reference's evaluation, as device
 * kernels.  They take no context (stateless; current HIP device); all pointers are device pointers.
 *
 * yl_eval_match replaces the greedy per-(image, category) matching of build_curves_from_coco
 * (scripts/data/p_r_f1.py:31-78 and :100-118).  Keys k = 0..num_keys-1 are the distinct
 * (image_id, category_id) pairs; detections of key k are rows det_off[k]..det_off[k+1]-1 of det_xywh,
 * ALREADY in score-descending stable order; ground truths likewise through gt_off.  Boxes are
 * [x,y,w,h] float64 exactly as the reference's python floats; IoU is evaluated in float64 with the
 * reference's operation order (iou_xywh, p_r_f1.py:31-41).  For every detection, in order: the
 * not-yet-matched ground truth with the largest IoU > 0 (first index on ties) is taken; if that IoU
 * >= iou_thr the detection is a true positive and the ground truth becomes matched.
 *   tp_dev[Nd]     1 = true positive, 0 = false positive
 *   match_dev[Nd]  index (within the key) of the matched ground truth, -1 if none   (may be NULL)
 *   gt_matched_dev[Ng]  scratch AND output: zeroed by the call, 1 where a ground truth was matched */
yl_status yl_eval_match(const double* det_xywh_dev, const int32_t* det_off_dev, const double* gt_xywh_dev,
                        const int32_t* gt_off_dev, int32_t num_keys, int32_t num_gt, double iou_thr,
                        uint8_t* tp_dev, int32_t* match_dev, uint8_t* gt_matched_dev, void* stream);
/* The 0..1 confidence sweep of build_curves_from_coco (p_r_f1.py:96-124): for every threshold
 * thr[s] (ascending float64, the reference uses np.linspace(0,1,steps)) the number of true / false
 * positives among the counted detections with score >= thr[s].  Because the greedy matching of a
 * score-descending list is prefix-stable, these are suffix sums of a histogram of the tp flags of ONE
 * matching pass (yl_eval_match) -- the reference re-runs the matching per threshold.
 *   counted_dev[n] 1 = detection takes part in the sweep (its key has ground truth), may be NULL = all
 *   tp_ge_dev[steps], fp_ge_dev[steps]  int32 outputs                                                   */
yl_status yl_eval_sweep(const double* score_dev, const uint8_t* tp_dev, const uint8_t* counted_dev, int32_t n,
                        const double* thr_dev, int32_t steps, int32_t* tp_ge_dev, int32_t* fp_ge_dev,
                        void* stream);
/* Replaces the matching loops of create_confusion_matrix (scripts/helpers/evaluate.py:96-153).
 * Images i = 0..num_images-1 are the images that HAVE ground truth; their detections (already
 * filtered by score >= score_thresh and in score-descending stable order) are rows
 * det_off[i]..det_off[i+1]-1; boxes are float32 [x1,y1,x2,y2] as produced by the reference's
 * xywh_to_xyxy (:23-25); IoU in float32 with iou_matrix's operation order (:27-57, union clipped at
 * 1e-6).  Class-agnostic: the best ground truth is the FIRST argmax over all ground truths of the
 * image; true positive iff its IoU >= iou_thr and it is unmatched (:128-140).
 *   cm_dev[(C+1)*(C+1)] int32, row = true class, column = predicted class, index C = background;
 *   zeroed by the call.                                                                               */
yl_status yl_eval_confusion(const float* det_xyxy_dev, const int32_t* det_cls_dev, const int32_t* det_off_dev,
                            const float* gt_xyxy_dev, const int32_t* gt_cls_dev, const int32_t* gt_off_dev,
                            int32_t num_images, int32_t num_gt, int32_t num_classes, float iou_thr,
                            int32_t* cm_dev, uint8_t* gt_matched_dev, void* stream);

/* ---- Kalman-SORT tracker bank (reference tools/tracker.py:9-326) -------------
 * The reference's KalmanSortTracker follows ONE stream on the host.  A yl_tracker holds `num_streams`
 * independent trackers on the device (capacity `max_tracks` tracks each); yl_track_update advances all of
 * them by one frame from the packed detections of yl_predict / yl_postprocess without a host round trip:
 *   dets_dev   [S][max_out][6] = x1,y1,x2,y2,score,class      counts_dev [S] (clamped to max_out)
 * Semantics per stream = KalmanSortTracker.update(boxes, scores, classes) (tools/tracker.py:211-326):
 * predict, greedy IoU assignment (descending IoU, optional same-class mask, stop below iou_threshold),
 * Kalman update, new tracks for unmatched detections (ids from 1, detection order), drop tracks unseen
 * for more than max_age frames, report tracks updated this frame with hits >= min_hits, in list order.
 * Constructor defaults of the reference: iou_threshold 0.3, max_age 15, min_hits 2, match_by_class 1.
 * Outputs (device): out_id/out_cls [S][max_tracks] int32, out_box [S][max_tracks][4] xyxy,
 * out_score [S][max_tracks], out_count [S].  fp32 like the reference; the 7x7 / 4x4 products are not
 * summed in BLAS order, so boxes agree to rounding (tests: 1e-3 px), ids / classes / counts exactly.
 * Equal IoUs are taken in flat-index order (numpy's argsort leaves ties unspecified).
 * Tracks beyond max_tracks are not created; yl_track_stats reports how many were lost.                */
typedef struct yl_tracker yl_tracker;
yl_status yl_track_create(int32_t device, int32_t num_streams, int32_t max_tracks, float iou_threshold,
                          int32_t max_age, int32_t min_hits, int32_t match_by_class, yl_tracker** out);
void yl_track_destroy(yl_tracker* t);
/* reset() of the reference (tools/tracker.py:191-193) for one stream, or all with stream_index = -1 */
yl_status yl_track_reset(yl_tracker* t, int32_t stream_index, void* stream);
yl_status yl_track_update(yl_tracker* t, const float* dets_dev, const int32_t* counts_dev, int32_t max_out,
                          int32_t* out_id_dev, float* out_box_dev, int32_t* out_cls_dev, float* out_score_dev,
                          int32_t* out_count_dev, void* stream);
/* Grows the per-stream capacity to new_max_tracks (<= 4096), keeping every stream's state; synchronises the
 * device.  The output tensors of yl_track_update must then have the new row length.  No-op when not larger. */
yl_status yl_track_grow(yl_tracker* t, int32_t new_max_tracks);
/* synchronises the device; host arrays [num_streams] (either may be NULL) */
yl_status yl_track_stats(yl_tracker* t, int32_t* ntracks_host, int32_t* overflow_host);

#ifdef __cplusplus
}
#endif
#endif /* YOLOLITE_HIP_H */
