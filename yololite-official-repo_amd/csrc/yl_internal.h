// Internal declarations shared by the HIP translation units (not part of the C ABI).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include "../../include/yololite_hip.h"

// activations that are not a clamp: SiLU runs in the conv kernels' generic epilogue (the fast clamp epilogues refuse it);
// GELU and ReLU + learnable affine (ABI v5: YL_ACT_POSTPASS) never reach a conv kernel -- the executor launches the layer with
// no activation (and no residual) and applies them in an element-wise pass over the output (yl_ops.hip: yl_act_kernel), so the
// hot kernels carry no code for them
#define YL_SMOOTH(a) ((a) >= YL_ACT_SILU)
#define YL_ACT_POSTPASS(a) ((a) >= YL_ACT_GELU)

#define YL_NUM_CU 256          // MI355X: 8 XCDs x 32 CUs
#define YL_LDS_KEYS_MAX 16384  // 64-bit sort keys that fit the 160 KiB LDS of one CU (128 KiB)

// ---- detection-level geometry handed to the post-processing kernels by value -------------------
struct YlLevels {
  const float* ptr[YL_MAX_LEVELS];  // level l: [B, A, S, S, E] contiguous
  int S[YL_MAX_LEVELS];
  int A[YL_MAX_LEVELS];
  int off[YL_MAX_LEVELS + 1];       // candidate offset of level l inside [0, N)
  float stride[YL_MAX_LEVELS];      // img_size / float(S)   (utils_ms.py:71)
  int L, N, E, C;
  float hi;                         // img_size - 1
};

struct YlDecodeP {
  int mode;            // YL_POST_*
  int center_mode, wh_mode;
  float4* boxes;       // [B][N]
  float* scores;       // [B][N]   (-inf for candidates removed by the fallback min-side filter)
  int* cls;            // [B][N]
};

struct YlNmsP {
  const float4* boxes;   // [B][N]
  const float* scores;   // [B][N]
  const int* cls;        // [B][N] or nullptr (single class 0)
  int N, C;
  float conf_thr, iou_thr;
  int impl;              // YL_NMS_*
  int cap;               // per-class cap (INT_MAX = none)
  int topk;              // global top-k (0 = none)
  int max_out;
  int* cls_ws;           // [B][4][C]: start, end, kept, off
  unsigned long long* gkeys;  // [B][gP] global key storage when survivors exceed the LDS capacity
  int gP;
  int lds_cap;           // keys that fit the dynamic LDS of this launch
  float* dets;           // [B][max_out][6]
  int* counts;           // [B]
  int* keep_idx;         // [B][max_out] or nullptr
  const float* backmap;  // [B][5] or nullptr
  float* tmp_dets;       // [B][N][6]  (fallback top-k staging) or nullptr
  int* tmp_idx;          // [B][N]
  // class-group split: G workgroups per image (grid.y), group g owns the classes c % G == g; a small merge kernel
  // (next in the stream) turns the per-class kept lists into the ordered output.  G == 1: one workgroup does it all.
  int G;
  int* kept_list;        // [B][G][N] candidate indices of the kept boxes, class segments at their sorted positions
  int* done;             // [B][64 ints = 256 bytes]: class -> group table written by group 0 for the merge kernel
};

// ---- activation element type of a translation unit.  fp32 everywhere except the FOURTH compilation of the conv units
// (-DYL_BF16=1 -DYL_F16=1 -DYL_F16S=1, option "store_f16", round 6): activation tensors live in HBM as fp16 -- the storage side
// of the reference's fp16 autocast (scripts/helpers/evaluate.py:399,415).  The struct layout is the same in every unit (these
// are pointers); the element type only decides pointer arithmetic and the load / store helpers (yl_dev.h: yl_ld4 / yl_st4).
#if defined(YL_F16S) && YL_F16S
typedef _Float16 yl_act_t;
#else
typedef float yl_act_t;
#endif

// ---- conv layer parameters (one struct for all conv kernels) -----------------------------------
struct YlConvP {
  const yl_act_t* x;     // input: NHWC [B,H,W,Cin]  (STEM: the fp32 NCHW network input [B,Cin,H,W], whatever yl_act_t is)
  const float* wp;       // packed weights (see yl_api.cpp pack_* for the layouts)
  const float* bias;     // [Npad16] (zero padded) or nullptr
  const yl_act_t* res;   // residual NHWC [B,OH,OW,N] or nullptr
  const yl_act_t* up;    // NHWC [B,UH,UW,N] nearest-upsampled and added, or nullptr
  yl_act_t* out;         // (out_f32: an fp32 tensor -- detection levels, mask prototypes -- also in the fp16-storage unit)
  const yl_act_t* zeros; // >= 1 KiB of zero bytes in HBM: out-of-range taps load from here (no select after the load)
  const float* dw_w;     // [dw_k*dw_k][Cin] tap-major
  const float* dw_b;     // [Cin] or nullptr
  int B, H, W, Cin;      // input tensor
  int OH, OW, N;         // output tensor (N = Cout)
  int k, stride, pad_t, pad_l;
  int act;
  int in_shift;          // KXK: input coordinates are shifted right by this (nearest 2^n upsample of x)
  int dw_k, dw_stride, dw_pad_t, dw_pad_l, dw_act;
  int MH, MW;            // grid the main conv reads (== H,W without prologue; dw output grid with it)
  int KB;                // ceil(Cin/16): 16-wide k blocks per tap
  int TK;                // taps * KB
  int NTtot;             // ceil(N/16)
  int CH;                // k-steps (of 16) per LDS weight chunk
  int UH, UW;
  long out_bstride;      // floats between images in `out` (OH*OW*N, or A*OH*OW*E for a head level)
  int M;                 // B*OH*OW
  int ntiles;            // M tiles
  // fused stem block (yl_stemblock_kernel): stem -> 3x3 s2 -> optional 1x1
  const float* w2p;      // packed [9][C1/16][ceil(C2/16)][64][4]
  const float* b2;       // padded
  const float* w3p;      // packed [ceil(C2/16)][ceil(C3/16)][64][4] or nullptr
  const float* b3;
  int C1, C2, C3, act2, act3;
  int SH, SW;            // stem output grid
  int tiles_x, tiles_y;  // 8x8 output tiles per image
  int sb_strip;          // yl_stemblock_kernel: tiles per strip (set by the launcher)
  // decode fused into the head-output conv (yl_predict): the epilogue turns the lane-distributed row of a
  // candidate (tx,ty,tw,th,obj,cls...) into box / score / class and writes the NMS inputs directly; the
  // raw level tensor is then only written when something else needs it (dec_raw: mask coefficients).
  float4* dec_boxes;     // [B][dec_N] of this launch's images, nullptr = off
  float* dec_scores;
  int* dec_cls;
  int dec_N;             // candidates per image (all levels)
  int dec_off;           // candidate index of this layer's cell 0 (level offset + anchor * S*S)
  int dec_C;             // classes
  int dec_mode, dec_center, dec_wh;
  int dec_raw;           // also write the raw rows
  float dec_stride, dec_hi;
  // level-batched launches (YlConvMulti): this problem owns blocks [blk0, blk0 + nblk) of grid.x; nblk == 0
  // means the whole grid (single-problem launch)
  int blk0, nblk;
  // Winograd F(2x2,3x3) image of a dense 3x3 stride-1 conv (yl_conv_wino_kernel, option "winograd"), or nullptr:
  // U = G g G^T per (cout, cin), packed [n-group of 32 couts][k-block][xi 0..15][2 n-tiles][64 lanes][4]
  const float* wino;
  // squeeze-excite gate (YL_OP_SE output, [B][Cin]) multiplying the 1x1 conv's input per image and channel, or nullptr
  const float* scale;
  // developer kernel-selection word of the context (yl_set_option "dev_select"; 0 in production): YL_DEV_*
  unsigned dev;
  // stand-alone depthwise conv feeding a squeeze-excite gate: the launch also leaves per-image partial channel sums of its
  // (activated) output, pool[b][r][C] for r < pool_wpi -- wave (b, r) owns every pool_wpi-th 4x2 pixel block of image b, so
  // the partials are a fixed function of the shape (deterministic) and the gate needs no pass over the tensor.  nullptr = off
  float* pool;
  int pool_wpi;
  // output row pitch in floats when it is not N (0 = N): the mask-coefficient part of a split head-output conv stores
  // its 32 columns into rows of 5+C+NM floats (yl_epi_fast only; scalar stores: the rows are not 16-byte aligned)
  int ldo;
  // fp16-storage mode: this layer's output tensor is fp32 all the same (head outputs -> the detection level buffers the
  // decode / NMS / mask kernels read, the mask prototypes); 0 in the fp32-storage units
  int out_f32;
};

// YlConvP::dev -- developer kernel-selection switches (A/B runs, bitwise kernel-equivalence tests); per context, never
// process-wide (the launchers used to read getenv switches)
#define YL_DEV_DW_TILE_OFF (1u << 0)   // stand-alone depthwise: yl_dw_kernel instead of yl_dw_tile_kernel
#define YL_DEV_PWS_OFF (1u << 1)       // wide 1x1 layers: yl_conv_pwt_kernel instead of yl_conv_pws_kernel
#define YL_DEV_S2C_OFF (1u << 2)       // 3x3 s2 + chained 1x1: yl_conv_mfma_kernel instead of yl_conv_s2c_kernel
#define YL_DEV_DWC_ALL (1u << 3)       // yl_conv_dwc_kernel (tile_m 7) on every shape it supports
#define YL_DEV_DWT_OFF (1u << 4)       // depthwise -> 1x1: yl_conv_dwh_kernel instead of yl_conv_dwt_kernel
#define YL_DEV_DWK(d) (((d) >> 5) & 3u)   // yl_conv_dwk_kernel: 0 default (a wave holds every n-group), 1 one n-group per item, 2 off
#define YL_DEV_KXK_NW(d) (((d) >> 7) & 3u) // yl_conv_kxk_kernel waves per workgroup: 0 auto, 1 four, 2 eight, 3 off (4-n-tile layers)
#define YL_DEV_KXK_MT2 (1u << 9)       // ... two m-tiles per wave in the 4-wave form
#define YL_DEV_DWT_NOSPLIT (1u << 10)  // depthwise -> 1x1 on <= 20x20 grids: one wave per tile instead of the split-K form
#define YL_DEV_WINO_V1 (1u << 11)      // Winograd: yl_conv_wino_kernel (all positions per wave) instead of yl_conv_wino2_kernel
#define YL_DEV_WINO_SHAPE(d) (((d) >> 12) & 3u) // yl_conv_wino2_kernel item shape: 0 auto, 1 (4,4), 2 (2,7), 3 two m-tiles
#define YL_DEV_DWL_OFF (1u << 14)      // depthwise 3x3 -> wide 1x1: yl_conv_dwk_kernel (taps from L1/L2) instead of yl_conv_dwl_kernel
#define YL_DEV_DWL_ALL (1u << 15)      // ... yl_conv_dwl_kernel on every grid it supports (partial windows, few items: the bitwise test)
#define YL_DEV_DPW_OFF (1u << 16)      // fused head launch: yl_conv_dpp_kernel (taps from L1/L2) instead of yl_conv_dpw_kernel (window in LDS)
#define YL_DEV_K3W_OFF (1u << 17)     // small-channel dense 3x3: the Winograd / direct kernels instead of yl_conv_k3w_kernel (bitwise A/B)

// squeeze-excite gate (yl_se.hip): fixed-order two-pass spatial mean + the two FCs + sigmoid
struct YlSeP {
  const float* x;        // [B][HW][C]
  float* partial;        // [B][P][C] scratch
  const float* w1;       // [RD][C]
  const float* b1;       // [RD]
  const float* w2;       // [RD][C] (transposed at upload: coalesced reads in the expand FC)
  const float* b2;       // [C]
  float* gate;           // [B][C]
  int B, HW, C, RD, P, act;
};
hipError_t yl_launch_se(const YlSeP& p, bool pooled, hipStream_t st);   // pooled: p.partial already holds P partials per image
int yl_se_parts(int HW, int C);            // partial sums per image the pool pass produces for this shape
int yl_dw_pool_wpi(int k, int stride, int cin, int n, int oh, int ow);   // > 0: yl_launch_dw can pool (waves per image), else 0

// element-wise / reduction ops of the hgnetv2 / convnextv2 backbones (yl_ops.hip; ABI v5 YL_OP_POOL .. YL_OP_NHWC4)
struct YlOpP {
  const float* x;        // input NHWC [B,H,W,C]  (YL_OP_NHWC4: the NCHW network input)
  float* out;
  const float* w;        // LN weight / GRN weight [C]
  const float* b;        // LN bias [C]
  float* partial;        // GRN: [B][P][C] scratch
  int B, H, W, C, OH, OW;
  int k, stride, pad_t, pad_l;   // pool
  int ldo, ch_off;       // copy: channels of the destination rows, first channel written
  int P;                 // GRN: partial sums per image
  float eps;
  const float* res;      // activation pass: residual added after the activation, or nullptr
  int act;               // activation pass: YL_ACT_GELU / YL_ACT_RELU_LAB
  float lab_s, lab_b;    // YL_ACT_RELU_LAB: lab_s * relu(v) + lab_b (timm LearnableAffineBlock of hgnetv2)
};
#define YL_OP_ACTPASS 100   // internal op code of yl_launch_op: the activation pass above (not an ABI op)
hipError_t yl_launch_op(int op, const YlOpP& p, hipStream_t st);
int yl_grn_parts(int HW);

// Up to 4 independent convolutions of identical kernel configuration in ONE launch (the FPN smooth blocks,
// head trunks and head outputs of all pyramid levels): every block serves one problem, the grid is split in
// proportion to the tile counts.  The coarse levels (20x20, 40x40: latency-bound on their own, 5 % and 19 % of
// the tiles) ride along with the 80x80 level instead of paying their own launches.
struct YlConvMulti {
  int n;
  YlConvP p[4];
};

// launchers implemented in the .hip files
hipError_t yl_launch_decode_score(const YlLevels& lv, int B, const YlDecodeP& p, hipStream_t st);
hipError_t yl_launch_decode_only(const YlLevels& lv, int B, int center_mode, int wh_mode, float* box,
                                 float* obj, float* cls, hipStream_t st);
hipError_t yl_launch_nms(const YlNmsP& p, int B, hipStream_t st);
hipError_t yl_launch_preprocess(const unsigned char* src, const void* imgs, int B, int S, float* out, int norm_mode,
                                hipStream_t st);
hipError_t yl_launch_masks(const YlLevels& lv, int B, const float* proto, int PH, int PW, int NM, int img_size,
                           const float4* boxes, const int* counts, const int* keep_idx, int max_out, float thr,
                           unsigned char* masks, hipStream_t st);
hipError_t yl_launch_masks_image(const YlLevels& lv, int B, const float* proto, int PH, int PW, int NM, int S,
                                 const float* dets, const int* counts, const int* keep_idx, int max_out, float thr,
                                 const float* backmap, const int* out_hw, const long long* mask_off, int max_h, int max_w,
                                 int packed, unsigned char* masks, hipStream_t st);
hipError_t yl_post_init();   // one-time function attributes (large dynamic LDS)

hipError_t yl_launch_stem(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv(const YlConvP& p, int tile_hint, hipStream_t st);
hipError_t yl_launch_conv_multi(const YlConvP* ps, int n, int tile_hint, hipStream_t st);   // n <= 4, same config
hipError_t yl_launch_dw(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_stemblock(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_stemdw(const YlConvP& p, hipStream_t st);   // stem -> depthwise 3x3 s1 -> 1x1 (EfficientNet-Lite entry, round 6)
hipError_t yl_stemblock_init();
// depthwise 3x3 -> 1x1 -> 1x1 head output + decode as one launch (yl_dpp.hip)
hipError_t yl_launch_conv_dpp(const YlConvP* ps, int n, hipStream_t st);
bool yl_dpp_supported(int cin, int cout, int c3, int oh, int ow);
hipError_t yl_dpp_init();
// dense 3x3 stride-2 conv (16 -> 48) + chained 1x1 from an LDS-staged patch (yl_dpp.hip)
hipError_t yl_launch_conv_s2c(const YlConvP& p, hipStream_t st);
// depthwise 3x3 -> 1x1 expand -> 1x1 project (+residual) as one launch (yl_dpp.hip)
hipError_t yl_launch_conv_dpq(const YlConvP& p, hipStream_t st);
// dense 3x3 with 16 / 32 -> <= 16 channels on large grids, window in LDS (yl_dpp.hip, round 6; fp32 only)
hipError_t yl_launch_conv_k3w(const YlConvP& p, hipStream_t st);
bool yl_dpq_supported(int cin, int cmid, int cout, int oh, int ow);
bool yl_stemblock_supported(int c1, int c2, int c3);
hipError_t yl_conv_init();
bool yl_uib_supported(int c1, int cmid, int n, int dk);
// block-cooperative depthwise -> 1x1 kernel (yl_convc.hip); hipErrorNotSupported = shape outside its limits
hipError_t yl_launch_conv_dwc(YlConvMulti& m, hipStream_t st);
hipError_t yl_launch_conv_dwc_bf16(YlConvMulti& m, hipStream_t st);
// wave-autonomous 1x1 conv for small pixel counts (yl_convc.hip); hipErrorNotSupported = not a plain 1x1 layer
hipError_t yl_launch_conv_pwt(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_pwt_bf16(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_pwt_multi(const YlConvP* ps, int n, hipStream_t st);
hipError_t yl_launch_conv_pwt_multi_bf16(const YlConvP* ps, int n, hipStream_t st);
// fused inverted-residual block with a workgroup-level halo (yl_convc.hip, round 3); hipErrorNotSupported = yl_uib_kernel
hipError_t yl_launch_conv_ir(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_ir_bf16(const YlConvP& p, hipStream_t st);
bool yl_ir_supported(int c1, int cmid, int n, int dk, int ds, int oh, int ow);
// plain 1x1 conv with a double-buffered weight stream for wide layers (yl_convc.hip); hipErrorNotSupported = pwt runs it
hipError_t yl_launch_conv_pws(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_pws_bf16(const YlConvP& p, hipStream_t st);
// dense k x k conv with a double-buffered weight stream (yl_convc.hip); hipErrorNotSupported = other kernel runs it
hipError_t yl_launch_conv_kxk(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_kxk_bf16(const YlConvP& p, hipStream_t st);
// wave-autonomous depthwise -> 1x1 kernel (yl_convc.hip); hipErrorNotSupported = shape outside its limits
hipError_t yl_launch_conv_dwt(YlConvMulti& m, hipStream_t st);
hipError_t yl_launch_conv_dwt_bf16(YlConvMulti& m, hipStream_t st);
// streamed-weight depthwise 3x3 -> 1x1 kernel for K >= 192 and more than 8 n-tiles (yl_convc.hip)
hipError_t yl_launch_conv_dwk(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_dwk_bf16(const YlConvP& p, hipStream_t st);
// depthwise k x k -> 1x1 for >= 192 depthwise channels: streamed 1x1 weights AND tap weights, halo patch through LDS (yl_convc.hip,
// round 5); hipErrorNotSupported = shape not instantiated
hipError_t yl_launch_conv_dws(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_dws_bf16(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_dws_f16(const YlConvP& p, hipStream_t st);
bool yl_dws_supported(int cin, int n, int dk, int ds, int oh, int ow);
// Winograd F(2x2,3x3) dense 3x3 (yl_convc.hip); hipErrorNotSupported = not this layer
hipError_t yl_launch_conv_wino(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_conv_wino_bf16(const YlConvP& p, hipStream_t st);
hipError_t yl_convc_init();
hipError_t yl_convc_init_bf16();
// bf16-MFMA builds of yl_conv.hip / yl_stemblock.hip (compiled a second time with -DYL_BF16=1, see yl_dev.h)
hipError_t yl_launch_conv_bf16(const YlConvP& p, int tile_hint, hipStream_t st);
hipError_t yl_launch_conv_multi_bf16(const YlConvP* ps, int n, int tile_hint, hipStream_t st);
hipError_t yl_conv_init_bf16();
hipError_t yl_launch_stemblock_bf16(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_stemdw_bf16(const YlConvP& p, hipStream_t st);
hipError_t yl_stemblock_init_bf16();
// fp16-MFMA builds (third compilation, -DYL_BF16=1 -DYL_F16=1: option "mfma_f16")
hipError_t yl_launch_conv_f16(const YlConvP& p, int tile_hint, hipStream_t st);
hipError_t yl_launch_conv_multi_f16(const YlConvP* ps, int n, int tile_hint, hipStream_t st);
hipError_t yl_conv_init_f16();
hipError_t yl_convc_init_f16();
hipError_t yl_launch_stemblock_f16(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_stemdw_f16(const YlConvP& p, hipStream_t st);
hipError_t yl_stemblock_init_f16();
// fp16-STORAGE builds (fourth compilation, -DYL_BF16=1 -DYL_F16=1 -DYL_F16S=1: option "store_f16"): fp16 operands and fp16
// activation tensors in HBM; YlConvP's activation pointers are _Float16* in those units (same struct layout)
hipError_t yl_launch_conv_f16s(const YlConvP& p, int tile_hint, hipStream_t st);
hipError_t yl_launch_conv_multi_f16s(const YlConvP* ps, int n, int tile_hint, hipStream_t st);
hipError_t yl_launch_stem_f16s(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_dw_f16s(const YlConvP& p, hipStream_t st);
hipError_t yl_conv_init_f16s();
hipError_t yl_convc_init_f16s();
hipError_t yl_launch_stemblock_f16s(const YlConvP& p, hipStream_t st);
hipError_t yl_launch_stemdw_f16s(const YlConvP& p, hipStream_t st);
hipError_t yl_stemblock_init_f16s();
