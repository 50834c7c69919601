// C ABI (include/yololite_hip.h): context, weight packing, forward executor, post-processing driver.
#include "yl_internal.h"

#include <dlfcn.h>
#include <limits.h>
#include <math.h>
#include <stdio.h>
#include <stddef.h>
#include <string.h>

#include <algorithm>
#include <memory>
#include <chrono>
#include <string>
#include <vector>

#define YL_DW_LDS_MAX (32 * 1024)

namespace {

struct DevLayer {
  yl_layer d;            // host description (pointers nulled after packing)
  float* wp = nullptr;   // packed weights (device)
  float* bias = nullptr; // padded bias (device)
  float* dw_w = nullptr; // [taps][Cin] (device)
  float* dw_b = nullptr;
  float* w2p = nullptr;  // stem block: packed second / third conv
  float* b2 = nullptr;
  float* w3p = nullptr;
  float* b3 = nullptr;
  float* wino = nullptr; // Winograd F(2x2,3x3) weight image of a dense 3x3 stride-1 conv (pack_wino), or nullptr
  // head-output layer of a model with mask coefficients (round 4): the same 1x1 conv as TWO weight images -- rows [0, 5+C)
  // (decode fused in the epilogue, no raw rows: the fast wave-autonomous path) and rows [5+C, 5+C+NM) (plain 1x1 whose 32
  // columns land in the level rows the mask kernels read)
  float* wp_det = nullptr; float* b_det = nullptr; float* wp_mc = nullptr; float* b_mc = nullptr;
  int in_h = 0, in_w = 0, out_h = 0, out_w = 0;
  int head_anchor = -1;  // head layers: anchor index handled by this layer
};

struct Slot {
  int h, w, c;
  size_t sz = 0;         // bytes per image
  size_t off = 0;        // byte offset of the slot inside a chunk arena, per image of the chunk (x chunk capacity)
  bool pinned = false;   // whole-batch contiguous allocation of its own (tensors the ABI hands out: mask prototypes)
  float* pin = nullptr;
};

}  // namespace

#define YL_GRAPH_SLOTS 4
#define YL_NMS_GROUPS 4
struct yl_ctx {
  int device = 0;
  int img_size = 0, in_ch = 3, C = 0, L = 0, N = 0, E = 0, NM = 0, proto_slot = -1;
  int level_S[YL_MAX_LEVELS] = {0}, level_A[YL_MAX_LEVELS] = {0}, level_off[YL_MAX_LEVELS + 1] = {0};
  std::vector<Slot> slots;
  std::vector<DevLayer> layers;
  float* zeros = nullptr;                  // 1 KiB of zeros (padding source for the conv kernels)
  int cap_batch = 0;                       // images the pinned slots / level buffers are allocated for (capacity)
  int act_batch = 0;                       // batch of the last forward: what yl_masks* / yl_read_slot may address
  // Activation memory: ONE arena per batch chunk (chunks run concurrently on their own streams), slots placed by
  // liveness -- a slot's bytes are reused by later tensors once its last consumer (launch group) has run.  edge_n,
  // B = 64: 3.8 GB with one buffer per tensor -> a few hundred MB, so that a chunk's producer -> consumer pairs have
  // a chance to meet in the 256 MB Infinity Cache instead of HBM.  "reuse_slots" 0 keeps every tensor (debugging).
  char* arena[4] = {nullptr, nullptr, nullptr, nullptr};
  int arena_capi[4] = {0, 0, 0, 0};        // images arena i is allocated for (a single-chunk call grows arena 0 only)
  float* se_scratch[4] = {nullptr, nullptr, nullptr, nullptr};   // YL_OP_SE partial sums of a chunk: [capi][se_unit]
  size_t se_unit = 0;                      // floats per image: max over the SE layers of P * C
  int plan_n = 0;                          // the current job's batch is split into this many chunks
  int arena_n = 0, arena_cap = 0;          // allocated: arenas, images of the largest one (plan_n / plan_cap <= these)
  int plan_b0[5] = {0, 0, 0, 0, 0};        // chunk i covers images [plan_b0[i], plan_b0[i + 1])
  int plan_cap = 0;                        // images of the largest chunk
  size_t arena_unit = 0;                   // arena bytes per image of a chunk (peak of the live set)
  bool plan_reuse = false;
  int opt_reuse = 1;
  int opt_time_split = 0;    // yl_predict records HIP events around the conv layers and the NMS (one chunk; eager or two hipGraphs)
  hipEvent_t ev_t[3] = {nullptr, nullptr, nullptr};
  bool timing_valid = false;
  int opt_pre_norm = 0;      // yl_preprocess: 0 = tools/infer.py arithmetic, 1 = the evaluate path's A.Normalize
  float* level_buf[YL_MAX_LEVELS] = {nullptr};
  // post-processing workspace
  float4* ws_boxes = nullptr;
  float* ws_scores = nullptr;
  int* ws_cls = nullptr;
  int* ws_clsws = nullptr;
  int* ws_kept_list = nullptr;               // [B][YL_NMS_GROUPS][N] class-group split of the NMS kernel
  int* ws_done = nullptr;                    // [B][256 bytes] class -> NMS workgroup table
  unsigned long long* ws_gkeys = nullptr;
  int gP = 0;
  float* ws_tmp_dets = nullptr;
  int* ws_tmp_idx = nullptr;
  int post_cap_batch = 0;
  // standalone nms scratch
  int* ws_nms_clsws = nullptr;
  unsigned long long* ws_nms_gkeys = nullptr;
  int nms_gP = 0;
  // options
  int opt_graph = 0, opt_tile_m = 0, opt_streams = 2;
  int opt_nms_groups = 0;    // workgroups per image in the NMS kernel: 1..4, 0 = auto (by confidence threshold, see do_post)
  int opt_hybrid = 0;        // (off: measured -0.5 % at B=64) full-batch launches for the high-resolution layers, batch chunks on the internal streams only
                             // for the run of low-resolution (<= 1/16) layers, see plan_segments()
  int small_lo = 0, small_hi = 0;   // that run: layers [small_lo, small_hi)
  int tiny_lo = 0, tiny_hi = 0;     // "hybrid" 2 (the reverse plan): the run of <= 1/32-resolution layers goes out as FULL-batch
                                    // launches between two chunked segments (half as many latency-bound launches)
  int opt_batch_levels = 1;  // runs of independent, identically shaped layers (FPN smooth / head trunk / head out of all
                             // levels) go out as ONE launch (YlConvMulti)
  int opt_winograd = 1;      // dense 3x3 stride-1 layers with >= 64 channels through Winograd F(2x2,3x3) (2.25x fewer MACs;
                             // NOT bit-identical to the direct convolution: fp32 rounding of the transforms); 2 = only the
                             // >= 64-channel layers on the LARGEST grid they occur on (the finest pyramid level's smooth block).
                             // 1 = every eligible layer (>= 16 channels in and out) -- the DEFAULT since round 4: measured score
                             // error vs the oracle over 4 weight seeds x 32 images x 8400 candidates 0.9-1.6e-5 (yololite_m) /
                             // 0.7-1.2e-5 (yololite_m v2), no worse than the direct convolution's 1.4-1.7e-5 / 1.0-1.3e-5
                             // (profiles/r04_winograd_margin*.json)
  int wino_max_hw = 0;       // that grid: max out_h * out_w over the layers that carry a Winograd weight image
  int opt_fuse_decode = 1;   // yl_predict: decode in the head-output conv's epilogue (no raw level tensor, no decode kernel)
  int opt_fuse_head = 1;     // ... and the head trunk (depthwise 3x3 -> 1x1) in the same launch (yl_conv_dpp_kernel)
  int opt_split_k = 0; // depthwise -> 1x1 layers on <= 20x20 grids in the split-K form (yl_conv_dwt_kernel<.., SK = 4>): -9 % batch-1
                      // latency, -0.7 % throughput at B = 64 (measured, edge_n) -> off by default, the pip API turns it on
  int opt_dev = 0;    // developer kernel-selection word (YL_DEV_*, "dev_select"); rides in every YlConvP
  int opt_bf16 = 0;   // reduced-precision MFMA mode: 1 = conv / stem-block launches use the bf16-MFMA builds, 2 = the fp16-MFMA
                      // builds (fp32 storage, fp32 accumulate either way); 0 = fp32 (the parity path)
  // batch chunks run on `opt_streams` internal streams (fork/join around every call): the
  // latency-bound low-resolution layers of one chunk overlap the bandwidth-bound layers of another
  hipStream_t work[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_fork = nullptr, ev_join[4] = {nullptr, nullptr, nullptr, nullptr};
  // branch lanes inside a chunk: neck/head layers that only feed the heads of the coarser levels (smooth5,
  // head5, smooth4, head4, ...) run on a side stream next to the 80x80 chain (lateral3 -> smooth3 -> head3);
  // lane[i] is derived from the slot graph at yl_create.  Off by default ("lanes" option): measured no gain
  // (29.3 k img/s either way) -- the conv grids are occupancy-sized persistent grids, so a second kernel only
  // gets CUs at the tail of the first
  int opt_lanes = 0;
  std::vector<unsigned char> lane;
  hipStream_t side[4] = {nullptr, nullptr, nullptr, nullptr};
  hipEvent_t ev_la[4] = {nullptr, nullptr, nullptr, nullptr}, ev_lb[4] = {nullptr, nullptr, nullptr, nullptr};
  // single-entry hipGraph cache keyed on everything baked into the captured launches
  // hipGraph cache: up to YL_GRAPH_SLOTS jobs (a serving loop alternates between a few input / output buffers,
  // e.g. the two slots of the pipelined all-gather), least recently used evicted
  struct GraphEntry {
    std::vector<hipGraphExec_t> execs;   // one graph per (segment, chunk) piece, in launch order
    int n = 0;
    std::vector<unsigned char> key;
    unsigned long long stamp = 0;
  };
  std::vector<GraphEntry> graphs;
  unsigned long long graph_clock = 0;
  // The packed weights (every device pointer inside `layers`, and `zeros`) are immutable after yl_create and may be
  // SHARED by several contexts (yl_clone: one context per batch in flight of a serving pipeline); the last owner frees them
  std::shared_ptr<int> weights_owner;
  std::string err;
};

namespace {

// function attributes (dynamic-LDS opt-ins) are per DEVICE: one flag per device ordinal
bool g_inited[64] = {false};

yl_status fail(yl_ctx* c, yl_status s, const std::string& msg) {
  if (c) c->err = msg;
  return s;
}

#define HIPCHK(ctx, expr)                                                                        \
  do {                                                                                           \
    hipError_t _e = (expr);                                                                      \
    if (_e != hipSuccess) {                                                                      \
      char _b[512];                                                                              \
      snprintf(_b, sizeof(_b), "%s failed: %s (%s:%d)", #expr, hipGetErrorString(_e), __FILE__,  \
               __LINE__);                                                                        \
      return fail(ctx, (_e == hipErrorOutOfMemory) ? YL_ERR_NOMEM : YL_ERR_HIP, _b);             \
    }                                                                                            \
  } while (0)

template <typename T>
yl_status upload(yl_ctx* c, const std::vector<T>& h, T** d) {
  *d = nullptr;
  if (h.empty()) return YL_OK;
  HIPCHK(c, hipMalloc((void**)d, h.size() * sizeof(T)));
  HIPCHK(c, hipMemcpy(*d, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice));
  return YL_OK;
}

inline int cdiv(int a, int b) { return (a + b - 1) / b; }

// MFMA fragment order: [tap][kblock][ntile][lane][s]  with
//   n = ntile*16 + (lane & 15),  c = kblock*16 + 4*(lane >> 4) + s     (see yl_conv.hip)
void pack_conv(const float* w, int cout, int cin, int k, std::vector<float>& out) {
  const int KB = cdiv(cin, 16), NT = cdiv(cout, 16), taps = k * k;
  out.assign((size_t)taps * KB * NT * 256, 0.0f);
  for (int tap = 0; tap < taps; ++tap)
    for (int kb = 0; kb < KB; ++kb)
      for (int nt = 0; nt < NT; ++nt)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int n = nt * 16 + (lane & 15);
            const int c = kb * 16 + 4 * (lane >> 4) + s;
            if (n < cout && c < cin)
              out[((((size_t)tap * KB + kb) * NT + nt) * 64 + lane) * 4 + s] =
                  w[((size_t)n * cin + c) * taps + tap];
          }
}

// Winograd F(2x2,3x3): U = G g G^T (4x4 per (cout, cin)), G = [[1,0,0],[.5,.5,.5],[.5,-.5,.5],[0,0,1]], in the MFMA
// fragment order of pack_conv per transform position xi = 4i + j, grouped so that one (n-group of 2 n-tiles, k-block)
// chunk is 32 KiB contiguous: [ngroup][kblock][xi][nt 0..1][lane][s]   (yl_conv_wino_kernel)
void pack_wino(const float* w, int cout, int cin, std::vector<float>& out) {
  const int KB = cdiv(cin, 16), NG = cdiv(cdiv(cout, 16), 2);
  static const float G[4][3] = {{1.f, 0.f, 0.f}, {0.5f, 0.5f, 0.5f}, {0.5f, -0.5f, 0.5f}, {0.f, 0.f, 1.f}};
  out.assign((size_t)NG * KB * 16 * 2 * 256, 0.0f);
  for (int ng = 0; ng < NG; ++ng)
    for (int kb = 0; kb < KB; ++kb)
      for (int t = 0; t < 2; ++t)
        for (int lane = 0; lane < 64; ++lane)
          for (int s = 0; s < 4; ++s) {
            const int n = (ng * 2 + t) * 16 + (lane & 15);
            const int c = kb * 16 + 4 * (lane >> 4) + s;
            if (n >= cout || c >= cin) continue;
            const float* g = w + ((size_t)n * cin + c) * 9;
            float tmp[4][3];
            for (int i = 0; i < 4; ++i)
              for (int b = 0; b < 3; ++b) tmp[i][b] = G[i][0] * g[0 * 3 + b] + G[i][1] * g[1 * 3 + b] + G[i][2] * g[2 * 3 + b];
            for (int i = 0; i < 4; ++i)
              for (int j = 0; j < 4; ++j) {
                const float u = tmp[i][0] * G[j][0] + tmp[i][1] * G[j][1] + tmp[i][2] * G[j][2];
                out[(((((size_t)ng * KB + kb) * 16 + (i * 4 + j)) * 2 + t) * 64 + lane) * 4 + s] = u;
              }
          }
}

// depthwise [c][1][k][k] -> [tap][c]
void pack_dw(const float* w, int ch, int k, std::vector<float>& out) {
  out.assign((size_t)k * k * ch, 0.0f);
  for (int c = 0; c < ch; ++c)
    for (int t = 0; t < k * k; ++t) out[(size_t)t * ch + c] = w[(size_t)c * k * k + t];
}

// stem [cout][3][3][3] -> MFMA A fragments [kstep(7)][ntile][lane]: n = ntile*16 + (lane&15),
// k = 4*kstep + (lane>>4) with k = c*9 + ky*3 + kx (PyTorch OIHW flattening), zero for k >= 27
void pack_stem(const float* w, int cout, int cin, int k, std::vector<float>& out) {
  const int K = cin * k * k, KS = cdiv(K, 4), NT = cdiv(cout, 16);
  out.assign((size_t)KS * NT * 64, 0.0f);
  for (int s = 0; s < KS; ++s)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = nt * 16 + (lane & 15), kk = 4 * s + (lane >> 4);
        if (n < cout && kk < K) out[((size_t)s * NT + nt) * 64 + lane] = w[(size_t)n * K + kk];
      }
}

// stem of the fused entry block (3 input channels, 3x3): K order by input rows, see yl_stemblock.hip
// bias (may be null): rides in the K = 27 -> 28 pad slot (s = 6, lane group 3) -- the kernel feeds 1.0 there, so the
// shift is the LAST product of every output's fma chain (the rounding of conv + shift)
void pack_stem_rows(const float* w, const float* bias, int cout, std::vector<float>& out) {
  const int KS = 7, NT = cdiv(cout, 16);
  out.assign((size_t)KS * NT * 64, 0.0f);
  for (int s = 0; s < KS; ++s)
    for (int nt = 0; nt < NT; ++nt)
      for (int lane = 0; lane < 64; ++lane) {
        const int n = nt * 16 + (lane & 15), kq = lane >> 4;
        int row, kx;
        if (s < 3) { row = 2 * kq; kx = s; }
        else if (s < 6) { row = 2 * kq + 1; kx = s - 3; }
        else if (kq < 3) { row = 8; kx = kq; }
        else {                                                       // pad slot: the bias
          if (n < cout && bias) out[((size_t)s * NT + nt) * 64 + lane] = bias[n];
          continue;
        }
        if (n < cout) out[((size_t)s * NT + nt) * 64 + lane] = w[(size_t)n * 27 + row * 3 + kx];
      }
}

void drop_graph(yl_ctx* c);

void free_post_ws(yl_ctx* c) {
  drop_graph(c);                        // cached graphs have the workspace pointers baked in
  hipFree(c->ws_boxes); hipFree(c->ws_scores); hipFree(c->ws_cls); hipFree(c->ws_clsws);
  hipFree(c->ws_kept_list); hipFree(c->ws_done);
  c->ws_kept_list = nullptr; c->ws_done = nullptr;
  hipFree(c->ws_gkeys); hipFree(c->ws_tmp_dets); hipFree(c->ws_tmp_idx);
  c->ws_boxes = nullptr; c->ws_scores = nullptr; c->ws_cls = nullptr; c->ws_clsws = nullptr;
  c->ws_gkeys = nullptr; c->ws_tmp_dets = nullptr; c->ws_tmp_idx = nullptr;
  c->post_cap_batch = 0;
}

void drop_graph(yl_ctx* c) {
  for (auto& g : c->graphs)
    for (auto e : g.execs)
      if (e) hipGraphExecDestroy(e);
  c->graphs.clear();
}

void free_act(yl_ctx* c) {
  drop_graph(c);
  for (int i = 0; i < 4; ++i) {
    hipFree(c->arena[i]); c->arena[i] = nullptr; c->arena_capi[i] = 0;
    hipFree(c->se_scratch[i]); c->se_scratch[i] = nullptr;
  }
  for (auto& s : c->slots) { hipFree(s.pin); s.pin = nullptr; }
  for (int l = 0; l < YL_MAX_LEVELS; ++l) { hipFree(c->level_buf[l]); c->level_buf[l] = nullptr; }
  c->cap_batch = 0; c->plan_n = 0; c->arena_n = 0; c->arena_cap = 0; c->act_batch = 0;
}

int pow2ceil(int v) { int p = 64; while (p < v) p <<= 1; return p; }

// layers [i, end) that run_layers sends out as ONE launch (level-batched heads / smooth blocks): same kernel
// configuration, no residual / upsample operands, no dependency inside the run
size_t layer_group_end(const yl_ctx* c, size_t i, size_t lend) {
  auto same_shape = [&](size_t x, size_t y) {
    const yl_layer& a = c->layers[x].d; const yl_layer& e = c->layers[y].d;
    return a.op == YL_OP_CONV && e.op == YL_OP_CONV && a.cin == e.cin && a.cout == e.cout && a.k == e.k &&
           a.stride == e.stride && a.pad_t == e.pad_t && a.pad_l == e.pad_l && a.act == e.act &&
           !YL_ACT_POSTPASS(a.act) && a.in_shift == e.in_shift && a.dw_k == e.dw_k && a.dw_stride == e.dw_stride && a.dw_pad_t == e.dw_pad_t &&
           a.dw_pad_l == e.dw_pad_l && a.dw_act == e.dw_act && a.c2 == 0 && e.c2 == 0 && a.scale_slot < 0 && e.scale_slot < 0 && a.res_slot < 0 &&
           e.res_slot < 0 && a.up_slot < 0 && e.up_slot < 0 && (a.head_level >= 0) == (e.head_level >= 0) &&
           (a.cout + 15) / 16 <= 8;
  };
  size_t gend = i + 1;
  if (c->layers[i].d.op != YL_OP_CONV) return gend;
  while (gend < lend && gend - i < 4 && same_shape(i, gend)) {
    bool dep = false;
    for (size_t q = i; q < gend; ++q)
      if (c->layers[q].d.head_level < 0 && c->layers[q].d.out_slot == c->layers[gend].d.in_slot) dep = true;
    if (dep) break;
    ++gend;
  }
  return gend;
}

// Place the slots inside a chunk arena.  Liveness is tracked per LAUNCH GROUP (a level-batched run reads and writes
// all of its layers' tensors at once): slot s is live from the group that produces it to the group of its last
// consumer, inclusive; first-fit over the gaps of the live set.  reuse == false: every slot gets its own range.
bool pair_fusable(const yl_ctx* c, size_t i, size_t lend, bool ignore_options);

void plan_slots(yl_ctx* c, bool reuse) {
  const size_t NL = c->layers.size(), NS = c->slots.size();
  std::vector<int> grp(NL, 0);
  int g = 0;
  for (size_t i = 0; i < NL; ++g) {
    size_t e = layer_group_end(c, i, NL);
    // two layers that run_layers may send out as ONE launch (yl_conv_dpq_kernel) are one group: the second layer's
    // output is written while the first layer's inputs are still being read by other tiles, so it must not be placed
    // over them (whatever the options say when the plan is made)
    // (also when the first layer of the pair is the LAST member of a level-batched run -- with "batch_levels"
    // 0 run_layers sees it alone and may fuse it with its successor)
    if (pair_fusable(c, e - 1, NL, true)) e = e + 1;
    for (size_t q = i; q < e; ++q) grp[q] = g;
    i = e;
  }
  const int INF = 1 << 30;
  std::vector<int> def(NS, INF), last(NS, -1);
  for (size_t i = 0; i < NL; ++i) {
    const yl_layer& d = c->layers[i].d;
    if (d.head_level < 0 && d.out_slot >= 0 && grp[i] < def[d.out_slot]) def[d.out_slot] = grp[i];
    const int ins[4] = {(d.op == YL_OP_STEM || d.op == YL_OP_STEMBLOCK || d.op == YL_OP_NHWC4) ? -1 : d.in_slot, d.res_slot, d.up_slot, d.scale_slot};
    for (int k = 0; k < 4; ++k)
      if (ins[k] >= 0 && grp[i] > last[ins[k]]) last[ins[k]] = grp[i];
  }
  for (size_t s = 0; s < NS; ++s) {
    Slot& t = c->slots[s];
    t.pinned = (int)s == c->proto_slot;
    // fp16-storage mode ("store_f16"): activation tensors are fp16; the pinned prototypes (the mask kernels read them) and the
    // squeeze-excite gates (yl_se.hip writes them, the gated 1x1 convs read them: fp32 [B][C] vectors) stay fp32
    bool f32 = c->opt_bf16 != 3 || t.pinned;
    for (size_t i = 0; i < NL && !f32; ++i) f32 = c->layers[i].d.op == YL_OP_SE && c->layers[i].d.out_slot == (int)s;
    t.sz = (size_t)t.h * t.w * t.c * (f32 ? sizeof(float) : sizeof(_Float16));   // images of a slot are contiguous (a multiple of 8 B)
    if (last[s] < def[s]) last[s] = def[s] == INF ? -1 : def[s];     // produced, never consumed: live in its own group
  }
  struct Iv { size_t off, sz; int last; };
  std::vector<Iv> live;
  size_t peak = 0;
  std::vector<size_t> order;
  for (size_t s = 0; s < NS; ++s) if (!c->slots[s].pinned && def[s] != INF) order.push_back(s);
  std::stable_sort(order.begin(), order.end(), [&](size_t a, size_t b) { return def[a] < def[b]; });
  size_t bump = 0;
  for (size_t s : order) {
    Slot& t = c->slots[s];
    const size_t need = (t.sz + 255) & ~(size_t)255;                  // placement granule: slot bases stay 256-B aligned
    if (!reuse) { t.off = bump; bump += need; peak = bump; continue; }
    live.erase(std::remove_if(live.begin(), live.end(), [&](const Iv& v) { return v.last < def[s]; }), live.end());
    std::sort(live.begin(), live.end(), [](const Iv& a, const Iv& b) { return a.off < b.off; });
    size_t at = 0;
    for (const Iv& v : live) {
      if (at + need <= v.off) break;
      if (v.off + v.sz > at) at = v.off + v.sz;
    }
    t.off = at;
    live.push_back({at, need, last[s]});
    if (at + need > peak) peak = at + need;
  }
  c->arena_unit = peak;
}

// number of batch chunks a job of B images is split into (submit / walk_plan)
int chunks_for(const yl_ctx* c, int B) {
  if (c->opt_time_split) return 1;          // the infer / post split is defined on ONE stream
  int n = c->opt_streams < 1 ? 1 : (c->opt_streams > 4 ? 4 : c->opt_streams);
  if (B < 4 * n) n = 1;
  return n;
}

// activations for a batch of B images processed as n chunks (the split walk_plan uses), one arena per chunk.
// The hybrid plan mixes full-batch and chunked segments and the side-lane option runs layers of one chunk
// concurrently: both get ONE arena with every tensor kept (any image range addressable, nothing reused).
yl_status ensure_act(yl_ctx* c, int B, int n) {
  const bool flat = c->opt_hybrid || c->opt_lanes;
  if (flat) n = 1;
  const bool reuse = c->opt_reuse && !flat;
  // the chunk split of THIS job
  int b0s[5] = {0, 0, 0, 0, 0}, cap = 0;
  {
    const int base = B / n, rem = B % n;
    int b0 = 0;
    for (int i = 0; i < n; ++i) {
      const int bn = base + (i < rem ? 1 : 0);
      b0s[i] = b0;
      b0 += bn;
      if (bn > cap) cap = bn;
    }
    b0s[n] = B;
  }
  // CAPACITY only grows: a smaller batch (a tail batch, alternating B = 4 / 16, a "time_split" toggle) re-plans its
  // chunk ranges inside the existing allocations -- no hipFree (a device sync), no dropped hipGraphs (their keys carry
  // the batch, so graphs of several batch sizes live side by side in the LRU), and the level buffers / prototypes of
  // the previous batch stay where yl_masks* reads them.  Slot addresses inside an arena scale with plan_cap
  // (slot_addr), i.e. a job's addresses are a function of (B, n) as long as the allocation does not move.
  // Per-arena capacity: a single-chunk call (time_split, per-layer timing) after a two-chunk run grows
  // arena 0 to the full batch and leaves the others at the chunk size -- not every arena at the full batch.
  bool grow = B > c->cap_batch || n > c->arena_n || reuse != c->plan_reuse || c->arena_n == 0;
  for (int i = 0; i < n && !grow; ++i) grow = cap > c->arena_capi[i];
  if (grow) {
    const int newB = B > c->cap_batch ? B : c->cap_batch, newn = n > c->arena_n ? n : c->arena_n;
    int capi[4], newcap = 0;
    for (int i = 0; i < 4; ++i) {
      capi[i] = i < newn ? c->arena_capi[i] : 0;
      if (i < n && cap > capi[i]) capi[i] = cap;
      if (capi[i] > newcap) newcap = capi[i];
    }
    free_act(c);                                             // also drops the cached graphs (addresses change)
    c->plan_reuse = reuse;
    plan_slots(c, reuse);
    for (int i = 0; i < newn; ++i) {
      // + 256 spare bytes, and the arena starts out as zeros: the kernels that read through buffer descriptors do not mask the
      // channel tail of a tensor's last k-block -- those lanes read up to 48 bytes behind the pixel (the next pixel, the next slot,
      // at the very end the spare bytes) and meet zero weights; 0 x garbage must not be 0 x NaN on the first run after an allocation
      if (c->arena_unit) {
        HIPCHK(c, hipMalloc((void**)&c->arena[i], c->arena_unit * (size_t)capi[i] + 256));
        HIPCHK(c, hipMemset(c->arena[i], 0, c->arena_unit * (size_t)capi[i] + 256));
      }
      if (c->se_unit) HIPCHK(c, hipMalloc((void**)&c->se_scratch[i], c->se_unit * sizeof(float) * (size_t)capi[i]));
      c->arena_capi[i] = capi[i];
    }
    for (auto& s : c->slots)
      if (s.pinned) {
        HIPCHK(c, hipMalloc((void**)&s.pin, s.sz * (size_t)newB + 256));
        HIPCHK(c, hipMemset(s.pin, 0, s.sz * (size_t)newB + 256));
      }
    for (int l = 0; l < c->L; ++l)
      HIPCHK(c, hipMalloc((void**)&c->level_buf[l],
                          (size_t)newB * c->level_A[l] * c->level_S[l] * c->level_S[l] * c->E * sizeof(float)));
    c->cap_batch = newB; c->arena_n = newn; c->arena_cap = newcap;
  }
  c->plan_n = n; c->plan_cap = cap;
  for (int i = 0; i <= n; ++i) c->plan_b0[i] = b0s[i];
  c->act_batch = B;
  return YL_OK;
}

// image b of slot `sl` (b must lie in the planned chunk that contains it)
float* slot_addr(const yl_ctx* c, int sl, int b) {
  const Slot& t = c->slots[sl];
  if (t.pinned) return (float*)((char*)t.pin + (size_t)b * t.sz);
  int ch = 0;
  while (ch + 1 < c->plan_n && b >= c->plan_b0[ch + 1]) ++ch;
  return (float*)(c->arena[ch] + t.off * (size_t)c->plan_cap + (size_t)(b - c->plan_b0[ch]) * t.sz);
}

yl_status ensure_post(yl_ctx* c, int B) {
  if (B <= c->post_cap_batch) return YL_OK;
  free_post_ws(c);
  const size_t n = (size_t)B * c->N;
  HIPCHK(c, hipMalloc((void**)&c->ws_boxes, n * sizeof(float4)));
  HIPCHK(c, hipMalloc((void**)&c->ws_scores, n * sizeof(float)));
  HIPCHK(c, hipMalloc((void**)&c->ws_cls, n * sizeof(int)));
  const int Cw = c->C > 0 ? c->C : 1;
  HIPCHK(c, hipMalloc((void**)&c->ws_clsws, (size_t)B * 4 * Cw * sizeof(int)));
  c->gP = pow2ceil(c->N);
  if (c->gP > YL_LDS_KEYS_MAX) HIPCHK(c, hipMalloc((void**)&c->ws_gkeys, (size_t)B * c->gP * 8));
  HIPCHK(c, hipMalloc((void**)&c->ws_tmp_dets, n * 6 * sizeof(float)));
  HIPCHK(c, hipMalloc((void**)&c->ws_tmp_idx, n * sizeof(int)));
  HIPCHK(c, hipMalloc((void**)&c->ws_kept_list, n * YL_NMS_GROUPS * sizeof(int)));
  HIPCHK(c, hipMalloc((void**)&c->ws_done, (size_t)B * 256));        // class -> NMS workgroup table per image
  c->post_cap_batch = B;
  return YL_OK;
}

void fill_levels(const yl_ctx* c, const float* const* ptrs, YlLevels& lv) {
  memset(&lv, 0, sizeof(lv));
  lv.L = c->L; lv.N = c->N; lv.E = c->E; lv.C = c->C;
  lv.hi = (float)(c->img_size - 1);
  for (int l = 0; l < c->L; ++l) {
    lv.ptr[l] = ptrs[l];
    lv.S[l] = c->level_S[l];
    lv.A[l] = c->level_A[l];
    lv.off[l] = c->level_off[l];
    lv.stride[l] = (float)((double)c->img_size / (double)c->level_S[l]);   // utils_ms.py:71
  }
  lv.off[c->L] = c->N;
}

// builds the kernel parameter block of layer i for batch B
void layer_params(const yl_ctx* c, const DevLayer& L, int b0, int B, const float* x, float* const* level_out,
                  YlConvP& p) {
  memset(&p, 0, sizeof(p));
  const yl_layer& d = L.d;
  p.wp = L.wp; p.bias = L.bias; p.dw_w = L.dw_w; p.dw_b = L.dw_b;
  // fp32 only: the committed Winograd error margins (profiles/r04_winograd_margin*.json) were measured with fp32 operands; with
  // 16-bit operands the transformed inputs and U = G g G^T would be rounded before the multiply
  p.wino = !c->opt_bf16 && (c->opt_winograd == 1 ||
            (c->opt_winograd == 2 && d.cin >= 64 && d.cout >= 64 && L.out_h * L.out_w >= c->wino_max_hw)) ? L.wino : nullptr;
  p.zeros = c->zeros;
  p.B = B; p.H = L.in_h; p.W = L.in_w; p.Cin = d.cin;
  p.OH = L.out_h; p.OW = L.out_w; p.N = d.cout;
  p.k = d.k; p.stride = d.stride; p.pad_t = d.pad_t; p.pad_l = d.pad_l; p.act = d.act;
  p.in_shift = d.in_shift;
  p.dw_k = d.dw_k; p.dw_stride = d.dw_stride; p.dw_pad_t = d.dw_pad_t; p.dw_pad_l = d.dw_pad_l; p.dw_act = d.dw_act;
  p.MH = L.out_h; p.MW = L.out_w;
  p.KB = cdiv(d.cin, 16);
  p.TK = d.k * d.k * p.KB;
  p.NTtot = cdiv(d.cout, 16);
  p.M = B * L.out_h * L.out_w;
  auto slot_ptr = [&](int sl) { return slot_addr(c, sl, b0); };
  p.x = (d.op == YL_OP_STEM || d.op == YL_OP_STEMBLOCK || d.op == YL_OP_NHWC4) ? x + (size_t)b0 * 3 * L.in_h * L.in_w : slot_ptr(d.in_slot);
  if (d.op == YL_OP_CONV && d.c2 > 0) {      // fused expand -> depthwise -> project
    p.w2p = L.w2p; p.b2 = L.b2; p.C1 = d.c2; p.act2 = d.act2;
  }
  if (d.op == YL_OP_CONV && d.c3 > 0) {      // dense k x k conv with a chained 1x1 conv (yl_conv_mfma_kernel epilogue)
    p.w3p = L.w3p; p.b3 = L.b3; p.C3 = d.c3; p.act3 = d.act3;
  }
  if (d.op == YL_OP_STEMBLOCK) {
    p.w2p = L.w2p; p.b2 = L.b2; p.w3p = L.w3p; p.b3 = L.b3;
    p.C1 = d.cout; p.C2 = d.c2; p.C3 = d.c3; p.act2 = d.act2; p.act3 = d.act3;
    p.SH = p.SW = (L.in_h + d.pad_t + (d.k - 1 - d.pad_t) - d.k) / d.stride + 1;
    p.tiles_x = cdiv(L.out_w, 8); p.tiles_y = cdiv(L.out_h, 8);
    p.ntiles = B * p.tiles_x * p.tiles_y;
    p.N = d.c3 > 0 ? d.c3 : d.c2;
  }
  if (d.res_slot >= 0) p.res = slot_ptr(d.res_slot);
  if (YL_ACT_POSTPASS(d.act)) { p.act = YL_ACT_NONE; p.res = nullptr; }     // applied by the activation pass (run_layers)
  if (d.scale_slot >= 0) p.scale = slot_ptr(d.scale_slot);
  p.dev = (unsigned)c->opt_dev | (c->opt_split_k ? 0u : YL_DEV_DWT_NOSPLIT);
  if (d.up_slot >= 0) {
    p.up = slot_ptr(d.up_slot);
    p.UH = c->slots[d.up_slot].h; p.UW = c->slots[d.up_slot].w;
  }
  p.out_f32 = (c->opt_bf16 == 3 && (d.head_level >= 0 || d.out_slot == c->proto_slot)) ? 1 : 0;
  if (d.head_level >= 0) {
    const int l = d.head_level;
    const int ss = c->level_S[l] * c->level_S[l];
    p.out_bstride = (long)c->level_A[l] * ss * c->E;
    p.out = level_out[l] + (size_t)b0 * p.out_bstride + (size_t)L.head_anchor * ss * c->E;
  } else {
    p.out = slot_ptr(d.out_slot);
    p.out_bstride = (long)L.out_h * L.out_w * ((d.op == YL_OP_CONV && d.c3 > 0) ? d.c3 : d.cout);
  }
}

// decode can be fused into the head-output convs when every one of them is a plain 1x1 conv whose 5+C(+masks)
// columns fit one block's n-tiles (<= 128)
bool can_fuse_decode(const yl_ctx* c) {
  if (!c->opt_fuse_decode) return false;
  bool any = false;
  for (const auto& L : c->layers) {
    if (L.d.head_level < 0) continue;
    if (L.d.op != YL_OP_CONV || L.d.k != 1 || L.d.dw_k > 0 || L.d.c2 > 0 || L.d.cout > 128 || L.d.cout != c->E ||
        L.d.act != YL_ACT_NONE || L.d.res_slot >= 0 || L.d.up_slot >= 0)
      return false;
    any = true;
  }
  return any;
}

// Head branches as ONE launch (yl_conv_dpp_kernel): the level-batched run [i, gend) of head trunks (depthwise 3x3 -> 1x1)
// is followed by the run of their head-output convs in the same order, nothing else reads the trunk tensors, decode is
// fused (yl_predict, no mask coefficients) and the shape is instantiated
bool head_run_fusable(const yl_ctx* c, size_t i, size_t gend, size_t lend) {
  const size_t n = gend - i;
  if (!c->opt_fuse_head || c->opt_bf16 || c->NM > 0 || gend + n > lend) return false;
  for (size_t q = 0; q < n; ++q) {
    const DevLayer& T = c->layers[i + q];
    const DevLayer& O = c->layers[gend + q];
    const yl_layer& t = T.d; const yl_layer& o = O.d;
    if (t.op != YL_OP_CONV || t.k != 1 || t.dw_k != 3 || t.dw_stride != 1 || t.c2 > 0 || t.c3 > 0 || t.head_level >= 0 ||
        t.res_slot >= 0 || t.up_slot >= 0 || t.in_shift || YL_SMOOTH(t.act) || YL_SMOOTH(t.dw_act) || t.out_slot < 0)
      return false;
    if (o.op != YL_OP_CONV || o.head_level < 0 || o.k != 1 || o.dw_k > 0 || o.c2 > 0 || o.c3 > 0 || o.in_slot != t.out_slot ||
        o.cin != t.cout || o.act != YL_ACT_NONE || o.res_slot >= 0 || o.up_slot >= 0 || o.in_shift)
      return false;
    if (T.in_h != T.out_h || T.in_w != T.out_w || !yl_dpp_supported(t.cin, t.cout, o.cout, T.out_h, T.out_w)) return false;
    for (size_t r = 0; r < c->layers.size(); ++r) {
      if (r == gend + q) continue;
      const yl_layer& e = c->layers[r].d;
      const bool reads_in = e.op != YL_OP_STEM && e.op != YL_OP_STEMBLOCK && e.in_slot == t.out_slot;
      if (reads_in || e.res_slot == t.out_slot || e.up_slot == t.out_slot) return false;
    }
  }
  return true;
}

// depthwise 3x3 -> 1x1 expand -> 1x1 project (+residual) as ONE launch (yl_conv_dpq_kernel): layer i is the depthwise +
// expand conv, layer i + 1 the plain 1x1 that consumes it, nothing else reads the expanded tensor, shape instantiated
bool pair_fusable(const yl_ctx* c, size_t i, size_t lend, bool ignore_options) {
  if ((!ignore_options && (!c->opt_fuse_head || c->opt_bf16)) || i + 1 >= lend) return false;
  const DevLayer& T = c->layers[i];
  const yl_layer& t = T.d; const yl_layer& o = c->layers[i + 1].d;
  if (t.op != YL_OP_CONV || t.k != 1 || t.dw_k != 3 || t.dw_stride != 1 || t.c2 > 0 || t.c3 > 0 || t.head_level >= 0 ||
      t.res_slot >= 0 || t.up_slot >= 0 || t.in_shift || YL_SMOOTH(t.act) || YL_SMOOTH(t.dw_act) || t.out_slot < 0)
    return false;
  if (o.op != YL_OP_CONV || o.head_level >= 0 || o.k != 1 || o.dw_k > 0 || o.c2 > 0 || o.c3 > 0 || o.in_slot != t.out_slot ||
      o.cin != t.cout || YL_SMOOTH(o.act) || o.up_slot >= 0 || o.in_shift || o.res_slot == t.out_slot ||
      o.scale_slot >= 0 || t.scale_slot >= 0)
    return false;
  if (T.in_h != T.out_h || T.in_w != T.out_w || !yl_dpq_supported(t.cin, t.cout, o.cout, T.out_h, T.out_w)) return false;
  for (size_t r = 0; r < c->layers.size(); ++r) {
    if (r == i + 1) continue;
    const yl_layer& e = c->layers[r].d;
    const bool reads_in = e.op != YL_OP_STEM && e.op != YL_OP_STEMBLOCK && e.in_slot == t.out_slot;
    if (reads_in || e.res_slot == t.out_slot || e.up_slot == t.out_slot) return false;
  }
  return true;
}

hipError_t conv_multi(const yl_ctx* c, const YlConvP* ps, int n, hipStream_t st) {
  switch (c->opt_bf16) {
    case 1: return yl_launch_conv_multi_bf16(ps, n, c->opt_tile_m, st);
    case 2: return yl_launch_conv_multi_f16(ps, n, c->opt_tile_m, st);
    case 3: return yl_launch_conv_multi_f16s(ps, n, c->opt_tile_m, st);
    default: return yl_launch_conv_multi(ps, n, c->opt_tile_m, st);
  }
}

yl_status run_layers(yl_ctx* c, const float* x, int b0, int B, float* const* level_out, hipStream_t st,
                     hipEvent_t* evs /*nullable: num_layers+1 events*/, int chunk = 0,
                     const yl_post_cfg* fuse = nullptr /*non-null: head outputs decode in their epilogue*/,
                     int lo = 0, int hi = -1 /*layer range [lo, hi), -1 = to the end*/) {
  if (evs) HIPCHK(c, hipEventRecord(evs[0], st));
  // two lanes (see yl_ctx::lane): lane-1 layers go to the chunk's side stream; an event edge is inserted
  // wherever a layer reads a slot produced on the other lane, and the side stream is joined at the end.
  // Per-layer timing (evs) keeps everything on one stream.
  const bool lanes = !evs && c->opt_lanes && !c->lane.empty() && chunk >= 0 && chunk < 4;
  hipStream_t sd = nullptr;
  if (lanes) {
    if (!c->side[chunk]) HIPCHK(c, hipStreamCreateWithFlags(&c->side[chunk], hipStreamNonBlocking));
    if (!c->ev_la[chunk]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_la[chunk], hipEventDisableTiming));
    if (!c->ev_lb[chunk]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_lb[chunk], hipEventDisableTiming));
    sd = c->side[chunk];
  }
  std::vector<unsigned char> prod(c->slots.size(), 0);     // lane that produced each slot
  bool side_used = false;
  int pooled_slot = -1, pooled_P = 0;                      // slot whose partial channel sums the last depthwise launch left
  auto params = [&](size_t i, YlConvP& p) {
    layer_params(c, c->layers[i], b0, B, x, level_out, p);
    const yl_layer& d = c->layers[i].d;
    if (fuse && d.head_level >= 0) {
      const int l = d.head_level, S = c->level_S[l];
      const size_t o = (size_t)b0 * c->N;
      p.dec_boxes = c->ws_boxes + o; p.dec_scores = c->ws_scores + o; p.dec_cls = c->ws_cls + o;
      p.dec_N = c->N; p.dec_off = c->level_off[l] + c->layers[i].head_anchor * S * S; p.dec_C = c->C;
      p.dec_mode = fuse->mode; p.dec_center = fuse->center_mode; p.dec_wh = fuse->wh_mode;
      p.dec_raw = c->NM > 0 ? 1 : 0;                          // mask coefficients are read from the raw rows
      p.dec_stride = (float)((double)c->img_size / (double)S);   // utils_ms.py:71, as fill_levels
      p.dec_hi = (float)(c->img_size - 1);
    }
  };
  const size_t lend = hi < 0 ? c->layers.size() : (size_t)hi;
  for (size_t i = (size_t)lo; i < lend;) {
    const yl_layer& d = c->layers[i].d;
#ifdef YL_VARIANT_SKIP_LAYERS
    // VARIANT BUILDS ONLY (tools/build_variant.sh ... -DYL_VARIANT_SKIP_LAYERS; never in libyololite_hip.so): leave out the
    // layers YL_SKIP="lo-hi" -- results are WRONG; answers "what would the step be if these launches were free"
    {
      static int slo = -2, shi = -2;
      if (slo == -2) { slo = -1; const char* e = getenv("YL_SKIP"); if (e) sscanf(e, "%d-%d", &slo, &shi); }
      if ((int)i >= slo && (int)i <= shi) { if (evs) hipEventRecord(evs[i + 1], st); ++i; continue; }
    }
#endif
    // ---- level-batched run starting at i (not under per-layer timing, not with side lanes)
    size_t gend = i + 1;
    if (!evs && !lanes && c->opt_batch_levels && d.op == YL_OP_CONV) gend = layer_group_end(c, i, lend);
    if (fuse && !evs && !lanes && d.op == YL_OP_CONV && d.dw_k == 3 && head_run_fusable(c, i, gend, lend)) {
      YlConvP ps[4];
      const size_t n = gend - i;
      for (size_t q = 0; q < n; ++q) {
        YlConvP o;
        params(i + q, ps[q]);
        params(gend + q, o);                               // the head-output conv: weights, bias, decode targets
        ps[q].w3p = o.wp; ps[q].b3 = o.bias; ps[q].C3 = o.N;
        ps[q].dec_boxes = o.dec_boxes; ps[q].dec_scores = o.dec_scores; ps[q].dec_cls = o.dec_cls;
        ps[q].dec_N = o.dec_N; ps[q].dec_off = o.dec_off; ps[q].dec_C = o.dec_C; ps[q].dec_mode = o.dec_mode;
        ps[q].dec_center = o.dec_center; ps[q].dec_wh = o.dec_wh; ps[q].dec_raw = o.dec_raw;
        ps[q].dec_stride = o.dec_stride; ps[q].dec_hi = o.dec_hi;
      }
      const hipError_t e = yl_launch_conv_dpp(ps, (int)n, st);
      if (e == hipSuccess) { i = gend + n; continue; }
      if (e != hipErrorNotSupported) {
        char b[256];
        snprintf(b, sizeof(b), "layers %zu..%zu head launch failed: %s", i, gend + n - 1, hipGetErrorString(e));
        return fail(c, YL_ERR_HIP, b);
      }
    }
    // wide depthwise 3x3 -> 1x1 layer (more than 96 outputs) on its own: yl_conv_dpq_kernel's expand-only form computes
    // the depthwise part once per pixel
    if (!evs && !lanes && gend == i + 1 && d.op == YL_OP_CONV && d.dw_k == 3 && d.dw_stride == 1 && d.k == 1 && d.c2 == 0 &&
        d.c3 == 0 && c->opt_fuse_head && !c->opt_bf16 && d.head_level < 0 && d.res_slot < 0 && d.up_slot < 0 && !d.in_shift &&
        d.cout > 96 && !pair_fusable(c, i, lend, false) && c->layers[i].in_h == c->layers[i].out_h &&
        yl_dpq_supported(d.cin, d.cout, 0, c->layers[i].out_h, c->layers[i].out_w)) {
      YlConvP pt;
      params(i, pt);
      pt.w3p = nullptr;
      const hipError_t e = yl_launch_conv_dpq(pt, st);
      if (e == hipSuccess) { ++i; continue; }
      if (e != hipErrorNotSupported) {
        char b[256];
        snprintf(b, sizeof(b), "layer %zu launch failed: %s", i, hipGetErrorString(e));
        return fail(c, YL_ERR_HIP, b);
      }
    }
    if (!evs && !lanes && gend == i + 1 && d.op == YL_OP_CONV && d.dw_k == 3 && pair_fusable(c, i, lend, false)) {
      YlConvP pt, po;
      params(i, pt);
      params(i + 1, po);
      pt.w3p = po.wp; pt.b3 = po.bias; pt.C3 = po.N; pt.act3 = po.act; pt.res = po.res; pt.out = po.out;
      const hipError_t e = yl_launch_conv_dpq(pt, st);
      if (e == hipSuccess) { i += 2; continue; }
      if (e != hipErrorNotSupported) {
        char b[256];
        snprintf(b, sizeof(b), "layers %zu..%zu fused launch failed: %s", i, i + 1, hipGetErrorString(e));
        return fail(c, YL_ERR_HIP, b);
      }
    }
    // head-output conv(s) of a model with mask coefficients under yl_predict: det rows through the decode epilogue (no raw
    // rows), the mask coefficients as a second plain 1x1 launch into the level rows (bit-identical values: same k order)
    if (fuse && !evs && !lanes && d.op == YL_OP_CONV && d.head_level >= 0 && c->layers[i].wp_det && c->opt_fuse_head) {
      bool all = true;
      for (size_t q = i; q < gend; ++q) all = all && c->layers[q].wp_det != nullptr;
      if (all) {
        YlConvP pd[4];
        const int nd = 5 + c->C;
        for (size_t q = i; q < gend; ++q) {
          YlConvP o;
          params(q, o);
          YlConvP& a = pd[q - i];
          a = o; a.wp = c->layers[q].wp_det; a.bias = c->layers[q].b_det; a.N = nd; a.NTtot = cdiv(nd, 16); a.dec_raw = 0;
          // the coefficient part rides along as the second weight image (yl_launch_conv_multi: one pass where it pays, else a
          // second plain 1x1 launch): its columns go into the level rows behind the detection columns
          a.w3p = c->layers[q].wp_mc; a.b3 = c->layers[q].b_mc; a.C3 = c->NM;
          a.out = o.out + nd; a.ldo = c->E;
        }
        const int n = (int)(gend - i);
        const hipError_t e = conv_multi(c, pd, n, st);
        if (e != hipSuccess) {
          char b[256];
          snprintf(b, sizeof(b), "layers %zu..%zu split head launch failed: %s", i, gend - 1, hipGetErrorString(e));
          return fail(c, YL_ERR_HIP, b);
        }
        i = gend;
        continue;
      }
    }
    if (gend - i > 1) {
      YlConvP ps[4];
      for (size_t q = i; q < gend; ++q) params(q, ps[q - i]);
      const hipError_t e = conv_multi(c, ps, (int)(gend - i), st);
      if (e != hipSuccess) {
        char b[256];
        snprintf(b, sizeof(b), "layers %zu..%zu batched launch failed: %s", i, gend - 1, hipGetErrorString(e));
        return fail(c, YL_ERR_HIP, b);
      }
      i = gend;
      continue;
    }
    YlConvP p;
    params(i, p);
    const int ln = (lanes && c->lane[i]) ? 1 : 0;
    hipStream_t ls = ln ? sd : st;
    if (lanes) {
      bool cross = false;
      const int ins[4] = {d.in_slot, d.res_slot, d.up_slot, d.scale_slot};
      for (int k = 0; k < 4; ++k)
        if (ins[k] >= 0 && d.op != YL_OP_STEM && d.op != YL_OP_STEMBLOCK && d.op != YL_OP_NHWC4 && prod[ins[k]] != ln) cross = true;
      if (ln == 1 && !side_used) cross = true;             // first side launch: order after everything enqueued so far
      if (cross) {
        hipEvent_t ev = ln ? c->ev_la[chunk] : c->ev_lb[chunk];
        HIPCHK(c, hipEventRecord(ev, ln ? st : sd));
        HIPCHK(c, hipStreamWaitEvent(ls, ev, 0));
      }
      if (ln) side_used = true;
      if (d.head_level < 0 && d.out_slot >= 0) prod[d.out_slot] = (unsigned char)ln;
    }
    hipError_t e;
    switch (d.op) {
      case YL_OP_SE: {
        const DevLayer& L = c->layers[i];
        YlSeP sp;
        sp.x = p.x; sp.gate = p.out;
        sp.w1 = L.wp; sp.b1 = L.bias; sp.w2 = L.w2p; sp.b2 = L.b2;
        sp.B = B; sp.HW = L.in_h * L.in_w; sp.C = d.cin; sp.RD = d.cout; sp.act = d.act;
        const bool pooled = pooled_slot == d.in_slot;        // the depthwise launch in front left the partial sums
        sp.P = pooled ? pooled_P : yl_se_parts(sp.HW, sp.C);
        int ch = 0;                                          // the chunk arena these images live in
        while (ch + 1 < c->plan_n && b0 >= c->plan_b0[ch + 1]) ++ch;
        sp.partial = c->se_scratch[ch] + (size_t)(b0 - c->plan_b0[ch]) * c->se_unit;
        if (c->opt_bf16 == 3 && !pooled) return fail(c, YL_ERR_UNSUPPORTED, "store_f16: squeeze-excite pooling needs the pooled depthwise launch in front of it");
        e = yl_launch_se(sp, pooled, ls);
        break;
      }
      case YL_OP_DW: {
        // feeding a squeeze-excite gate next (efficientnetv2 MBConv): pool in the same launch
        pooled_slot = -1;
        if (i + 1 < lend && c->layers[i + 1].d.op == YL_OP_SE && c->layers[i + 1].d.in_slot == d.out_slot && d.res_slot < 0 &&
            (!c->opt_bf16 || c->opt_bf16 == 3) && !(c->opt_dev & YL_DEV_DW_TILE_OFF)) {
          const int wpi = yl_dw_pool_wpi(d.k, d.stride, d.cin, d.cout, c->layers[i].out_h, c->layers[i].out_w);
          if (wpi > 0 && c->se_unit >= (size_t)wpi * d.cin) {
            int ch = 0;
            while (ch + 1 < c->plan_n && b0 >= c->plan_b0[ch + 1]) ++ch;
            p.pool = c->se_scratch[ch] + (size_t)(b0 - c->plan_b0[ch]) * c->se_unit;
            p.pool_wpi = wpi;
            pooled_slot = d.out_slot; pooled_P = wpi;
          }
        }
        e = c->opt_bf16 == 3 ? yl_launch_dw_f16s(p, ls) : yl_launch_dw(p, ls);
        break;
      }
      case YL_OP_POOL: case YL_OP_COPY: case YL_OP_LN: case YL_OP_GRN: case YL_OP_NHWC4: {
        if (c->opt_bf16 == 3) return fail(c, YL_ERR_UNSUPPORTED, "store_f16: the element-wise ops of the hgnetv2 / convnextv2 backbones are fp32-storage only");
        const DevLayer& L = c->layers[i];
        YlOpP q;
        memset(&q, 0, sizeof(q));
        q.x = p.x; q.out = p.out; q.w = L.wp; q.b = L.bias;
        q.B = B; q.H = L.in_h; q.W = L.in_w; q.C = d.cin;
        q.OH = d.op == YL_OP_GRN ? L.in_h : L.out_h; q.OW = d.op == YL_OP_GRN ? L.in_w : L.out_w;
        q.k = d.k; q.stride = d.stride; q.pad_t = d.pad_t; q.pad_l = d.pad_l;
        q.ldo = d.op == YL_OP_COPY ? c->slots[d.out_slot].c : d.cin; q.ch_off = d.out_ch_off;
        q.eps = d.eps;
        if (d.op == YL_OP_GRN) {
          q.P = yl_grn_parts(L.in_h * L.in_w);
          int ch = 0;                                          // the chunk arena these images live in
          while (ch + 1 < c->plan_n && b0 >= c->plan_b0[ch + 1]) ++ch;
          q.partial = c->se_scratch[ch] + (size_t)(b0 - c->plan_b0[ch]) * c->se_unit;
        }
        e = yl_launch_op(d.op, q, ls);
        break;
      }
      case YL_OP_STEM: e = c->opt_bf16 == 3 ? yl_launch_stem_f16s(p, ls) : yl_launch_stem(p, ls); break;
      case YL_OP_CONV:
        // small-channel dense 3x3 on large grids: window-in-LDS kernel (fp32 units only; "tile_m" 6 keeps the generic kernel)
        e = (!c->opt_bf16 && c->opt_tile_m != 6) ? yl_launch_conv_k3w(p, ls) : hipErrorNotSupported;
        if (e != hipErrorNotSupported) break;
        e = c->opt_bf16 == 1 ? yl_launch_conv_bf16(p, c->opt_tile_m, ls)
            : c->opt_bf16 == 2 ? yl_launch_conv_f16(p, c->opt_tile_m, ls)
            : c->opt_bf16 == 3 ? yl_launch_conv_f16s(p, c->opt_tile_m, ls) : yl_launch_conv(p, c->opt_tile_m, ls);
        break;
      case YL_OP_STEMBLOCK:
        if (d.dw_k == 3)
          e = c->opt_bf16 == 1 ? yl_launch_stemdw_bf16(p, ls) : c->opt_bf16 == 2 ? yl_launch_stemdw_f16(p, ls)
              : c->opt_bf16 == 3 ? yl_launch_stemdw_f16s(p, ls) : yl_launch_stemdw(p, ls);
        else
        e = c->opt_bf16 == 1 ? yl_launch_stemblock_bf16(p, ls) : c->opt_bf16 == 2 ? yl_launch_stemblock_f16(p, ls)
            : c->opt_bf16 == 3 ? yl_launch_stemblock_f16s(p, ls) : yl_launch_stemblock(p, ls);
        break;
      default: e = c->opt_bf16 == 3 ? yl_launch_dw_f16s(p, ls) : yl_launch_dw(p, ls); break;
    }
    if (e == hipSuccess && YL_ACT_POSTPASS(d.act) && c->opt_bf16 == 3) return fail(c, YL_ERR_UNSUPPORTED, "store_f16: GELU / ReLU + affine passes are fp32-storage only");
    if (e == hipSuccess && YL_ACT_POSTPASS(d.act)) {
      // GELU / ReLU + learnable affine (+ the residual behind it): element-wise pass over the layer's output, in place
      const DevLayer& L = c->layers[i];
      YlOpP q;
      memset(&q, 0, sizeof(q));
      q.x = p.out; q.out = p.out;
      q.B = B; q.OH = L.out_h; q.OW = L.out_w; q.C = d.cout;
      q.act = d.act; q.lab_s = d.lab_scale; q.lab_b = d.lab_bias;
      if (d.res_slot >= 0) q.res = slot_addr(c, d.res_slot, b0);
      e = yl_launch_op(YL_OP_ACTPASS, q, ls);
    }
    if (e != hipSuccess) {
      char b[256];
      snprintf(b, sizeof(b), "layer %zu launch failed: %s", i, hipGetErrorString(e));
      return fail(c, YL_ERR_HIP, b);
    }
    if (evs) HIPCHK(c, hipEventRecord(evs[i + 1], st));
    ++i;
  }
  if (side_used) {                                          // join: decode / the caller see both lanes
    HIPCHK(c, hipEventRecord(c->ev_lb[chunk], sd));
    HIPCHK(c, hipStreamWaitEvent(st, c->ev_lb[chunk], 0));
  }
  return YL_OK;
}

// lane assignment from the slot graph: a layer goes to lane 1 iff everything it feeds ends in head outputs of
// levels >= 1 only (the finest level's chain, the backbone, the top-down laterals and the prototype branch stay
// on lane 0)
// the longest run of consecutive layers whose input AND output grids are <= 1/16 of the image (the 40x40 / 20x20
// stages of the backbone and the coarse part of the top-down pass): the part of the network that is chunked over
// the internal streams by the hybrid plan
void assign_small_run(yl_ctx* c) {
  const int lim = c->img_size / 16;
  int best_lo = 0, best_hi = 0, lo = -1;
  const int n = (int)c->layers.size();
  for (int i = 0; i <= n; ++i) {
    const bool small = i < n && c->layers[i].d.op == YL_OP_CONV && c->layers[i].in_h <= lim && c->layers[i].out_h <= lim &&
                       c->layers[i].d.head_level < 0;
    if (small && lo < 0) lo = i;
    if (!small && lo >= 0) {
      if (i - lo > best_hi - best_lo) { best_lo = lo; best_hi = i; }
      lo = -1;
    }
  }
  c->small_lo = best_lo; c->small_hi = best_hi;
  const int lim2 = c->img_size / 32;
  best_lo = best_hi = 0; lo = -1;
  for (int i = 0; i <= n; ++i) {
    const bool small = i < n && (c->layers[i].d.op == YL_OP_CONV || c->layers[i].d.op == YL_OP_DW || c->layers[i].d.op == YL_OP_SE) &&
                       c->layers[i].in_h <= lim2 && c->layers[i].out_h <= lim2 && c->layers[i].d.head_level < 0;
    if (small && lo < 0) lo = i;
    if (!small && lo >= 0) {
      if (i - lo > best_hi - best_lo) { best_lo = lo; best_hi = i; }
      lo = -1;
    }
  }
  c->tiny_lo = best_lo; c->tiny_hi = best_hi;
}

void assign_lanes(yl_ctx* c) {
  const size_t n = c->layers.size();
  c->lane.assign(n, 0);
  std::vector<unsigned> reach(n, 0u);
  for (size_t ii = n; ii-- > 0;) {
    const yl_layer& d = c->layers[ii].d;
    unsigned r = 0;
    if (d.head_level >= 0) r |= 1u << (d.head_level > 30 ? 30 : d.head_level);
    if (d.out_slot >= 0 && d.out_slot == c->proto_slot) r |= 1u << 31;
    if (d.head_level < 0 && d.out_slot >= 0)
      for (size_t j = ii + 1; j < n; ++j) {
        const yl_layer& e = c->layers[j].d;
        if (e.in_slot == d.out_slot || e.res_slot == d.out_slot || e.up_slot == d.out_slot || e.scale_slot == d.out_slot) r |= reach[j];
      }
    reach[ii] = r;
  }
  bool any = false;
  for (size_t i = 0; i < n; ++i) {
    c->lane[i] = (reach[i] != 0 && (reach[i] & 1u) == 0 && (reach[i] >> 31) == 0) ? 1 : 0;
    any |= c->lane[i] != 0;
  }
  if (!any) c->lane.clear();
}

yl_status check_cfg(yl_ctx* c, const yl_post_cfg* cfg) {
  if (!cfg) return fail(c, YL_ERR_INVALID, "cfg is NULL");
  if (cfg->mode < YL_POST_MAIN || cfg->mode > YL_POST_EVAL) return fail(c, YL_ERR_INVALID, "bad post mode");
  if (cfg->max_out <= 0) return fail(c, YL_ERR_INVALID, "max_out must be > 0");
  if (cfg->center_mode < 0 || cfg->center_mode > 1 || cfg->wh_mode < 0 || cfg->wh_mode > 2)
    return fail(c, YL_ERR_INVALID, "bad center/wh mode");
  if (cfg->fallback_nms != YL_NMS_TORCHVISION && cfg->fallback_nms != YL_NMS_GREEDY)
    return fail(c, YL_ERR_INVALID, "bad fallback_nms");
  return YL_OK;
}

// post-processing of images [b0, b0+B) of a batch (workspaces must already cover b0+B images)
yl_status do_post(yl_ctx* c, const float* const* levels_all, int b0, int B, const yl_post_cfg* cfg, float* dets,
                  int* counts, int* keep_idx, hipStream_t st, bool decoded = false /*NMS inputs already written*/) {
  const float* levels[YL_MAX_LEVELS];
  for (int l = 0; l < c->L; ++l)
    levels[l] = levels_all[l] + (size_t)b0 * c->level_A[l] * c->level_S[l] * c->level_S[l] * c->E;
  const size_t o = (size_t)b0 * c->N;
  YlLevels lv;
  fill_levels(c, levels, lv);
  YlDecodeP dp;
  dp.mode = cfg->mode; dp.center_mode = cfg->center_mode; dp.wh_mode = cfg->wh_mode;
  dp.boxes = c->ws_boxes + o; dp.scores = c->ws_scores + o; dp.cls = c->ws_cls + o;
  if (!decoded) HIPCHK(c, yl_launch_decode_score(lv, B, dp, st));
  YlNmsP np;
  memset(&np, 0, sizeof(np));
  np.boxes = c->ws_boxes + o; np.scores = c->ws_scores + o; np.cls = c->ws_cls + o;
  np.N = c->N; np.C = c->C > 0 ? c->C : 1;
  np.conf_thr = cfg->conf_thr; np.iou_thr = cfg->iou_thr;
  np.impl = (cfg->mode == YL_POST_FALLBACK && cfg->fallback_nms == YL_NMS_GREEDY) ? YL_NMS_GREEDY : YL_NMS_TORCHVISION;
  np.cap = (cfg->per_class_cap > 0) ? cfg->per_class_cap : INT_MAX;
  np.topk = (cfg->mode == YL_POST_FALLBACK && cfg->topk > 0) ? cfg->topk : 0;
  np.max_out = cfg->max_out;
  np.cls_ws = c->ws_clsws + (size_t)b0 * 4 * np.C;
  np.gkeys = c->ws_gkeys ? c->ws_gkeys + (size_t)b0 * c->gP : nullptr; np.gP = c->gP;
  np.lds_cap = c->gP < YL_LDS_KEYS_MAX ? c->gP : YL_LDS_KEYS_MAX;
  np.dets = dets + (size_t)b0 * cfg->max_out * 6; np.counts = counts + b0;
  np.keep_idx = keep_idx ? keep_idx + (size_t)b0 * cfg->max_out : nullptr;
  np.backmap = cfg->backmap_dev ? cfg->backmap_dev + (size_t)b0 * 5 : nullptr;
  np.tmp_dets = c->ws_tmp_dets + o * 6; np.tmp_idx = c->ws_tmp_idx + o;
  // several workgroups per image (classes split mod G) unless the fallback's global top-k needs the whole kept
  // set in one workgroup, or survivors could exceed the LDS key capacity
  // auto: the class-group split (histogram + assignment prologue, merge kernel) pays when an image has thousands of
  // survivors -- evaluation thresholds (conf 0.001: +2.7 % on a 2000-survivor workload) -- and costs 0.8 % at the
  // few hundred survivors of a detector operating point (conf 0.4, calibrated benchmark workload)
  const int groups = c->opt_nms_groups > 0 ? c->opt_nms_groups : (cfg->conf_thr < 0.05f ? YL_NMS_GROUPS : 1);
  np.G = (groups > 1 && np.topk == 0 && c->C > 1 && c->C <= 256 && c->N <= YL_LDS_KEYS_MAX) ? groups : 1;
  np.kept_list = c->ws_kept_list + o * YL_NMS_GROUPS;
  np.done = c->ws_done + (size_t)b0 * 64;                              // 256 bytes per image
  HIPCHK(c, yl_launch_nms(np, B, st));
  return YL_OK;
}

struct Job {
  const float* x = nullptr;          // nullptr: no forward (post-processing of caller levels only)
  int B = 0;
  float* outs[YL_MAX_LEVELS] = {nullptr};
  const yl_post_cfg* cfg = nullptr;  // nullptr: forward only
  float* dets = nullptr;
  int* counts = nullptr;
  int* keep_idx = nullptr;
};

// A job runs as a sequence of SEGMENTS: layer range [lo, hi), either as one full-batch piece on the caller's
// stream or as one piece per batch chunk on the internal streams (fork/join); the last segment carries the
// post-processing.  Plans:
//   plain    [0, L) chunked, post per chunk                      (1 segment; n == 1: a single piece)
//   hybrid   [0, lo) full | [lo, hi) chunked | [hi, L) full + post
// hybrid: the high-resolution layers (stem block, 160^2 / 80^2 stages, the 80^2 neck / heads) fill the machine on
// their own -- their grids are persistent and occupancy-sized, so a half-batch launch of another chunk cannot
// overlap them, it only queues behind them -- while the run of <= 1/16-resolution layers is latency-bound with a
// few hundred tiles: those runs of the two chunks interleave on two streams.
struct Seg { int lo, hi; bool chunked, post; };

int plan_segments(const yl_ctx* c, const Job& j, int n, Seg* segs) {
  const int L = (int)c->layers.size();
  if (!j.x) { segs[0] = {0, 0, n > 1, true}; return 1; }
  if (c->opt_time_split && j.cfg) {          // one chunk (chunks_for): conv layers | post-processing, an event between them
    segs[0] = {0, L, false, false};
    segs[1] = {L, L, false, true};
    return 2;
  }
  if (n > 1 && c->opt_hybrid == 2 && !c->opt_lanes && c->tiny_hi - c->tiny_lo >= 6) {
    int k = 0;
    if (c->tiny_lo > 0) segs[k++] = {0, c->tiny_lo, true, false};
    segs[k++] = {c->tiny_lo, c->tiny_hi, false, false};
    segs[k++] = {c->tiny_hi, L, true, j.cfg != nullptr};
    return k;
  }
  if (n > 1 && c->opt_hybrid == 1 && !c->opt_lanes && c->small_hi - c->small_lo >= 6) {
    int k = 0;
    if (c->small_lo > 0) segs[k++] = {0, c->small_lo, false, false};
    segs[k++] = {c->small_lo, c->small_hi, true, false};
    segs[k++] = {c->small_hi, L, false, j.cfg != nullptr};
    return k;
  }
  segs[0] = {0, L, n > 1, j.cfg != nullptr};
  return 1;
}

// one piece: layers [lo, hi) of images [b0, b0 + bn), then (last segment) post-processing of those images
yl_status run_piece(yl_ctx* c, const Job& j, const Seg& sg, int b0, int bn, hipStream_t st, int chunk) {
  yl_status s = YL_OK;
  const bool fused = j.x && j.cfg && can_fuse_decode(c);
  // ("time_split": the events are recorded by walk_plan AROUND the pieces -- conv layers | post-processing -- so that a
  // piece can also be a replayed hipGraph)
  if (j.x && sg.hi > sg.lo) s = run_layers(c, j.x, b0, bn, j.outs, st, nullptr, chunk, fused ? j.cfg : nullptr, sg.lo, sg.hi);
  if (s == YL_OK && sg.post && j.cfg) s = do_post(c, j.outs, b0, bn, j.cfg, j.dets, j.counts, j.keep_idx, st, fused);
  return s;
}

// ---- stream / hardware-queue aliasing probe (yl_streams_overlap, include/yololite_hip.h)
__global__ void yl_spin_kernel(long long ticks) {
  const long long t0 = wall_clock64();                       // 100 MHz constant clock
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(8);
}

// 1 = kernel CHAINS on a and b overlap, 0 = they serialise, -1 = HIP error.  Two rounds, the better one counts (a busy device
// can only make a pair look serialised).  Measured (round 6): a good pair finishes two chains of 8 x 25 us in ~231 us, an aliasing
// pair in ~465 us; with ONE 200 us kernel per stream both kinds finish in ~224 us -- lone kernels do not show the aliasing.
int streams_overlap(hipStream_t a, hipStream_t b) {
  const long long ticks = 20000;                             // 200 us
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  hipLaunchKernelGGL(yl_spin_kernel, dim3(1), dim3(64), 0, a, 100LL);   // code object load / first-launch cost off the clock
  hipLaunchKernelGGL(yl_spin_kernel, dim3(1), dim3(64), 0, b, 100LL);
  if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
  // CHAINS of dependent kernels, not one kernel per stream: two lone kernels overlap on any pair of streams; what a serving
  // loop needs is that the command processor keeps dispatching stream b's chain while stream a's next kernel waits for its
  // predecessor (two streams whose hardware queues share a pipe fail exactly there)
  const int links = 8;
  double best = 1e30;
  for (int r = 0; r < 2; ++r) {
    const auto t0 = std::chrono::steady_clock::now();
    for (int i = 0; i < links; ++i) {
      hipLaunchKernelGGL(yl_spin_kernel, dim3(1), dim3(64), 0, a, ticks / links);
      hipLaunchKernelGGL(yl_spin_kernel, dim3(1), dim3(64), 0, b, ticks / links);
    }
    if (hipStreamSynchronize(a) != hipSuccess || hipStreamSynchronize(b) != hipSuccess) return -1;
    const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
    if (us < best) best = us;
  }
  return best < 1.5 * 200.0 ? 1 : 0;
}

// a new non-blocking stream whose kernels overlap those of every stream in `others` (up to 12 candidates: the hardware-queue
// assignment walks round-robin with the streams created; the rejected ones are destroyed afterwards -- destroying one right
// away would hand the same queue to the next candidate's successor).  Falls back to the last candidate.
yl_status make_overlapping_stream(yl_ctx* c, const hipStream_t* others, int n, hipStream_t* out) {
  std::vector<hipStream_t> rejected;
  hipStream_t s = nullptr;
  for (int attempt = 0; attempt < 12; ++attempt) {
    HIPCHK(c, hipStreamCreateWithFlags(&s, hipStreamNonBlocking));
    bool ok = true;
    for (int i = 0; i < n && ok; ++i) ok = streams_overlap(others[i], s) != 0;
    if (ok || attempt == 11) break;
    rejected.push_back(s);
    s = nullptr;
  }
  for (hipStream_t r : rejected) hipStreamDestroy(r);
  *out = s;
  return YL_OK;
}

// Only the streams this job needs (round 6): n - 1 chunk streams, side streams only under the "lanes" option.  A context
// used to create all seven at its first call; ROCm hands streams to the GPU_MAX_HW_QUEUES hardware queues as they appear,
// and with two cloned contexts per serving pipeline (14 idle streams) the two LANE streams of the pipeline ended up
// sharing a queue: 38.6 k instead of 44.9 k images/s (edge_n B=64, two batches in flight).  A one-stream context now
// creates no stream at all.
yl_status ensure_streams(yl_ctx* c, int n, hipStream_t st) {
  for (int i = 0; i < 4 && i < n; ++i) {
    if (i > 0 && !c->work[i]) {
      // chunk stream i must run beside the caller's stream (chunk 0) and beside its siblings: probe, do not assume
      hipStream_t others[4] = {st, nullptr, nullptr, nullptr};
      int no = 1;
      for (int k = 1; k < i; ++k) others[no++] = c->work[k];
      hipStreamCaptureStatus cap = hipStreamCaptureStatusNone;
      if (hipStreamIsCapturing(st, &cap) != hipSuccess || cap != hipStreamCaptureStatusNone) {
        (void)hipGetLastError();
        HIPCHK(c, hipStreamCreateWithFlags(&c->work[i], hipStreamNonBlocking));   // caller is capturing: no probe possible
      } else {
        yl_status ms = make_overlapping_stream(c, others, no, &c->work[i]);
        if (ms != YL_OK) return ms;
      }
    }
    if (i > 0 && !c->ev_join[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_join[i], hipEventDisableTiming));
    if (!c->opt_lanes) continue;
    if (!c->side[i]) HIPCHK(c, hipStreamCreateWithFlags(&c->side[i], hipStreamNonBlocking));
    if (!c->ev_la[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_la[i], hipEventDisableTiming));
    if (!c->ev_lb[i]) HIPCHK(c, hipEventCreateWithFlags(&c->ev_lb[i], hipEventDisableTiming));
  }
  if (!c->ev_fork) HIPCHK(c, hipEventCreateWithFlags(&c->ev_fork, hipEventDisableTiming));
  return YL_OK;
}

// Walk the plan.  `piece(seg, chunk, b0, bn, stream)` either enqueues the piece's kernels (eager) or launches its
// captured graph; chunked segments are forked over the internal streams and joined back on `st`.
template <typename F>
yl_status walk_plan(yl_ctx* c, const Job& j, hipStream_t st, int n, const Seg* segs, int nseg, F&& piece) {
  const int base = j.B / n, rem = j.B % n;
  const bool timed = c->opt_time_split && j.x && j.cfg && nseg == 2 && c->ev_t[0];
  for (int g = 0; g < nseg; ++g) {
    if (!segs[g].chunked || n == 1) {
      if (timed) HIPCHK(c, hipEventRecord(c->ev_t[g], st));       // before the conv layers / between them and the NMS
      yl_status s = piece(g, 0, 0, j.B, st);
      if (s != YL_OK) return s;
      if (timed && g == 1) { HIPCHK(c, hipEventRecord(c->ev_t[2], st)); c->timing_valid = true; }
      continue;
    }
    HIPCHK(c, hipEventRecord(c->ev_fork, st));
    int b0 = 0;
    for (int i = 0; i < n; ++i) {
      const int bn = base + (i < rem ? 1 : 0);
      hipStream_t ws = (i == 0) ? st : c->work[i];
      if (i > 0) HIPCHK(c, hipStreamWaitEvent(ws, c->ev_fork, 0));
      yl_status s = piece(g, i, b0, bn, ws);
      if (s != YL_OK) return s;
      if (i > 0) {
        HIPCHK(c, hipEventRecord(c->ev_join[i], ws));
        HIPCHK(c, hipStreamWaitEvent(st, c->ev_join[i], 0));
      }
      b0 += bn;
    }
  }
  return YL_OK;
}

// run a job eagerly, or by replaying cached hipGraphs of exactly this job (one graph per piece: capturing several
// chunks plus side lanes into ONE graph -- 4 streams -- crashed hipGraphInstantiate on ROCm 7.2)
yl_status submit(yl_ctx* c, const Job& j, hipStream_t st, bool allow_graph = true) {
  const int n = chunks_for(c, j.B);
  Seg segs[3];
  const int nseg = plan_segments(c, j, n, segs);
  yl_status s = ensure_streams(c, n, st);
  if (s != YL_OK) return s;
  if (!c->opt_graph || !allow_graph)
    return walk_plan(c, j, st, n, segs, nseg, [&](int g, int i, int b0, int bn, hipStream_t ws) -> yl_status {
      return run_piece(c, j, segs[g], b0, bn, ws, i);
    });
  std::vector<unsigned char> key(sizeof(Job) + sizeof(yl_post_cfg) + 3 * sizeof(int), 0);
  memcpy(key.data(), &j, sizeof(Job));
  if (j.cfg) memcpy(key.data() + sizeof(Job), j.cfg, sizeof(yl_post_cfg));
  const int optkey = c->opt_streams | (c->opt_lanes << 8) | ((c->opt_bf16 & 1) << 9) | ((c->opt_bf16 >> 1) << 25) | (c->opt_fuse_decode << 10) |
                     (c->opt_batch_levels << 11) | ((c->opt_hybrid & 1) << 12) | (c->opt_nms_groups << 13) | ((c->opt_hybrid >> 1) << 24) |
                     (c->opt_winograd << 17) | (c->opt_fuse_head << 19);
  const int devkey = c->opt_dev | (c->opt_time_split << 16) | (c->opt_split_k << 17);
  memcpy(key.data() + sizeof(Job) + sizeof(yl_post_cfg), &optkey, sizeof(int));
  memcpy(key.data() + sizeof(Job) + sizeof(yl_post_cfg) + sizeof(int), &c->opt_tile_m, sizeof(int));
  memcpy(key.data() + sizeof(Job) + sizeof(yl_post_cfg) + 2 * sizeof(int), &devkey, sizeof(int));
  // the cfg POINTER is part of Job but not of the identity of the work: blank it in the key
  memset(key.data() + offsetof(Job, cfg), 0, sizeof(void*));
  yl_ctx::GraphEntry* ge = nullptr;
  for (auto& g : c->graphs)
    if (g.key == key) { ge = &g; break; }
  if (!ge) {
    yl_ctx::GraphEntry fresh;
    hipStream_t cs;
    HIPCHK(c, hipStreamCreateWithFlags(&cs, hipStreamNonBlocking));
    hipError_t e = hipSuccess;
    // capture every piece in plan order (walk_plan's fork/join calls are NOT issued here: pieces are independent
    // captures on the scratch stream)
    const int base = j.B / n, rem = j.B % n;
    for (int g = 0; g < nseg && s == YL_OK && e == hipSuccess; ++g) {
      const int pieces = (segs[g].chunked && n > 1) ? n : 1;
      int b0 = 0;
      for (int i = 0; i < pieces && s == YL_OK && e == hipSuccess; ++i) {
        const int bn = pieces == 1 ? j.B : base + (i < rem ? 1 : 0);
        hipGraph_t gr = nullptr;
        hipGraphExec_t ex = nullptr;
        e = hipStreamBeginCapture(cs, hipStreamCaptureModeThreadLocal);
        if (e != hipSuccess) break;
        s = run_piece(c, j, segs[g], b0, bn, cs, i);
        e = hipStreamEndCapture(cs, &gr);
        if (s == YL_OK && e == hipSuccess) e = hipGraphInstantiate(&ex, gr, nullptr, nullptr, 0);
        if (gr) hipGraphDestroy(gr);
        fresh.execs.push_back(ex);
        b0 += bn;
      }
    }
    hipStreamDestroy(cs);
    if (s != YL_OK || e != hipSuccess) {
      for (auto ex : fresh.execs)
        if (ex) hipGraphExecDestroy(ex);
      if (s != YL_OK) return s;
      HIPCHK(c, e);
    }
    fresh.key = key;
    fresh.n = n;
    if (c->graphs.size() >= YL_GRAPH_SLOTS) {                 // evict the least recently used entry
      size_t v = 0;
      for (size_t i = 1; i < c->graphs.size(); ++i)
        if (c->graphs[i].stamp < c->graphs[v].stamp) v = i;
      for (auto ex : c->graphs[v].execs)
        if (ex) hipGraphExecDestroy(ex);
      c->graphs.erase(c->graphs.begin() + v);
    }
    c->graphs.push_back(fresh);
    ge = &c->graphs.back();
  }
  ge->stamp = ++c->graph_clock;
  size_t at = 0;
  std::vector<size_t> first(nseg);
  for (int g = 0; g < nseg; ++g) { first[g] = at; at += (segs[g].chunked && n > 1) ? n : 1; }
  return walk_plan(c, j, st, n, segs, nseg, [&](int g, int i, int, int, hipStream_t ws) -> yl_status {
    HIPCHK(c, hipGraphLaunch(ge->execs[first[g] + i], ws));
    return YL_OK;
  });
}

}  // namespace

// ================================================================================================
extern "C" {

int32_t yl_abi_version(void) { return YL_ABI_VERSION; }

const char* yl_strerror(yl_status s) {
  switch (s) {
    case YL_OK: return "ok";
    case YL_ERR_INVALID: return "invalid argument";
    case YL_ERR_HIP: return "HIP runtime error";
    case YL_ERR_NOMEM: return "out of memory";
    case YL_ERR_STATE: return "invalid state for this call";
    case YL_ERR_UNSUPPORTED: return "unsupported configuration";
    case YL_ERR_CAPACITY: return "output buffer too small";
    default: return "unknown yl_status";
  }
}

const char* yl_last_error(const yl_ctx* ctx) { return ctx ? ctx->err.c_str() : "null context"; }

void yl_destroy(yl_ctx* c) {
  if (!c) return;
  hipSetDevice(c->device);
  free_act(c);
  free_post_ws(c);
  hipFree(c->ws_nms_clsws);
  for (int i = 0; i < 4; ++i) {
    if (c->work[i]) hipStreamDestroy(c->work[i]);
    if (c->ev_join[i]) hipEventDestroy(c->ev_join[i]);
    if (c->side[i]) hipStreamDestroy(c->side[i]);
    if (c->ev_la[i]) hipEventDestroy(c->ev_la[i]);
    if (c->ev_lb[i]) hipEventDestroy(c->ev_lb[i]);
  }
  if (c->ev_fork) hipEventDestroy(c->ev_fork);
  for (int i = 0; i < 3; ++i) if (c->ev_t[i]) hipEventDestroy(c->ev_t[i]);
  hipFree(c->ws_nms_gkeys);
  if (c->weights_owner.use_count() <= 1) {                   // last context that shares these weights (yl_clone)
    hipFree(c->zeros);
    for (auto& L : c->layers) {
      hipFree(L.wp); hipFree(L.bias); hipFree(L.dw_w); hipFree(L.dw_b);
      hipFree(L.w2p); hipFree(L.b2); hipFree(L.w3p); hipFree(L.b3); hipFree(L.wino);
      hipFree(L.wp_det); hipFree(L.b_det); hipFree(L.wp_mc); hipFree(L.b_mc);
    }
  }
  delete c;
}

yl_status yl_clone(const yl_ctx* src, yl_ctx** out) {
  if (!src || !out) return YL_ERR_INVALID;
  *out = nullptr;
  if (hipSetDevice(src->device) != hipSuccess) return YL_ERR_HIP;
  yl_ctx* c = new (std::nothrow) yl_ctx();
  if (!c) return YL_ERR_NOMEM;
  *out = c;
  // the model: geometry, layer program and the (shared, immutable) packed weights
  c->device = src->device;
  c->img_size = src->img_size; c->in_ch = src->in_ch; c->C = src->C; c->L = src->L; c->N = src->N; c->E = src->E;
  c->NM = src->NM; c->proto_slot = src->proto_slot;
  memcpy(c->level_S, src->level_S, sizeof(c->level_S));
  memcpy(c->level_A, src->level_A, sizeof(c->level_A));
  memcpy(c->level_off, src->level_off, sizeof(c->level_off));
  c->slots.resize(src->slots.size());
  for (size_t i = 0; i < src->slots.size(); ++i) { c->slots[i].h = src->slots[i].h; c->slots[i].w = src->slots[i].w; c->slots[i].c = src->slots[i].c; }
  c->layers = src->layers;
  c->zeros = src->zeros;
  c->weights_owner = src->weights_owner;
  c->se_unit = src->se_unit;
  c->wino_max_hw = src->wino_max_hw;
  c->lane = src->lane;
  c->small_lo = src->small_lo; c->small_hi = src->small_hi; c->tiny_lo = src->tiny_lo; c->tiny_hi = src->tiny_hi;
  // the options as they are now; activation arenas, workspaces, streams, events and cached graphs are the clone's own
  c->opt_reuse = src->opt_reuse; c->opt_pre_norm = src->opt_pre_norm; c->opt_graph = src->opt_graph; c->opt_tile_m = src->opt_tile_m;
  c->opt_streams = src->opt_streams; c->opt_nms_groups = src->opt_nms_groups; c->opt_hybrid = src->opt_hybrid;
  c->opt_batch_levels = src->opt_batch_levels; c->opt_winograd = src->opt_winograd; c->opt_fuse_decode = src->opt_fuse_decode;
  c->opt_fuse_head = src->opt_fuse_head; c->opt_split_k = src->opt_split_k; c->opt_dev = src->opt_dev; c->opt_bf16 = src->opt_bf16;
  c->opt_lanes = src->opt_lanes;
  return YL_OK;
}

yl_status yl_streams_overlap(int32_t device_id, void* a, void* b, int32_t* overlap) {
  if (!overlap) return YL_ERR_INVALID;
  *overlap = 0;
  if (hipSetDevice(device_id) != hipSuccess) return YL_ERR_HIP;
  const int r = streams_overlap((hipStream_t)a, (hipStream_t)b);
  if (r < 0) return YL_ERR_HIP;
  *overlap = r;
  return YL_OK;
}

yl_status yl_create(const yl_model_desc* d, int32_t device_id, yl_ctx** out) {
  if (!d || !out) return YL_ERR_INVALID;
  *out = nullptr;
  if (d->abi_version != YL_ABI_VERSION) return YL_ERR_INVALID;
  if (d->num_levels < 1 || d->num_levels > YL_MAX_LEVELS) return YL_ERR_INVALID;
  if (d->num_classes < 0 || d->num_classes > 4096) return YL_ERR_UNSUPPORTED;
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || device_id < 0 || device_id >= ndev) return YL_ERR_HIP;
  if (hipSetDevice(device_id) != hipSuccess) return YL_ERR_HIP;
  if (device_id >= 64) return YL_ERR_UNSUPPORTED;
  if (!g_inited[device_id]) {
    if (yl_post_init() != hipSuccess || yl_conv_init() != hipSuccess || yl_stemblock_init() != hipSuccess ||
        yl_conv_init_bf16() != hipSuccess || yl_stemblock_init_bf16() != hipSuccess || yl_convc_init() != hipSuccess ||
        yl_convc_init_bf16() != hipSuccess || yl_dpp_init() != hipSuccess || yl_conv_init_f16() != hipSuccess ||
        yl_stemblock_init_f16() != hipSuccess || yl_convc_init_f16() != hipSuccess || yl_conv_init_f16s() != hipSuccess ||
        yl_stemblock_init_f16s() != hipSuccess || yl_convc_init_f16s() != hipSuccess)
      return YL_ERR_HIP;
    g_inited[device_id] = true;
  }
  yl_ctx* c = new (std::nothrow) yl_ctx();
  if (!c) return YL_ERR_NOMEM;
  *out = c;   // handed out even on failure so that yl_last_error() can be read; caller destroys it
  c->weights_owner = std::make_shared<int>(0);
  c->device = device_id;
  c->img_size = d->img_size; c->in_ch = d->in_channels; c->C = d->num_classes; c->L = d->num_levels;
  c->NM = d->num_masks; c->proto_slot = d->proto_slot;
  if (c->NM < 0 || c->NM > 64) return fail(c, YL_ERR_UNSUPPORTED, "num_masks must be in [0,64]");
  c->E = 5 + c->C + c->NM;
  int off = 0;
  for (int l = 0; l < c->L; ++l) {
    c->level_S[l] = d->level_size[l]; c->level_A[l] = d->level_anchors[l];
    if (c->level_S[l] < 1 || c->level_A[l] < 1) return fail(c, YL_ERR_INVALID, "bad level geometry");
    c->level_off[l] = off;
    off += c->level_A[l] * c->level_S[l] * c->level_S[l];
  }
  c->level_off[c->L] = off;
  c->N = off;
  if (c->N >= (1 << 20)) return fail(c, YL_ERR_UNSUPPORTED, "more than 2^20 candidates per image");
  if (d->num_layers == 0) return YL_OK;
  HIPCHK(c, hipMalloc((void**)&c->zeros, 1024));
  HIPCHK(c, hipMemset(c->zeros, 0, 1024));
  if (d->in_channels != 3) return fail(c, YL_ERR_UNSUPPORTED, "network input must have 3 channels");
  if (!d->layers || !d->slot_h || !d->slot_w || !d->slot_c) return fail(c, YL_ERR_INVALID, "null layer/slot arrays");
  c->slots.resize(d->num_slots);
  for (int i = 0; i < d->num_slots; ++i) {
    c->slots[i].h = d->slot_h[i]; c->slots[i].w = d->slot_w[i]; c->slots[i].c = d->slot_c[i];
    if (c->slots[i].h < 1 || c->slots[i].w < 1 || c->slots[i].c < 1 || (c->slots[i].c & 3))
      return fail(c, YL_ERR_UNSUPPORTED, "slot channels must be a positive multiple of 4");
  }
  std::vector<int> head_seen(c->L, 0);
  char msg[256];
  for (int i = 0; i < d->num_layers; ++i) {
    const yl_layer& l = d->layers[i];
    DevLayer L;
    L.d = l;
    auto bad = [&](const char* what) {
      snprintf(msg, sizeof(msg), "layer %d: %s", i, what);
      return fail(c, YL_ERR_INVALID, msg);
    };
    if (l.op < YL_OP_STEM || l.op > YL_OP_NHWC4) return bad("unknown op");
    if (l.reserved0 != 0) return bad("reserved0 must be 0");
    if (l.act < YL_ACT_NONE || l.act > YL_ACT_RELU_LAB || l.dw_act < 0 || l.dw_act > YL_ACT_SILU || l.act2 < 0 || l.act2 > YL_ACT_SILU ||
        l.act3 < 0 || l.act3 > YL_ACT_SILU)
      return bad("unknown activation (GELU / ReLU+affine are valid as `act` only)");
    if (YL_ACT_POSTPASS(l.act) && ((l.op != YL_OP_STEM && l.op != YL_OP_CONV && l.op != YL_OP_DW) || l.head_level >= 0 || l.up_slot >= 0 ||
                                   l.c2 > 0 || l.c3 > 0 || (l.cout & 3)))
      return bad("GELU / ReLU + learnable affine: plain STEM / CONV / DW layers with cout % 4 == 0 only");
    if (l.out_ch_off != 0 && l.op != YL_OP_COPY) return bad("out_ch_off is a YL_OP_COPY field");
    if (l.op >= YL_OP_POOL) {
      // element-wise / reduction ops (ABI v5): in_slot -> out_slot, no conv fields
      if (l.out_slot < 0 || l.out_slot >= d->num_slots) return bad("bad out_slot");
      if (l.op != YL_OP_NHWC4 && (l.in_slot < 0 || l.in_slot >= d->num_slots)) return bad("bad in_slot");
      if (l.res_slot >= 0 || l.up_slot >= 0 || l.scale_slot >= 0 || l.head_level >= 0 || l.dw_k || l.c2 || l.c3 || l.in_shift || l.act)
        return bad("element-wise op: plain layer fields only");
      const Slot& so = c->slots[l.out_slot];
      if (l.op == YL_OP_NHWC4) {
        if (so.h != d->img_size || so.w != d->img_size || so.c != 4 || l.cout != 4) return bad("NHWC4: out_slot must be [S,S,4]");
        L.in_h = L.in_w = d->img_size; L.out_h = L.out_w = d->img_size;
      } else {
        const Slot& si = c->slots[l.in_slot];
        if (si.c != l.cin || (l.cin & 3)) return bad("cin does not match the input slot / not a multiple of 4");
        L.in_h = si.h; L.in_w = si.w; L.out_h = so.h; L.out_w = so.w;
        if (l.op == YL_OP_POOL) {
          if (l.k < 1 || l.stride < 1 || l.cout != l.cin || so.c != l.cin || l.pad_t < 0 || l.pad_l < 0 || l.pad_t >= l.k || l.pad_l >= l.k ||
              (so.h - 1) * l.stride - l.pad_t >= si.h || (so.w - 1) * l.stride - l.pad_l >= si.w)
            return bad("pool: bad geometry");
        } else if (l.op == YL_OP_COPY) {
          if (so.h != si.h || so.w != si.w || l.out_ch_off < 0 || (l.out_ch_off & 3) || l.out_ch_off + l.cin > so.c)
            return bad("copy: channel slice outside the output slot");
        } else if (l.op == YL_OP_LN) {
          if (so.h != si.h || so.w != si.w || so.c != l.cin || l.cout != l.cin || !l.w || !l.b || !(l.eps > 0.0f)) return bad("layer norm: needs w, b [cin], eps > 0, same shape out");
        } else {     // GRN
          if (so.h != 1 || so.w != 1 || so.c != l.cin || !l.w || !(l.eps > 0.0f) || l.cin > 16384) return bad("GRN: out_slot must be [1,1,cin], needs w [cin], eps > 0");
          const size_t unit = (size_t)64 * l.cin;
          if (unit > c->se_unit) c->se_unit = unit;
          L.out_h = L.out_w = 1;
        }
      }
      yl_status s2;
      if (l.w && (l.op == YL_OP_LN || l.op == YL_OP_GRN)) {
        std::vector<float> wv(l.w, l.w + l.cin);
        if ((s2 = upload(c, wv, &L.wp)) != YL_OK) return s2;
      }
      if (l.b && l.op == YL_OP_LN) {
        std::vector<float> bv(l.b, l.b + l.cin);
        if ((s2 = upload(c, bv, &L.bias)) != YL_OK) return s2;
      }
      L.d.w = L.d.b = L.d.dw_w = L.d.dw_b = nullptr;
      L.d.w2 = L.d.b2 = L.d.w3 = L.d.b3 = nullptr;
      c->layers.push_back(L);
      continue;
    }
    if (!l.w) return bad("weights are NULL");
    if (l.k < 1 || l.stride < 1) return bad("bad kernel geometry");
    if (l.op == YL_OP_SE) {
      // squeeze-excite gate: in_slot [H,W,cin] -> out_slot [1,1,cin]; w/b = conv_reduce [cout][cin], w2/b2 = conv_expand [cin][cout]
      if (l.in_slot < 0 || l.in_slot >= d->num_slots || l.out_slot < 0 || l.out_slot >= d->num_slots) return bad("bad slot");
      const Slot& si = c->slots[l.in_slot]; const Slot& so = c->slots[l.out_slot];
      if (si.c != l.cin || so.c != l.cin || so.h != 1 || so.w != 1) return bad("squeeze-excite: out_slot must be [1,1,cin]");
      if (!l.w2 || !l.b || !l.b2 || l.c2 != l.cin || l.cout < 1 || l.cout > 256 || l.cin > 4096 || (l.cin & 3))
        return bad("squeeze-excite: needs w [cout][cin], b, w2 [cin][cout], b2, c2 == cin, cout <= 256, cin % 4 == 0 and <= 4096");
      if (l.k != 1 || l.stride != 1 || l.dw_k || l.c3 || l.res_slot >= 0 || l.up_slot >= 0 || l.scale_slot >= 0 || l.head_level >= 0 ||
          l.in_shift)
        return bad("squeeze-excite: plain layer fields only");
      L.in_h = si.h; L.in_w = si.w; L.out_h = L.out_w = 1;
      std::vector<float> w1(l.w, l.w + (size_t)l.cout * l.cin), b1(l.b, l.b + l.cout);
      std::vector<float> w2((size_t)l.cin * l.cout), b2(l.b2, l.b2 + l.cin);
      for (int cc = 0; cc < l.cin; ++cc)                       // conv_expand [cin][cout] -> [cout][cin]
        for (int j = 0; j < l.cout; ++j) w2[(size_t)j * l.cin + cc] = l.w2[(size_t)cc * l.cout + j];
      yl_status s2;
      if ((s2 = upload(c, w1, &L.wp)) != YL_OK || (s2 = upload(c, b1, &L.bias)) != YL_OK ||
          (s2 = upload(c, w2, &L.w2p)) != YL_OK || (s2 = upload(c, b2, &L.b2)) != YL_OK)
        return s2;
      // scratch: partial sums of the stand-alone pool pass, or of the depthwise launch that produces the tensor (<= 64 each)
      const size_t unit = (size_t)64 * l.cin;
      if (unit > c->se_unit) c->se_unit = unit;
      L.d.w = L.d.b = L.d.dw_w = L.d.dw_b = nullptr;
      L.d.w2 = L.d.b2 = L.d.w3 = L.d.b3 = nullptr;
      c->layers.push_back(L);
      continue;
    }
    if (l.scale_slot >= 0) {
      if (l.scale_slot >= d->num_slots || l.op != YL_OP_CONV || l.k != 1 || l.stride != 1 || l.dw_k || l.c2 || l.c3 || l.in_shift)
        return bad("scale_slot needs a plain 1x1 stride-1 conv");
      const Slot& g = c->slots[l.scale_slot];
      if (g.h != 1 || g.w != 1 || g.c != l.cin) return bad("scale_slot must be [1,1,cin]");
    }
    if (l.op == YL_OP_STEMBLOCK) {
      L.in_h = L.in_w = d->img_size;
      if (l.cin != 3 || l.k != 3) return fail(c, YL_ERR_UNSUPPORTED, "stem must be 3x3 with 3 input channels");
      if (!l.w2 || l.c2 < 1 || l.c3 < 0 || (l.c3 > 0 && !l.w3)) return bad("stem block needs w2 (and w3 when c3 > 0)");
      if (YL_SMOOTH(l.act) || YL_SMOOTH(l.act2) || YL_SMOOTH(l.act3))
        return fail(c, YL_ERR_UNSUPPORTED, "stem block: ReLU-family activations only");
      if (l.dw_k == 3) {       // second conv DEPTHWISE 3x3 stride 1 pad 1, then the 1x1: the EfficientNet-Lite entry (yl_stemdw_kernel)
        if (l.cout != 32 || l.c2 != 32 || l.c3 < 4 || l.c3 > 32 || (l.c3 & 3) || !l.w3 || l.dw_stride != 1 || l.dw_pad_t != 1 || l.dw_pad_l != 1)
          return fail(c, YL_ERR_UNSUPPORTED, "stem block with a depthwise second conv: 3 -> 32 -> dw3x3 s1 pad 1 -> 1x1 (4..32 outputs)");
      } else if (l.dw_k != 0) {
        return fail(c, YL_ERR_UNSUPPORTED, "stem block: dw_k must be 0 (dense 3x3 s2 second conv) or 3 (depthwise 3x3 s1)");
      } else if (!yl_stemblock_supported(l.cout, l.c2, l.c3))
        return fail(c, YL_ERR_UNSUPPORTED, "stem block: c1 in {16,32}, c2,c3 <= 32 and multiples of 4");
    } else if (l.op == YL_OP_STEM) {
      L.in_h = L.in_w = d->img_size;
      if (l.cin != 3 || l.k != 3) return fail(c, YL_ERR_UNSUPPORTED, "stem must be 3x3 with 3 input channels");
      if (l.cout != 16 && l.cout != 32) return fail(c, YL_ERR_UNSUPPORTED, "stem cout must be 16 or 32");
    } else {
      if (l.in_slot < 0 || l.in_slot >= d->num_slots) return bad("bad in_slot");
      L.in_h = c->slots[l.in_slot].h; L.in_w = c->slots[l.in_slot].w;
      if (l.in_shift != 0) {
        if (l.op != YL_OP_CONV || l.k < 2 || l.dw_k != 0 || l.in_shift < 0 || l.in_shift > 3)
          return bad("in_shift needs a kxk (k>1) conv without depthwise prologue");
        L.in_h <<= l.in_shift; L.in_w <<= l.in_shift;          // dims of the virtually upsampled input
      }
      const bool uib = (l.op == YL_OP_CONV && l.c2 > 0);
      if (c->slots[l.in_slot].c != (uib ? l.c2 : l.cin)) return bad("cin does not match the input slot");
      if (uib) {
        if (!l.w2 || l.dw_k == 0 || l.dw_stride < 1 || l.k != 1 || l.head_level >= 0)
          return bad("fused expand->depthwise->project block: needs w2, a depthwise prologue, 1x1 projection");
        // workgroup-level halo kernel (yl_ir_kernel: stride 1 / 2, TF-SAME pads) or the per-wave one (yl_uib_kernel)
        const bool ir = l.out_slot >= 0 && l.out_slot < d->num_slots &&
                        yl_ir_supported(l.c2, l.cin, l.cout, l.dw_k, l.dw_stride, c->slots[l.out_slot].h, c->slots[l.out_slot].w);
        if (!ir) {
          if (l.up_slot >= 0) return fail(c, YL_ERR_UNSUPPORTED, "fused block with an upsample-add: shape not instantiated");
          if (l.dw_stride != 1) return fail(c, YL_ERR_UNSUPPORTED, "fused inverted-residual block: stride-2 shape not instantiated");
          if (!yl_uib_supported(l.c2, l.cin, l.cout, l.dw_k))
            return fail(c, YL_ERR_UNSUPPORTED, "fused inverted-residual block: shape not instantiated / LDS budget exceeded");
          if ((c->slots[l.in_slot].h & 3) || (c->slots[l.in_slot].w & 3))
            return fail(c, YL_ERR_UNSUPPORTED, "fused inverted-residual block needs H,W multiples of 4");
        }
      }
    }
    // output geometry.  Sizes are declared by the host (slot / level dims); pad_t/pad_l are explicit
    // and the bottom/right padding is implied, so only reachability is checked here.
    if (l.op == YL_OP_CONV && l.dw_k > 0) {
      if (l.k != 1 || l.stride != 1) return fail(c, YL_ERR_UNSUPPORTED, "dw prologue needs a 1x1 stride-1 main conv");
      if (!l.dw_w) return bad("dw prologue weights are NULL");
      if (l.dw_stride < 1) return bad("bad dw_stride");
      if ((size_t)(l.dw_k * l.dw_k + 1) * l.cin * sizeof(float) > YL_DW_LDS_MAX) {
        // beyond the tap image of the generic depthwise-prologue kernels: only the streamed-tap kernel (yl_conv_dws_kernel) runs it
        const bool dws = l.out_slot >= 0 && l.out_slot < d->num_slots && l.c2 == 0 && l.c3 == 0 && l.scale_slot < 0 && l.head_level < 0 &&
                         yl_dws_supported(l.cin, l.cout, l.dw_k, l.dw_stride, c->slots[l.out_slot].h, c->slots[l.out_slot].w);
        if (!dws)
          return fail(c, YL_ERR_UNSUPPORTED, "dw prologue: taps+bias of all input channels must fit 32 KiB of LDS (or the layer must "
                                             "be one yl_query_dw_prologue reports as 2)");
      }
    }
    if (l.head_level >= 0) {
      if (l.op != YL_OP_CONV || l.head_level >= c->L) return bad("bad head_level");
      L.out_h = L.out_w = c->level_S[l.head_level];
      if (l.cout != c->E) return bad("head layers must have cout = 5+C+NM (one layer per anchor)");
      L.head_anchor = head_seen[l.head_level]++;
      if (L.head_anchor >= c->level_A[l.head_level]) return bad("more head layers than anchors for this level");
      if (l.res_slot >= 0 || l.up_slot >= 0) return bad("head layers take no residual/upsample input");
    } else {
      if (l.out_slot < 0 || l.out_slot >= d->num_slots) return bad("bad out_slot");
      L.out_h = c->slots[l.out_slot].h; L.out_w = c->slots[l.out_slot].w;
      const int oc = (l.op == YL_OP_STEMBLOCK) ? (l.c3 > 0 ? l.c3 : l.c2) : ((l.op == YL_OP_CONV && l.c3 > 0) ? l.c3 : l.cout);
      if (c->slots[l.out_slot].c != oc) return bad("cout does not match the output slot");
    }
    if (l.op == YL_OP_STEMBLOCK) {
      const int sh = (L.in_h + 2 * 0 + l.pad_t + (l.k - 1 - l.pad_t) - l.k) / l.stride + 1;   // symmetric / SAME stem
      if (L.out_h != (l.dw_k == 3 ? sh : (sh + 2 - 3) / 2 + 1) || L.out_w != L.out_h) return bad("stem block output size mismatch");
    } else {
      const bool pro = (l.op == YL_OP_CONV && l.dw_k > 0);
      const int st = pro ? l.dw_stride : l.stride, pt = pro ? l.dw_pad_t : l.pad_t, pl = pro ? l.dw_pad_l : l.pad_l;
      const int kk = pro ? l.dw_k : l.k;
      // the last window must start inside the tensor
      if ((L.out_h - 1) * st - pt >= L.in_h || (L.out_w - 1) * st - pl >= L.in_w || pt >= kk || pl >= kk ||
          L.out_h < 1 || L.out_w < 1)
        return bad("output size inconsistent with stride/padding");
    }
    if (l.res_slot >= 0) {
      if (l.res_slot >= d->num_slots) return bad("bad res_slot");
      const Slot& r = c->slots[l.res_slot];
      if (r.h != L.out_h || r.w != L.out_w || r.c != l.cout) return bad("residual shape mismatch");
    }
    if (l.up_slot >= 0) {
      if (l.up_slot >= d->num_slots || l.op != YL_OP_CONV) return bad("bad up_slot");
      // (fused block: the addend joins the EXPANDED tensor, cin channels)
      if (c->slots[l.up_slot].c != ((l.op == YL_OP_CONV && l.c2 > 0) ? l.cin : l.cout)) return bad("upsample source channel mismatch");
    }
    if (l.op == YL_OP_DW && l.cin != l.cout) return bad("depthwise needs cin == cout");
    if (l.op != YL_OP_STEM && l.op != YL_OP_STEMBLOCK && (l.cin & 3)) return fail(c, YL_ERR_UNSUPPORTED, "cin must be a multiple of 4");
    if ((l.res_slot >= 0 || l.up_slot >= 0 || YL_SMOOTH(l.act)) && (l.cout & 3) && l.op == YL_OP_CONV)
      return fail(c, YL_ERR_UNSUPPORTED, "residual/upsample/SiLU epilogue needs cout % 4 == 0");

    // ---- pack + upload
    std::vector<float> wp, bias;
    yl_status s;
    if (l.op == YL_OP_STEM || l.op == YL_OP_STEMBLOCK) {
      if (l.op == YL_OP_STEMBLOCK) pack_stem_rows(l.w, l.b, l.cout, wp);
      else pack_stem(l.w, l.cout, l.cin, l.k, wp);
      bias.assign(l.cout, 0.0f);
      if (l.b) memcpy(bias.data(), l.b, l.cout * sizeof(float));
      if (l.op == YL_OP_STEMBLOCK) {
        std::vector<float> w2, b2v((size_t)cdiv(l.c2, 16) * 16, 0.0f);
        if (l.dw_k == 3) {                                           // depthwise taps [c][1][3][3] -> tap-major [9][c]
          w2.assign((size_t)9 * l.c2, 0.0f);
          for (int ch = 0; ch < l.c2; ++ch)
            for (int t = 0; t < 9; ++t) w2[(size_t)t * l.c2 + ch] = l.w2[(size_t)ch * 9 + t];
        } else
        pack_conv(l.w2, l.c2, l.cout, 3, w2);
        if (l.b2) memcpy(b2v.data(), l.b2, l.c2 * sizeof(float));
        if ((s = upload(c, w2, &L.w2p)) != YL_OK) return s;
        if ((s = upload(c, b2v, &L.b2)) != YL_OK) return s;
        if (l.c3 > 0) {
          // the 1x1 conv's k-blocks are the second conv's 16-wide n-tiles: pack with cin padded to that
          std::vector<float> w3, b3v((size_t)cdiv(l.c3, 16) * 16, 0.0f);
          pack_conv(l.w3, l.c3, l.c2, 1, w3);
          if (l.b3) memcpy(b3v.data(), l.b3, l.c3 * sizeof(float));
          if ((s = upload(c, w3, &L.w3p)) != YL_OK) return s;
          if ((s = upload(c, b3v, &L.b3)) != YL_OK) return s;
        }
      }
    } else if (l.op == YL_OP_CONV) {
      pack_conv(l.w, l.cout, l.cin, l.k, wp);
      bias.assign((size_t)cdiv(l.cout, 16) * 16 + 128, 0.0f);
      if (l.b) memcpy(bias.data(), l.b, l.cout * sizeof(float));
      if (l.k == 3 && l.stride == 1 && l.dw_k == 0 && l.c2 == 0 && l.c3 == 0 && l.pad_t == 1 && l.pad_l == 1 && l.in_shift <= 1 &&
          l.cin >= 16 && l.cout >= 16 && (l.cout & 3) == 0 && l.head_level < 0 && l.up_slot < 0) {
        std::vector<float> wn;
        pack_wino(l.w, l.cout, l.cin, wn);
        if ((s = upload(c, wn, &L.wino)) != YL_OK) return s;
        if (l.cin >= 64 && l.cout >= 64 && L.out_h * L.out_w > c->wino_max_hw) c->wino_max_hw = L.out_h * L.out_w;
      }
      if (l.head_level >= 0 && c->NM > 0 && (c->NM & 3) == 0 && l.k == 1 && l.dw_k == 0 && l.c2 == 0 && l.c3 == 0 &&
          5 + c->C <= 96 && l.cout == c->E && c->level_A[l.head_level] == 1) {
        const int nd = 5 + c->C;
        std::vector<float> wd, wm, bd((size_t)cdiv(nd, 16) * 16 + 128, 0.0f), bm((size_t)cdiv(c->NM, 16) * 16 + 128, 0.0f);
        pack_conv(l.w, nd, l.cin, 1, wd);
        pack_conv(l.w + (size_t)nd * l.cin, c->NM, l.cin, 1, wm);
        if (l.b) { memcpy(bd.data(), l.b, nd * sizeof(float)); memcpy(bm.data(), l.b + nd, c->NM * sizeof(float)); }
        if ((s = upload(c, wd, &L.wp_det)) != YL_OK || (s = upload(c, bd, &L.b_det)) != YL_OK ||
            (s = upload(c, wm, &L.wp_mc)) != YL_OK || (s = upload(c, bm, &L.b_mc)) != YL_OK)
          return s;
      }
      if (l.c3 > 0) {       // chained 1x1 conv [c3][cout][1][1]: its k-blocks are this conv's 16-wide n-tiles
        if (!l.w3 || l.k < 2 || l.dw_k > 0 || l.c2 > 0 || l.head_level >= 0 || l.res_slot >= 0 || l.up_slot >= 0 || l.in_shift ||
            (l.cout & 3) || (l.c3 & 3) || l.c3 > 32 || l.cout > 96 || YL_SMOOTH(l.act) || YL_SMOOTH(l.act3))
          return fail(c, YL_ERR_UNSUPPORTED, "chained 1x1 conv: needs a plain dense k x k conv (<= 96 channels out), c3 <= 32, ReLU-family activations");
        std::vector<float> w3, b3v((size_t)cdiv(l.c3, 16) * 16, 0.0f);
        pack_conv(l.w3, l.c3, l.cout, 1, w3);
        if (l.b3) memcpy(b3v.data(), l.b3, l.c3 * sizeof(float));
        if ((s = upload(c, w3, &L.w3p)) != YL_OK) return s;
        if ((s = upload(c, b3v, &L.b3)) != YL_OK) return s;
      }
      if (l.c2 > 0) {       // expansion conv of a fused inverted-residual block: [cin][c2][1][1]
        std::vector<float> w2, b2v((size_t)cdiv(l.cin, 16) * 16, 0.0f);
        pack_conv(l.w2, l.cin, l.c2, 1, w2);
        if (l.b2) memcpy(b2v.data(), l.b2, l.cin * sizeof(float));
        if ((s = upload(c, w2, &L.w2p)) != YL_OK) return s;
        if ((s = upload(c, b2v, &L.b2)) != YL_OK) return s;
      }
      if (l.dw_k > 0) {
        std::vector<float> dw, dwb;
        pack_dw(l.dw_w, l.cin, l.dw_k, dw);
        if ((s = upload(c, dw, &L.dw_w)) != YL_OK) return s;
        if (l.dw_b) {
          dwb.assign(l.dw_b, l.dw_b + l.cin);
          if ((s = upload(c, dwb, &L.dw_b)) != YL_OK) return s;
        }
      }
    } else {
      pack_dw(l.w, l.cout, l.k, wp);
      if (l.b) bias.assign(l.b, l.b + l.cout);
    }
    if ((s = upload(c, wp, &L.wp)) != YL_OK) return s;
    if ((s = upload(c, bias, &L.bias)) != YL_OK) return s;
    L.d.w = L.d.b = L.d.dw_w = L.d.dw_b = nullptr;
    L.d.w2 = L.d.b2 = L.d.w3 = L.d.b3 = nullptr;
    c->layers.push_back(L);
  }
  for (int l = 0; l < c->L; ++l)
    if (head_seen[l] != c->level_A[l]) return fail(c, YL_ERR_INVALID, "every level needs one head layer per anchor");
  if (c->NM > 0) {
    if (c->proto_slot < 0 || c->proto_slot >= d->num_slots || c->slots[c->proto_slot].c != c->NM)
      return fail(c, YL_ERR_INVALID, "num_masks > 0 needs proto_slot with num_masks channels");
  }
  assign_lanes(c);
  assign_small_run(c);
  return YL_OK;
}

yl_status yl_set_option(yl_ctx* c, const char* name, int32_t value) {
  if (!c || !name) return YL_ERR_INVALID;
  if (!strcmp(name, "graph")) { c->opt_graph = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "mfma_bf16")) { if (c->opt_bf16 == 3) free_act(c); c->opt_bf16 = value ? 1 : (c->opt_bf16 == 1 ? 0 : c->opt_bf16); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "mfma_f16")) { if (c->opt_bf16 == 3) free_act(c); c->opt_bf16 = value ? 2 : (c->opt_bf16 == 2 ? 0 : c->opt_bf16); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "store_f16")) {      // fp16 operands AND fp16 activation tensors in HBM (round 6): the arenas are re-planned
    const int nv = value ? 3 : (c->opt_bf16 == 3 ? 0 : c->opt_bf16);
    if ((nv == 3) != (c->opt_bf16 == 3)) free_act(c);
    c->opt_bf16 = nv; drop_graph(c); return YL_OK;
  }
  if (!strcmp(name, "nms_groups")) { c->opt_nms_groups = value < 0 ? 0 : (value > YL_NMS_GROUPS ? YL_NMS_GROUPS : value); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "time_split")) {
    // (no drop_graph: the setting is part of the graph key -- a serving loop toggles it per call, api.YoloLite.predict)
    c->opt_time_split = value ? 1 : 0;
    for (int i = 0; i < 3 && value; ++i)
      if (!c->ev_t[i]) HIPCHK(c, hipEventCreate(&c->ev_t[i]));
    c->timing_valid = false;
    return YL_OK;
  }
  if (!strcmp(name, "pre_norm")) { c->opt_pre_norm = value ? 1 : 0; return YL_OK; }
  if (!strcmp(name, "reuse_slots")) { c->opt_reuse = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "hybrid")) { c->opt_hybrid = value < 0 ? 0 : (value > 2 ? 2 : value); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "batch_levels")) { c->opt_batch_levels = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "fuse_decode")) { c->opt_fuse_decode = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "fuse_head")) { c->opt_fuse_head = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "winograd")) { c->opt_winograd = value < 0 ? 0 : (value > 2 ? 2 : value); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "lanes")) { c->opt_lanes = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "tile_m")) { c->opt_tile_m = value; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "streams")) { c->opt_streams = value < 1 ? 1 : (value > 4 ? 4 : value); drop_graph(c); return YL_OK; }
  if (!strcmp(name, "split_k")) { c->opt_split_k = value ? 1 : 0; drop_graph(c); return YL_OK; }
  if (!strcmp(name, "dev_select")) { c->opt_dev = value & 0x3ffffff; drop_graph(c); return YL_OK; }
  return fail(c, YL_ERR_INVALID, std::string("unknown option ") + name);
}

yl_status yl_get_option(const yl_ctx* c, const char* name, int32_t* value) {
  if (!c || !name || !value) return YL_ERR_INVALID;
  const struct { const char* n; int v; } tab[] = {
      {"graph", c->opt_graph}, {"mfma_bf16", c->opt_bf16 == 1}, {"mfma_f16", c->opt_bf16 == 2}, {"store_f16", c->opt_bf16 == 3}, {"nms_groups", c->opt_nms_groups}, {"time_split", c->opt_time_split},
      {"pre_norm", c->opt_pre_norm}, {"reuse_slots", c->opt_reuse}, {"hybrid", c->opt_hybrid}, {"batch_levels", c->opt_batch_levels},
      {"fuse_decode", c->opt_fuse_decode}, {"fuse_head", c->opt_fuse_head}, {"winograd", c->opt_winograd}, {"lanes", c->opt_lanes},
      {"tile_m", c->opt_tile_m}, {"streams", c->opt_streams}, {"dev_select", c->opt_dev}, {"split_k", c->opt_split_k}};
  for (const auto& t : tab)
    if (!strcmp(name, t.n)) { *value = t.v; return YL_OK; }
  return YL_ERR_INVALID;
}

int32_t yl_query_fused_block(int32_t c_in, int32_t c_mid, int32_t c_out, int32_t dw_k, int32_t dw_stride, int32_t out_h,
                             int32_t out_w) {
  if (c_in < 1 || c_mid < 1 || c_out < 1 || out_h < 1 || out_w < 1) return 0;
  if (yl_ir_supported(c_in, c_mid, c_out, dw_k, dw_stride, out_h, out_w)) return 1;
  // per-wave kernel: stride 1 (input grid == output grid), grids multiples of 4 (the checks of yl_create)
  if (dw_stride == 1 && !(out_h & 3) && !(out_w & 3) && yl_uib_supported(c_in, c_mid, c_out, dw_k)) return 2;
  return 0;
}

int32_t yl_query_dw_prologue(int32_t c_in, int32_t c_out, int32_t dw_k, int32_t dw_stride, int32_t out_h, int32_t out_w) {
  if (c_in < 1 || c_out < 1 || dw_k < 1 || dw_stride < 1 || out_h < 1 || out_w < 1) return 0;
  if (yl_dws_supported(c_in, c_out, dw_k, dw_stride, out_h, out_w)) return 2;
  if ((size_t)(dw_k * dw_k + 1) * c_in * sizeof(float) <= YL_DW_LDS_MAX) return 1;
  return 0;
}

static yl_status forward_impl(yl_ctx* c, const float* x, int B, float* const* level_out, hipStream_t st,
                              float* layer_ms, const yl_post_cfg* cfg, float* dets, int* counts, int* keep_idx = nullptr) {
  if (!c) return YL_ERR_INVALID;
  if (c->layers.empty()) return fail(c, YL_ERR_STATE, "context was created without layers");
  if (!x || B < 1) return fail(c, YL_ERR_INVALID, "bad input");
  HIPCHK(c, hipSetDevice(c->device));
  yl_status s = ensure_act(c, B, layer_ms ? 1 : chunks_for(c, B));
  if (s != YL_OK) return s;
  if (cfg && (s = ensure_post(c, B)) != YL_OK) return s;
  Job j;
  j.x = x; j.B = B; j.cfg = cfg; j.dets = dets; j.counts = counts; j.keep_idx = keep_idx;
  for (int l = 0; l < c->L; ++l) j.outs[l] = (level_out && level_out[l]) ? level_out[l] : c->level_buf[l];
  if (layer_ms) {        // measurement path: one stream, one chunk, an event pair around every launch
    std::vector<hipEvent_t> ev(c->layers.size() + 1);
    for (auto& e : ev) HIPCHK(c, hipEventCreate(&e));
    s = run_layers(c, x, 0, B, j.outs, st, ev.data());
    if (s == YL_OK) {
      HIPCHK(c, hipStreamSynchronize(st));
      for (size_t i = 0; i < c->layers.size(); ++i) hipEventElapsedTime(&layer_ms[i], ev[i], ev[i + 1]);
    }
    for (auto& e : ev) hipEventDestroy(e);
    return s;
  }
  return submit(c, j, st);
}

yl_status yl_forward(yl_ctx* c, const float* x, int32_t B, float* const* level_out, void* stream) {
  return forward_impl(c, x, B, level_out, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr);
}

yl_status yl_forward_timed(yl_ctx* c, const float* x, int32_t B, float* const* level_out, void* stream,
                           float* layer_ms) {
  if (!layer_ms) return YL_ERR_INVALID;
  return forward_impl(c, x, B, level_out, (hipStream_t)stream, layer_ms, nullptr, nullptr, nullptr);
}

yl_status yl_last_timing(yl_ctx* c, float* infer_ms, float* post_ms) {
  if (!c || !infer_ms || !post_ms) return YL_ERR_INVALID;
  if (!c->opt_time_split || !c->timing_valid) return fail(c, YL_ERR_STATE, "set option time_split and call yl_predict first");
  HIPCHK(c, hipEventSynchronize(c->ev_t[2]));
  HIPCHK(c, hipEventElapsedTime(infer_ms, c->ev_t[0], c->ev_t[1]));
  HIPCHK(c, hipEventElapsedTime(post_ms, c->ev_t[1], c->ev_t[2]));
  return YL_OK;
}

int64_t yl_activation_bytes(const yl_ctx* c) {
  if (!c || c->plan_n < 1) return 0;
  int64_t t = 0;
  for (int i = 0; i < c->arena_n; ++i) t += (int64_t)(c->arena_unit + c->se_unit * sizeof(float)) * c->arena_capi[i];
  for (const auto& s : c->slots)
    if (s.pinned) t += (int64_t)s.sz * c->cap_batch;
  return t;
}

yl_status yl_read_slot(yl_ctx* c, int32_t slot, int32_t B, float* dst, void* stream) {
  if (!c || !dst) return YL_ERR_INVALID;
  if (slot < 0 || slot >= (int)c->slots.size()) return fail(c, YL_ERR_INVALID, "bad slot");
  if (B > c->act_batch || c->plan_n < 1) return fail(c, YL_ERR_STATE, "no forward of >= B images has produced this slot");
  HIPCHK(c, hipSetDevice(c->device));
  const Slot& s = c->slots[slot];
  if (c->opt_bf16 == 3 && !s.pinned) return fail(c, YL_ERR_UNSUPPORTED, "store_f16: activation slots are fp16 (yl_read_slot hands out fp32)");
  // (with "reuse_slots" on, a tensor that is not an output of the network may have been overwritten by later layers)
  for (int i = 0; i < c->plan_n; ++i) {
    const int b0 = c->plan_b0[i], b1 = c->plan_b0[i + 1] < B ? c->plan_b0[i + 1] : B;
    if (b1 <= b0) break;
    HIPCHK(c, hipMemcpyAsync((char*)dst + (size_t)b0 * s.sz, slot_addr(c, slot, b0), (size_t)(b1 - b0) * s.sz,
                             hipMemcpyDeviceToDevice, (hipStream_t)stream));
  }
  return YL_OK;
}

yl_status yl_preprocess(yl_ctx* c, const uint8_t* packed, const yl_pre_image* imgs, int32_t B, float* x, void* stream) {
  if (!c || !packed || !imgs || !x || B < 1) return YL_ERR_INVALID;
  HIPCHK(c, hipSetDevice(c->device));
  static_assert(sizeof(yl_pre_image) == 32, "yl_pre_image layout");
  HIPCHK(c, yl_launch_preprocess(packed, imgs, B, c->img_size, x, c->opt_pre_norm, (hipStream_t)stream));
  return YL_OK;
}

yl_status yl_decode(yl_ctx* c, const float* const* levels, int32_t B, int32_t center_mode, int32_t wh_mode,
                    float* box, float* obj, float* cls, void* stream) {
  if (!c || !levels || !box || !obj || B < 1) return YL_ERR_INVALID;
  if (c->C > 0 && !cls) return fail(c, YL_ERR_INVALID, "cls_dev is NULL");
  if (center_mode < 0 || center_mode > 1 || wh_mode < 0 || wh_mode > 2) return fail(c, YL_ERR_INVALID, "bad mode");
  HIPCHK(c, hipSetDevice(c->device));
  YlLevels lv;
  fill_levels(c, levels, lv);
  HIPCHK(c, yl_launch_decode_only(lv, B, center_mode, wh_mode, box, obj, cls, (hipStream_t)stream));
  return YL_OK;
}

yl_status yl_forward_decoded(yl_ctx* c, const float* x, int32_t B, int32_t center_mode, int32_t wh_mode, float* box,
                             float* obj, float* cls, void* stream) {
  if (!c || !box || !obj) return YL_ERR_INVALID;
  if (c->C > 0 && !cls) return fail(c, YL_ERR_INVALID, "cls_dev is NULL");
  if (center_mode < 0 || center_mode > 1 || wh_mode < 0 || wh_mode > 2) return fail(c, YL_ERR_INVALID, "bad mode");
  // the raw levels go to the context's own buffers (two chunk streams, hipGraph: as yl_forward), the decode follows on
  // the caller's stream behind the join
  yl_status s = forward_impl(c, x, B, nullptr, (hipStream_t)stream, nullptr, nullptr, nullptr, nullptr);
  if (s != YL_OK) return s;
  YlLevels lv;
  fill_levels(c, c->level_buf, lv);
  HIPCHK(c, yl_launch_decode_only(lv, B, center_mode, wh_mode, box, obj, cls, (hipStream_t)stream));
  return YL_OK;
}

yl_status yl_postprocess(yl_ctx* c, const float* const* levels, int32_t B, const yl_post_cfg* cfg, float* dets,
                         int32_t* counts, int32_t* keep_idx, void* stream) {
  if (!c || !levels || !dets || !counts || B < 1) return YL_ERR_INVALID;
  yl_status s = check_cfg(c, cfg);
  if (s != YL_OK) return s;
  HIPCHK(c, hipSetDevice(c->device));
  if ((s = ensure_post(c, B)) != YL_OK) return s;
  Job j;
  j.B = B; j.cfg = cfg; j.dets = dets; j.counts = counts; j.keep_idx = keep_idx;
  for (int l = 0; l < c->L; ++l) j.outs[l] = const_cast<float*>(levels[l]);
  return submit(c, j, (hipStream_t)stream, false);
}

yl_status yl_predict(yl_ctx* c, const float* x, int32_t B, const yl_post_cfg* cfg, float* dets, int32_t* counts,
                     int32_t* keep_idx, void* stream) {
  if (!c || !dets || !counts) return YL_ERR_INVALID;
  yl_status s = check_cfg(c, cfg);
  if (s != YL_OK) return s;
  return forward_impl(c, x, B, nullptr, (hipStream_t)stream, nullptr, cfg, dets, counts, keep_idx);
}

yl_status yl_masks(yl_ctx* c, const float* const* levels, int32_t B, const int32_t* counts, const int32_t* keep_idx,
                   int32_t max_out, float thr, uint8_t* masks, void* stream) {
  if (!c || !counts || !keep_idx || !masks || B < 1 || max_out < 1) return YL_ERR_INVALID;
  if (c->NM <= 0 || c->proto_slot < 0) return fail(c, YL_ERR_STATE, "context has no mask branch");
  if (B > c->act_batch || B > c->post_cap_batch || !c->slots[c->proto_slot].pin)
    return fail(c, YL_ERR_STATE, "yl_masks needs a preceding yl_predict / yl_forward+yl_postprocess of this batch");
  HIPCHK(c, hipSetDevice(c->device));
  const float* lp[YL_MAX_LEVELS];
  for (int l = 0; l < c->L; ++l) lp[l] = (levels && levels[l]) ? levels[l] : c->level_buf[l];
  YlLevels lv;
  fill_levels(c, lp, lv);
  const Slot& ps = c->slots[c->proto_slot];
  HIPCHK(c, yl_launch_masks(lv, B, ps.pin, ps.h, ps.w, c->NM, c->img_size, c->ws_boxes, counts, keep_idx, max_out, thr,
                            masks, (hipStream_t)stream));
  return YL_OK;
}

yl_status yl_masks_image(yl_ctx* c, const float* const* levels, int32_t B, const float* dets, const int32_t* counts,
                         const int32_t* keep_idx, int32_t max_out, float thr, const float* backmap,
                         const int32_t* out_hw, const int64_t* mask_off, int32_t max_h, int32_t max_w, int32_t packed,
                         uint8_t* masks, void* stream) {
  if (!c || !dets || !counts || !keep_idx || !out_hw || !mask_off || !masks || B < 1 || max_out < 1 || max_h < 1 || max_w < 1)
    return YL_ERR_INVALID;
  if (c->NM <= 0 || c->proto_slot < 0) return fail(c, YL_ERR_STATE, "context has no mask branch");
  if (B > c->act_batch || !c->slots[c->proto_slot].pin)
    return fail(c, YL_ERR_STATE, "yl_masks_image needs a preceding yl_predict / yl_forward of this batch");
  HIPCHK(c, hipSetDevice(c->device));
  const float* lp[YL_MAX_LEVELS];
  for (int l = 0; l < c->L; ++l) lp[l] = (levels && levels[l]) ? levels[l] : c->level_buf[l];
  YlLevels lv;
  fill_levels(c, lp, lv);
  const Slot& ps = c->slots[c->proto_slot];
  static_assert(sizeof(long long) == sizeof(int64_t), "offset type");
  HIPCHK(c, yl_launch_masks_image(lv, B, ps.pin, ps.h, ps.w, c->NM, c->img_size, dets, counts, keep_idx, max_out, thr,
                                  backmap, out_hw, (const long long*)mask_off, max_h, max_w, packed ? 1 : 0, masks,
                                  (hipStream_t)stream));
  return YL_OK;
}

// the one exchange of the multi-GPU path, for hosts that do not go through torch.distributed.  RCCL is
// NOT a link-time dependency of this library: ncclAllGather is looked up among the libraries already loaded in the
// process, i.e. the RCCL that created the caller's communicator (PyTorch ships its own copy).
yl_status yl_allgather_dets(yl_ctx* c, void* comm, const float* local_dev, int64_t row_floats, float* all_dev, void* stream) {
  if (!c || !comm || !local_dev || !all_dev || row_floats <= 0) return c ? fail(c, YL_ERR_INVALID, "bad argument") : YL_ERR_INVALID;
  typedef int (*allgather_fn)(const void*, void*, size_t, int /*ncclDataType_t*/, void* /*ncclComm_t*/, hipStream_t);
  static allgather_fn fn = nullptr;
  if (!fn) fn = (allgather_fn)dlsym(RTLD_DEFAULT, "ncclAllGather");
  if (!fn) return fail(c, YL_ERR_UNSUPPORTED, "ncclAllGather not found: load RCCL (librccl.so) in this process first");
  HIPCHK(c, hipSetDevice(c->device));
  const int rc = fn(local_dev, all_dev, (size_t)row_floats, 7 /*ncclFloat32*/, comm, (hipStream_t)stream);
  if (rc != 0) {
    char b[96];
    snprintf(b, sizeof(b), "ncclAllGather failed with ncclResult_t %d", rc);
    return fail(c, YL_ERR_HIP, b);
  }
  return YL_OK;
}

yl_status yl_nms(yl_ctx* c, const float* boxes, const float* scores, int32_t n, float iou_thr, int32_t impl,
                 int32_t max_det, int32_t* keep, int32_t* count, void* stream) {
  if (!c || !boxes || !scores || !keep || !count || n < 0 || max_det < 1) return YL_ERR_INVALID;
  if (impl != YL_NMS_TORCHVISION && impl != YL_NMS_GREEDY) return fail(c, YL_ERR_INVALID, "bad nms_impl");
  if (n >= (1 << 20)) return fail(c, YL_ERR_UNSUPPORTED, "n must be < 2^20");
  HIPCHK(c, hipSetDevice(c->device));
  hipStream_t st = (hipStream_t)stream;
  if (n == 0) {
    HIPCHK(c, hipMemsetAsync(count, 0, sizeof(int), st));
    return YL_OK;
  }
  if (!c->ws_nms_clsws) HIPCHK(c, hipMalloc((void**)&c->ws_nms_clsws, 4 * sizeof(int)));
  const int P = pow2ceil(n);
  if (P > YL_LDS_KEYS_MAX && P > c->nms_gP) {
    hipFree(c->ws_nms_gkeys);
    c->ws_nms_gkeys = nullptr;
    HIPCHK(c, hipMalloc((void**)&c->ws_nms_gkeys, (size_t)P * 8));
    c->nms_gP = P;
  }
  // dets rows are not wanted here: reuse the top-k staging path's buffers? No -- a scratch dets
  // buffer of max_det rows is taken from the tmp area sized for n rows.
  float* scratch = nullptr;
  HIPCHK(c, hipMallocAsync((void**)&scratch, (size_t)max_det * 6 * sizeof(float), st));
  YlNmsP np;
  memset(&np, 0, sizeof(np));
  np.boxes = (const float4*)boxes; np.scores = scores; np.cls = nullptr;
  np.N = n; np.C = 1;
  np.conf_thr = -INFINITY; np.iou_thr = iou_thr; np.impl = impl;
  np.cap = max_det; np.topk = 0; np.max_out = max_det;
  np.cls_ws = c->ws_nms_clsws;
  np.gkeys = c->ws_nms_gkeys; np.gP = c->nms_gP;
  np.lds_cap = P < YL_LDS_KEYS_MAX ? P : YL_LDS_KEYS_MAX;
  np.dets = scratch; np.counts = count; np.keep_idx = keep; np.backmap = nullptr;
  hipError_t e = yl_launch_nms(np, 1, st);
  hipFreeAsync(scratch, st);
  HIPCHK(c, e);
  return YL_OK;
}

}  // extern "C"
