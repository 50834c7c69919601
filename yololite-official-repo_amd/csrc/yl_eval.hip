// Evaluate-path consumers on the device: the O(detections x ground truths)
// greedy matching loops of the reference's evaluation.
//   yl_eval_match      <- build_curves_from_coco     scripts/data/p_r_f1.py:31-78, :100-118  (float64)
//   yl_eval_sweep      <- its 0..1 confidence sweep  scripts/data/p_r_f1.py:96-124
//   yl_eval_confusion  <- create_confusion_matrix    scripts/helpers/evaluate.py:23-57, :96-153 (float32)
// Compiled with -ffp-contract=off: every + - * / is the IEEE operation the reference's python / numpy
// arithmetic performs, in the same order, so the match decisions are bit-identical.
//
// Work decomposition: the matching of one key (image, category) -- or one image for the confusion
// matrix -- is inherently sequential over its score-ordered detections, and independent of every other
// key.  One 64-lane wave owns a key: detections in order, lanes across the ground truths (strided when
// there are more than 64), a butterfly reduction for (max IoU, first index).  The matched flag of a
// ground truth is written by the lane that will read it again (lane = index & 63): no fences.  The
// kernels are latency-bound integer/compare work on a few KB per key; they exist so the evaluation
// loop never leaves the device, not to fill the machine.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "../../include/yololite_hip.h"

namespace {

// ---- (max value, min index) butterfly over the 64 lanes; every lane ends with the result
template <typename T>
__device__ __forceinline__ void yl_wave_argmax(T& v, int& j) {
#pragma unroll
  for (int m = 32; m >= 1; m >>= 1) {
    const T ov = __shfl_xor(v, m, 64);
    const int oj = __shfl_xor(j, m, 64);
    if (ov > v || (ov == v && oj < j)) { v = ov; j = oj; }
  }
}

// iou_xywh, p_r_f1.py:31-41 (python floats = IEEE double)
__device__ __forceinline__ double yl_iou_xywh(double ax, double ay, double aw, double ah, double bx, double by,
                                              double bw, double bh) {
  const double ax2 = ax + aw, ay2 = ay + ah;
  const double bx2 = bx + bw, by2 = by + bh;
  const double ix1 = fmax(ax, bx), iy1 = fmax(ay, by);
  const double ix2 = fmin(ax2, bx2), iy2 = fmin(ay2, by2);
  const double iw = fmax(0.0, ix2 - ix1), ih = fmax(0.0, iy2 - iy1);
  const double inter = iw * ih;
  const double ua = fmax(0.0, aw * ah) + fmax(0.0, bw * bh) - inter;
  return ua > 0.0 ? inter / ua : 0.0;
}

__global__ __launch_bounds__(256) void yl_eval_match_kernel(const double* __restrict__ det,
                                                            const int* __restrict__ det_off,
                                                            const double* __restrict__ gt,
                                                            const int* __restrict__ gt_off, int num_keys, double thr,
                                                            uint8_t* __restrict__ tp, int* __restrict__ match,
                                                            uint8_t* gt_matched) {
  const int lane = threadIdx.x & 63;
  const int key = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (key >= num_keys) return;
  const int d0 = det_off[key], d1 = det_off[key + 1];
  const int g0 = gt_off[key], ng = gt_off[key + 1] - g0;
  for (int d = d0; d < d1; ++d) {
    const double ax = det[4 * (size_t)d], ay = det[4 * (size_t)d + 1];
    const double aw = det[4 * (size_t)d + 2], ah = det[4 * (size_t)d + 3];
    double best = 0.0;                 // p_r_f1.py:67 -- only an IoU strictly above 0 can be "best"
    int bj = 0x7fffffff;
    for (int j = lane; j < ng; j += 64) {
      if (gt_matched[g0 + j]) continue;
      const double* g = gt + 4 * (size_t)(g0 + j);
      const double v = yl_iou_xywh(ax, ay, aw, ah, g[0], g[1], g[2], g[3]);
      if (v > best) { best = v; bj = j; }       // strict: the first index keeps a tie (:72-73)
    }
    yl_wave_argmax(best, bj);
    const bool hit = bj != 0x7fffffff && best >= thr;          // :74
    if (hit && lane == (bj & 63)) gt_matched[g0 + bj] = 1;
    if (lane == 0) {
      tp[d] = hit ? 1 : 0;
      if (match) match[d] = hit ? bj : -1;
    }
  }
}

// histogram of the tp / fp flags over the threshold bins: bin(s) = largest k with thr[k] <= s
__global__ void yl_eval_hist_kernel(const double* __restrict__ score, const uint8_t* __restrict__ tp,
                                    const uint8_t* __restrict__ counted, int n, const double* __restrict__ thr,
                                    int steps, int* tp_hist, int* fp_hist) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  if (counted && !counted[i]) return;
  const double s = score[i];
  int lo = 0, hi = steps;              // first index with thr[idx] > s   (score >= thr, p_r_f1.py:104)
  while (lo < hi) {
    const int mid = (lo + hi) >> 1;
    if (thr[mid] <= s) lo = mid + 1; else hi = mid;
  }
  if (lo == 0) return;                 // below every threshold (or NaN): never counted
  atomicAdd(tp[i] ? &tp_hist[lo - 1] : &fp_hist[lo - 1], 1);
}

// in-place suffix sums of both histograms (steps is a few hundred: one block, serial tail is fine)
__global__ void yl_eval_suffix_kernel(int* tp_hist, int* fp_hist, int steps) {
  if (threadIdx.x == 0) {
    int a = 0;
    for (int k = steps - 1; k >= 0; --k) { a += tp_hist[k]; tp_hist[k] = a; }
  } else if (threadIdx.x == 64) {
    int a = 0;
    for (int k = steps - 1; k >= 0; --k) { a += fp_hist[k]; fp_hist[k] = a; }
  }
}

// iou_matrix, evaluate.py:27-57 (numpy float32)
__device__ __forceinline__ float yl_iou_xyxy_f32(const float* a, const float* b) {
  const float ix1 = fmaxf(a[0], b[0]), iy1 = fmaxf(a[1], b[1]);
  const float ix2 = fminf(a[2], b[2]), iy2 = fminf(a[3], b[3]);
  const float iw = fmaxf(ix2 - ix1, 0.0f), ih = fmaxf(iy2 - iy1, 0.0f);
  const float inter = iw * ih;
  const float area1 = (a[2] - a[0]) * (a[3] - a[1]);
  const float area2 = (b[2] - b[0]) * (b[3] - b[1]);
  float uni = area1 + area2 - inter;
  uni = fmaxf(uni, 1e-6f);
  return inter / uni;
}

__global__ __launch_bounds__(256) void yl_eval_confusion_kernel(const float* __restrict__ det,
                                                                const int* __restrict__ det_cls,
                                                                const int* __restrict__ det_off,
                                                                const float* __restrict__ gt,
                                                                const int* __restrict__ gt_cls,
                                                                const int* __restrict__ gt_off, int num_images,
                                                                int C, float thr, int* cm, uint8_t* gt_matched) {
  const int lane = threadIdx.x & 63;
  const int img = blockIdx.x * 4 + (threadIdx.x >> 6);
  if (img >= num_images) return;
  const int d0 = det_off[img], d1 = det_off[img + 1];
  const int g0 = gt_off[img], ng = gt_off[img + 1] - g0;
  const int W = C + 1;
  for (int d = d0; d < d1; ++d) {
    float a[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) a[r] = det[4 * (size_t)d + r];
    // np.argmax over the whole row (matched or not): first index of the maximum (:126-128)
    float best = -INFINITY;
    int bj = 0x7fffffff;
    for (int j = lane; j < ng; j += 64) {
      const float v = yl_iou_xyxy_f32(a, gt + 4 * (size_t)(g0 + j));
      if (v > best) { best = v; bj = j; }
    }
    yl_wave_argmax(best, bj);
    if (ng == 0) {                                                    // :119-123 (not reachable: images have GT)
      if (lane == 0) atomicAdd(&cm[C * W + det_cls[d]], 1);
      continue;
    }
    if (bj == 0x7fffffff) bj = 0;                                     // all-NaN row: argmax returns the first NaN
    const bool owner = lane == (bj & 63);
    int hit = 0;
    if (owner) {
      hit = (best >= thr && !gt_matched[g0 + bj]) ? 1 : 0;            // :130
      if (hit) {
        gt_matched[g0 + bj] = 1;
        atomicAdd(&cm[gt_cls[g0 + bj] * W + det_cls[d]], 1);          // true positive: (gt class, det class)
      } else {
        atomicAdd(&cm[C * W + det_cls[d]], 1);                        // false positive: (background, det class)
      }
    }
  }
  for (int j = lane; j < ng; j += 64)                                 // false negatives (:143-149)
    if (!gt_matched[g0 + j]) atomicAdd(&cm[gt_cls[g0 + j] * W + C], 1);
}

inline yl_status yl_hip(hipError_t e) { return e == hipSuccess ? YL_OK : YL_ERR_HIP; }

}  // namespace

extern "C" {

yl_status yl_eval_match(const double* det_xywh_dev, const int32_t* det_off_dev, const double* gt_xywh_dev,
                        const int32_t* gt_off_dev, int32_t num_keys, int32_t num_gt, double iou_thr, uint8_t* tp_dev,
                        int32_t* match_dev, uint8_t* gt_matched_dev, void* stream) {
  if (num_keys < 0 || num_gt < 0) return YL_ERR_INVALID;
  if (num_keys == 0) return YL_OK;
  if (!det_off_dev || !gt_off_dev || !tp_dev || (num_gt > 0 && (!gt_xywh_dev || !gt_matched_dev)))
    return YL_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (num_gt > 0) {
    const hipError_t e = hipMemsetAsync(gt_matched_dev, 0, (size_t)num_gt, st);
    if (e != hipSuccess) return YL_ERR_HIP;
  }
  hipLaunchKernelGGL(yl_eval_match_kernel, dim3((num_keys + 3) / 4), dim3(256), 0, st, det_xywh_dev, det_off_dev,
                     gt_xywh_dev, gt_off_dev, num_keys, iou_thr, tp_dev, match_dev, gt_matched_dev);
  return yl_hip(hipGetLastError());
}

yl_status yl_eval_sweep(const double* score_dev, const uint8_t* tp_dev, const uint8_t* counted_dev, int32_t n,
                        const double* thr_dev, int32_t steps, int32_t* tp_ge_dev, int32_t* fp_ge_dev, void* stream) {
  if (n < 0 || steps <= 0 || !thr_dev || !tp_ge_dev || !fp_ge_dev) return YL_ERR_INVALID;
  if (n > 0 && (!score_dev || !tp_dev)) return YL_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  if (hipMemsetAsync(tp_ge_dev, 0, sizeof(int32_t) * (size_t)steps, st) != hipSuccess) return YL_ERR_HIP;
  if (hipMemsetAsync(fp_ge_dev, 0, sizeof(int32_t) * (size_t)steps, st) != hipSuccess) return YL_ERR_HIP;
  if (n > 0)
    hipLaunchKernelGGL(yl_eval_hist_kernel, dim3((n + 255) / 256), dim3(256), 0, st, score_dev, tp_dev, counted_dev, n,
                       thr_dev, steps, tp_ge_dev, fp_ge_dev);
  hipLaunchKernelGGL(yl_eval_suffix_kernel, dim3(1), dim3(128), 0, st, tp_ge_dev, fp_ge_dev, steps);
  return yl_hip(hipGetLastError());
}

yl_status yl_eval_confusion(const float* det_xyxy_dev, const int32_t* det_cls_dev, const int32_t* det_off_dev,
                            const float* gt_xyxy_dev, const int32_t* gt_cls_dev, const int32_t* gt_off_dev,
                            int32_t num_images, int32_t num_gt, int32_t num_classes, float iou_thr, int32_t* cm_dev,
                            uint8_t* gt_matched_dev, void* stream) {
  if (num_images < 0 || num_gt < 0 || num_classes <= 0 || !cm_dev) return YL_ERR_INVALID;
  hipStream_t st = (hipStream_t)stream;
  const size_t W = (size_t)num_classes + 1;
  if (hipMemsetAsync(cm_dev, 0, sizeof(int32_t) * W * W, st) != hipSuccess) return YL_ERR_HIP;
  if (num_images == 0) return YL_OK;
  if (!det_off_dev || !gt_off_dev || (num_gt > 0 && (!gt_xyxy_dev || !gt_cls_dev || !gt_matched_dev)))
    return YL_ERR_INVALID;
  if (num_gt > 0 && hipMemsetAsync(gt_matched_dev, 0, (size_t)num_gt, st) != hipSuccess) return YL_ERR_HIP;
  hipLaunchKernelGGL(yl_eval_confusion_kernel, dim3((num_images + 3) / 4), dim3(256), 0, st, det_xyxy_dev,
                     det_cls_dev, det_off_dev, gt_xyxy_dev, gt_cls_dev, gt_off_dev, num_images, num_classes, iou_thr,
                     cm_dev, gt_matched_dev);
  return yl_hip(hipGetLastError());
}

}  // extern "C"
