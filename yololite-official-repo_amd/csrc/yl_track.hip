// Kalman-SORT tracker bank on the device (reference tools/tracker.py:9-326).
//
// The reference tracks ONE video stream with python lists of KalmanFilter objects (7-state constant
// velocity filter on [cx,cy,s,r], numpy float32) and a greedy global IoU assignment.  Here a *bank* of S
// independent streams lives in HBM (structure of arrays, capacity `max_tracks` per stream) and one
// 256-thread workgroup advances one stream per frame, directly on the packed detections yl_predict /
// yl_postprocess leave on the device ([S][max_out][6] = x1,y1,x2,y2,score,class + counts): a B-camera
// batch is tracked without the detections ever visiting the host.
//
// Per stream and frame (tools/tracker.py:211-326):
//   1. predict every track: x <- F x, P <- F P F^T + Q.  F is I plus three ones, so every element is at
//      most one float add -- identical to numpy's sgemm result bit for bit.            (:223-227, :123-125)
//   2. greedy assignment: repeatedly take the largest IoU(track box, detection box) * [same class]
//      among unmatched rows/columns until it drops below iou_threshold (:257-283).  Implemented with a
//      per-track "best unmatched detection" cache: block arg-max over the cache, then only the rows
//      whose cached detection was just taken are rescanned -- O(T*D + matches*T) instead of sorting T*D
//      entries.  Ties: smallest flat index i*D+j (what a stable descending sort yields; numpy's
//      introsort leaves tie order unspecified).
//   3. Kalman update of matched tracks (:127-137) -- 4x4 inverse by Gauss-Jordan with partial pivoting;
//      fp32 like the reference, but BLAS/LAPACK summation order is not reproduced: tolerance, not bits.
//   4. unmatched detections start tracks (ids in detection order, :291-294), tracks unseen for more than
//      max_age frames are dropped (:297), survivors are compacted in list order.
//   5. output = tracks updated this frame with hits >= min_hits, in list order (:300-313).
// State moves main -> scratch (predict/update in place) -> main (compaction), so no in-place hazards.
#include <hip/hip_runtime.h>
#include <stdint.h>

#include <new>

#include "../../include/yololite_hip.h"

struct yl_tracker {
  int device, S, T;             // streams, capacity per stream
  float iou_thr;
  int max_age, min_hits, by_class;
  // main and scratch state: x [S][T][7], P [S][T][49], score [S][T], ints [S][T][5] = id, cls, hits, age, tsu
  float *x[2], *P[2], *score[2];
  int* meta[2];
  int *ntracks, *next_id;       // [S]
  int* overflow;                // [S] tracks that could not be created (capacity)
};

namespace {

constexpr int NT = 256;

struct TrackP {
  float *x0, *P0, *sc0; int* m0;      // main
  float *x1, *P1, *sc1; int* m1;      // scratch
  int *ntracks, *next_id, *overflow;
  int T;
  float thr; int max_age, min_hits, by_class;
  const float* dets; const int* counts; int max_out;
  int* out_id; float* out_box; int* out_cls; float* out_score; int* out_count;
};

// tools/tracker.py:9-24 (float32 scalar arithmetic)
__device__ __forceinline__ void yl_xyxy_to_z(const float* b, float* z) {
  const float w = b[2] - b[0], h = b[3] - b[1];
  z[0] = b[0] + w / 2.0f;
  z[1] = b[1] + h / 2.0f;
  z[2] = w * h;
  z[3] = w / (h + 1e-6f);
}
// tools/tracker.py:27-39
__device__ __forceinline__ void yl_z_to_xyxy(const float* x, float* b) {
  const float w = sqrtf(x[2] * x[3]);
  const float h = x[2] / (w + 1e-6f);
  b[0] = x[0] - w / 2.0f; b[1] = x[1] - h / 2.0f;
  b[2] = x[0] + w / 2.0f; b[3] = x[1] + h / 2.0f;
}
// tools/tracker.py:42-71
__device__ __forceinline__ float yl_iou(const float* a, const float* b) {
  const float ix1 = fmaxf(a[0], b[0]), iy1 = fmaxf(a[1], b[1]);
  const float ix2 = fminf(a[2], b[2]), iy2 = fminf(a[3], b[3]);
  const float iw = fmaxf(0.0f, ix2 - ix1), ih = fmaxf(0.0f, iy2 - iy1);
  const float inter = iw * ih;
  const float uni = (a[2] - a[0]) * (a[3] - a[1]) + (b[2] - b[0]) * (b[3] - b[1]) - inter;
  return uni > 0.0f ? inter / uni : 0.0f;
}

// block-wide exclusive scan of 0/1 flags over n items (n arbitrary), result through `pos`, total returned
__device__ int yl_block_scan(const uint8_t* flag, int n, int* pos, int* sh) {
  const int tid = threadIdx.x;
  const int per = (n + NT - 1) / NT;
  const int b = tid * per, e = min(n, b + per);
  int c = 0;
  for (int i = b; i < e; ++i) c += flag[i];
  sh[tid] = c;
  __syncthreads();
  if (tid == 0) {
    int a = 0;
    for (int i = 0; i < NT; ++i) { const int v = sh[i]; sh[i] = a; a += v; }
    sh[NT] = a;
  }
  __syncthreads();
  int a = sh[tid];
  for (int i = b; i < e; ++i) { pos[i] = a; a += flag[i]; }
  const int total = sh[NT];
  __syncthreads();
  return total;
}

// Kalman measurement update (tools/tracker.py:127-137) on one track's state, H = [I4 0]
__device__ void yl_kf_update(float* x, float* P, const float* z) {
  float y[4], Sm[4][4], inv[4][4];
  for (int i = 0; i < 4; ++i) y[i] = z[i] - x[i];
  for (int i = 0; i < 4; ++i)
    for (int j = 0; j < 4; ++j) {
      Sm[i][j] = P[i * 7 + j] + (i == j ? 1.0f : 0.0f);       // H P H^T + R
      inv[i][j] = i == j ? 1.0f : 0.0f;
    }
  for (int c = 0; c < 4; ++c) {                                // Gauss-Jordan, partial pivoting
    int p = c;
    float best = fabsf(Sm[c][c]);
    for (int r = c + 1; r < 4; ++r)
      if (fabsf(Sm[r][c]) > best) { best = fabsf(Sm[r][c]); p = r; }
    if (p != c)
      for (int k = 0; k < 4; ++k) {
        float t = Sm[c][k]; Sm[c][k] = Sm[p][k]; Sm[p][k] = t;
        t = inv[c][k]; inv[c][k] = inv[p][k]; inv[p][k] = t;
      }
    const float d = 1.0f / Sm[c][c];
    for (int k = 0; k < 4; ++k) { Sm[c][k] *= d; inv[c][k] *= d; }
    for (int r = 0; r < 4; ++r) {
      if (r == c) continue;
      const float f = Sm[r][c];
      for (int k = 0; k < 4; ++k) { Sm[r][k] -= f * Sm[c][k]; inv[r][k] -= f * inv[c][k]; }
    }
  }
  float K[7][4];                                               // P H^T S^-1
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 4; ++j) {
      float a = 0.0f;
      for (int k = 0; k < 4; ++k) a += P[i * 7 + k] * inv[k][j];
      K[i][j] = a;
    }
  for (int i = 0; i < 7; ++i) {
    float a = 0.0f;
    for (int k = 0; k < 4; ++k) a += K[i][k] * y[k];
    x[i] += a;
  }
  float Pn[49];                                                // (I - K H) P
  for (int i = 0; i < 7; ++i)
    for (int j = 0; j < 7; ++j) {
      float a = 0.0f;
      for (int k = 0; k < 7; ++k) {
        const float ikh = (i == k ? 1.0f : 0.0f) - (k < 4 ? K[i][k] : 0.0f);
        a += ikh * P[k * 7 + j];
      }
      Pn[i * 7 + j] = a;
    }
  for (int i = 0; i < 49; ++i) P[i] = Pn[i];
}

__global__ __launch_bounds__(NT) void yl_track_update_kernel(TrackP p) {
  extern __shared__ __attribute__((aligned(16))) unsigned char smem[];
  const int s = blockIdx.x, tid = threadIdx.x;
  const int T = p.T;
  const int nt = p.ntracks[s];
  int D = p.counts[s];
  D = D < 0 ? 0 : (D > p.max_out ? p.max_out : D);
  const float* det = p.dets + (size_t)s * p.max_out * 6;
  float* x0 = p.x0 + (size_t)s * T * 7;   float* x1 = p.x1 + (size_t)s * T * 7;
  float* P0 = p.P0 + (size_t)s * T * 49;  float* P1 = p.P1 + (size_t)s * T * 49;
  float* sc0 = p.sc0 + (size_t)s * T;     float* sc1 = p.sc1 + (size_t)s * T;
  int* m0 = p.m0 + (size_t)s * T * 5;     int* m1 = p.m1 + (size_t)s * T * 5;

  // LDS carve-up
  float* tbox = reinterpret_cast<float*>(smem);               // [T][4] predicted boxes
  float* rbest = tbox + 4 * T;                                // [T] cached best IoU of the row
  int* rarg = reinterpret_cast<int*>(rbest + T);              // [T] its detection
  int* tmatch = rarg + T;                                     // [T] matched detection or -1
  int* pos = tmatch + T;                                      // [max(T, max_out)] scan positions
  const int NP = T > p.max_out ? T : p.max_out;
  int* sh = pos + NP;                                         // [NT + 1] + reduction scratch [2*NT]
  float* redv = reinterpret_cast<float*>(sh + NT + 1);
  int* redi = reinterpret_cast<int*>(redv + NT);
  uint8_t* dmatched = reinterpret_cast<uint8_t*>(redi + NT);  // [max_out]
  uint8_t* flag = dmatched + p.max_out;                       // [max(T, max_out)]

  // 1. predict -> scratch
  for (int t = tid; t < nt; t += NT) {
    float x[7], P[49];
    for (int i = 0; i < 7; ++i) x[i] = x0[t * 7 + i];
    for (int i = 0; i < 49; ++i) P[i] = P0[t * 49 + i];
    x[0] += x[4]; x[1] += x[5]; x[2] += x[6];                  // F x
    for (int j = 0; j < 7; ++j) {                              // F P
      P[0 * 7 + j] += P[4 * 7 + j]; P[1 * 7 + j] += P[5 * 7 + j]; P[2 * 7 + j] += P[6 * 7 + j];
    }
    for (int i = 0; i < 7; ++i) {                              // (F P) F^T
      P[i * 7 + 0] += P[i * 7 + 4]; P[i * 7 + 1] += P[i * 7 + 5]; P[i * 7 + 2] += P[i * 7 + 6];
    }
    for (int i = 0; i < 7; ++i) P[i * 7 + i] += 0.01f;         // + Q
    for (int i = 0; i < 7; ++i) x1[t * 7 + i] = x[i];
    for (int i = 0; i < 49; ++i) P1[t * 49 + i] = P[i];
    sc1[t] = sc0[t];
    m1[t * 5 + 0] = m0[t * 5 + 0]; m1[t * 5 + 1] = m0[t * 5 + 1]; m1[t * 5 + 2] = m0[t * 5 + 2];
    m1[t * 5 + 3] = m0[t * 5 + 3] + 1;                         // age
    m1[t * 5 + 4] = m0[t * 5 + 4] + 1;                         // time_since_update
    yl_z_to_xyxy(x, tbox + 4 * t);
    tmatch[t] = -1;
  }
  for (int j = tid; j < D; j += NT) dmatched[j] = 0;
  __syncthreads();

  // 2. greedy assignment
  auto scan_row = [&](int t) {                                 // best unmatched detection of track t
    float best = -1.0f; int arg = -1;
    const int tc = m1[t * 5 + 1];
    for (int j = 0; j < D; ++j) {
      if (dmatched[j]) continue;
      float v = yl_iou(tbox + 4 * t, det + 6 * j);
      if (p.by_class && (int)det[6 * j + 5] != tc) v = v * 0.0f;
      if (v > best) { best = v; arg = j; }
    }
    rbest[t] = best; rarg[t] = arg;
  };
  if (nt > 0 && D > 0) {
    for (int t = tid; t < nt; t += NT) scan_row(t);
    __syncthreads();
    const int rounds = nt < D ? nt : D;
    for (int it = 0; it < rounds; ++it) {
      float bv = -2.0f; int bi = 0x7fffffff;
      for (int t = tid; t < nt; t += NT)
        if (tmatch[t] < 0 && rarg[t] >= 0 && (rbest[t] > bv)) { bv = rbest[t]; bi = t; }   // ascending t: first max
      redv[tid] = bv; redi[tid] = bi;
      __syncthreads();
      for (int m = NT / 2; m >= 1; m >>= 1) {
        if (tid < m) {
          const float ov = redv[tid + m]; const int oi = redi[tid + m];
          if (ov > redv[tid] || (ov == redv[tid] && oi < redi[tid])) { redv[tid] = ov; redi[tid] = oi; }
        }
        __syncthreads();
      }
      const float gv = redv[0]; const int gi = redi[0];
      __syncthreads();
      if (gi == 0x7fffffff || gv < p.thr) break;               // :276-277
      const int gj = rarg[gi];
      if (tid == 0) { tmatch[gi] = gj; dmatched[gj] = 1; }
      __syncthreads();
      for (int t = tid; t < nt; t += NT)
        if (tmatch[t] < 0 && rarg[t] == gj) scan_row(t);
      __syncthreads();
    }
  }
  __syncthreads();

  // 3. Kalman update of the matched tracks (scratch, in place)
  for (int t = tid; t < nt; t += NT) {
    const int j = tmatch[t];
    if (j < 0) continue;
    float x[7], P[49], z[4];
    for (int i = 0; i < 7; ++i) x[i] = x1[t * 7 + i];
    for (int i = 0; i < 49; ++i) P[i] = P1[t * 49 + i];
    yl_xyxy_to_z(det + 6 * j, z);
    yl_kf_update(x, P, z);
    for (int i = 0; i < 7; ++i) x1[t * 7 + i] = x[i];
    for (int i = 0; i < 49; ++i) P1[t * 49 + i] = P[i];
    sc1[t] = fmaxf(sc1[t], det[6 * j + 4]);
    if (!p.by_class) m1[t * 5 + 1] = (int)det[6 * j + 5];
    m1[t * 5 + 2] += 1;                                        // hits
    m1[t * 5 + 4] = 0;                                         // time_since_update
  }
  __syncthreads();

  // 4a. survivors: scratch -> main, compacted in list order
  for (int t = tid; t < nt; t += NT) flag[t] = m1[t * 5 + 4] <= p.max_age ? 1 : 0;
  __syncthreads();
  const int nkeep = yl_block_scan(flag, nt, pos, sh);
  for (int t = tid; t < nt; t += NT) {
    if (!flag[t]) continue;
    const int d = pos[t];
    for (int i = 0; i < 7; ++i) x0[d * 7 + i] = x1[t * 7 + i];
    for (int i = 0; i < 49; ++i) P0[d * 49 + i] = P1[t * 49 + i];
    sc0[d] = sc1[t];
    for (int i = 0; i < 5; ++i) m0[d * 5 + i] = m1[t * 5 + i];
  }
  __syncthreads();
  // 4b. new tracks from the unmatched detections, ids in detection order.  The reference appends them
  // BEFORE pruning; a fresh track (time_since_update 0) always survives, so the order is the same.
  for (int j = tid; j < D; j += NT) flag[j] = dmatched[j] ? 0 : 1;
  __syncthreads();
  const int nnew = yl_block_scan(flag, D, pos, sh);
  const int nid = p.next_id[s];
  for (int j = tid; j < D; j += NT) {
    if (!flag[j]) continue;
    const int d = nkeep + pos[j];
    if (d >= T) continue;                                      // capacity: counted in `overflow`
    float z[4];
    yl_xyxy_to_z(det + 6 * j, z);
    for (int i = 0; i < 7; ++i) x0[d * 7 + i] = i < 4 ? z[i] : 0.0f;
    for (int i = 0; i < 49; ++i) P0[d * 49 + i] = (i % 8 == 0) ? 10.0f : 0.0f;
    sc0[d] = det[6 * j + 4];
    m0[d * 5 + 0] = nid + pos[j];
    m0[d * 5 + 1] = (int)det[6 * j + 5];
    m0[d * 5 + 2] = 1; m0[d * 5 + 3] = 1; m0[d * 5 + 4] = 0;
  }
  __syncthreads();
  int ntot = nkeep + nnew;
  if (tid == 0) {
    if (ntot > T) { p.overflow[s] += ntot - T; }
    p.ntracks[s] = ntot > T ? T : ntot;
    p.next_id[s] = nid + nnew;
  }
  if (ntot > T) ntot = T;
  __syncthreads();

  // 5. outputs: updated this frame and stable, in list order
  for (int t = tid; t < ntot; t += NT) flag[t] = (m0[t * 5 + 4] == 0 && m0[t * 5 + 2] >= p.min_hits) ? 1 : 0;
  __syncthreads();
  const int nout = yl_block_scan(flag, ntot, pos, sh);
  for (int t = tid; t < ntot; t += NT) {
    if (!flag[t]) continue;
    const size_t o = (size_t)s * T + pos[t];
    float b[4];
    yl_z_to_xyxy(x0 + t * 7, b);
    p.out_id[o] = m0[t * 5 + 0];
    p.out_cls[o] = m0[t * 5 + 1];
    p.out_score[o] = sc0[t];
    for (int i = 0; i < 4; ++i) p.out_box[4 * o + i] = b[i];
  }
  if (tid == 0) p.out_count[s] = nout;
}

__global__ void yl_track_reset_kernel(int* ntracks, int* next_id, int* overflow, int S, int only) {
  const int s = blockIdx.x * blockDim.x + threadIdx.x;
  if (s >= S || (only >= 0 && s != only)) return;
  ntracks[s] = 0; next_id[s] = 1; overflow[s] = 0;
}

size_t yl_track_lds(int T, int max_out) {
  const size_t NP = (size_t)(T > max_out ? T : max_out);
  return sizeof(float) * 4 * T + sizeof(float) * T + sizeof(int) * 2 * T + sizeof(int) * NP + sizeof(int) * (NT + 1) +
         sizeof(float) * NT + sizeof(int) * NT + (size_t)max_out + NP + 64;
}

}  // namespace

extern "C" {

yl_status yl_track_create(int32_t device, int32_t num_streams, int32_t max_tracks, float iou_threshold,
                          int32_t max_age, int32_t min_hits, int32_t match_by_class, yl_tracker** out) {
  if (!out || num_streams <= 0 || max_tracks <= 0 || max_tracks > 4096) return YL_ERR_INVALID;
  if (hipSetDevice(device) != hipSuccess) return YL_ERR_HIP;
  yl_tracker* t = new (std::nothrow) yl_tracker();
  if (!t) return YL_ERR_NOMEM;
  t->device = device; t->S = num_streams; t->T = max_tracks;
  t->iou_thr = iou_threshold; t->max_age = max_age; t->min_hits = min_hits; t->by_class = match_by_class ? 1 : 0;
  const size_t n = (size_t)num_streams * max_tracks;
  bool ok = true;
  for (int k = 0; k < 2; ++k) {
    ok = ok && hipMalloc(&t->x[k], n * 7 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&t->P[k], n * 49 * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&t->score[k], n * sizeof(float)) == hipSuccess;
    ok = ok && hipMalloc(&t->meta[k], n * 5 * sizeof(int)) == hipSuccess;
  }
  ok = ok && hipMalloc(&t->ntracks, num_streams * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc(&t->next_id, num_streams * sizeof(int)) == hipSuccess;
  ok = ok && hipMalloc(&t->overflow, num_streams * sizeof(int)) == hipSuccess;
  if (!ok) { yl_track_destroy(t); return YL_ERR_NOMEM; }
  hipLaunchKernelGGL(yl_track_reset_kernel, dim3((num_streams + 255) / 256), dim3(256), 0, 0, t->ntracks, t->next_id,
                     t->overflow, num_streams, -1);
  if (hipDeviceSynchronize() != hipSuccess) { yl_track_destroy(t); return YL_ERR_HIP; }
  *out = t;
  return YL_OK;
}

void yl_track_destroy(yl_tracker* t) {
  if (!t) return;
  hipSetDevice(t->device);
  for (int k = 0; k < 2; ++k) { hipFree(t->x[k]); hipFree(t->P[k]); hipFree(t->score[k]); hipFree(t->meta[k]); }
  hipFree(t->ntracks); hipFree(t->next_id); hipFree(t->overflow);
  delete t;
}

yl_status yl_track_reset(yl_tracker* t, int32_t stream_index, void* stream) {
  if (!t || stream_index >= t->S) return YL_ERR_INVALID;
  if (hipSetDevice(t->device) != hipSuccess) return YL_ERR_HIP;
  hipLaunchKernelGGL(yl_track_reset_kernel, dim3((t->S + 255) / 256), dim3(256), 0, (hipStream_t)stream, t->ntracks,
                     t->next_id, t->overflow, t->S, stream_index);
  return hipGetLastError() == hipSuccess ? YL_OK : YL_ERR_HIP;
}

yl_status yl_track_update(yl_tracker* t, const float* dets_dev, const int32_t* counts_dev, int32_t max_out,
                          int32_t* out_id_dev, float* out_box_dev, int32_t* out_cls_dev, float* out_score_dev,
                          int32_t* out_count_dev, void* stream) {
  if (!t || !counts_dev || max_out < 0 || (max_out > 0 && !dets_dev) || !out_id_dev || !out_box_dev || !out_cls_dev ||
      !out_score_dev || !out_count_dev)
    return YL_ERR_INVALID;
  const size_t lds = yl_track_lds(t->T, max_out);
  if (lds > 150 * 1024) return YL_ERR_CAPACITY;
  if (hipSetDevice(t->device) != hipSuccess) return YL_ERR_HIP;
  static bool attr_done[64] = {false};                       // the dynamic-LDS opt-in is a per-device attribute
  if (t->device < 0 || t->device >= 64) return YL_ERR_UNSUPPORTED;
  if (!attr_done[t->device]) {
    if (hipFuncSetAttribute((const void*)yl_track_update_kernel, hipFuncAttributeMaxDynamicSharedMemorySize,
                            150 * 1024) != hipSuccess)
      return YL_ERR_HIP;
    attr_done[t->device] = true;
  }
  TrackP p;
  p.x0 = t->x[0]; p.P0 = t->P[0]; p.sc0 = t->score[0]; p.m0 = t->meta[0];
  p.x1 = t->x[1]; p.P1 = t->P[1]; p.sc1 = t->score[1]; p.m1 = t->meta[1];
  p.ntracks = t->ntracks; p.next_id = t->next_id; p.overflow = t->overflow;
  p.T = t->T; p.thr = t->iou_thr; p.max_age = t->max_age; p.min_hits = t->min_hits; p.by_class = t->by_class;
  p.dets = dets_dev; p.counts = counts_dev; p.max_out = max_out;
  p.out_id = out_id_dev; p.out_box = out_box_dev; p.out_cls = out_cls_dev; p.out_score = out_score_dev;
  p.out_count = out_count_dev;
  hipLaunchKernelGGL(yl_track_update_kernel, dim3(t->S), dim3(NT), lds, (hipStream_t)stream, p);
  return hipGetLastError() == hipSuccess ? YL_OK : YL_ERR_HIP;
}

// Re-allocates the bank with a larger per-stream capacity and copies every stream's state (synchronises the device).
// The reference keeps an unbounded python list (tools/tracker.py:166,299-305); the single-stream mirror
// (tracker.KalmanSortTracker) grows its bank through this call BEFORE an update could overflow.
yl_status yl_track_grow(yl_tracker* t, int32_t new_max_tracks) {
  if (!t || new_max_tracks > 4096) return YL_ERR_INVALID;
  if (new_max_tracks <= t->T) return YL_OK;
  if (hipSetDevice(t->device) != hipSuccess) return YL_ERR_HIP;
  if (hipDeviceSynchronize() != hipSuccess) return YL_ERR_HIP;
  const size_t S = (size_t)t->S, To = (size_t)t->T, Tn = (size_t)new_max_tracks;
  // all eight new arrays first, the swap only when every allocation and copy has succeeded: on failure the bank is
  // untouched (arrays grown one by one would leave row pitches Tn and To mixed under an unchanged T -- wrong rows for
  // every stream >= 1)
  void** ptrs[8] = {(void**)&t->x[0], (void**)&t->P[0], (void**)&t->score[0], (void**)&t->meta[0],
                    (void**)&t->x[1], (void**)&t->P[1], (void**)&t->score[1], (void**)&t->meta[1]};
  const size_t eb[4] = {7 * sizeof(float), 49 * sizeof(float), sizeof(float), 5 * sizeof(int)};
  void* nw[8] = {nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr};
  bool ok = true;
  for (int k = 0; k < 8 && ok; ++k) {
    const size_t e = eb[k & 3];
    ok = hipMalloc(&nw[k], S * Tn * e) == hipSuccess && hipMemset(nw[k], 0, S * Tn * e) == hipSuccess &&
         hipMemcpy2D(nw[k], Tn * e, *ptrs[k], To * e, To * e, S, hipMemcpyDeviceToDevice) == hipSuccess;
  }
  if (!ok) {
    for (int k = 0; k < 8; ++k) hipFree(nw[k]);
    (void)hipGetLastError();
    return YL_ERR_NOMEM;       // nothing was swapped: the bank is exactly as before
  }
  for (int k = 0; k < 8; ++k) { hipFree(*ptrs[k]); *ptrs[k] = nw[k]; }
  t->T = new_max_tracks;
  return YL_OK;
}

yl_status yl_track_stats(yl_tracker* t, int32_t* ntracks_host, int32_t* overflow_host) {
  if (!t) return YL_ERR_INVALID;
  if (hipSetDevice(t->device) != hipSuccess) return YL_ERR_HIP;
  if (hipDeviceSynchronize() != hipSuccess) return YL_ERR_HIP;
  if (ntracks_host && hipMemcpy(ntracks_host, t->ntracks, sizeof(int) * t->S, hipMemcpyDeviceToHost) != hipSuccess)
    return YL_ERR_HIP;
  if (overflow_host && hipMemcpy(overflow_host, t->overflow, sizeof(int) * t->S, hipMemcpyDeviceToHost) != hipSuccess)
    return YL_ERR_HIP;
  return YL_OK;
}

}  // extern "C"
