// pipeline.hip — the fused per-image path: Tester_FRCNN:testOne -> ImageDetect:detect ->
// model:forward (models/vgg.lua:23-31) -> per-class NMS -> keep_top_k, as ONE stream of kernels with
// no host synchronisation between the image upload and the final detections.
//
// Memory plan (sized for 288 GB HBM, nothing is re-used to save bytes): every layer output has its
// own C8P buffer whose halo is zeroed once; packed weights (fc6 alone 411 MB) stay resident; the
// ROI-pooled matrix, fc activations, scored boxes and NMS outputs are fixed-capacity buffers
// allocated at create time.  The reference's 500-ROI chunking (ImageDetect.lua:116-124) is not
// needed: all ROIs go through each GEMM at once (rows are independent, results identical), so the
// fc weights stream from HBM once per image instead of twice.
#include <cstdlib>
#include <iterator>
#include <algorithm>
#include <map>
#include <string>
#include <tuple>
#include <vector>

#include "dense.h"
#include "resnet.h"

namespace mpn {
int launch_bbox_decode(const float *d_boxes, const float *d_deltas, int N, int C, float *d_out, int clamp, float im_w,
                       float im_h, hipStream_t s);

// softmax over head[:, 0:C] (row stride ld) -> scores [M,C]; one wave per row
__global__ __launch_bounds__(256) void head_softmax_kernel(const float *__restrict__ head, int ld, int M, int C,
                                                           float *__restrict__ scores) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float *r = head + (size_t)row * ld;
  float mx = -INFINITY;
  for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
  float sum = 0.f;
  for (int c = lane; c < C; c += 64) sum += expf(r[c] - mx);
#pragma unroll
  for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
  for (int c = lane; c < C; c += 64) scores[(size_t)row * C + c] = expf(r[c] - mx) / sum;
}

// BBoxNorm (modules/BBoxNorm.lua:28-29) + convertFrom (utils.lua:229-247) + optional clamp
// (Tester_FRCNN.lua:75-78) on head[:, C:5C]; thread per (roi, class)
__global__ void head_decode_kernel(const float *__restrict__ head, int ld, int col0, int M, int C, const float *__restrict__ boxes,
                                   int has_norm, float m0, float m1, float m2, float m3, float s0, float s1, float s2,
                                   float s3, int clamp, float im_w, float im_h, float *__restrict__ raw, float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)M * C) return;
  int i = (int)(t / C), c = (int)(t - (size_t)i * C);
  const float *d = head + (size_t)i * ld + col0 + 4 * c;
  float d0 = d[0], d1 = d[1], d2 = d[2], d3 = d[3];
  if (has_norm) {
    d0 = d0 * s0; d0 = d0 + m0;
    d1 = d1 * s1; d1 = d1 + m1;
    d2 = d2 * s2; d2 = d2 + m2;
    d3 = d3 * s3; d3 = d3 + m3;
  }
  if (raw) { float *r = raw + 4 * t; r[0] = d0; r[1] = d1; r[2] = d2; r[3] = d3; }
  const float *bx = boxes + 4 * (size_t)i;
  float x1 = bx[0], y1 = bx[1], x2 = bx[2], y2 = bx[3];
  float xc = (x1 + x2) * 0.5f, yc = (y1 + y2) * 0.5f;
  float w = x2 - x1, h = y2 - y1;
  float p0 = d0 * w, p1 = d1 * h;
  float xtc = xc + p0, ytc = yc + p1;
  float wt = expf(d2) * w, ht = expf(d3) * h;
  float hw = wt * 0.5f, hh = ht * 0.5f;
  float o0 = xtc - hw, o1 = ytc - hh, o2 = xtc + hw, o3 = ytc + hh;
  if (clamp) {  // Tester_FRCNN.lua:75-78 — the first detect() of testOne only
    o0 = o0 < 1.0f ? 1.0f : (o0 > im_w ? im_w : o0);
    o2 = o2 < 1.0f ? 1.0f : (o2 > im_w ? im_w : o2);
    o1 = o1 < 1.0f ? 1.0f : (o1 > im_h ? im_h : o1);
    o3 = o3 < 1.0f ? 1.0f : (o3 > im_h ? im_h : o3);
  }
  float *o = out + 4 * t;
  o[0] = o0; o[1] = o1; o[2] = o2; o[3] = o3;
}

// integral eval head (model_utils.lua:296-313): K classifiers -> softmax each -> mean over K.
// logits [M, K*C] row-major; one wave per row; sum over k in k order, then * (1/K) like nn.Mean.
__global__ __launch_bounds__(256) void integral_softmax_mean_kernel(const float *__restrict__ logits, int M, int K, int C,
                                                                    float *__restrict__ scores) {
  int row = blockIdx.x * 4 + (threadIdx.x >> 6), lane = threadIdx.x & 63;
  if (row >= M) return;
  const float inv = 1.0f / (float)K;
  // per k: max and sum over all C columns (wave reductions), then accumulate this lane's columns
  float accv[4] = {0.f, 0.f, 0.f, 0.f};  // supports C <= 256
  for (int k = 0; k < K; ++k) {
    const float *r = logits + ((size_t)row * K + k) * C;
    float mx = -INFINITY;
    for (int c = lane; c < C; c += 64) mx = fmaxf(mx, r[c]);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) mx = fmaxf(mx, __shfl_xor(mx, off));
    float sum = 0.f;
    for (int c = lane; c < C; c += 64) sum += expf(r[c] - mx);
#pragma unroll
    for (int off = 32; off >= 1; off >>= 1) sum += __shfl_xor(sum, off);
    int q = 0;
    for (int c = lane; c < C; c += 64, ++q) accv[q] += expf(r[c] - mx) / sum;
  }
  int q = 0;
  for (int c = lane; c < C; c += 64, ++q) scores[(size_t)row * C + c] = accv[q] * inv;
}

// pooled C8 matrix [cb*PP+bin][Mp][8] -> reference order [N, C*PP] (debug / parity only)
__global__ void unpack_pooled_kernel(const float *__restrict__ xc8, int N, int C, int PP, int Mp, float *__restrict__ out) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  size_t total = (size_t)N * C * PP;
  if (t >= total) return;
  int bin = (int)(t % PP); size_t r = t / PP;
  int c = (int)(r % C); int n = (int)(r / C);
  out[t] = xc8[(((size_t)(c >> 3) * PP + bin) * Mp + n) * 8 + (c & 7)];
}

__global__ void copy_cols_kernel(const float *__restrict__ src, int ld, int col0, int M, int ncols, float *__restrict__ dst) {
  size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (t >= (size_t)M * ncols) return;
  int m = (int)(t / ncols), c = (int)(t - (size_t)m * ncols);
  dst[t] = src[(size_t)m * ld + col0 + c];
}
// ---- proposal (ROI) sharding of one image (mpn_frcnn_shard_*): record pack / unpack ------------------------------------------
// Balanced contiguous partition of n items over `world` ranks: the first n % world ranks own one item more.
__host__ __device__ inline void shard_bounds(int n, int world, int rank, int *lo, int *hi) {
  const int base = n / world, rem = n % world;
  *lo = rank * base + (rank < rem ? rank : rem);
  *hi = *lo + base + (rank < rem ? 1 : 0);
}
__device__ inline void shard_owner(int item, int n, int world, int *rank, int *local) {
  const int base = n / world, rem = n % world, cut = rem * (base + 1);
  if (item < cut) { *rank = item / (base + 1); *local = item - *rank * (base + 1); }
  else { const int q = (item - cut) / base; *rank = rem + q; *local = item - cut - q * base; }
}

// this rank's joined score / box tables (P passes x n_local rows, pass-major) -> its row record
// [scores: P x chunk x C][boxes: P x chunk x 4C], rows at / beyond n_local zeroed
__global__ void shard_pack_rows_kernel(const float *__restrict__ sc, const float *__restrict__ bb, int n_local, int P, int C, int chunk,
                                       float *__restrict__ rec) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ns = (size_t)P * chunk * C, nb = ns * 4;
  if (t >= ns + nb) return;
  const bool is_box = t >= ns;
  const size_t u = is_box ? t - ns : t;
  const int w = is_box ? 4 * C : C;
  const int col = (int)(u % w);
  const size_t r = u / w;
  const int i = (int)(r % chunk), k = (int)(r / chunk);
  float v = 0.0f;
  if (i < n_local) v = (is_box ? bb : sc)[((size_t)k * n_local + i) * w + col];
  rec[t] = v;
}

// all ranks' row records [world][rec_floats] -> the image's joined tables sc [P*N, C], bb [P*N, 4C] in the unsharded row order
__global__ void shard_unpack_rows_kernel(const float *__restrict__ all, int N, int world, int P, int C, int chunk, size_t rec_floats,
                                         float *__restrict__ sc, float *__restrict__ bb) {
  const size_t t = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
  const size_t ns = (size_t)P * N * C, nb = ns * 4;
  if (t >= ns + nb) return;
  const bool is_box = t >= ns;
  const size_t u = is_box ? t - ns : t;
  const int w = is_box ? 4 * C : C;
  const int col = (int)(u % w);
  const size_t r = u / w;
  const int row = (int)(r % N), k = (int)(r / N);
  int owner, i;
  shard_owner(row, N, world, &owner, &i);
  const size_t plane = is_box ? (size_t)P * chunk * C : 0;
  (is_box ? bb : sc)[u] = all[(size_t)owner * rec_floats + plane + ((size_t)k * chunk + i) * w + col];
}

// class record of one rank: [n_keep: cmax ints][keep: cmax x rows x 5][keep_idx: cmax x rows ints][voted: cmax x rows x 5 (if voting)]
__host__ __device__ inline size_t shard_class_rec_floats(int cmax, int rows, int voting) {
  return (size_t)cmax * (1 + (size_t)rows * (voting ? 11 : 6));
}
// grid (ceil(rows / 256), cmax): slot j = class c0 + j of this rank's range [c0, c1)
__global__ void shard_pack_classes_kernel(const float *__restrict__ keep, const int *__restrict__ kidx, const int *__restrict__ n_keep,
                                          const float *__restrict__ voted, int c0, int c1, int rows, int cmax, float *__restrict__ rec) {
  const int j = blockIdx.y, c = c0 + j, i = blockIdx.x * blockDim.x + threadIdx.x;
  int *rn = reinterpret_cast<int *>(rec);
  const int n = c < c1 ? min(max(n_keep[c], 0), rows) : 0;
  if (i == 0) rn[j] = n;
  if (i >= n) return;
  float *rk = rec + cmax + ((size_t)j * rows + i) * 5;
  const float *k = keep + ((size_t)c * rows + i) * 5;
  rk[0] = k[0]; rk[1] = k[1]; rk[2] = k[2]; rk[3] = k[3]; rk[4] = k[4];
  reinterpret_cast<int *>(rec + cmax + (size_t)cmax * rows * 5)[(size_t)j * rows + i] = kidx[(size_t)c * rows + i];
  if (voted) {
    float *rv = rec + cmax + (size_t)cmax * rows * 6 + ((size_t)j * rows + i) * 5;
    const float *v = voted + ((size_t)c * rows + i) * 5;
    rv[0] = v[0]; rv[1] = v[1]; rv[2] = v[2]; rv[3] = v[3]; rv[4] = v[4];
  }
}
// all ranks' class records -> the image's per-class tables; grid (ceil(rows / 256), n_cls)
__global__ void shard_unpack_classes_kernel(const float *__restrict__ all, int n_cls, int world, int rows, int cmax, size_t rec_floats,
                                            float *__restrict__ keep, int *__restrict__ kidx, int *__restrict__ n_keep,
                                            float *__restrict__ voted) {
  const int c = blockIdx.y, i = blockIdx.x * blockDim.x + threadIdx.x;
  int owner, j;
  shard_owner(c, n_cls, world, &owner, &j);
  const float *rec = all + (size_t)owner * rec_floats;
  const int n = min(max(reinterpret_cast<const int *>(rec)[j], 0), rows);
  if (i == 0) n_keep[c] = n;
  if (i >= n) return;
  const float *rk = rec + cmax + ((size_t)j * rows + i) * 5;
  float *k = keep + ((size_t)c * rows + i) * 5;
  k[0] = rk[0]; k[1] = rk[1]; k[2] = rk[2]; k[3] = rk[3]; k[4] = rk[4];
  kidx[(size_t)c * rows + i] = reinterpret_cast<const int *>(rec + cmax + (size_t)cmax * rows * 5)[(size_t)j * rows + i];
  if (voted) {
    const float *rv = rec + cmax + (size_t)cmax * rows * 6 + ((size_t)j * rows + i) * 5;
    float *v = voted + ((size_t)c * rows + i) * 5;
    v[0] = rv[0]; v[1] = rv[1]; v[2] = rv[2]; v[3] = rv[3]; v[4] = rv[4];
  }
}
}  // namespace mpn

using namespace mpn;

struct ConvLayer {
  int Cin, Cout, pool;
  float *wpk = nullptr, *bpk = nullptr, *wino = nullptr;  // direct-conv and Winograd-transformed weights
  float *w36 = nullptr;     // first layer (<= 4 input channels, no pool): the K = 36 formulation's weights
  float *out = nullptr;     // C8P buffer for the conv output (max image size)
  float *pooled = nullptr;  // C8P buffer for the pooled output (when pool)
};

struct mpn_frcnn {
  mpn_frcnn_config cfg;
  std::vector<int> cout, pool_after;
  std::vector<ConvLayer> conv;
  float *img_c8p = nullptr;
  size_t act_total_bytes = 0;
  std::vector<std::pair<float *, size_t>> act_bufs;  // for re-zeroing when the image size changes
  int last_h = -1, last_w = -1;
  int feat_c = 0;
  // head
  int K6 = 0, Mp = 0, n_head = 0;
  float *w6 = nullptr, *b6 = nullptr, *w7 = nullptr, *b7 = nullptr, *wh = nullptr, *bh = nullptr;
  float *rois = nullptr, *x6 = nullptr, *y6 = nullptr, *y7 = nullptr, *head = nullptr;
  float *feat_pm = nullptr;   // pixel-major copy of the last trunk map (plain Fast R-CNN head: roi_pool_pm)
  bool feat_pm_valid = false;
  float *scores = nullptr, *bbox = nullptr, *bbox_raw = nullptr;
  // NMS-stage buffers: two sets so that image i's NMS (side stream) overlaps image i+1's trunk
  float *scored_b[2] = {nullptr, nullptr}, *keep_b[2] = {nullptr, nullptr}, *thresh_b[2] = {nullptr, nullptr};
  float *voted_b[2] = {nullptr, nullptr}, *voted = nullptr;      // bbox-voted tables (opt.test_bbox_voting)
  float *it_scores = nullptr, *it_bbox = nullptr, *it_boxes = nullptr;  // iterative localisation: rows of both passes
  float *scaled = nullptr, *scale_tmp = nullptr;  // getImages' rescaled image (ImageDetect.lua:34-43), grown on demand
  size_t scaled_bytes = 0, scale_tmp_bytes = 0;
  int net_h = 0, net_w = 0;                       // size of the image the trunk last saw
  int *counts_b[2] = {nullptr, nullptr}, *keep_idx_b[2] = {nullptr, nullptr}, *n_keep_b[2] = {nullptr, nullptr};
  float *scored = nullptr, *keep = nullptr, *thresh = nullptr;   // set of the most recent call
  int *counts = nullptr, *keep_idx = nullptr, *n_keep = nullptr;
  hipStream_t side = nullptr;           // side stream (default priority: see create) for the heads and the NMS / top-k tail of the pipelined forms
  // Deferred heads (pipelined forms of the plain Fast R-CNN head): cls / bbox GEMM + softmax + decode + select of image i run on `side`
  // too, under image i + 1's first trunk layers — they are 51 us of kernels that leave most of the GPU idle.  What they read is held per
  // buffer set: fc7's output (y7_b) and a copy of the caller's boxes (boxes_b); join_tail(b) orders their reuse two calls later.
  hipStream_t defer_stream = nullptr;   // non-null while run_detect is to hand the heads over to it
  bool was_deferred = false;            // the previous pipelined call handed its heads over
  int defer_set = 0;
  float *y7_b[2] = {nullptr, nullptr}, *boxes_b[2] = {nullptr, nullptr}, *y7_last = nullptr;  // y7_last: where the last head left fc7's output
  hipEvent_t ev_fc7 = nullptr;
  hipEvent_t ev_head[2] = {nullptr, nullptr}, ev_tail[2] = {nullptr, nullptr};
  bool tail_pending[2] = {false, false};
  unsigned long long seq = 0;
  float *dbg = nullptr;
  size_t dbg_bytes = 0;
  int last_n = 0, last_rows = 0;
  int fuse_pool = 1;
  // ---- MultiPathNet head (models/multipathnet.lua:64-120); empty for plain Fast R-CNN
  struct Tower { int region, use4, use3, total_feat; float *mix_w, *mix_b, *w6, *b6, *w7, *b7; unsigned short *w6_s3 = nullptr, *w7_s3 = nullptr; };
  bool is_mpnet = false;
  std::vector<int> rn_region;   // ResNet towers: Foveal region per tower (empty = plain resnet.lua)
  ResNetGraph *rn = nullptr;  // ResNet Fast R-CNN (mpn_resnet_create): trunk + per-ROI layer4 replace the VGG convs / fc6 / fc7
  int tap3 = -1, tap4 = -1, n_integral = 1;
  bool conv345_norm = true;  // model_conv345_norm (model_utils.lua:209): false = the MulConstant(1, 1/30, 1/200) branch
  std::vector<Tower> towers;
  float *fov = nullptr, *tx = nullptr, *ty = nullptr, *tz6 = nullptr, *cat = nullptr, *cls_rm = nullptr, *bbox_rm = nullptr;
  float *wcls = nullptr, *bcls = nullptr, *wbbox = nullptr, *bbbox = nullptr;
  Act tap_act[3];  // conv5, conv4, conv3 of the last trunk run
  float *vmax_tab[3] = {nullptr, nullptr, nullptr};  // vertical range-max tables of the three maps (MultiPathNet ROI pools)
  bool vmax_built[3] = {false, false, false};         // built for the current tap_act maps (per map: the pooling stream builds a map's tables where its first pooling is enqueued)
  bool vmax_pm = false;                               // ... in the pixel-major form
  float *mix_scale = nullptr;                         // [2 tower parities][3][Mp]: per-(map, ROI) nn.Normalize scales the mix GEMM applies
  // tower t + 1's skip pooling (L2 -> L1 bound, no matrix work) runs on its own stream under tower t's GEMMs (matrix-bound):
  float *tx2 = nullptr;                               // second pooled-operand buffer (towers alternate between tx and tx2)
  // round 6: two towers that pool the SAME Foveal region, one's maps a prefix of the other's (models/multipathnet.lua:74-113: the "het"
  // tower = region 2 with conv5 + conv4 + conv3, tower 2 = region 2 with conv5 + conv4), share ONE pooled operand: the wider one is pooled
  // once into tx3, the narrower tower's mix GEMM reads its K prefix (the per-map nn.Normalize scales are per (map, region, ROI): the same)
  float *tx3 = nullptr;
  int share_provider = -1, share_consumer = -1;       // tower indices (-1: no such pair)
  // The pooling stream IS the side stream (the NMS / top-k tail's) since the end of round 6: the tail of image i - 1 runs under image i's
  // trunk and is long over when image i's first pooling is enqueued behind it, and the handle needs one stream fewer.  With a stream of its own
  // the host-fed form drove five streams on ROCm's four hardware queues, and whichever stream shared a queue with the upload stream waited
  // behind the upload's completion marker: 0.2 ms per image (configs[2] host-fed 13.26-13.31 -> 13.07-13.13 ms, profiles/r06_hw_queues.txt).
  hipStream_t pool_stream = nullptr;   // alias of `side` (never destroyed on its own); nullptr = no overlapped pooling (plain Fast R-CNN handles)
  bool pool_on_side = false;
  hipEvent_t ev_pool_done[3] = {nullptr, nullptr, nullptr}, ev_mix_done[3] = {nullptr, nullptr, nullptr}, ev_pool_go = nullptr;
  // two tower LANES (round 6): the towers of one image are independent until the concat (ModelParallelTable.lua:195-242 ran them on
  // different GPUs), so towers 1, 3 run on the handle's second tower stream with their own mix / fc6 buffers beside towers 0, 2, 4 on the
  // caller's stream: one lane's short-K mix GEMM (6.1 block rounds on 256 CUs, 40 stages per tile) and the prologue / epilogue of every
  // launch run under the other lane's fc6 / fc7 instead of leaving the matrix pipe idle.  Pure scheduling: bit-identical results.
  unsigned short *w6_s3 = nullptr, *x6_s3 = nullptr;  // MPN_FC_SPLIT3: fc6's weights (packed once) and operand (per image) as three bf16 planes
  unsigned short *w7_s3 = nullptr, *y6_s3 = nullptr;  // ... and fc7's
  unsigned short *ty_s3[2] = {nullptr, nullptr}, *tz6_s3[2] = {nullptr, nullptr};  // MultiPathNet towers: the per-lane fc6 / fc7 operands as planes
  hipStream_t tower_stream = nullptr;
  hipEvent_t ev_lane_go = nullptr, ev_lane_done = nullptr;
  float *ty2 = nullptr, *tz6_2 = nullptr;
  std::vector<void *> allocs;
  Scratch scratch;  // split-K slabs, NMS masks, ... of THIS handle (bound to the calling thread by ScratchScope in every entry point)
  int device = 0;   // the handle lives on the device that was current at creation
  // host-fed throughput form (mpn_frcnn_test_one_pipelined_host): three staging sets filled by the copy stream
  hipStream_t copy = nullptr;
  static constexpr int kStage = 3;
  float *stage_img[kStage] = {}, *stage_boxes[kStage] = {};
  size_t stage_cap[kStage] = {};
  hipEvent_t ev_up[kStage] = {}, ev_consumed[kStage] = {};
  bool used_pending[kStage] = {};
  unsigned long long up_seq = 0;
  // proposal sharding (mpn_frcnn_test_one_sharded): this rank's row / class records and the gathered ones, grown on demand
  float *sh_buf[4] = {nullptr, nullptr, nullptr, nullptr};
  size_t sh_bytes[4] = {0, 0, 0, 0};
  // ---- captured launch graphs (round 4): the kernel chain of a SEGMENT of the per-image path — the head (transform .. decode, the
  // iterative-localisation passes) or the tail (per-class NMS, voting, top-k) — is captured once per (pointers, shape) with
  // hipStreamBeginCapture on the handle's capture stream and replayed with hipGraphLaunch on the caller's stream: one host call instead
  // of 30-60 launches.  A segment is replayed only when (a) the previous execution of that segment kind on this handle had the same
  // shape — the host-side state a real run leaves (cached-feature flags, sizes) is then exactly what it would be — (b) no library buffer
  // was replaced since the capture (alloc_generation), (c) profiling is off.  Everything between the segments (cross-stream events,
  // the select kernel, uploads) stays ordinary stream work, so the pipelined forms keep their overlap.
  struct GraphKey {
    int kind; const void *a, *b, *c, *d; int i0, i1, i2, i3;
    bool operator<(const GraphKey &o) const {
      return std::tie(kind, a, b, c, d, i0, i1, i2, i3) < std::tie(o.kind, o.a, o.b, o.c, o.d, o.i0, o.i1, o.i2, o.i3);
    }
  };
  struct GraphEntry { hipGraphExec_t exec = nullptr; unsigned long long gen = 0, last_use = 0; bool failed = false; int seen = 0; hipStream_t last_stream = nullptr; bool launched = false; };
  unsigned long long graph_clock = 0;
  std::map<GraphKey, GraphEntry> graphs;
  // the last few caller-pointer keys seen ONCE, per segment kind (a small ring: the pipelined forms alternate two output buffer sets, a host
  // may rotate a handful): such a key enters `graphs` only at its second sighting while still in the ring, so a host that hands in fresh
  // buffers every call never occupies the cache
  static constexpr int kUnseen = 8;
  GraphKey unseen[4][kUnseen] = {};
  bool unseen_valid[4][kUnseen] = {};
  int unseen_next[4] = {0, 0, 0, 0};
  int graphs_on = 0;                 // mpn_frcnn_set_graphs / MPN_GRAPHS (opt-in: see create_impl)
  hipStream_t cap_stream = nullptr;  // capture happens here (the caller's stream may be the legacy NULL stream, which cannot capture)
  int seg_shape[4][4] = {{-1, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}, {-1, -1, -1, -1}};  // shape of the last execution per segment kind
  long graph_replays = 0, graph_captures = 0;
  // optional per-kernel-group timing with HIP events recorded on the launch stream
  bool prof = false;
  std::vector<hipEvent_t> ev_pool;
  std::vector<int> ev_tag;   // one tag per (begin,end) pair
  size_t ev_used = 0;
  double prof_ms[MPN_PROF_NTAGS] = {0};
  long prof_cnt[MPN_PROF_NTAGS] = {0};
};

struct ProfScope {
  mpn_frcnn *p; hipStream_t s; bool on;
  ProfScope(mpn_frcnn *p_, int tag, hipStream_t s_) : p(p_), s(s_), on(p_->prof) {
    if (!on) return;
    if (p->ev_used + 2 > p->ev_pool.size()) {
      for (int i = 0; i < 64; ++i) { hipEvent_t e; if (hipEventCreate(&e) != hipSuccess) { on = false; return; } p->ev_pool.push_back(e); }
    }
    p->ev_tag.push_back(tag);
    (void)hipEventRecord(p->ev_pool[p->ev_used], s);
  }
  ~ProfScope() {
    if (!on) return;
    (void)hipEventRecord(p->ev_pool[p->ev_used + 1], s);
    p->ev_used += 2;
  }
};

MPN_KNOB(int, g_fuse_pool, 1);
MPN_KNOB(int, g_first_k36, 1);  // 0: the first layer on the generic direct kernel
MPN_KNOB(int, g_roi_pool_pm, 1);  // 0: ROI pooling straight from the C8P map (roi_pool_c8_kernel)
MPN_KNOB(int, g_mix_fold, 1);     // 0: MultiPathNet's nn.Normalize scales applied in place (l2norm_apply) instead of inside the mix GEMM
MPN_KNOB(int, g_tower_lanes, 1);  // 0: the towers of an image one after the other on the caller's stream (rounds 2-5) instead of two lanes (mpn_debug_set_tower_lanes)
#ifdef MPN_DEBUG_HOOKS
static int g_mpn_pool_knock = 0;
extern "C" void mpn_debug_set_mpn_pool_knock(int v) { g_mpn_pool_knock = v; }
#endif
MPN_KNOB(int, g_mix_packed, 1);   // 0: the mix GEMM's rows padded to 128 per bin as the fc operands are (rounds 2-5; mpn_debug_set_mix_packed)
MPN_KNOB(int, g_tables_lazy, 1);  // 0: all range-max tables on the launch stream in front of the head (rounds 3-5) instead of per map on the pooling stream (mpn_debug_set_tables_lazy)
MPN_KNOB(int, g_tower_order, 1);  // 0: the towers in index order instead of cheapest pooling first (mpn_debug_set_tower_order)
MPN_KNOB(int, g_tower_share, 1);  // 0: every tower pools its own operand even where two of them pool the same region's maps (mpn_debug_set_tower_share)
MPN_KNOB(int, g_pool_overlap, 1); // 0: MultiPathNet's skip pooling on the launch stream instead of its own stream under the previous tower's GEMMs
MPN_KNOB(int, g_defer_heads, 1);  // 0: the pipelined forms keep heads / softmax / decode / select on the launch stream (rounds 1-5a); 2 (test): the
                                  // side stream is held back 1 ms before the heads, so that the launch stream runs far ahead of it
#ifdef MPN_DEBUG_HOOKS
__global__ void side_stream_delay_kernel(long long ticks) {  // wall_clock64: 100 MHz
  const long long t0 = wall_clock64();
  while (wall_clock64() - t0 < ticks) __builtin_amdgcn_s_sleep(64);
}
#endif
MPN_KNOB(int, g_halo_memset, 0);  // 1: a size change clears every activation buffer whole (rounds 1-4) instead of re-laying the halos only
#ifdef MPN_DEBUG_HOOKS
extern "C" void mpn_debug_set_fuse_pool(int v) { g_fuse_pool = v; }
extern "C" void mpn_debug_set_first_k36(int v) { g_first_k36 = v; }
extern "C" void mpn_debug_set_roi_pool_pm(int v) { g_roi_pool_pm = v; }
extern "C" void mpn_debug_set_mix_fold(int v) { g_mix_fold = v; }
extern "C" void mpn_debug_set_pool_overlap(int v) { g_pool_overlap = v; }
extern "C" void mpn_debug_set_tower_lanes(int v) { g_tower_lanes = v; }
extern "C" void mpn_debug_set_tower_share(int v) { g_tower_share = v; }
extern "C" void mpn_debug_set_tower_order(int v) { g_tower_order = v; }
extern "C" void mpn_debug_set_tables_lazy(int v) { g_tables_lazy = v; }
extern "C" void mpn_debug_set_mix_packed(int v) { g_mix_packed = v; }
extern "C" void mpn_debug_set_halo_memset(int v) { g_halo_memset = v; }
extern "C" void mpn_debug_set_defer_heads(int v) { g_defer_heads = v; }
#endif

template <typename T>
static int dev_alloc(mpn_frcnn *p, T **ptr, size_t bytes, bool zero) {
  void *q = nullptr;
  hipError_t e = hipMalloc(&q, bytes ? bytes : 16);
  if (e != hipSuccess) { set_error("mpn_frcnn_create: hipMalloc(%zu) failed: %s", bytes, hipGetErrorString(e)); return MPN_ENOMEM; }
  if (zero) {
    e = hipMemset(q, 0, bytes ? bytes : 16);
    if (e != hipSuccess) { set_error("mpn_frcnn_create: hipMemset failed: %s", hipGetErrorString(e)); return MPN_EHIP; }
  }
  p->allocs.push_back(q);
  *ptr = static_cast<T *>(q);
  return MPN_OK;
}

extern "C" void mpn_frcnn_destroy(mpn_frcnn *p) {
  if (!p) return;
  (void)hipDeviceSynchronize();
  for (auto &kv : p->graphs) if (kv.second.exec) (void)hipGraphExecDestroy(kv.second.exec);
  if (p->cap_stream) (void)hipStreamDestroy(p->cap_stream);
  for (hipEvent_t e : p->ev_pool) (void)hipEventDestroy(e);
  for (int i = 0; i < 2; ++i) { if (p->ev_head[i]) (void)hipEventDestroy(p->ev_head[i]); if (p->ev_tail[i]) (void)hipEventDestroy(p->ev_tail[i]); }
  if (p->ev_fc7) (void)hipEventDestroy(p->ev_fc7);
  if (p->side) (void)hipStreamDestroy(p->side);
  for (int i = 0; i < 3; ++i) { if (p->ev_pool_done[i]) (void)hipEventDestroy(p->ev_pool_done[i]); if (p->ev_mix_done[i]) (void)hipEventDestroy(p->ev_mix_done[i]); }
  if (p->ev_pool_go) (void)hipEventDestroy(p->ev_pool_go);
  if (p->tower_stream) (void)hipStreamDestroy(p->tower_stream);
  if (p->ev_lane_go) (void)hipEventDestroy(p->ev_lane_go);
  if (p->ev_lane_done) (void)hipEventDestroy(p->ev_lane_done);
  if (p->copy) (void)hipStreamDestroy(p->copy);
  for (int i = 0; i < mpn_frcnn::kStage; ++i) { if (p->ev_up[i]) (void)hipEventDestroy(p->ev_up[i]); if (p->ev_consumed[i]) (void)hipEventDestroy(p->ev_consumed[i]); }
  p->scratch.release();
  for (int i = 0; i < mpn_frcnn::kStage; ++i) if (p->stage_img[i]) (void)hipFree(p->stage_img[i]);
  for (void *q : p->allocs) (void)hipFree(q);
  resnet_free(p->rn);
  if (p->scaled) (void)hipFree(p->scaled);
  if (p->scale_tmp) (void)hipFree(p->scale_tmp);
  if (p->dbg) (void)hipFree(p->dbg);
  for (int i = 0; i < 4; ++i) if (p->sh_buf[i]) (void)hipFree(p->sh_buf[i]);
  delete p;
}

static int create_impl(const mpn_frcnn_config *cfg, const float *const *d_conv_w, const float *const *d_conv_b,
                       const float *d_fc6_w, const float *d_fc6_b, const float *d_fc7_w, const float *d_fc7_b,
                       const float *d_cls_w, const float *d_cls_b, const float *d_bbox_w, const float *d_bbox_b,
                       const mpn_mpnet_weights *mw, mpn_frcnn **out, const mpn_resnet_weights *rw = nullptr, const mpn_graph_weights *gw = nullptr) {
  MPN_CHECK_ARG(cfg && d_cls_w && d_bbox_w && out);
  const bool graph_net = rw || gw;  // the trunk / per-ROI stage live in a ResNetGraph object
  MPN_CHECK_ARG(graph_net || (d_conv_w && d_conv_b));
  MPN_CHECK_ARG(graph_net || mw || (d_fc6_w && d_fc7_w));
  MPN_CHECK_ARG(graph_net || (cfg->n_conv > 0 && cfg->conv_cout && cfg->pool_after && cfg->fc_dim > 0));
  MPN_CHECK_ARG(cfg->pooled_h > 0 && cfg->pooled_w > 0 && cfg->n_classes > 1);
  MPN_CHECK_ARG(cfg->max_h > 0 && cfg->max_w > 0 && cfg->max_rois > 0 && cfg->max_rois <= MPN_NMS_MAX_BOXES);
  MPN_CHECK_ARG(cfg->top_k > 0);
  mpn_frcnn *p = new mpn_frcnn();
  p->cfg = *cfg;
  {  // captured launch graphs: OFF unless asked for (MPN_GRAPHS=1 in the environment, or mpn_frcnn_set_graphs per handle).  Measured on
     // MI355X (profiles/r04_launch_graphs.txt): the host's enqueue time per AlexNet image drops 484 -> 74 us, but the device-side timeline
     // does not change (the gaps between dependent kernels are the command processor's, not the host's) and the throughput lines read
     // 0.5 % LOWER with replays (c2 287.2 vs 289.1 k, c1 373.9 vs 376.4 k proposals/s): a host-CPU saving, not a speed-up, so it is opt-in
    const char *e = getenv("MPN_GRAPHS");
    if (e && (e[0] == '0' || e[0] == '1')) p->graphs_on = e[0] == '1';
  }
  if (hipGetDevice(&p->device) != hipSuccess) { delete p; set_error("mpn_frcnn_create: no current HIP device"); return MPN_EHIP; }
  p->scratch.device = p->device;
  ScratchScope scratch_scope(&p->scratch);
  const int tower_heads = rw ? rw->n_heads : (gw ? gw->n_heads : 0);  // > 1: MultiPathNet towers on a ResNet / op-list backbone
  const int *tower_region = rw ? rw->head_region : (gw ? gw->head_region : nullptr);
  if (tower_heads > 1) p->n_integral = (rw ? rw->n_integral : gw->n_integral) > 0 ? (rw ? rw->n_integral : gw->n_integral) : 1;
  if (mw) { p->is_mpnet = true; p->tap3 = mw->tap_conv3; p->tap4 = mw->tap_conv4; p->n_integral = mw->n_integral > 0 ? mw->n_integral : 1; p->conv345_norm = !mw->conv345_unnormalized; }
  const int n_conv = graph_net ? 0 : cfg->n_conv;
  p->cfg.n_conv = n_conv;
  if (n_conv) { p->cout.assign(cfg->conv_cout, cfg->conv_cout + n_conv); p->pool_after.assign(cfg->pool_after, cfg->pool_after + n_conv); }
  p->cfg.conv_cout = p->cout.data();
  p->cfg.pool_after = p->pool_after.data();
  int rc = MPN_OK;
#define TRY(x) do { rc = (x); if (rc != MPN_OK) { mpn_frcnn_destroy(p); return rc; } } while (0)
  TRY((cfg->fc_arith == MPN_FC_FP32 || (cfg->fc_arith == MPN_FC_SPLIT3 && !graph_net)) ? MPN_OK : (set_error("mpn_frcnn_config.fc_arith: %d (MPN_FC_SPLIT3 is for mpn_frcnn_create / mpn_mpnet_create pipelines)", cfg->fc_arith), MPN_EINVAL));
  TRY((cfg->roi_bin_rule == MPN_ROI_BINS_CAFFE || cfg->roi_bin_rule == MPN_ROI_BINS_ADAPTIVE) ? MPN_OK : (set_error("mpn_frcnn_config.roi_bin_rule: %d is not an MPN_ROI_BINS_* value", cfg->roi_bin_rule), MPN_EINVAL));
  // ---- trunk buffers + packed weights
  int h = cfg->max_h, w = cfg->max_w, cin = 3;
  size_t b = act_bytes(3, h, w);
  TRY(dev_alloc(p, &p->img_c8p, b, true));
  p->act_bufs.push_back({p->img_c8p, b});
  for (int l = 0; l < n_conv; ++l) {
    MPN_CHECK_ARG(d_conv_w[l] != nullptr);
    ConvLayer L;
    L.Cin = cin; L.Cout = p->cout[l]; L.pool = p->pool_after[l];
    TRY(dev_alloc(p, &L.wpk, conv_wpk_elems(L.Cin, L.Cout) * sizeof(float), false));
    TRY(dev_alloc(p, &L.bpk, (size_t)conv_coutp(L.Cout) * sizeof(float), false));
    TRY(pack_conv_weights(d_conv_w[l], d_conv_b[l], L.Cin, L.Cout, L.wpk, L.bpk, nullptr));
    if (L.Cin <= 4 && !L.pool && !(mw && (l == mw->tap_conv3 || l == mw->tap_conv4))) {
      TRY(dev_alloc(p, &L.w36, conv_first_elems(L.Cout) * sizeof(float), false));
      TRY(pack_conv_weights_first(d_conv_w[l], L.Cin, L.Cout, L.w36, nullptr));
    }
    if (L.Cin >= 16) {  // fewer input channels: direct kernels (the 3-channel first layer is bound by its output stores either way)
      TRY(dev_alloc(p, &L.wino, conv_wino_elems(L.Cin, L.Cout) * sizeof(float), false));
      TRY(pack_conv_weights_wino(d_conv_w[l], L.Cin, L.Cout, L.wino, nullptr));
    }
    b = act_bytes(L.Cout, h, w);
    TRY(dev_alloc(p, &L.out, b, true));
    p->act_bufs.push_back({L.out, b});
    if (mw && (l == mw->tap_conv3 || l == mw->tap_conv4 || l == n_conv - 1)) {  // range-max tables of the maps the towers pool
      const int slot = l == n_conv - 1 ? 0 : (l == mw->tap_conv4 ? 1 : 2);
      TRY(dev_alloc(p, &p->vmax_tab[slot], (size_t)(vmax_levels_for(h) + 1) * b, false));  // levels 1..L (C8P form) or 0..L (pixel-major form)
    }
    if (L.pool) {
      h = (h + 1) / 2; w = (w + 1) / 2;
      b = act_bytes(L.Cout, h, w);
      TRY(dev_alloc(p, &L.pooled, b, true));
      p->act_bufs.push_back({L.pooled, b});
    }
    cin = L.Cout;
    p->conv.push_back(L);
  }
  p->feat_c = cin;
  if (graph_net) {  // ResNet / op-list graph: the graph object owns the trunk and per-ROI weights and activations; the cls + bbox heads read its pooled vector
    if (rw) TRY(resnet_build(rw, cfg->max_h, cfg->max_w, cfg->max_rois, cfg->pooled_h, &p->rn));
    else TRY(graph_build(gw, cfg->max_h, cfg->max_w, cfg->max_rois, cfg->pooled_h, &p->rn));
    p->feat_c = resnet_feat_channels(p->rn);
    resnet_set_roi_bins(p->rn, cfg->roi_bin_rule);
  }
  // ---- head
  const int PP = cfg->pooled_h * cfg->pooled_w, C = cfg->n_classes, F = graph_net ? resnet_out_channels(p->rn) : cfg->fc_dim;
  p->cfg.fc_dim = F;
  MPN_CHECK_ARG(p->feat_c % 8 == 0);
  p->K6 = p->feat_c * PP;
  p->Mp = lin_mp(cfg->max_rois);
  p->n_head = 5 * C;
  const int K6_32 = round_up(p->K6, 64), F32 = round_up(F, 64);
  const size_t M = cfg->max_rois;
  if (mw) {
    MPN_CHECK_ARG(mw->n_towers >= 2 && mw->n_towers <= 8 && p->tap3 >= 0 && p->tap4 > p->tap3 && p->tap4 < cfg->n_conv - 1);
    MPN_CHECK_ARG(C <= 256 && F % 128 == 0);
    const int c5 = p->feat_c, c4 = p->cout[p->tap4], c3 = p->cout[p->tap3];
    MPN_CHECK_ARG(c4 % 8 == 0 && c3 % 8 == 0);
    int max_feat = 0;
    for (int t = 0; t < mw->n_towers; ++t) {
      mpn_frcnn::Tower T{};
      T.region = mw->region[t]; T.use4 = mw->use_conv4[t]; T.use3 = mw->use_conv3[t];
      MPN_CHECK_ARG(T.region >= 0 && T.region < 4 && mw->mix_w[t] && mw->fc6_w[t] && mw->fc7_w[t]);
      T.total_feat = c5 + (T.use4 ? c4 : 0) + (T.use3 ? c3 : 0);
      if (T.total_feat > max_feat) max_feat = T.total_feat;
      const int TF64 = round_up(T.total_feat, 64);
      TRY(dev_alloc(p, &T.mix_w, lin_wpk_elems(TF64, c5) * sizeof(float), false));
      TRY(dev_alloc(p, &T.mix_b, (size_t)lin_np(c5) * sizeof(float), false));
      TRY(pack_linear_weights(mw->mix_w[t], mw->mix_b[t], T.total_feat, c5, 1, T.mix_w, T.mix_b, nullptr));
      TRY(dev_alloc(p, &T.w6, lin_wpk_elems(K6_32, F) * sizeof(float), false));
      TRY(dev_alloc(p, &T.b6, (size_t)lin_np(F) * sizeof(float), false));
      TRY(pack_linear_weights(mw->fc6_w[t], mw->fc6_b[t], p->K6, F, PP, T.w6, T.b6, nullptr));
      TRY(dev_alloc(p, &T.w7, lin_wpk_elems(F32, F) * sizeof(float), false));
      TRY(dev_alloc(p, &T.b7, (size_t)lin_np(F) * sizeof(float), false));
      TRY(pack_linear_weights(mw->fc7_w[t], mw->fc7_b[t], F, F, 1, T.w7, T.b7, nullptr));
      if (cfg->fc_arith == MPN_FC_SPLIT3) {  // the towers' fc6 / fc7 weights as three bf16 planes (include/mpn.h)
        float *tmp = nullptr;
        TRY(dev_alloc(p, &tmp, split3_plane_elems(p->K6, lin_np(F)) * sizeof(unsigned short), true));
        T.w6_s3 = reinterpret_cast<unsigned short *>(tmp);
        TRY(split3_planes(T.w6, p->K6, lin_np(F), lin_np(F), T.w6_s3, nullptr));
        TRY(dev_alloc(p, &tmp, split3_plane_elems(F, lin_np(F)) * sizeof(unsigned short), true));
        T.w7_s3 = reinterpret_cast<unsigned short *>(tmp);
        TRY(split3_planes(T.w7, F, lin_np(F), lin_np(F), T.w7_s3, nullptr));
      }
      p->towers.push_back(T);
    }
    const int n_fov = mw->n_towers - 1, K = p->n_integral;
    const int KC64 = round_up(n_fov * F, 64);
    TRY(dev_alloc(p, &p->wcls, lin_wpk_elems(KC64, K * C) * sizeof(float), false));
    TRY(dev_alloc(p, &p->bcls, (size_t)lin_np(K * C) * sizeof(float), false));
    TRY(pack_linear_weights(d_cls_w, d_cls_b, n_fov * F, K * C, 1, p->wcls, p->bcls, nullptr));
    TRY(dev_alloc(p, &p->wbbox, lin_wpk_elems(F32, 4 * C) * sizeof(float), false));
    TRY(dev_alloc(p, &p->bbbox, (size_t)lin_np(4 * C) * sizeof(float), false));
    TRY(pack_linear_weights(d_bbox_w, d_bbox_b, F, 4 * C, 1, p->wbbox, p->bbbox, nullptr));
    const size_t rows = (size_t)PP * p->Mp;
    TRY(dev_alloc(p, &p->fov, M * 20 * sizeof(float), true));
    TRY(dev_alloc(p, &p->mix_scale, (size_t)3 * 3 * p->Mp * sizeof(float), true));
    TRY(dev_alloc(p, &p->tx2, (size_t)(round_up(max_feat, 64) / 8) * rows * 8 * sizeof(float), true));
    for (int a = 0; a < (int)p->towers.size() && p->share_provider < 0; ++a)      // the first (provider, consumer) pair, if any
      for (int b = 0; b < (int)p->towers.size() && p->share_provider < 0; ++b) {
        const mpn_frcnn::Tower &A = p->towers[a], &B = p->towers[b];
        if (a == b || A.region != B.region || A.total_feat <= B.total_feat) continue;
        int la[3], lb[3], na = 0, nb = 0;   // the towers' map lists in concat order (conv345Combine: conv5, [conv4], [conv3])
        la[na++] = 0; if (A.use4) la[na++] = 1; if (A.use3) la[na++] = 2;
        lb[nb++] = 0; if (B.use4) lb[nb++] = 1; if (B.use3) lb[nb++] = 2;
        bool prefix = nb <= na;
        for (int i = 0; prefix && i < nb; ++i) prefix = la[i] == lb[i];
        if (prefix) { p->share_provider = a; p->share_consumer = b; }
      }
    if (p->share_provider >= 0) TRY(dev_alloc(p, &p->tx3, (size_t)(round_up(max_feat, 64) / 8) * rows * 8 * sizeof(float), true));
    {
      hipError_t e = hipSuccess;
      p->pool_on_side = true;   // (p->side is created below)
      for (int i = 0; i < 3 && e == hipSuccess; ++i) {
        e = hipEventCreateWithFlags(&p->ev_pool_done[i], hipEventDisableTiming);
        if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_mix_done[i], hipEventDisableTiming);
      }
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_pool_go, hipEventDisableTiming);
      if (e != hipSuccess) { set_error("mpn_mpnet_create: pooling events: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
    }
    TRY(dev_alloc(p, &p->tx, (size_t)(round_up(max_feat, 64) / 8) * rows * 8 * sizeof(float), true));
    TRY(dev_alloc(p, &p->ty, (size_t)(lin_np(c5) / 8) * rows * 8 * sizeof(float), true));
    TRY(dev_alloc(p, &p->tz6, (size_t)(lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
    if (cfg->fc_arith == MPN_FC_SPLIT3)
      for (int ln = 0; ln < (mw->n_towers > 1 ? 2 : 1); ++ln) {
        float *tmp = nullptr;
        TRY(dev_alloc(p, &tmp, split3_plane_elems(p->K6, p->Mp) * sizeof(unsigned short), true));
        p->ty_s3[ln] = reinterpret_cast<unsigned short *>(tmp);
        TRY(dev_alloc(p, &tmp, split3_plane_elems(F, p->Mp) * sizeof(unsigned short), true));
        p->tz6_s3[ln] = reinterpret_cast<unsigned short *>(tmp);
      }
    if (mw->n_towers > 1) {  // the second tower lane
      TRY(dev_alloc(p, &p->ty2, (size_t)(lin_np(c5) / 8) * rows * 8 * sizeof(float), true));
      TRY(dev_alloc(p, &p->tz6_2, (size_t)(lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
      hipError_t e = hipStreamCreateWithFlags(&p->tower_stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_lane_go, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_lane_done, hipEventDisableTiming);
      if (e != hipSuccess) { set_error("mpn_mpnet_create: tower stream / events: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
    }
    TRY(dev_alloc(p, &p->cat, (size_t)mw->n_towers * (lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
    TRY(dev_alloc(p, &p->cls_rm, M * K * C * sizeof(float), true));
    TRY(dev_alloc(p, &p->bbox_rm, M * 4 * C * sizeof(float), true));
  } else if (tower_heads > 1) {  // ResNet / graph towers (this library's extension, see mpn_resnet_weights): same classifier stage as MultiPathNet
    const int n_fov = tower_heads - 1, K = p->n_integral;
    MPN_CHECK_ARG(C <= 256 && tower_heads <= 8);
    const int KC64 = round_up(n_fov * F, 64);
    TRY(dev_alloc(p, &p->wcls, lin_wpk_elems(KC64, K * C) * sizeof(float), false));
    TRY(dev_alloc(p, &p->bcls, (size_t)lin_np(K * C) * sizeof(float), false));
    TRY(pack_linear_weights(d_cls_w, d_cls_b, n_fov * F, K * C, 1, p->wcls, p->bcls, nullptr));
    TRY(dev_alloc(p, &p->wbbox, lin_wpk_elems(F32, 4 * C) * sizeof(float), false));
    TRY(dev_alloc(p, &p->bbbox, (size_t)lin_np(4 * C) * sizeof(float), false));
    TRY(pack_linear_weights(d_bbox_w, d_bbox_b, F, 4 * C, 1, p->wbbox, p->bbbox, nullptr));
    TRY(dev_alloc(p, &p->fov, M * 20 * sizeof(float), true));
    TRY(dev_alloc(p, &p->cat, (size_t)tower_heads * (lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
    TRY(dev_alloc(p, &p->cls_rm, M * K * C * sizeof(float), true));
    TRY(dev_alloc(p, &p->bbox_rm, M * 4 * C * sizeof(float), true));
    for (int t = 0; t < tower_heads; ++t) { MPN_CHECK_ARG(tower_region[t] >= 0 && tower_region[t] < 4); p->rn_region.push_back(tower_region[t]); }
    if (resnet_has_second_lane(p->rn)) {  // the second tower lane's stream (run_detect; debug flavour only)
      hipError_t e = hipStreamCreateWithFlags(&p->tower_stream, hipStreamNonBlocking);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_lane_go, hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_lane_done, hipEventDisableTiming);
      if (e != hipSuccess) { set_error("mpn_resnet_create / mpn_graph_create: tower stream / events: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
    }
  } else {
  if (!graph_net) {
  TRY(dev_alloc(p, &p->w6, lin_wpk_elems(K6_32, F) * sizeof(float), false));
  TRY(dev_alloc(p, &p->b6, (size_t)lin_np(F) * sizeof(float), false));
  TRY(pack_linear_weights(d_fc6_w, d_fc6_b, p->K6, F, PP, p->w6, p->b6, nullptr));
  if (cfg->fc_arith == MPN_FC_SPLIT3) {  // the packed fp32 weights [K/8][NP][8] split once into three bf16 planes [3][K/8][NP↑256][8]; the operand's planes per image
    float *tmp = nullptr;
    TRY(dev_alloc(p, &tmp, split3_plane_elems(p->K6, lin_np(F)) * sizeof(unsigned short), true));
    p->w6_s3 = reinterpret_cast<unsigned short *>(tmp);
    TRY(split3_planes(p->w6, p->K6, lin_np(F), lin_np(F), p->w6_s3, nullptr));
    TRY(dev_alloc(p, &tmp, split3_plane_elems(p->K6, p->Mp) * sizeof(unsigned short), true));
    p->x6_s3 = reinterpret_cast<unsigned short *>(tmp);
  }
  TRY(dev_alloc(p, &p->w7, lin_wpk_elems(F32, F) * sizeof(float), false));
  TRY(dev_alloc(p, &p->b7, (size_t)lin_np(F) * sizeof(float), false));
  TRY(pack_linear_weights(d_fc7_w, d_fc7_b, F, F, 1, p->w7, p->b7, nullptr));
  if (cfg->fc_arith == MPN_FC_SPLIT3) {
    float *tmp = nullptr;
    TRY(dev_alloc(p, &tmp, split3_plane_elems(F, lin_np(F)) * sizeof(unsigned short), true));
    p->w7_s3 = reinterpret_cast<unsigned short *>(tmp);
    TRY(split3_planes(p->w7, F, lin_np(F), lin_np(F), p->w7_s3, nullptr));
    TRY(dev_alloc(p, &tmp, split3_plane_elems(F, p->Mp) * sizeof(unsigned short), true));
    p->y6_s3 = reinterpret_cast<unsigned short *>(tmp);
  }
  }
  {  // cls and bbox heads share their input -> one [5C, F] GEMM (model_utils.lua:105-119 ConcatTable)
    float *tmp_w = nullptr, *tmp_b = nullptr;
    TRY(dev_alloc(p, &tmp_w, (size_t)5 * C * F * sizeof(float), false));
    TRY(dev_alloc(p, &tmp_b, (size_t)5 * C * sizeof(float), true));
    hipError_t e = hipMemcpy(tmp_w, d_cls_w, (size_t)C * F * sizeof(float), hipMemcpyDeviceToDevice);
    if (e == hipSuccess) e = hipMemcpy(tmp_w + (size_t)C * F, d_bbox_w, (size_t)4 * C * F * sizeof(float), hipMemcpyDeviceToDevice);
    if (e == hipSuccess && d_cls_b) e = hipMemcpy(tmp_b, d_cls_b, (size_t)C * sizeof(float), hipMemcpyDeviceToDevice);
    if (e == hipSuccess && d_bbox_b) e = hipMemcpy(tmp_b + C, d_bbox_b, (size_t)4 * C * sizeof(float), hipMemcpyDeviceToDevice);
    if (e != hipSuccess) { set_error("mpn_frcnn_create: head weight copy failed: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
    TRY(dev_alloc(p, &p->wh, lin_wpk_elems(F32, 5 * C) * sizeof(float), false));
    TRY(dev_alloc(p, &p->bh, (size_t)lin_np(5 * C) * sizeof(float), false));
    TRY(pack_linear_weights(tmp_w, tmp_b, F, 5 * C, 1, p->wh, p->bh, nullptr));
  }
  if (!graph_net) {
  {  // pixel-major copy of the final map at its largest size
    int fh = cfg->max_h, fw = cfg->max_w;
    for (int l = 0; l < n_conv; ++l) if (p->pool_after[l]) { fh = (fh + 1) / 2; fw = (fw + 1) / 2; }
    TRY(dev_alloc(p, &p->feat_pm, (size_t)fh * fw * ((p->feat_c + 7) / 8) * 8 * sizeof(float), false));
  }
  TRY(dev_alloc(p, &p->x6, (size_t)(K6_32 / 8) * p->Mp * 8 * sizeof(float), true));
  TRY(dev_alloc(p, &p->y6, (size_t)(lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
  }
  TRY(dev_alloc(p, &p->y7, (size_t)(lin_np(F) / 8) * p->Mp * 8 * sizeof(float), true));
  TRY(dev_alloc(p, &p->head, M * 5 * C * sizeof(float), true));
  }
  TRY(dev_alloc(p, &p->rois, M * 5 * sizeof(float), true));
  TRY(dev_alloc(p, &p->scores, M * C * sizeof(float), true));
  TRY(dev_alloc(p, &p->bbox, M * 4 * C * sizeof(float), true));
  TRY(dev_alloc(p, &p->bbox_raw, M * 4 * C * sizeof(float), true));
  const int n_it = cfg->num_iter > 1 ? cfg->num_iter : 1;
  MPN_CHECK_ARG((size_t)n_it * M <= MPN_NMS_MAX_BOXES);
  MPN_CHECK_ARG(!cfg->use_rbox_scores || n_it > 1);  // Tester_FRCNN.lua:92 assert(#all_output > 1)
  p->cfg.num_iter = n_it;
  const size_t MR = M * n_it;  // rows that reach NMS per class
  if (n_it > 1) {
    TRY(dev_alloc(p, &p->it_scores, MR * C * sizeof(float), true));
    TRY(dev_alloc(p, &p->it_bbox, MR * 4 * C * sizeof(float), true));
    TRY(dev_alloc(p, &p->it_boxes, M * 4 * sizeof(float), true));
  }
  for (int i = 0; i < 2; ++i) {
    TRY(dev_alloc(p, &p->scored_b[i], (size_t)(C - 1) * MR * 5 * sizeof(float), true));
    TRY(dev_alloc(p, &p->keep_b[i], (size_t)(C - 1) * MR * 5 * sizeof(float), true));
    if (cfg->bbox_voting) TRY(dev_alloc(p, &p->voted_b[i], (size_t)(C - 1) * MR * 5 * sizeof(float), true));
    TRY(dev_alloc(p, &p->keep_idx_b[i], (size_t)(C - 1) * MR * sizeof(int), true));
    TRY(dev_alloc(p, &p->counts_b[i], (size_t)(C - 1) * sizeof(int), true));
    TRY(dev_alloc(p, &p->n_keep_b[i], (size_t)(C - 1) * sizeof(int), true));
    TRY(dev_alloc(p, &p->thresh_b[i], 16, true));
  }
  p->scored = p->scored_b[0]; p->keep = p->keep_b[0]; p->keep_idx = p->keep_idx_b[0];
  p->counts = p->counts_b[0]; p->n_keep = p->n_keep_b[0]; p->thresh = p->thresh_b[0];
  {
    int lo = 0, hi = 0;
    hipError_t e = hipDeviceGetStreamPriorityRange(&lo, &hi);
    // The side stream (NMS / top-k tail under the next image's trunk) has the DEFAULT priority since round 5.  Rounds 1-4 created it with
    // the highest one; what that bought the headline is nothing measurable (3.457-3.476 ms either way, six alternating runs), and what it can
    // cost is large: a priority stream lands in a different hardware-queue class, and depending on which queue HIP's round-robin hands it,
    // EVERY dispatch of the launch queue took 30-50 us longer while the two queues were both active — a mixed-size stream inside bench.py ran
    // at 4.59 ms per image with the highest priority, 4.05 with the lowest, 3.12 with the default (profiles/r05_mixed_sizes_timeline.txt).
    (void)lo; (void)hi;
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&p->side, hipStreamNonBlocking);
    for (int i = 0; i < 2 && e == hipSuccess; ++i) {
      e = hipEventCreateWithFlags(&p->ev_head[i], hipEventDisableTiming);
      if (e == hipSuccess) e = hipEventCreateWithFlags(&p->ev_tail[i], hipEventDisableTiming);
    }
    if (e != hipSuccess) { set_error("mpn_frcnn_create: side stream/events: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
    if (p->pool_on_side) p->pool_stream = p->side;
  }
#undef TRY
  hipError_t e = hipDeviceSynchronize();
  if (e != hipSuccess) { set_error("mpn_frcnn_create: %s", hipGetErrorString(e)); mpn_frcnn_destroy(p); return MPN_EHIP; }
  *out = p;
  return MPN_OK;
}

static int run_trunk(mpn_frcnn *p, const float *d_image, int H, int W, hipStream_t s, Act *feat_out) {
  const mpn_frcnn_config &c = p->cfg;
  if (H != p->last_h || W != p->last_w) {  // halo positions move with the image size: re-lay the zero halo of every activation, once
    if (g_halo_memset) {
      for (auto &b : p->act_bufs) MPN_CHECK_HIP(hipMemsetAsync(b.first, 0, b.second, s));
    } else {
      std::vector<Act> acts;
      acts.push_back(make_act(p->img_c8p, 3, H, W));
      int hh = H, ww = W;
      for (auto &L : p->conv) {
        if (L.out) acts.push_back(make_act(L.out, L.Cout, hh, ww));
        if (L.pool) { hh = (hh + 1) / 2; ww = (ww + 1) / 2; if (L.pooled) acts.push_back(make_act(L.pooled, L.Cout, hh, ww)); }
      }
      int rc_h = c8p_zero_halos(acts.data(), (int)acts.size(), s);
      if (rc_h) return rc_h;
    }
    p->last_h = H; p->last_w = W;
  }
  Act cur = make_act(p->img_c8p, 3, H, W);
  p->vmax_built[0] = p->vmax_built[1] = p->vmax_built[2] = false;
  p->feat_pm_valid = false;
  int rc;
  { ProfScope ps(p, MPN_PROF_TRANSFORM, s);
    rc = image_transform_c8p(d_image, H, W, c.tf_swap, c.tf_scale, c.tf_mean, c.tf_std, c.tf_std[0] != 0.0, cur, s); }
  if (rc) return rc;
  int h = H, w = W;
  int li = 0;
  for (auto &L : p->conv) {
    Act out = make_act(L.out, L.Cout, h, w);
    const int ctag = conv3x3_variant_for(L.Cout, L.wino != nullptr) == 7 ? MPN_PROF_CONV_WINO : MPN_PROF_CONV_DIRECT;
    const bool is_tap = p->is_mpnet && (li == p->tap3 || li == p->tap4);
    if (is_tap) p->tap_act[li == p->tap4 ? 1 : 2] = out;
    ++li;
    if (L.pool) {
      Act pooled = make_act(L.pooled, L.Cout, (h + 1) / 2, (w + 1) / 2);
      if (g_fuse_pool) {
        ProfScope ps(p, ctag, s);
        rc = conv3x3_c8p(cur, L.wpk, L.bpk, L.Cout, 1, is_tap ? out : Act{}, pooled, s, L.wino);  // tap layers keep the pre-pool map too
      } else {
        { ProfScope ps(p, ctag, s); rc = conv3x3_c8p(cur, L.wpk, L.bpk, L.Cout, 1, out, Act{}, s, L.wino); }
        if (rc == MPN_OK) { ProfScope ps(p, MPN_PROF_POOL, s); rc = maxpool2x2_c8p(out, pooled, s); }
      }
      if (rc) return rc;
      cur = pooled; h = pooled.H; w = pooled.W;
    } else {
      { ProfScope ps(p, ctag, s);
        rc = (L.w36 && g_first_k36) ? conv3x3_first_c8p(cur, L.w36, L.bpk, L.Cout, 1, out, s) : conv3x3_c8p(cur, L.wpk, L.bpk, L.Cout, 1, out, Act{}, s, L.wino); }
      if (rc) return rc;
      cur = out;
    }
  }
  *feat_out = cur;
  p->tap_act[0] = cur;
  return MPN_OK;
}

// models/multipathnet.lua:64-120 + model_utils.lua:209-251,296-313 on the C8 layouts: Foveal -> per tower
// {ROI pools of conv5 / conv4 / conv3 written side by side = the channel concat, per-map L2 normalise * 1000,
// 1x1 conv mix as a GEMM over (bin, roi) rows whose output IS the fc6 operand, fc6, fc7 into the tower concat}
// -> K integral classifiers (mean of softmaxes) + bbox regressor on the "het" tower.
static int run_integral_heads(mpn_frcnn *p, const float *d_boxes, int N, int H, int W, int n_fov, hipStream_t s, int clamp);

static int run_mpnet_head(mpn_frcnn *p, const float *d_boxes, int N, int H, int W, hipStream_t s, int clamp) {
  const mpn_frcnn_config &c = p->cfg;
  const int F = c.fc_dim, PP = c.pooled_h * c.pooled_w, Mp = lin_mp(N);
  int rc = mpn_foveal_forward(p->rois, N, p->fov, s);
  if (rc) return rc;
  const Act maps[3] = {p->tap_act[0], p->tap_act[1], p->tap_act[2]};
  const float scales[3] = {c.spatial_scale, c.spatial_scale * 2.0f, c.spatial_scale * 4.0f};
  const float kConv345Factor[3] = {1.0f, (float)(1.0 / 30), (float)(1.0 / 200)};  // model_utils.lua:231-237 normFactor (Lua doubles -> float)
  const int Fcb = lin_np(F) / 8;
  const bool pm = g_roi_pool_pm != 0;  // pixel-major range-max tables + fused sum of squares (round 3); 0 = the C8P form (test hook)
  // The range-max tables: once per trunk run and map (iterative localisation on the cached maps reuses them).  Round 6: with the pooling
  // stream, a map's tables are built THERE, where the first pooling that reads them is enqueued — the first tower in execution order pools
  // conv5 alone (cheapest first), so only conv5's small tables (1/7 of the bytes) stand between the trunk and the first GEMM; conv4's and
  // conv3's levels (HBM-bound writes) are built under that tower's GEMMs instead of in front of everything.
  if (p->vmax_pm != pm) { p->vmax_built[0] = p->vmax_built[1] = p->vmax_built[2] = false; p->vmax_pm = pm; }
  auto build_tables = [&](int m, hipStream_t st) -> int {
    if (p->vmax_built[m]) return MPN_OK;
    ProfScope ps(p, MPN_PROF_ROIPOOL, st);
    const int rcb = pm ? build_vmax_tables_pm(maps[m], p->vmax_tab[m], st) : build_vmax_tables(maps[m], p->vmax_tab[m], st);
    if (rcb == MPN_OK) p->vmax_built[m] = true;
    return rcb;
  };
  // nn.Normalize's per-(ROI, map) scale folded into the mix GEMM (linear_c8_rowscaled).  Fold or in-place l2norm_apply is decided by
  // the network alone (channel counts), never by the ROI count, and the mix GEMM — K = the channel concat, >= 49 row tiles for any N —
  // always runs as ONE un-split accumulation chain, so a ROI's mix output does not depend on the rows it is batched with (ADVICE r3:
  // a 128-row shard of a feat_c <= 256 trunk used to take l2norm_apply + split-K where the full table took the fold)
  const bool fold_scale = pm && p->conv345_norm && p->mix_scale && g_mix_fold &&
                          maps[0].C % 32 == 0 && maps[1].C % 32 == 0 && maps[2].C % 32 == 0;  // segment boundaries fall on 32-k stages
  // Two streams: a tower's skip pooling is L2 -> L1 bound (no matrix work), its GEMMs are matrix-bound, and the folding GEMM leaves
  // half of every CU's registers and 94 KB of LDS free.  Tower t's pooling therefore runs on the handle's pooling stream, into the
  // operand buffer / scale set of parity t & 1, under tower t - 1's GEMMs on `s`; `s` waits for the pooling only where the mix GEMM
  // starts, the pooling stream waits for the mix GEMM of tower t - 2 (the previous user of its buffers).  Every result is still
  // ordered on `s`.  (Profiling scopes time each stream's own work; with overlap their sum exceeds the wall time.)
  // Rows of the pooled operand = the mix GEMM's rows = (bin, roi).  With the scale fold (the product path) they are PACKED: Nr = N rounded up to 8
  // rows per bin instead of the Mp = N rounded up to 128 the fc GEMMs' operands carry, and the mix GEMM scatters its output rows into fc6's
  // [cout block][bin][Mp][8] operand (GemmArgs::bin_rows).  1000 proposals: 49 x 1000 = 383 row tiles instead of 49 x 1024 = 392 — 1532 blocks
  // instead of 1568 on 512 resident slots, i.e. 2.99 rounds instead of 3.06 (a whole block time per mix GEMM); 300 proposals: 115 instead of 147.
  const int Nr = (fold_scale && g_mix_packed) ? round_up(N, 8) : Mp;
  const int R = PP * Nr, Rp = round_up(R, 128);
  const bool overlap = pm && p->pool_stream && p->tx2 && g_pool_overlap;
  hipStream_t ps_stream = overlap ? p->pool_stream : s;
  if (!(overlap && g_tables_lazy))   // one stream (or the round-3..5 order, hook tables_lazy = 0): all tables up front
    for (int m = 0; m < 3; ++m) if ((rc = build_tables(m, s)) != MPN_OK) return rc;
  if (overlap) {  // the tables, the Foveal regions and everything before them on `s`
    MPN_CHECK_HIP(hipEventRecord(p->ev_pool_go, s));
    MPN_CHECK_HIP(hipStreamWaitEvent(ps_stream, p->ev_pool_go, 0));
  }
  const int n_tow = (int)p->towers.size();
  MPN_CHECK_ARG(n_tow <= 8);
  // ---- the operand plan of this image: which pooled-operand buffer each tower's mix GEMM reads, and which tower's arrival pools it.
  // Plain towers alternate between tx and tx2 (buffers 0 / 1); the (provider, consumer) pair of mpn_frcnn::share_provider shares tx3
  // (buffer 2): whichever of the two comes first pools the PROVIDER's maps there, the other one finds them.  prev_user = the tower whose
  // mix GEMM read the buffer last (its ev_mix_done is what a re-pooling waits for; by construction it is at least two towers back, so
  // its mix GEMM has been enqueued whenever the pooling is).
  const bool share = overlap && p->tx3 && p->share_provider >= 0 && g_tower_share;
  float *const tx_of[3] = {p->tx, p->tx2, p->tx3};
  int buf_of[8], prev_user[8], pool_src[8], order[8];
  bool pools[8];
  // Execution order: the towers meet only in `cat` (each writes its own slice), so the order is free — cheapest pooling first.  The first
  // tower's pooling has nothing to hide behind (it follows the trunk directly): 512 channels (the conv5-only tower) instead of 1280.
  for (int t = 0; t < n_tow; ++t) order[t] = t;
  if (overlap && g_tower_order) {
    auto cost = [&](int t) { return (share && (t == p->share_provider || t == p->share_consumer)) ? p->towers[p->share_provider].total_feat : p->towers[t].total_feat; };
    std::stable_sort(order, order + n_tow, [&](int x, int y) { return cost(x) < cost(y); });
  }
  {
    int nplain = 0, last_user[3] = {-1, -1, -1};
    bool shared_pooled = false;
    for (int k = 0; k < n_tow; ++k) {
      const int t = order[k];
      if (share && (t == p->share_provider || t == p->share_consumer)) {
        buf_of[t] = 2; pools[t] = !shared_pooled; shared_pooled = true; pool_src[t] = p->share_provider;
      } else {
        buf_of[t] = overlap ? (nplain & 1) : 0; ++nplain; pools[t] = true; pool_src[t] = t;
      }
      prev_user[t] = pools[t] ? last_user[buf_of[t]] : -1;
      last_user[buf_of[t]] = t;
    }
  }
  GemmRowScale grs_of[8];
  for (int t = 0; t < n_tow; ++t) {  // the per-(map, ROI) scale vectors tower t's mix GEMM applies: those of its buffer, its own K segments
    const mpn_frcnn::Tower &T = p->towers[t];
    GemmRowScale &grs = grs_of[t];
    grs = GemmRowScale{};
    grs.rs_mod = Nr;
    if (!fold_scale) continue;
    if (Nr != Mp) { grs.bin_rows = Nr; grs.out_Mp = Mp; grs.x_pitch = R; }
    const int used[3] = {1, T.use4, T.use3};
    int cb_off = 0;
    for (int m = 0; m < 3; ++m) {
      if (!used[m]) continue;
      grs.scale[grs.n_seg] = p->mix_scale + ((size_t)buf_of[t] * 3 + grs.n_seg) * p->Mp;
      if (grs.n_seg < 2) grs.k_end[grs.n_seg] = (cb_off + maps[m].Cb()) * 8;
      ++grs.n_seg;
      cb_off += maps[m].Cb();
    }
  }
  auto pool_tower = [&](int t) -> int {  // pooling (+ normalisation scales) of tower t's operand on the pooling stream
    if (!pools[t]) return MPN_OK;          // the other tower of the shared pair pooled it
#ifdef MPN_DEBUG_HOOKS
    if (g_mpn_pool_knock) {                // mpn_debug_set_mpn_pool_knock (timing only, stale operands): what the skip pooling costs the towers' GEMMs
      if (overlap) MPN_CHECK_HIP(hipEventRecord(p->ev_pool_done[buf_of[t]], ps_stream));
      return MPN_OK;
    }
#endif
    const mpn_frcnn::Tower &T = p->towers[pool_src[t]];
    const int b = buf_of[t];
    float *txb = tx_of[b];
    const float *reg = p->fov + 5 * T.region;  // rows 4n + region of the Foveal table
    if (overlap && prev_user[t] >= 0) MPN_CHECK_HIP(hipStreamWaitEvent(ps_stream, p->ev_mix_done[b], 0));  // the buffer's last reader
    int cb_off = 0, seg = 0;
    const int used[3] = {1, T.use4, T.use3};
    int rcl = MPN_OK;
    for (int m = 0; m < 3; ++m) {
      if (!used[m]) continue;
      float *dst = txb + (size_t)cb_off * R * 8;
      if ((rcl = build_tables(m, ps_stream)) != MPN_OK) return rcl;   // (first reader of the map's tables this trunk run)
      { ProfScope ps(p, MPN_PROF_ROIPOOL, ps_stream);
        if (pm) {
          float *sc_out = fold_scale ? p->mix_scale + ((size_t)b * 3 + seg) * p->Mp : nullptr;
          rcl = roi_pool_pm_rmq(maps[m], p->vmax_tab[m], reg, N, c.pooled_h, c.pooled_w, scales[m], RoiRule{1.0f, 0, c.roi_bin_rule}, dst, ps_stream, 20, Nr, p->conv345_norm ? 1 : 0,
                                p->conv345_norm ? 1000.0f : kConv345Factor[m], sc_out);
          ++seg;
        } else {
          rcl = roi_pool_c8_rmq(maps[m], p->vmax_tab[m], reg, N, c.pooled_h, c.pooled_w, scales[m], RoiRule{1.0f, 0, c.roi_bin_rule}, dst, ps_stream, 20, Nr);
          if (rcl == MPN_OK) rcl = p->conv345_norm ? l2norm_scale_c8(dst, maps[m].Cb() * PP, Nr, N, 1000.0f, ps_stream)
                                                   : mul_const_c8(dst, maps[m].Cb() * PP, Nr, N, kConv345Factor[m], ps_stream);
        } }
      if (rcl) return rcl;
      cb_off += maps[m].Cb();
    }
    if (overlap) MPN_CHECK_HIP(hipEventRecord(p->ev_pool_done[b], ps_stream));
    return MPN_OK;
  };
  // Two tower lanes (mpn_frcnn::tower_stream): tower ti runs on lane ti & 1 — its own stream, its own mix-output / fc6-output buffers.
  // The pooling stream runs one tower ahead of EACH lane (towers 0 and 1 up front, tower ti + 2 as soon as tower ti's mix GEMM is
  // enqueued).  The lanes join before the classifiers.  (profiling keeps one lane: its per-group scopes time the launch stream)
  const bool lanes = overlap && p->tower_stream && p->ty2 && p->tz6_2 && g_tower_lanes && n_tow > 1 && !p->prof;
  hipStream_t lane_s[2] = {s, lanes ? p->tower_stream : s};
  float *lane_ty[2] = {p->ty, lanes ? p->ty2 : p->ty}, *lane_tz6[2] = {p->tz6, lanes ? p->tz6_2 : p->tz6};
  if (lanes) {  // everything the towers read (Foveal regions, tables, the previous image's use of the lane buffers) is ordered on `s`
    MPN_CHECK_HIP(hipEventRecord(p->ev_lane_go, s));
    MPN_CHECK_HIP(hipStreamWaitEvent(lane_s[1], p->ev_lane_go, 0));
  }
  rc = pool_tower(order[0]);
  if (rc) return rc;
  if (lanes) { rc = pool_tower(order[1]); if (rc) return rc; }
  for (int k = 0; k < n_tow; ++k) {
    const int ti = order[k];
    const mpn_frcnn::Tower &T = p->towers[ti];
    const int b = buf_of[ti];
    const int ln = lanes ? (k & 1) : 0;
    hipStream_t ls = lane_s[ln];
    SplitkSlotScope lane_slabs(ln ? SCR_GEMM_SPLITK_LANE : SCR_GEMM_SPLITK);  // a GEMM of few tiles (small ROI shards) runs split-K: each lane its own slabs
    const float *txb = tx_of[b];
    if (overlap) {
      MPN_CHECK_HIP(hipStreamWaitEvent(ls, p->ev_pool_done[b], 0));
      if (!lanes && k + 1 < n_tow) { rc = pool_tower(order[k + 1]); if (rc) return rc; }  // enqueued now: runs under this tower's GEMMs
    }
    const GemmRowScale &grs = grs_of[ti];
    // 1x1 conv mix: rows = (bin, roi), K = concat channels, N = feat_c; output layout == fc6 operand layout
    { ProfScope ps(p, MPN_PROF_HEADS, ls);
      rc = fold_scale ? linear_c8_rowscaled(txb, R, T.total_feat, T.mix_w, T.mix_b, p->feat_c, 0, lane_ty[ln], ls, Rp, grs)
                      : linear_c8(txb, PP * Mp, T.total_feat, T.mix_w, T.mix_b, p->feat_c, 0, lane_ty[ln], nullptr, ls, PP * Mp, nullptr, 2); }
    if (rc) return rc;
    if (overlap) MPN_CHECK_HIP(hipEventRecord(p->ev_mix_done[b], ls));
    if (lanes && k + 2 < n_tow) { rc = pool_tower(order[k + 2]); if (rc) return rc; }  // waits for its buffer's last reader; runs under this lane's fc6 and the other lane's tower
    { ProfScope ps(p, MPN_PROF_FC6, ls);
      if (T.w6_s3) {  // MPN_FC_SPLIT3 (auxiliary arithmetic)
        rc = split3_planes(lane_ty[ln], p->K6, Mp, Mp, p->ty_s3[ln], ls);
        if (rc == MPN_OK) rc = linear_c8_split3(p->ty_s3[ln], N, p->K6, T.w6_s3, T.b6, F, 1, lane_tz6[ln], ls);
      } else
        rc = linear_c8(lane_ty[ln], N, p->K6, T.w6, T.b6, F, 1, lane_tz6[ln], nullptr, ls, Mp, nullptr, 1); }
    if (rc) return rc;
    { ProfScope ps(p, MPN_PROF_FC7, ls);
      if (T.w7_s3) {
        rc = split3_planes(lane_tz6[ln], F, Mp, Mp, p->tz6_s3[ln], ls);
        if (rc == MPN_OK) rc = linear_c8_split3(p->tz6_s3[ln], N, F, T.w7_s3, T.b7, F, 1, p->cat + (size_t)ti * Fcb * Mp * 8, ls);
      } else
        rc = linear_c8(lane_tz6[ln], N, F, T.w7, T.b7, F, 1, p->cat + (size_t)ti * Fcb * Mp * 8, nullptr, ls, Mp, nullptr, 1); }
    if (rc) return rc;
    if (!overlap && k + 1 < n_tow) { rc = pool_tower(order[k + 1]); if (rc) return rc; }
  }
  if (lanes) {
    MPN_CHECK_HIP(hipEventRecord(p->ev_lane_done, lane_s[1]));
    MPN_CHECK_HIP(hipStreamWaitEvent(s, p->ev_lane_done, 0));
  }
  return run_integral_heads(p, d_boxes, N, H, W, (int)p->towers.size() - 1, s, clamp);
}

// the stage after the towers (model_utils.lua:296-315, multipathnet.lua:112-120): K classifier clones on the concatenated
// classification towers -> mean of their softmaxes; box regressor on the last tower; decode + clamp
static int run_integral_heads(mpn_frcnn *p, const float *d_boxes, int N, int H, int W, int n_fov, hipStream_t s, int clamp) {
  const mpn_frcnn_config &c = p->cfg;
  const int C = c.n_classes, F = c.fc_dim, Mp = lin_mp(N), K = p->n_integral;
  const int Fcb = lin_np(F) / 8;
  int rc;
  { ProfScope ps(p, MPN_PROF_HEADS, s);
    rc = linear_c8(p->cat, N, n_fov * F, p->wcls, p->bcls, K * C, 0, nullptr, p->cls_rm, s, Mp, nullptr, 1);
    if (rc == MPN_OK) rc = linear_c8(p->cat + (size_t)n_fov * Fcb * Mp * 8, N, F, p->wbbox, p->bbbox, 4 * C, 0, nullptr, p->bbox_rm, s, Mp, nullptr, 1); }
  if (rc) return rc;
  ProfScope ps_post(p, MPN_PROF_POST, s);
  hipLaunchKernelGGL(integral_softmax_mean_kernel, dim3(cdiv(N, 4)), dim3(256), 0, s, p->cls_rm, N, K, C, p->scores);
  MPN_CHECK_LAUNCH();
  const bool norm = c.bbox_std[0] != 0.0f;
  hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)cdiv_sz((size_t)N * C, 256)), dim3(256), 0, s, p->bbox_rm, 4 * C, 0, N, C, d_boxes,
                     norm ? 1 : 0, c.bbox_mean[0], c.bbox_mean[1], c.bbox_mean[2], c.bbox_mean[3], c.bbox_std[0], c.bbox_std[1],
                     c.bbox_std[2], c.bbox_std[3], clamp, (float)W, (float)H, p->bbox_raw, p->bbox);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// d_image == nullptr: recompute_features = false (ImageDetect.lua:107-111) — reuse the trunk output of the last
// call on this handle (iterative localisation, Tester_FRCNN.lua:82-89) and run only the ROI head on new boxes.
static int run_detect(mpn_frcnn *p, const float *d_image, int H0, int W0, const float *d_boxes, int N, hipStream_t s, int clamp = 1) {
  const mpn_frcnn_config &c = p->cfg;
  MPN_CHECK_ARG(p && d_boxes);
  p->seg_shape[0][0] = -1;  // SEG_HEAD: the host-side state a captured head graph relies on is being rewritten (run_segment re-stamps it after its body)
  MPN_CHECK_ARG(H0 > 0 && W0 > 0 && N > 0 && N <= c.max_rois);
  // getImages (ImageDetect.lua:34-43): s = target/min side, capped so that round(s*max side) <= max_size; the image is
  // resampled to (long)(H*s) x (long)(W*s).  scale_target == 0 keeps the image as it is (s = 1).
  double sc = 1.0;
  int H = H0, W = W0;
  if (c.scale_target > 0.0) {
    sc = mpn_pick_scale(H0, W0, c.scale_target, c.scale_max > 0.0 ? c.scale_max : 1e30);
    if (sc != 1.0) { H = (int)((double)H0 * sc); W = (int)((double)W0 * sc); }
  }
  if (H <= 0 || W <= 0 || H > c.max_h || W > c.max_w) {
    set_error("run_detect: %dx%d image (scaled to %dx%d) exceeds the pipeline's %dx%d", H0, W0, H, W, c.max_h, c.max_w);
    return MPN_EINVAL;
  }
  Act feat;
  int rc = MPN_OK;
  if (d_image) {
    const float *img = d_image;
    if (sc != 1.0) {
      const size_t need = (size_t)3 * H * W * sizeof(float), need_t = (size_t)3 * H0 * W * sizeof(float);
      if (need > p->scaled_bytes || need_t > p->scale_tmp_bytes) {
        MPN_CHECK_HIP(hipStreamSynchronize(s));
        bump_alloc_generation();  // BEFORE the frees: a hipMalloc that fails below must not leave captured graphs holding a freed pointer
        if (need > p->scaled_bytes) { if (p->scaled) (void)hipFree(p->scaled); p->scaled = nullptr; p->scaled_bytes = 0; MPN_CHECK_HIP(hipMalloc(&p->scaled, need)); p->scaled_bytes = need; }
        if (need_t > p->scale_tmp_bytes) { if (p->scale_tmp) (void)hipFree(p->scale_tmp); p->scale_tmp = nullptr; p->scale_tmp_bytes = 0; MPN_CHECK_HIP(hipMalloc(&p->scale_tmp, need_t)); p->scale_tmp_bytes = need_t; }
      }
      rc = mpn_image_scale(d_image, 3, H0, W0, H, W, p->scale_tmp, p->scaled, s);
      if (rc) return rc;
      img = p->scaled;
    }
    if (p->rn) {
      ProfScope ps(p, MPN_PROF_CONV_DIRECT, s);
      rc = resnet_trunk_forward(p->rn, img, H, W, c.tf_swap, c.tf_scale, c.tf_mean, c.tf_std, c.tf_std[0] != 0.0, s);
    } else {
      rc = run_trunk(p, img, H, W, s, &feat);
    }
  } else if (p->rn) {
    if (!resnet_has_features(p->rn, H, W)) { set_error("run_detect: no cached features for a %dx%d image", H0, W0); return MPN_ESTATE; }
  } else {
    if (p->last_h != H || p->last_w != W || !p->tap_act[0].p) { set_error("run_detect: no cached features for a %dx%d image", H0, W0); return MPN_ESTATE; }
    feat = p->tap_act[0];
  }
  if (rc) return rc;
  const bool defer_heads = p->defer_stream && !p->rn && !p->is_mpnet;  // the pipelined forms of the plain VGG head (pipelined_impl)
  rc = defer_heads ? project_im_rois_copy(d_boxes, N, sc, p->rois, p->boxes_b[p->defer_set], s) : mpn_project_im_rois(d_boxes, N, sc, p->rois, s);
  if (rc) return rc;
  // decode uses the ORIGINAL boxes and clamps to the ORIGINAL image (ImageDetect.lua:183-185, Tester_FRCNN.lua:75-78)
  H = H0; W = W0;
  if (p->is_mpnet) {
    rc = run_mpnet_head(p, d_boxes, N, H, W, s, clamp);
    p->last_n = N;
    return rc;
  }
  const int C = c.n_classes, F = c.fc_dim;
  if (p->rn && !p->rn_region.empty()) {  // ResNet towers: Foveal region t -> ROI pool -> layer4 copy t -> average pool -> its slice of `cat`
    rc = mpn_foveal_forward(p->rois, N, p->fov, s);
    if (rc) return rc;
    const int Fcb = lin_np(F) / 8, Mp = lin_mp(N);
    // Two tower lanes (see mpn_frcnn::tower_stream): tower t on lane t & 1 — its own stream and activation buffers; the sorted / range-max
    // images of the feature map every tower pools from are built before the fork.  Tower t + 1's ROI pooling (a store-bound launch with no
    // matrix work), the ragged last block round and the launch gaps of each of a tower's ~20-80 convolutions then run under the other
    // lane's convolutions.  Pure scheduling: bit-identical results (the towers meet only in `cat`, each writing its own slice).
    // MEASURED (profiles/r06_tower_lanes_ab.txt): 10.86 -> 10.87 ms on configs[3] bf16, 23.61 -> 23.69 ms on configs[4] — nothing: two
    // towers' convolutions contend for the same vector-memory path that bounds each of them alone (the VGG towers' matrix-bound GEMMs are
    // a different story: 13.71 -> 13.43 ms, run_mpnet_head).  The graph lanes therefore exist in the debug flavour only (tower_lanes = 2).
    const bool lanes = p->tower_stream && resnet_has_second_lane(p->rn) && g_tower_lanes == 2 && p->rn_region.size() > 1 && !p->prof;
    if (lanes) {
      rc = resnet_heads_prepare(p->rn, s);
      if (rc) return rc;
      MPN_CHECK_HIP(hipEventRecord(p->ev_lane_go, s));
      MPN_CHECK_HIP(hipStreamWaitEvent(p->tower_stream, p->ev_lane_go, 0));
    }
    for (size_t t = 0; t < p->rn_region.size(); ++t) {
      const int ln = lanes ? (int)(t & 1) : 0;
      hipStream_t ls = ln ? p->tower_stream : s;
      SplitkSlotScope lane_slabs(ln ? SCR_GEMM_SPLITK_LANE : SCR_GEMM_SPLITK);  // (fp32 graphs: a pointwise layer of few tiles on the split-K GEMM)
      ProfScope ps(p, MPN_PROF_FC6, ls);
      rc = resnet_head_forward(p->rn, (int)t, p->fov + 5 * p->rn_region[t], 20, N, c.spatial_scale, p->cat + t * (size_t)Fcb * Mp * 8, Mp, ls, ln);
      if (rc) return rc;
    }
    if (lanes) {
      MPN_CHECK_HIP(hipEventRecord(p->ev_lane_done, p->tower_stream));
      MPN_CHECK_HIP(hipStreamWaitEvent(s, p->ev_lane_done, 0));
    }
    rc = run_integral_heads(p, d_boxes, N, H, W, (int)p->rn_region.size() - 1, s, clamp);
    p->last_n = N;
    return rc;
  }
  // the pipelined forms of the plain VGG head hand the heads over to the side stream after fc7 (mpn_frcnn::defer_stream)
  hipStream_t hs = defer_heads ? p->defer_stream : s;
  float *y7 = hs != s ? p->y7_b[p->defer_set] : p->y7;
  p->y7_last = y7;
  const float *dec_boxes = d_boxes;
  if (p->rn) {  // resnet.lua:40-48: ROIPooling(14,14) -> layer4 -> average pool -> View; lands in y7 as the heads' operand
    ProfScope ps(p, MPN_PROF_FC6, s);
    rc = resnet_head_forward(p->rn, 0, p->rois, 5, N, c.spatial_scale, p->y7, lin_mp(N), s);
    if (rc) return rc;
  } else {
  { ProfScope ps(p, MPN_PROF_ROIPOOL, s);
    if (p->feat_pm && g_roi_pool_pm) {
      if (!p->feat_pm_valid) { rc = c8p_to_pixel_major(feat, p->feat_pm, s); if (rc) return rc; p->feat_pm_valid = true; }  // once per trunk run
      rc = roi_pool_pm(feat, p->feat_pm, p->rois, N, c.pooled_h, c.pooled_w, c.spatial_scale, RoiRule{1.0f, 0, c.roi_bin_rule}, p->x6, s);
    } else {
      rc = roi_pool_c8(feat, p->rois, N, c.pooled_h, c.pooled_w, c.spatial_scale, RoiRule{1.0f, 0, c.roi_bin_rule}, p->x6, nullptr, s);
    } }
  if (rc) return rc;
  { ProfScope ps(p, MPN_PROF_FC6, s);
    if (p->w6_s3) {  // MPN_FC_SPLIT3: the pooled operand -> three bf16 planes, six bf16 products per k-step with fp32 accumulation, fixed K ranges
      rc = split3_planes(p->x6, p->K6, lin_mp(N), lin_mp(N), p->x6_s3, s);
      if (rc == MPN_OK) rc = linear_c8_split3(p->x6_s3, N, p->K6, p->w6_s3, p->b6, F, 1, p->y6, s);
    } else
      rc = linear_c8(p->x6, N, p->K6, p->w6, p->b6, F, 1, p->y6, nullptr, s, 0, nullptr, 1); }
  if (rc) return rc;
  { ProfScope ps(p, MPN_PROF_FC7, s);
    if (p->w7_s3) {
      rc = split3_planes(p->y6, F, lin_mp(N), lin_mp(N), p->y6_s3, s);
      if (rc == MPN_OK) rc = linear_c8_split3(p->y6_s3, N, F, p->w7_s3, p->b7, F, 1, y7, s);
    } else
      rc = linear_c8(p->y6, N, F, p->w7, p->b7, F, 1, y7, nullptr, s, 0, nullptr, 1); }
  if (rc) return rc;
  }
  if (hs != s) {  // hand over: the side stream continues from here (its work on this buffer set is ordered by ev_tail[set])
    MPN_CHECK_HIP(hipEventRecord(p->ev_fc7, s));
    MPN_CHECK_HIP(hipStreamWaitEvent(hs, p->ev_fc7, 0));
    dec_boxes = p->boxes_b[p->defer_set];
#ifdef MPN_DEBUG_HOOKS
    if (g_defer_heads == 2) hipLaunchKernelGGL(side_stream_delay_kernel, dim3(1), dim3(64), 0, hs, 100000ll);
#endif
  }
  { ProfScope ps(p, MPN_PROF_HEADS, hs);
    SplitkSlotScope sk(hs != s ? SCR_GEMM_SPLITK_SIDE : SCR_GEMM_SPLITK);  // its partial sums must not share a buffer with fc6 / fc7 of the next image
    rc = linear_c8(y7, N, F, p->wh, p->bh, 5 * C, 0, nullptr, p->head, hs, 0, nullptr, 1); }
  if (rc) return rc;
  ProfScope ps_post(p, MPN_PROF_POST, hs);
  hipLaunchKernelGGL(head_softmax_kernel, dim3(cdiv(N, 4)), dim3(256), 0, hs, p->head, 5 * C, N, C, p->scores);
  MPN_CHECK_LAUNCH();
  const bool norm = c.bbox_std[0] != 0.0f;
  hipLaunchKernelGGL(head_decode_kernel, dim3((unsigned)cdiv_sz((size_t)N * C, 256)), dim3(256), 0, hs, p->head, 5 * C, C, N, C, dec_boxes,
                     norm ? 1 : 0, c.bbox_mean[0], c.bbox_mean[1], c.bbox_mean[2], c.bbox_mean[3], c.bbox_std[0], c.bbox_std[1],
                     c.bbox_std[2], c.bbox_std[3], clamp, (float)W, (float)H, p->bbox_raw, p->bbox);
  MPN_CHECK_LAUNCH();
  p->last_n = N;
  return MPN_OK;
}

extern "C" int mpn_frcnn_detect(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N,
                                float *d_scores, float *d_bbox, int clamp, void *stream) {
  MPN_CHECK_ARG(p != nullptr);
  ScratchScope scratch_scope(&p->scratch);
  { int rcf = mpn_frcnn_flush(p, stream); if (rcf) return rcf; }  // a pipelined predecessor's side-stream work may still use the head buffers
  hipStream_t s = as_stream(stream);
  int rc = run_detect(p, d_image, H, W, d_boxes, N, s, clamp ? 1 : 0);
  if (rc) return rc;
  const int C = p->cfg.n_classes;
  if (d_scores) MPN_CHECK_HIP(hipMemcpyAsync(d_scores, p->scores, (size_t)N * C * sizeof(float), hipMemcpyDeviceToDevice, s));
  if (d_bbox) MPN_CHECK_HIP(hipMemcpyAsync(d_bbox, p->bbox, (size_t)N * 4 * C * sizeof(float), hipMemcpyDeviceToDevice, s));
  return MPN_OK;
}

static void select_set(mpn_frcnn *p, int b) {
  p->scored = p->scored_b[b]; p->keep = p->keep_b[b]; p->keep_idx = p->keep_idx_b[b];
  p->counts = p->counts_b[b]; p->n_keep = p->n_keep_b[b]; p->thresh = p->thresh_b[b];
  p->voted = p->voted_b[b];
}

// ---- captured launch graphs (see mpn_frcnn::GraphKey) --------------------------------------------------------------------------
enum { SEG_HEAD = 0, SEG_TAIL = 1, SEG_SHARD_NMS = 2, SEG_SHARD_FIN = 3 };
constexpr size_t kMaxGraphs = 64;  // per handle: a caller that passes fresh pointers on every call must not grow the cache without bound

// Runs body(stream) — a pure chain of stream work plus host-side bookkeeping — for segment `kind`, or replays its captured graph.
// stable_ptrs: the key's pointers are library-owned (staging sets): capture at the first sighting; caller-provided pointers are captured
// at their second sighting (a caller that hands in fresh buffers every call never pays for a capture).
template <class F>
static int run_segment(mpn_frcnn *p, int kind, const mpn_frcnn::GraphKey &key, const int (&shape)[4], bool stable_ptrs, hipStream_t s, F &&body) {
  int *last = p->seg_shape[kind];
  const bool same_shape = memcmp(last, shape, sizeof(shape)) == 0;
  auto direct = [&]() -> int {
    const int rc = body(s);
    memcpy(last, shape, sizeof(shape));
    if (rc) last[0] = -2;  // a failed run leaves no state to rely on
    return rc;
  };
  if (!p->graphs_on || p->prof) return direct();
  auto it = p->graphs.find(key);
  int seen_before = 0;
  if (it == p->graphs.end()) {
    if (!stable_ptrs) {  // first sighting of caller-provided pointers: remember the key in the side ring only
      int hit = -1;
      for (int q = 0; q < mpn_frcnn::kUnseen; ++q) {
        const mpn_frcnn::GraphKey &u = p->unseen[kind][q];
        if (p->unseen_valid[kind][q] && !(u < key) && !(key < u)) { hit = q; break; }
      }
      if (hit < 0) {
        const int q = p->unseen_next[kind];
        p->unseen[kind][q] = key; p->unseen_valid[kind][q] = true;
        p->unseen_next[kind] = (q + 1) % mpn_frcnn::kUnseen;
        return direct();
      }
      p->unseen_valid[kind][hit] = false;
      seen_before = 1;
    }
    if (p->graphs.size() >= kMaxGraphs) {  // full: drop the entries that hold no executable graph (failed / never captured) before giving up
      for (auto j = p->graphs.begin(); j != p->graphs.end();) j = j->second.exec ? std::next(j) : p->graphs.erase(j);
      if (p->graphs.size() >= kMaxGraphs) {  // all live: evict the least recently used one
        auto lru = p->graphs.begin();
        for (auto j = p->graphs.begin(); j != p->graphs.end(); ++j) if (j->second.last_use < lru->second.last_use) lru = j;
        // its last replay may still be in flight (the pipelined forms put no bound on how far the host runs ahead): wait for the stream it
        // was launched on before the executable goes away.  Evictions are rare — a host must rotate > 64 live keys to get here.
        if (lru->second.launched) (void)hipStreamSynchronize(lru->second.last_stream);
        (void)hipGraphExecDestroy(lru->second.exec);
        p->graphs.erase(lru);
      }
    }
    it = p->graphs.emplace(key, mpn_frcnn::GraphEntry()).first;
    it->second.seen = seen_before;
  }
  mpn_frcnn::GraphEntry &e = it->second;
  e.last_use = ++p->graph_clock;
  if (e.exec && e.gen != alloc_generation()) {  // a library buffer was replaced since (the regrow synchronised its stream; a replay on another stream may still run)
    if (e.launched && e.last_stream != s) (void)hipStreamSynchronize(e.last_stream);
    (void)hipGraphExecDestroy(e.exec); e.exec = nullptr;
  }
  if (e.exec && same_shape) {
    MPN_CHECK_HIP(hipGraphLaunch(e.exec, s));
    e.last_stream = s; e.launched = true;
    ++p->graph_replays;
    return MPN_OK;
  }
  ++e.seen;
  if (!same_shape || e.failed || (!stable_ptrs && e.seen < 2)) return direct();  // the warm-up run for this shape / this key
  if (!p->cap_stream && hipStreamCreateWithFlags(&p->cap_stream, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); e.failed = true; return direct(); }
  if (hipStreamBeginCapture(p->cap_stream, hipStreamCaptureModeRelaxed) != hipSuccess) { (void)hipGetLastError(); e.failed = true; return direct(); }
  const int rc_body = body(p->cap_stream);  // host bookkeeping happens, the stream work is recorded instead of executed
  hipGraph_t g = nullptr;
  const hipError_t ec = hipStreamEndCapture(p->cap_stream, &g);
  hipGraphExec_t ex = nullptr;
  if (rc_body == MPN_OK && ec == hipSuccess && g && hipGraphInstantiate(&ex, g, nullptr, nullptr, 0) == hipSuccess && ex) {
    (void)hipGraphDestroy(g);
    e.exec = ex; e.gen = alloc_generation();
    ++p->graph_captures;
    memcpy(last, shape, sizeof(shape));
    MPN_CHECK_HIP(hipGraphLaunch(e.exec, s));
    e.last_stream = s; e.launched = true;
    return MPN_OK;
  }
  if (g) (void)hipGraphDestroy(g);
  (void)hipGetLastError();
  e.failed = true;     // this chain cannot be captured here: never try again, run it as ordinary launches
  return direct();
}

// Tester_FRCNN.lua:106-125 + keep_top_k: per class j=1..C-1 select (score > thresh) -> NMS -> top-k, on stream `t`
static int run_tail(mpn_frcnn *p, int N, float *d_dets, int top_cap, int *d_n_dets, hipStream_t sel_stream, hipStream_t t,
                    hipEvent_t after_select) {
  const mpn_frcnn_config &c = p->cfg;
  const int C = c.n_classes;
  int rc;
  const float *sc = p->scores, *bb = p->bbox;
  if (c.num_iter > 1) { sc = p->it_scores; bb = p->it_bbox; }  // rows of both localisation passes (utils.joinTable, Tester_FRCNN.lua:99-100)
  { ProfScope ps(p, MPN_PROF_SELECT, sel_stream);
    rc = mpn_select_scored(sc, bb, N, C, 1, c.score_thresh, p->scored, p->counts, nullptr, sel_stream); }
  if (rc) return rc;
  if (after_select) {  // hand over to the side stream
    MPN_CHECK_HIP(hipEventRecord(after_select, sel_stream));
    MPN_CHECK_HIP(hipStreamWaitEvent(t, after_select, 0));
  }
  const mpn_frcnn::GraphKey key{SEG_TAIL, d_dets, d_n_dets, p->scored, nullptr, N, top_cap, 0, 0};  // p->scored names the buffer set
  const int shape[4] = {N, top_cap, c.bbox_voting ? 1 : 0, 0};
  return run_segment(p, SEG_TAIL, key, shape, false, t, [&](hipStream_t q) -> int {
    int r;
    { ProfScope ps(p, MPN_PROF_NMS, q);
      r = after_select ? nms_batched_under_trunk(p->scored, p->counts, C - 1, N, c.nms_thresh, p->keep, p->keep_idx, p->n_keep, q)
                       : mpn_nms_batched(p->scored, p->counts, C - 1, N, c.nms_thresh, p->keep, p->keep_idx, p->n_keep, q); }
    if (r) return r;
    const float *final_tables = p->keep;
    if (c.bbox_voting) {  // Tester_FRCNN.lua:118-124
      r = mpn_bbox_vote_batched(p->keep, p->n_keep, p->scored, p->counts, C - 1, N, c.bbox_vote_thresh,
                                c.bbox_vote_score_pow != 0.0f ? c.bbox_vote_score_pow : 1.0f, p->voted, q);
      if (r) return r;
      final_tables = p->voted;
    }
    ProfScope ps(p, MPN_PROF_TOPK, q);
    return mpn_keep_top_k_sorted(final_tables, p->n_keep, C - 1, N, c.top_k, p->thresh, d_dets, top_cap, d_n_dets, q);  // NMS / voted tables: scores non-increasing per class
  });
}

// Tester_FRCNN.lua:72-100: detect (clamped, :75-78); for i = 2..num_iter: SelectBoxes on the previous pass -> detect on the refined
// boxes with recompute_features = false (NOT clamped: only the first bbox_pred is); the rows of all passes are concatenated
// before the per-class NMS.  opt.test_use_rbox_scores (:91-97): the scores of pass i+1 go with the boxes of pass i.
static int run_detect_iter(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, hipStream_t s, int *n_rows) {
  int rc = run_detect(p, d_image, H, W, d_boxes, N, s, 1);
  *n_rows = N;
  const mpn_frcnn_config &c = p->cfg;
  if (rc || c.num_iter <= 1) return rc;
  const int C = c.n_classes;
  const size_t srow = (size_t)N * C, brow = (size_t)N * 4 * C;
  const bool rbox = c.use_rbox_scores != 0;
  for (int it = 1;; ++it) {  // pass `it` (1-based) has just run: p->scores / p->bbox hold its output
    const int sslot = rbox ? it - 2 : it - 1, bslot = it - 1;  // rbox: drop the first score table and the last box table
    if (sslot >= 0) MPN_CHECK_HIP(hipMemcpyAsync(p->it_scores + sslot * srow, p->scores, srow * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (!(rbox && it == c.num_iter)) MPN_CHECK_HIP(hipMemcpyAsync(p->it_bbox + bslot * brow, p->bbox, brow * sizeof(float), hipMemcpyDeviceToDevice, s));
    if (it == c.num_iter) break;
    rc = mpn_select_boxes_forward(p->scores, p->bbox, N, C, p->it_boxes, s);
    if (rc) return rc;
    rc = run_detect(p, nullptr, H, W, p->it_boxes, N, s, 0);
    if (rc) return rc;
  }
  *n_rows = (rbox ? c.num_iter - 1 : c.num_iter) * N;
  return MPN_OK;
}

// the head segment: run_detect_iter as a captured graph when it starts from an image (the cached-features form keeps host-side
// checks of what is cached and always runs as ordinary launches)
static int run_head(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, hipStream_t s, int *n_rows, bool stable_ptrs) {
  if (!d_image || !d_boxes || N <= 0 || N > p->cfg.max_rois) {
    const int rc = run_detect_iter(p, d_image, H, W, d_boxes, N, s, n_rows);
    p->seg_shape[SEG_HEAD][0] = -1;
    return rc;
  }
  const mpn_frcnn_config &c = p->cfg;
  *n_rows = c.num_iter > 1 ? (c.use_rbox_scores ? c.num_iter - 1 : c.num_iter) * N : N;
  const mpn_frcnn::GraphKey key{SEG_HEAD, d_image, d_boxes, nullptr, nullptr, H, W, N, 0};
  const int shape[4] = {H, W, N, 1};
  const int rc = run_segment(p, SEG_HEAD, key, shape, stable_ptrs, s, [&](hipStream_t q) -> int {
    int rows = 0;
    return run_detect_iter(p, d_image, H, W, d_boxes, N, q, &rows);
  });
  if (rc == MPN_OK) p->last_n = N;  // (a replay runs no host code)
  return rc;
}

static int join_tail(mpn_frcnn *p, int b, hipStream_t s) {
  if (p->tail_pending[b]) {
    MPN_CHECK_HIP(hipStreamWaitEvent(s, p->ev_tail[b], 0));
    p->tail_pending[b] = false;
  }
  return MPN_OK;
}

extern "C" int mpn_frcnn_flush(mpn_frcnn *p, void *stream) {
  MPN_CHECK_ARG(p != nullptr);
  hipStream_t s = as_stream(stream);
  int rc = join_tail(p, 0, s);
  if (rc) return rc;
  return join_tail(p, 1, s);
}

extern "C" int mpn_frcnn_test_one(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N,
                                  float *d_dets, int top_cap, int *d_n_dets, void *stream) {
  MPN_CHECK_ARG(p != nullptr && d_n_dets && (top_cap == 0 || d_dets) && top_cap >= 0);
  ScratchScope scratch_scope(&p->scratch);
  hipStream_t s = as_stream(stream);
  int rc = mpn_frcnn_flush(p, stream);  // a pipelined predecessor may still own a buffer set
  if (rc) return rc;
  int rows = N;
  rc = run_head(p, d_image, H, W, d_boxes, N, s, &rows, false);
  if (rc) return rc;
  select_set(p, 0);
  p->last_rows = rows;
  return run_tail(p, rows, d_dets, top_cap, d_n_dets, s, s, nullptr);
}

// ------------------------------------------------------------------------------------------------------------------------------
// Proposal (ROI) sharding of ONE image across the GPUs of a node — the latency mode of SURVEY §8e / north_star's
// "images+proposals shard across the 8 GPUs".  Replaces ModelParallelTable.lua:195-242 (broadcast the whole input to every tower
// GPU, run the towers, copy their outputs back and concatenate): here every rank runs the trunk on the image (1.6 ms of replicated
// work instead of a 5-62 MB feature broadcast over xGMI), the ROI head on ITS rows of the proposal table, the per-class NMS on
// ITS classes; what travels is scored boxes only — one all-gather of the decoded rows ([N/G, 5C] per rank: 420 KB in total for
// VOC, 1.6 MB for COCO) and one of the kept tables.  Rows are independent (ImageDetect.lua:126-133's chunk invariance) and
// classes are independent (Tester_FRCNN.lua:106-125's loop), so the result is the unsharded mpn_frcnn_test_one's bit for bit.
// The three steps take caller-provided records so that the exchange between them can be any transport; mpn_frcnn_test_one_sharded
// chains them over an mpn_comm (RCCL all-gather, comm.hip).
static int shard_passes(const mpn_frcnn_config &c) { return c.num_iter > 1 ? (c.use_rbox_scores ? c.num_iter - 1 : c.num_iter) : 1; }

extern "C" int mpn_shard_range(int n, int world, int rank, int *lo, int *hi) {
  MPN_CHECK_ARG(n >= 0 && world >= 1 && rank >= 0 && rank < world && lo && hi);
  shard_bounds(n, world, rank, lo, hi);
  return MPN_OK;
}

extern "C" size_t mpn_frcnn_shard_rows_floats(const mpn_frcnn *p, int N, int world) {
  if (!p || N <= 0 || world < 1) return 0;
  const int chunk = (N + world - 1) / world;
  return (size_t)shard_passes(p->cfg) * chunk * 5 * p->cfg.n_classes;
}

extern "C" size_t mpn_frcnn_shard_class_floats(const mpn_frcnn *p, int N, int world) {
  if (!p || N <= 0 || world < 1) return 0;
  const int n_cls = p->cfg.n_classes - 1, cmax = (n_cls + world - 1) / world;
  return shard_class_rec_floats(cmax, shard_passes(p->cfg) * N, p->cfg.bbox_voting);
}

extern "C" int mpn_frcnn_shard_head(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, int rank, int world,
                                    float *d_rows_rec, void *stream) {
  MPN_CHECK_ARG(p != nullptr && d_boxes && d_rows_rec && N > 0 && N <= p->cfg.max_rois && world >= 1 && rank >= 0 && rank < world);
  ScratchScope scratch_scope(&p->scratch);
  hipStream_t s = as_stream(stream);
  int rc = mpn_frcnn_flush(p, stream);
  if (rc) return rc;
  const mpn_frcnn_config &c = p->cfg;
  const int C = c.n_classes, P = shard_passes(c), chunk = (N + world - 1) / world;
  int lo, hi;
  shard_bounds(N, world, rank, &lo, &hi);
  const int n_local = hi - lo;
  const size_t total = (size_t)P * chunk * 5 * C;
  if (n_local == 0) {  // more ranks than proposals: nothing to score here
    MPN_CHECK_HIP(hipMemsetAsync(d_rows_rec, 0, total * sizeof(float), s));
    return MPN_OK;
  }
  int rows = n_local;
  rc = run_head(p, d_image, H, W, d_boxes + 4 * (size_t)lo, n_local, s, &rows, false);
  if (rc) return rc;
  const float *sc = c.num_iter > 1 ? p->it_scores : p->scores, *bb = c.num_iter > 1 ? p->it_bbox : p->bbox;
  hipLaunchKernelGGL(shard_pack_rows_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, s, sc, bb, n_local, P, C, chunk, d_rows_rec);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

extern "C" int mpn_frcnn_shard_nms(mpn_frcnn *p, const float *d_rows_all, int N, int rank, int world, float *d_class_rec, void *stream) {
  MPN_CHECK_ARG(p != nullptr && d_rows_all && d_class_rec && N > 0 && N <= p->cfg.max_rois && world >= 1 && rank >= 0 && rank < world);
  ScratchScope scratch_scope(&p->scratch);
  hipStream_t s = as_stream(stream);
  int rc = mpn_frcnn_flush(p, stream);
  if (rc) return rc;
  const mpn_frcnn_config &c = p->cfg;
  const int C = c.n_classes, P = shard_passes(c), chunk = (N + world - 1) / world, rows = P * N;
  // the whole image's joined tables, in the unsharded row order, where run_tail reads them
  float *sc = c.num_iter > 1 ? p->it_scores : p->scores, *bb = c.num_iter > 1 ? p->it_bbox : p->bbox;
  const size_t rec_floats = (size_t)P * chunk * 5 * C, total = (size_t)rows * 5 * C;
  select_set(p, 0);
  p->last_rows = rows;
  p->last_n = N;
  const int n_cls = C - 1, cmax = (n_cls + world - 1) / world;
  int c0, c1;
  shard_bounds(n_cls, world, rank, &c0, &c1);
  const mpn_frcnn::GraphKey key{SEG_SHARD_NMS, d_rows_all, d_class_rec, nullptr, nullptr, N, rank, world, 0};
  const int shape[4] = {N, rank, world, c.bbox_voting ? 1 : 0};
  return run_segment(p, SEG_SHARD_NMS, key, shape, false, s, [&](hipStream_t q) -> int {
    hipLaunchKernelGGL(shard_unpack_rows_kernel, dim3((unsigned)cdiv_sz(total, 256)), dim3(256), 0, q, d_rows_all, N, world, P, C, chunk, rec_floats, sc, bb);
    MPN_CHECK_LAUNCH();
    int r;
    { ProfScope ps(p, MPN_PROF_SELECT, q);
      r = mpn_select_scored(sc, bb, rows, C, 1, c.score_thresh, p->scored, p->counts, nullptr, q); }
    if (r) return r;
    if (c1 > c0) {
      const size_t off = (size_t)c0 * rows;
      { ProfScope ps(p, MPN_PROF_NMS, q);
        r = mpn_nms_batched(p->scored + off * 5, p->counts + c0, c1 - c0, rows, c.nms_thresh, p->keep + off * 5, p->keep_idx + off, p->n_keep + c0, q); }
      if (r) return r;
      if (c.bbox_voting) {
        r = mpn_bbox_vote_batched(p->keep + off * 5, p->n_keep + c0, p->scored + off * 5, p->counts + c0, c1 - c0, rows, c.bbox_vote_thresh,
                                  c.bbox_vote_score_pow != 0.0f ? c.bbox_vote_score_pow : 1.0f, p->voted + off * 5, q);
        if (r) return r;
      }
    }
    hipLaunchKernelGGL(shard_pack_classes_kernel, dim3(cdiv(rows, 256), cmax), dim3(256), 0, q, p->keep, p->keep_idx, p->n_keep,
                       c.bbox_voting ? p->voted : nullptr, c0, c1, rows, cmax, d_class_rec);
    MPN_CHECK_LAUNCH();
    return MPN_OK;
  });
}

extern "C" int mpn_frcnn_shard_finish(mpn_frcnn *p, const float *d_class_all, int N, int world, float *d_dets, int top_cap, int *d_n_dets,
                                      void *stream) {
  MPN_CHECK_ARG(p != nullptr && d_class_all && N > 0 && N <= p->cfg.max_rois && world >= 1);
  MPN_CHECK_ARG(d_n_dets && (top_cap == 0 || d_dets) && top_cap >= 0);
  ScratchScope scratch_scope(&p->scratch);
  hipStream_t s = as_stream(stream);
  const mpn_frcnn_config &c = p->cfg;
  const int n_cls = c.n_classes - 1, cmax = (n_cls + world - 1) / world, rows = shard_passes(c) * N;
  select_set(p, 0);
  p->last_rows = rows;
  const mpn_frcnn::GraphKey key{SEG_SHARD_FIN, d_class_all, d_dets, d_n_dets, nullptr, N, world, top_cap, 0};
  const int shape[4] = {N, world, top_cap, c.bbox_voting ? 1 : 0};
  return run_segment(p, SEG_SHARD_FIN, key, shape, false, s, [&](hipStream_t q) -> int {
    hipLaunchKernelGGL(shard_unpack_classes_kernel, dim3(cdiv(rows, 256), n_cls), dim3(256), 0, q, d_class_all, n_cls, world, rows, cmax,
                       shard_class_rec_floats(cmax, rows, c.bbox_voting), p->keep, p->keep_idx, p->n_keep, c.bbox_voting ? p->voted : nullptr);
    MPN_CHECK_LAUNCH();
    ProfScope ps(p, MPN_PROF_TOPK, q);
    return mpn_keep_top_k_sorted(c.bbox_voting ? p->voted : p->keep, p->n_keep, n_cls, rows, c.top_k, p->thresh, d_dets, top_cap, d_n_dets, q);
  });
}

static int shard_buf(mpn_frcnn *p, int i, size_t floats, hipStream_t s) {
  const size_t need = floats * sizeof(float);
  if (need <= p->sh_bytes[i]) return MPN_OK;
  MPN_CHECK_HIP(hipStreamSynchronize(s));
  bump_alloc_generation();  // before the free: see run_detect's regrow
  if (p->sh_buf[i]) (void)hipFree(p->sh_buf[i]);
  p->sh_buf[i] = nullptr; p->sh_bytes[i] = 0;
  MPN_CHECK_HIP(hipMalloc(&p->sh_buf[i], need));
  p->sh_bytes[i] = need;
  return MPN_OK;
}

extern "C" int mpn_frcnn_test_one_sharded(mpn_frcnn *p, mpn_comm *comm, const float *d_image, int H, int W, const float *d_boxes, int N,
                                          float *d_dets, int top_cap, int *d_n_dets, void *stream) {
  MPN_CHECK_ARG(p != nullptr && comm != nullptr && N > 0);
  hipStream_t s = as_stream(stream);
  const int world = mpn_comm_world(comm), rank = mpn_comm_rank(comm);
  MPN_CHECK_ARG(world >= 1 && rank >= 0);
  const size_t rr = mpn_frcnn_shard_rows_floats(p, N, world), cr = mpn_frcnn_shard_class_floats(p, N, world);
  int rc;
  if ((rc = shard_buf(p, 0, rr, s)) || (rc = shard_buf(p, 1, rr * world, s)) || (rc = shard_buf(p, 2, cr, s)) || (rc = shard_buf(p, 3, cr * world, s))) return rc;
  rc = mpn_frcnn_shard_head(p, d_image, H, W, d_boxes, N, rank, world, p->sh_buf[0], stream);
  if (rc) return rc;
  rc = mpn_gather_rows(comm, p->sh_buf[0], rr, p->sh_buf[1], stream);
  if (rc) return rc;
  rc = mpn_frcnn_shard_nms(p, p->sh_buf[1], N, rank, world, p->sh_buf[2], stream);
  if (rc) return rc;
  rc = mpn_gather_rows(comm, p->sh_buf[2], cr, p->sh_buf[3], stream);
  if (rc) return rc;
  return mpn_frcnn_shard_finish(p, p->sh_buf[3], N, world, d_dets, top_cap, d_n_dets, stream);
}

// Throughput form for a loop over images (Tester:test, Tester_FRCNN.lua:150-157): trunk + heads + select of
// image i on `stream`; NMS + top-k of image i on the pipeline's side stream, overlapping image
// i+1's MFMA kernels (they are latency-bound on ~20 CUs).  d_dets / d_n_dets of call i are ordered on `stream`
// only after call i+1 returns or after mpn_frcnn_flush(); the caller alternates two output buffers.
static int pipelined_impl(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, float *d_dets, int top_cap, int *d_n_dets,
                          void *stream, bool stable_ptrs) {
  MPN_CHECK_ARG(p != nullptr && d_n_dets && (top_cap == 0 || d_dets) && top_cap >= 0);
  ScratchScope scratch_scope(&p->scratch);
  hipStream_t s = as_stream(stream);
  const int b = (int)(p->seq & 1);
  int rc = join_tail(p, b, s);  // buffer set b was last used two calls ago
  if (rc) return rc;
  int rows = N;
  // heads + softmax + decode + select on the side stream too (plain VGG head, one localisation pass, no captured graphs, not profiling)
  const bool defer = g_defer_heads && !p->is_mpnet && !p->rn && p->cfg.num_iter == 1 && !p->graphs_on && !p->prof && d_image && d_boxes;
  if (defer) {
    if (!p->y7_b[1]) {
      p->y7_b[0] = p->y7;
      int rca = dev_alloc(p, &p->y7_b[1], (size_t)(lin_np(p->cfg.fc_dim) / 8) * p->Mp * 8 * sizeof(float), true);
      for (int i = 0; i < 2 && !rca; ++i) rca = dev_alloc(p, &p->boxes_b[i], (size_t)p->cfg.max_rois * 4 * sizeof(float), false);
      if (rca) return rca;
      MPN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_fc7, hipEventDisableTiming));
    }
    p->defer_stream = p->side; p->defer_set = b;
  } else if (p->was_deferred) {  // the form changed under us (graphs / profiling switched on): the side stream may still read y7_b[0] == y7
    rc = mpn_frcnn_flush(p, stream);
    if (rc) return rc;
  }
  p->was_deferred = defer;
  rc = run_head(p, d_image, H, W, d_boxes, N, s, &rows, stable_ptrs);
  p->defer_stream = nullptr;
  if (rc) return rc;
  select_set(p, b);
  p->last_rows = rows;
  rc = run_tail(p, rows, d_dets, top_cap, d_n_dets, defer ? p->side : s, p->side, p->ev_head[b]);
  if (rc) return rc;
  MPN_CHECK_HIP(hipEventRecord(p->ev_tail[b], p->side));
  p->tail_pending[b] = true;
  rc = join_tail(p, b ^ 1, s);  // the previous image's detections become visible to `stream` here
  p->seq++;
  return rc;
}

extern "C" int mpn_frcnn_test_one_pipelined(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N,
                                            float *d_dets, int top_cap, int *d_n_dets, void *stream) {
  return pipelined_impl(p, d_image, H, W, d_boxes, N, d_dets, top_cap, d_n_dets, stream, false);
}

// Host-fed throughput form: the reference's loop hands testOne a CPU image and CPU boxes (Tester_FRCNN.lua:64-66) and
// ImageDetect copies them to the GPU (ImageDetect.lua:148-151).  Here the upload of image i is issued on the handle's copy
// stream into staging set i & 1 — it runs while image i-1's kernels are still executing — and `stream` waits for it only
// at the point where the trunk starts.  The copy stream in turn waits until image i-2 (the previous user of the set) has
// been consumed.
extern "C" int mpn_frcnn_test_one_pipelined_host(mpn_frcnn *p, const float *h_image, int H, int W, const float *h_boxes, int N,
                                                 float *d_dets, int top_cap, int *d_n_dets, void *stream) {
  MPN_CHECK_ARG(p != nullptr && h_image && h_boxes && H > 0 && W > 0 && N > 0 && N <= p->cfg.max_rois);
  hipStream_t s = as_stream(stream);
  const size_t img_n = (size_t)3 * H * W;
  if (!p->copy) {  // first use: copy stream, events, the two box staging buffers
    MPN_CHECK_HIP(hipStreamCreateWithFlags(&p->copy, hipStreamNonBlocking));
    for (int i = 0; i < mpn_frcnn::kStage; ++i) {
      MPN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_up[i], hipEventDisableTiming));
      MPN_CHECK_HIP(hipEventCreateWithFlags(&p->ev_consumed[i], hipEventDisableTiming));
      int rc0 = dev_alloc(p, &p->stage_boxes[i], (size_t)p->cfg.max_rois * 4 * sizeof(float), false);
      if (rc0) return rc0;
    }
  }
  const int b = (int)(p->up_seq % mpn_frcnn::kStage);
  if (img_n > p->stage_cap[b]) {  // image staging grows on demand (getImages may be handed images larger than max_h x max_w)
    MPN_CHECK_HIP(hipStreamSynchronize(p->copy));
    MPN_CHECK_HIP(hipStreamSynchronize(s));
    bump_alloc_generation();  // before the free: see run_detect's regrow
    if (p->stage_img[b]) (void)hipFree(p->stage_img[b]);
    p->stage_img[b] = nullptr; p->stage_cap[b] = 0;
    size_t cap = (size_t)3 * p->cfg.max_h * p->cfg.max_w;
    if (cap < img_n) cap = img_n;
    MPN_CHECK_HIP(hipMalloc(&p->stage_img[b], cap * sizeof(float)));
    p->stage_cap[b] = cap;
  }
  // The staging set is free once the image that used it (three calls ago) has been consumed.  Waited for on the HOST, not with
  // hipStreamWaitEvent on the copy stream: a copy that depends on a compute-queue event leaves the SDMA path (measured on AlexNet,
  // tools/host_enqueue_probe.py: 0.99 ms / image with the stream wait, 0.86 resident), and this bounds the host's run-ahead to
  // three images, as a queue should.
  if (p->used_pending[b]) { MPN_CHECK_HIP(hipEventSynchronize(p->ev_consumed[b])); p->used_pending[b] = false; }
  MPN_CHECK_HIP(hipMemcpyAsync(p->stage_img[b], h_image, img_n * sizeof(float), hipMemcpyHostToDevice, p->copy));
  // (Round 5 tried the proposal table's copy on the launch stream instead — small copies are shader blits, i.e. a kernel on the copy stream's
  // hardware queue — and took it back: a 32-KB table there stalls the launch stream for 0.25 ms per image, 5.40 -> 5.65 ms at 2000 proposals.)
  MPN_CHECK_HIP(hipMemcpyAsync(p->stage_boxes[b], h_boxes, (size_t)N * 4 * sizeof(float), hipMemcpyHostToDevice, p->copy));
  MPN_CHECK_HIP(hipEventRecord(p->ev_up[b], p->copy));
  MPN_CHECK_HIP(hipStreamWaitEvent(s, p->ev_up[b], 0));
  int rc = pipelined_impl(p, p->stage_img[b], H, W, p->stage_boxes[b], N, d_dets, top_cap, d_n_dets, stream, true);  // staging sets: stable pointers
  if (rc) return rc;
  // the image is consumed by the trunk's first kernel and the boxes by the decode kernel: both are behind this point of `stream`
  MPN_CHECK_HIP(hipEventRecord(p->ev_consumed[b], s));
  p->used_pending[b] = true;
  p->up_seq++;
  return MPN_OK;
}

extern "C" int mpn_frcnn_create(const mpn_frcnn_config *cfg, const float *const *d_conv_w, const float *const *d_conv_b,
                                const float *d_fc6_w, const float *d_fc6_b, const float *d_fc7_w, const float *d_fc7_b,
                                const float *d_cls_w, const float *d_cls_b, const float *d_bbox_w, const float *d_bbox_b,
                                mpn_frcnn **out) {
  return create_impl(cfg, d_conv_w, d_conv_b, d_fc6_w, d_fc6_b, d_fc7_w, d_fc7_b, d_cls_w, d_cls_b, d_bbox_w, d_bbox_b, nullptr, out);
}

extern "C" int mpn_mpnet_create(const mpn_frcnn_config *cfg, const float *const *d_conv_w, const float *const *d_conv_b,
                                const mpn_mpnet_weights *mw, const float *d_cls_w, const float *d_cls_b,
                                const float *d_bbox_w, const float *d_bbox_b, mpn_frcnn **out) {
  MPN_CHECK_ARG(mw != nullptr);
  return create_impl(cfg, d_conv_w, d_conv_b, nullptr, nullptr, nullptr, nullptr, d_cls_w, d_cls_b, d_bbox_w, d_bbox_b, mw, out);
}

extern "C" int mpn_resnet_create(const mpn_frcnn_config *cfg, const mpn_resnet_weights *rw, const float *d_cls_w, const float *d_cls_b,
                                 const float *d_bbox_w, const float *d_bbox_b, mpn_frcnn **out) {
  MPN_CHECK_ARG(rw != nullptr);
  return create_impl(cfg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_cls_w, d_cls_b, d_bbox_w, d_bbox_b, nullptr, out, rw);
}

extern "C" int mpn_graph_create(const mpn_frcnn_config *cfg, const mpn_graph_weights *gw, const float *d_cls_w, const float *d_cls_b,
                                const float *d_bbox_w, const float *d_bbox_b, mpn_frcnn **out) {
  MPN_CHECK_ARG(gw != nullptr);
  return create_impl(cfg, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, d_cls_w, d_cls_b, d_bbox_w, d_bbox_b, nullptr, out, nullptr, gw);
}

extern "C" int mpn_frcnn_set_graphs(mpn_frcnn *p, int enable) {
  MPN_CHECK_ARG(p != nullptr);
  p->graphs_on = enable ? 1 : 0;
  return MPN_OK;
}

extern "C" int mpn_frcnn_graph_stats(const mpn_frcnn *p, long *captures, long *replays) {
  MPN_CHECK_ARG(p != nullptr);
  if (captures) *captures = p->graph_captures;
  if (replays) *replays = p->graph_replays;
  return MPN_OK;
}

extern "C" int mpn_frcnn_set_profiling(mpn_frcnn *p, int enable) {
  MPN_CHECK_ARG(p != nullptr);
  p->prof = enable != 0;
  return MPN_OK;
}

extern "C" int mpn_frcnn_get_profile(mpn_frcnn *p, double *ms, long *counts, int n_tags, int reset) {
  MPN_CHECK_ARG(p != nullptr && ms && counts && n_tags >= MPN_PROF_NTAGS);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  for (size_t i = 0; i + 1 < p->ev_used + 1 && i / 2 < p->ev_tag.size(); i += 2) {
    float t = 0.f;
    if (hipEventElapsedTime(&t, p->ev_pool[i], p->ev_pool[i + 1]) == hipSuccess) {
      p->prof_ms[p->ev_tag[i / 2]] += t;
      p->prof_cnt[p->ev_tag[i / 2]] += 1;
    }
  }
  p->ev_used = 0;
  p->ev_tag.clear();
  for (int t = 0; t < MPN_PROF_NTAGS; ++t) { ms[t] = p->prof_ms[t]; counts[t] = p->prof_cnt[t]; }
  if (reset) for (int t = 0; t < MPN_PROF_NTAGS; ++t) { p->prof_ms[t] = 0; p->prof_cnt[t] = 0; }
  return MPN_OK;
}

extern "C" int mpn_frcnn_nms_results(mpn_frcnn *p, const float **d_keep, const int **d_keep_idx, const int **d_n_keep,
                                     int *m_stride) {
  MPN_CHECK_ARG(p != nullptr);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  if (d_keep) *d_keep = (p->cfg.bbox_voting && p->voted) ? p->voted : p->keep;  // what testOne returns as img_boxes[j]
  if (d_keep_idx) *d_keep_idx = p->keep_idx;
  if (d_n_keep) *d_n_keep = p->n_keep;
  if (m_stride) *m_stride = p->last_rows ? p->last_rows : p->last_n;
  return MPN_OK;
}

#ifdef MPN_DEBUG_HOOKS
// bench.py's `power_sensitivity` leg (debug flavour only): fc6 of the VGG Fast R-CNN pipeline issued `iters` times BACK TO BACK on the
// operand the last detect() left in HBM (the ROI-pooled, post-ReLU conv5 features and the handle's own fc6 weights) — the same GEMM that
// mpn_debug_bench_linear times on dense random operands.  Inside the pipeline fc6 follows the trunk's phases and runs at a higher clock.
extern "C" int mpn_debug_bench_fc6(mpn_frcnn *p, int iters, float *ms_out) {
  MPN_CHECK_ARG(p && iters > 0 && ms_out);
  if (p->rn || p->is_mpnet || p->last_n <= 0) { set_error("mpn_debug_bench_fc6: needs a VGG Fast R-CNN handle after a detect()"); return MPN_ESTATE; }
  ScratchScope scratch_scope(&p->scratch);
  const int N = p->last_n, F = p->cfg.fc_dim;
  hipEvent_t e0, e1;
  MPN_CHECK_HIP(hipEventCreate(&e0)); MPN_CHECK_HIP(hipEventCreate(&e1));
  int rc = MPN_OK;
  for (int i = 0; i < 2 && rc == MPN_OK; ++i) rc = linear_c8(p->x6, N, p->K6, p->w6, p->b6, F, 1, p->y6, nullptr, nullptr, 0, nullptr, 1);
  MPN_CHECK_HIP(hipDeviceSynchronize());
  MPN_CHECK_HIP(hipEventRecord(e0, nullptr));
  for (int i = 0; i < iters && rc == MPN_OK; ++i) rc = linear_c8(p->x6, N, p->K6, p->w6, p->b6, F, 1, p->y6, nullptr, nullptr, 0, nullptr, 1);
  MPN_CHECK_HIP(hipEventRecord(e1, nullptr));
  MPN_CHECK_HIP(hipEventSynchronize(e1));
  float ms = 0.f;
  MPN_CHECK_HIP(hipEventElapsedTime(&ms, e0, e1));
  *ms_out = ms / iters;
  (void)hipEventDestroy(e0); (void)hipEventDestroy(e1);
  return rc;
}
#endif

extern "C" int mpn_frcnn_debug_tensor(mpn_frcnn *p, const char *name, const float **d_ptr, size_t *n_elems) {
  MPN_CHECK_ARG(p && name && d_ptr && n_elems);
  if (p->last_n <= 0 || (!p->rn && p->last_h <= 0)) { set_error("mpn_frcnn_debug_tensor: run detect first"); return MPN_ESTATE; }
  const mpn_frcnn_config &c = p->cfg;
  const int N = p->last_n, C = c.n_classes, F = c.fc_dim, PP = c.pooled_h * c.pooled_w;
  int h = p->last_h, w = p->last_w;
  for (auto &L : p->conv) if (L.pool) { h = (h + 1) / 2; w = (w + 1) / 2; }
  std::string nm(name);
  size_t n = 0;
  if (nm == "conv5") n = (size_t)p->feat_c * h * w;
  else if (nm == "pooled") n = (size_t)N * p->feat_c * PP;
  else if (nm == "fc7") n = (size_t)N * F;
  else if (nm == "cls") n = (size_t)N * C;
  else if (nm == "bbox_raw") n = (size_t)N * 4 * C;
  else if (nm == "cls_k") n = (size_t)N * p->n_integral * C;                       // integral heads: the K classifiers' logits [N, K * C] (pre-softmax)
  else if (nm == "cat") n = (size_t)N * (p->is_mpnet ? p->towers.size() : p->rn_region.size()) * F;  // tower outputs side by side [N, towers * F]
  else { set_error("mpn_frcnn_debug_tensor: unknown tensor '%s'", name); return MPN_EINVAL; }
  const bool towers = p->is_mpnet || (p->rn && !p->rn_region.empty());
  if ((nm == "cls_k" || nm == "cat") && (!towers || lin_np(F) != F)) { set_error("mpn_frcnn_debug_tensor: '%s' needs a tower model (integral heads)", name); return MPN_EINVAL; }
  if (p->is_mpnet && nm != "conv5" && nm != "bbox_raw" && nm != "cls_k" && nm != "cat") { set_error("mpn_frcnn_debug_tensor: '%s' is not kept by the MultiPathNet head", name); return MPN_EINVAL; }
  if (p->rn && nm != "bbox_raw" && nm != "cls_k" && nm != "cat" && !((nm == "cls" || nm == "fc7") && p->rn_region.empty())) { set_error("mpn_frcnn_debug_tensor: '%s' is not kept by the op-list / ResNet pipelines", name); return MPN_EINVAL; }
  MPN_CHECK_HIP(hipDeviceSynchronize());
  if (n * sizeof(float) > p->dbg_bytes) {
    if (p->dbg) (void)hipFree(p->dbg);
    p->dbg = nullptr; p->dbg_bytes = 0;
    MPN_CHECK_HIP(hipMalloc(&p->dbg, n * sizeof(float)));
    p->dbg_bytes = n * sizeof(float);
  }
  int rc = MPN_OK;
  if (nm == "conv5") {
    const ConvLayer &L = p->conv.back();
    rc = c8p_to_nchw(make_act(L.pool ? L.pooled : L.out, p->feat_c, h, w), p->dbg, nullptr);
  } else if (nm == "pooled") {
    hipLaunchKernelGGL(unpack_pooled_kernel, dim3((unsigned)cdiv_sz(n, 256)), dim3(256), 0, nullptr, p->x6, N, p->feat_c, PP, lin_mp(N), p->dbg);
    MPN_CHECK_LAUNCH();
  } else if (nm == "fc7") {
    rc = c8_to_rowmajor(p->y7_last ? p->y7_last : p->y7, N, F, p->dbg, nullptr);  // rows at stride lin_mp(N), as linear_c8 wrote them
  } else if (nm == "cls_k") {
    MPN_CHECK_HIP(hipMemcpy(p->dbg, p->cls_rm, n * sizeof(float), hipMemcpyDeviceToDevice));
  } else if (nm == "cat") {
    rc = c8_to_rowmajor(p->cat, N, (int)(n / N), p->dbg, nullptr);  // towers' [F/8][Mp][8] blocks back to back == one C8 matrix of towers * F columns
  } else if (nm == "cls") {
    hipLaunchKernelGGL(copy_cols_kernel, dim3((unsigned)cdiv_sz(n, 256)), dim3(256), 0, nullptr, p->head, 5 * C, 0, N, C, p->dbg);
    MPN_CHECK_LAUNCH();
  } else {
    MPN_CHECK_HIP(hipMemcpy(p->dbg, p->bbox_raw, n * sizeof(float), hipMemcpyDeviceToDevice));
  }
  if (rc) return rc;
  MPN_CHECK_HIP(hipDeviceSynchronize());
  *d_ptr = p->dbg;
  *n_elems = n;
  return MPN_OK;
}
