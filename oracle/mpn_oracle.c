/* mpn_oracle.c — CPU restatement of the MultiPathNet per-image detection hot path.
 *
 * ===========================  TEST INFRASTRUCTURE ONLY  ===========================
 * This file is the *checker*.  Only tests/, __graft_entry__.smoke() and the cpu_baseline leg of
 * bench.py may load it.  The product (multipathnet_amd/, libmpn_hip.so) never links, imports or
 * calls anything here, and fails loudly when its HIP library is missing.
 *
 * Every function cites the reference file:line (relative to /root/reference) it restates.
 *
 * Parity status:
 *   pinned   : orc_overlap / orc_nms / orc_bbox_vote  — checked bit-for-bit against the reference's
 *              own nms.c compiled unmodified (oracle/_ref/libnms_ref.so) and against the IoU
 *              known-answer vector of test.lua:40-52.
 *   pinned by in-tree Lua source (restated, no executable reference here): foveal, context_region,
 *              bbox_norm, bbox_decode (convertFrom), clamp, keep_top_k, image transformer,
 *              project_im_rois, select_boxes, integral mean.
 *   PARITY UNPINNED: roi_pool (inn.ROIPooling), conv3x3 (cudnn.SpatialConvolution), maxpool
 *              (nn.SpatialMaxPooling, ceil mode), linear (nn.Linear), softmax (nn.SoftMax),
 *              l2 normalize (nn.Normalize) — their source is in un-vendored, unpinned luarocks
 *              (inn, cudnn.torch, nn); restated from their published semantics and cross-checked
 *              against PyTorch-CPU in tests/test_oracle_dense.py.
 *
 * Arithmetic conventions: fp32 everywhere the reference is fp32, built with -ffp-contract=off so
 * a*b+c rounds twice (as x86-64 gcc -O3 without -mfma does for TH / nms.c).  Dense contractions
 * accumulate in fp32 in ascending-k order (k = cin*9 + ky*3 + kx for conv, k = input index for
 * linear), vectorised across *outputs* only, so no reassociation happens.
 */
#include <float.h>
#include <math.h>
#include <stddef.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

#define ORC_MIN(a, b) (((a) < (b)) ? (a) : (b))
#define ORC_MAX(a, b) (((a) > (b)) ? (a) : (b))

/* ------------------------------------------------------------------------------------------
 * NMS family — restates nms.c
 * ---------------------------------------------------------------------------------------- */

/* nms.c:14-41  IoU with the "+1" pixel convention; 0 when the intersection is empty. */
float orc_overlap(const float *a, const float *b) {
  float x1 = ORC_MAX(a[0], b[0]);
  float y1 = ORC_MAX(a[1], b[1]);
  float x2 = ORC_MIN(a[2], b[2]);
  float y2 = ORC_MIN(a[3], b[3]);
  float w = x2 - x1 + 1;
  float h = y2 - y1 + 1;
  float inter = w * h;
  float aarea = (a[2] - a[0] + 1) * (a[3] - a[1] + 1);
  float barea = (b[2] - b[0] + 1) * (b[3] - b[1] + 1);
  float iou = inter / (aarea + barea - inter);
  return (w <= 0 || h <= 0) ? 0 : iou;
}

/* nms.c:43-56  boxoverlap: IoU of each of N boxes [N,4] against one box b[4]. */
void orc_boxoverlap(const float *a, int n, const float *b, float *out) {
  for (int i = 0; i < n; ++i) out[i] = orc_overlap(a + 4 * i, b);
}

/* nms.c:59-108  greedy NMS.  scored_boxes [n,5] = {x1,y1,x2,y2,score}; writes kept rows (in
 * selection order) to keep[.,5] and, for index-level parity checks, the original row index of
 * every kept box to keep_idx.  Returns the number kept.
 *
 * Restated with an index permutation instead of a pointer array; the swap (nms.c:83-85) and the
 * stable partition with swaps (nms.c:91-98) are reproduced literally because they define the
 * winner among bit-equal scores.  Scores must be > -1e7 (nms.c:75) — otherwise the reference
 * reads boxes[-1]; we stop instead (documented deviation on undefined behaviour). */
int orc_nms(const float *sb, int n, float thr, float *keep, int *keep_idx) {
  if (n <= 0) return 0;
  int *order = (int *)malloc(sizeof(int) * (size_t)n);
  for (int i = 0; i < n; ++i) order[i] = i;
  int *cur = order;
  int num = n, kept = 0;
  while (num) {
    int best = -1;
    float bestS = -10000000;
    for (int i = 0; i < num; ++i) {
      if (sb[5 * cur[i] + 4] > bestS) { bestS = sb[5 * cur[i] + 4]; best = i; }
    }
    if (best < 0) break; /* reference: UB */
    int b = cur[best];
    int tmp = cur[0]; cur[0] = cur[best]; cur[best] = tmp;
    cur++;
    if (keep) memcpy(keep + 5 * kept, sb + 5 * b, sizeof(float) * 5);
    if (keep_idx) keep_idx[kept] = b;
    kept++;
    int good = 0;
    for (int i = 0; i < num - 1; ++i) {
      float iou = orc_overlap(sb + 5 * b, sb + 5 * cur[i]);
      if (iou <= thr) { tmp = cur[good]; cur[good++] = cur[i]; cur[i] = tmp; }
    }
    num = good;
  }
  free(order);
  return kept;
}

/* nms.c:110-142  bbox voting: for each kept box, score-weighted mean of all scored boxes with
 * IoU > thr (strict), accumulated sequentially in j order; score column = the NMS score. */
void orc_bbox_vote(const float *nms_boxes, int n_nms, const float *sb, int n, float thr, float *res) {
  for (int i = 0; i < n_nms; ++i) {
    float acc[5] = {0, 0, 0, 0, 0};
    for (int j = 0; j < n; ++j) {
      float ov = orc_overlap(sb + 5 * j, nms_boxes + 5 * i);
      if (ov > thr) {
        for (int f = 0; f < 4; ++f) acc[f] += sb[5 * j + f] * sb[5 * j + 4];
        acc[4] += sb[5 * j + 4];
      }
    }
    for (int f = 0; f < 4; ++f) res[5 * i + f] = acc[f] / acc[4];
    res[5 * i + 4] = nms_boxes[5 * i + 4];
  }
}

/* ------------------------------------------------------------------------------------------
 * Image side — ImageTransformer.lua, ImageDetect.lua
 * ---------------------------------------------------------------------------------------- */

/* modules/ImageTransformer.lua:19-33.  in [3,H,W]; out[i] = (in[swap[i]]*scale - mean[i]) / std[i].
 * The reference runs this on the DoubleTensor image and casts to float when it is copied into the
 * FloatTensor batch (ImageDetect.lua:44-50): compute in f64, round once to f32.
 * swap is 0-based here (Lua {3,2,1} -> {2,1,0}); has_std=0 skips the division (RossTransformer,
 * model_utils.lua:138-140). */
void orc_image_transform(const float *in, int H, int W, const int *swap, double scale, const double *mean,
                         const double *std, int has_std, float *out) {
  size_t plane = (size_t)H * W;
  for (int c = 0; c < 3; ++c) {
    const float *src = in + (size_t)swap[c] * plane;
    float *dst = out + (size_t)c * plane;
    for (size_t i = 0; i < plane; ++i) {
      double v = (double)src[i];
      if (scale != 1.0) v = v * scale;
      v = v + (-mean[c]);
      if (has_std) v = v / std[c];
      dst[i] = (float)v;
    }
  }
}

/* ImageDetect.lua:34-43  scale selection: s = target/min_side; if round(s*max_side) > max_size
 * then s = max_size/max_side.  (torch.round = half away from zero.) */
double orc_pick_scale(int H, int W, double target, double max_size) {
  double mn = H < W ? H : W, mx = H < W ? W : H;
  double s = target / mn;
  if (round(s * mx) > max_size) s = max_size / mx;
  return s;
}

/* ImageDetect.lua:54-73 (single-scale branch :66-70): rois[:,0]=1; rois[:,1:5]=(boxes-1)*s+1 as
 * three successive FloatTensor ops (add(-1), mul(s), add(1)) => three fp32 roundings, s cast to
 * fp32 by THFloatTensor_mul. */
void orc_project_im_rois(const float *boxes, int n, double scale, float *rois) {
  float s = (float)scale;
  for (int i = 0; i < n; ++i) {
    rois[5 * i] = 1.0f;
    for (int f = 0; f < 4; ++f) {
      float v = boxes[4 * i + f];
      v = v + (-1.0f);
      v = v * s;
      v = v + 1.0f;
      rois[5 * i + 1 + f] = v;
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Dense trunk ops — external L1 kernels (cudnn / nn); PARITY UNPINNED, restated semantics.
 * ---------------------------------------------------------------------------------------- */

/* cudnn.SpatialConvolution(Cin,Cout,3,3,1,1,1,1) + nn.ReLU as used in `features`
 * (models/vgg.lua:15,25; layer list multipathnet.lua:34-46): cross-correlation, stride 1, zero pad 1,
 * bias, optional ReLU.  in [Cin,H,W], w [Cout,Cin,3,3], out [Cout,H,W].
 * Accumulation order per output: bias first? — nn/cudnn add bias after the contraction; we start
 * from 0, add the 9*Cin products in ascending (cin,ky,kx) order, then add bias, then ReLU. */
void orc_conv3x3(const float *in, int Cin, int H, int W, const float *w, const float *bias, int Cout, int relu,
                 float *out) {
  size_t plane = (size_t)H * W;
#pragma omp parallel for schedule(dynamic, 1)
  for (int co = 0; co < Cout; ++co) {
    float *o = out + (size_t)co * plane;
    memset(o, 0, sizeof(float) * plane);
    for (int ci = 0; ci < Cin; ++ci) {
      const float *ip = in + (size_t)ci * plane;
      for (int ky = 0; ky < 3; ++ky) {
        for (int kx = 0; kx < 3; ++kx) {
          float wv = w[(((size_t)co * Cin + ci) * 3 + ky) * 3 + kx];
          int dy = ky - 1, dx = kx - 1;
          int y0 = dy < 0 ? 1 : 0, y1 = dy > 0 ? H - 1 : H;
          int x0 = dx < 0 ? 1 : 0, x1 = dx > 0 ? W - 1 : W;
          for (int y = y0; y < y1; ++y) {
            const float *irow = ip + (size_t)(y + dy) * W + dx;
            float *orow = o + (size_t)y * W;
            for (int x = x0; x < x1; ++x) orow[x] += wv * irow[x];
          }
        }
      }
    }
    float b = bias ? bias[co] : 0.0f;
    for (size_t i = 0; i < plane; ++i) {
      float v = o[i] + b;
      o[i] = (relu && v < 0.0f) ? 0.0f : v;
    }
  }
}

/* nn.SpatialMaxPooling(2,2,2,2):ceil() inside the Caffe-converted `features` blob
 * (test.lua:142 — a 600-px side maps to 38 rows, which only ceil mode gives).
 * in [C,H,W] -> out [C,ceil(H/2),ceil(W/2)]; windows hanging over the edge use in-bounds elements. */
void orc_maxpool2x2_ceil(const float *in, int C, int H, int W, float *out) {
  int Ho = (H + 1) / 2, Wo = (W + 1) / 2;
#pragma omp parallel for
  for (int c = 0; c < C; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float *op = out + (size_t)c * Ho * Wo;
    for (int y = 0; y < Ho; ++y)
      for (int x = 0; x < Wo; ++x) {
        float m = -INFINITY;
        for (int dy = 0; dy < 2; ++dy)
          for (int dx = 0; dx < 2; ++dx) {
            int yy = 2 * y + dy, xx = 2 * x + dx;
            if (yy < H && xx < W) { float v = ip[(size_t)yy * W + xx]; if (v > m) m = v; }
          }
        op[(size_t)y * Wo + x] = m;
      }
  }
}

/* nn.Linear (+ optional nn.ReLU): y[m][n] = sum_k x[m][k]*W[n][k] + b[n]  (vgg.lua:16,30 `top`;
 * model_utils.lua:105-119 heads).  x [M,K], W [N,K] row-major.  fp32 accumulation in ascending k,
 * bias added last.  Vectorised across n via a transposed copy of W (no reassociation). */
void orc_linear(const float *x, int M, int K, const float *W, const float *bias, int N, int relu, float *y) {
  float *Wt = (float *)malloc(sizeof(float) * (size_t)K * N);
#pragma omp parallel for
  for (int k0 = 0; k0 < K; k0 += 64)
    for (int n = 0; n < N; ++n)
      for (int k = k0; k < ORC_MIN(k0 + 64, K); ++k) Wt[(size_t)k * N + n] = W[(size_t)n * K + k];
  enum { MB = 4, NB = 64 };
  int nblk = (N + NB - 1) / NB;
#pragma omp parallel for schedule(dynamic, 1)
  for (int nb = 0; nb < nblk; ++nb) {
    int n0 = nb * NB, nn = ORC_MIN(NB, N - n0);
    for (int m0 = 0; m0 < M; m0 += MB) {
      int mm = ORC_MIN(MB, M - m0);
      float acc[MB][NB];
      for (int i = 0; i < MB; ++i)
        for (int j = 0; j < NB; ++j) acc[i][j] = 0.0f;
      for (int k = 0; k < K; ++k) {
        const float *wr = Wt + (size_t)k * N + n0;
        for (int i = 0; i < mm; ++i) {
          float xv = x[(size_t)(m0 + i) * K + k];
          for (int j = 0; j < nn; ++j) acc[i][j] += xv * wr[j];
        }
      }
      for (int i = 0; i < mm; ++i)
        for (int j = 0; j < nn; ++j) {
          float v = acc[i][j] + (bias ? bias[n0 + j] : 0.0f);
          y[(size_t)(m0 + i) * N + n0 + j] = (relu && v < 0.0f) ? 0.0f : v;
        }
    }
  }
  free(Wt);
}

/* nn.SoftMax over the class dimension (ImageDetect.lua:19,189-191): exp(x-max)/sum, fp32. */
void orc_softmax(const float *x, int M, int C, float *y) {
  for (int m = 0; m < M; ++m) {
    const float *r = x + (size_t)m * C;
    float *o = y + (size_t)m * C;
    float mx = -INFINITY;
    for (int c = 0; c < C; ++c) mx = ORC_MAX(mx, r[c]);
    float sum = 0.0f;
    for (int c = 0; c < C; ++c) { o[c] = expf(r[c] - mx); sum += o[c]; }
    for (int c = 0; c < C; ++c) o[c] = o[c] / sum;
  }
}

/* ------------------------------------------------------------------------------------------
 * inn.ROIPooling(W,H,scale) — external, source absent; PARITY UNPINNED.
 * Fast R-CNN ROI max pooling restated from the published algorithm, with the reference README's
 * "v2" coordinate fix (README.md:202-203: 1-based pixel coords are shifted before scaling):
 *     start = round((x1 - coord_offset) * scale)      end = round((x2 - coord_offset) * scale) + end_adjust
 *     roi_w = max(end_w - start_w + 1, 1)             bin_w = roi_w / PW   (fp32)
 *     wstart = floor(pw * bin_w) + start_w            wend = ceil((pw+1) * bin_w) + start_w   (clipped to [0,W])
 * max over the bin, empty bin -> 0 and argmax -1.  rois [N,5] = {batch(1-based), x1,y1,x2,y2}.
 * feat [B,C,H,W], out [N,C,PH,PW], argmax [N,C,PH,PW] int32 = h*W+w inside the feature plane.
 * round() is C roundf (half away from zero) on the fp32 product.
 * Call sites: vgg.lua:28 (7,7,1/16), alexnet.lua:23 (6,6,1/16), resnet.lua:48, inceptionv3.lua:41,
 * model_utils.lua:215 (7x7 at 1/16, 1/8, 1/4). */
void orc_roi_pool(const float *feat, int B, int C, int H, int W, const float *rois, int N, int PH, int PW,
                  float scale, float coord_offset, int end_adjust, float *out, int32_t *argmax) {
#pragma omp parallel for schedule(dynamic, 4)
  for (int n = 0; n < N; ++n) {
    const float *r = rois + 5 * n;
    int b = (int)r[0] - 1;
    if (b < 0) b = 0;
    if (b >= B) b = B - 1;
    int sw = (int)roundf((r[1] - coord_offset) * scale);
    int sh = (int)roundf((r[2] - coord_offset) * scale);
    int ew = (int)roundf((r[3] - coord_offset) * scale) + end_adjust;
    int eh = (int)roundf((r[4] - coord_offset) * scale) + end_adjust;
    int rw = ORC_MAX(ew - sw + 1, 1), rh = ORC_MAX(eh - sh + 1, 1);
    float bh = (float)rh / (float)PH, bw = (float)rw / (float)PW;
    for (int c = 0; c < C; ++c) {
      const float *fp = feat + ((size_t)b * C + c) * H * W;
      for (int ph = 0; ph < PH; ++ph)
        for (int pw = 0; pw < PW; ++pw) {
          int hs = (int)floorf((float)ph * bh) + sh, he = (int)ceilf((float)(ph + 1) * bh) + sh;
          int ws = (int)floorf((float)pw * bw) + sw, we = (int)ceilf((float)(pw + 1) * bw) + sw;
          hs = ORC_MIN(ORC_MAX(hs, 0), H); he = ORC_MIN(ORC_MAX(he, 0), H);
          ws = ORC_MIN(ORC_MAX(ws, 0), W); we = ORC_MIN(ORC_MAX(we, 0), W);
          int empty = (he <= hs) || (we <= ws);
          float m = empty ? 0.0f : -INFINITY;
          int mi = -1;
          for (int h = hs; h < he; ++h)
            for (int w = ws; w < we; ++w) {
              float v = fp[(size_t)h * W + w];
              if (v > m) { m = v; mi = h * W + w; }
            }
          size_t o = (((size_t)n * C + c) * PH + ph) * PW + pw;
          out[o] = m;
          if (argmax) argmax[o] = mi;
        }
    }
  }
}

/* inn.ROIPooling's OTHER branch — the module's CPU path (the "CPU nn path" of BASELINE configs[0]; call sites
 * models/alexnet.lua:23, models/vgg.lua:28 when the tensors are float): PARITY UNPINNED (external rock, source absent), restated as
 *     rois[{{},{2,5}}]:add(-1):mul(spatial_scale):add(1):round()      -- corners in 1-based map coordinates, fp32 tensor ops in this order
 *     x1, x2 clipped to the map width, y1, y2 to its height           -- the reference clips the upper side (cmin) and indexes the tensor
 *                                                                        with the lower side, which must be >= 1; we clip both sides
 *     im = data[{b, {}, {y1, y2}, {x1, x2}}]                          -- the inclusive crop
 *     out[n] = nn.SpatialAdaptiveMaxPooling(PW, PH):forward(im)       -- bin i of a side of length L: [floor(i * L / P), ceil((i + 1) * L / P))
 *                                                                        (THNN's START_IND / END_IND: (int)floor((float)(i * L) / P)),
 *                                                                        max from -FLT_MAX upwards with `>`
 * bin_rule 0 is orc_roi_pool (the CUDA branch: bins of the un-clipped window, clipped afterwards, empty bins -> 0); 1 is this one.  The two
 * differ on windows whose rounded corners leave the map (Foveal's regions, border boxes) and, once in a thousand, by one cell.  argmax = h * W + w in the feature plane for both.
 * Written as crop-then-pool, as the module does it — deliberately not sharing code with the CUDA-branch restatement above. */
void orc_roi_pool_rule(const float *feat, int B, int C, int H, int W, const float *rois, int N, int PH, int PW, float scale,
                       float coord_offset, int end_adjust, int bin_rule, float *out, int32_t *argmax) {
  if (bin_rule == 0) { orc_roi_pool(feat, B, C, H, W, rois, N, PH, PW, scale, coord_offset, end_adjust, out, argmax); return; }
#pragma omp parallel for schedule(dynamic, 4)
  for (int n = 0; n < N; ++n) {
    const float *r = rois + 5 * n;
    int b = (int)r[0] - 1;
    if (b < 0) b = 0;
    if (b >= B) b = B - 1;
    float c1[4];
    for (int k = 0; k < 4; ++k) {
      float t = r[1 + k] - coord_offset;
      t = t * scale;
      t = t + 1.0f;
      c1[k] = roundf(t);
    }
    int x1 = (int)c1[0], y1 = (int)c1[1], x2 = (int)c1[2] + end_adjust, y2 = (int)c1[3] + end_adjust;  /* 1-based, inclusive */
    x1 = ORC_MIN(ORC_MAX(x1, 1), W); x2 = ORC_MIN(ORC_MAX(x2, 1), W);
    y1 = ORC_MIN(ORC_MAX(y1, 1), H); y2 = ORC_MIN(ORC_MAX(y2, 1), H);
    int iw = x2 - x1 + 1, ih = y2 - y1 + 1;
    if (iw < 1) iw = 1;
    if (ih < 1) ih = 1;
    for (int c = 0; c < C; ++c) {
      const float *crop = feat + ((size_t)b * C + c) * H * W + (size_t)(y1 - 1) * W + (x1 - 1);  /* crop(y, x) = crop[y * W + x] */
      for (int i = 0; i < PH; ++i) {
        int ys = (int)floorf((float)(i * ih) / (float)PH), ye = (int)ceilf((float)((i + 1) * ih) / (float)PH);
        for (int j = 0; j < PW; ++j) {
          int xs = (int)floorf((float)(j * iw) / (float)PW), xe = (int)ceilf((float)((j + 1) * iw) / (float)PW);
          float m = -FLT_MAX;
          int mi = -1;
          for (int y = ys; y < ye; ++y)
            for (int x = xs; x < xe; ++x) {
              float v = crop[(size_t)y * W + x];
              if (v > m) { m = v; mi = (y1 - 1 + y) * W + (x1 - 1 + x); }
            }
          size_t o = (((size_t)n * C + c) * PH + i) * PW + j;
          out[o] = m;
          if (argmax) argmax[o] = mi;
        }
      }
    }
  }
}

/* ------------------------------------------------------------------------------------------
 * Box-geometry modules — modules/X.lua files
 * ---------------------------------------------------------------------------------------- */

/* modules/Foveal.lua:15-44: each ROI {id,x,y,x2,y2} -> 4 rows {id,x',y',x'+w',y'+h'} with
 * (x',y',w',h') = itself; (x-.25w, y-.25h, 1.5w, 1.5h); (x-.5w, y-.5h, 2w, 2h); (x-1.5w, y-1.5h,
 * 4w, 4h).  The Lua loop runs in double (box:totable()) and rounds to fp32 on the FloatTensor
 * store (:26-28,:32-39).  out [4N,5], rows grouped per ROI. */
void orc_foveal(const float *rois, int n, float *out) {
  static const double off[4] = {0.0, 0.25, 0.5, 1.5};
  static const double mul[4] = {1.0, 1.5, 2.0, 4.0};
  for (int i = 0; i < n; ++i) {
    double id = rois[5 * i], x = rois[5 * i + 1], y = rois[5 * i + 2], x2 = rois[5 * i + 3], y2 = rois[5 * i + 4];
    double w = x2 - x, h = y2 - y;
    float *o = out + 20 * (size_t)i;
    memcpy(o, rois + 5 * i, sizeof(float) * 5); /* base[1]:copy(box) */
    for (int r = 1; r < 4; ++r) {
      double xx = x - w * off[r], yy = y - h * off[r], ww = w * mul[r], hh = h * mul[r];
      o[5 * r + 0] = (float)id;
      o[5 * r + 1] = (float)xx;
      o[5 * r + 2] = (float)yy;
      o[5 * r + 3] = (float)(xx + ww);
      o[5 * r + 4] = (float)(yy + hh);
    }
  }
}

/* modules/ContextRegion.lua:14-32: out[:,0]=in[:,0]; out[:,1:5] = in[:,1:5] * T, T built in double
 * from a=(1+s)/2, b=(1-s)/2 then cast to the module's fp32 type.  The fp32 mm has two non-zero
 * terms per output; we evaluate it as fl(fl(a*p) + fl(b*q)) with the products in ascending input
 * column order (zeros contribute exactly 0). */
void orc_context_region(const float *rois, int n, double scale, float *out) {
  float a = (float)((1.0 + scale) / 2.0), b = (float)((1.0 - scale) / 2.0);
  for (int i = 0; i < n; ++i) {
    const float *r = rois + 5 * i;
    float *o = out + 5 * i;
    o[0] = r[0];
    o[1] = a * r[1] + b * r[3];
    o[2] = a * r[2] + b * r[4];
    o[3] = b * r[1] + a * r[3];
    o[4] = b * r[2] + a * r[4];
  }
}

/* modules/BBoxNorm.lua:18-32 (eval): view(-1,4) * std + mean, in place, fp32 (cmul then add). */
void orc_bbox_norm(float *bbox, int n, int C4, const float *mean4, const float *std4) {
  for (int i = 0; i < n; ++i)
    for (int j = 0; j < C4; ++j) {
      float v = bbox[(size_t)i * C4 + j] * std4[j & 3];
      bbox[(size_t)i * C4 + j] = v + mean4[j & 3];
    }
}

/* utils.lua:229-247 convertFrom (2-D path), applied per 4-column class block with the ORIGINAL image
 * boxes (ImageDetect.lua:183-185).  boxes [N,4], deltas [N,4C] -> out [N,4C].  Tensor-op order:
 * xc=(x1+x2)*0.5; w=x2-x1 (no +1); xtc=xc+dx*w (addcmul: product rounded, then sum);
 * wt=exp(dw)*w; out = xtc -/+ wt*0.5. */
void orc_bbox_decode(const float *boxes, const float *deltas, int n, int C, float *out) {
  for (int i = 0; i < n; ++i) {
    const float *bx = boxes + 4 * i;
    float xc = (bx[0] + bx[2]) * 0.5f, yc = (bx[1] + bx[3]) * 0.5f;
    float w = bx[2] - bx[0], h = bx[3] - bx[1];
    for (int c = 0; c < C; ++c) {
      const float *d = deltas + (size_t)i * 4 * C + 4 * c;
      float *o = out + (size_t)i * 4 * C + 4 * c;
      float p0 = d[0] * w, p1 = d[1] * h;
      float xtc = xc + p0, ytc = yc + p1;
      float wt = expf(d[2]) * w, ht = expf(d[3]) * h;
      float hw = wt * 0.5f, hh = ht * 0.5f;
      o[0] = xtc - hw; o[1] = ytc - hh; o[2] = xtc + hw; o[3] = ytc + hh;
    }
  }
}

/* Tester_FRCNN.lua:75-78: view(-1,2); x clamped to [1, im_W], y clamped to [1, im_H]. */
void orc_clamp_boxes(float *bbox, size_t n_pairs, float im_w, float im_h) {
  for (size_t i = 0; i < n_pairs; ++i) {
    float x = bbox[2 * i], y = bbox[2 * i + 1];
    bbox[2 * i] = x < 1.0f ? 1.0f : (x > im_w ? im_w : x);
    bbox[2 * i + 1] = y < 1.0f ? 1.0f : (y > im_h ? im_h : y);
  }
}

/* Tester_FRCNN.lua:106-116: class j (1..C-1): rows with score > thresh -> [M,5] {box_j, score_j}.
 * scores [N,C], bbox [N,4C].  Returns M; also writes source row indices when idx != NULL. */
int orc_select_scored(const float *scores, const float *bbox, int n, int C, int cls, float thresh, float *sb,
                      int *idx) {
  int m = 0;
  for (int i = 0; i < n; ++i) {
    float s = scores[(size_t)i * C + cls];
    if (s > thresh) {
      memcpy(sb + 5 * (size_t)m, bbox + (size_t)i * 4 * C + 4 * cls, sizeof(float) * 4);
      sb[5 * (size_t)m + 4] = s;
      if (idx) idx[m] = i;
      ++m;
    }
  }
  return m;
}

static int cmp_desc(const void *a, const void *b) {
  float x = *(const float *)a, y = *(const float *)b;
  return (x < y) - (x > y);
}
/* utils.lua:75-96 keep_top_k: thresh = k-th largest score over all classes' kept boxes (or the
 * smallest if fewer than k); every row with score >= thresh survives (ties kept).  Returns thresh;
 * total==0 -> returns 0 (reference returns `boxes, 0`). */
float orc_topk_threshold(const float *scores, int total, int k) {
  if (total <= 0) return 0.0f;
  float *tmp = (float *)malloc(sizeof(float) * (size_t)total);
  memcpy(tmp, scores, sizeof(float) * (size_t)total);
  qsort(tmp, (size_t)total, sizeof(float), cmp_desc);
  float t = tmp[ORC_MIN(total, k) - 1];
  free(tmp);
  return t;
}

/* modules/SelectBoxes.lua:26-56: per row, arg-max class (first max) -> its 4 regressed coords. */
void orc_select_boxes(const float *scores, const float *bbox, int n, int C, float *out) {
  for (int i = 0; i < n; ++i) {
    int best = 0;
    for (int c = 1; c < C; ++c)
      if (scores[(size_t)i * C + c] > scores[(size_t)i * C + best]) best = c;
    memcpy(out + 4 * (size_t)i, bbox + (size_t)i * 4 * C + 4 * best, sizeof(float) * 4);
  }
}

/* nn.Normalize(2) as used by conv345Combine (model_utils.lua:216-223): per row x / (sum x^2 + eps)^0.5,
 * eps=1e-10 (nn.Normalize default).  PARITY UNPINNED (external nn). rows [M,D]. */
void orc_l2_normalize(const float *x, int M, int D, float *y) {
#pragma omp parallel for
  for (int m = 0; m < M; ++m) {
    const float *r = x + (size_t)m * D;
    float s = 0.0f;
    for (int d = 0; d < D; ++d) s += r[d] * r[d];
    float nrm = sqrtf(s + 1e-10f);
    for (int d = 0; d < D; ++d) y[(size_t)m * D + d] = r[d] / nrm;
  }
}

/* model_utils.lua:296-313 integral eval head: mean over K of per-classifier softmaxes.
 * probs [K,N,C] -> out [N,C]: sum in k order then divide by K (nn.Mean = sum * (1/K)). */
void orc_mean_over_k(const float *probs, int K, int N, int C, float *out) {
  size_t nc = (size_t)N * C;
  float inv = 1.0f / (float)K;
  for (size_t i = 0; i < nc; ++i) {
    float s = 0.0f;
    for (int k = 0; k < K; ++k) s += probs[(size_t)k * nc + i];
    out[i] = s * inv;
  }
}


/* ------------------------------------------------------------------------------------------
 * image.scale(src, width, height) in its default 'bilinear' mode — external `image` rock (ImageDetect.lua:41);
 * PARITY UNPINNED (source not in the tree).  Restated from the published algorithm of torch/image's
 * generic/image.c (scaleLinear_rowcol): separable, rows first then columns through a float intermediate;
 *   upscale   : dst[d] = (1-f)*src[i] + f*src[i+1] with i+f = d*(src_len-1)/(dst_len-1), last sample copied;
 *   downscale : fractional box average over [d*scale, (d+1)*scale), scale = src_len/dst_len;
 *   equal     : copy.
 * Output size as ImageDetect.lua:40 builds it: (long)(H*s) x (long)(W*s). */
static void orc_scale_line(const float *src, long sstride, long slen, float *dst, long dstride, long dlen) {
  if (dlen > slen) {
    if (slen == 1) { for (long d = 0; d < dlen; ++d) dst[d * dstride] = src[0]; return; }
    float scale = (float)(slen - 1) / (float)(dlen - 1);
    for (long d = 0; d < dlen - 1; ++d) {
      float sf = (float)d * scale;
      long si = (long)sf;
      sf -= (float)si;
      dst[d * dstride] = (1.0f - sf) * src[si * sstride] + sf * src[(si + 1) * sstride];
    }
    dst[(dlen - 1) * dstride] = src[(slen - 1) * sstride];
  } else if (dlen < slen) {
    float scale = (float)slen / (float)dlen;
    for (long d = 0; d < dlen; ++d) {
      float s0 = (float)d * scale, s1 = (float)(d + 1) * scale;
      long i0 = (long)s0, i1 = (long)s1;
      float f0 = s0 - (float)i0, f1 = s1 - (float)i1;
      float acc = (1.0f - f0) * src[i0 * sstride], n = 1.0f - f0;
      for (long i = i0 + 1; i < i1; ++i) { acc += src[i * sstride]; n += 1.0f; }
      if (i1 < slen && i1 > i0) { acc += f1 * src[i1 * sstride]; n += f1; }
      dst[d * dstride] = acc / n;
    }
  } else {
    for (long d = 0; d < dlen; ++d) dst[d * dstride] = src[d * sstride];
  }
}

void orc_image_scale(const float *in, int C, int H, int W, int H2, int W2, float *out) {
  float *tmp = (float *)malloc(sizeof(float) * (size_t)H * W2);
  for (int c = 0; c < C; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float *op = out + (size_t)c * H2 * W2;
    for (int y = 0; y < H; ++y) orc_scale_line(ip + (size_t)y * W, 1, W, tmp + (size_t)y * W2, 1, W2);
    for (int x = 0; x < W2; ++x) orc_scale_line(tmp + x, W2, H, op + x, W2, H2);
  }
  free(tmp);
}

/* ---- ResNet Fast R-CNN (models/resnet.lua:24-50) -------------------------------------------------
 * The network itself comes from a fb.resnet.torch `.t7` that is not in the tree (resnet.lua:17,25): layers 1-7 =
 * conv1 7x7/2 pad 3, BN, ReLU, SpatialMaxPooling(3,3,2,2,1,1), layer1-3; layers 8-10 = layer4, 7x7 average pool, View.
 * BN is folded to a fixed per-channel scale/shift at load (inn.utils.BNtoFixed, resnet.lua:34-36) and, here, further
 * into the preceding convolution's weights and bias by the caller.  PARITY UNPINNED: restated from the public
 * fb.resnet.torch definition (bottleneck: 1x1, 3x3 carrying the stride, 1x1 x4; shortcut type B = strided 1x1 conv). */

/* generic cross-correlation: in [B,Cin,H,W], w [Cout,Cin,KH,KW], stride s, zero pad p -> out [B,Cout,OH,OW],
 * OH = (H + 2p - KH)/s + 1 (floor).  Sum in ascending (cin,ky,kx) order, then + bias, + residual (same shape as out,
 * may be NULL), then optional ReLU. */
void orc_conv2d(const float *in, int B, int Cin, int H, int W, const float *w, const float *bias, int Cout, int KH, int KW,
                int stride, int pad, const float *residual, int relu, float *out) {
  const int OH = (H + 2 * pad - KH) / stride + 1, OW = (W + 2 * pad - KW) / stride + 1;
  const size_t oplane = (size_t)OH * OW, iplane = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float *o = out + ((size_t)b * Cout + co) * oplane;
      memset(o, 0, sizeof(float) * oplane);
      for (int ci = 0; ci < Cin; ++ci) {
        const float *ip = in + ((size_t)b * Cin + ci) * iplane;
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < KW; ++kx) {
            const float wv = w[(((size_t)co * Cin + ci) * KH + ky) * KW + kx];
            for (int oy = 0; oy < OH; ++oy) {
              const int iy = oy * stride + ky - pad;
              if (iy < 0 || iy >= H) continue;
              const float *irow = ip + (size_t)iy * W;
              float *orow = o + (size_t)oy * OW;
              for (int ox = 0; ox < OW; ++ox) {
                const int ix = ox * stride + kx - pad;
                if (ix >= 0 && ix < W) orow[ox] += wv * irow[ix];
              }
            }
          }
      }
      const float bv = bias ? bias[co] : 0.0f;
      const float *r = residual ? residual + ((size_t)b * Cout + co) * oplane : NULL;
      for (size_t i = 0; i < oplane; ++i) {
        float v = o[i] + bv;
        if (r) v += r[i];
        o[i] = (relu && v < 0.0f) ? 0.0f : v;
      }
    }
}

/* nn.SpatialMaxPooling(k,k,s,s,p,p), floor mode (fb.resnet.torch's 3x3/2 pad 1): padded positions never win. */
void orc_maxpool2d(const float *in, int BC, int H, int W, int k, int stride, int pad, float *out) {
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
#pragma omp parallel for
  for (int c = 0; c < BC; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float *op = out + (size_t)c * OH * OW;
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        float m = -INFINITY;
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) {
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) { const float v = ip[(size_t)iy * W + ix]; if (v > m) m = v; }
          }
        op[(size_t)oy * OW + ox] = m;
      }
  }
}

/* nn.SpatialMaxPooling(k,k,s,s,p,p) with an explicit output size (:ceil() / Caffe rounding is decided by the caller): windows are
 * clipped to the map, padded cells never win.  models/alexnet.lua's pool1 / pool2 (external nn rock; parity unpinned, cross-checked
 * against PyTorch-CPU max_pool2d(ceil_mode=True)). */
void orc_maxpool2d_out(const float *in, int BC, int H, int W, int k, int stride, int pad, int OH, int OW, float *out) {
#pragma omp parallel for
  for (int c = 0; c < BC; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float *op = out + (size_t)c * OH * OW;
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        float m = -INFINITY;
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) {
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) { const float v = ip[(size_t)iy * W + ix]; if (v > m) m = v; }
          }
        op[(size_t)oy * OW + ox] = m;
      }
  }
}

/* nn.SpatialCrossMapLRN(size, alpha, beta, k) (external nn rock; models/alexnet.lua's norm1 / norm2; parity unpinned, cross-checked
 * against PyTorch-CPU local_response_norm): out_c = in_c * (k + alpha/size * sum_{|c'-c| <= (size-1)/2} in_c'^2) ^ -beta, squares
 * summed in ascending channel order, fp32. */
void orc_lrn(const float *in, int B, int C, int HW, int size, float alpha, float beta, float k, float *out) {
  const int half = (size - 1) / 2;
  const float a = alpha / (float)size;
#pragma omp parallel for
  for (int b = 0; b < B; ++b)
    for (int c = 0; c < C; ++c)
      for (int p = 0; p < HW; ++p) {
        float ssum = 0.0f;
        for (int d = -half; d <= half; ++d) {
          const int cc = c + d;
          if (cc >= 0 && cc < C) { const float x = in[((size_t)b * C + cc) * HW + p]; ssum = ssum + x * x; }
        }
        const float sc = k + a * ssum;
        out[((size_t)b * C + c) * HW + p] = in[((size_t)b * C + c) * HW + p] * powf(sc, -beta);
      }
}

/* nn.SpatialAveragePooling over the whole map (7x7 after layer4): sum in row-major order, then * 1/(H*W). */
void orc_avgpool_global(const float *in, int BC, int H, int W, float *out) {
  const float inv = 1.0f / (float)(H * W);
#pragma omp parallel for
  for (int c = 0; c < BC; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float sacc = 0.0f;
    for (int i = 0; i < H * W; ++i) sacc += ip[i];
    out[c] = sacc * inv;
  }
}

/* ---- branching graphs (models/inceptionv3.lua:27-43): asymmetric kernels and count_include_pad average pooling -------------- */
void orc_conv2d_rect(const float *in, int B, int Cin, int H, int W, const float *w, const float *bias, int Cout, int KH, int KW, int sh,
                     int sw, int ph, int pw, int relu, float *out) {
  const int OH = (H + 2 * ph - KH) / sh + 1, OW = (W + 2 * pw - KW) / sw + 1;
  const size_t oplane = (size_t)OH * OW, iplane = (size_t)H * W;
#pragma omp parallel for collapse(2) schedule(dynamic, 1)
  for (int b = 0; b < B; ++b)
    for (int co = 0; co < Cout; ++co) {
      float *o = out + ((size_t)b * Cout + co) * oplane;
      memset(o, 0, sizeof(float) * oplane);
      for (int ci = 0; ci < Cin; ++ci) {
        const float *ip = in + ((size_t)b * Cin + ci) * iplane;
        for (int ky = 0; ky < KH; ++ky)
          for (int kx = 0; kx < KW; ++kx) {
            const float wv = w[(((size_t)co * Cin + ci) * KH + ky) * KW + kx];
            for (int oy = 0; oy < OH; ++oy) {
              const int iy = oy * sh + ky - ph;
              if (iy < 0 || iy >= H) continue;
              const float *irow = ip + (size_t)iy * W;
              float *orow = o + (size_t)oy * OW;
              for (int ox = 0; ox < OW; ++ox) {
                const int ix = ox * sw + kx - pw;
                if (ix >= 0 && ix < W) orow[ox] += wv * irow[ix];
              }
            }
          }
      }
      const float bv = bias ? bias[co] : 0.0f;
      for (size_t i = 0; i < oplane; ++i) {
        const float v = o[i] + bv;
        o[i] = (relu && v < 0.0f) ? 0.0f : v;
      }
    }
}

/* nn.SpatialAveragePooling(k,k,s,s,p,p) with count_include_pad = true (torch's default): in-map cells summed in row-major
 * order, divided by k*k whatever the window's overlap with the padding. */
void orc_avgpool2d(const float *in, int BC, int H, int W, int k, int stride, int pad, float *out) {
  const int OH = (H + 2 * pad - k) / stride + 1, OW = (W + 2 * pad - k) / stride + 1;
  const float inv = 1.0f / (float)(k * k);
#pragma omp parallel for
  for (int c = 0; c < BC; ++c) {
    const float *ip = in + (size_t)c * H * W;
    float *op = out + (size_t)c * OH * OW;
    for (int oy = 0; oy < OH; ++oy)
      for (int ox = 0; ox < OW; ++ox) {
        float sacc = 0.0f;
        for (int ky = 0; ky < k; ++ky)
          for (int kx = 0; kx < k; ++kx) {
            const int iy = oy * stride + ky - pad, ix = ox * stride + kx - pad;
            if (iy >= 0 && iy < H && ix >= 0 && ix < W) sacc += ip[(size_t)iy * W + ix];
          }
        op[(size_t)oy * OW + ox] = sacc * inv;
      }
  }
}
