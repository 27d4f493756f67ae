/* A HOST WRITTEN IN PLAIN C over the C ABI of include/mpn.h — what the reference's LuaJIT host does through ffi.cdef
 * (utils.lua:15-39 for libnms; INTEGRATION.md §3 for the whole path), minus Lua, which this image does not have.
 * No torch, no Python, no C++: device memory comes from the HIP runtime's C API, exactly as a cutorch tensor's would.
 *
 *   frcnn_host <model+inputs blob> <out file>
 *
 * blob (little-endian, written by tests/test_c_host.py with numpy):
 *   int32  magic 0x4d504e31, n_conv, conv_cout[n_conv], pool_after[n_conv], fc_dim, n_classes, pooled, H, W, N
 *   float  per conv layer: w[cout][cin][3][3], b[cout]   (cin of layer 0 = 3)
 *   float  fc6_w[fc][C5*pooled^2], fc6_b, fc7_w[fc][fc], fc7_b, cls_w[C][fc], cls_b, bbox_w[4C][fc], bbox_b
 *   float  bbox_mean[4], bbox_std[4]   (nn.BBoxNorm; std[0] == 0: module absent)
 *   float  image[3][H][W] (RGB in [0,1], as image.load returns it), boxes[N][4] (x1 y1 x2 y2, 1-based)
 * out:  int32 n_dets, float dets[n_dets][6] {x1,y1,x2,y2,score,class}  from Tester:testOne + keep_top_k —
 *       once through mpn_frcnn_test_one (device inputs) and once through the pipelined host-fed loop form; the program fails
 *       if the two disagree in a single bit. */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include <hip/hip_runtime_api.h>

#include "mpn.h"

#define CHECK_HIP(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); return 2; } } while (0)
#define CHECK_MPN(x) do { int r_ = (x); if (r_ != MPN_OK) { fprintf(stderr, "%s: %d (%s)\n", #x, r_, mpn_last_error()); return 3; } } while (0)

static const float *cursor;

static int upload(size_t n, float **d) {
  CHECK_HIP(hipMalloc((void **)d, n * sizeof(float)));
  CHECK_HIP(hipMemcpy(*d, cursor, n * sizeof(float), hipMemcpyHostToDevice));
  cursor += n;
  return 0;
}

int main(int argc, char **argv) {
  if (argc != 3) { fprintf(stderr, "usage: %s <blob> <out>\n", argv[0]); return 1; }
  FILE *f = fopen(argv[1], "rb");
  if (!f) { perror(argv[1]); return 1; }
  fseek(f, 0, SEEK_END);
  long sz = ftell(f);
  fseek(f, 0, SEEK_SET);
  char *blob = (char *)malloc((size_t)sz);
  if (!blob || fread(blob, 1, (size_t)sz, f) != (size_t)sz) { fprintf(stderr, "short read\n"); return 1; }
  fclose(f);
  const int32_t *hi = (const int32_t *)blob;
  if (hi[0] != 0x4d504e31) { fprintf(stderr, "bad magic\n"); return 1; }
  const int n_conv = hi[1];
  const int32_t *cout = hi + 2, *pool_after = hi + 2 + n_conv;
  const int32_t *tail = hi + 2 + 2 * n_conv;
  const int fc = tail[0], C = tail[1], pooled = tail[2], H = tail[3], W = tail[4], N = tail[5];
  cursor = (const float *)(tail + 6);

  int ndev = 0;
  CHECK_HIP(hipGetDeviceCount(&ndev));
  if (ndev < 1) { fprintf(stderr, "no HIP device (the library has no CPU path)\n"); return 2; }
  CHECK_HIP(hipSetDevice(0));
  printf("libmpn_hip version %d\n", mpn_version());

  float *conv_w[32], *conv_b[32];
  int cin = 3;
  for (int l = 0; l < n_conv; ++l) {
    if (upload((size_t)cout[l] * cin * 9, &conv_w[l]) || upload((size_t)cout[l], &conv_b[l])) return 2;
    cin = cout[l];
  }
  const size_t k6 = (size_t)cin * pooled * pooled;
  float *fc6_w, *fc6_b, *fc7_w, *fc7_b, *cls_w, *cls_b, *bbox_w, *bbox_b, *d_im, *d_boxes;
  if (upload((size_t)fc * k6, &fc6_w) || upload(fc, &fc6_b) || upload((size_t)fc * fc, &fc7_w) || upload(fc, &fc7_b) ||
      upload((size_t)C * fc, &cls_w) || upload(C, &cls_b) || upload((size_t)4 * C * fc, &bbox_w) || upload((size_t)4 * C, &bbox_b)) return 2;
  const float *h_bbox_norm = cursor;  /* mean[4], std[4]: host-side configuration, not a device tensor */
  cursor += 8;
  const float *h_im = cursor, *h_boxes = cursor + (size_t)3 * H * W;
  if (upload((size_t)3 * H * W, &d_im) || upload((size_t)N * 4, &d_boxes)) return 2;
  if ((const char *)cursor != blob + sz) { fprintf(stderr, "blob size mismatch\n"); return 1; }

  mpn_frcnn_config cfg;
  memset(&cfg, 0, sizeof cfg);
  int cc[32], pa[32];
  for (int l = 0; l < n_conv; ++l) { cc[l] = cout[l]; pa[l] = pool_after[l]; }
  cfg.n_conv = n_conv; cfg.conv_cout = cc; cfg.pool_after = pa;
  cfg.pooled_h = cfg.pooled_w = pooled;
  int n_pool = 0;
  for (int l = 0; l < n_conv; ++l) n_pool += pa[l];
  cfg.spatial_scale = 1.0f / (float)(1 << n_pool);
  cfg.fc_dim = fc; cfg.n_classes = C; cfg.max_h = H; cfg.max_w = W; cfg.max_rois = N;
  /* fbcoco.ImageTransformer as models/vgg.lua builds it ("Ross" Caffe-style): x255, BGR, mean subtraction, no std */
  cfg.tf_scale = 255.0; cfg.tf_mean[0] = 102.9801; cfg.tf_mean[1] = 115.9465; cfg.tf_mean[2] = 122.7717;
  cfg.tf_swap[0] = 2; cfg.tf_swap[1] = 1; cfg.tf_swap[2] = 0;
  for (int i = 0; i < 4; ++i) { cfg.bbox_mean[i] = h_bbox_norm[i]; cfg.bbox_std[i] = h_bbox_norm[4 + i]; }
  cfg.nms_thresh = 0.3f; cfg.score_thresh = -1.5f; cfg.top_k = 100; cfg.num_iter = 1;

  mpn_frcnn *net = NULL;
  CHECK_MPN(mpn_frcnn_create(&cfg, (const float *const *)conv_w, (const float *const *)conv_b, fc6_w, fc6_b, fc7_w, fc7_b, cls_w, cls_b, bbox_w,
                             bbox_b, &net));

  const int cap = 4 * cfg.top_k + 64;
  float *d_dets[2];
  int *d_n[2];
  for (int i = 0; i < 2; ++i) {
    CHECK_HIP(hipMalloc((void **)&d_dets[i], (size_t)cap * 6 * sizeof(float)));
    CHECK_HIP(hipMalloc((void **)&d_n[i], sizeof(int)));
  }
  /* Tester:testOne on device-resident inputs */
  CHECK_MPN(mpn_frcnn_test_one(net, d_im, H, W, d_boxes, N, d_dets[0], cap, d_n[0], NULL));
  /* Tester:test's loop form fed from HOST buffers: three images (the same one), the tail of each overlapping the next trunk */
  for (int it = 0; it < 3; ++it) CHECK_MPN(mpn_frcnn_test_one_pipelined_host(net, h_im, H, W, h_boxes, N, d_dets[1], cap, d_n[1], NULL));
  CHECK_MPN(mpn_frcnn_flush(net, NULL));
  CHECK_HIP(hipDeviceSynchronize());

  int n[2] = {0, 0};
  float *h_dets[2];
  for (int i = 0; i < 2; ++i) {
    CHECK_HIP(hipMemcpy(&n[i], d_n[i], sizeof(int), hipMemcpyDeviceToHost));
    if (n[i] < 0) { fprintf(stderr, "n_dets %d is negative\n", n[i]); return 4; }
    /* *d_n_dets is the UNTRUNCATED survivor count of keep_top_k (ties at the threshold all survive): only min(n, cap) rows were
     * written (include/mpn.h, mpn_frcnn_test_one).  Clamp before reading; a larger count means rows were dropped. */
    if (n[i] > cap) { fprintf(stderr, "warning: %d detections survive the top-k rule, the %d-row buffer holds the first %d\n", n[i], cap, cap); n[i] = cap; }
    h_dets[i] = (float *)malloc((size_t)cap * 6 * sizeof(float));
    CHECK_HIP(hipMemcpy(h_dets[i], d_dets[i], (size_t)n[i] * 6 * sizeof(float), hipMemcpyDeviceToHost));
  }
  if (n[0] != n[1] || memcmp(h_dets[0], h_dets[1], (size_t)n[0] * 6 * sizeof(float)) != 0) {
    fprintf(stderr, "test_one and the pipelined host-fed loop disagree (%d vs %d detections)\n", n[0], n[1]);
    return 4;
  }
  printf("%d detections; first: [%.2f %.2f %.2f %.2f] score %.6f class %d\n", n[0], n[0] ? h_dets[0][0] : 0.f, n[0] ? h_dets[0][1] : 0.f,
         n[0] ? h_dets[0][2] : 0.f, n[0] ? h_dets[0][3] : 0.f, n[0] ? h_dets[0][4] : 0.f, n[0] ? (int)h_dets[0][5] : 0);
  FILE *o = fopen(argv[2], "wb");
  if (!o) { perror(argv[2]); return 1; }
  int32_t n32 = n[0];
  fwrite(&n32, sizeof n32, 1, o);
  fwrite(h_dets[0], sizeof(float), (size_t)n[0] * 6, o);
  fclose(o);
  mpn_frcnn_destroy(net);
  return 0;
}
