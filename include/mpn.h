/* mpn.h — C ABI of libmpn_hip.so: the MI355X (gfx950) per-image detection hot path of MultiPathNet.
 *
 * Drop-in boundary.  Every entry point mirrors ONE reference surface — an
 * nn.Module:updateOutput, a utils.lua helper, or an nms.c export — and cites it.  Plain pointers and
 * sizes only; no torch / TH types.  (The TH-struct-compatible `NMS` / `bbox_vote` pair that
 * utils.lua:15-19 binds lives in include/mpn_libnms.h → libnms.so.)
 *
 * Conventions
 *   - `d_` pointers are DEVICE (HBM) pointers on the current HIP device, fp32, contiguous, row-major,
 *     Torch layout (NCHW images/features, [N,5] rois = {batch(1-based), x1,y1,x2,y2} 1-based pixels).
 *   - `h_` pointers are host pointers.
 *   - `stream` is a hipStream_t passed as void* (NULL = default stream).  Calls enqueue work and
 *     return; they never synchronise unless the name ends in `_host` / `_sync`.
 *   - Return value: MPN_OK or a negative mpn_status; mpn_last_error() gives a thread-local message.
 *     Nothing aborts.  Module-level calls allocate only what a `ws` (workspace) argument documents, plus the shared
 *     scratch named under Concurrency; pipeline handles allocate everything at creation.
 *   - Concurrency: re-entrant.  The reference host runs one worker thread per GPU inside ONE process
 *     (test_runner.lua:55-66: Threads(nGPU) + cutorch.setDevice per thread); this library has no process-global device
 *     state.  A pipeline handle (mpn_frcnn) owns every buffer it uses, including its split-K / NMS scratch, and lives on
 *     the device that was current at creation: drive ONE handle from one thread at a time (with that device current);
 *     different handles — same or different devices — may run concurrently from different host threads and streams.
 *     Module-level calls keep their grow-on-demand scratch per (device, stream) pair, so calls on different streams
 *     or devices never share it either — but ONE thread at a time per (device, stream): two host threads issuing
 *     module-level calls on the same stream (the NULL stream included) would grow and use one scratch concurrently.
 *     A host that creates streams per image releases a stream's scratch with mpn_stream_release before it destroys
 *     the stream (tens of MB of NMS masks per entry otherwise stay until mpn_release_all_scratch / process exit).
 *   - Integer results (argmax, keep indices, counts) are bit-exact vs the reference semantics; fp32
 *     box arithmetic is evaluated without FMA contraction, in the reference's operation order.
 */
#ifndef MPN_H
#define MPN_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MPN_VERSION 600

typedef enum mpn_status {
  MPN_OK = 0,
  MPN_EINVAL = -1,   /* bad argument (null pointer, non-positive size, unsupported shape) */
  MPN_EHIP = -2,     /* a HIP runtime call or kernel launch failed */
  MPN_ENOMEM = -3,   /* workspace too small / allocation failed */
  MPN_ENCCL = -4,    /* an RCCL call failed (mpn_comm_*, mpn_gather_dets) */
  MPN_ESTATE = -5    /* object used in the wrong state */
} mpn_status;

int mpn_version(void);
const char *mpn_last_error(void);
/* Fills name[] with the device's gcnArchName; returns MPN_EHIP when no HIP device is usable. */
int mpn_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes);
/* Frees the module-level scratch kept for (current device, stream) after synchronising that stream; call it before
 * hipStreamDestroy when streams come and go.  mpn_release_all_scratch does it for every (device, stream) entry (the host
 * must have quiesced all module-level work).  Pipeline handles are unaffected: their scratch dies with the handle. */
int mpn_stream_release(void *stream);
int mpn_release_all_scratch(void);

/* ------------------------------------------------------------------------------------------------
 * NMS family — replaces nms.c (the reference's only native code)
 * ---------------------------------------------------------------------------------------------- */

/* nms.c:59-108 `NMS`, batched over classes (Tester_FRCNN.lua:106-125 calls it once per class).
 *   d_scored [n_cls, m_stride, 5] {x1,y1,x2,y2,score}; class c has d_counts[c] valid rows
 *   (d_counts == NULL -> every class has m_stride rows).
 *   d_keep      [n_cls, m_stride, 5]  kept rows in SELECTION order (as nms.c:102-105 writes them)
 *   d_keep_idx  [n_cls, m_stride]     original row index of each kept box (may be NULL)
 *   d_n_keep    [n_cls]               number kept
 * Greedy selection, IoU with the +1 convention (nms.c:14-41), suppression when IoU > thr,
 * tie-breaking among bit-equal scores identical to nms.c:74-98 (swap + stable partition history).
 * Tables of up to 1024 rows per class take ONE launch (sort, suppression mask sliced over the GPU, greedy selection with the exact position
 * rule in one CU's LDS); tables up to MPN_NMS_MAX_BOXES rows a chain of launches (bitonic sort -> suppression bitmask -> wave scan; also
 * what the pipelined mpn_frcnn_* forms run above 384 rows, their NMS sharing the GPU with the next image's trunk);
 * wider tables are accepted too (nms.c has no size limit) and take the exact sweep kernel on HBM-resident arrays. */
#define MPN_NMS_MAX_BOXES 6144
int mpn_nms_batched(const float *d_scored, const int *d_counts, int n_cls, int m_stride, float thr, float *d_keep,
                    int *d_keep_idx, int *d_n_keep, void *stream);

/* Single-class convenience (== utils.nms, utils.lua:29-33) on device buffers. */
int mpn_nms(const float *d_scored, int m, float thr, float *d_keep, int *d_keep_idx, int *d_n_keep, void *stream);

/* utils.nms_dense (utils.lua:402-462, called by demo.lua:85): the index-returning NMS — sort by score (descending), walk the
 * sorted list, a picked box suppresses every box whose IoU with it (areas with the +1 convention, intersection clamped at 0)
 * exceeds `overlap` (strict).  d_boxes [m,5] {x1,y1,x2,y2,score}; d_pick [m] int32 receives the picks as 1-BASED row indices
 * in pick order (the LongTensor the Lua function returns), *d_n_pick their number.  Any m (tables wider than the 8192 rows the
 * LDS sort holds take a counting-rank + sequential-walk form: the reference function has no size limit).  The order among
 * bit-equal scores is the sort's: torch.sort is TH's (unstable) quicksort and TH is absent from the reference tree — PARITY
 * UNPINNED there; this library sorts ties by ascending index, NaN scores last. */
int mpn_nms_dense(const float *d_boxes, int m, float overlap, int *d_pick, int *d_n_pick, void *stream);

/* Host-buffer form used by the libnms.so drop-in, synchronous.  h_keep [m,5].  Tables of up to 1024 rows are staged in a per-thread pinned,
 * device-mapped buffer that the kernel reads and writes directly (no device allocation, no hipMemcpy: one launch + one stream sync);
 * wider ones go H2D -> kernels -> D2H. */
int mpn_nms_host(const float *h_scored, int m, float thr, float *h_keep, int *h_keep_idx, int *n_keep);

/* nms.c:110-142 `bbox_vote`: d_res[i] = score-weighted mean of all scored boxes with IoU(j,i) > thr
 * (strict), accumulated sequentially in j order; d_res[i][4] = nms score.  d_nms [n_nms,5],
 * d_scored [m,5], d_res [n_nms,5].  d_n_nms (device int, may be NULL) overrides n_nms at run time. */
int mpn_bbox_vote(const float *d_nms, int n_nms, const int *d_n_nms, const float *d_scored, int m, float thr,
                  float *d_res, void *stream);
/* The per-class voting loop of Tester_FRCNN.lua:118-124 in one launch: class c votes d_keep[c] (n_keep[c] rows) against
 * d_scored[c] (counts[c] rows) with weights score^score_pow (opt.test_bbox_voting_score_pow; 1 = nms.c arithmetic exactly). */
int mpn_bbox_vote_batched(const float *d_keep, const int *d_n_keep, const float *d_scored, const int *d_counts, int n_cls,
                          int m_stride, float thr, float score_pow, float *d_res, void *stream);
int mpn_bbox_vote_host(const float *h_nms, int n_nms, const float *h_scored, int m, float thr, float *h_res);

/* nms.c:43-56 `boxoverlap`: IoU of n boxes [n,4] against one box (h_b[4], host). */
int mpn_boxoverlap(const float *d_a, int n, const float *h_b, float *d_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Image / ROI preparation — ImageTransformer.lua, ImageDetect.lua
 * ---------------------------------------------------------------------------------------------- */

/* fbcoco.ImageTransformer:updateOutput (modules/ImageTransformer.lua:19-33):
 * out[i] = (in[swap[i]]*scale - mean[i]) / std[i], f64 arithmetic rounded once to fp32
 * (the reference transforms a DoubleTensor, ImageDetect.lua:29,44-50).  swap is 0-based;
 * h_std == NULL skips the division.  d_in, d_out [3,H,W]. */
int mpn_image_transform(const float *d_in, int H, int W, const int *h_swap, double scale, const double *h_mean,
                        const double *h_std, float *d_out, void *stream);

/* image.scale(src, W2, H2) in its default bilinear mode (external `image` rock; ImageDetect.lua:41).  PARITY
 * UNPINNED: restated from torch/image's published separable algorithm — rows then columns through a float
 * intermediate; upscaling interpolates with i+f = d*(src-1)/(dst-1) (last sample copied), downscaling is a
 * fractional box average over [d*s,(d+1)*s), s = src/dst.  d_in [C,H,W] -> d_out [C,H2,W2]; d_tmp holds C*H*W2 floats.
 * ImageDetect.lua:40 sizes the output as (long)(H*scale) x (long)(W*scale). */
int mpn_image_scale(const float *d_in, int C, int H, int W, int H2, int W2, float *d_tmp, float *d_out, void *stream);

/* ImageDetect.lua:34-43: scale factor for one image (host arithmetic). */
double mpn_pick_scale(int H, int W, double target, double max_size);

/* project_im_rois (ImageDetect.lua:66-70): rois = {1, (boxes-1)*s+1}.  d_boxes [n,4] -> d_rois [n,5]. */
int mpn_project_im_rois(const float *d_boxes, int n, double scale, float *d_rois, void *stream);

/* ------------------------------------------------------------------------------------------------
 * nn.Module:updateOutput mirrors
 * ---------------------------------------------------------------------------------------------- */

/* inn.ROIPooling(PW,PH,scale):updateOutput{feat, rois} (vgg.lua:28, alexnet.lua:23, resnet.lua:48,
 * inceptionv3.lua:41, model_utils.lua:215).  d_feat [B,C,H,W], d_rois [N,5] -> d_out [N,C,PH,PW],
 * d_argmax [N,C,PH,PW] int32 (h*W+w within the plane, -1 for an empty bin; may be NULL).
 * start=round((x1-coord_offset)*scale), end=round((x2-coord_offset)*scale)+end_adjust, ROI forced
 * >= 1x1, bins floor/ceil of fp32 bin size, clipped to the map, empty bin -> 0.
 * Reference default ("v2" fix, README.md:202-203): coord_offset=1, end_adjust=0. */
int mpn_roi_pool_forward(const float *d_feat, int B, int C, int H, int W, const float *d_rois, int N, int PH, int PW,
                         float scale, float coord_offset, int end_adjust, float *d_out, int32_t *d_argmax,
                         void *stream);
/* The same module with an explicit BIN RULE (inn.ROIPooling is external and its source is not in the reference tree: both of its branches are
 * restated, SURVEY §8a-6, parity unpinned):
 *   MPN_ROI_BINS_CAFFE    — the CUDA branch (what mpn_roi_pool_forward computes): bins from the un-clipped window, each bin clipped to the map
 *                           afterwards, a bin that falls outside the map is empty (0, argmax -1);
 *   MPN_ROI_BINS_ADAPTIVE — the CPU branch ("CPU nn path", BASELINE configs[0]; call sites models/alexnet.lua:23, models/vgg.lua:28 with float
 *                           tensors): corners = round((x - coord_offset) * scale + 1) in 1-based map coordinates, CLIPPED to the map first
 *                           (the reference clips the upper side — cmin — and indexes the tensor with the lower side; we clip both), the crop
 *                           goes through nn.SpatialAdaptiveMaxPooling(PW, PH): bin p = [floor(p * size / P), ceil((p + 1) * size / P)) of the
 *                           clipped crop, never empty.  The two rules differ on windows whose rounded corners leave the map (Foveal's regions; border boxes whose
 *                           round((x2 - 1) * scale) is W) and, about once in a thousand windows, by one cell where the fp32 bin size rounds
 *                           (measured: tests/test_oracle_roipool_adaptive.py).
 * d_argmax is h * W + w in the feature plane under both rules. */
#define MPN_ROI_BINS_CAFFE 0
#define MPN_ROI_BINS_ADAPTIVE 1
int mpn_roi_pool_forward_rule(const float *d_feat, int B, int C, int H, int W, const float *d_rois, int N, int PH, int PW,
                              float scale, float coord_offset, int end_adjust, int bin_rule, float *d_out, int32_t *d_argmax,
                              void *stream);

/* nn.Foveal:updateOutput (modules/Foveal.lua:15-44): [N,5] -> [4N,5], f64 arithmetic rounded to fp32. */
int mpn_foveal_forward(const float *d_rois, int N, float *d_out, void *stream);

/* nn.ContextRegion(scale):updateOutput (modules/ContextRegion.lua:14-32): [N,5] -> [N,5]. */
int mpn_context_region_forward(const float *d_rois, int N, double scale, float *d_out, void *stream);

/* nn.BBoxNorm:updateOutput, evaluate mode (modules/BBoxNorm.lua:18-32): in place, view(-1,4)*std+mean. */
int mpn_bbox_norm_forward(float *d_bbox, int N, int C4, const float *h_mean4, const float *h_std4, void *stream);

/* nn.SelectBoxes:updateOutput (modules/SelectBoxes.lua:26-56): first arg-max class -> its 4 coords. */
int mpn_select_boxes_forward(const float *d_scores, const float *d_bbox, int N, int C, float *d_out, void *stream);

/* nn.SoftMax:updateOutput over dim 2 (ImageDetect.lua:19,189-191).  [M,C] -> [M,C]. */
int mpn_softmax_forward(const float *d_x, int M, int C, float *d_y, void *stream);

/* cudnn.SpatialConvolution(Cin,Cout,3,3,1,1,1,1):updateOutput (+ fused nn.ReLU) as in `features`
 * (models/vgg.lua:15,25; multipathnet.lua:34-46).  NCHW in/out, d_w [Cout,Cin,3,3], d_b [Cout] or NULL.
 * fp32 MFMA implicit GEMM.  The activations are converted to the library's channel-blocked HBM layout
 * inside `ws`; mpn_conv3x3_workspace_bytes gives the size needed. */
size_t mpn_conv3x3_workspace_bytes(int B, int Cin, int H, int W, int Cout);
int mpn_conv3x3_forward(const float *d_in, int B, int Cin, int H, int W, const float *d_w, const float *d_b,
                        int Cout, int relu, float *d_out, void *d_ws, size_t ws_bytes, void *stream);

/* nn.SpatialMaxPooling(2,2,2,2):ceil():updateOutput.  [B*C,H,W] -> [B*C,ceil(H/2),ceil(W/2)]. */
int mpn_maxpool2x2_ceil_forward(const float *d_in, int BC, int H, int W, float *d_out, void *stream);

/* nn.Linear(K,N):updateOutput (+ fused nn.ReLU) (vgg.lua:16,30 `top`; model_utils.lua:105-119).
 * y[M,N] = x[M,K] W[N,K]^T + b.  fp32 MFMA; per-output k-ascending fmaf chain, independent of M
 * (so chunked == un-chunked exactly, test.lua:140-179). */
int mpn_linear_forward(const float *d_x, int M, int K, const float *d_w, const float *d_b, int N, int relu,
                       float *d_y, void *stream);

/* ------------------------------------------------------------------------------------------------
 * utils.lua / Tester_FRCNN.lua helpers
 * ---------------------------------------------------------------------------------------------- */

/* utils.convertFrom, 2-D path, for every 4-column class block (utils.lua:229-247,
 * ImageDetect.lua:183-185).  d_boxes [N,4] original-image boxes, d_deltas [N,4C] -> d_out [N,4C]. */
int mpn_bbox_decode(const float *d_boxes, const float *d_deltas, int N, int C, float *d_out, void *stream);

/* Tester_FRCNN.lua:75-78: in place, x -> [1,im_w], y -> [1,im_h] on the (x,y) pairs of d_bbox. */
int mpn_clamp_boxes(float *d_bbox, size_t n_pairs, float im_w, float im_h, void *stream);

/* Tester_FRCNN.lua:106-116 for all classes j = first_cls .. C-1 at once: rows with score > thresh,
 * in row order -> d_scored [C-first_cls, N, 5], d_counts [C-first_cls], d_src_idx (may be NULL). */
int mpn_select_scored(const float *d_scores, const float *d_bbox, int N, int C, int first_cls, float thresh,
                      float *d_scored, int *d_counts, int *d_src_idx, void *stream);

/* utils.keep_top_k (utils.lua:75-96): threshold = k-th largest kept score over all classes (ties
 * survive).  d_keep [n_cls, m_stride, 5] / d_n_keep [n_cls] as written by mpn_nms_batched.
 * Writes *d_thresh (device float) and compacts survivors into d_out [max_out, 6] =
 * {x1,y1,x2,y2,score,class(1-based, as float)} class-major.  *d_n_out receives the UNTRUNCATED number of survivors:
 * when it exceeds max_out (many ties at the threshold), only the first max_out rows were written and the caller
 * knows rows were dropped (utils.keep_top_k itself never truncates). */
int mpn_keep_top_k(const float *d_keep, const int *d_n_keep, int n_cls, int m_stride, int k, float *d_thresh,
                   float *d_out, int max_out, int *d_n_out, void *stream);
/* The same on tables whose rows are in NON-INCREASING score order inside every class — what mpn_nms_batched emits (each greedy round picks the
 * largest remaining score, nms.c:74-81) and what mpn_bbox_vote_batched keeps (voted rows carry their NMS scores, nms.c:139).  Only the first
 * min(k, n_c) rows of each class can reach the k-th largest score and a class's survivors are a prefix of it, so the kernel stages
 * n_cls * k keys instead of every kept row.  Same outputs, bit for bit, as mpn_keep_top_k ON SUCH TABLES; on unsorted tables use
 * mpn_keep_top_k.  The test_one forms of the pipeline call this one. */
int mpn_keep_top_k_sorted(const float *d_keep, const int *d_n_keep, int n_cls, int m_stride, int k, float *d_thresh,
                          float *d_out, int max_out, int *d_n_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Wire formats either side of the path
 * ---------------------------------------------------------------------------------------------- */

/* testCoco/init.lua:65-85: {x1,y1,x2,y2,score,class} rows (mpn_keep_top_k output) -> COCO result rows
 * {image_id, x1-1, y1-1, x2-x1, y2-y1, score, category_id}; d_cat_ids [n_cat] maps class (1-based) to the
 * dataset's category id (NULL: the class index itself).  d_n_dets (device int, may be NULL) bounds max_n. */
int mpn_dets_to_coco_rows(const float *d_dets, const int *d_n_dets, int max_n, float image_id, const float *d_cat_ids,
                          int n_cat, float *d_rows, void *stream);

/* DataSetJSON.lua:157-239 (loadROIDB): proposal rows come as {y1,x1,y2,x2}; emits the {2,1,4,3}-permuted
 * {x1,y1,x2,y2} rows and, in d_keep (may be NULL), 1 for rows whose area (x2-x1)*(y2-y1) > min_area
 * (min_area == 0 keeps everything — filterArea, DataSetJSON.lua:172-186). */
int mpn_proposals_permute_filter(const float *d_in, int n, float min_area, float *d_out, int *d_keep, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Fused per-image pipeline: Tester_FRCNN:testOne -> ImageDetect:detect -> model:forward -> NMS
 * (Tester_FRCNN.lua:54-139, ImageDetect.lua:156-193, models/vgg.lua:23-31)
 * ---------------------------------------------------------------------------------------------- */

typedef struct mpn_frcnn_config {
  int n_conv;              /* number of 3x3 conv layers in the trunk (VGG-16: 13) */
  const int *conv_cout;    /* [n_conv] output channels */
  const int *pool_after;   /* [n_conv] 1 = ceil-mode 2x2 max-pool after this layer's ReLU */
  int pooled_h, pooled_w;  /* ROIPooling bins (VGG: 7,7) */
  float spatial_scale;     /* VGG: 1/16 */
  int fc_dim;              /* 4096 */
  int n_classes;           /* C incl. background (VOC: 21) */
  int max_h, max_w;        /* largest input image (600 x 1000) */
  int max_rois;            /* largest ROI batch (1000) */
  double tf_scale;         /* ImageTransformer: scale, mean[3], std[3] (std[0]==0 -> none), swap[3] */
  double tf_mean[3];
  double tf_std[3];
  int tf_swap[3];
  float bbox_mean[4];      /* BBoxNorm (std[0]==0 -> module absent) */
  float bbox_std[4];
  float nms_thresh;        /* 0.3 */
  float score_thresh;      /* -1.5 (Tester_FRCNN.lua:50) */
  int top_k;               /* 100 (Tester_FRCNN.lua:163) */
  /* accuracy knobs of Tester_FRCNN (all off in the reference's default config): */
  int num_iter;            /* opt.test_num_iterative_loc (1 = off; i = 2..num_iter: SelectBoxes + a head-only pass on the cached
                              trunk features; num_iter * max_rois <= MPN_NMS_MAX_BOXES) */
  int bbox_voting;         /* opt.test_bbox_voting */
  float bbox_vote_thresh;  /* opt.test_bbox_voting_nms_threshold (0.5).  The reference passes an unset field here
                              (Tester_FRCNN.lua:123 vs :29); we use the configured threshold. */
  float bbox_vote_score_pow; /* opt.test_bbox_voting_score_pow (1) */
  /* getImages (ImageDetect.lua:34-43): 0 = feed the image as it is; otherwise rescale so that the short side is
   * scale_target (600), capped so that the long side stays <= scale_max (1000).  max_h / max_w bound the RESCALED image. */
  double scale_target, scale_max;
  int use_rbox_scores;     /* opt.test_use_rbox_scores (Tester_FRCNN.lua:91-97): needs num_iter > 1; the scores of pass i+1 are
                              paired with the boxes of pass i (the first score table and the last box table are dropped), so
                              (num_iter - 1) * N rows reach the NMS */
  int roi_bin_rule;        /* MPN_ROI_BINS_CAFFE (0, default: inn.ROIPooling's CUDA branch) | MPN_ROI_BINS_ADAPTIVE (its CPU branch): the
                              rule of every ROI pooling of the pipeline — mpn_roi_pool_forward_rule.  (added in MPN_VERSION 600, at the END) */
  int fc_arith;            /* MPN_FC_FP32 (0, default): fc6 on the fp32 matrix pipe (v_mfma_f32_32x32x2_f32).  MPN_FC_SPLIT3 (1; mpn_frcnn_create / mpn_mpnet_create): fc6 and fc7
                              on the bf16 pipe with fp32-level results — both operands split exactly into three bf16 planes (h + m + l = the fp32 value),
                              the six plane products of weight >= 2^-16 accumulated in fp32, the three <= 2^-24 ones (the size of an fp32 product's own
                              rounding) dropped.  An AUXILIARY arithmetic: the headline and every parity claim are MPN_FC_FP32.  (MPN_VERSION 600) */
} mpn_frcnn_config;
#define MPN_FC_FP32 0
#define MPN_FC_SPLIT3 1

typedef struct mpn_frcnn mpn_frcnn; /* opaque */

/* Builds the pipeline on the current device: allocates activations/workspace once, re-packs weights
 * into MFMA-fragment order in HBM.  Weight pointers are DEVICE pointers in Torch layout
 * (conv [Cout,Cin,3,3]; linear [out,in]); they are read during this call only.
 * d_conv_w/d_conv_b: arrays (host) of n_conv device pointers. */
int mpn_frcnn_create(const mpn_frcnn_config *cfg, const float *const *d_conv_w, const float *const *d_conv_b,
                     const float *d_fc6_w, const float *d_fc6_b, const float *d_fc7_w, const float *d_fc7_b,
                     const float *d_cls_w, const float *d_cls_b, const float *d_bbox_w, const float *d_bbox_b,
                     mpn_frcnn **out);
void mpn_frcnn_destroy(mpn_frcnn *p);

/* MultiPathNet head (models/multipathnet.lua:64-120): nn.Foveal -> n_towers region towers, each
 * {conv345Combine (model_utils.lua:209-251: ROI pools of conv5 @scale, conv4 @2*scale, conv3 @4*scale,
 * per-map nn.Normalize(2), channel concat, *1000, 1x1 conv mix to conv5's width) -> fc6 -> fc7};
 * the first n_towers-1 ("foveal") towers are concatenated for the K integral classifiers
 * (model_utils.lua:275-317: eval = mean of the K softmaxes), the last ("het") tower feeds the box
 * regressor.  This replaces nn.ModelParallelTable's broadcast/concat (ModelParallelTable.lua:195-242)
 * with zero-copy concatenation on one GPU.  All pointers are device pointers in Torch layout:
 * mix_w[t] [c5, total_feat_t] (the 1x1 conv), fc6_w[t] [fc, c5*PH*PW], fc7_w[t] [fc, fc];
 * d_cls_w [n_integral*C, (n_towers-1)*fc] (the K classifiers stacked), d_bbox_w [4C, fc]. */
typedef struct mpn_mpnet_weights {
  int n_towers;              /* 5 in the reference: regions {0,1,2,3} + het on region 1 */
  int region[8];             /* 0-based Foveal region of each tower (Foveal.lua:36-39 rows) */
  int use_conv4[8], use_conv3[8];
  int tap_conv3, tap_conv4;  /* 0-based conv-layer indices whose pre-pool outputs are "conv3"/"conv4" (VGG-16: 6, 9) */
  int n_integral;            /* K (opt.nDonkeys in the reference, 6 in scripts/train_multipathnet_coco.sh:8) */
  const float *mix_w[8], *mix_b[8], *fc6_w[8], *fc6_b[8], *fc7_w[8], *fc7_b[8];
  int conv345_unnormalized;  /* 0 (default, opt.model_conv345_norm = true): per-map nn.Normalize(2) then MulConstant(1000);
                              * 1: the isNormalized = false branch, MulConstant(1), (1/30), (1/200) on conv5 / conv4 / conv3 and
                              * no x1000 (model_utils.lua:214-223,231-244).  (added in MPN_VERSION 201, at the END of the struct) */
} mpn_mpnet_weights;
int mpn_mpnet_create(const mpn_frcnn_config *cfg, const float *const *d_conv_w, const float *const *d_conv_b,
                     const mpn_mpnet_weights *mw, const float *d_cls_w, const float *d_cls_b, const float *d_bbox_w,
                     const float *d_bbox_b, mpn_frcnn **out);

/* ResNet Fast R-CNN (models/resnet.lua:24-50): net:get(1..7) = conv1 7x7/2, BN, ReLU, max-pool 3x3/2 pad 1,
 * layer1-3 run on the image; inn.ROIPooling(14,14,1/16); net:get(8..10) = layer4 + 7x7 average pool + View run per ROI;
 * classAndBBoxLinear on the pooled vector.  The fb.resnet.torch `.t7` is not in the tree: the caller describes the graph —
 * every convolution in execution order (conv1, then per residual block its convolutions followed by its shortcut
 * convolution if it has one), BatchNorm already folded into w / b (inn.utils.BNtoFixed, resnet.lua:34-36).  A block is
 * relu(conv_n(... relu(conv_1(x)) ...) + shortcut(x)).  Uses cfg->{max_h,max_w,max_rois,n_classes,pooled_h,spatial_scale,
 * tf_*,bbox_*,nms_*,score_thresh,top_k,num_iter,bbox_voting,...}; cfg->n_conv / conv_cout / pool_after / fc_dim are ignored.
 * All pointers are device pointers; fp32. */
typedef struct mpn_resnet_weights {
  int n_convs;
  const float *const *w;       /* [n_convs] -> [cout, cin, k, k] */
  const float *const *b;       /* [n_convs] -> [cout] (may be NULL = no biases) */
  const int *cin, *cout, *ksize, *stride, *pad;   /* [n_convs] */
  int n_blocks;
  const int *block_n_convs;       /* [n_blocks] convolutions on the residual path */
  const int *block_has_shortcut;  /* [n_blocks] 1 = a shortcut convolution follows the block's convolutions in the list */
  int n_trunk_blocks;             /* blocks [0, n_trunk_blocks) = layer1-3 (image trunk); the rest = layer4 (per-ROI head) */
  /* MultiPathNet on a ResNet backbone (BASELINE configs[3]).  The reference tree has NO such model: this is
   * this library's extension, shaped like models/multipathnet.lua:64-120 — nn.Foveal regions over the single stride-16 map, one
   * layer4 copy ("tower") per region, the classification towers' pooled vectors concatenated for n_integral classifier clones
   * (mean of their softmaxes, model_utils.lua:296-315), the LAST tower feeding the box regressor.  n_heads <= 1: plain
   * resnet.lua.  n_heads >= 2: the blocks after the trunk are n_heads equal groups in tower order; d_cls_w is
   * [n_integral*C, (n_heads-1)*out_c], d_bbox_w [4C, out_c]. */
  int n_heads;
  int head_region[8];             /* 0-based Foveal region of each tower */
  int n_integral;
  int bf16;                       /* 1: trunk and per-ROI convolutions in bf16 (bf16 activations and weights, fp32 accumulate; bias, residual,
                                     ReLU in fp32; the cls / bbox head stays fp32) — the dtype SURVEY §8f rank 3 asks for.  0: fp32 */
} mpn_resnet_weights;
int mpn_resnet_create(const mpn_frcnn_config *cfg, const mpn_resnet_weights *rw, const float *d_cls_w, const float *d_cls_b,
                      const float *d_bbox_w, const float *d_bbox_b, mpn_frcnn **out);

/* Branching conv graphs (models/inceptionv3.lua:27-43, BASELINE configs[4]; models/alexnet.lua:14-27, BASELINE configs[0], whose
 * `top` — fc6 / fc7 on the flattened 6x6 ROI-pooled map — is a 6x6 and a 1x1 convolution in the head list): the trunk (net:get(1..25)) and the per-ROI
 * classifier (net:get(26..30)) as two op lists over numbered tensors.  Tensor 0 of the trunk list is the transformed image
 * (3 channels); tensor 0 of the head list is the ROI-pooled map (cfg->pooled_h x pooled_w, channels of `feat_tensor`); the head's
 * `out_tensor` is averaged over its whole map (the graph's final SpatialAveragePooling + View) and feeds classAndBBoxLinear.
 * An op reads ALL channels of `src` and writes `cout` (conv) / src's (pools) channels of `dst` starting at channel
 * `dst_c_off` — an Inception module's DepthConcat is its branches writing side by side into one tensor (offsets and widths
 * must be multiples of 8; of 16 with bf16).  BatchNorm folded into w / b by the caller (utils.BNtoFixed, inceptionv3.lua:23).
 * The `.t7` (Moodstocks' conversion of Google's Inception-v3) is not in the tree: PARITY UNPINNED, structure from the public
 * definition. */
typedef struct mpn_graph_op {
  int kind;               /* 0 = convolution (+ bias, ReLU if relu), 1 = max-pool (padded cells never win; floor mode unless ceil_mode),
                             2 = average pool, count_include_pad (nn.SpatialAveragePooling's default: always / (kh*kw)),
                             3 = cross-channel LRN, nn.SpatialCrossMapLRN(size = kh, lrn_alpha, lrn_beta, lrn_k) (alexnet.lua's trunk):
                                 out_c = in_c * (lrn_k + lrn_alpha / size * sum_{|c' - c| <= (size-1)/2} in_c'^2) ^ -lrn_beta */
  int src, dst, dst_c_off;
  int cin, cout;          /* conv: weight shape [cout, cin, kh, kw]; pools: cin = channels of src */
  int kh, kw, sh, sw, ph, pw;
  int relu;
  const float *w, *b;     /* conv only, device pointers */
  int src_c_off;          /* the op reads channels [src_c_off, src_c_off + cin) of `src` (a multiple of 8; 0 = from the first): a grouped
                             convolution (alexnet.lua's conv2 / conv4 / conv5, groups = 2) is one op per group */
  int ceil_mode;          /* max-pool: nn.SpatialMaxPooling(...):ceil() / Caffe output size — ceil((H + 2p - k) / s) + 1, minus one when the
                             last window would start beyond the padded input */
  float lrn_alpha, lrn_beta, lrn_k;   /* kind 3 */
} mpn_graph_op;
typedef struct mpn_graph_weights {
  int n_trunk_ops;  const mpn_graph_op *trunk_ops;  int n_trunk_tensors;  const int *trunk_tensor_c;  int feat_tensor;
  int n_head_ops;   const mpn_graph_op *head_ops;   int n_head_tensors;   const int *head_tensor_c;   int out_tensor;
  int bf16;               /* as mpn_resnet_weights.bf16 */
  /* MultiPathNet towers on this backbone (BASELINE configs[4]; this library's extension, see mpn_resnet_weights.n_heads):
   * n_heads >= 2: head_ops holds n_heads * n_head_ops entries, tower-major, every tower the same op structure with its own
   * weights; tower t pools Foveal region head_region[t]; the last tower feeds the box regressor. */
  int n_heads;
  int head_region[8];
  int n_integral;
} mpn_graph_weights;
int mpn_graph_create(const mpn_frcnn_config *cfg, const mpn_graph_weights *gw, const float *d_cls_w, const float *d_cls_b,
                     const float *d_bbox_w, const float *d_bbox_b, mpn_frcnn **out);

/* ImageDetect:detect (ImageDetect.lua:156-193; getImages' rescaling per cfg.scale_target):
 * d_image [3,H,W] fp32 in [0,1]; d_boxes [N,4].  Outputs (all optional, device):
 *   d_scores [N,C] softmax, d_bbox [N,4C] decoded boxes.  clamp = 0 returns them as ImageDetect:detect does (unclamped);
 *   clamp = 1 additionally applies Tester_FRCNN.lua:75-78's clamp to the image, which the reference applies to the FIRST
 *   detect() of testOne only (passes 2..num_iter of iterative localisation stay unclamped, Tester_FRCNN.lua:82-89).
 * d_image == NULL means recompute_features = false (ImageDetect.lua:107-111): the trunk output of the previous
 * call on this handle is reused and only the ROI head runs on the new boxes (iterative localisation,
 * Tester_FRCNN.lua:82-89); H, W must equal the cached image's size. */
int mpn_frcnn_detect(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, float *d_scores,
                     float *d_bbox, int clamp, void *stream);

/* Tester:testOne + keep_top_k: detect, per-class NMS, global top-k.
 *   d_dets [top_cap,6] {x1,y1,x2,y2,score,class}, *d_n_dets; raw per-class NMS results stay readable
 *   through mpn_frcnn_nms_results until the next call.
 * CONTRACT of *d_n_dets (all test_one forms, as mpn_keep_top_k's *d_n_out): the UNTRUNCATED number of survivors of the
 * top-k rule.  utils.keep_top_k keeps every row tied at the threshold, so it can exceed top_k and even top_cap; only
 * min(*d_n_dets, top_cap) rows of d_dets were written.  A caller must clamp before it reads rows — copy
 * min(n, top_cap) * 6 floats — and may treat n > top_cap as "rows were dropped, enlarge top_cap". */
int mpn_frcnn_test_one(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, float *d_dets,
                       int top_cap, int *d_n_dets, void *stream);
/* Throughput form for a loop over images (Tester:test, Tester_FRCNN.lua:150-157): same work, but the
 * latency-bound NMS + top-k tail of image i runs on an internal side stream (default priority) and overlaps image
 * i+1's trunk.  d_dets / d_n_dets of call i are ordered on `stream` only after call i+1 (on the same
 * handle) or mpn_frcnn_flush(); alternate two output buffers between consecutive calls.  For the plain Fast R-CNN head
 * (one localisation pass) the class / box GEMM, softmax, decode and select of image i run on that internal stream as well
 * (51 us of kernels that leave most of the GPU idle, now under image i+1's first layers); what they read is kept per
 * buffer set inside the handle — d_image and d_boxes are consumed on `stream` before the call's work there ends, exactly as
 * in the un-pipelined form. */
int mpn_frcnn_test_one_pipelined(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N,
                                 float *d_dets, int top_cap, int *d_n_dets, void *stream);
int mpn_frcnn_flush(mpn_frcnn *p, void *stream);
/* The same throughput form fed from HOST buffers, as the reference's loop is (Tester_FRCNN.lua:64-66 gets a CPU image and
 * CPU boxes; ImageDetect.lua:148-151 copies them to the GPU): the upload of image i (7.2 MB + 16 KB at 600x1000 / 1000 ROIs)
 * runs on the handle's copy stream into one of three handle-owned staging sets and overlaps image i-1's kernels; `stream`
 * waits for it only where the trunk starts.  A staging set is reused once the image that filled it three calls earlier has
 * been consumed: the call waits for that on the HOST (a copy that depends on a compute-queue event leaves the DMA engines),
 * so the host runs at most three images ahead of the device.  h_image / h_boxes should be pinned (hipHostMalloc / hipHostRegister /
 * torch pin_memory) — pageable memory works but serialises the copy; they may be reused as soon as the call returns
 * only if pinned memory is NOT rewritten before the copy ran: alternate two host buffers like the output buffers. */
int mpn_frcnn_test_one_pipelined_host(mpn_frcnn *p, const float *h_image, int H, int W, const float *h_boxes, int N,
                                      float *d_dets, int top_cap, int *d_n_dets, void *stream);
int mpn_frcnn_nms_results(mpn_frcnn *p, const float **d_keep, const int **d_keep_idx, const int **d_n_keep,
                          int *m_stride);
/* ------------------------------------------------------------------------------------------------
 * Multi-GPU: images shard across GPUs, the only exchange is an RCCL all-gather of SCORED BOXES
 * (replaces test_runner.lua:91-104's result hand-back and ModelParallelTable.lua:204-236's feature broadcast)
 * ---------------------------------------------------------------------------------------------- */
typedef struct mpn_comm mpn_comm; /* opaque: one RCCL communicator rank, bound to the device current at creation */
#define MPN_UNIQUE_ID_BYTES 128
/* One process per GPU: rank 0 calls mpn_comm_get_unique_id and hands the 128 bytes to the other ranks out of band
 * (a file, an env var, torch.distributed's store ...); every rank then calls mpn_comm_init_rank (collective).
 * world == 1 with id128 == NULL never loads RCCL (the gather degenerates to the pack kernel); world == 1 with an id
 * creates a real one-rank communicator. */
int mpn_comm_get_unique_id(void *id128);
int mpn_comm_init_rank(const void *id128, int world, int rank, mpn_comm **out);
/* The reference's process model — ONE process, one worker thread per GPU (test_runner.lua:55-66): creates the n_dev
 * communicators at once; worker i uses out[i] with device h_devices[i] (NULL: device i) current. */
int mpn_comm_init_all(int n_dev, const int *h_devices, mpn_comm **out);
int mpn_comm_world(const mpn_comm *c);
int mpn_comm_rank(const mpn_comm *c);
/* What RCCL ITSELF reports for the communicator (ncclCommCount), 0 when there is none (world 1 without an id), -1 when the loaded RCCL
 * has no ncclCommCount symbol (nothing was cross-checked: UNVERIFIED, never the caller's own world size).  Both init entries
 * cross-check ncclCommCount / ncclCommUserRank against the caller's (world, rank) and fail with MPN_ENCCL on a mismatch, and
 * mpn_comm_init_rank waits a bounded time for its peers (MPN_COMM_INIT_TIMEOUT_S, default 120 s) instead of hanging: a multi-GPU
 * run can state — and a bench line can carry — how many ranks RCCL really saw (test_runner.lua:55-66: worker k IS GPU k). */
int mpn_comm_rccl_ranks(const mpn_comm *c);
void mpn_comm_destroy(mpn_comm *c);
/* A rank's record for one image: top_cap rows {x1,y1,x2,y2,score,class} (rows at / beyond the count zeroed) + the count
 * as a float = top_cap*6 + 1 floats (~10 KB for top_cap = 464). */
size_t mpn_det_record_floats(int top_cap);
int mpn_pack_det_record(const float *d_dets, const int *d_n_dets, int top_cap, float *d_rec, void *stream);
/* Packs this rank's (d_dets, *d_n_dets) — mpn_frcnn_test_one's outputs — and all-gathers the records of all ranks into
 * d_out [world, top_cap*6 + 1]; record r belongs to the image rank r processed in this step.  Stream-ordered on
 * `stream`, no host synchronisation; every rank must call it the same number of times. */
int mpn_gather_dets(mpn_comm *c, const float *d_dets, const int *d_n_dets, int top_cap, float *d_out, void *stream);
/* Generic fixed-size all-gather of float records: every rank contributes n_floats from d_send, d_out is [world, n_floats]
 * (rank order).  Stream-ordered, no host synchronisation; without an RCCL communicator (world 1) a device copy.  The calling
 * thread's current device must be the communicator's (MPN_ESTATE otherwise; mpn_gather_dets checks the same). */
int mpn_gather_rows(mpn_comm *c, const float *d_send, size_t n_floats, float *d_out, void *stream);

/* ------------------------------------------------------------------------------------------------
 * Proposal (ROI) sharding of ONE image across the GPUs of a node — the latency mode (north_star "images+proposals
 * shard across the 8 GPUs").  Replaces ModelParallelTable.lua:195-242 (broadcast the input to every tower GPU, run, copy back,
 * concatenate) for a single image: every rank holds the full model and is handed the SAME image and the SAME proposal table;
 * it runs the trunk, the ROI head on its contiguous slice of the proposals (mpn_shard_range(N, world, rank)), the per-class NMS
 * on its contiguous slice of the foreground classes (mpn_shard_range(C-1, world, rank)), and two all-gathers of scored boxes
 * connect the three steps.  Rows and classes are independent, so d_dets / d_n_dets and mpn_frcnn_nms_results equal the
 * unsharded mpn_frcnn_test_one's bit for bit on every rank; iterative localisation and box voting are supported (each rank
 * refines its own rows).  The partition contract is test_runner.lua:91-104's (every worker the same code, a disjoint share).
 *   mpn_frcnn_shard_head   trunk + head on this rank's proposals -> d_rows_rec [mpn_frcnn_shard_rows_floats]
 *   (exchange)             d_rows_all [world, rows_floats] = all ranks' records in rank order (mpn_gather_rows or any transport)
 *   mpn_frcnn_shard_nms    joined tables of the whole image (kept on the handle: what detect() returns), select, NMS (+ vote)
 *                          of this rank's classes -> d_class_rec [mpn_frcnn_shard_class_floats]
 *   (exchange)             d_class_all [world, class_floats]
 *   mpn_frcnn_shard_finish every class's kept table on the handle (mpn_frcnn_nms_results) + keep_top_k -> d_dets, *d_n_dets
 * mpn_frcnn_test_one_sharded chains the five steps over an mpn_comm on `stream` (no host synchronisation in steady state).
 * The same image size / N / world must be passed to all steps; a rank that owns no proposals (world > N) writes a zero record. */
int mpn_shard_range(int n, int world, int rank, int *lo, int *hi); /* balanced contiguous: the first n % world ranks own one more */
size_t mpn_frcnn_shard_rows_floats(const mpn_frcnn *p, int N, int world);
size_t mpn_frcnn_shard_class_floats(const mpn_frcnn *p, int N, int world);
int mpn_frcnn_shard_head(mpn_frcnn *p, const float *d_image, int H, int W, const float *d_boxes, int N, int rank, int world,
                         float *d_rows_rec, void *stream);
int mpn_frcnn_shard_nms(mpn_frcnn *p, const float *d_rows_all, int N, int rank, int world, float *d_class_rec, void *stream);
int mpn_frcnn_shard_finish(mpn_frcnn *p, const float *d_class_all, int N, int world, float *d_dets, int top_cap, int *d_n_dets,
                           void *stream);
int mpn_frcnn_test_one_sharded(mpn_frcnn *p, mpn_comm *comm, const float *d_image, int H, int W, const float *d_boxes, int N,
                               float *d_dets, int top_cap, int *d_n_dets, void *stream);

/* Captured launch graphs.  The kernel chains of the per-image path — the head (transform .. decode) and the tail (per-class NMS,
 * voting, top-k) of mpn_frcnn_test_one / _pipelined / _pipelined_host, and the bodies of mpn_frcnn_shard_head / _shard_nms /
 * _shard_finish — are captured with hipStreamBeginCapture once per (buffer pointers, H, W, N) and replayed with hipGraphLaunch: one host
 * call per segment instead of 30-60 kernel launches (Tester_FRCNN.lua:54-139 calls testOne in a loop over same-sized inputs).  Results are
 * bit-identical to the ordinary launches.  Caller-provided buffers are captured at their second sighting, so a host that passes the
 * same device buffers every image gets the replays and one that allocates fresh ones never pays for a capture; a graph is dropped when a
 * library buffer it references is replaced.  OFF by default — opt in with enable = 1 here or MPN_GRAPHS=1 in the environment: measured
 * on MI355X the host's enqueue time per image drops 6.5x (AlexNet 484 -> 74 us) while the device-side timeline, and so the proposals/s
 * of a device-bound loop, does not improve (-0.5 %): it frees the host thread, it does not speed up the GPU.  Profiling
 * (mpn_frcnn_set_profiling) suspends it. */
int mpn_frcnn_set_graphs(mpn_frcnn *p, int enable);
int mpn_frcnn_graph_stats(const mpn_frcnn *p, long *captures, long *replays);

/* Per-kernel-group timing with HIP events recorded on the launch stream (bench.py's roofline leg).
 * Tags index the arrays returned by mpn_frcnn_get_profile (accumulated ms and launch-group counts). */
enum {
  MPN_PROF_TRANSFORM = 0, MPN_PROF_CONV_WINO /* Winograd F(2x2,3x3) layers */, MPN_PROF_CONV_DIRECT /* direct implicit-GEMM layers */, MPN_PROF_POOL, MPN_PROF_ROIPOOL, MPN_PROF_FC6,
  MPN_PROF_FC7, MPN_PROF_HEADS, MPN_PROF_POST, MPN_PROF_SELECT, MPN_PROF_NMS, MPN_PROF_TOPK, MPN_PROF_NTAGS
};
int mpn_frcnn_set_profiling(mpn_frcnn *p, int enable);
int mpn_frcnn_get_profile(mpn_frcnn *p, double *ms, long *counts, int n_tags, int reset);

/* Intermediate activations for parity tests: name in {"conv5","pooled","fc7","cls","bbox_raw"}; tower models (MultiPathNet, the
 * ResNet / op-list tower forms) keep "bbox_raw", "cls_k" (the K integral classifiers' pre-softmax logits, [N, K * C] row-major) and
 * "cat" (the towers' outputs side by side, [N, towers * fc_dim]); plain ResNet / op-list models keep "cls", "fc7" (the head's
 * average-pooled features) and "bbox_raw". */
int mpn_frcnn_debug_tensor(mpn_frcnn *p, const char *name, const float **d_ptr, size_t *n_elems);

#ifdef __cplusplus
}
#endif
#endif /* MPN_H */
