// resnet.h — ResNet Fast R-CNN graph (models/resnet.lua:24-50): trunk conv1 / max-pool / layer1-3 on the
// image, ROIPooling(14,14,1/16), per-ROI layer4 + global average pool.  fp32 MFMA, BN folded into the convolutions by the
// caller.  Used by pipeline.hip (mpn_resnet_create); kernels in resnet.hip.
//
// Layout "C8I": [C/8][B*H*W rows, pitch rounded up to 128][8] fp32 — a batch of channel-blocked maps WITHOUT halos: the generic
// convolution gathers its input pixels with explicit bounds checks (7x7 pad 3, strides 1/2, 14x14 and 7x7 per-ROI maps), so a
// halo would only inflate the thousand small per-ROI maps.  A pixel's 8 channels are one 32-byte record, as in C8P, and one
// channel-block plane holds ALL maps' pixels: the whole batch is a C8 matrix (dense.h) whose rows are the pixels, so a 1x1 /
// stride-1 convolution is exactly dense.h's linear_c8 GEMM on it (residual add + ReLU fused in its epilogue).
#pragma once
#include "dense.h"

struct mpn_resnet_weights;
struct mpn_graph_weights;

namespace mpn {

struct ResNetGraph;

int resnet_build(const mpn_resnet_weights *rw, int max_h, int max_w, int max_rois, int pooled, ResNetGraph **out);
// the same object driven by two op lists (branching graphs: Inception-v3)
int graph_build(const mpn_graph_weights *gw, int max_h, int max_w, int max_rois, int pooled, ResNetGraph **out);
void resnet_free(ResNetGraph *g);
void resnet_set_roi_bins(ResNetGraph *g, int bin_rule);  // MPN_ROI_BINS_* of every ROI pooling of the graph (default: the CUDA branch's rule)
int resnet_feat_channels(const ResNetGraph *g);   // layer3 output channels (what the ROI pool reads)
int resnet_out_channels(const ResNetGraph *g);    // layer4 output channels (what the cls / bbox heads read)
// image [3,H,W] fp32 -> transformed -> conv1 -> pool -> layer1..3; the feature map is cached in the graph
int resnet_trunk_forward(ResNetGraph *g, const float *d_image, int H, int W, const int *swap, double scale, const double *mean,
                         const double *std, int has_std, hipStream_t s);
bool resnet_has_features(const ResNetGraph *g, int H, int W);
int resnet_n_heads(const ResNetGraph *g);
// rois (roi_stride floats apart; 5 = a plain [N,5] table, 20 = one region of a Foveal [4N,5] table) -> ROI pool -> tower `head`'s
// layer4 -> average pool -> C8 matrix [out_c/8][Mp][8] (row = roi)
// lane: which set of per-ROI activation buffers the tower uses (0, or 1 where resnet_has_second_lane): towers on different lanes may run
// concurrently on different streams once resnet_heads_prepare has run for the image on a stream both are ordered after
int resnet_head_forward(ResNetGraph *g, int head, const float *d_rois, int roi_stride, int N, float spatial_scale, float *d_feat_c8, int Mp,
                        hipStream_t s, int lane = 0);
int resnet_heads_prepare(ResNetGraph *g, hipStream_t s);
bool resnet_has_second_lane(const ResNetGraph *g);

}  // namespace mpn
