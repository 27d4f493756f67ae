// dense.h — internal interface of the fp32-MFMA dense kernels (conv3x3 implicit GEMM, linear GEMM,
// 2x2 ceil max-pool, layout converters, weight packers).  Used by the module-level C entry points
// (dense.hip) and by the fused pipeline (pipeline.hip).
//
// HBM layouts (MI355X-first; chosen so that every LDS image is a linear copy of HBM and every MFMA
// operand fetch is one conflict-free ds_read_b128):
//
//   C8P activation  [Cb][Hp][Wp][8] fp32, Cb = ceil(C/8).  Pixel (y,x) of channel c lives at
//        ((c/8 * Hp + y + 1) * Wp + x + 1) * 8 + c%8.  Row 0, column 0 and everything at/after row
//        H+1 / column W+1 is a ZERO halo: the 3x3 conv's padding and the ragged tile edges are read
//        straight from HBM with no bounds checks.  Pad channels (c >= C) are zero.
//   C8 matrix       [K/8][Mp][8] fp32 — the same thing for a [M,K] matrix (rows = "pixels").
//   packed conv W   [Cin8/8][9][CoutP][8]  (tap = ky*3+kx; element j of the last dim = cin chunk*8+j)
//   packed linear W [K8/8][NP][8]
//   The k-pair of one v_mfma_f32_32x32x2_f32 is (cin j, cin j+4) of a chunk: lanes 0-31 fetch floats
//   0-3 of the 32-byte pixel/row record, lanes 32-63 floats 4-7, identically for both operands.
#pragma once
#include "mpn_internal.h"

namespace mpn {

struct Act {      // C8P activation view
  float *p;
  int C, H, W;    // logical dims
  int Hp, Wp;     // physical padded dims (rows, cols)
  __host__ __device__ int Cb() const { return (C + 7) / 8; }
  __host__ __device__ size_t plane() const { return (size_t)Hp * Wp * 8; }  // floats per channel block
  __host__ __device__ size_t elems() const { return plane() * Cb(); }
};

// physical dims for a logical HxW map: 1 halo row/col in front, >= 1 behind, rounded so that whole
// conv tiles (8 rows x 32 cols + halo) can be read without leaving the allocation.
inline int act_hp(int H) { return ((H + 15) / 16) * 16 + 2; }
inline int act_wp(int W) { return ((W + 31) / 32) * 32 + 2; }
inline Act make_act(float *p, int C, int H, int W) { return Act{p, C, H, W, act_hp(H), act_wp(W)}; }
inline size_t act_bytes(int C, int H, int W) { return (size_t)((C + 7) / 8) * act_hp(H) * act_wp(W) * 8 * sizeof(float); }

inline int round_up(int a, int b) { return (a + b - 1) / b * b; }
inline int conv_coutp(int Cout) { return round_up(Cout, 128); }
inline size_t conv_wpk_elems(int Cin, int Cout) { return (size_t)((Cin + 7) / 8) * 9 * conv_coutp(Cout) * 8; }
inline size_t conv_wino_elems(int Cin, int Cout) { return (size_t)((Cin + 7) / 8) * 16 * conv_coutp(Cout) * 8; }
inline int lin_np(int N) { return round_up(N, 128); }
inline int lin_mp(int M) { return round_up(M, 128); }
inline size_t lin_wpk_elems(int K, int N) { return (size_t)((K + 7) / 8) * lin_np(N) * 8; }
inline size_t mat_c8_elems(int M, int K) { return (size_t)((K + 7) / 8) * lin_mp(M) * 8; }

// --- packers / converters (all enqueue on `s`) -------------------------------------------------
int pack_conv_weights(const float *d_w, const float *d_b, int Cin, int Cout, float *d_wpk, float *d_bpk, hipStream_t s);
// Winograd F(2x2,3x3) filter transform G g G^T -> [Cin8/8][16][CoutP][8]
int pack_conv_weights_wino(const float *d_w, int Cin, int Cout, float *d_wino, hipStream_t s);
// inner: K index permutation k_src = (q/inner*8 + j)*inner + q%inner for chunk q, element j
// (inner=1: plain; inner=PH*PW: fc6 behind a channel-blocked ROI pool).
int pack_linear_weights(const float *d_w, const float *d_b, int K, int N, int inner, float *d_wpk, float *d_bpk,
                        hipStream_t s);
int nchw_to_c8p(const float *d_in, int C, int H, int W, Act out, hipStream_t s);   // also zeroes pad channels
int c8p_to_nchw(Act in, float *d_out, hipStream_t s);
int rowmajor_to_c8(const float *d_x, int M, int K, float *d_c8, hipStream_t s);
int c8_to_rowmajor(const float *d_c8, int M, int N, float *d_y, hipStream_t s);
// image transformer fused with the NCHW->C8P conversion (channels 3..7 zero)
int image_transform_c8p(const float *d_in, int H, int W, const int *swap, double scale, const double *mean,
                        const double *std, int has_std, Act out, hipStream_t s);

// --- compute ------------------------------------------------------------------------------------
// out = relu?(conv3x3(in) + b); optional fused ceil-mode 2x2 max-pool writes `pooled` as well
// (out.p may be null when only the pooled map is needed).
// d_wino (optional): Winograd-transformed weights; used when the variant selector picks the Winograd kernel.
// batch_invariant (Winograd only): the launch plan must not depend on the map's height — no split-K, no tail split, the 8 x 32-px block
// geometry — so that a cell of a mosaic of per-ROI maps (resnet.hip) gets the same bits whatever the number of maps in the mosaic.
int conv3x3_c8p(Act in, const float *d_wpk, const float *d_bpk, int Cout, int relu, Act out, Act pooled, hipStream_t s,
                const float *d_wino = nullptr, bool batch_invariant = false);
// first layer (<= 4 input channels): K = 9 taps x 4 channels formulation, weights [36][CoutP] (pack_conv_weights_first);
// bias from the direct packing's d_bpk; no fused pool / split-K (the layer is bound by its output stores)
size_t conv_first_elems(int Cout);
int pack_conv_weights_first(const float *d_w, int Cin, int Cout, float *d_w36, hipStream_t s);
int conv3x3_first_c8p(Act in, const float *d_w36, const float *d_bpk, int Cout, int relu, Act out, hipStream_t s);
int conv3x3_variant_for(int Cout, bool has_wino = false);  // 7 = Winograd, 1 = direct 128 couts x 4 rows x 32 cols tile, 2 = direct 64 x 8 x 32
int maxpool2x2_c8p(Act in, Act out, hipStream_t s);
// zero the halo (everything of each plane outside the H x W interior) of n C8P activations, one launch per kHaloMax of them
constexpr int kHaloMax = 24;
struct HaloDesc { float *p; int H, W, Hp, Wp; };
struct HaloTable { int n; size_t total; size_t first[kHaloMax]; HaloDesc d[kHaloMax]; };
int c8p_zero_halos(const Act *acts, int n, hipStream_t s);
// y = relu?(x W^T + b).  x: C8 matrix [K8/8][Mp][8]; y: C8 matrix [NP/8][Mp][8] (y_c8) and/or
// row-major [M,N] (y_rm); either may be null.
// Mp_override: row pitch of x / y when it is not lin_mp(M) (a ROI-pooled matrix viewed as (bin, roi) rows).
// d_res_c8 (optional, y_c8's layout): added before the ReLU; only with the direct form (>= 128 output tiles, C8 output only).
// row_invariant: the K summation of every output row follows ONE canonical order that depends on (K, N) only — K cut into fixed
// segments, each accumulated from zero, the segment sums added in segment order — whether the launch runs split-K (few row tiles:
// one block per segment + the reduce kernel) or un-split (many row tiles: one block walks the segments and folds its accumulator
// into a running total at each boundary).  A row's result is then bit-identical for ANY number of rows in the call, which is what
// lets the ROI-sharded mode (mpn_frcnn_shard_*) equal the unsharded one exactly and makes memoryEfficientForward's chunk
// invariance (ImageDetect.lua:126-133) hold at every size.  Used by the ROI heads (fc6 / fc7 / cls + bbox / integral heads).
// row_invariant = 2: the degenerate canonical order — ONE segment, the launch never splits K whatever the row count (short-K layers
// whose callers always bring dozens of row tiles: MultiPathNet's 1x1 mix over (bin, roi) rows, per-ROI pointwise convolutions over
// (roi, pixel) rows); combines with a residual.
int linear_c8(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8,
              float *d_y_rm, hipStream_t s, int Mp_override = 0, const float *d_res_c8 = nullptr, int row_invariant = 0);
bool linear_c8_is_direct(int M, int N, int Mp_override = 0);  // would linear_c8 run un-split for this shape?
// The scratch slot a split-K linear_c8 of this thread keeps its partial sums in.  A launch that runs on a SECOND stream of the same handle
// beside the launch stream's GEMMs (the pipelined forms' deferred heads) takes SCR_GEMM_SPLITK_SIDE for its duration.
struct SplitkSlotScope {
  explicit SplitkSlotScope(ScratchSlot slot);
  ~SplitkSlotScope();
  ScratchSlot prev;
};
// y = sum over up to three K segments of scale_seg[row % rs_mod] * (x_seg . w_seg) (+ b, ReLU): MultiPathNet's mix GEMM with nn.Normalize
// of its three pooled maps applied where the accumulator is folded, instead of a read-modify-write pass over the pooled matrix.
// Always launched un-split (like row_invariant = 2, so a row's result does not depend on the row count); k_end = the K index (multiple of 32) at which segment i ends.
// bin_rows > 0 (round 6): the rows are packed (bin, roi) pairs, bin_rows per bin, x_pitch rows between the operand's K chunks, and the output is
// scattered into [N / 8][M / bin_rows][out_Mp][8] — the fc6 operand of the MultiPathNet towers (GemmArgs in dense.hip)
struct GemmRowScale { int n_seg; int k_end[2]; const float *scale[3]; int rs_mod; int bin_rows, out_Mp, x_pitch; };
int linear_c8_rowscaled(const float *d_x_c8, int M, int K, const float *d_wpk, const float *d_bpk, int N, int relu, float *d_y_c8, hipStream_t s,
                        int Mp_override, const GemmRowScale &rs);
// fc6 on the bf16 matrix pipe with fp32 results (dense.hip: gemm_c8_split3_kernel): operands as three bf16 planes [3][K64 / 8][rows↑256][8]
size_t split3_plane_elems(int K, int rows);
int split3_planes(const float *d_c8, int K, int rows_src, int rows_valid, unsigned short *d_planes, hipStream_t s);
int linear_c8_split3(const unsigned short *d_x3, int M, int K, const unsigned short *d_w3, const float *d_bpk, int N, int relu, float *d_y_c8, hipStream_t s);
// ROI max-pool reading a C8P feature map and writing the C8 matrix the fc6 GEMM consumes:
// chunk q = cb*PH*PW + bin, row = roi.  argmax (optional) [N,C,PH,PW] int32 as the NCHW kernel.
// roi_stride: floats between consecutive rois (5; 20 selects one Foveal region out of the [4N,5] table);
// Mp: row pitch of the output matrix (0 = lin_mp(N)).
int roi_pool_c8(Act feat, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr,
                float *d_x_c8, int32_t *d_argmax, hipStream_t s, int roi_stride = 5, int Mp = 0);
// The same pooling from a pixel-major copy [y][x][Cb*8] of the map (one contiguous 1 KiB per pixel and 256 channels; bit-identical
// output, no argmax): c8p_to_pixel_major once per image, then roi_pool_pm.
size_t pixel_major_elems(Act feat);
int c8p_to_pixel_major(Act feat, float *d_pm, hipStream_t s);
int roi_pool_pm(Act feat, const float *d_pm, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr,
                float *d_x_c8, hipStream_t s, int roi_stride = 5, int Mp = 0);
// vertical range-max tables of a C8P map (levels 1..vmax_levels_for(H), each feat.elems() floats) and the ROI max-pool that
// reads them: identical output to roi_pool_c8 (no argmax), cost 2 x bin-width reads per bin instead of bin-height x bin-width
int vmax_levels_for(int H);
int build_vmax_tables(Act feat, float *d_tables, hipStream_t s);
int roi_pool_c8_rmq(Act feat, const float *d_tables, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr, float *d_x_c8, hipStream_t s, int roi_stride = 5, int Mp = 0);
// The same on PIXEL-MAJOR tables (levels 0 .. vmax_levels_for(H), pixel_major_elems(feat) floats each; level 0 = the map): coalesced
// 1-KiB wave loads; `normalize` fuses nn.Normalize(2)'s sum of squares into the pooling launch and applies x * (mul / norm),
// otherwise nn.MulConstant(mul).  Pooled values bit-identical to roi_pool_c8 / roi_pool_c8_rmq.
int build_vmax_tables_pm(Act feat, float *d_tables, hipStream_t s);
// d_scale_out (normalize only, [Mp]): write the per-ROI scale mul / norm there and leave the pooled matrix unscaled (the consumer applies it).
int roi_pool_pm_rmq(Act feat, const float *d_tables_pm, const float *d_rois, int N, int PH, int PW, float scale, RoiRule rr, float *d_x_c8, hipStream_t s, int roi_stride, int Mp, int normalize, float mul, float *d_scale_out = nullptr);
// in-place x * (mul / sqrt(sum x^2 + 1e-10)) per ROI over n_records 8-float records of a C8 matrix
int l2norm_scale_c8(float *d_x_c8, int n_records, int Mp, int N, float mul, hipStream_t s);
int mul_const_c8(float *d_x_c8, int n_records, int Mp, int N, float mul, hipStream_t s);  // nn.MulConstant on the same layout

}  // namespace mpn
