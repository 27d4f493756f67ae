// th_compat.cpp -> libnms.so : TH-ABI drop-in for the reference's nms.c exports (see
// include/mpn_libnms.h).  Links against libmpn_hip.so; the THFloatTensor_* helpers are left
// undefined on purpose: in a Torch7 process they resolve to libTH, in this repo's tests to the
// oracle's TH shim (loaded RTLD_GLOBAL first).
#include <cstdio>
#include <cstdlib>

#include "../../include/mpn.h"
#include "../../include/mpn_libnms.h"

extern "C" {
float *THFloatTensor_data(const THFloatTensor *self);
void THFloatTensor_resize2d(THFloatTensor *self, long size0, long size1);
void THFloatTensor_resizeAs(THFloatTensor *self, THFloatTensor *src);
int THFloatTensor_isContiguous(const THFloatTensor *self);
}

// libTH's error hook (TH/THGeneral.h: #define THError(...) _THError(__FILE__, __LINE__, __VA_ARGS__)); weak: resolved in a
// Torch7 process, absent elsewhere
extern "C" void _THError(const char *file, const int line, const char *fmt, ...) __attribute__((weak));

static void die(const char *what) {
  // the reference raises a Lua error through THAssert -> THError (longjmp back into the interpreter); do the same when
  // libTH is in the process, otherwise there is nobody to catch it: print and abort
  if (_THError) _THError(__FILE__, __LINE__, "libnms.so (mpn): %s: %s", what, mpn_last_error());
  std::fprintf(stderr, "libnms.so (mpn): %s: %s\n", what, mpn_last_error());
  std::abort();
}

extern "C" void NMS(THFloatTensor *keep, THFloatTensor *scored_boxes, float overlap) {
  const long n = scored_boxes->nDimension > 0 ? scored_boxes->size[0] : 0;
  if (n == 0) { THFloatTensor_resize2d(keep, 0, 5); return; }
  if (!THFloatTensor_isContiguous(scored_boxes) || scored_boxes->size[1] != 5) die("NMS: scored_boxes must be contiguous [M,5]");
  THFloatTensor_resize2d(keep, n, 5);  // upper bound, shrunk below (same storage)
  int kept = 0;
  if (mpn_nms_host(THFloatTensor_data(scored_boxes), (int)n, overlap, THFloatTensor_data(keep), nullptr, &kept) != MPN_OK) die("NMS");
  THFloatTensor_resize2d(keep, kept, 5);
}

extern "C" void bbox_vote(THFloatTensor *res, THFloatTensor *nms_boxes, THFloatTensor *scored_boxes, float threshold) {
  if (!THFloatTensor_isContiguous(nms_boxes) || !THFloatTensor_isContiguous(scored_boxes)) die("bbox_vote: inputs must be contiguous");
  THFloatTensor_resizeAs(res, nms_boxes);
  const long n_nms = nms_boxes->nDimension > 0 ? nms_boxes->size[0] : 0;
  const long m = scored_boxes->nDimension > 0 ? scored_boxes->size[0] : 0;
  if (n_nms == 0) return;
  if (nms_boxes->size[1] != 5 || (m > 0 && scored_boxes->size[1] != 5)) die("bbox_vote: expected [.,5] tensors");
  if (mpn_bbox_vote_host(THFloatTensor_data(nms_boxes), (int)n_nms, m ? THFloatTensor_data(scored_boxes) : nullptr, (int)m, threshold,
                         THFloatTensor_data(res)) != MPN_OK)
    die("bbox_vote");
}
