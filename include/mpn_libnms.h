/* mpn_libnms.h — the libnms.so drop-in: exactly the two symbols utils.lua:15-19 binds through the
 * LuaJIT FFI (`ffi.load('./libnms.so')`, utils.lua:21-26), with Torch7's THFloatTensor ABI.
 *
 *   void NMS(THFloatTensor *keep, THFloatTensor *scored_boxes, float overlap);        (nms.c:59)
 *   void bbox_vote(THFloatTensor *res, THFloatTensor *nms_boxes,
 *                  THFloatTensor *scored_boxes, float threshold);                      (nms.c:110)
 *
 * Ownership/behaviour as the reference: the caller passes an empty FloatTensor for the result, the
 * callee resizes it (THFloatTensor_resize2d / resizeAs — resolved from the libTH the host process
 * has already loaded) and fills it; inputs are borrowed, never modified; M = 0 gives a [0,5] result.
 * Both calls are synchronous and run the wavefront kernels of libmpn_hip.so on the current device
 * (mpn_nms_host / mpn_bbox_vote_host); any M is accepted, as nms.c does.  A failure (non-contiguous input, no usable
 * device) raises through libTH's _THError like the reference's THAssert when libTH is in the process, and aborts otherwise.
 */
#ifndef MPN_LIBNMS_H
#define MPN_LIBNMS_H
#include <stddef.h>
#ifdef __cplusplus
extern "C" {
#endif

/* Field order of Torch7 lib/TH/generic/THTensor.h (the fields nms.c reads: size, nDimension). */
typedef struct THFloatStorage THFloatStorage;
typedef struct THFloatTensor {
  long *size;
  long *stride;
  int nDimension;
  THFloatStorage *storage;
  ptrdiff_t storageOffset;
  int refcount;
  char flag;
} THFloatTensor;

void NMS(THFloatTensor *keep, THFloatTensor *scored_boxes, float overlap);
void bbox_vote(THFloatTensor *res, THFloatTensor *nms_boxes, THFloatTensor *scored_boxes, float threshold);

#ifdef __cplusplus
}
#endif
#endif
