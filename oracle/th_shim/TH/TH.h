/* Minimal TH/TH.h stand-in — TEST INFRASTRUCTURE ONLY (never shipped, never on the product path).
 *
 * Purpose: lets the reference's own nms.c (/root/reference/nms.c) compile *unmodified, from where it
 * lies* into oracle/_ref/libnms_ref.so, so that the C restatement in mpn_oracle.c and the HIP NMS can
 * be pinned against the real reference code.  Only the 8 symbols nms.c touches are declared.
 *
 * The struct mirrors the field order of Torch7's generic/THTensor.h (size, stride, nDimension,
 * storage, storageOffset, refcount, flag) so that the same layout is used by the product's
 * libnms.so drop-in (multipathnet_amd/csrc/th_compat.h); here `storage` is simplified to a
 * {data,size} pair because no real libTH exists in this container.
 */
#ifndef MPN_ORACLE_TH_SHIM_H
#define MPN_ORACLE_TH_SHIM_H
#include <stddef.h>
#include <stdlib.h>
#include <string.h>

typedef struct THFloatStorage {
  float *data;
  ptrdiff_t size;
} THFloatStorage;

typedef struct THFloatTensor {
  long *size;
  long *stride;
  int nDimension;
  THFloatStorage *storage;
  ptrdiff_t storageOffset;
  int refcount;
  char flag;
} THFloatTensor;

float *THFloatTensor_data(const THFloatTensor *self);
void THFloatTensor_resize1d(THFloatTensor *self, long size0);
void THFloatTensor_resize2d(THFloatTensor *self, long size0, long size1);
void THFloatTensor_resizeAs(THFloatTensor *self, THFloatTensor *src);
void THFloatTensor_zero(THFloatTensor *self);
int THFloatTensor_isContiguous(const THFloatTensor *self);
void mpn_th_shim_assert_fail(const char *expr, const char *file, int line);

#define THAssert(exp) \
  do { if (!(exp)) mpn_th_shim_assert_fail(#exp, __FILE__, __LINE__); } while (0)

/* shim-only helpers used by the Python test harness */
THFloatTensor *mpn_th_shim_new(void);
void mpn_th_shim_free(THFloatTensor *t);
THFloatTensor *mpn_th_shim_from(const float *data, long n0, long n1);
long mpn_th_shim_size(const THFloatTensor *t, int dim);
int mpn_th_shim_ndim(const THFloatTensor *t);

#endif
