/* Implementation of the 6 TH entry points nms.c needs — TEST INFRASTRUCTURE ONLY. */
#include <TH/TH.h>
#include <stdio.h>

void mpn_th_shim_assert_fail(const char *expr, const char *file, int line) {
  fprintf(stderr, "THAssert failed: %s (%s:%d)\n", expr, file, line);
  abort();
}

float *THFloatTensor_data(const THFloatTensor *self) {
  return self->storage ? self->storage->data + self->storageOffset : NULL;
}

static void resize_nd(THFloatTensor *self, int nd, const long *sz) {
  ptrdiff_t total = 1;
  self->size = (long *)realloc(self->size, sizeof(long) * (nd > 0 ? nd : 1));
  self->stride = (long *)realloc(self->stride, sizeof(long) * (nd > 0 ? nd : 1));
  for (int d = nd - 1; d >= 0; --d) {
    self->size[d] = sz[d];
    self->stride[d] = total;
    total *= sz[d];
  }
  self->nDimension = nd;
  if (!self->storage) self->storage = (THFloatStorage *)calloc(1, sizeof(THFloatStorage));
  if (self->storage->size < total) {
    self->storage->data = (float *)realloc(self->storage->data, sizeof(float) * (total > 0 ? total : 1));
    self->storage->size = total;
  }
  self->storageOffset = 0;
}

void THFloatTensor_resize1d(THFloatTensor *self, long size0) { long s[1] = {size0}; resize_nd(self, 1, s); }
void THFloatTensor_resize2d(THFloatTensor *self, long size0, long size1) {
  long s[2] = {size0, size1};
  resize_nd(self, 2, s);
}
void THFloatTensor_resizeAs(THFloatTensor *self, THFloatTensor *src) {
  long s[8];
  for (int d = 0; d < src->nDimension; ++d) s[d] = src->size[d];
  resize_nd(self, src->nDimension, s);
}
void THFloatTensor_zero(THFloatTensor *self) {
  ptrdiff_t total = 1;
  for (int d = 0; d < self->nDimension; ++d) total *= self->size[d];
  if (self->nDimension > 0 && total > 0) memset(THFloatTensor_data(self), 0, sizeof(float) * total);
}
int THFloatTensor_isContiguous(const THFloatTensor *self) {
  long z = 1;
  for (int d = self->nDimension - 1; d >= 0; --d) {
    if (self->size[d] != 1 && self->stride[d] != z) return 0;
    z *= self->size[d];
  }
  return 1;
}

THFloatTensor *mpn_th_shim_new(void) { return (THFloatTensor *)calloc(1, sizeof(THFloatTensor)); }
void mpn_th_shim_free(THFloatTensor *t) {
  if (!t) return;
  if (t->storage) { free(t->storage->data); free(t->storage); }
  free(t->size); free(t->stride); free(t);
}
THFloatTensor *mpn_th_shim_from(const float *data, long n0, long n1) {
  THFloatTensor *t = mpn_th_shim_new();
  THFloatTensor_resize2d(t, n0, n1);
  if (n0 * n1 > 0) memcpy(THFloatTensor_data(t), data, sizeof(float) * n0 * n1);
  return t;
}
long mpn_th_shim_size(const THFloatTensor *t, int dim) { return dim < t->nDimension ? t->size[dim] : 0; }
int mpn_th_shim_ndim(const THFloatTensor *t) { return t->nDimension; }
