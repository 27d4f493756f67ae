// common.hip — error reporting, version, device query.
#include <atomic>
#include <map>
#include <mutex>
#include <set>
#include <utility>

#include "mpn_internal.h"

namespace mpn {
static thread_local char g_err[512] = "";
void set_error(const char *fmt, ...) {
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(g_err, sizeof(g_err), fmt, ap);
  va_end(ap);
}

static std::atomic<unsigned long long> g_alloc_gen{1};
unsigned long long alloc_generation() { return g_alloc_gen.load(std::memory_order_relaxed); }
void bump_alloc_generation() { g_alloc_gen.fetch_add(1, std::memory_order_relaxed); }

void Scratch::release() {
  for (int i = 0; i < SCR_NSLOTS; ++i) {
    if (buf[i]) (void)hipFree(buf[i]);
    buf[i] = nullptr; bytes[i] = 0;
  }
}

static thread_local Scratch *t_scratch = nullptr;
ScratchScope::ScratchScope(Scratch *sc) : prev(t_scratch) { t_scratch = sc; }
ScratchScope::~ScratchScope() { t_scratch = prev; }

static std::mutex g_reg_mu;
static std::map<std::pair<int, hipStream_t>, Scratch *> g_registry;   // module-level calls: one Scratch per (device, stream)
static std::set<std::pair<const void *, int>> g_attr_done;            // (kernel, device) pairs whose LDS limit is raised

static int scratch_get_impl(ScratchSlot slot, size_t need, hipStream_t s, void **out, bool zero);
int scratch_get(ScratchSlot slot, size_t need, hipStream_t s, void **out) { return scratch_get_impl(slot, need, s, out, false); }
int scratch_get_zeroed(ScratchSlot slot, size_t need, hipStream_t s, void **out) { return scratch_get_impl(slot, need, s, out, true); }

static int scratch_get_impl(ScratchSlot slot, size_t need, hipStream_t s, void **out, bool zero) {
  Scratch *sc = t_scratch;
  if (!sc) {
    int dev = 0;
    MPN_CHECK_HIP(hipGetDevice(&dev));
    std::lock_guard<std::mutex> lk(g_reg_mu);
    Scratch *&e = g_registry[{dev, s}];
    if (!e) { e = new Scratch(); e->device = dev; }
    sc = e;
  }
  if (need > sc->bytes[slot]) {  // grows monotonically; steady state allocates nothing
    MPN_CHECK_HIP(hipStreamSynchronize(s));
    bump_alloc_generation();  // captured launch graphs hold the old pointer (bumped BEFORE the free: a failing hipMalloc must not leave them replayable)
    if (sc->buf[slot]) (void)hipFree(sc->buf[slot]);
    sc->buf[slot] = nullptr; sc->bytes[slot] = 0;
    MPN_CHECK_HIP(hipMalloc(&sc->buf[slot], need));
    sc->bytes[slot] = need;
    if (zero) MPN_CHECK_HIP(hipMemsetAsync(sc->buf[slot], 0, need, s));
  }
  *out = sc->buf[slot];
  return MPN_OK;
}

int set_max_dyn_lds(const void *fn, int bytes) {
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  std::lock_guard<std::mutex> lk(g_reg_mu);
  if (g_attr_done.count({fn, dev})) return MPN_OK;
  MPN_CHECK_HIP(hipFuncSetAttribute(fn, hipFuncAttributeMaxDynamicSharedMemorySize, bytes));
  g_attr_done.insert({fn, dev});
  return MPN_OK;
}
}  // namespace mpn

extern "C" int mpn_stream_release(void *stream) {
  hipStream_t s = mpn::as_stream(stream);
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  mpn::Scratch *sc = nullptr;
  {
    std::lock_guard<std::mutex> lk(mpn::g_reg_mu);
    auto it = mpn::g_registry.find({dev, s});
    if (it == mpn::g_registry.end()) return MPN_OK;
    sc = it->second;
    mpn::g_registry.erase(it);
  }
  // ownership was taken above: free on every path (ADVICE r3 — an early return here leaked the Scratch and its buffers).  hipFree
  // synchronises the device itself, so a failed stream sync only changes the status that is reported.
  const hipError_t e = hipStreamSynchronize(s);
  sc->release();
  delete sc;
  if (e != hipSuccess) { mpn::set_error("mpn_stream_release: hipStreamSynchronize: %s (scratch freed)", hipGetErrorString(e)); return MPN_EHIP; }
  return MPN_OK;
}

extern "C" int mpn_release_all_scratch(void) {
  std::map<std::pair<int, hipStream_t>, mpn::Scratch *> taken;
  {
    std::lock_guard<std::mutex> lk(mpn::g_reg_mu);
    taken.swap(mpn::g_registry);
  }
  int cur = 0, rc = MPN_OK;
  const bool have_cur = hipGetDevice(&cur) == hipSuccess;  // no early return once the registry was taken: everything is freed below
  if (!have_cur) rc = MPN_EHIP;
  for (auto &kv : taken) {
    if (hipSetDevice(kv.first.first) != hipSuccess || hipDeviceSynchronize() != hipSuccess) rc = MPN_EHIP;
    kv.second->release();
    delete kv.second;
  }
  if (have_cur) (void)hipSetDevice(cur);
  if (rc) mpn::set_error("mpn_release_all_scratch: a device could not be synchronised");
  return rc;
}

extern "C" int mpn_version(void) { return MPN_VERSION; }
extern "C" const char *mpn_last_error(void) { return mpn::g_err; }

extern "C" int mpn_device_info(char *name, int name_len, int *cu_count, size_t *hbm_bytes) {
  int dev = 0;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  hipDeviceProp_t prop;
  MPN_CHECK_HIP(hipGetDeviceProperties(&prop, dev));
  if (name && name_len > 0) {
    strncpy(name, prop.gcnArchName, (size_t)name_len - 1);
    name[name_len - 1] = 0;
  }
  if (cu_count) *cu_count = prop.multiProcessorCount;
  if (hbm_bytes) *hbm_bytes = prop.totalGlobalMem;
  return MPN_OK;
}
