// comm.hip — the ONLY cross-GPU exchange of the path: an RCCL all-gather of each rank's scored-box record.
//
// Replaces test_runner.lua:91-104 (every worker thread serialises its per-image result tables back to the main Lua
// thread) and ModelParallelTable.lua:204-236 (per-forward broadcast of whole feature maps to tower GPUs).  Images
// shard across GPUs (rank r owns images r, r+G, ...), weights are resident per rank, and what travels is one
// fixed-size record per image: top_cap rows of {x1,y1,x2,y2,score,class} + a count (~10 KB) — never features or logits.
// xGMI bandwidth is irrelevant at this size; the call is stream-ordered and has no host synchronisation, so it
// overlaps the next image's trunk.
//
// RCCL is bound at run time (dlopen by SONAME): a process that already mapped an RCCL (PyTorch ships one) shares it,
// a Lua host gets /opt/rocm/lib's, and a single-GPU host never loads the 0.5 GB library at all.
#include <dlfcn.h>

#include <chrono>
#include <cstdlib>
#include <future>
#include <memory>
#include <mutex>
#include <thread>

#include "mpn_internal.h"

namespace mpn {

// the slice of rccl.h this file uses (rccl/rccl.h:40-43,187,220,236,260,339,678; ABI-stable NCCL 2 surface)
typedef struct ncclComm *ncclComm_t;
typedef struct { char internal[128]; } ncclUniqueId;
typedef int ncclResult_t;       // ncclSuccess == 0
constexpr int kNcclFloat = 7;   // ncclFloat32 (rccl.h ncclDataType_t)

struct Rccl {
  void *h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId *) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t *, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t *, int, const int *) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void *, void *, size_t, int, ncclComm_t, hipStream_t) = nullptr;
  const char *(*GetErrorString)(ncclResult_t) = nullptr;
  ncclResult_t (*CommCount)(const ncclComm_t, int *) = nullptr;     // optional: what RCCL itself says the communicator is
  ncclResult_t (*CommUserRank)(const ncclComm_t, int *) = nullptr;
};
static Rccl g_rccl;
static std::mutex g_rccl_mu;

static int rccl_load() {
  std::lock_guard<std::mutex> lk(g_rccl_mu);
  if (g_rccl.h) return MPN_OK;
  const char *names[] = {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
  void *h = nullptr;
  for (const char *n : names) { h = dlopen(n, RTLD_NOW | RTLD_GLOBAL); if (h) break; }
  if (!h) { set_error("mpn_comm: cannot load RCCL (librccl.so.1): %s", dlerror()); return MPN_ENCCL; }
  Rccl r;
  r.h = h;
#define SYM(field, name) do { r.field = reinterpret_cast<decltype(r.field)>(dlsym(h, name)); if (!r.field) { set_error("mpn_comm: RCCL lacks %s", name); dlclose(h); return MPN_ENCCL; } } while (0)
  SYM(GetUniqueId, "ncclGetUniqueId");
  SYM(CommInitRank, "ncclCommInitRank");
  SYM(CommInitAll, "ncclCommInitAll");
  SYM(CommDestroy, "ncclCommDestroy");
  SYM(AllGather, "ncclAllGather");
  SYM(GetErrorString, "ncclGetErrorString");
#undef SYM
  r.CommCount = reinterpret_cast<decltype(r.CommCount)>(dlsym(h, "ncclCommCount"));
  r.CommUserRank = reinterpret_cast<decltype(r.CommUserRank)>(dlsym(h, "ncclCommUserRank"));
  g_rccl = r;
  return MPN_OK;
}

#define MPN_CHECK_NCCL(expr)                                                                                   \
  do {                                                                                                         \
    ncclResult_t _r = (expr);                                                                                  \
    if (_r != 0) {                                                                                             \
      ::mpn::set_error("%s: %s failed: %s", __func__, #expr, g_rccl.GetErrorString ? g_rccl.GetErrorString(_r) : "?"); \
      return MPN_ENCCL;                                                                                        \
    }                                                                                                          \
  } while (0)

// record = top_cap rows of 6 floats (rows at and beyond the count are zero) + the count as a float
__global__ void pack_det_record_kernel(const float *__restrict__ dets, const int *__restrict__ n_dets, int top_cap, float *__restrict__ rec) {
  const int t = blockIdx.x * blockDim.x + threadIdx.x;
  const int n = min(max(*n_dets, 0), top_cap);
  if (t < top_cap * 6) rec[t] = (t / 6 < n) ? dets[t] : 0.0f;
  if (t == 0) rec[top_cap * 6] = (float)n;
}

}  // namespace mpn

using namespace mpn;

struct mpn_comm {
  ncclComm_t comm = nullptr;
  int world = 1, rank = 0, device = 0;
  int rccl_ranks = 0;      // ncclCommCount of `comm` (0: no RCCL communicator — world 1 without an id; -1: the RCCL has no ncclCommCount, unverified)
  float *send = nullptr;   // this rank's packed record
  size_t send_floats = 0;
};

extern "C" int mpn_comm_get_unique_id(void *id128) {
  MPN_CHECK_ARG(id128 != nullptr);
  int rc = rccl_load();
  if (rc) return rc;
  ncclUniqueId id;
  MPN_CHECK_NCCL(g_rccl.GetUniqueId(&id));
  memcpy(id128, id.internal, sizeof(id.internal));
  return MPN_OK;
}

// The first N > 1 run must verify itself (test_runner.lua:55-66: thread k IS GPU k): ask RCCL what it built — ncclCommCount /
// ncclCommUserRank — and refuse a communicator that is not the (world, rank) the caller asked for.
static int verify_comm(mpn_comm *c, const char *who) {
  if (!c->comm) return MPN_OK;
  if (g_rccl.CommCount) {
    int n = -1;
    ncclResult_t r = g_rccl.CommCount(c->comm, &n);
    if (r != 0) { set_error("%s: ncclCommCount failed: %s", who, g_rccl.GetErrorString(r)); return MPN_ENCCL; }
    c->rccl_ranks = n;
    if (n != c->world) { set_error("%s: RCCL built a communicator of %d ranks, %d were asked for", who, n, c->world); return MPN_ENCCL; }
  } else {
    c->rccl_ranks = -1;  // an RCCL without ncclCommCount: nothing was cross-checked — reported as -1 (unverified), never as the world size
  }
  if (g_rccl.CommUserRank) {
    int ur = -1;
    ncclResult_t r = g_rccl.CommUserRank(c->comm, &ur);
    if (r != 0) { set_error("%s: ncclCommUserRank failed: %s", who, g_rccl.GetErrorString(r)); return MPN_ENCCL; }
    if (ur != c->rank) { set_error("%s: RCCL says this is rank %d, the caller said %d", who, ur, c->rank); return MPN_ENCCL; }
  }
  return MPN_OK;
}

// ncclCommInitRank blocks until every rank has arrived.  A rank that never comes (a crashed peer, a wrong id) would hang the caller — and
// the GPU lease — forever, so the call runs on a helper thread and the caller waits a bounded time (MPN_COMM_INIT_TIMEOUT_S, default
// 120 s).  On a timeout the helper is left behind (it cannot be cancelled) and the caller gets MPN_ENCCL with a message: exit the process.
static int init_timeout_s() {
  const char *e = getenv("MPN_COMM_INIT_TIMEOUT_S");
  const int v = e ? atoi(e) : 0;
  return v > 0 ? v : 120;
}

extern "C" int mpn_comm_init_rank(const void *id128, int world, int rank, mpn_comm **out) {
  MPN_CHECK_ARG(out != nullptr && world >= 1 && rank >= 0 && rank < world && (world == 1 || id128 != nullptr));
  mpn_comm *c = new mpn_comm();
  c->world = world; c->rank = rank;
  if (hipGetDevice(&c->device) != hipSuccess) { delete c; set_error("mpn_comm_init_rank: no current HIP device"); return MPN_EHIP; }
  if (world > 1 || id128) {  // world == 1 with an id: a real one-rank RCCL communicator (exercises the collective path on one GPU)
    int rc = rccl_load();
    if (rc) { delete c; return rc; }
    ncclUniqueId id;
    memcpy(id.internal, id128, sizeof(id.internal));
    struct Job { ncclComm_t comm = nullptr; ncclResult_t r = 0; };
    auto job = std::make_shared<Job>();
    auto done = std::make_shared<std::promise<void>>();
    std::future<void> fut = done->get_future();
    const int device = c->device;
    std::thread([job, done, world, id, rank, device]() {
      if (hipSetDevice(device) != hipSuccess) job->r = -1;
      else job->r = g_rccl.CommInitRank(&job->comm, world, id, rank);
      done->set_value();
    }).detach();
    const int tmo = init_timeout_s();
    if (fut.wait_for(std::chrono::seconds(tmo)) != std::future_status::ready) {
      set_error("mpn_comm_init_rank: ncclCommInitRank (rank %d of %d) did not return within %d s — a peer rank never arrived or the unique id "
                "differs between ranks; the helper thread is abandoned, exit the process (MPN_COMM_INIT_TIMEOUT_S changes the bound)", rank, world, tmo);
      delete c;
      return MPN_ENCCL;
    }
    if (job->r != 0) { set_error("mpn_comm_init_rank: ncclCommInitRank failed: %s", job->r == -1 ? "hipSetDevice on the helper thread" : g_rccl.GetErrorString(job->r)); delete c; return MPN_ENCCL; }
    c->comm = job->comm;
    int rcv = verify_comm(c, "mpn_comm_init_rank");
    if (rcv) { mpn_comm_destroy(c); return rcv; }
  }
  *out = c;
  return MPN_OK;
}

// The reference's own process model: ONE process, one worker thread per GPU (test_runner.lua:55-66).  Creates n_dev
// communicators at once (ncclCommInitAll); thread i then uses out[i] with device h_devices[i] current.
extern "C" int mpn_comm_init_all(int n_dev, const int *h_devices, mpn_comm **out) {
  MPN_CHECK_ARG(n_dev >= 1 && out != nullptr);
  int cur = 0;
  MPN_CHECK_HIP(hipGetDevice(&cur));
  if (n_dev == 1) {
    mpn_comm *c = new mpn_comm();
    c->device = h_devices ? h_devices[0] : cur;
    out[0] = c;
    return MPN_OK;
  }
  int rc = rccl_load();
  if (rc) return rc;
  ncclComm_t *comms = new ncclComm_t[n_dev];
  ncclResult_t r = g_rccl.CommInitAll(comms, n_dev, h_devices);
  if (r != 0) { set_error("mpn_comm_init_all: ncclCommInitAll failed: %s", g_rccl.GetErrorString(r)); delete[] comms; return MPN_ENCCL; }
  for (int i = 0; i < n_dev; ++i) {
    mpn_comm *c = new mpn_comm();
    c->comm = comms[i]; c->world = n_dev; c->rank = i; c->device = h_devices ? h_devices[i] : i;
    out[i] = c;
  }
  delete[] comms;
  for (int i = 0; i < n_dev; ++i) {
    int rcv = verify_comm(out[i], "mpn_comm_init_all");
    if (rcv) { for (int j = 0; j < n_dev; ++j) { mpn_comm_destroy(out[j]); out[j] = nullptr; } return rcv; }
  }
  return MPN_OK;
}

extern "C" int mpn_comm_world(const mpn_comm *c) { return c ? c->world : 0; }
extern "C" int mpn_comm_rank(const mpn_comm *c) { return c ? c->rank : -1; }
extern "C" int mpn_comm_rccl_ranks(const mpn_comm *c) { return c ? c->rccl_ranks : 0; }

extern "C" void mpn_comm_destroy(mpn_comm *c) {
  if (!c) return;
  if (c->comm && g_rccl.CommDestroy) (void)g_rccl.CommDestroy(c->comm);
  if (c->send) (void)hipFree(c->send);
  delete c;
}

extern "C" size_t mpn_det_record_floats(int top_cap) { return top_cap > 0 ? (size_t)top_cap * 6 + 1 : 1; }

extern "C" int mpn_pack_det_record(const float *d_dets, const int *d_n_dets, int top_cap, float *d_rec, void *stream) {
  MPN_CHECK_ARG(d_dets && d_n_dets && d_rec && top_cap > 0);
  hipLaunchKernelGGL(pack_det_record_kernel, dim3(cdiv(top_cap * 6, 256)), dim3(256), 0, as_stream(stream), d_dets, d_n_dets, top_cap, d_rec);
  MPN_CHECK_LAUNCH();
  return MPN_OK;
}

// the communicator (and its send buffer) lives on the device that was current at creation: a worker thread that forgot
// hipSetDevice would hand RCCL cross-device buffers
static int check_device(const mpn_comm *c, const char *who) {
  int dev = -1;
  MPN_CHECK_HIP(hipGetDevice(&dev));
  if (dev != c->device) { set_error("%s: the communicator belongs to device %d but device %d is current", who, c->device, dev); return MPN_ESTATE; }
  return MPN_OK;
}

// Generic fixed-size all-gather of float records (the exchange steps of the ROI-sharded mode, pipeline.hip): every rank
// contributes n_floats from d_send; d_out [world, n_floats].  Without an RCCL communicator (world 1) it is a device copy.
extern "C" int mpn_gather_rows(mpn_comm *c, const float *d_send, size_t n_floats, float *d_out, void *stream) {
  MPN_CHECK_ARG(c && d_send && d_out && n_floats > 0);
  hipStream_t s = as_stream(stream);
  int rc = check_device(c, "mpn_gather_rows");
  if (rc) return rc;
  if (!c->comm) {
    if (d_send != d_out) MPN_CHECK_HIP(hipMemcpyAsync(d_out, d_send, n_floats * sizeof(float), hipMemcpyDeviceToDevice, s));
    return MPN_OK;
  }
  MPN_CHECK_NCCL(g_rccl.AllGather(d_send, d_out, n_floats, kNcclFloat, c->comm, s));
  return MPN_OK;
}

extern "C" int mpn_gather_dets(mpn_comm *c, const float *d_dets, const int *d_n_dets, int top_cap, float *d_out, void *stream) {
  MPN_CHECK_ARG(c && d_dets && d_n_dets && d_out && top_cap > 0);
  hipStream_t s = as_stream(stream);
  int rcd = check_device(c, "mpn_gather_dets");
  if (rcd) return rcd;
  const size_t rec = mpn_det_record_floats(top_cap);
  if (!c->comm) return mpn_pack_det_record(d_dets, d_n_dets, top_cap, d_out, stream);  // single rank without RCCL
  if (rec > c->send_floats) {
    MPN_CHECK_HIP(hipStreamSynchronize(s));
    if (c->send) (void)hipFree(c->send);
    c->send = nullptr; c->send_floats = 0;
    MPN_CHECK_HIP(hipMalloc(&c->send, rec * sizeof(float)));
    c->send_floats = rec;
  }
  int rc = mpn_pack_det_record(d_dets, d_n_dets, top_cap, c->send, stream);
  if (rc) return rc;
  MPN_CHECK_NCCL(g_rccl.AllGather(c->send, d_out, rec, kNcclFloat, c->comm, s));
  return MPN_OK;
}
