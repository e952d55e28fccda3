// comm.hip — the collectives of SURVEY.md 8(e) behind the C ABI, for hosts that do not bring a process group of their own: RCCL over xGMI, one process per GPU.
//
// Reference seam: the process group of main.py:39-40 (`torch.distributed.init_process_group`) and the DistributedDataParallel wrap of engine/vision_engine.py:313,510
// (gradient all-reduce, overlapped with backward), plus the query all-gather of the sharded CBIR search (path B).  The Python host of this repo keeps using
// c10d (visiondk_amd/comm.py: the maintainer's process already owns that communicator); these entry points are the same exchange for a C / C++ host:
//
//     vdk_comm_unique_id(id)              rank 0, once; the host ships the 128 bytes to the other ranks by whatever rendezvous it has (file, socket, MPI)
//     vdk_comm_init(id, rank, world, &c)  every rank, on its GPU (hipSetDevice first): communicator + a dedicated non-blocking stream for the collectives
//     vdk_vit_backward(..., on_ready = [](user, off, n) { vdk_allreduce_bucket(c, grads, off, n, launch_stream); }, ...)
//     vdk_comm_finish(c, launch_stream)   before vdk_sumsq_f32 / vdk_sgd_step: the launch stream waits for every collective issued since the last finish
//     vdk_allgather(c, q_local, q_all, bytes, stream)   the sharded search's query exchange
//
// A bucket's all-reduce is ordered AFTER the kernels that produced it (an event recorded on the launch stream at call time = right after they were enqueued, which is when
// vdk_*_backward fires the callback) and runs on the communicator's own stream, so it overlaps the rest of the backward -- the c10d behaviour, without c10d.
// librccl is opened at first use (dlopen; a process that already mapped it -- PyTorch's copy -- gets that one back), so the library has no link-time dependency on it and a
// single-GPU host never loads it.
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include "vdk_host.h"

#ifndef VDK_EMU
#include <dlfcn.h>
#include <rccl/rccl.h>
#include <mutex>
#include <new>
#include <vector>

namespace {
struct Rccl {
  void* h = nullptr;
  ncclResult_t (*GetUniqueId)(ncclUniqueId*) = nullptr;
  ncclResult_t (*CommInitRank)(ncclComm_t*, int, ncclUniqueId, int) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
};
Rccl g_rccl;
std::once_flag g_once;
bool rccl_load() {
  std::call_once(g_once, [] {
    const char* env = getenv("VDK_RCCL_LIB");
    const char* names[] = {env, "librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"};
    for (const char* n : names) {
      if (!n) continue;
      g_rccl.h = dlopen(n, RTLD_NOW | RTLD_GLOBAL);
      if (g_rccl.h) break;
    }
    if (!g_rccl.h) return;
    g_rccl.GetUniqueId = (decltype(g_rccl.GetUniqueId))dlsym(g_rccl.h, "ncclGetUniqueId");
    g_rccl.CommInitRank = (decltype(g_rccl.CommInitRank))dlsym(g_rccl.h, "ncclCommInitRank");
    g_rccl.CommDestroy = (decltype(g_rccl.CommDestroy))dlsym(g_rccl.h, "ncclCommDestroy");
    g_rccl.AllReduce = (decltype(g_rccl.AllReduce))dlsym(g_rccl.h, "ncclAllReduce");
    g_rccl.AllGather = (decltype(g_rccl.AllGather))dlsym(g_rccl.h, "ncclAllGather");
    g_rccl.GetErrorString = (decltype(g_rccl.GetErrorString))dlsym(g_rccl.h, "ncclGetErrorString");
  });
  return g_rccl.h && g_rccl.GetUniqueId && g_rccl.CommInitRank && g_rccl.CommDestroy && g_rccl.AllReduce && g_rccl.AllGather;
}
int rccl_fail(const char* what, ncclResult_t r) {
  char msg[256];
  snprintf(msg, sizeof(msg), "%s: %s", what, g_rccl.GetErrorString ? g_rccl.GetErrorString(r) : "RCCL error");
  return vdk_fail(VDK_ELAUNCH, msg);
}
}  // namespace

#define VDK_COMM_READY_RING 64
struct VdkComm {
  ncclComm_t comm;
  hipStream_t stream;          // the collectives' stream
  hipEvent_t ready_ring[VDK_COMM_READY_RING]; int ready_next;      // launch stream -> comm stream: a RING of events, one per collective in flight (re-recording a single event
                                                                   // while the wait on its previous record is still pending in the other queue made the host call block on this stack)
  hipEvent_t done;             // comm stream -> launch stream
  int rank, world;
  int64_t issued;              // collectives since the last vdk_comm_finish
  // timing trace (vdk_comm_trace): per all-reduce a (start, end) event pair on the collectives' stream, and marks on the launch stream; read back relative to the first mark
  bool trace;
  std::vector<hipEvent_t> ar0, ar1, marks;
  std::vector<int64_t> ar_numel;
};

extern "C" {

int vdk_comm_unique_id(void* id128) {
  if (!id128) return vdk_fail(VDK_EINVAL, "vdk_comm_unique_id: null pointer");
  if (!rccl_load()) return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_unique_id: librccl not found (set VDK_RCCL_LIB)");
  static_assert(sizeof(ncclUniqueId) == 128, "ncclUniqueId is 128 bytes");
  ncclUniqueId id;
  const ncclResult_t r = g_rccl.GetUniqueId(&id);
  if (r != ncclSuccess) return rccl_fail("ncclGetUniqueId", r);
  memcpy(id128, &id, sizeof(id));
  return VDK_OK;
}

int vdk_comm_init(const void* id128, int32_t rank, int32_t world, VdkComm** out) {
  if (!id128 || !out || world < 1 || rank < 0 || rank >= world) return vdk_fail(VDK_EINVAL, "vdk_comm_init: bad argument");
  if (!rccl_load()) return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_init: librccl not found (set VDK_RCCL_LIB)");
  VdkComm* c = new (std::nothrow) VdkComm();
  if (!c) return vdk_fail(VDK_EINVAL, "vdk_comm_init: out of memory");
  ncclUniqueId id;
  memcpy(&id, id128, sizeof(id));
  ncclResult_t r = g_rccl.CommInitRank(&c->comm, world, id, rank);
  if (r != ncclSuccess) { delete c; return rccl_fail("ncclCommInitRank", r); }
  // The collectives' stream is a HIGH-PRIORITY stream (as torch's NCCL streams are): HIP multiplexes the streams of a process onto a few hardware queues
  // (GPU_MAX_HW_QUEUES, 4 by default) per priority level, and a normal-priority stream created late can land on the SAME queue as the launch stream -- every collective then
  // runs serialised between the backward's kernels instead of beside them (measured: tools/overlap_probe.py, the step grew by exactly the collectives' duration).
  int prio_least = 0, prio_greatest = 0;
  (void)hipDeviceGetStreamPriorityRange(&prio_least, &prio_greatest);
  bool ok = hipStreamCreateWithPriority(&c->stream, hipStreamNonBlocking, prio_greatest) == hipSuccess && hipEventCreateWithFlags(&c->done, hipEventDisableTiming) == hipSuccess;
  for (int i = 0; ok && i < VDK_COMM_READY_RING; ++i) ok = hipEventCreateWithFlags(&c->ready_ring[i], hipEventDisableTiming) == hipSuccess;
  c->ready_next = 0;
  if (!ok) {
    g_rccl.CommDestroy(c->comm); delete c;
    return vdk_fail(VDK_ELAUNCH, "vdk_comm_init: stream / event creation failed");
  }
  c->rank = rank; c->world = world; c->issued = 0; c->trace = false;
  *out = c;
  return VDK_OK;
}

int vdk_comm_destroy(VdkComm* c) {
  if (!c) return VDK_OK;
  (void)hipStreamSynchronize(c->stream);
  g_rccl.CommDestroy(c->comm);
  for (int i = 0; i < VDK_COMM_READY_RING; ++i) (void)hipEventDestroy(c->ready_ring[i]);
  (void)hipEventDestroy(c->done);
  for (auto* v : {&c->ar0, &c->ar1, &c->marks}) for (hipEvent_t e : *v) (void)hipEventDestroy(e);
  (void)hipStreamDestroy(c->stream);
  delete c;
  return VDK_OK;
}

// SUM all-reduce of grads[offset, offset + numel) (fp32, in place) over the communicator, ordered after everything enqueued on launch_stream so far; returns at once.
int vdk_allreduce_bucket(VdkComm* c, float* grads, int64_t offset, int64_t numel, void* launch_stream) {
  if (!c || !grads || offset < 0 || numel < 0) return vdk_fail(VDK_EINVAL, "vdk_allreduce_bucket: bad argument");
  if (numel == 0) return VDK_OK;
  // (more than VDK_COMM_READY_RING collectives in flight before vdk_comm_finish re-use an event whose wait may not have been consumed: re-recording it then orders the
  //  OLDER collective behind newer kernels -- later, never earlier -- so results stay correct; only the overlap of that one bucket suffers.  64 covers 2 GB at 32 MB.)
  hipEvent_t ready = c->ready_ring[c->ready_next]; c->ready_next = (c->ready_next + 1) % VDK_COMM_READY_RING;
  if (hipEventRecord(ready, (hipStream_t)launch_stream) != hipSuccess || hipStreamWaitEvent(c->stream, ready, 0) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_allreduce_bucket: event record / wait failed");
  hipEvent_t e0 = nullptr, e1 = nullptr;
  if (c->trace) {
    if (hipEventCreate(&e0) != hipSuccess || hipEventCreate(&e1) != hipSuccess || hipEventRecord(e0, c->stream) != hipSuccess) {
      if (e0) (void)hipEventDestroy(e0);
      if (e1) (void)hipEventDestroy(e1);
      return vdk_fail(VDK_ELAUNCH, "vdk_allreduce_bucket: trace event");
    }
  }
  static const bool skip = getenv("VDK_COMM_SKIP_RCCL") && atoi(getenv("VDK_COMM_SKIP_RCCL")) > 0;      // diagnosis only: everything but the RCCL call itself
  const ncclResult_t r = skip ? ncclSuccess : g_rccl.AllReduce(grads + offset, grads + offset, (size_t)numel, ncclFloat32, ncclSum, c->comm, c->stream);
  if (r != ncclSuccess) {
    if (e0) (void)hipEventDestroy(e0);
    if (e1) (void)hipEventDestroy(e1);
    return rccl_fail("ncclAllReduce", r);
  }
  if (c->trace) { c->ar0.push_back(e0); c->ar1.push_back(e1); c->ar_numel.push_back(numel); }      // (the end event is recorded by vdk_comm_trace_close_last, after an optional stand-in)
  ++c->issued;
  return VDK_OK;
}

// ---- timing trace: where in time the collectives sit relative to the launch stream's kernels (the one-GPU evidence of communication / compute overlap) -----------------------
int vdk_comm_trace(VdkComm* c, int32_t enable) {
  if (!c) return vdk_fail(VDK_EINVAL, "vdk_comm_trace: null communicator");
  for (auto* v : {&c->ar0, &c->ar1, &c->marks}) { for (hipEvent_t e : *v) (void)hipEventDestroy(e); v->clear(); }
  c->ar_numel.clear();
  c->trace = enable != 0;
  return VDK_OK;
}
// end-of-collective event of the all-reduce issued last (after anything the caller enqueued on the collectives' stream behind it: tests put a stand-in kernel there)
int vdk_comm_trace_close_last(VdkComm* c) {
  if (!c || !c->trace || c->ar1.empty()) return VDK_OK;
  return hipEventRecord(c->ar1.back(), c->stream) == hipSuccess ? VDK_OK : vdk_fail(VDK_ELAUNCH, "vdk_comm_trace_close_last: event record failed");
}
// a time stamp on the launch stream (mark 0 is the origin of every time vdk_comm_trace_read returns)
int vdk_comm_mark(VdkComm* c, void* launch_stream) {
  if (!c) return vdk_fail(VDK_EINVAL, "vdk_comm_mark: null communicator");
  if (!c->trace) return VDK_OK;
  hipEvent_t e;
  if (hipEventCreate(&e) != hipSuccess || hipEventRecord(e, (hipStream_t)launch_stream) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_comm_mark: event record failed");
  c->marks.push_back(e);
  return VDK_OK;
}
// SYNCHRONISES (a diagnostic): ar_ms [2 * n_ar] = (start, end) of every traced all-reduce, marks_ms [n_marks], all in milliseconds after mark 0; numel [n_ar]
int vdk_comm_trace_read(VdkComm* c, float* ar_ms, int64_t* numel, int32_t ar_cap, int32_t* n_ar, float* marks_ms, int32_t marks_cap, int32_t* n_marks) {
  if (!c || !n_ar || !n_marks) return vdk_fail(VDK_EINVAL, "vdk_comm_trace_read: bad argument");
  *n_ar = (int32_t)c->ar0.size(); *n_marks = (int32_t)c->marks.size();
  if (c->marks.empty()) return vdk_fail(VDK_EINVAL, "vdk_comm_trace_read: no mark recorded (vdk_comm_mark)");
  if (hipStreamSynchronize(c->stream) != hipSuccess || hipDeviceSynchronize() != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_comm_trace_read: synchronise failed");
  for (int i = 0; i < *n_marks && i < marks_cap && marks_ms; ++i)
    if (hipEventElapsedTime(&marks_ms[i], c->marks[0], c->marks[i]) != hipSuccess) return vdk_fail(VDK_ELAUNCH, "vdk_comm_trace_read: elapsed time");
  for (int i = 0; i < *n_ar && i < ar_cap && ar_ms; ++i) {
    if (hipEventElapsedTime(&ar_ms[2 * i], c->marks[0], c->ar0[i]) != hipSuccess || hipEventElapsedTime(&ar_ms[2 * i + 1], c->marks[0], c->ar1[i]) != hipSuccess)
      return vdk_fail(VDK_ELAUNCH, "vdk_comm_trace_read: elapsed time (was vdk_comm_trace_close_last called for every all-reduce?)");
    if (numel) numel[i] = c->ar_numel[i];
  }
  return VDK_OK;
}
void* vdk_comm_stream(VdkComm* c) { return c ? (void*)c->stream : nullptr; }

// launch_stream waits (on the device, no host synchronisation) for every collective issued on this communicator since the last call
int vdk_comm_finish(VdkComm* c, void* launch_stream) {
  if (!c) return vdk_fail(VDK_EINVAL, "vdk_comm_finish: null communicator");
  if (c->issued == 0) return VDK_OK;
  if (hipEventRecord(c->done, c->stream) != hipSuccess || hipStreamWaitEvent((hipStream_t)launch_stream, c->done, 0) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_comm_finish: event record / wait failed");
  c->issued = 0;
  return VDK_OK;
}

// recv[r * bytes_per_rank ...] = rank r's send buffer (the sharded search's query all-gather); ordered after launch_stream's earlier work, and launch_stream's later
// work after it
int vdk_allgather(VdkComm* c, const void* send, void* recv, int64_t bytes_per_rank, void* launch_stream) {
  if (!c || !send || !recv || bytes_per_rank <= 0) return vdk_fail(VDK_EINVAL, "vdk_allgather: bad argument");
  // (more than VDK_COMM_READY_RING collectives in flight before vdk_comm_finish re-use an event whose wait may not have been consumed: re-recording it then orders the
  //  OLDER collective behind newer kernels -- later, never earlier -- so results stay correct; only the overlap of that one bucket suffers.  64 covers 2 GB at 32 MB.)
  hipEvent_t ready = c->ready_ring[c->ready_next]; c->ready_next = (c->ready_next + 1) % VDK_COMM_READY_RING;
  if (hipEventRecord(ready, (hipStream_t)launch_stream) != hipSuccess || hipStreamWaitEvent(c->stream, ready, 0) != hipSuccess)
    return vdk_fail(VDK_ELAUNCH, "vdk_allgather: event record / wait failed");
  const ncclResult_t r = g_rccl.AllGather(send, recv, (size_t)bytes_per_rank, ncclInt8, c->comm, c->stream);
  if (r != ncclSuccess) return rccl_fail("ncclAllGather", r);
  ++c->issued;
  return vdk_comm_finish(c, launch_stream);
}

int vdk_comm_rank(const VdkComm* c) { return c ? c->rank : -1; }
int vdk_comm_world(const VdkComm* c) { return c ? c->world : -1; }

}  // extern "C"

#else   // the CPU SIMT emulation of the tests has no RCCL: the entry points exist (the ABI table binds every symbol) and say so

struct VdkComm { int unused; };
extern "C" {
int vdk_comm_unique_id(void*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_init(const void*, int32_t, int32_t, VdkComm**) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_destroy(VdkComm*) { return VDK_OK; }
int vdk_allreduce_bucket(VdkComm*, float*, int64_t, int64_t, void*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_finish(VdkComm*, void*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_allgather(VdkComm*, const void*, void*, int64_t, void*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_rank(const VdkComm*) { return -1; }
int vdk_comm_world(const VdkComm*) { return -1; }
int vdk_comm_trace(VdkComm*, int32_t) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_trace_close_last(VdkComm*) { return VDK_OK; }
int vdk_comm_mark(VdkComm*, void*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
int vdk_comm_trace_read(VdkComm*, float*, int64_t*, int32_t, int32_t*, float*, int32_t, int32_t*) { return vdk_fail(VDK_EUNSUPPORTED, "vdk_comm_*: no RCCL in the emulator build"); }
void* vdk_comm_stream(VdkComm*) { return nullptr; }
}
#endif
