// TEST INFRASTRUCTURE (GPU tier): an in-process DEVICE transport for nepmi_dist_* -- the same contract as
// nepmi_transport_rccl (device buffers, everything enqueued on the caller's stream, no host copy of the payload), for
// several ranks that live in ONE process as threads and share one GPU.  RCCL refuses two ranks on one device, and the
// TCP transport is a host transport (no speculative enqueue, no device-side vote), so this is what lets a 1-GPU box run
// the code path an 8-GPU node runs: device_buffers = 1 with more than one rank (tests/test_dist_inproc.py).
//
// exchange: a send records an event on the sender's stream and is queued per (source, destination); the matching
// receive makes the receiver's stream wait for that event, copies device-to-device on the receiver's stream and records
// a completion event the sender's stream then waits for (the send buffer may be overwritten afterwards).  All sends of
// a call are posted before any receive is waited for (ncclGroup semantics: no ordering deadlock), messages between one
// pair match in order.  allreduce: every rank waits for all ranks' inputs, reduces them in rank order into its own
// scratch (identical bits everywhere), and copies the result over its buffer once every rank has finished reading.
#include <hip/hip_runtime.h>

#include <condition_variable>
#include <cstdint>
#include <deque>
#include <mutex>
#include <vector>

#include "../../include/nepmi.h"

namespace {

constexpr int kMaxRanks = 8;

struct SendPost {
  const void* src;
  int64_t bytes;
  hipEvent_t ready, done;
  bool consumed = false;
};

struct Group {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::deque<SendPost*>> q; // [src * n + dst]
  // reusable barrier
  int bar_count = 0;
  long bar_gen = 0;
  // all-reduce slots of the collective in flight
  void* ar_buf[kMaxRanks];
  hipEvent_t ar_ready[kMaxRanks], ar_read[kMaxRanks];
  int failed = 0;
};

struct Ctx {
  Group* g;
  int rank;
  void* scratch = nullptr; // 4 KB of device memory
  std::vector<hipEvent_t> garbage;
};

void barrier(Group* g)
{
  std::unique_lock<std::mutex> lk(g->mu);
  const long gen = g->bar_gen;
  if (++g->bar_count == g->n) {
    g->bar_count = 0;
    ++g->bar_gen;
    g->cv.notify_all();
  } else {
    g->cv.wait(lk, [&] { return g->bar_gen != gen; });
  }
}

hipEvent_t new_event(Ctx* c)
{
  hipEvent_t e;
  if (hipEventCreateWithFlags(&e, hipEventDisableTiming) != hipSuccess)
    c->g->failed = 1;
  c->garbage.push_back(e);
  return e;
}

struct Ptrs {
  const void* p[kMaxRanks];
};

template <class T>
__global__ void reduce_kernel(Ptrs in, int n, int64_t count, int op, T* out)
{
  const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= count)
    return;
  T acc = ((const T*)in.p[0])[i];
  for (int r = 1; r < n; ++r) {
    const T v = ((const T*)in.p[r])[i];
    acc = op == 0 ? acc + v : (v > acc ? v : acc);
  }
  out[i] = acc;
}

int inproc_exchange(void* vctx, int ns, const nepmi_msg* sends, int nr, const nepmi_msg* recvs, void* vstream)
{
  Ctx* c = (Ctx*)vctx;
  Group* g = c->g;
  hipStream_t stream = (hipStream_t)vstream;
  std::vector<SendPost*> mine;
  for (int k = 0; k < ns; ++k) {
    SendPost* p = new SendPost();
    p->src = sends[k].buf;
    p->bytes = sends[k].bytes;
    p->ready = new_event(c);
    p->done = new_event(c);
    if (hipEventRecord(p->ready, stream) != hipSuccess)
      return -1;
    {
      std::lock_guard<std::mutex> lk(g->mu);
      g->q[(size_t)c->rank * g->n + sends[k].peer].push_back(p);
    }
    g->cv.notify_all();
    mine.push_back(p);
  }
  for (int k = 0; k < nr; ++k) {
    SendPost* p = nullptr;
    {
      std::unique_lock<std::mutex> lk(g->mu);
      auto& dq = g->q[(size_t)recvs[k].peer * g->n + c->rank];
      g->cv.wait(lk, [&] { return !dq.empty(); });
      p = dq.front();
      dq.pop_front();
    }
    if (p->bytes != recvs[k].bytes)
      return -2; // the two sides disagree about a message
    if (hipStreamWaitEvent(stream, p->ready, 0) != hipSuccess ||
        hipMemcpyAsync(recvs[k].buf, p->src, (size_t)p->bytes, hipMemcpyDeviceToDevice, stream) != hipSuccess ||
        hipEventRecord(p->done, stream) != hipSuccess)
      return -1;
    {
      std::lock_guard<std::mutex> lk(g->mu);
      p->consumed = true;
    }
    g->cv.notify_all();
  }
  for (SendPost* p : mine) {
    {
      std::unique_lock<std::mutex> lk(g->mu);
      g->cv.wait(lk, [&] { return p->consumed; });
    }
    if (hipStreamWaitEvent(stream, p->done, 0) != hipSuccess)
      return -1;
    delete p;
  }
  return 0;
}

int inproc_allreduce(void* vctx, void* buf, int64_t count, int dtype, int op, void* vstream)
{
  Ctx* c = (Ctx*)vctx;
  Group* g = c->g;
  hipStream_t stream = (hipStream_t)vstream;
  const size_t esz = dtype == 1 ? 4 : 8;
  if ((size_t)count * esz > 4096)
    return -3;
  hipEvent_t ready = new_event(c), read = new_event(c);
  if (hipEventRecord(ready, stream) != hipSuccess)
    return -1;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ar_buf[c->rank] = buf;
    g->ar_ready[c->rank] = ready;
    g->ar_read[c->rank] = read;
  }
  barrier(g); // every rank has posted its input
  Ptrs in;
  for (int r = 0; r < g->n; ++r) {
    in.p[r] = g->ar_buf[r];
    if (hipStreamWaitEvent(stream, g->ar_ready[r], 0) != hipSuccess)
      return -1;
  }
  const unsigned grid = (unsigned)((count + 63) / 64);
  if (dtype == 0)
    hipLaunchKernelGGL(reduce_kernel<double>, dim3(grid), dim3(64), 0, stream, in, g->n, count, op, (double*)c->scratch);
  else if (dtype == 1)
    hipLaunchKernelGGL(reduce_kernel<int>, dim3(grid), dim3(64), 0, stream, in, g->n, count, op, (int*)c->scratch);
  else
    hipLaunchKernelGGL(reduce_kernel<long long>, dim3(grid), dim3(64), 0, stream, in, g->n, count, op, (long long*)c->scratch);
  if (hipEventRecord(read, stream) != hipSuccess)
    return -1;
  barrier(g); // every rank has enqueued its reads
  for (int r = 0; r < g->n; ++r)
    if (hipStreamWaitEvent(stream, g->ar_read[r], 0) != hipSuccess)
      return -1;
  if (hipMemcpyAsync(buf, c->scratch, (size_t)count * esz, hipMemcpyDeviceToDevice, stream) != hipSuccess)
    return -1;
  barrier(g); // the slots may be reused
  return g->failed ? -1 : 0;
}

void inproc_destroy(void* vctx)
{
  Ctx* c = (Ctx*)vctx;
  if (!c)
    return;
  (void)hipDeviceSynchronize();
  for (hipEvent_t e : c->garbage)
    (void)hipEventDestroy(e);
  (void)hipFree(c->scratch);
  delete c;
}

} // namespace

extern "C" void* inproc_group_create(int nranks)
{
  if (nranks < 1 || nranks > kMaxRanks)
    return nullptr;
  Group* g = new Group();
  g->n = nranks;
  g->q.resize((size_t)nranks * nranks);
  return g;
}

extern "C" void inproc_group_destroy(void* g) { delete (Group*)g; }

extern "C" int inproc_transport(void* group, int rank, nepmi_transport* out)
{
  Group* g = (Group*)group;
  if (!g || !out || rank < 0 || rank >= g->n)
    return -1;
  Ctx* c = new Ctx();
  c->g = g;
  c->rank = rank;
  if (hipMalloc(&c->scratch, 4096) != hipSuccess) {
    delete c;
    return -1;
  }
  out->ctx = c;
  out->rank = rank;
  out->nranks = g->n;
  out->device_buffers = 1;
  out->exchange = inproc_exchange;
  out->allreduce = inproc_allreduce;
  out->destroy = inproc_destroy;
  return 0;
}
