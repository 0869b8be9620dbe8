// TEST INFRASTRUCTURE (CPU tier): the host twin of inproc_transport.hip for the kernel-logic emulator, where "device"
// memory is host memory and a stream is nothing: the same queues and barriers, plain memcpy instead of events and
// device copies.  It reports device_buffers = 1, so the decomposed driver takes the code path of a device transport
// (speculative enqueue, device-side vote, no host staging) with several ranks -- as threads of one process.
#include <condition_variable>
#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <deque>
#include <mutex>
#include <vector>

#include "../../include/nepmi.h"

namespace {

constexpr int kMaxRanks = 8;

struct SendPost {
  const void* src;
  int64_t bytes;
  bool consumed = false;
};

struct Group {
  int n;
  std::mutex mu;
  std::condition_variable cv;
  std::vector<std::deque<SendPost*>> q;
  int bar_count = 0;
  long bar_gen = 0;
  void* ar_buf[kMaxRanks];
};

struct Ctx {
  Group* g;
  int rank;
  int calls = 0;
  unsigned char scratch[4096];
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

int host_exchange(void* vctx, int ns, const nepmi_msg* sends, int nr, const nepmi_msg* recvs, void*)
{
  Ctx* c = (Ctx*)vctx;
  Group* g = c->g;
  if (std::getenv("INPROC_TRACE")) {
    std::fprintf(stderr, "[r%d] exchange #%d ns=%d nr=%d", c->rank, c->calls++, ns, nr);
    for (int k = 0; k < ns; ++k) std::fprintf(stderr, " s(%d,%lld)", sends[k].peer, (long long)sends[k].bytes);
    for (int k = 0; k < nr; ++k) std::fprintf(stderr, " r(%d,%lld)", recvs[k].peer, (long long)recvs[k].bytes);
    std::fprintf(stderr, "\n");
  }
  std::vector<SendPost*> mine;
  for (int k = 0; k < ns; ++k) {
    SendPost* p = new SendPost();
    p->src = sends[k].buf;
    p->bytes = sends[k].bytes;
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
      return -2;
    std::memcpy(recvs[k].buf, p->src, (size_t)p->bytes);
    {
      std::lock_guard<std::mutex> lk(g->mu);
      p->consumed = true;
    }
    g->cv.notify_all();
  }
  for (SendPost* p : mine) {
    std::unique_lock<std::mutex> lk(g->mu);
    g->cv.wait(lk, [&] { return p->consumed; });
    lk.unlock();
    delete p;
  }
  return 0;
}

template <class T>
void reduce(Group* g, int64_t count, int op, T* out)
{
  for (int64_t i = 0; i < count; ++i) {
    T acc = ((const T*)g->ar_buf[0])[i];
    for (int r = 1; r < g->n; ++r) {
      const T v = ((const T*)g->ar_buf[r])[i];
      acc = op == 0 ? acc + v : (v > acc ? v : acc);
    }
    out[i] = acc;
  }
}

int host_allreduce(void* vctx, void* buf, int64_t count, int dtype, int op, void*)
{
  Ctx* c = (Ctx*)vctx;
  Group* g = c->g;
  const size_t esz = dtype == 1 ? 4 : 8;
  if (std::getenv("INPROC_TRACE"))
    std::fprintf(stderr, "[r%d] allreduce #%d count=%lld dtype=%d op=%d\n", c->rank, c->calls++, (long long)count, dtype, op);
  if ((size_t)count * esz > sizeof(c->scratch))
    return -3;
  {
    std::lock_guard<std::mutex> lk(g->mu);
    g->ar_buf[c->rank] = buf;
  }
  barrier(g);
  if (dtype == 0)
    reduce<double>(g, count, op, (double*)c->scratch);
  else if (dtype == 1)
    reduce<int>(g, count, op, (int*)c->scratch);
  else
    reduce<long long>(g, count, op, (long long*)c->scratch);
  barrier(g);
  std::memcpy(buf, c->scratch, (size_t)count * esz);
  barrier(g);
  return 0;
}

void host_destroy(void* vctx) { delete (Ctx*)vctx; }

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
  out->ctx = c;
  out->rank = rank;
  out->nranks = g->n;
  out->device_buffers = 1;
  out->exchange = host_exchange;
  out->allreduce = host_allreduce;
  out->destroy = host_destroy;
  return 0;
}
