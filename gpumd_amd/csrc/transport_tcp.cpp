// Host transport of the domain-decomposed driver over TCP sockets (include/nepmi.h: nepmi_transport_tcp).
//
// For ranks that cannot use RCCL: the CPU test tier, several ranks sharing one GPU, a node without xGMI.  Buffers
// are HOST memory (device_buffers = 0: the driver stages payloads through the host).  Rank 0 listens on
// master_addr:port; every rank connects to it and learns the listening ports of the others, then the ranks connect
// pairwise (full mesh: a rank talks to at most 26 neighbours).  exchange() moves all messages of a call concurrently
// with poll(), so that two ranks sending to each other cannot deadlock; all-reduce = gather at rank 0 in rank
// order + broadcast, which gives every rank bit-identical sums.
#include <arpa/inet.h>
#include <netinet/in.h>
#include <netdb.h>
#include <netinet/tcp.h>
#include <poll.h>
#include <sys/socket.h>
#include <unistd.h>

#include <cerrno>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#if defined(__HIPCC__) // the product build: the C-ABI functions are compiled as nepmi_*__impl (capi_dispatch.inc: the public
#define NEPMI_CAPI_RENAME // names are per-handle trampolines); tests/emu compiles this file with g++ and keeps the plain names
#include "capi_dispatch.inc"
#undef NEPMI_CAPI_RENAME
#endif
#include "../../include/nepmi.h"

namespace {

struct TcpCtx {
  int rank = 0, nranks = 1;
  std::vector<int> fd; // socket to every other rank (-1 for self)
};

bool send_all(int fd, const void* buf, size_t n)
{
  const char* p = (const char*)buf;
  while (n > 0) {
    const ssize_t k = ::send(fd, p, n, MSG_NOSIGNAL);
    if (k < 0) {
      if (errno == EINTR)
        continue;
      return false;
    }
    p += k;
    n -= (size_t)k;
  }
  return true;
}
bool recv_all(int fd, void* buf, size_t n)
{
  char* p = (char*)buf;
  while (n > 0) {
    const ssize_t k = ::recv(fd, p, n, 0);
    if (k <= 0) {
      if (k < 0 && errno == EINTR)
        continue;
      return false;
    }
    p += k;
    n -= (size_t)k;
  }
  return true;
}

// dotted quad or host name (MASTER_ADDR=localhost is common under torchrun / mpirun) -> IPv4 address; false if unknown
bool resolve_ipv4(const char* addr, in_addr* out)
{
  if (!addr) {
    out->s_addr = htonl(INADDR_LOOPBACK);
    return true;
  }
  if (inet_pton(AF_INET, addr, out) == 1)
    return true;
  addrinfo hints;
  std::memset(&hints, 0, sizeof hints);
  hints.ai_family = AF_INET;
  hints.ai_socktype = SOCK_STREAM;
  addrinfo* res = nullptr;
  if (getaddrinfo(addr, nullptr, &hints, &res) != 0 || !res)
    return false;
  *out = ((sockaddr_in*)res->ai_addr)->sin_addr;
  freeaddrinfo(res);
  return true;
}

// accept with a deadline: a rank that never starts must not hang the others for ever
int accept_within(int lsock, int timeout_ms)
{
  pollfd pf{lsock, POLLIN, 0};
  for (;;) {
    const int r = ::poll(&pf, 1, timeout_ms);
    if (r < 0 && errno == EINTR)
      continue;
    if (r <= 0)
      return -1;
    return ::accept(lsock, nullptr, nullptr);
  }
}

int listen_on(const char* addr, int port, int* port_out)
{
  const int s = ::socket(AF_INET, SOCK_STREAM, 0);
  if (s < 0)
    return -1;
  int one = 1;
  ::setsockopt(s, SOL_SOCKET, SO_REUSEADDR, &one, sizeof one);
  sockaddr_in a;
  std::memset(&a, 0, sizeof a);
  a.sin_family = AF_INET;
  a.sin_port = htons((uint16_t)port);
  if (!resolve_ipv4(addr, &a.sin_addr)) {
    ::close(s);
    return -1;
  }
  if (::bind(s, (sockaddr*)&a, sizeof a) != 0 || ::listen(s, 128) != 0) {
    ::close(s);
    return -1;
  }
  socklen_t len = sizeof a;
  ::getsockname(s, (sockaddr*)&a, &len);
  *port_out = ntohs(a.sin_port);
  return s;
}

int connect_to(const char* addr, int port)
{
  in_addr ip;
  if (!resolve_ipv4(addr, &ip))
    return -1;
  for (int attempt = 0; attempt < 600; ++attempt) { // the peer may not be listening yet: retry for ~60 s
    const int s = ::socket(AF_INET, SOCK_STREAM, 0);
    if (s < 0)
      return -1;
    sockaddr_in a;
    std::memset(&a, 0, sizeof a);
    a.sin_family = AF_INET;
    a.sin_port = htons((uint16_t)port);
    a.sin_addr = ip;
    if (::connect(s, (sockaddr*)&a, sizeof a) == 0) {
      int one = 1;
      ::setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
      return s;
    }
    ::close(s);
    ::usleep(100000);
  }
  return -1;
}

int tcp_exchange(void* vctx, int ns, const nepmi_msg* sends, int nr, const nepmi_msg* recvs, void*)
{
  TcpCtx* c = (TcpCtx*)vctx;
  // per peer: queue of sends and of recvs, each processed in order (stream sockets keep the order)
  struct Op {
    char* p;
    int64_t left;
    int peer;
    bool is_send;
  };
  std::vector<Op> ops;
  for (int k = 0; k < ns; ++k)
    ops.push_back(Op{(char*)sends[k].buf, sends[k].bytes, sends[k].peer, true});
  for (int k = 0; k < nr; ++k)
    ops.push_back(Op{(char*)recvs[k].buf, recvs[k].bytes, recvs[k].peer, false});
  // messages to oneself: copy send k-th to recv k-th of the same peer
  {
    std::vector<int> srcs, dsts;
    for (int i = 0; i < (int)ops.size(); ++i)
      if (ops[i].peer == c->rank)
        (ops[i].is_send ? srcs : dsts).push_back(i);
    if (srcs.size() != dsts.size())
      return -1;
    for (size_t k = 0; k < srcs.size(); ++k) {
      if (ops[srcs[k]].left != ops[dsts[k]].left)
        return -1;
      std::memcpy(ops[dsts[k]].p, ops[srcs[k]].p, (size_t)ops[srcs[k]].left);
      ops[srcs[k]].left = ops[dsts[k]].left = 0;
    }
  }
  for (;;) {
    // the first unfinished send and the first unfinished recv of every peer are active
    std::vector<pollfd> pf;
    std::vector<int> which;
    std::vector<char> seen_s((size_t)c->nranks, 0), seen_r((size_t)c->nranks, 0);
    for (int i = 0; i < (int)ops.size(); ++i) {
      Op& o = ops[i];
      if (o.left == 0)
        continue;
      std::vector<char>& seen = o.is_send ? seen_s : seen_r;
      if (seen[o.peer])
        continue;
      seen[o.peer] = 1;
      pollfd p;
      p.fd = c->fd[o.peer];
      p.events = o.is_send ? POLLOUT : POLLIN;
      p.revents = 0;
      pf.push_back(p);
      which.push_back(i);
    }
    if (pf.empty())
      return 0;
    int pr;
    do
      pr = ::poll(pf.data(), (nfds_t)pf.size(), 60000);
    while (pr < 0 && errno == EINTR);
    if (pr <= 0)
      return -1;
    for (size_t k = 0; k < pf.size(); ++k) {
      Op& o = ops[which[k]];
      if (pf[k].revents & (POLLERR | POLLHUP | POLLNVAL)) {
        if (!(pf[k].revents & POLLIN))
          return -1;
      }
      if (o.is_send && (pf[k].revents & POLLOUT)) {
        const ssize_t n = ::send(pf[k].fd, o.p, (size_t)(o.left < (1 << 20) ? o.left : (1 << 20)), MSG_NOSIGNAL | MSG_DONTWAIT);
        if (n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
          return -1;
        if (n > 0) {
          o.p += n;
          o.left -= n;
        }
      } else if (!o.is_send && (pf[k].revents & POLLIN)) {
        const ssize_t n = ::recv(pf[k].fd, o.p, (size_t)o.left, MSG_DONTWAIT);
        if (n == 0)
          return -1;
        if (n < 0 && errno != EAGAIN && errno != EWOULDBLOCK && errno != EINTR)
          return -1;
        if (n > 0) {
          o.p += n;
          o.left -= n;
        }
      }
    }
  }
}

template <class T>
void reduce_into(T* acc, const T* in, int64_t n, int op)
{
  for (int64_t i = 0; i < n; ++i)
    acc[i] = op == 0 ? (T)(acc[i] + in[i]) : (in[i] > acc[i] ? in[i] : acc[i]);
}

int tcp_allreduce(void* vctx, void* buf, int64_t count, int dtype, int op, void*)
{
  TcpCtx* c = (TcpCtx*)vctx;
  if (c->nranks == 1)
    return 0;
  const size_t esz = dtype == 1 ? 4 : 8;
  const size_t bytes = esz * (size_t)count;
  if (c->rank == 0) {
    std::vector<char> in(bytes);
    for (int r = 1; r < c->nranks; ++r) { // fixed order: the same sum on every run
      if (!recv_all(c->fd[r], in.data(), bytes))
        return -1;
      if (dtype == 0)
        reduce_into((double*)buf, (const double*)in.data(), count, op);
      else if (dtype == 1)
        reduce_into((int*)buf, (const int*)in.data(), count, op);
      else
        reduce_into((int64_t*)buf, (const int64_t*)in.data(), count, op);
    }
    for (int r = 1; r < c->nranks; ++r)
      if (!send_all(c->fd[r], buf, bytes))
        return -1;
  } else {
    if (!send_all(c->fd[0], buf, bytes) || !recv_all(c->fd[0], buf, bytes))
      return -1;
  }
  return 0;
}

void tcp_destroy(void* vctx)
{
  TcpCtx* c = (TcpCtx*)vctx;
  if (!c)
    return;
  for (int f : c->fd)
    if (f >= 0)
      ::close(f);
  delete c;
}

} // namespace

extern "C" int nepmi_transport_tcp(const char* master_addr, int port, int rank, int nranks, nepmi_transport* out)
{
  if (!out || rank < 0 || rank >= nranks || !master_addr)
    return NEPMI_ERR_ARG;
  TcpCtx* c = new TcpCtx();
  c->rank = rank;
  c->nranks = nranks;
  c->fd.assign((size_t)nranks, -1);
  out->ctx = c;
  out->rank = rank;
  out->nranks = nranks;
  out->device_buffers = 0;
  out->exchange = tcp_exchange;
  out->allreduce = tcp_allreduce;
  out->destroy = tcp_destroy;
  if (nranks == 1)
    return NEPMI_OK;
  // every rank opens its own listening socket (any port); rank 0's is the rendezvous
  int my_port = 0;
  const int lsock = listen_on(master_addr, rank == 0 ? port : 0, &my_port);
  if (lsock < 0) {
    tcp_destroy(c);
    return NEPMI_ERR_IO;
  }
  std::vector<int> ports((size_t)nranks, 0);
  ports[rank] = my_port;
  bool ok = true;
  if (rank == 0) {
    // accept the nranks - 1 others; each first says who it is and where it listens
    for (int k = 1; k < nranks && ok; ++k) {
      const int s = accept_within(lsock, 120000);
      int hello[2] = {0, 0};
      ok = s >= 0 && recv_all(s, hello, sizeof hello) && hello[0] > 0 && hello[0] < nranks && c->fd[hello[0]] < 0;
      if (ok) {
        int one = 1;
        ::setsockopt(s, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        c->fd[hello[0]] = s;
        ports[hello[0]] = hello[1];
      }
    }
    for (int r = 1; r < nranks && ok; ++r)
      ok = send_all(c->fd[r], ports.data(), sizeof(int) * (size_t)nranks);
  } else {
    const int s = connect_to(master_addr, port);
    int hello[2] = {rank, my_port};
    ok = s >= 0 && send_all(s, hello, sizeof hello) && recv_all(s, ports.data(), sizeof(int) * (size_t)nranks);
    c->fd[0] = s;
    // pairwise: the lower rank accepts, the higher one connects
    for (int r = 1; r < rank && ok; ++r) {
      const int t = connect_to(master_addr, ports[r]);
      int who = rank;
      ok = t >= 0 && send_all(t, &who, sizeof who);
      c->fd[r] = t;
    }
    for (int k = rank + 1; k < nranks && ok; ++k) {
      const int t = accept_within(lsock, 120000);
      int who = -1;
      ok = t >= 0 && recv_all(t, &who, sizeof who) && who > rank && who < nranks && c->fd[who] < 0;
      if (ok) {
        int one = 1;
        ::setsockopt(t, IPPROTO_TCP, TCP_NODELAY, &one, sizeof one);
        c->fd[who] = t;
      }
    }
  }
  ::close(lsock);
  if (!ok) {
    tcp_destroy(c);
    out->ctx = nullptr;
    return NEPMI_ERR_IO;
  }
  return NEPMI_OK;
}
