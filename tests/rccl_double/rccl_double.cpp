// TEST-ONLY double of the RCCL entry points the engine resolves (ten, plus ncclCommAbort) (csrc/hxv_comm.cpp: rccl()), for ranks that are host THREADS of one
// process sharing one GPU -- RCCL itself refuses two ranks on one device, and no round of this project has had a second GPU.  It exists
// so that the RCCL branches of hxv_comm.cpp (the grouped ncclSend / ncclRecv of the column exchange and of both transposes, the in-place
// ncclAllGather, the ncclSum / ncclMax all-reduces) EXECUTE with 2-4 ranks and are checked against the oracle: counts, offsets and
// pointers are the engine's, only the transport underneath is replaced.  Loaded through HXV_RCCL_LIB; never linked into the product.
//
// Semantics kept from NCCL: calls are ordered on the caller's stream; a send matches the peer's next receive from this rank, in issue
// order, and their byte counts must agree (ncclInvalidArgument otherwise); operations between ncclGroupStart / ncclGroupEnd are issued
// together, so a rank may post sends and receives towards several peers without deadlock; collectives are entered by every rank of the
// communicator.  Unlike NCCL the calls block the host thread until the peers have issued their side (device work stays asynchronous).
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>

#include <atomic>
#include <chrono>
#include <condition_variable>
#include <cstring>
#include <deque>
#include <map>
#include <memory>
#include <mutex>
#include <string>
#include <vector>

namespace {

constexpr double TIMEOUT_S = 120.0;

struct Msg {
  const void* buf = nullptr;
  size_t bytes = 0;
  hipEvent_t ready = nullptr, consumed = nullptr;
  bool taken = false;
  ~Msg() {
    if (ready) (void)hipEventDestroy(ready);
    if (consumed) (void)hipEventDestroy(consumed);
  }
};

struct World {
  int n = 0;
  std::string key;
  std::mutex mu;
  std::condition_variable cv;
  int joined = 0, alive = 0;
  int arrived = 0;
  uint64_t gen = 0;
  bool broken = false;  // ncclCommAbort on any of its communicators: every wait gives up
  struct Post {
    const void* send = nullptr;
    void* recv = nullptr;
    size_t bytes = 0;
  };
  std::vector<Post> post;                           // by rank: the collective under way
  std::vector<std::vector<double>> red;             // by rank: all-reduce contributions
  std::vector<hipEvent_t> ready, done;              // by rank
  std::vector<std::deque<std::shared_ptr<Msg>>> q;  // [src * n + dst]: sends not yet matched
  // every rank of the world calls it; false on timeout
  bool barrier() {
    std::unique_lock<std::mutex> lk(mu);
    const uint64_t g = gen;
    if (++arrived == n) {
      arrived = 0;
      ++gen;
      cv.notify_all();
      return true;
    }
    return cv.wait_for(lk, std::chrono::duration<double>(TIMEOUT_S), [&] { return gen != g || broken; }) && !broken;
  }
};

struct Comm {
  std::shared_ptr<World> w;
  int rank = 0;
};

std::mutex g_mu;
std::map<std::string, std::shared_ptr<World>> g_worlds;
std::atomic<uint64_t> g_next{1};

struct Op {
  bool send;
  Comm* c;
  int peer;
  void* buf;
  size_t bytes;
  hipStream_t st;
};
thread_local int tl_depth = 0;
thread_local std::vector<Op> tl_ops;

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclFloat16: case ncclBfloat16: return 2;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

#define HIPOK(expr)                                  \
  do {                                               \
    if ((expr) != hipSuccess) return ncclUnhandledCudaError; \
  } while (0)

ncclResult_t run_group(std::vector<Op>& ops) {
  ncclResult_t res = ncclSuccess;
  std::vector<std::shared_ptr<Msg>> mine(ops.size());
  // 1. post every send first: no rank waits for anything before its own offers are visible
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& o = ops[i];
    if (!o.send) continue;
    auto m = std::make_shared<Msg>();
    m->buf = o.buf;
    m->bytes = o.bytes;
    HIPOK(hipEventCreateWithFlags(&m->ready, hipEventDisableTiming));
    HIPOK(hipEventCreateWithFlags(&m->consumed, hipEventDisableTiming));
    HIPOK(hipEventRecord(m->ready, o.st));
    World& w = *o.c->w;
    {
      std::lock_guard<std::mutex> lk(w.mu);
      w.q[(size_t)o.c->rank * w.n + o.peer].push_back(m);
    }
    w.cv.notify_all();
    mine[i] = m;
  }
  // 2. receives: take the peer's next send towards this rank
  for (Op& o : ops) {
    if (o.send) continue;
    World& w = *o.c->w;
    std::shared_ptr<Msg> m;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      auto& dq = w.q[(size_t)o.peer * w.n + o.c->rank];
      if (!w.cv.wait_for(lk, std::chrono::duration<double>(TIMEOUT_S), [&] { return !dq.empty() || w.broken; }) || w.broken) return ncclSystemError;
      m = dq.front();
      dq.pop_front();
    }
    if (m->bytes != o.bytes) res = ncclInvalidArgument;  // the two ranks disagree about the size of this block
    const size_t nb = std::min(m->bytes, o.bytes);
    HIPOK(hipStreamWaitEvent(o.st, m->ready, 0));
    if (nb) HIPOK(hipMemcpyAsync(o.buf, m->buf, nb, hipMemcpyDeviceToDevice, o.st));
    HIPOK(hipEventRecord(m->consumed, o.st));
    {
      std::lock_guard<std::mutex> lk(w.mu);
      m->taken = true;
    }
    w.cv.notify_all();
  }
  // 3. my send buffers are free again (for later work on my stream) once the receivers have read them
  for (size_t i = 0; i < ops.size(); ++i) {
    if (!mine[i]) continue;
    World& w = *ops[i].c->w;
    {
      std::unique_lock<std::mutex> lk(w.mu);
      if (!w.cv.wait_for(lk, std::chrono::duration<double>(TIMEOUT_S), [&] { return mine[i]->taken || w.broken; }) || w.broken) return ncclSystemError;
    }
    HIPOK(hipStreamWaitEvent(ops[i].st, mine[i]->consumed, 0));
  }
  return res;
}

ncclResult_t p2p(bool send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t tb = type_bytes(t);
  if (!c || !tb || peer < 0 || peer >= c->w->n || peer == c->rank || (!buf && count)) return ncclInvalidArgument;
  tl_ops.push_back(Op{send, c, peer, const_cast<void*>(buf), count * tb, st});
  if (tl_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(tl_ops);
  return run_group(ops);
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "hxv-rccl-double-%llu", (unsigned long long)g_next.fetch_add(1));
  return ncclSuccess;
}

// TEST HOOK (not an RCCL entry point): how many ncclCommInitRank / ncclCommDestroy calls this library has served since it was loaded --
// what tests/test_gpu_ranks.py reads to show that a sweep of many sectors builds ONE communicator per rank.
static std::atomic<long long> g_n_init{0}, g_n_destroy{0};
long long rccl_double_comm_inits() { return g_n_init.load(); }
long long rccl_double_comm_destroys() { return g_n_destroy.load(); }

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  ++g_n_init;
  const std::string key(id.internal, sizeof(id.internal));
  std::shared_ptr<World> w;
  {
    std::lock_guard<std::mutex> lk(g_mu);
    auto& slot = g_worlds[key];
    if (!slot) {
      slot = std::make_shared<World>();
      slot->n = nranks;
      slot->key = key;
      slot->post.resize(nranks);
      slot->red.resize(nranks);
      slot->ready.assign(nranks, nullptr);
      slot->done.assign(nranks, nullptr);
      slot->q.resize((size_t)nranks * nranks);
    }
    w = slot;
  }
  if (w->n != nranks) return ncclInvalidArgument;
  HIPOK(hipEventCreateWithFlags(&w->ready[rank], hipEventDisableTiming));
  HIPOK(hipEventCreateWithFlags(&w->done[rank], hipEventDisableTiming));
  {
    std::unique_lock<std::mutex> lk(w->mu);
    ++w->joined;
    ++w->alive;
    w->cv.notify_all();
    if (!w->cv.wait_for(lk, std::chrono::duration<double>(TIMEOUT_S), [&] { return w->joined >= w->n; })) return ncclSystemError;
  }
  Comm* c = new Comm();
  c->w = w;
  c->rank = rank;
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  ++g_n_destroy;
  std::shared_ptr<World> w = c->w;
  bool last;
  {
    std::lock_guard<std::mutex> lk(w->mu);
    last = --w->alive == 0;
  }
  if (last) {
    for (auto e : w->ready)
      if (e) (void)hipEventDestroy(e);
    for (auto e : w->done)
      if (e) (void)hipEventDestroy(e);
    std::lock_guard<std::mutex> lk(g_mu);
    g_worlds.erase(w->key);
  }
  delete c;
  return ncclSuccess;
}

// ncclCommAbort: any thread may call it while the communicator's own rank waits in a collective.  The double marks the WORLD broken (a
// lost rank breaks every collective of the communicator anyway) and leaves the Comm object alone: its rank may be reading it right now.
ncclResult_t ncclCommAbort(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  {
    std::lock_guard<std::mutex> lk(c->w->mu);
    c->w->broken = true;
  }
  c->w->cv.notify_all();
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t tb = type_bytes(datatype);
  if (!c || !tb || !sendbuff || !recvbuff) return ncclInvalidArgument;
  World& w = *c->w;
  const size_t bytes = sendcount * tb;
  const int r = c->rank;
  w.post[r] = World::Post{sendbuff, recvbuff, bytes};
  HIPOK(hipEventRecord(w.ready[r], stream));
  if (!w.barrier()) return ncclSystemError;
  ncclResult_t res = ncclSuccess;
  for (int p = 0; p < w.n; ++p) {
    if (w.post[p].bytes != bytes) res = ncclInvalidArgument;  // every rank contributes the same count
    char* dst = static_cast<char*>(recvbuff) + (size_t)p * bytes;
    if (p == r) {
      if (dst != sendbuff && bytes) HIPOK(hipMemcpyAsync(dst, sendbuff, bytes, hipMemcpyDeviceToDevice, stream));  // (in place: nothing to do)
      continue;
    }
    HIPOK(hipStreamWaitEvent(stream, w.ready[p], 0));
    if (bytes && res == ncclSuccess) HIPOK(hipMemcpyAsync(dst, w.post[p].send, bytes, hipMemcpyDeviceToDevice, stream));
  }
  HIPOK(hipEventRecord(w.done[r], stream));
  if (!w.barrier()) return ncclSystemError;
  for (int p = 0; p < w.n; ++p)
    if (p != r) HIPOK(hipStreamWaitEvent(stream, w.done[p], 0));
  return res;
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || datatype != ncclFloat64 || (op != ncclSum && op != ncclMax) || !sendbuff || !recvbuff) return ncclInvalidArgument;
  World& w = *c->w;
  const int r = c->rank;
  std::vector<double> mine(count);
  HIPOK(hipMemcpyAsync(mine.data(), sendbuff, count * sizeof(double), hipMemcpyDeviceToHost, stream));
  HIPOK(hipStreamSynchronize(stream));
  w.red[r] = mine;
  if (!w.barrier()) return ncclSystemError;
  ncclResult_t res = ncclSuccess;
  std::vector<double> tot(count, 0.0);
  for (int p = 0; p < w.n; ++p) {
    if (w.red[p].size() != count) {
      res = ncclInvalidArgument;
      continue;
    }
    for (size_t i = 0; i < count; ++i) tot[i] = (p == 0) ? w.red[p][i] : (op == ncclSum ? tot[i] + w.red[p][i] : std::max(tot[i], w.red[p][i]));
  }
  if (!w.barrier()) return ncclSystemError;
  HIPOK(hipMemcpyAsync(recvbuff, tot.data(), count * sizeof(double), hipMemcpyHostToDevice, stream));
  HIPOK(hipStreamSynchronize(stream));
  return res;
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return p2p(true, sendbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t comm, hipStream_t stream) {
  return p2p(false, recvbuff, count, datatype, peer, comm, stream);
}

ncclResult_t ncclGroupStart() {
  ++tl_depth;
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (tl_depth <= 0) return ncclInvalidUsage;
  if (--tl_depth > 0) return ncclSuccess;
  std::vector<Op> ops;
  ops.swap(tl_ops);
  return ops.empty() ? ncclSuccess : run_group(ops);
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl double: HIP call failed";
    case ncclSystemError: return "rccl double: a peer did not arrive in time";
    case ncclInvalidArgument: return "rccl double: invalid argument (peers disagree about a count, or a bad pointer / rank / type)";
    case ncclInvalidUsage: return "rccl double: invalid usage";
    default: return "rccl double: error";
  }
}

}  // extern "C"
