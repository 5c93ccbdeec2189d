// TEST-ONLY double of the ten RCCL entry points the engine resolves (csrc/hxv_comm.cpp: rccl()) for ranks that are separate PROCESSES
// sharing ONE GPU -- the process model of `python -m torch.distributed.run --nproc-per-node N bench.py --gpus N`, which RCCL itself refuses
// on one device.  Companion of rccl_double.cpp (thread ranks of one process).  Loaded through HXV_RCCL_LIB; never linked into the product.
//
// Transport: POSIX shared memory named after the unique id.  A sender stages its data device -> host into its window of the segment and posts
// (offset, bytes) in the mailbox of the (source, destination) pair; the receiver copies host -> device and acknowledges.  Every call blocks
// the host thread until its part is done (real RCCL is asynchronous; the ORDER of effects on the caller's stream is the same).  Kept from
// NCCL: a send matches the peer's next receive from this rank in issue order and their byte counts must agree (ncclInvalidArgument),
// operations between ncclGroupStart / ncclGroupEnd are issued together (all sends are posted before any receive waits), collectives are
// entered by every rank.  Limits: up to 16 messages in flight per ordered pair and HXV_RCCL_MP_MB (default 256) MiB of staging per rank
// and group -- a rehearsal transport for small sectors, not a data path.
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <vector>

namespace {

constexpr double TIMEOUT_S = 120.0;
constexpr int QDEPTH = 16;
constexpr int MAXRANKS = 16;

struct Slot {
  uint64_t off, bytes;
};
struct Mailbox {  // ordered pair (src, dst): src posts, dst consumes
  std::atomic<uint64_t> posted, consumed;
  Slot slot[QDEPTH];
};
struct Header {
  std::atomic<int> joined, left;
  int n;
  uint64_t window_bytes;
  Mailbox box[MAXRANKS][MAXRANKS];
};

struct Comm {
  Header* hd = nullptr;
  char* data = nullptr;  // n windows of window_bytes behind the header
  size_t map_bytes = 0;
  int n = 0, rank = 0;
  std::string name;
  uint64_t cursor = 0;  // staging cursor of the group under way (this rank's window)
};

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

template <typename F>
bool spin_until(F cond) {
  const auto t0 = std::chrono::steady_clock::now();
  int it = 0;
  while (!cond()) {
    if ((++it & 1023) == 0) {
      sched_yield();
      if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > TIMEOUT_S) return false;
    }
  }
  return true;
}

#define HIPOK(expr)                                           \
  do {                                                        \
    if ((expr) != hipSuccess) return ncclUnhandledCudaError;  \
  } while (0)

ncclResult_t run_group(std::vector<Op>& ops) {
  ncclResult_t res = ncclSuccess;
  // 1. every send: stage device -> host into this rank's window, post the slot
  std::vector<uint64_t> my_seq(ops.size(), 0);
  for (size_t i = 0; i < ops.size(); ++i) {
    Op& o = ops[i];
    if (!o.send) continue;
    Comm& c = *o.c;
    Mailbox& mb = c.hd->box[c.rank][o.peer];
    const uint64_t seq = mb.posted.load(std::memory_order_relaxed);
    if (!spin_until([&] { return seq - mb.consumed.load(std::memory_order_acquire) < QDEPTH; })) return ncclSystemError;
    if (c.cursor + o.bytes > c.hd->window_bytes) return ncclInternalError;  // staging window too small: raise HXV_RCCL_MP_MB
    char* dst = c.data + (size_t)c.rank * c.hd->window_bytes + c.cursor;
    HIPOK(hipStreamSynchronize(o.st));  // what is sent has been produced
    if (o.bytes) HIPOK(hipMemcpy(dst, o.buf, o.bytes, hipMemcpyDeviceToHost));
    mb.slot[seq % QDEPTH] = Slot{c.cursor, o.bytes};
    c.cursor += (o.bytes + 255) & ~(uint64_t)255;
    mb.posted.store(seq + 1, std::memory_order_release);
    my_seq[i] = seq + 1;
  }
  // 2. every receive: the peer's next message towards this rank
  for (Op& o : ops) {
    if (o.send) continue;
    Comm& c = *o.c;
    Mailbox& mb = c.hd->box[o.peer][c.rank];
    const uint64_t seq = mb.consumed.load(std::memory_order_relaxed);
    if (!spin_until([&] { return mb.posted.load(std::memory_order_acquire) > seq; })) return ncclSystemError;
    const Slot s = mb.slot[seq % QDEPTH];
    if (s.bytes != o.bytes) res = ncclInvalidArgument;  // the two ranks disagree about the size of this block
    const size_t nb = (size_t)std::min<uint64_t>(s.bytes, o.bytes);
    if (nb) {
      HIPOK(hipMemcpyAsync(o.buf, c.data + (size_t)o.peer * c.hd->window_bytes + s.off, nb, hipMemcpyHostToDevice, o.st));
      HIPOK(hipStreamSynchronize(o.st));  // (the window may be reused as soon as the message is acknowledged)
    }
    mb.consumed.store(seq + 1, std::memory_order_release);
  }
  // 3. my sends have been read: the window is free for the next group
  for (size_t i = 0; i < ops.size(); ++i) {
    if (!ops[i].send) continue;
    Comm& c = *ops[i].c;
    Mailbox& mb = c.hd->box[c.rank][ops[i].peer];
    if (!spin_until([&] { return mb.consumed.load(std::memory_order_acquire) >= my_seq[i]; })) return ncclSystemError;
  }
  for (Op& o : ops) o.c->cursor = 0;
  return res;
}

ncclResult_t p2p(bool send, const void* buf, size_t count, ncclDataType_t t, int peer, ncclComm_t comm, hipStream_t st) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t tb = type_bytes(t);
  if (!c || !tb || peer < 0 || peer >= c->n || peer == c->rank || (!buf && count)) return ncclInvalidArgument;
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
  static std::atomic<unsigned> counter{0};
  std::memset(id, 0, sizeof(*id));
  std::snprintf(id->internal, sizeof(id->internal), "hxvdbl_%d_%u_%llx", (int)getpid(), counter.fetch_add(1),
                (unsigned long long)std::chrono::steady_clock::now().time_since_epoch().count());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* comm, int nranks, ncclUniqueId id, int rank) {
  if (!comm || nranks < 1 || nranks > MAXRANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  size_t window = 256;
  if (const char* e = getenv("HXV_RCCL_MP_MB")) window = (size_t)std::max(1, atoi(e));
  window <<= 20;
  Comm* c = new Comm();
  c->n = nranks;
  c->rank = rank;
  c->name = std::string("/") + std::string(id.internal, strnlen(id.internal, sizeof(id.internal)));
  c->map_bytes = sizeof(Header) + window * (size_t)nranks;
  const int fd = shm_open(c->name.c_str(), O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) {  // (a fresh segment is zero-filled: all counters start at 0)
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->hd = static_cast<Header*>(p);
  c->data = static_cast<char*>(p) + sizeof(Header);
  c->hd->n = nranks;
  c->hd->window_bytes = window;
  c->hd->joined.fetch_add(1, std::memory_order_acq_rel);
  if (!spin_until([&] { return c->hd->joined.load(std::memory_order_acquire) >= nranks; })) {
    munmap(p, c->map_bytes);
    delete c;
    return ncclSystemError;
  }
  *comm = reinterpret_cast<ncclComm_t>(c);
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t comm) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c) return ncclSuccess;
  const bool last = c->hd->left.fetch_add(1, std::memory_order_acq_rel) + 1 == c->n;
  munmap(c->hd, c->map_bytes);
  if (last) shm_unlink(c->name.c_str());
  delete c;
  return ncclSuccess;
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

// the collectives are groups of pairwise messages among all ranks
ncclResult_t ncclAllGather(const void* sendbuff, void* recvbuff, size_t sendcount, ncclDataType_t datatype, ncclComm_t comm, hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  const size_t tb = type_bytes(datatype);
  if (!c || !tb || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t bytes = sendcount * tb;
  char* mine = static_cast<char*>(recvbuff) + (size_t)c->rank * bytes;
  if (mine != sendbuff && bytes) HIPOK(hipMemcpyAsync(mine, sendbuff, bytes, hipMemcpyDeviceToDevice, stream));  // (in place: nothing to do)
  std::vector<Op> ops;
  for (int p = 0; p < c->n; ++p)
    if (p != c->rank) ops.push_back(Op{true, c, p, const_cast<void*>(sendbuff), bytes, stream});
  for (int p = 0; p < c->n; ++p)
    if (p != c->rank) ops.push_back(Op{false, c, p, static_cast<char*>(recvbuff) + (size_t)p * bytes, bytes, stream});
  return ops.empty() ? ncclSuccess : run_group(ops);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op, ncclComm_t comm,
                           hipStream_t stream) {
  Comm* c = reinterpret_cast<Comm*>(comm);
  if (!c || datatype != ncclFloat64 || (op != ncclSum && op != ncclMax) || !sendbuff || !recvbuff) return ncclInvalidArgument;
  const size_t bytes = count * sizeof(double);
  double* d_all = nullptr;
  HIPOK(hipMalloc((void**)&d_all, bytes * (size_t)c->n));
  ncclResult_t r = ncclAllGather(sendbuff, d_all, count, ncclFloat64, comm, stream);
  std::vector<double> all(count * (size_t)c->n);
  if (r == ncclSuccess && hipMemcpyAsync(all.data(), d_all, bytes * (size_t)c->n, hipMemcpyDeviceToHost, stream) != hipSuccess) r = ncclUnhandledCudaError;
  if (r == ncclSuccess && hipStreamSynchronize(stream) != hipSuccess) r = ncclUnhandledCudaError;
  (void)hipFree(d_all);
  if (r != ncclSuccess) return r;
  std::vector<double> tot(count);
  for (size_t i = 0; i < count; ++i) {
    double t = all[i];  // rank order: the same bits on every rank
    for (int p = 1; p < c->n; ++p) t = op == ncclSum ? t + all[(size_t)p * count + i] : std::max(t, all[(size_t)p * count + i]);
    tot[i] = t;
  }
  HIPOK(hipMemcpyAsync(recvbuff, tot.data(), bytes, hipMemcpyHostToDevice, stream));
  HIPOK(hipStreamSynchronize(stream));
  return ncclSuccess;
}

const char* ncclGetErrorString(ncclResult_t result) {
  switch (result) {
    case ncclSuccess: return "no error";
    case ncclUnhandledCudaError: return "rccl process double: HIP call failed";
    case ncclSystemError: return "rccl process double: a peer did not arrive in time, or the shared segment could not be made";
    case ncclInternalError: return "rccl process double: staging window too small (HXV_RCCL_MP_MB)";
    case ncclInvalidArgument: return "rccl process double: invalid argument (peers disagree about a count, or a bad pointer / rank / type)";
    case ncclInvalidUsage: return "rccl process double: invalid usage";
    default: return "rccl process double: error";
  }
}

}  // extern "C"
