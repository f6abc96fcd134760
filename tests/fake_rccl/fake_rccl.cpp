// fake_rccl.cpp -- TEST INFRASTRUCTURE, never shipped, never loaded by the product unless DSEA_RCCL_LIB names it.
//
// A stand-in for the ten RCCL entry points libdsea binds at run time (csrc/dsea_partitioned.hip: rccl_init), for N
// PROCESSES SHARING ONE GPU -- the situation real RCCL refuses ("two ranks on one device") and the only one a one-GPU
// box can offer.  With it the COMM_RCCL_OWNED branch of the row-partitioned solvers (unique ids, ncclCommInitRank of
// two communicators, the ncclGroupStart / ncclSend / ncclRecv / ncclGroupEnd all-to-all and pair exchange, ncclAllReduce
// on the solver stream while the exchange communicator works on the side stream) executes at world 2 / 4 / 8.
//
// Transport: one POSIX shared-memory segment per communicator (its name travels in the ncclUniqueId).  Every rank
// pair owns a one-slot mailbox (sequence counters `sent` / `recvd`, SLOT_BYTES of payload; longer messages go through
// it in pieces, all operations of a group progressed round-robin so that two ranks sending to each other cannot
// block one another).  All-reduce: every rank deposits its contribution, barrier, every rank adds the P deposits in
// RANK ORDER (so the result is bit-identical on all ranks and reproducible), barrier.
//
// Stream semantics (what a caller of RCCL may rely on): an operation starts after everything enqueued earlier on its
// stream and is finished before anything enqueued later on that stream starts.
//   * blocking mode (default): the API call synchronises the stream, moves the bytes, returns.
//   * asynchronous mode (FAKE_RCCL_ASYNC=1): the call records an event, hands the operation to the communicator's
//     progress thread and enqueues a device-side wait (hipStreamWaitValue64) for its completion word; the host returns
//     at once, two communicators on two streams really are in flight together, and a dependency the CALLER forgot
//     (reading a receive buffer without joining the exchange stream) is not hidden by a host-side synchronisation.
//
// Fault injection for the bench watchdog tests: FAKE_RCCL_HANG=<kind>[@comm<i>][:<after>] with kind = allreduce | p2p
// makes the (after+1)-th operation of that kind (on the i-th communicator this process created, default: any) sleep
// forever on the host, on the ranks FAKE_RCCL_HANG_RANK selects (default: all).
// FAKE_RCCL_FAIL=<allreduce|send>[:<after>] makes the (after+1)-th call of that kind return ncclInternalError, once, on
// every rank alike (no operation is recorded): what the caller does with a failed call inside an open group.
#include <fcntl.h>
#include <hip/hip_runtime_api.h>
#include <rccl/rccl.h>
#include <sched.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include <sys/mman.h>
#include <sys/stat.h>
#include <time.h>
#include <unistd.h>

#include <atomic>
#include <condition_variable>
#include <deque>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

namespace {

constexpr uint32_t MAGIC = 0xFA4ECC1u;
constexpr size_t SLOT_BYTES = 1u << 20;          // mailbox payload
constexpr size_t AR_BYTES = 1u << 16;            // all-reduce deposit per rank and round
constexpr int MAX_RANKS = 64;

struct Mailbox {
  std::atomic<uint64_t> sent, recvd;
  char pad[64 - 2 * sizeof(std::atomic<uint64_t>)];
  char data[SLOT_BYTES];
};
struct Header {
  std::atomic<uint32_t> magic;
  std::atomic<int32_t> attached, detached;
  std::atomic<uint64_t> barrier;
  char pad[4096 - 24];
};
static_assert(sizeof(Header) == 4096, "header page");

struct Op {
  int kind;            // 0 send, 1 recv, 2 all-reduce
  const void* src;
  void* dst;
  size_t bytes, done;  // p2p: payload and progress
  int peer;
  size_t count;        // all-reduce
  ncclDataType_t dtype;
};
struct Work {
  std::vector<Op> ops;
  hipEvent_t ready;
  uint64_t seq;
};

}  // namespace

struct ncclComm {
  int rank, world, device, index;
  char name[64];
  size_t bytes;
  char* base;
  Header* hdr;
  char* ar;            // world deposits of AR_BYTES
  Mailbox* box;        // world * world, [src * world + dst]
  uint64_t barriers;
  // asynchronous mode
  bool async;
  hipStream_t pstream;
  uint64_t* done_word;         // signal memory: last finished sequence number
  uint64_t issued;
  std::thread worker;
  std::mutex mu;
  std::condition_variable cv;
  std::deque<Work> queue;
  bool stop;
  std::atomic<int> failed;
};

namespace {

std::atomic<uint64_t> g_stats[8];     // 0 all-reduce 1 send 2 recv 3 group_end 4 comms created 5 async ops 6 bytes p2p
std::atomic<int> g_comm_index{0};
thread_local int t_group_depth = 0;
struct Pending {
  ncclComm* comm;
  hipStream_t stream;
  Op op;
};
thread_local std::vector<Pending>* t_pending = nullptr;

double now_s() {
  timespec ts;
  clock_gettime(CLOCK_MONOTONIC, &ts);
  return ts.tv_sec + 1e-9 * ts.tv_nsec;
}
double timeout_s() {
  static double v = [] {
    const char* e = getenv("FAKE_RCCL_TIMEOUT_S");
    return e ? atof(e) : 300.0;
  }();
  return v;
}
void relax(int& spins) {
  if (++spins < 200) {
    sched_yield();
  } else {
    usleep(50);
  }
}

// ---- fault injection ------------------------------------------------------------------------------------------------
struct Hang {
  int kind = -1;        // 0 all-reduce, 1 p2p
  int comm_index = -1;  // -1 any
  long after = 0;
  int rank = -1;        // -1 all
  std::atomic<long> seen{0};
};
Hang& hang_cfg() {
  static Hang h;
  static std::once_flag once;
  std::call_once(once, [] {
    const char* e = getenv("FAKE_RCCL_HANG");
    if (!e || !*e) return;
    std::string s(e);
    size_t colon = s.find(':');
    if (colon != std::string::npos) {
      h.after = atol(s.c_str() + colon + 1);
      s.resize(colon);
    }
    size_t at = s.find("@comm");
    if (at != std::string::npos) {
      h.comm_index = atoi(s.c_str() + at + 5);
      s.resize(at);
    }
    if (s == "allreduce") h.kind = 0;
    if (s == "p2p") h.kind = 1;
    const char* r = getenv("FAKE_RCCL_HANG_RANK");
    if (r && *r) h.rank = atoi(r);
  });
  return h;
}
void maybe_hang(ncclComm* c, int kind) {
  Hang& h = hang_cfg();
  if (h.kind != kind) return;
  if (h.comm_index >= 0 && h.comm_index != c->index) return;
  if (h.rank >= 0 && h.rank != c->rank) return;
  if (h.seen.fetch_add(1) < h.after) return;
  fprintf(stderr, "[fake_rccl] rank %d: injected hang (%s on communicator %d)\n", c->rank, kind == 0 ? "allreduce" : "p2p",
          c->index);
  fflush(stderr);
  for (;;) pause();
}

bool injected_failure(int kind) {      // kind 0 all-reduce, 1 send
  static int want = -2;
  static long after = 0;
  static std::atomic<long> seen{0};
  static std::once_flag once;
  std::call_once(once, [] {
    want = -1;
    const char* e = getenv("FAKE_RCCL_FAIL");
    if (!e || !*e) return;
    std::string s(e);
    size_t colon = s.find(':');
    if (colon != std::string::npos) {
      after = atol(s.c_str() + colon + 1);
      s.resize(colon);
    }
    if (s == "allreduce") want = 0;
    if (s == "send") want = 1;
  });
  return want == kind && seen.fetch_add(1) == after;
}

// ---- shared-memory protocol --------------------------------------------------------------------------------------------
bool barrier(ncclComm* c) {
  const uint64_t target = (++c->barriers) * (uint64_t)c->world;
  c->hdr->barrier.fetch_add(1, std::memory_order_acq_rel);
  const double t0 = now_s();
  int spins = 0;
  while (c->hdr->barrier.load(std::memory_order_acquire) < target) {
    relax(spins);
    if ((spins & 1023) == 0 && now_s() - t0 > timeout_s()) return false;
  }
  return true;
}

size_t dtype_size(ncclDataType_t t) {
  switch (t) {
    case ncclInt8: case ncclUint8: return 1;
    case ncclInt32: case ncclUint32: case ncclFloat32: return 4;
    case ncclInt64: case ncclUint64: case ncclFloat64: return 8;
    default: return 0;
  }
}

template <typename T>
void sum_in_rank_order(const char* deposits, int world, size_t count, T* out) {
  for (size_t i = 0; i < count; ++i) {
    T acc = reinterpret_cast<const T*>(deposits)[i];
    for (int r = 1; r < world; ++r) acc += reinterpret_cast<const T*>(deposits + (size_t)r * AR_BYTES)[i];
    out[i] = acc;
  }
}

// one all-reduce (sum) executed NOW; device data moved on `st`, which must be idle with respect to the operands
ncclResult_t run_allreduce(ncclComm* c, const Op& op, hipStream_t st) {
  const size_t esz = dtype_size(op.dtype);
  const size_t per_round = AR_BYTES / esz;
  std::vector<char> result(AR_BYTES);
  for (size_t off = 0; off < op.count; off += per_round) {
    const size_t cnt = op.count - off < per_round ? op.count - off : per_round;
    char* mine = c->ar + (size_t)c->rank * AR_BYTES;
    if (hipMemcpyAsync(mine, (const char*)op.src + off * esz, cnt * esz, hipMemcpyDeviceToHost, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return ncclUnhandledCudaError;
    std::atomic_thread_fence(std::memory_order_seq_cst);
    if (!barrier(c)) return ncclSystemError;
    switch (op.dtype) {
      case ncclFloat64: sum_in_rank_order<double>(c->ar, c->world, cnt, (double*)result.data()); break;
      case ncclFloat32: sum_in_rank_order<float>(c->ar, c->world, cnt, (float*)result.data()); break;
      case ncclInt64: sum_in_rank_order<int64_t>(c->ar, c->world, cnt, (int64_t*)result.data()); break;
      case ncclUint64: sum_in_rank_order<uint64_t>(c->ar, c->world, cnt, (uint64_t*)result.data()); break;
      case ncclInt32: sum_in_rank_order<int32_t>(c->ar, c->world, cnt, (int32_t*)result.data()); break;
      case ncclUint32: sum_in_rank_order<uint32_t>(c->ar, c->world, cnt, (uint32_t*)result.data()); break;
      default: return ncclInvalidArgument;
    }
    if (!barrier(c)) return ncclSystemError;      // every rank has read the deposits: they may be overwritten
    if (hipMemcpyAsync((char*)op.dst + off * esz, result.data(), cnt * esz, hipMemcpyHostToDevice, st) != hipSuccess ||
        hipStreamSynchronize(st) != hipSuccess)
      return ncclUnhandledCudaError;
  }
  return ncclSuccess;
}

// a group of sends and receives executed NOW, progressed round-robin
ncclResult_t run_p2p(ncclComm* c, std::vector<Op>& ops, hipStream_t st) {
  size_t open = 0;
  for (Op& o : ops) {
    o.done = 0;
    if (o.bytes > 0) ++open;
  }
  const double t0 = now_s();
  int spins = 0;
  while (open > 0) {
    bool moved = false;
    for (Op& o : ops) {
      if (o.done == o.bytes) continue;
      const size_t piece = o.bytes - o.done < SLOT_BYTES ? o.bytes - o.done : SLOT_BYTES;
      if (o.kind == 0) {
        Mailbox& m = c->box[(size_t)c->rank * c->world + o.peer];
        const uint64_t s = m.sent.load(std::memory_order_relaxed);
        if (m.recvd.load(std::memory_order_acquire) != s) continue;      // previous piece not yet taken
        if (hipMemcpyAsync(m.data, (const char*)o.src + o.done, piece, hipMemcpyDeviceToHost, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
          return ncclUnhandledCudaError;
        m.sent.store(s + 1, std::memory_order_release);
      } else {
        Mailbox& m = c->box[(size_t)o.peer * c->world + c->rank];
        const uint64_t r = m.recvd.load(std::memory_order_relaxed);
        if (m.sent.load(std::memory_order_acquire) == r) continue;       // nothing there yet
        if (hipMemcpyAsync((char*)o.dst + o.done, m.data, piece, hipMemcpyHostToDevice, st) != hipSuccess ||
            hipStreamSynchronize(st) != hipSuccess)
          return ncclUnhandledCudaError;
        m.recvd.store(r + 1, std::memory_order_release);
      }
      o.done += piece;
      moved = true;
      if (o.done == o.bytes) --open;
    }
    if (moved) {
      spins = 0;
    } else {
      relax(spins);
      if ((spins & 1023) == 0 && now_s() - t0 > timeout_s()) {
        fprintf(stderr, "[fake_rccl] rank %d: point-to-point group timed out after %.0f s\n", c->rank, timeout_s());
        return ncclSystemError;
      }
    }
  }
  return ncclSuccess;
}

ncclResult_t run_ops(ncclComm* c, std::vector<Op>& ops, hipStream_t st) {
  std::vector<Op> p2p;
  for (Op& o : ops) {
    if (o.kind == 2) {
      maybe_hang(c, 0);
      ncclResult_t rc = run_allreduce(c, o, st);
      if (rc != ncclSuccess) return rc;
    } else {
      p2p.push_back(o);
    }
  }
  if (!p2p.empty()) {
    maybe_hang(c, 1);
    return run_p2p(c, p2p, st);
  }
  return ncclSuccess;
}

// ---- asynchronous mode ---------------------------------------------------------------------------------------------------
void worker_main(ncclComm* c) {
  (void)hipSetDevice(c->device);
  for (;;) {
    Work w;
    {
      std::unique_lock<std::mutex> lk(c->mu);
      c->cv.wait(lk, [&] { return c->stop || !c->queue.empty(); });
      if (c->queue.empty()) return;
      w = std::move(c->queue.front());
      c->queue.pop_front();
    }
    ncclResult_t rc = ncclSuccess;
    if (hipEventSynchronize(w.ready) != hipSuccess) rc = ncclUnhandledCudaError;
    if (rc == ncclSuccess) rc = run_ops(c, w.ops, c->pstream);
    (void)hipEventDestroy(w.ready);
    if (rc != ncclSuccess) {
      c->failed.store((int)rc);
      fprintf(stderr, "[fake_rccl] rank %d: asynchronous operation %llu failed (%d); the waiting stream is released\n", c->rank,
              (unsigned long long)w.seq, (int)rc);
    }
    // release the stream that waits for this operation (also after a failure: a stuck queue would hide the error)
    __atomic_store_n(c->done_word, w.seq, __ATOMIC_RELEASE);
  }
}

ncclResult_t submit(ncclComm* c, std::vector<Op>& ops, hipStream_t st) {
  if (ops.empty()) return ncclSuccess;
  if (c->failed.load() != 0) return (ncclResult_t)c->failed.load();
  if (!c->async) {
    if (hipStreamSynchronize(st) != hipSuccess) return ncclUnhandledCudaError;
    return run_ops(c, ops, st);
  }
  Work w;
  w.ops = ops;
  if (hipEventCreateWithFlags(&w.ready, hipEventDisableTiming) != hipSuccess) return ncclUnhandledCudaError;
  if (hipEventRecord(w.ready, st) != hipSuccess) return ncclUnhandledCudaError;
  w.seq = ++c->issued;
  const uint64_t seq = w.seq;
  {
    std::lock_guard<std::mutex> lk(c->mu);
    c->queue.push_back(std::move(w));
  }
  c->cv.notify_one();
  g_stats[5].fetch_add(1);
  if (hipStreamWaitValue64(st, c->done_word, seq, hipStreamWaitValueGte, ~0ull) != hipSuccess) return ncclUnhandledCudaError;
  return ncclSuccess;
}

bool env_flag(const char* name) {
  const char* e = getenv(name);
  return e && *e && strcmp(e, "0") != 0;
}

}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  if (!id) return ncclInvalidArgument;
  memset(id, 0, sizeof(*id));
  static std::atomic<uint32_t> serial{0};
  timespec ts;
  clock_gettime(CLOCK_REALTIME, &ts);
  snprintf(id->internal, sizeof(id->internal), "/fake_rccl_%d_%ld%09ld_%u", (int)getpid(), (long)ts.tv_sec, (long)ts.tv_nsec,
           serial.fetch_add(1));
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int nranks, ncclUniqueId id, int rank) {
  if (!out || nranks < 1 || nranks > MAX_RANKS || rank < 0 || rank >= nranks) return ncclInvalidArgument;
  if (strncmp(id.internal, "/fake_rccl_", 11) != 0 || !memchr(id.internal, 0, 64)) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = nranks;
  c->barriers = 0;
  c->issued = 0;
  c->stop = false;
  c->failed.store(0);
  c->done_word = nullptr;
  c->pstream = nullptr;
  c->index = g_comm_index.fetch_add(1);
  if (hipGetDevice(&c->device) != hipSuccess) {
    delete c;
    return ncclUnhandledCudaError;
  }
  strncpy(c->name, id.internal, sizeof(c->name) - 1);
  c->name[sizeof(c->name) - 1] = 0;
  c->bytes = sizeof(Header) + (size_t)nranks * AR_BYTES + (size_t)nranks * nranks * sizeof(Mailbox);
  // every rank opens-or-creates and sizes the segment (idempotent; new pages are zero: counters start at 0; pages of
  // mailboxes no pair ever uses are never touched, hence never committed)
  int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);
  if (fd < 0 || ftruncate(fd, (off_t)c->bytes) != 0) {
    if (fd >= 0) close(fd);
    delete c;
    return ncclSystemError;
  }
  c->base = (char*)mmap(nullptr, c->bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (c->base == MAP_FAILED) {
    delete c;
    return ncclSystemError;
  }
  c->hdr = (Header*)c->base;
  c->ar = c->base + sizeof(Header);
  c->box = (Mailbox*)(c->ar + (size_t)nranks * AR_BYTES);
  c->hdr->magic.store(MAGIC);
  c->hdr->attached.fetch_add(1);
  const double t0 = now_s();
  int spins = 0;
  while (c->hdr->attached.load(std::memory_order_acquire) < nranks) {       // ncclCommInitRank synchronises the ranks
    relax(spins);
    if ((spins & 1023) == 0 && now_s() - t0 > timeout_s()) {
      munmap(c->base, c->bytes);
      delete c;
      return ncclSystemError;
    }
  }
  // every rank has the segment mapped: the NAME can go now (a killed rank then leaves nothing behind in /dev/shm)
  if (rank == 0) shm_unlink(c->name);
  c->async = env_flag("FAKE_RCCL_ASYNC");
  if (c->async) {
    void* p = nullptr;
    if (hipExtMallocWithFlags(&p, 8, hipMallocSignalMemory) != hipSuccess || !p ||
        hipStreamCreateWithFlags(&c->pstream, hipStreamNonBlocking) != hipSuccess) {
      fprintf(stderr, "[fake_rccl] rank %d: no signal memory / stream for the asynchronous mode; blocking mode\n", rank);
      (void)hipGetLastError();
      c->async = false;
    } else {
      c->done_word = (uint64_t*)p;
      __atomic_store_n(c->done_word, 0ull, __ATOMIC_RELEASE);
      c->worker = std::thread(worker_main, c);
    }
  }
  g_stats[4].fetch_add(1);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  if (c->async) {
    {
      std::lock_guard<std::mutex> lk(c->mu);
      c->stop = true;
    }
    c->cv.notify_all();
    if (c->worker.joinable()) c->worker.join();
    (void)hipStreamDestroy(c->pstream);
    (void)hipFree(c->done_word);
  }
  c->hdr->detached.fetch_add(1);
  munmap(c->base, c->bytes);                       // (the name was removed when the last rank attached)
  delete c;
  return ncclSuccess;
}

ncclResult_t ncclCommCount(const ncclComm_t c, int* count) {
  if (!c || !count) return ncclInvalidArgument;
  *count = c->world;
  return ncclSuccess;
}

ncclResult_t ncclCommUserRank(const ncclComm_t c, int* rank) {
  if (!c || !rank) return ncclInvalidArgument;
  *rank = c->rank;
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() {
  if (t_group_depth++ == 0) {
    if (!t_pending) t_pending = new std::vector<Pending>();
    t_pending->clear();
  }
  return ncclSuccess;
}

ncclResult_t ncclGroupEnd() {
  if (t_group_depth <= 0) return ncclInvalidUsage;
  if (--t_group_depth > 0) return ncclSuccess;
  g_stats[3].fetch_add(1);
  // one submission per (communicator, stream) in order of first appearance
  std::vector<Pending>& all = *t_pending;
  std::vector<char> taken(all.size(), 0);
  ncclResult_t rc = ncclSuccess;
  for (size_t i = 0; i < all.size() && rc == ncclSuccess; ++i) {
    if (taken[i]) continue;
    std::vector<Op> ops;
    for (size_t j = i; j < all.size(); ++j)
      if (!taken[j] && all[j].comm == all[i].comm && all[j].stream == all[i].stream) {
        ops.push_back(all[j].op);
        taken[j] = 1;
      }
    rc = submit(all[i].comm, ops, all[i].stream);
  }
  all.clear();
  return rc;
}

static ncclResult_t enqueue(ncclComm* c, hipStream_t st, const Op& op) {
  if (t_group_depth > 0) {
    t_pending->push_back(Pending{c, st, op});
    return ncclSuccess;
  }
  std::vector<Op> one(1, op);
  return submit(c, one, st);
}

ncclResult_t ncclAllReduce(const void* sendbuff, void* recvbuff, size_t count, ncclDataType_t datatype, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t stream) {
  if (!c || !sendbuff || !recvbuff) return ncclInvalidArgument;
  if (op != ncclSum || dtype_size(datatype) < 4) return ncclInvalidArgument;
  g_stats[0].fetch_add(1);
  if (injected_failure(0)) return ncclInternalError;
  if (count == 0) return ncclSuccess;
  Op o{};
  o.kind = 2;
  o.src = sendbuff;
  o.dst = recvbuff;
  o.count = count;
  o.dtype = datatype;
  return enqueue(c, stream, o);
}

ncclResult_t ncclSend(const void* sendbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t c, hipStream_t stream) {
  if (!c || (!sendbuff && count) || peer < 0 || peer >= c->world || peer == c->rank || dtype_size(datatype) == 0)
    return ncclInvalidArgument;
  g_stats[1].fetch_add(1);
  if (injected_failure(1)) return ncclInternalError;
  g_stats[6].fetch_add(count * dtype_size(datatype));
  Op o{};
  o.kind = 0;
  o.src = sendbuff;
  o.bytes = count * dtype_size(datatype);
  o.peer = peer;
  return enqueue(c, stream, o);
}

ncclResult_t ncclRecv(void* recvbuff, size_t count, ncclDataType_t datatype, int peer, ncclComm_t c, hipStream_t stream) {
  if (!c || (!recvbuff && count) || peer < 0 || peer >= c->world || peer == c->rank || dtype_size(datatype) == 0)
    return ncclInvalidArgument;
  g_stats[2].fetch_add(1);
  Op o{};
  o.kind = 1;
  o.dst = recvbuff;
  o.bytes = count * dtype_size(datatype);
  o.peer = peer;
  return enqueue(c, stream, o);
}

// not part of RCCL: call counters for the tests (0 all-reduce, 1 send, 2 recv, 3 group_end, 4 communicators created,
// 5 operations handed to a progress thread, 6 bytes sent point-to-point)
void fake_rccl_stats(uint64_t* out8) {
  for (int i = 0; i < 8; ++i) out8[i] = g_stats[i].load();
}

}  // extern "C"
