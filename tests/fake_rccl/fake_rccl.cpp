// TEST INFRASTRUCTURE ONLY.  A stand-in for librccl.so that lets several ranks (processes) share ONE GPU, so that the
// solver's multi-rank control flow (union of block patterns, all-reduce of the reduced camera system, all-reduce of
// the step statistics) can be exercised on the one-GPU test box.  RCCL itself refuses two ranks on one device.
//
// Every collective is synchronous: wait for the stream, copy this rank's operand to a shared-memory slot, meet the
// other ranks, combine the slots in rank order on the host, copy the result back.  That is stronger ordering than
// RCCL's (stream-ordered), never weaker, so a schedule that is right here is right there as far as data flow goes.
// Every wait has a deadline: a rank that does not show up aborts the process instead of hanging the box.
//
// Only tests/test_gpu_two_ranks.py loads this (through BA_RCCL_LIBRARY in the rank processes it starts).
#include <fcntl.h>
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>
#include <sys/mman.h>
#include <unistd.h>

#include <atomic>
#include <chrono>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <random>

namespace {
constexpr size_t SLOT_BYTES = (size_t)8 << 20;  // per-rank operand limit of this shim
constexpr int MAX_RANKS = 8;
constexpr double DEADLINE_S = 30.0;

struct Shared {
  std::atomic<int> arrived;
  std::atomic<int> generation;
  char pad[56];
};

[[noreturn]] void die(const char* what) {
  std::fprintf(stderr, "fake_rccl: %s\n", what);
  std::_Exit(86);
}
}  // namespace

struct ncclComm {
  int rank, world;
  Shared* sh;
  char* slots;  // world slots of SLOT_BYTES
  size_t map_bytes;
  char name[64];
};

namespace {
void barrier(ncclComm* c) {
  const int gen = c->sh->generation.load(std::memory_order_acquire);
  if (c->sh->arrived.fetch_add(1, std::memory_order_acq_rel) + 1 == c->world) {
    c->sh->arrived.store(0, std::memory_order_relaxed);
    c->sh->generation.store(gen + 1, std::memory_order_release);
    return;
  }
  const auto t0 = std::chrono::steady_clock::now();
  while (c->sh->generation.load(std::memory_order_acquire) == gen) {
    if (std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count() > DEADLINE_S)
      die("a rank did not reach the collective in time");
    usleep(20);
  }
}

size_t type_bytes(ncclDataType_t t) {
  switch (t) {
    case ncclDouble: case ncclInt64: case ncclUint64: return 8;
    case ncclFloat: case ncclInt32: case ncclUint32: return 4;
    case ncclInt8: case ncclUint8: return 1;
    default: die("data type not handled by the shim");
  }
}

// stage this rank's operand into its slot and meet the others
void publish(ncclComm* c, const void* send, size_t bytes, hipStream_t stream) {
  if (bytes > SLOT_BYTES) die("operand larger than the shim's slot");
  if (hipStreamSynchronize(stream) != hipSuccess) die("stream synchronise failed");
  if (hipMemcpy(c->slots + (size_t)c->rank * SLOT_BYTES, send, bytes, hipMemcpyDeviceToHost) != hipSuccess)
    die("device-to-host copy failed");
  barrier(c);
}

void deliver(ncclComm* c, void* recv, const void* host, size_t bytes) {
  if (hipMemcpy(recv, host, bytes, hipMemcpyHostToDevice) != hipSuccess) die("host-to-device copy failed");
  barrier(c);  // nobody overwrites a slot before everybody has read it
}
}  // namespace

extern "C" {

ncclResult_t ncclGetUniqueId(ncclUniqueId* id) {
  std::memset(id, 0, sizeof *id);
  std::random_device rd;
  std::snprintf(id->internal, sizeof id->internal, "/fake_rccl_%08x%08x", rd(), rd());
  return ncclSuccess;
}

ncclResult_t ncclCommInitRank(ncclComm_t* out, int world, ncclUniqueId id, int rank) {
  if (world < 1 || world > MAX_RANKS || rank < 0 || rank >= world) return ncclInvalidArgument;
  ncclComm* c = new ncclComm();
  c->rank = rank;
  c->world = world;
  std::snprintf(c->name, sizeof c->name, "%s", id.internal);
  c->map_bytes = sizeof(Shared) + (size_t)world * SLOT_BYTES;
  const int fd = shm_open(c->name, O_CREAT | O_RDWR, 0600);  // a fresh object reads as zeros: counters start at 0
  if (fd < 0 || ftruncate(fd, (off_t)c->map_bytes) != 0) die("shared memory object");
  void* p = mmap(nullptr, c->map_bytes, PROT_READ | PROT_WRITE, MAP_SHARED, fd, 0);
  close(fd);
  if (p == MAP_FAILED) die("mmap");
  c->sh = static_cast<Shared*>(p);
  c->slots = static_cast<char*>(p) + sizeof(Shared);
  barrier(c);
  *out = c;
  return ncclSuccess;
}

ncclResult_t ncclCommDestroy(ncclComm_t c) {
  if (!c) return ncclSuccess;
  munmap(c->sh, c->map_bytes);
  shm_unlink(c->name);  // the first rank to leave removes the name; the others keep their mapping until munmap
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

ncclResult_t ncclAllReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op,
                           ncclComm_t c, hipStream_t stream) {
  if (type != ncclDouble || (op != ncclSum && op != ncclMax)) die("all-reduce flavour not handled by the shim");
  const size_t bytes = count * sizeof(double);
  publish(c, send, bytes, stream);
  double* acc = static_cast<double*>(std::malloc(bytes ? bytes : 8));
  std::memcpy(acc, c->slots, bytes);
  for (int r = 1; r < c->world; ++r) {
    const double* x = reinterpret_cast<const double*>(c->slots + (size_t)r * SLOT_BYTES);
    for (size_t i = 0; i < count; ++i) acc[i] = op == ncclSum ? acc[i] + x[i] : (x[i] > acc[i] ? x[i] : acc[i]);
  }
  deliver(c, recv, acc, bytes);
  std::free(acc);
  return ncclSuccess;
}

// (sum to one rank; the other ranks' receive buffers are left alone, as RCCL leaves them)
ncclResult_t ncclReduce(const void* send, void* recv, size_t count, ncclDataType_t type, ncclRedOp_t op, int root,
                        ncclComm_t c, hipStream_t stream) {
  if (type != ncclDouble || op != ncclSum) die("reduce flavour not handled by the shim");
  const size_t bytes = count * sizeof(double);
  publish(c, send, bytes, stream);
  if (c->rank == root) {
    double* acc = static_cast<double*>(std::malloc(bytes ? bytes : 8));
    std::memcpy(acc, c->slots, bytes);
    for (int r = 1; r < c->world; ++r) {
      const double* x = reinterpret_cast<const double*>(c->slots + (size_t)r * SLOT_BYTES);
      for (size_t i = 0; i < count; ++i) acc[i] += x[i];
    }
    if (hipMemcpy(recv, acc, bytes, hipMemcpyHostToDevice) != hipSuccess) die("host-to-device copy failed");
    std::free(acc);
  }
  barrier(c);  // nobody overwrites a slot before the root has read it
  return ncclSuccess;
}

ncclResult_t ncclBroadcast(const void* send, void* recv, size_t count, ncclDataType_t type, int root, ncclComm_t c,
                           hipStream_t stream) {
  const size_t bytes = count * type_bytes(type);
  publish(c, send, bytes, stream);
  deliver(c, recv, c->slots + (size_t)root * SLOT_BYTES, bytes);
  return ncclSuccess;
}

ncclResult_t ncclAllGather(const void* send, void* recv, size_t count, ncclDataType_t type, ncclComm_t c,
                           hipStream_t stream) {
  const size_t bytes = count * type_bytes(type);
  publish(c, send, bytes, stream);
  for (int r = 0; r < c->world; ++r)
    if (hipMemcpy(static_cast<char*>(recv) + (size_t)r * bytes, c->slots + (size_t)r * SLOT_BYTES, bytes,
                  hipMemcpyHostToDevice) != hipSuccess)
      die("host-to-device copy failed");
  barrier(c);
  return ncclSuccess;
}

ncclResult_t ncclGroupStart() { return ncclSuccess; }  // the grouped calls here are independent: run them in turn
ncclResult_t ncclGroupEnd() { return ncclSuccess; }
const char* ncclGetErrorString(ncclResult_t) { return "fake_rccl error"; }

}  // extern "C"
