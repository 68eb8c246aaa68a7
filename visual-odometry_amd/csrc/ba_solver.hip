// Host side of libba_amd.so: problem set-up (track sort, Schur block pattern), the
// Levenberg-Marquardt driver and the extern "C" entry points declared in include/ba_amd.h.
// Replaces the scipy least_squares call of reference src/bundle_adjustment.py:201-219.
#include "../../include/ba_amd.h"
#include "ba_kernels.h"
#include "ba_dense.h"
#include "ba_banded.h"
#include "ba_bcr.h"
#include "ba_prior.h"
#include "ba_coarse.h"
#include "ba_components.h"
#include "host_parallel.h"

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <numeric>
#include <memory>
#include <string>
#include <system_error>
#include <thread>
#include <limits>
#include <mutex>
#include <vector>

#include <dlfcn.h>
#include <hip/hip_ext.h>
#include <rccl/rccl.h>  // types and prototypes only: librccl.so is dlopen()ed by ba_comm_init

namespace {

using clk = std::chrono::steady_clock;
inline double ms_since(clk::time_point t0) {
  return std::chrono::duration<double, std::milli>(clk::now() - t0).count();
}

// RCCL entry points, resolved at run time so that the single-GPU path has no RCCL dependency.
struct Rccl {
  void* lib = nullptr;
  decltype(&ncclGetUniqueId) GetUniqueId = nullptr;
  decltype(&ncclCommInitRank) CommInitRank = nullptr;
  decltype(&ncclCommDestroy) CommDestroy = nullptr;
  decltype(&ncclCommCount) CommCount = nullptr;
  decltype(&ncclCommUserRank) CommUserRank = nullptr;
  decltype(&ncclAllReduce) AllReduce = nullptr;
  decltype(&ncclBroadcast) Broadcast = nullptr;
  decltype(&ncclAllGather) AllGather = nullptr;
  decltype(&ncclReduce) Reduce = nullptr;
  decltype(&ncclGroupStart) GroupStart = nullptr;
  decltype(&ncclGroupEnd) GroupEnd = nullptr;
  decltype(&ncclGetErrorString) GetErrorString = nullptr;
};

Rccl* rccl_load() {
  static Rccl r;
  static bool tried = false;
  if (tried) return r.lib ? &r : nullptr;
  tried = true;
  // BA_RCCL_LIBRARY names the library file outright (a site build of RCCL; the test stand-in of tests/fake_rccl)
  const char* named = getenv("BA_RCCL_LIBRARY");
  void* lib = named && *named ? dlopen(named, RTLD_NOW | RTLD_LOCAL) : dlopen("librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (named && *named && !lib) return nullptr;  // a named library that does not load is an error, not a cue to look elsewhere
  if (!lib) lib = dlopen("librccl.so.1", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) lib = dlopen("/opt/rocm/lib/librccl.so", RTLD_NOW | RTLD_GLOBAL);
  if (!lib) return nullptr;
#define RCCL_SYM(name)                                                        \
  r.name = reinterpret_cast<decltype(r.name)>(dlsym(lib, "nccl" #name));      \
  if (!r.name) return nullptr;
  RCCL_SYM(GetUniqueId) RCCL_SYM(CommInitRank) RCCL_SYM(CommDestroy) RCCL_SYM(CommCount) RCCL_SYM(CommUserRank)
  RCCL_SYM(AllReduce) RCCL_SYM(Broadcast) RCCL_SYM(Reduce)
  RCCL_SYM(AllGather) RCCL_SYM(GroupStart) RCCL_SYM(GroupEnd) RCCL_SYM(GetErrorString)
#undef RCCL_SYM
  r.lib = lib;
  return &r;
}

}  // namespace

// ---------------------------------------------------------------------------------------------- static scratch
// ba_evaluate_static allocates nothing: everything it needs on the device beyond the caller's own arrays
// (staged inputs, the residuals before they are copied out, per-block partial sums) lives in this file-scope
// device array.  Every GPU the library runs on holds its own copy at its own address until the process ends;
// host code never uses the array's name as an address, only the pointer hipGetSymbolAddress returned for
// the device (kept per device in static_scratch_ptr).  Regions, in doubles:
//   [partial 3 * blocks | out 4 | poses 7 * MAX_CAMS | points 3 * MAX_PTS | uv 2 * MAX_OBS | res 2 * MAX_OBS |
//    cam (int32) MAX_OBS / 2 | pt (int32) MAX_OBS / 2]
namespace {
constexpr long ST_MAX_OBS = BA_STATIC_MAX_OBS, ST_MAX_CAMS = BA_STATIC_MAX_CAMS, ST_MAX_PTS = BA_STATIC_MAX_PTS;
constexpr long ST_BLOCKS = (ST_MAX_OBS + 255) / 256;
constexpr long ST_OFF_PARTIAL = 0;
constexpr long ST_OFF_OUT = ST_OFF_PARTIAL + 3 * ST_BLOCKS;
constexpr long ST_OFF_POSES = ST_OFF_OUT + 4;
constexpr long ST_OFF_PTS = ST_OFF_POSES + 7 * ST_MAX_CAMS;
constexpr long ST_OFF_UV = ST_OFF_PTS + 3 * ST_MAX_PTS;
constexpr long ST_OFF_RES = ST_OFF_UV + 2 * ST_MAX_OBS;
constexpr long ST_OFF_CAM = ST_OFF_RES + 2 * ST_MAX_OBS;
constexpr long ST_OFF_PT = ST_OFF_CAM + ST_MAX_OBS / 2;
constexpr long ST_DOUBLES = ST_OFF_PT + ST_MAX_OBS / 2;
static_assert(ST_MAX_OBS % 2 == 0 && ST_OFF_POSES % 2 == 0, "regions stay 16-byte aligned");
constexpr int ST_MAX_DEVICES = 64;
}  // namespace
static __device__ __align__(256) double scratch[ST_DOUBLES];
namespace {
double* static_scratch_ptr[ST_MAX_DEVICES] = {};  // per device: the address of that GPU's copy of `scratch`
std::mutex static_scratch_mutex;                  // calls that overlap on one GPU would race on its array: serialised
}  // namespace

struct PendingDev { void** out; const void* src; size_t bytes; };  // a queued device array (queue_alloc / queue_upload)

struct ba_handle {
  ba_options opt;
  std::string err;
  int device = 0;
  hipStream_t stream = nullptr;
  bool has_problem = false, has_guess = false, has_estimate = false;
  double setup_ms = 0.0;
  bool digest_on = false;               // BA_TABLE_DIGEST=1: dev_upload records a digest of every table it uploads
  std::vector<uint64_t> table_digests;  // in upload order (ba_debug_table_digests)
  std::vector<PendingDev> pending;        // queue_alloc / queue_upload requests until dev_commit

  // host copies of the structure
  int C = 0, Cf = 0, L = 0, L_obs = 0, n = 0;
  std::vector<int> lm_internal_to_user, lm_user_to_internal;  // landmark permutation
  std::vector<int> obs_src;                                   // internal obs k -> caller's index
  std::vector<int> cam2free, free2cam;
  std::vector<int> urow_ptr, ucol;
  std::vector<uint8_t> fixed;
  int g_lin = 16, g_schur = 64, lpg_lin = 1, lpg_schur = 1, lpg_back = 1;
  long contrib_rows = 0;  // rows of the contribution tiles that k_schur_staged writes per launch
  bool use_flat_backsub = true;  // BA_BACKSUB_FLAT=0: the group-per-track kernel + k_cost
  bool gather_wave = false;      // k_gather_S_wave (a wave per S block): few contribution rows per block (BA_GATHER_WAVE overrides)
  bool matrix_free = false;      // no W buffer: k_schur_staged<MF> and k_backsub_flat rebuild W (BA_MATRIX_FREE=0: stored factors)
  bool lin_win = false;          // k_linearize_win takes the linearisation (BA_LIN_ROWS=0 BA_LIN_WIN=1: round 2's kernel, atomics)
  bool lin_rows = false;         // k_linearize_rows takes it: U | gc and the statistics summed in a fixed order (chunk tables exist)
  struct PosePrior {             // ba_set_pose_prior: the caller's prior, applied by every ba_set_problem until replaced
    std::vector<int> cams;
    std::vector<double> ref, H, g;
    double c0 = 0.0;
  } prior;
  int flat_chunks = 0;  // chunks of whole tracks (<= 64 observations) for the lane-per-observation kernels; 0: none
  int lin_cpw = 0;  // 1: k_linearize_flat (one chunk of whole tracks per wave) replaces k_linearize<G>
  int schur_cpb = 0;  // > 0: k_schur_staged with this many chunks per block
  bool small_solve = false;
  bool comp_solve = false;  // dense Cholesky per connected component of the camera graph
  bool comp_lm = false;     // ... and Levenberg-Marquardt per component (damping, accept / reject): CompLm
  size_t comp_lds = 0;
  bool dense_solve = false;  // whole-GPU dense Cholesky of the reduced system (ba_dense.h)
  bool banded_solve = false; // banded + arrow block Cholesky in one workgroup (ba_banded.h)
  BandedShape bshape{};
  double* banded_B = nullptr;
  size_t banded_lds = 0;
  // two-level (segmented) variant: P segments in parallel + a level-2 system of the separators
  int bseg_P = 0, bseg_bw = 0;
  // several ranks, BA_SHARDED_SOLVE=1: every rank eliminates the segments [sh_q0, sh_q1) of the two-level banded solver
  bool sharded = false;
  int sh_q0 = 0, sh_q1 = 0;
  std::vector<int> h_cam_owner;       // rank that adds the damping and the rhs of a free camera
  const int* cam_owner = nullptr;
  long bseg_base2 = 0, bseg_total = 0;
  size_t bseg_lds = 0;
  const BandedSeg* bseg_segs = nullptr;
  const long* bseg_dst = nullptr; const unsigned char* bseg_tr = nullptr; const long* bseg_rdst = nullptr;
  const long* bseg_cdst = nullptr; const unsigned char* bseg_ctr = nullptr; const long* bseg_crdst = nullptr;
  const int* bseg_l2 = nullptr;
  double* bseg_x2 = nullptr;
  std::vector<BandedSeg> h_segs; std::vector<long> h_dst, h_rdst, h_cdst, h_crdst; std::vector<unsigned char> h_tr, h_ctr;
  std::vector<int> h_l2;
  // block cyclic reduction of a trajectory's reduced system (ba_bcr.h)
  bool bcr_solve = false;
  BcrPlan bcr_plan{};
  BcrTop bcr_top{};          // the back-substitution levels that k_bcr_tail runs itself
  double* bcr_B = nullptr;
  size_t bcr_zero = 0;  // doubles of the level-0 arrays (cleared once, in ba_set_problem)
  bool bcr_direct = false;  // k_gather_S_wave writes the level-0 arrays itself (no k_bcr_assemble launch in solver mode)
  size_t bcr_lds_factor = 0, bcr_lds_update = 0, bcr_lds_tail = 0, bcr_lds_level = 0;
  bool bcr_fused = true;  // BA_BCR_FUSED=0: k_bcr_update and k_bcr_factor as separate launches at every level
  const long *bcr_dst = nullptr, *bcr_dst2 = nullptr, *bcr_rdst = nullptr, *bcr_pad = nullptr;
  const unsigned char* bcr_kind = nullptr;
  const int *bcr_cam0 = nullptr, *bcr_cnt = nullptr;
  int bcr_npad = 0;
  std::vector<long> h_bdst, h_bdst2, h_brdst, h_bpad;
  std::vector<unsigned char> h_bkind;
  std::vector<int> h_bcam0, h_bcnt;
  double* dense_A = nullptr;
  int dense_nt = 0;
  size_t small_lds = 0;
  // aggregation coarse space of the one-workgroup PCG (ba_coarse.h): built on a second stream, one LM iteration behind
  bool coarse = false;
  double* csum[2] = {nullptr, nullptr};  // coarse sums of the LM iterations of parity p (k_gather_S adds, k_coarse_build clears)
  double* cinv[2] = {nullptr, nullptr};  // Sc^-1 built from csum[p]: read by the solves of parity p, two iterations later
  size_t coarse_lds = 0;
  hipStream_t side = nullptr;
  hipEvent_t ev_coarse[2] = {nullptr, nullptr};
  int coarse_seq = 0;                      // solver-mode reduced solves enqueued in the current lm_solve
  int coarse_dirty[2] = {-1, -1};          // the iteration whose gather added to csum[p], -1: csum[p] is clear
  bool coarse_inflight[2] = {false, false};  // a build on buffer p has been launched and not yet been seen complete
  bool coarse_have[2] = {false, false};      // cinv[p] holds an inverse

  DeviceProblem d{};
  std::vector<void*> allocs;  // everything to hipFree
  double *poses_cur = nullptr, *poses_init = nullptr;  // poses_cur / pts_cur: where the last estimate lives
  double *pts_cur = nullptr, *pts_init = nullptr;
  double* scal_host = nullptr;  // pinned
  double* stage = nullptr;      // pinned staging for pose / point transfers
  size_t stage_bytes = 0;

  // optional HIP-event timing per kernel class (opt.profile_kernels)
  std::vector<hipEvent_t> ev_pool;
  size_t ev_used = 0;
  struct Span { int cls; hipEvent_t a, b; int iter; };
  int span_iter = 0;
  LmCtrl* ctrl_host = nullptr;  // pinned
  LmMirror* mirror = nullptr;   // pinned, written by k_lm_decide
  LmMirror* mirror_dev = nullptr;
  std::vector<Span> spans;
  double ktime_ms[BA_KERNEL_CLASSES] = {};
  int kcount[BA_KERNEL_CLASSES] = {};

  // multi-GPU: landmark-sharded ranks, one RCCL communicator (ba_comm_init)
  Rccl* rccl = nullptr;
  ncclComm_t comm = nullptr;
  int rank = 0, world = 1;
  double* scal_fold = nullptr;  // device [SC_COUNT + 1]: folded scalars + point-gradient inf-norm
  double* agree_dev = nullptr;  // device [2]: status agreement between the ranks (lives as long as the communicator)
  hipError_t launch_err = hipSuccess;  // first kernel launch the runtime refused since the last check
  bool agree_failed = false;  // an agreement of the current call has already delivered an error: make no further one
  long n_global = 0;            // observations over all ranks
};

namespace {

int fail(ba_handle* h, int code, const char* fmt, ...) {
  char buf[512];
  va_list ap;
  va_start(ap, fmt);
  vsnprintf(buf, sizeof buf, fmt, ap);
  va_end(ap);
  if (h) h->err = buf;
  return code;
}

#define HIP_TRY(h, expr)                                                                        \
  do {                                                                                          \
    hipError_t e_ = (expr);                                                                     \
    if (e_ != hipSuccess) {                                                                     \
      (void)hipGetLastError(); /* reported here: must not resurface as a "failed launch" */     \
      return fail(h, e_ == hipErrorOutOfMemory ? BA_ERR_NOMEM : BA_ERR_DEVICE, "%s failed: %s", \
                  #expr, hipGetErrorString(e_));                                                \
    }                                                                                           \
  } while (0)

enum { KT_LIN = 0, KT_SCHUR, KT_BACKSUB, KT_COST, KT_PCG, KT_OTHER, KT_GATHER, KT_DECIDE };
static_assert(KT_DECIDE + 1 == BA_KERNEL_CLASSES, "kernel classes");

#define NCCL_TRY(h, expr)                                                                              \
  do {                                                                                                 \
    ncclResult_t r_ = (expr);                                                                          \
    if (r_ != ncclSuccess) return fail(h, BA_ERR_DEVICE, "%s failed: %s", #expr, (h)->rccl->GetErrorString(r_)); \
  } while (0)

// A launch that the runtime refuses (bad grid or LDS size) leaves no trace in the stream: remember the first such
// error; the entry points turn it into BA_ERR_DEVICE at their next synchronisation.  HIP's "last error" is per host
// thread and stays set until somebody reads it: every entry point that launches kernels calls begin_launches() first,
// so that an error left behind by an EARLIER call on this thread (a failed hipMalloc of another handle, somebody
// else's HIP code) is not taken for a refused launch of this one.
inline void begin_launches(ba_handle* h) {
  (void)hipGetLastError();
  h->launch_err = hipSuccess;
}
inline void note_launch(ba_handle* h) {
  const hipError_t e = hipGetLastError();
  if (e != hipSuccess && h->launch_err == hipSuccess) h->launch_err = e;
}
int check_launches(ba_handle* h, const char* where) {
  if (h->launch_err == hipSuccess) return BA_OK;
  const hipError_t e = h->launch_err;
  h->launch_err = hipSuccess;
  return fail(h, BA_ERR_DEVICE, "%s: a kernel launch failed: %s", where, hipGetErrorString(e));
}

// Multi-rank calls agree on a status BEFORE their next data collective: a rank that returned early (a bad shard,
// a failed allocation) would otherwise leave the others blocked inside RCCL, which has no deadline.  Every rank
// contributes its local status; all of them get the maximum and return it.
int agree_status(ba_handle* h, int local_rc, const char* where) {
  if (h->world <= 1 || !h->comm) return local_rc;
  double v[2] = {(double)local_rc, 0.0};
  if (hipMemcpyAsync(h->agree_dev, v, sizeof(double), hipMemcpyHostToDevice, h->stream) != hipSuccess ||
      h->rccl->AllReduce(h->agree_dev, h->agree_dev + 1, 1, ncclDouble, ncclMax, h->comm, h->stream) != ncclSuccess ||
      hipMemcpyAsync(v + 1, h->agree_dev + 1, sizeof(double), hipMemcpyDeviceToHost, h->stream) != hipSuccess ||
      hipStreamSynchronize(h->stream) != hipSuccess)
    return local_rc ? local_rc : fail(h, BA_ERR_DEVICE, "%s: the ranks could not agree on a status", where);
  const int global_rc = (int)v[1];
  if (global_rc) h->agree_failed = true;  // every rank stops making agreements for this call at the same point
  if (local_rc) return local_rc;  // this rank's own message stands
  if (global_rc) return fail(h, global_rc, "%s: another rank reported status %d (all ranks return it)", where, global_rc);
  return BA_OK;
}

struct DevTmp {  // a device allocation that lives for one scope
  void* p = nullptr;
  ~DevTmp() { if (p) (void)hipFree(p); }
};

hipEvent_t next_event(ba_handle* h) {
  if (h->ev_used == h->ev_pool.size()) {
    hipEvent_t e;
    (void)hipEventCreate(&e);
    h->ev_pool.push_back(e);
  }
  return h->ev_pool[h->ev_used++];
}

// Call after the stream is idle.  Launches enqueued after the solver terminated, and linearise launches
// of iterations that followed a rejected step, exited at their first instruction: they are not counted.
void collect_spans(ba_handle* h, const LmCtrl& c) {
  for (auto& s : h->spans) {
    bool ran = s.iter < c.it;
    if (ran && s.cls == KT_LIN && s.iter > 0)  // the linearisation of iteration i + 1 ran iff iteration i accepted its step
      ran = s.iter - 1 < LM_ACC_BITS ? (c.acc_bits[(s.iter - 1) >> 5] >> ((s.iter - 1) & 31)) & 1u : false;
    float ms = 0.f;
    if (ran && hipEventElapsedTime(&ms, s.a, s.b) == hipSuccess) {
      h->ktime_ms[s.cls] += ms;
      h->kcount[s.cls] += 1;
    }
  }
  h->spans.clear();
  h->ev_used = 0;
}

void free_problem(ba_handle* h) {
  if (h->side) (void)hipStreamSynchronize(h->side);  // a coarse build may still be reading the buffers freed below
  for (int i = 0; i < BA_KERNEL_CLASSES; ++i) { h->ktime_ms[i] = 0.0; h->kcount[i] = 0; }
  for (void* p : h->allocs) (void)hipFree(p);
  h->allocs.clear();
  if (h->stage) (void)hipHostFree(h->stage);
  h->stage = nullptr;
  h->stage_bytes = 0;
  h->has_problem = h->has_guess = h->has_estimate = false;
  h->d = DeviceProblem{};
}

template <class T>
int dev_alloc(ba_handle* h, T** out, size_t count) {
  void* p = nullptr;
  const size_t bytes = std::max<size_t>(count, 1) * sizeof(T);
  HIP_TRY(h, hipMalloc(&p, bytes));
  h->allocs.push_back(p);
  *out = static_cast<T*>(p);
  return BA_OK;
}

// 64-bit digest of a table's bytes (length included): what ba_debug_table_digests reports
uint64_t digest_bytes(const void* ptr, size_t bytes) {
  const unsigned char* b = static_cast<const unsigned char*>(ptr);
  uint64_t x = 0x9E3779B97F4A7C15ull ^ (uint64_t)bytes;
  size_t i = 0;
  for (; i + 8 <= bytes; i += 8) {
    uint64_t w;
    std::memcpy(&w, b + i, 8);
    x = (x ^ w) * 0x100000001B3ull;
    x ^= x >> 29;
  }
  for (; i < bytes; ++i) x = (x ^ b[i]) * 0x100000001B3ull;
  return x;
}

template <class T>
int dev_upload(ba_handle* h, const T** out, const std::vector<T>& v) {
  if (h->digest_on) h->table_digests.push_back(digest_bytes(v.data(), v.size() * sizeof(T)));
  T* p = nullptr;
  int rc = dev_alloc(h, &p, v.size());
  if (rc) return rc;
  if (!v.empty()) HIP_TRY(h, hipMemcpy(p, v.data(), v.size() * sizeof(T), hipMemcpyHostToDevice));
  *out = p;
  return BA_OK;
}

// ba_set_problem's tables and state arrays: queued, then ONE device allocation for all of them (dev_commit) instead of
// a hipMalloc and a blocking copy per table (about ninety of each: 2.3 ms of a 12 ms set-up at BASELINE configs[1]).
// The small tables travel in one staging copy, the large ones one copy each, straight from their vectors, which must
// stay alive until dev_commit.
template <class T>
void queue_alloc(ba_handle* h, T** out, size_t count) {
  h->pending.push_back(PendingDev{reinterpret_cast<void**>(out), nullptr, std::max<size_t>(count, 1) * sizeof(T)});
}
template <class T, class A>
void queue_upload(ba_handle* h, const T** out, const std::vector<T, A>& v) {
  if (h->digest_on) h->table_digests.push_back(digest_bytes(v.data(), v.size() * sizeof(T)));
  h->pending.push_back(PendingDev{reinterpret_cast<void**>(const_cast<T**>(out)), v.empty() ? nullptr : (const void*)v.data(),
                                  std::max<size_t>(v.size(), 1) * sizeof(T)});
}
int env_int(const char* name, int dflt);
// One allocation for every array of `todo`, pointers handed out, uploads copied.  Touches no handle (it also runs on
// ba_set_problem's helper thread); *base_out is the allocation (to be freed by the caller's owner) or null.
hipError_t commit_list(const std::vector<PendingDev>& todo, void** base_out) {
  constexpr size_t ALIGN = 256, SMALL = 64 * 1024;
  *base_out = nullptr;
  auto padded = [](size_t b) { return (b + ALIGN - 1) / ALIGN * ALIGN; };
  // small uploads first and contiguous (one staging copy), then the large uploads, then what is only allocated
  auto cls = [&](const PendingDev& q) { return q.src == nullptr ? 2 : (q.bytes <= SMALL ? 0 : 1); };
  std::vector<size_t> off(todo.size());
  size_t total = 0, small_bytes = 0;
  for (int c = 0; c < 3; ++c) {
    for (size_t i = 0; i < todo.size(); ++i)
      if (cls(todo[i]) == c) { off[i] = total; total += padded(todo[i].bytes); }
    if (c == 0) small_bytes = total;
  }
  if (total == 0) return hipSuccess;
  void* base = nullptr;
  const auto t_a = clk::now();
  hipError_t e = hipMalloc(&base, total);
  if (e != hipSuccess) return e;
  if (env_int("BA_SETUP_TRACE", 0)) std::fprintf(stderr, "[ba_set_problem]   hipMalloc of %.1f MB %8.3f ms\n", total / 1048576.0, ms_since(t_a));
  *base_out = base;
  char* b = static_cast<char*>(base);
  for (size_t i = 0; i < todo.size(); ++i) *todo[i].out = b + off[i];
  if (small_bytes > 0) {
    std::vector<char> staging(small_bytes, 0);
    for (size_t i = 0; i < todo.size(); ++i)
      if (cls(todo[i]) == 0) std::memcpy(staging.data() + off[i], todo[i].src, todo[i].bytes);
    if ((e = hipMemcpy(b, staging.data(), small_bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
  }
  for (size_t i = 0; i < todo.size(); ++i)
    if (cls(todo[i]) == 1 && (e = hipMemcpy(b + off[i], todo[i].src, todo[i].bytes, hipMemcpyHostToDevice)) != hipSuccess) return e;
  return hipSuccess;
}
int dev_commit(ba_handle* h) {
  std::vector<PendingDev> todo;
  todo.swap(h->pending);
  void* base = nullptr;
  const hipError_t e = commit_list(todo, &base);
  if (base) h->allocs.push_back(base);
  HIP_TRY(h, e);
  return BA_OK;
}

int pow2_at_least(int v, int lo, int hi) {
  int g = lo;
  while (g < v && g < hi) g <<= 1;
  return g;
}

int env_int(const char* name, int dflt) {
  const char* s = getenv(name);
  return (s && *s) ? atoi(s) : dflt;
}

constexpr size_t SCAL_BYTES = (size_t)SCAL_SLABS * SC_COUNT * sizeof(double);
static_assert(LM_HISTORY == BA_HISTORY, "device and ABI history lengths differ");
static_assert(PRIOR_MAX_CAMS == BA_PRIOR_MAX_CAMS, "device and ABI prior sizes differ");

// read back every copy of the scalar block and fold them into copy 0 (sum; max for the inf-norms)
int fetch_scalars(ba_handle* h);

inline int grid_for(long items, int block = 256) { return (int)std::max<long>(1, (items + block - 1) / block); }

// ---------------------------------------------------------------------------------------------- launches
// The four streaming kernels are launched through hipExtLaunchKernelGGL when kernel timing is on: the
// start / stop events are attached to the launch itself, so the measured span is the kernel's execution
// (agrees with rocprofv3's per-dispatch duration) and costs no extra API call.
template <class... KArgs, class... Args>
void launch_timed_lds(ba_handle* h, int cls, void (*kern)(KArgs...), int grid, int block, size_t lds, Args... args) {
  if (h->opt.profile_kernels) {
    hipEvent_t a = next_event(h), b = next_event(h);
    hipExtLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, h->stream, a, b, 0, args...);
    h->spans.push_back({cls, a, b, h->span_iter});
  } else {
    hipLaunchKernelGGL(kern, dim3(grid), dim3(block), lds, h->stream, args...);
  }
  note_launch(h);
}
template <class... KArgs, class... Args>
void launch_timed(ba_handle* h, int cls, void (*kern)(KArgs...), int grid, int block, Args... args) {
  launch_timed_lds(h, cls, kern, grid, block, 0, args...);
}

inline size_t linf_lds_bytes(int C) {
  return ((size_t)C * 34 + (size_t)LINF_WAVES * (64 * LINF_VSTRIDE + LINF_MAXLM * 3)) * sizeof(double);
}

// poses / pts == nullptr: solver mode (state and run/skip decision come from the device-side LM control)
void launch_linearize(ba_handle* h, const double* poses, const double* pts) {
  const DeviceProblem& d = h->d;
  if (h->lin_rows) {
    const int grid = grid_for(d.n_lin_chunks, LINF_WAVES);
    if (d.loss.kind == LOSS_LINEAR) launch_timed(h, KT_LIN, k_linearize_rows<false>, grid, 64 * LINF_WAVES, d, poses, pts);
    else launch_timed(h, KT_LIN, k_linearize_rows<true>, grid, 64 * LINF_WAVES, d, poses, pts);
    return;
  }
  if (h->lin_win) {
    const int grid = grid_for(d.n_lin_chunks, LINF_WAVES);
    if (d.loss.kind == LOSS_LINEAR) launch_timed(h, KT_LIN, k_linearize_win<false>, grid, 64 * LINF_WAVES, d, poses, pts);
    else launch_timed(h, KT_LIN, k_linearize_win<true>, grid, 64 * LINF_WAVES, d, poses, pts);
    return;
  }
  if (h->lin_cpw > 0) {
    const int grid = grid_for(d.n_lin_chunks, LINF_WAVES);
    const size_t lds = linf_lds_bytes(d.C);
    if (d.loss.kind == LOSS_LINEAR)
      launch_timed_lds(h, KT_LIN, k_linearize_flat<false>, grid, 64 * LINF_WAVES, lds, d, poses, pts);
    else
      launch_timed_lds(h, KT_LIN, k_linearize_flat<true>, grid, 64 * LINF_WAVES, lds, d, poses, pts);
    return;
  }
  const int G = h->g_lin, lpg = h->lpg_lin;
  const int grid = grid_for(d.L_obs, (64 / G) * lpg);
  if (d.loss.kind == LOSS_LINEAR) {
    switch (G) {
      case 8: launch_timed(h, KT_LIN, k_linearize<8, false>, grid, 64, d, poses, pts, lpg); break;
      case 16: launch_timed(h, KT_LIN, k_linearize<16, false>, grid, 64, d, poses, pts, lpg); break;
      case 32: launch_timed(h, KT_LIN, k_linearize<32, false>, grid, 64, d, poses, pts, lpg); break;
      default: launch_timed(h, KT_LIN, k_linearize<64, false>, grid, 64, d, poses, pts, lpg); break;
    }
  } else {
    switch (G) {
      case 8: launch_timed(h, KT_LIN, k_linearize<8, true>, grid, 64, d, poses, pts, lpg); break;
      case 16: launch_timed(h, KT_LIN, k_linearize<16, true>, grid, 64, d, poses, pts, lpg); break;
      case 32: launch_timed(h, KT_LIN, k_linearize<32, true>, grid, 64, d, poses, pts, lpg); break;
      default: launch_timed(h, KT_LIN, k_linearize<64, true>, grid, 64, d, poses, pts, lpg); break;
    }
  }
}

// lam < 0: solver mode.  Returns true if the trial cost was computed by the same launch (k_backsub_flat in solver mode).
bool launch_backsub(ba_handle* h, double lam, const double* pts, double* pts_t, const double* poses, double* poses_t) {
  const DeviceProblem& d = h->d;
  if (h->flat_chunks > 0 && h->use_flat_backsub) {
    const int pb = grid_for(d.C);
    const int grid = grid_for(h->flat_chunks, 4) + pb;
    const int with_cost = lam < 0.0;
    if (d.loss.kind == LOSS_LINEAR)
      if (d.W) launch_timed(h, KT_BACKSUB, k_backsub_flat<false, false>, grid, 256, d, lam, h->opt.min_diag, pts, pts_t, poses, poses_t, pb, 0, with_cost);
      else launch_timed(h, KT_BACKSUB, k_backsub_flat<false, true>, grid, 256, d, lam, h->opt.min_diag, pts, pts_t, poses, poses_t, pb, 0, with_cost);
    else
      if (d.W) launch_timed(h, KT_BACKSUB, k_backsub_flat<true, false>, grid, 256, d, lam, h->opt.min_diag, pts, pts_t, poses, poses_t, pb, 0, with_cost);
      else launch_timed(h, KT_BACKSUB, k_backsub_flat<true, true>, grid, 256, d, lam, h->opt.min_diag, pts, pts_t, poses, poses_t, pb, 0, with_cost);
    return with_cost != 0;
  }
  const int G = h->g_lin, lpg = h->lpg_back;
  const long groups = (d.L_obs + lpg - 1) / lpg;
  const int pb = grid_for(d.C);  // trailing blocks: pose retraction + camera step statistics
  const int grid = grid_for(groups * G) + pb;
  const double md = h->opt.min_diag;
  const int guard = 0;  // explicit mode (debug export): the plain step; solver mode takes the option from the control block
  switch (G) {
    case 8: launch_timed(h, KT_BACKSUB, k_backsub<8>, grid, 256, d, lam, md, pts, pts_t, lpg, poses, poses_t, pb, guard); break;
    case 16: launch_timed(h, KT_BACKSUB, k_backsub<16>, grid, 256, d, lam, md, pts, pts_t, lpg, poses, poses_t, pb, guard); break;
    case 32: launch_timed(h, KT_BACKSUB, k_backsub<32>, grid, 256, d, lam, md, pts, pts_t, lpg, poses, poses_t, pb, guard); break;
    default: launch_timed(h, KT_BACKSUB, k_backsub<64>, grid, 256, d, lam, md, pts, pts_t, lpg, poses, poses_t, pb, guard); break;
  }
  return false;
}

void launch_schur(ba_handle* h, double lam) {
  const DeviceProblem& d = h->d;
  const int G = h->g_schur, lpg = h->lpg_schur;
  const double md = h->opt.min_diag;
  if (h->schur_cpb > 0) {
    // the leading n_fold blocks sum the U | gc rows of k_linearize_rows into the slabs (fold_u_rows): a block per
    // (camera, segment)
    const int n_fold = h->lin_rows ? d.C * d.nslab_u : 0;
    const int cpb = h->schur_cpb, grid = grid_for(d.n_chunks, cpb) + n_fold;
    // explicit mode (lam >= 0): the point that ba_debug_linearization staged in buffer 0
    const double* P = lam < 0.0 ? nullptr : (const double*)d.poses_buf[0];
    const double* X = lam < 0.0 ? nullptr : (const double*)d.pts_buf[0];
    if (h->matrix_free) {
      switch (G) {
        case 8: launch_timed(h, KT_SCHUR, k_schur_staged<8, true>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        case 16: launch_timed(h, KT_SCHUR, k_schur_staged<16, true>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        case 32: launch_timed(h, KT_SCHUR, k_schur_staged<32, true>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        default: launch_timed(h, KT_SCHUR, k_schur_staged<64, true>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
      }
    } else {
      switch (G) {
        case 8: launch_timed(h, KT_SCHUR, k_schur_staged<8, false>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        case 16: launch_timed(h, KT_SCHUR, k_schur_staged<16, false>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        case 32: launch_timed(h, KT_SCHUR, k_schur_staged<32, false>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
        default: launch_timed(h, KT_SCHUR, k_schur_staged<64, false>, grid, 64, d, lam, md, cpb, P, X, n_fold); break;
      }
    }
    return;
  }
  const int grid = grid_for(d.L_obs, (64 / G) * lpg);
  switch (G) {
    case 8: launch_timed(h, KT_SCHUR, k_schur<8>, grid, 64, d, lam, md, lpg); break;
    case 16: launch_timed(h, KT_SCHUR, k_schur<16>, grid, 64, d, lam, md, lpg); break;
    case 32: launch_timed(h, KT_SCHUR, k_schur<32>, grid, 64, d, lam, md, lpg); break;
    default: launch_timed(h, KT_SCHUR, k_schur<64>, grid, 64, d, lam, md, lpg); break;
  }
}

void launch_cost(ba_handle* h, const double* poses, const double* pts, int cost_slot, int behind_slot,
                 double* res_out) {
  const int grid = std::min(grid_for(h->d.n), 256 * 16);
  if (h->d.loss.kind == LOSS_LINEAR)
    launch_timed(h, KT_COST, k_cost<false>, grid, 256, h->d, poses, pts, cost_slot, behind_slot, res_out);
  else
    launch_timed(h, KT_COST, k_cost<true>, grid, 256, h->d, poses, pts, cost_slot, behind_slot, res_out);
}

// One allocation holds [Ugc slabs | scalar copies | S slabs]: a linearising iteration clears all of it
// with ONE memset, a re-solve after a rejected step clears [scalar copies | S slabs].
int clear_accumulators(ba_handle* h, bool with_u) {
  const DeviceProblem& d = h->d;
  const size_t u_bytes = (size_t)d.nslab_u * h->C * 27 * sizeof(double);
  const size_t rest = SCAL_BYTES + (size_t)d.nslab_s * d.s_slab_stride * sizeof(double);
  if (with_u) HIP_TRY(h, hipMemsetAsync(d.Ugc_slab, 0, u_bytes + rest, h->stream));
  else HIP_TRY(h, hipMemsetAsync(d.scal, 0, rest, h->stream));
  return BA_OK;
}

// U V W gc gp at (poses, pts) [explicit mode] or at the current estimate if it moved [solver mode]
void enqueue_linearize(ba_handle* h, const double* poses, const double* pts) {
  launch_linearize(h, poses, pts);
  if (h->schur_cpb > 0) return;  // staged path: the Schur launch folds the U | gc rows, k_gather_S sums the slabs
  if (h->lin_rows) { k_fold_U<<<h->d.C * h->d.nslab_u, 64, 0, h->stream>>>(h->d, poses == nullptr); note_launch(h); }
  k_reduce_U<<<grid_for(27L * h->C, 16), 256, 0, h->stream>>>(h->d, poses == nullptr); note_launch(h);
}

// build S / rhs for damping lam (lam < 0: solver mode), precondition, solve for dc by block-Jacobi PCG
int do_reduced_solve(ba_handle* h, double lam, int* pcg_iters) {
  DeviceProblem& d = h->d;
  const ba_options& o = h->opt;
  const int mode = lam < 0.0;
  // coarse space of the one-workgroup PCG (ba_coarse.h): buffer p = parity of this LM iteration
  double* csum_arg = nullptr;
  const double* cinv_arg = nullptr;
  if (mode && h->coarse) {
    const int p = h->coarse_seq & 1;
    if (h->coarse_inflight[p]) {  // launched two iterations ago: complete, as a rule
      if (hipEventQuery(h->ev_coarse[p]) != hipSuccess) HIP_TRY(h, hipStreamWaitEvent(h->stream, h->ev_coarse[p], 0));
      h->coarse_inflight[p] = false;
      h->coarse_have[p] = true;
    }
    if (h->coarse_dirty[p] < 0) {  // clear (never used, or consumed by a build): this iteration's sums go there
      csum_arg = h->csum[p];
      h->coarse_dirty[p] = h->coarse_seq;
    }
    if (h->coarse_have[p]) cinv_arg = h->cinv[p];
    ++h->coarse_seq;
  }
  launch_schur(h, lam);
  // the gather also inverts the damped diagonal blocks when they are final there (one GPU, one-workgroup PCG)
  const int fused_precond = h->schur_cpb > 0 && h->world == 1 && h->small_solve && d.prior.n == 0;
  // one GPU, solver mode, no prior: S is final in the gather, which then fills the cyclic reduction's arrays itself
  const bool bcr_direct = mode && h->bcr_solve && h->bcr_direct && h->gather_wave && h->schur_cpb > 0 && !fused_precond;
  BcrScatter bsc{};
  if (bcr_direct)
    bsc = BcrScatter{h->bcr_B, h->bcr_dst, h->bcr_dst2, h->bcr_rdst, h->bcr_kind, h->bcr_plan.m, o.min_diag};
  if (h->schur_cpb > 0 && h->gather_wave && !fused_precond)
    launch_timed(h, KT_GATHER, k_gather_S_wave, grid_for(d.nu + (d.C - d.Cf), 4), 256, d, mode, bsc);
  else if (h->schur_cpb > 0)
    launch_timed(h, KT_GATHER, k_gather_S, d.nu + (d.C - d.Cf), GATHER_THREADS, d, mode, lam, o.min_diag, fused_precond, csum_arg);
  else launch_timed(h, KT_GATHER, k_reduce_S, grid_for(d.s_slab_stride + 27L * d.C), 256, d, mode);
  const size_t red_count = (size_t)d.nu * 36 + (size_t)6 * d.Cf + (size_t)27 * d.C;
  if (h->world > 1 && h->sharded)  // sharded reduced solve: only rhs | U, gc are summed everywhere; S travels by segment below
    NCCL_TRY(h, h->rccl->AllReduce(d.rhs, d.rhs, (size_t)6 * d.Cf + (size_t)27 * d.C, ncclDouble, ncclSum, h->comm, h->stream));
  else if (h->world > 1)  // the one exchange step of the path: partial S, reduced rhs, U / gc summed over the shards
    NCCL_TRY(h, h->rccl->AllReduce(d.S, d.S, red_count, ncclDouble, ncclSum, h->comm, h->stream));
  if (d.prior.n > 0) {  // the pose prior's Gauss-Newton terms, into the assembled system
    k_prior_add<<<1, 256, 0, h->stream>>>(d, mode ? nullptr : (const double*)d.poses_buf[0]);  // explicit mode: the point that ba_debug_linearization staged
    note_launch(h);
  }
  const double tol2 = o.pcg_tol * o.pcg_tol;
  if (h->comp_solve) {
    launch_timed_lds(h, KT_PCG, k_solve_components, d.n_comp, 64, h->comp_lds, d, lam, o.min_diag);
    *pcg_iters = -1;
    return BA_OK;
  }
  if (h->bcr_solve) {
    const LmCtrl* ctrl = mode ? d.ctrl : nullptr;
    const BcrPlan& pl = h->bcr_plan;
    double* B = h->bcr_B;
    if (!bcr_direct)
      launch_timed(h, KT_PCG, k_bcr_assemble, grid_for((long)d.nu * 36 + 6L * d.Cf + h->bcr_npad), 256, d, B, pl.m, h->bcr_dst,
                   h->bcr_dst2, h->bcr_kind, h->bcr_rdst, h->bcr_pad, h->bcr_npad, lam, o.min_diag, mode);
    const int m = pl.m;
    if (h->bcr_fused && pl.n_levels > 0) {
      // factor of level 0; then one launch per level: the update of level l with the factor of level l + 1; the last
      // update leaves the tail system
      launch_timed_lds(h, KT_PCG, k_bcr_factor, pl.lv[0].n_elim, BCR_THREADS, h->bcr_lds_factor, B, pl.lv[0], m, ctrl, d.scal + SC_BADV);
      for (int l = 0; l + 1 < pl.n_levels; ++l)
        launch_timed_lds(h, KT_PCG, k_bcr_level, pl.lv[l + 1].n, BCR_THREADS, h->bcr_lds_level, B, pl.lv[l], pl.lv[l + 1], m, l, ctrl,
                         d.scal + SC_BADV);
      launch_timed_lds(h, KT_PCG, k_bcr_update, pl.lv[pl.n_levels].n, BCR_THREADS, h->bcr_lds_update, B, pl.lv[pl.n_levels - 1],
                       pl.lv[pl.n_levels], m, ctrl);
    } else {
      for (int l = 0; l < pl.n_levels; ++l) {
        launch_timed_lds(h, KT_PCG, k_bcr_factor, pl.lv[l].n_elim, BCR_THREADS, h->bcr_lds_factor, B, pl.lv[l], m, ctrl, d.scal + SC_BADV);
        launch_timed_lds(h, KT_PCG, k_bcr_update, pl.lv[l + 1].n, BCR_THREADS, h->bcr_lds_update, B, pl.lv[l], pl.lv[l + 1], m, ctrl);
      }
    }
    launch_timed_lds(h, KT_PCG, k_bcr_tail, 1, BCR_THREADS, h->bcr_lds_tail, B, pl.lv[pl.n_levels], m, pl.X, pl.n_levels, h->bcr_cam0,
                     h->bcr_cnt, ctrl, d.dc, d.scal + SC_BADV, h->bcr_top);
    for (int l = pl.n_levels - 1 - h->bcr_top.n; l >= 0; --l) {
      if (bcr_wave_backsolve_ok(m))
        launch_timed(h, KT_PCG, k_bcr_backsolve_w, grid_for(pl.lv[l].n_elim, 4), BCR_THREADS, B, pl.lv[l], m, pl.X, l, h->bcr_cam0,
                     h->bcr_cnt, ctrl, d.dc);
      else
        launch_timed(h, KT_PCG, k_bcr_backsolve, pl.lv[l].n_elim, BCR_THREADS, B, pl.lv[l], m, pl.X, l, h->bcr_cam0, h->bcr_cnt, ctrl, d.dc);
    }
    *pcg_iters = -1;
    return BA_OK;
  }
  if (h->banded_solve && h->bseg_P > 0) {
    const LmCtrl* ctrl = mode ? d.ctrl : nullptr;
    const BandedShape& sh2 = h->bshape;  // the level-2 system
    const int P = h->bseg_P, bw = h->bseg_bw;
    double* Ball = h->banded_B;
    HIP_TRY(h, hipMemsetAsync(Ball, 0, (size_t)h->bseg_total * sizeof(double), h->stream));
    // Sharded (several ranks, BA_SHARDED_SOLVE=1): every rank scatters its PARTIAL S; the damping and the (all-reduced)
    // right-hand side of a camera are added by the rank that owns it.  The segments then travel to their owners (one
    // reduce per rank, grouped: half the volume of the all-reduce they replace), each rank eliminates the interiors of
    // its own segments, and what they leave on the separators is summed over the ranks together with the separators'
    // own blocks (an all-reduce of the small level-2 system).
    const int q0 = h->sharded ? h->sh_q0 : 0, q1 = h->sharded ? h->sh_q1 : P;
    launch_timed(h, KT_PCG, k_bseg_assemble, grid_for((long)d.nu * 36 + 6L * d.Cf), 256, d, Ball, h->bseg_dst, h->bseg_tr,
                 h->bseg_rdst, lam, o.min_diag, mode, h->sharded ? h->cam_owner : nullptr, h->rank);
    if (h->sharded) {
      const int per = P / h->world;
      NCCL_TRY(h, h->rccl->GroupStart());
      for (int t = 0; t < h->world; ++t) {
        const long lo = h->h_segs[(size_t)t * per].base;
        const long hi = t + 1 < h->world ? h->h_segs[(size_t)(t + 1) * per].base : h->bseg_base2;
        NCCL_TRY(h, h->rccl->Reduce(Ball + lo, Ball + lo, (size_t)(hi - lo), ncclDouble, ncclSum, t, h->comm, h->stream));
      }
      NCCL_TRY(h, h->rccl->GroupEnd());
    }
    launch_timed_lds(h, KT_PCG, k_bseg_factor, q1 - q0, BND_THREADS, h->bseg_lds, Ball, h->bseg_segs + q0, bw, ctrl, d.scal + SC_BADV);
    for (int parity = 0; parity < 2; ++parity)
      launch_timed(h, KT_PCG, k_bseg_corner_add, (q1 - q0 + 1) / 2, 256, Ball, h->bseg_segs, bw, parity, h->bseg_cdst, h->bseg_ctr,
                   h->bseg_crdst, ctrl, q0, q1);
    if (h->sharded)
      NCCL_TRY(h, h->rccl->AllReduce(Ball + h->bseg_base2, Ball + h->bseg_base2, (size_t)(h->bseg_total - h->bseg_base2), ncclDouble,
                                     ncclSum, h->comm, h->stream));
    launch_timed_lds(h, KT_PCG, k_banded_factor, 1, BND_THREADS, h->banded_lds, Ball + h->bseg_base2, sh2, ctrl,
                     d.scal + SC_BADV);
    launch_timed_lds(h, KT_PCG, k_banded_backsolve, 1, 64,
                     (size_t)((sh2.bw + 1 + std::max(sh2.na, 1)) * 6 + 66) * sizeof(double),
                     (const double*)(Ball + h->bseg_base2), sh2, ctrl, h->bseg_x2);
    if (h->sharded) HIP_TRY(h, hipMemsetAsync(d.dc, 0, (size_t)6 * d.Cf * sizeof(double), h->stream));
    launch_timed_lds(h, KT_PCG, k_bseg_backsolve, q1 - q0, 64, (size_t)((bw + 1 + 2 * bw) * 6 + 66) * sizeof(double),
                     (const double*)Ball, h->bseg_segs, bw, h->bseg_l2, (const double*)h->bseg_x2, ctrl, d.dc, q0);
    // every camera's step has exactly one writer among the ranks (zeros elsewhere): the sum hands every rank the same bits
    if (h->sharded) NCCL_TRY(h, h->rccl->AllReduce(d.dc, d.dc, (size_t)6 * d.Cf, ncclDouble, ncclSum, h->comm, h->stream));
    *pcg_iters = -1;
    return BA_OK;
  }
  if (h->banded_solve) {
    const LmCtrl* ctrl = mode ? d.ctrl : nullptr;
    const BandedShape& sh = h->bshape;
    HIP_TRY(h, hipMemsetAsync(h->banded_B, 0, (size_t)sh.Cf * sh.nrow * 36 * sizeof(double), h->stream));
    launch_timed(h, KT_PCG, k_banded_assemble, grid_for((long)d.nu * 36 + 6L * d.Cf), 256, d, h->banded_B, sh, lam,
                 o.min_diag, mode);
    launch_timed_lds(h, KT_PCG, k_banded_factor, 1, BND_THREADS, h->banded_lds, h->banded_B, sh, ctrl, d.scal + SC_BADV);
    launch_timed_lds(h, KT_PCG, k_banded_backsolve, 1, 64, (size_t)((sh.bw + 1 + std::max(sh.na, 1)) * 6 + 66) * sizeof(double),
                     (const double*)h->banded_B, sh, ctrl, d.dc);
    *pcg_iters = -1;
    return BA_OK;
  }
  if (h->dense_solve) {
    const LmCtrl* ctrl = mode ? d.ctrl : nullptr;
    const int nt = h->dense_nt;
    const long ld = (long)nt * DT;
    double* A = h->dense_A;
    HIP_TRY(h, hipMemsetAsync(A, 0, (size_t)(nt + 1) * DT * ld * sizeof(double), h->stream));
    launch_timed(h, KT_PCG, k_dense_assemble, grid_for((long)d.nu * 36 + (long)nt * DT), 256, d, A, ld, nt, lam, o.min_diag, mode);
    const size_t lds_trsm = ((size_t)DT * DT + DT) * sizeof(double);
    const size_t lds_back = ((size_t)DPACK + 2 * DT) * sizeof(double);
    for (int k = 0; k < nt; ++k) {
      const int R = nt - k;  // tile rows below the diagonal tile (the rhs row included)
      launch_timed(h, KT_PCG, k_dense_potrf, 1, 256, A, ld, k, ctrl, d.scal + SC_BADV);
      launch_timed_lds(h, KT_PCG, k_dense_trsm, R, 256, lds_trsm, A, ld, k, ctrl);
      launch_timed(h, KT_PCG, k_dense_update, R * (R + 1) / 2, 512, A, ld, k, nt, ctrl);
    }
    for (int k = nt - 1; k >= 0; --k)
      launch_timed_lds(h, KT_PCG, k_dense_backsolve, k + 1, 256, lds_back, A, ld, k, nt, ctrl, d.dc, 6 * d.Cf);
    *pcg_iters = -1;
    return BA_OK;
  }
  if (h->small_solve) {
    if (!fused_precond) launch_timed(h, KT_OTHER, k_block_precond, grid_for(d.Cf, 64), 64, d, lam, o.min_diag);
    const double* cin = cinv_arg;
    if (h->coarse)
      launch_timed_lds(h, KT_PCG, k_solve_small<1, true>, 1, SMALL_THREADS, h->small_lds, d, lam, o.min_diag, tol2, o.pcg_max_iter, cin);
    else if (d.nu <= SMALL_THREADS)
      launch_timed_lds(h, KT_PCG, k_solve_small<1, false>, 1, SMALL_THREADS, h->small_lds, d, lam, o.min_diag, tol2, o.pcg_max_iter, cin);
    else
      launch_timed_lds(h, KT_PCG, k_solve_small<2, false>, 1, SMALL_THREADS, h->small_lds, d, lam, o.min_diag, tol2, o.pcg_max_iter, cin);
    *pcg_iters = -1;  // lives in scal[SC_PCGIT]
    // every rank solved the same all-reduced system with a solver whose summation order is fixed (no
    // atomics): the camera step is bit-identical on all ranks and needs no broadcast
    return BA_OK;
  }
  k_damp_mirror<<<grid_for(d.nu), 256, 0, h->stream>>>(d, lam, o.min_diag); note_launch(h);
  const int gv = grid_for(6L * d.Cf);
  k_pcg_init<<<gv, 256, 0, h->stream>>>(d, mode); note_launch(h);
  const int every = std::max(1, o.pcg_check_every);
  int launched = 0;
  while (launched < o.pcg_max_iter) {
    const int nb = std::min(every, o.pcg_max_iter - launched);
    for (int i = 0; i < nb; ++i) {
      k_pcg_spmv<<<gv, 256, 0, h->stream>>>(d);
      k_pcg_update<<<gv, 256, 0, h->stream>>>(d);
      k_pcg_dir<<<gv, 256, 0, h->stream>>>(d, tol2, o.pcg_max_iter); note_launch(h);
    }
    launched += nb;
    if (h->world > 1)  // all ranks must take the same number of PCG batches: follow rank 0's scalars
      NCCL_TRY(h, h->rccl->Broadcast(d.scal + SC_RZ, d.scal + SC_RZ, 8, ncclDouble, 0, h->comm, h->stream));
    HIP_TRY(h, hipMemcpyAsync(h->scal_host + SC_RZ, d.scal + SC_RZ, 8 * sizeof(double), hipMemcpyDeviceToHost,
                              h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    if (h->scal_host[SC_DONE] != 0.0) break;
  }
  *pcg_iters = (int)h->scal_host[SC_PCGIT];
  if (h->world > 1) NCCL_TRY(h, h->rccl->Broadcast(d.dc, d.dc, (size_t)6 * d.Cf, ncclDouble, 0, h->comm, h->stream));
  return BA_OK;
}

// step statistics + trial state + trial cost (all arguments null / negative: solver mode)
void enqueue_trial(ba_handle* h, double lam, const double* poses, const double* pts, double* poses_t, double* pts_t) {
  if (!launch_backsub(h, lam, pts, pts_t, poses, poses_t))
    launch_cost(h, poses_t, pts_t, SC_COST_T, SC_NBEHIND_T, nullptr);
  // the per-chunk / per-block partial statistics of k_linearize_rows, k_backsub_flat and the pose blocks, summed in a
  // fixed order into the copies of the scalar block (no atomics: the LM path is the same in every run)
  // (batched solves, solver mode: the per-window sums ride in the same launch, ceil(n_comp / 4) blocks after the first 64)
  const int comp_blocks = lam < 0.0 && h->comp_lm ? grid_for(h->d.n_comp, 4) : 0;
  launch_timed(h, KT_DECIDE, k_fold_stats, SCAL_SLABS + comp_blocks, 256, h->d, lam < 0.0 ? 1 : 0, h->lin_rows ? 1 : 0,
               h->flat_chunks > 0 && h->use_flat_backsub ? 1 : 0);
  if (h->d.prior.n > 0) {
    k_prior_trial<<<1, 128, 0, h->stream>>>(h->d, poses, (const double*)poses_t);
    note_launch(h);
  }
}

int fetch_scalars(ba_handle* h) {
  double* s0 = h->scal_host;
  HIP_TRY(h, hipMemcpyAsync(s0, h->d.scal, SCAL_BYTES, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int k = 1; k < SCAL_SLABS; ++k) {
    const double* sk = s0 + (size_t)k * SC_COUNT;
    for (int i = 0; i < SC_NSUM; ++i) {
      if (i == SC_GINF || i == SC_GINF_C) s0[i] = std::max(s0[i], sk[i]); else s0[i] += sk[i];
    }
  }
  return BA_OK;
}

// ---------------------------------------------------------------------------------------------- transfers
// Quaternion handling follows the reference: free poses are normalised (bundle_adjustment.py:63),
// the fixed pose only when its norm is off by more than 1e-6 (geometry.py:189-190).
int stage_state(ba_handle* h, const double* poses, const double* points, double* d_poses, double* d_pts) {
  double* sp = h->stage;
  double* sx = h->stage + (size_t)h->C * 7;
  for (int c = 0; c < h->C; ++c) {
    const double* p = poses + 7 * c;
    for (int i = 0; i < 7; ++i)
      if (!std::isfinite(p[i])) return fail(h, BA_ERR_INVALID, "pose %d has a non-finite entry", c);
    const double nq = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    if (nq == 0.0) return fail(h, BA_ERR_INVALID, "pose %d has a zero quaternion", c);
    const bool renorm = !h->fixed[c] || std::fabs(nq - 1.0) > 1e-6;
    for (int i = 0; i < 3; ++i) sp[7 * c + i] = p[i];
    for (int i = 3; i < 7; ++i) sp[7 * c + i] = renorm ? p[i] / nq : p[i];
  }
  for (int j = 0; j < h->L; ++j) {
    const double* x = points + 3 * (size_t)h->lm_internal_to_user[j];
    if (!std::isfinite(x[0]) || !std::isfinite(x[1]) || !std::isfinite(x[2]))
      return fail(h, BA_ERR_INVALID, "landmark %d has a non-finite coordinate", h->lm_internal_to_user[j]);
    sx[3 * (size_t)j] = x[0]; sx[3 * (size_t)j + 1] = x[1]; sx[3 * (size_t)j + 2] = x[2];
  }
  HIP_TRY(h, hipMemcpyAsync(d_poses, sp, (size_t)h->C * 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipMemcpyAsync(d_pts, sx, (size_t)h->L * 3 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  return BA_OK;
}

int fetch_state(ba_handle* h, const double* d_poses, const double* d_pts, double* poses, double* points) {
  double* sp = h->stage;
  double* sx = h->stage + (size_t)h->C * 7;
  HIP_TRY(h, hipMemcpyAsync(sp, d_poses, (size_t)h->C * 7 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(sx, d_pts, (size_t)h->L * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  for (int c = 0; c < h->C; ++c)
    if (!h->fixed[c]) std::memcpy(poses + 7 * c, sp + 7 * c, 7 * sizeof(double));
  for (int j = 0; j < h->L_obs; ++j)  // unobserved landmarks are left untouched
    std::memcpy(points + 3 * (size_t)h->lm_internal_to_user[j], sx + 3 * (size_t)j, 3 * sizeof(double));
  return BA_OK;
}

// ---------------------------------------------------------------------------------------------- LM driver
// The host only ENQUEUES iterations; the accept / reject decision, the damping update and the
// termination tests run on the device (k_lm_decide).  Every lm_check_every iterations the small
// control block is read back; iterations enqueued after termination exit at their first instruction.
int lm_solve(ba_handle* h, ba_report* rep) {
  const ba_options& o = h->opt;
  DeviceProblem& d = h->d;
  std::memset(rep, 0, sizeof *rep);
  rep->nobs = (int32_t)h->n_global;
  rep->setup_ms = h->setup_ms;
  LmCtrl& c = *h->ctrl_host;
  std::memset(&c, 0, sizeof c);
  begin_launches(h);
  c.ftol = o.ftol; c.xtol = o.xtol; c.gtol = o.gtol;
  c.lambda_min = o.lambda_min; c.lambda_max = o.lambda_max; c.accept_ratio = o.accept_ratio;
  c.max_iters = o.max_iters; c.guard_cheirality = o.guard_cheirality;
  c.lam = o.lambda0; c.nu = 2.0; c.need_lin = 1;
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  const auto t0 = clk::now();
  {
    const long n_clear = (long)d.nslab_u * h->C * 27 + (long)(SCAL_BYTES / sizeof(double)) + (long)d.nslab_s * d.s_slab_stride;
    const long work = std::max<long>(3L * h->L, n_clear);
    k_init_solve<<<(int)std::min<long>(1024, (work + 255) / 256), 256, 0, h->stream>>>(
        d, (const double*)h->poses_init, (const double*)h->pts_init, static_cast<const LmHead&>(c), n_clear);
    note_launch(h);
    if (h->comp_lm) {
      k_comp_init<<<std::min(1024, grid_for(std::max(d.L_obs, d.n_comp))), 256, 0, h->stream>>>(d, o.lambda0);
      const int grid = grid_for(h->flat_chunks, 4);
      if (d.loss.kind == LOSS_LINEAR)
        k_comp_cost<false><<<grid, 256, 0, h->stream>>>(d, (const double*)d.poses_buf[0], (const double*)d.pts_buf[0]);
      else
        k_comp_cost<true><<<grid, 256, 0, h->stream>>>(d, (const double*)d.poses_buf[0], (const double*)d.pts_buf[0]);
      k_comp_fold_cost<<<grid_for(d.n_comp, 4), 256, 0, h->stream>>>(d);
      note_launch(h);
    }
  }

  // (with the coarse space a build on the side stream rewrites cinv[p] once the host has SEEN iteration coarse_dirty[p]
  // finish: with one iteration per batch the iteration in flight always has the other parity; longer batches would
  // let the build race the solves of its own parity that are still queued, so the coarse space polls every iteration)
  const int check_every =
      h->coarse ? 1
                : (h->small_solve || h->comp_solve || h->dense_solve || h->banded_solve || h->bcr_solve) ? std::max(1, o.lm_check_every) : 1;
  h->mirror->seq = 0;
  if (h->coarse) {  // no build of an earlier solve in flight, clear sums, no inverse yet
    HIP_TRY(h, hipStreamSynchronize(h->side));
    HIP_TRY(h, hipMemsetAsync(h->csum[0], 0, (size_t)2 * d.c_nc * d.c_nc * sizeof(double), h->stream));
    h->coarse_seq = 0;
    for (int p = 0; p < 2; ++p) { h->coarse_dirty[p] = -1; h->coarse_inflight[p] = h->coarse_have[p] = false; }
  }
  int enq = 0, observed = 0, rc;
  bool done = false;
  // On the staged path with a solver that keeps no state in the scalar block, the accumulators clean themselves
  // (k_gather_S zeroes the U / gc slabs it has summed, k_lm_decide the statistics it has folded): one memset per
  // solve instead of one per iteration.
  const bool self_cleaning =
      h->schur_cpb > 0 && (h->small_solve || h->comp_solve || h->dense_solve || h->banded_solve || h->bcr_solve);
  auto enqueue_batch = [&]() -> int {
    const int nb = std::min(check_every, o.max_iters - enq);
    for (int i = 0; i < nb; ++i, ++enq) {
      h->span_iter = enq;
      // a linearisation may or may not run in this iteration (device decides): always clear its slabs
      if (!self_cleaning && enq > 0 && (rc = clear_accumulators(h, true))) return rc;
      enqueue_linearize(h, nullptr, nullptr);
      int pcg_it = 0;
      if ((rc = do_reduced_solve(h, -1.0, &pcg_it))) return rc;
      enqueue_trial(h, -1.0, nullptr, nullptr, nullptr, nullptr);
      if (h->comp_lm) {
        // every connected component decides for itself; the accepted ones move into the current buffers
        launch_timed(h, KT_DECIDE, k_lm_decide_comp, 1, DECIDE_COMP_THREADS, d, h->mirror_dev, enq + 1);
        launch_timed(h, KT_OTHER, k_commit_comp, std::min(2048, grid_for(std::max(d.L_obs, d.Cf))), 256, d);
      } else if (h->world > 1) {
        // fold on the device, combine the ranks with one sum all-reduce (the max of the point-gradient norm
        // travels in per-rank slots)
        k_fold_scalars<<<1, 128, 0, h->stream>>>(d, h->scal_fold, h->rank, h->world); note_launch(h);
        NCCL_TRY(h, h->rccl->AllReduce(h->scal_fold, h->scal_fold, (size_t)SC_COUNT + h->world, ncclDouble, ncclSum,
                                       h->comm, h->stream));
        launch_timed(h, KT_DECIDE, k_lm_decide, 1, 256, d, (const double*)h->scal_fold, h->world, h->mirror_dev, enq + 1);
      } else {
        launch_timed(h, KT_DECIDE, k_lm_decide, 1, 256, d, (const double*)nullptr, 1, h->mirror_dev, enq + 1);
      }
    }
    return BA_OK;
  };
  // One batch AHEAD: the host puts the next batch into the stream before it waits for the current one, so the GPU never
  // idles through the round trip (k_lm_decide -> pinned memory -> host -> six launches: 8 us per iteration at config
  // 2, the only gap of an iteration in the kernel trace).  The batch that turns out to follow termination exits at the
  // first instruction of each kernel (~15 us once per solve).  With several ranks that batch also runs its collectives
  // (once per solve, on every rank alike: the LM state is replicated bit for bit, so every rank's host sees `done` after
  // the same batch and has enqueued the same number of batches by then; BA_LM_AHEAD=0 on every rank restores the
  // lock-step order).
  const bool ahead = !h->coarse && env_int("BA_LM_AHEAD", 1) != 0;
  if (o.max_iters > 0 && (rc = enqueue_batch())) return rc;
  int wait_end = enq;  // the host waits until iterations [0, wait_end) have decided
  while (wait_end > 0) {
    const int waiting = wait_end;
    if (ahead && enq < o.max_iters && (rc = enqueue_batch())) return rc;
    wait_end = enq;
    if (h->coarse) {
      // the iterations that the host has SEEN finish left final coarse sums: their inverses are built on the second
      // stream, beside the iterations just enqueued (launched after them: the launch costs the host a few us)
      for (int p = 0; p < 2; ++p)
        if (h->coarse_dirty[p] >= 0 && h->coarse_dirty[p] < observed && !h->coarse_inflight[p]) {
          k_coarse_build<<<1, COARSE_THREADS, h->coarse_lds, h->side>>>(d, h->csum[p], h->cinv[p]);
          note_launch(h);
          HIP_TRY(h, hipEventRecord(h->ev_coarse[p], h->side));
          h->coarse_inflight[p] = true;
          h->coarse_dirty[p] = -1;
        }
    }
    // k_lm_decide stores the control head into pinned host memory and then raises seq: spin on it (a copy +
    // stream synchronisation costs ~20 us of idle GPU per iteration); long iterations fall back to a blocking wait
    const auto spin_until = clk::now() + std::chrono::milliseconds(2);
    while (__atomic_load_n(&h->mirror->seq, __ATOMIC_ACQUIRE) < waiting) {
      if (clk::now() > spin_until) {
        HIP_TRY(h, hipStreamSynchronize(h->stream));
        if (__atomic_load_n(&h->mirror->seq, __ATOMIC_ACQUIRE) < waiting)
          return check_launches(h, "solve") ? BA_ERR_DEVICE
                                            : fail(h, BA_ERR_DEVICE, "the LM control block did not reach the host");
        break;
      }
    }
    done = h->mirror->head.done != 0;
    observed = waiting;
    if (done) break;
    if (wait_end == waiting) {  // nothing further in the stream (not ahead, or the iteration cap was reached)
      if (enq >= o.max_iters) break;
      if ((rc = enqueue_batch())) return rc;
      wait_end = enq;
    }
  }
  HIP_TRY(h, hipMemcpyAsync(&c, d.ctrl, sizeof c, hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipStreamSynchronize(h->stream));
  rep->solve_ms = ms_since(t0);
  collect_spans(h, c);
  if ((rc = check_launches(h, "solve"))) return rc;
  if (c.error == 1) return fail(h, BA_ERR_NUMERICAL, "cannot project points with zero Z coordinates (division by zero)");
  if (c.error == 2) return fail(h, BA_ERR_NUMERICAL, "non-finite cost at the initial guess");
  h->poses_cur = d.poses_buf[c.sel];
  h->pts_cur = d.pts_buf[c.sel];
  rep->cost0 = c.cost0;
  rep->cost = c.cost;
  rep->grad_inf = c.ginf;
  rep->lambda = c.lam;
  rep->iterations = c.it;
  rep->accepted = c.accepted;
  rep->linearizations = c.linearizations;
  rep->status = c.status;
  rep->pcg_iters = c.pcg_total;
  rep->bad_pivots = c.bad_pivots;
  rep->guard_rejections = c.guard_rejections;
  rep->shortened_steps = c.shortened_steps;
  rep->rms_reproj = h->n_global > 0 ? std::sqrt(2.0 * c.cost / (double)h->n_global) : 0.0;
  rep->n_history = c.n_hist;
  for (int i = 0; i < c.n_hist; ++i) {
    rep->cost_history[i] = c.cost_hist[i];
    rep->lambda_history[i] = c.lam_hist[i];
    rep->pcg_history[i] = c.pcg_hist[i];
    rep->accepted_history[i] = c.acc_hist[i];
    if (o.verbose)
      fprintf(stderr, "[ba] it %3d cost %.12e trial %.12e rho %8.3g lam %.2e |g| %.2e |dx| %.2e pcg %d %s\n", i + 1,
              c.cost_hist[i], c.trial_hist[i], c.rho_hist[i], c.lam_hist[i], c.ginf_hist[i], c.step_hist[i],
              c.pcg_hist[i], c.acc_hist[i] ? "accept" : "reject");
  }
  h->has_estimate = true;
  return BA_OK;
}

}  // namespace

// ================================================================================================ C ABI
extern "C" {

int ba_version(void) { return BA_AMD_VERSION; }

int ba_device_count(void) {
  int n = 0;
  if (hipGetDeviceCount(&n) != hipSuccess) return 0;
  return n;
}

void ba_default_options(ba_options* o) {
  if (!o) return;
  o->max_iters = 2000;  // the reference's cap: max_nfev=2000 (bundle_adjustment.py:210)
  o->ftol = 1e-8;
  o->xtol = 1e-12;
  o->gtol = 1e-10;
  o->lambda0 = 1e-4;
  o->lambda_min = 1e-12;
  o->lambda_max = 1e12;
  o->min_diag = 1e-12;
  o->accept_ratio = 1e-4;
  o->guard_cheirality = 1;
  o->linear_solver = BA_SOLVER_AUTO;
  o->pcg_tol = 3e-2;
  o->pcg_max_iter = 1000;
  o->pcg_check_every = 10;
  o->lm_check_every = 1;
  o->device = -1;
  o->verbose = 0;
  o->profile_kernels = 0;
  o->loss = BA_LOSS_LINEAR;
  o->f_scale = 1.0;
}

int ba_create(const ba_options* opt, ba_handle** out) {
  if (!out) return BA_ERR_INVALID;
  *out = nullptr;
  ba_handle* h = new (std::nothrow) ba_handle();
  if (!h) return BA_ERR_NOMEM;
  if (opt) h->opt = *opt; else ba_default_options(&h->opt);
  if (h->opt.loss < BA_LOSS_LINEAR || h->opt.loss > BA_LOSS_ARCTAN || !(h->opt.f_scale > 0.0) ||
      !std::isfinite(h->opt.f_scale)) {
    delete h;
    return BA_ERR_INVALID;  // scipy raises ValueError for an unknown loss or f_scale <= 0 as well
  }
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) {
    delete h;
    return BA_ERR_DEVICE;  // no GPU: the product path fails loudly, there is no CPU fallback
  }
  int dev = h->opt.device;
  if (dev < 0) (void)hipGetDevice(&dev);
  if (dev >= ndev) { delete h; return BA_ERR_INVALID; }
  h->device = dev;
  if (hipSetDevice(dev) != hipSuccess || hipStreamCreate(&h->stream) != hipSuccess ||
      hipHostMalloc((void**)&h->scal_host, SCAL_BYTES) != hipSuccess ||
      hipHostMalloc((void**)&h->ctrl_host, sizeof(LmCtrl)) != hipSuccess ||
      hipHostMalloc((void**)&h->mirror, sizeof(LmMirror), hipHostMallocMapped) != hipSuccess ||
      hipHostGetDevicePointer((void**)&h->mirror_dev, h->mirror, 0) != hipSuccess) {
    delete h;
    return BA_ERR_DEVICE;
  }
  *out = h;
  return BA_OK;
}

void ba_destroy(ba_handle* h) {
  if (!h) return;
  (void)hipSetDevice(h->device);
  free_problem(h);
  if (h->comm && h->rccl) (void)h->rccl->CommDestroy(h->comm);
  if (h->agree_dev) (void)hipFree(h->agree_dev);
  if (h->scal_host) (void)hipHostFree(h->scal_host);
  if (h->ctrl_host) (void)hipHostFree(h->ctrl_host);
  if (h->mirror) (void)hipHostFree(h->mirror);
  for (hipEvent_t e : h->ev_pool) (void)hipEventDestroy(e);
  for (hipEvent_t e : h->ev_coarse)
    if (e) (void)hipEventDestroy(e);
  if (h->side) (void)hipStreamDestroy(h->side);
  if (h->stream) (void)hipStreamDestroy(h->stream);
  delete h;
}

const char* ba_last_error(const ba_handle* h) { return h ? h->err.c_str() : "null handle"; }

int ba_get_kernel_times(ba_handle* h, double* total_ms, int32_t* launches) {
  if (!h || !total_ms || !launches) return BA_ERR_INVALID;
  for (int i = 0; i < BA_KERNEL_CLASSES; ++i) { total_ms[i] = h->ktime_ms[i]; launches[i] = h->kcount[i]; }
  return BA_OK;
}

int ba_set_kernel_timing(ba_handle* h, int32_t on) {
  if (!h) return BA_ERR_INVALID;
  h->opt.profile_kernels = on ? 1 : 0;
  return BA_OK;
}

int ba_comm_unique_id(char* id128) {
  Rccl* r = rccl_load();
  if (!r || !id128) return BA_ERR_DEVICE;
  ncclUniqueId id;
  if (r->GetUniqueId(&id) != ncclSuccess) return BA_ERR_DEVICE;
  std::memcpy(id128, id.internal, NCCL_UNIQUE_ID_BYTES);
  return BA_OK;
}

int ba_comm_init(ba_handle* h, int32_t rank, int32_t world, const char* id128) {
  if (!h) return BA_ERR_INVALID;
  if (world < 1 || world > FOLD_MAX_RANKS || rank < 0 || rank >= world || !id128)
    return fail(h, BA_ERR_INVALID, "bad rank/world/id (at most %d ranks)", FOLD_MAX_RANKS);
  if (h->comm) return fail(h, BA_ERR_STATE, "communicator already initialised");
  if (world > 1 && !h->prior.cams.empty()) return fail(h, BA_ERR_STATE, "comm_init: a pose prior is set, which several ranks do not support");
  h->rccl = rccl_load();
  if (!h->rccl) return fail(h, BA_ERR_DEVICE, "librccl.so could not be loaded: %s", dlerror());
  HIP_TRY(h, hipSetDevice(h->device));
  ncclUniqueId id;
  std::memcpy(id.internal, id128, NCCL_UNIQUE_ID_BYTES);
  HIP_TRY(h, hipMalloc((void**)&h->agree_dev, 2 * sizeof(double)));
  NCCL_TRY(h, h->rccl->CommInitRank(&h->comm, world, id, rank));
  h->rank = rank;
  h->world = world;
  free_problem(h);  // a problem set before the communicator existed has a rank-local pattern
  return BA_OK;
}

int ba_debug_table_digests(const ba_handle* h, uint64_t* out, int32_t cap, int32_t* count) {
  if (!h || !count || cap < 0 || (cap > 0 && !out)) return BA_ERR_INVALID;
  *count = (int32_t)h->table_digests.size();
  for (int32_t i = 0; i < std::min(cap, *count); ++i) out[i] = h->table_digests[(size_t)i];
  return BA_OK;
}

int ba_comm_info(const ba_handle* h, int32_t* rank, int32_t* world) {
  if (!h || !rank || !world) return BA_ERR_INVALID;
  *rank = 0;
  *world = 1;
  if (!h->comm) return BA_OK;  // no communicator: one GPU
  int n = 0, r = 0;  // what the communicator itself reports, not what ba_comm_init was told
  if (h->rccl->CommCount(h->comm, &n) != ncclSuccess || h->rccl->CommUserRank(h->comm, &r) != ncclSuccess)
    return BA_ERR_DEVICE;
  *rank = r;
  *world = n;
  return BA_OK;
}

int ba_reduced_solver(const ba_handle* h, int32_t* kind) {
  if (!h || !kind) return BA_ERR_INVALID;
  if (!h->has_problem) return BA_ERR_STATE;
  *kind = h->bcr_solve ? 5 : h->comp_solve ? 4 : h->banded_solve ? 3 : h->dense_solve ? 2 : h->small_solve ? 1 : 0;
  return BA_OK;
}

int ba_problem_info(const ba_handle* h, int64_t* info) {
  if (!h || !info) return BA_ERR_INVALID;
  if (!h->has_problem) return BA_ERR_STATE;
  const int64_t rows = h->schur_cpb > 0 && h->d.n_chunks > 0 ? (int64_t)h->contrib_rows : 0;
  const int64_t v[BA_INFO_COUNT] = {h->d.nu, rows, h->d.n_chunks, h->d.nslab_u,
                                    h->bcr_solve ? 5 : h->comp_solve ? 4 : h->banded_solve ? 3 : h->dense_solve ? 2 : h->small_solve ? 1 : 0,
                                    h->bcr_solve ? h->bcr_plan.m / 6 : h->banded_solve ? h->bshape.bw : 0,
                                    h->bcr_solve ? h->bcr_plan.n_levels : h->banded_solve ? h->bshape.na : 0,
                                    h->bcr_solve ? h->bcr_plan.N : h->bseg_P,
                                    h->Cf, h->L_obs, (h->lin_cpw > 0 ? 1 : 0) | (h->flat_chunks > 0 && h->use_flat_backsub ? 2 : 0) | (h->matrix_free ? 4 : 0) | (h->lin_win ? 8 : 0) | (h->lin_rows ? 16 : 0) | (h->sharded ? 32 : 0),
                                    h->d.n_comp, (int64_t)h->d.n_urows};
  for (int i = 0; i < BA_INFO_COUNT; ++i) info[i] = v[i];
  return BA_OK;
}

int ba_schur_block_count(const ba_handle* h, int64_t* nu) {
  if (!h || !nu || !h->has_problem) return BA_ERR_STATE;
  *nu = h->d.nu;
  return BA_OK;
}

}  // extern "C"

namespace {
int set_problem_impl(ba_handle* h, int32_t C, int32_t L, int32_t n, const int32_t* obs_cam, const int32_t* obs_pt,
                     const double* obs_uv, const double* intr, const uint8_t* fixed_in);

// everything that can be wrong with the arguments of ba_set_problem, checked before any state changes
int validate_problem(ba_handle* h, int32_t C, int32_t L, int32_t n, const int32_t* obs_cam, const int32_t* obs_pt,
                     const double* obs_uv, const double* intr, const uint8_t* fixed_in) {
  if (C <= 0 || L <= 0 || n <= 0 || !obs_cam || !obs_pt || !obs_uv || !intr)
    return fail(h, BA_ERR_INVALID, "set_problem: empty problem or null pointer (C=%d L=%d n=%d)%s", C, L, n,
                h->world > 1 ? "; in a multi-GPU solve every rank needs at least one landmark and one observation" : "");
  for (int i = 0; i < 7; ++i)
    if (!std::isfinite(intr[i])) return fail(h, BA_ERR_INVALID, "intrinsics[%d] is not finite", i);
  if (!(intr[0] > 0) || !(intr[1] > 0)) return fail(h, BA_ERR_INVALID, "focal lengths must be positive");
  {
    // the first observation that is wrong, if any (ranges of the list on host threads; the smallest index wins, as a
    // plain loop would report it)
    const int T = hostpar::threads_for(n, 65536);
    std::vector<long> first_bad((size_t)T, -1);
    hostpar::run(T, [&](int t, int nt) {
      for (long k = hostpar::part_begin(n, t, nt); k < hostpar::part_begin(n, t + 1, nt); ++k)
        if (obs_cam[k] < 0 || obs_cam[k] >= C || obs_pt[k] < 0 || obs_pt[k] >= L || !std::isfinite(obs_uv[2 * (size_t)k]) ||
            !std::isfinite(obs_uv[2 * (size_t)k + 1])) { first_bad[(size_t)t] = k; break; }
    });
    for (long kb : first_bad) {
      if (kb < 0) continue;
      const int k = (int)kb;
      if (obs_cam[k] < 0 || obs_cam[k] >= C) return fail(h, BA_ERR_INVALID, "obs_cam[%d]=%d out of range", k, obs_cam[k]);
      if (obs_pt[k] < 0 || obs_pt[k] >= L) return fail(h, BA_ERR_INVALID, "obs_pt[%d]=%d out of range", k, obs_pt[k]);
      return fail(h, BA_ERR_INVALID, "observation %d is not finite", k);
    }
  }
  if (!h->prior.cams.empty()) {
    if (h->world > 1) return fail(h, BA_ERR_STATE, "set_problem: a pose prior is set, which several ranks do not support");
    for (int c : h->prior.cams) {
      if (c >= C) return fail(h, BA_ERR_INVALID, "pose prior: camera %d is not a camera of this problem (C=%d)", c, C);
      if (fixed_in ? fixed_in[c] != 0 : c == 0) return fail(h, BA_ERR_INVALID, "pose prior: camera %d is fixed", c);
    }
  }
  if (fixed_in) {
    int nfree = 0;
    for (int c = 0; c < C; ++c) nfree += fixed_in[c] ? 0 : 1;
    if (nfree == 0) return fail(h, BA_ERR_INVALID, "every pose is fixed: nothing to optimise");
  } else if (C < 2) {
    return fail(h, BA_ERR_INVALID, "every pose is fixed: nothing to optimise");
  }
  return BA_OK;
}
}  // namespace

extern "C" {

int ba_set_pose_prior(ba_handle* h, int32_t n_prior, const int32_t* cams, const double* ref_poses, const double* H,
                      const double* g, double c0) {
  if (!h) return BA_ERR_INVALID;
  if (n_prior < 0 || n_prior > BA_PRIOR_MAX_CAMS)
    return fail(h, BA_ERR_INVALID, "pose prior: %d cameras (at most %d)", n_prior, BA_PRIOR_MAX_CAMS);
  if (n_prior > 0 && (!cams || !ref_poses || !H || !g)) return fail(h, BA_ERR_INVALID, "pose prior: null pointer");
  if (n_prior > 0 && h->world > 1) return fail(h, BA_ERR_STATE, "pose prior: not available with several ranks");
  const int n6 = 6 * n_prior;
  for (int a = 0; a < n_prior; ++a) {
    if (cams[a] < 0) return fail(h, BA_ERR_INVALID, "pose prior: camera index %d", cams[a]);
    for (int b = 0; b < a; ++b)
      if (cams[a] == cams[b]) return fail(h, BA_ERR_INVALID, "pose prior: camera %d appears twice", cams[a]);
    double nq = 0.0;
    for (int i = 0; i < 7; ++i) {
      if (!std::isfinite(ref_poses[7 * a + i])) return fail(h, BA_ERR_INVALID, "pose prior: reference pose %d is not finite", a);
      if (i >= 3) nq += ref_poses[7 * a + i] * ref_poses[7 * a + i];
    }
    if (std::fabs(std::sqrt(nq) - 1.0) > 1e-6) return fail(h, BA_ERR_INVALID, "pose prior: reference quaternion %d is not unit length", a);
  }
  for (int i = 0; i < n6; ++i) {
    if (!std::isfinite(g[i])) return fail(h, BA_ERR_INVALID, "pose prior: g[%d] is not finite", i);
    for (int j = 0; j < n6; ++j) {
      const double v = H[(size_t)i * n6 + j], w = H[(size_t)j * n6 + i];
      if (!std::isfinite(v)) return fail(h, BA_ERR_INVALID, "pose prior: H[%d][%d] is not finite", i, j);
      if (std::fabs(v - w) > 1e-9 * (std::fabs(v) + std::fabs(w)) + 1e-300)
        return fail(h, BA_ERR_INVALID, "pose prior: H is not symmetric at [%d][%d]", i, j);
    }
    if (H[(size_t)i * n6 + i] < 0.0) return fail(h, BA_ERR_INVALID, "pose prior: H[%d][%d] < 0", i, i);
  }
  if (!std::isfinite(c0)) return fail(h, BA_ERR_INVALID, "pose prior: c0 is not finite");
  h->prior.cams.assign(cams, cams + n_prior);
  h->prior.ref.assign(ref_poses, ref_poses + (size_t)7 * n_prior);
  h->prior.H.assign(H, H + (size_t)n6 * n6);
  h->prior.g.assign(g, g + n6);
  h->prior.c0 = n_prior > 0 ? c0 : 0.0;
  free_problem(h);  // the block pattern of a problem set earlier does not hold the prior's camera pairs
  return BA_OK;
}

int ba_set_problem(ba_handle* h, int32_t C, int32_t L, int32_t n, const int32_t* obs_cam, const int32_t* obs_pt,
                   const double* obs_uv, const double* intr, const uint8_t* fixed_in) {
  if (!h) return BA_ERR_INVALID;
  h->agree_failed = false;
  // multi-GPU: no rank goes on to the pattern-union collectives unless every rank's arguments are good, and no
  // rank keeps the problem unless every rank built it
  int rc = agree_status(h, validate_problem(h, C, L, n, obs_cam, obs_pt, obs_uv, intr, fixed_in), "set_problem");
  if (rc) return rc;
  try {  // (the tables are std::vectors, built on several host threads: no exception may cross the C boundary)
    rc = set_problem_impl(h, C, L, n, obs_cam, obs_pt, obs_uv, intr, fixed_in);
  } catch (const std::bad_alloc&) {
    rc = fail(h, BA_ERR_NOMEM, "set_problem: out of host memory while building the index tables");
  } catch (const std::exception& e) {
    rc = fail(h, BA_ERR_DEVICE, "set_problem: %s", e.what());
  }
  if (!h->agree_failed) rc = agree_status(h, rc, "set_problem");
  if (rc) free_problem(h);
  return rc;
}

}  // extern "C"

namespace {
int set_problem_impl(ba_handle* h, int32_t C, int32_t L, int32_t n, const int32_t* obs_cam, const int32_t* obs_pt,
                     const double* obs_uv, const double* intr, const uint8_t* fixed_in) {
  const auto t0 = clk::now();
  // BA_SETUP_TRACE=1: wall time of every stage of the set-up to stderr
  const bool trace = env_int("BA_SETUP_TRACE", 0) != 0;
  auto t_lap = t0;
  auto lap = [&](const char* what) {
    if (!trace) return;
    const auto now = clk::now();
    std::fprintf(stderr, "[ba_set_problem] %-28s %8.3f ms\n", what, std::chrono::duration<double, std::milli>(now - t_lap).count());
    t_lap = now;
  };
  HIP_TRY(h, hipSetDevice(h->device));
  free_problem(h);
  h->C = C; h->L = L; h->n = n;
  h->digest_on = env_int("BA_TABLE_DIGEST", 0) != 0;
  h->table_digests.clear();
  lap("free the previous problem");

  // ---- cameras
  h->fixed.assign(C, 0);
  if (fixed_in) { for (int c = 0; c < C; ++c) h->fixed[c] = fixed_in[c] ? 1 : 0; } else h->fixed[0] = 1;
  h->cam2free.assign(C, -1);
  h->free2cam.clear();
  for (int c = 0; c < C; ++c)
    if (!h->fixed[c]) { h->cam2free[c] = (int)h->free2cam.size(); h->free2cam.push_back(c); }
  h->Cf = (int)h->free2cam.size();
  if (h->Cf == 0) return fail(h, BA_ERR_INVALID, "every pose is fixed: nothing to optimise");
  const int Cf = h->Cf;

  // ---- tracks: observations grouped by landmark, camera-sorted inside a track.  On host threads: every thread owns a
  //      range of landmarks and scans the observation list for them (a counting sort without shared counters)
  std::vector<int> uptr(L + 1, 0);
  hostpar::raw_vector<int> track(n);
  {
    const int T = hostpar::threads_for(n);
    std::vector<int> cnt(L + 1, 0);
    hostpar::for_range(L, T, [&](long l0, long l1) {
      if (T == 1) { for (int k = 0; k < n; ++k) cnt[obs_pt[k] + 1]++; return; }
      for (int k = 0; k < n; ++k) { const int j = obs_pt[k]; if (j >= l0 && j < l1) cnt[j + 1]++; }
    });
    for (int j = 0; j < L; ++j) uptr[j + 1] = uptr[j] + cnt[j + 1];
    std::vector<int>& fill = cnt;  // reused: next free entry of every track
    hostpar::for_range(L, T, [&](long l0, long l1) {
      for (long j = l0; j < l1; ++j) fill[j] = uptr[j];
      if (T == 1) { for (int k = 0; k < n; ++k) track[fill[obs_pt[k]]++] = k; }
      else for (int k = 0; k < n; ++k) { const int j = obs_pt[k]; if (j >= l0 && j < l1) track[fill[j]++] = k; }
      for (long j = l0; j < l1; ++j) {
        int* b = track.data() + uptr[j];
        int* e = track.data() + uptr[j + 1];
        // entries arrive in ascending k: image-major input (the reference's packing order, bundle_adjustment.py:178-194)
        // is camera-sorted already
        bool sorted = true;
        for (int* q = b; q + 1 < e; ++q)
          if (obs_cam[q[0]] > obs_cam[q[1]]) { sorted = false; break; }
        if (!sorted) std::sort(b, e, [&](int a, int c2) { return obs_cam[a] != obs_cam[c2] ? obs_cam[a] < obs_cam[c2] : a < c2; });
      }
    });
  }
  lap("tracks");
  // ---- landmark order: observed first, lexicographic by camera signature (locality for the
  //      register run-length accumulation of k_linearize / k_schur), ties by index.  Sorted by a 64-bit key that packs
  //      the first cameras of the signature (camera + 1 per field, 0 = past the end of the track, so that a signature
  //      which is a prefix of another sorts first, as comparing the lengths would have it); the rest of a long signature
  //      is looked at only between equal keys
  std::vector<int> order(L);
  int L_obs = 0;
  {
    int bits = 1;
    while (((long)C >> bits) != 0) ++bits;  // camera + 1 <= C needs `bits` bits
    const int slots = 64 / bits;
    struct Key { uint64_t key; int idx; };
    for (int j = 0; j < L; ++j) L_obs += uptr[j + 1] > uptr[j] ? 1 : 0;
    std::vector<Key> keyed((size_t)L_obs);
    {
      int a = 0, b = L_obs;  // observed landmarks to the front (sorted below), the others behind them in index order
      for (int j = 0; j < L; ++j) {
        if (uptr[j + 1] > uptr[j]) keyed[(size_t)a++].idx = j;
        else order[b++] = j;
      }
    }
    const int T = hostpar::threads_for(L_obs, 8192);
    hostpar::for_range(L_obs, T, [&](long i0, long i1) {
      for (long i = i0; i < i1; ++i) {
        const int j = keyed[(size_t)i].idx, len = uptr[j + 1] - uptr[j];
        uint64_t key = 0;
        for (int q = 0; q < slots; ++q) key = (key << bits) | (q < len ? (uint64_t)(obs_cam[track[uptr[j] + q]] + 1) : 0u);
        keyed[(size_t)i].key = key;
      }
    });
    auto key_less = [&](const Key& x, const Key& y) {
      if (x.key != y.key) return x.key < y.key;
      const int a = x.idx, b = y.idx;
      const int la = uptr[a + 1] - uptr[a], lb = uptr[b + 1] - uptr[b];
      const int m = std::min(la, lb);
      for (int i = slots; i < m; ++i) {
        const int ca = obs_cam[track[uptr[a] + i]], cb = obs_cam[track[uptr[b] + i]];
        if (ca != cb) return ca < cb;
      }
      if (la > slots || lb > slots) { if (la != lb) return la < lb; }
      return a < b;
    };
    hostpar::sort(keyed, key_less, T);
    hostpar::for_range(L_obs, T, [&](long i0, long i1) { for (long i = i0; i < i1; ++i) order[i] = keyed[(size_t)i].idx; });
  }
  h->lm_internal_to_user = order;
  h->lm_user_to_internal.assign(L, 0);
  for (int j = 0; j < L; ++j) h->lm_user_to_internal[order[j]] = j;
  h->L_obs = L_obs;
  lap("landmark order");

  hostpar::raw_vector<int> o_cam(n), o_src(n), o_lm(n), o_free(n);  // written in full below, by the threads that first touch them
  hostpar::raw_vector<double> o_u(n), o_v(n);
  std::vector<int> lm_ptr(L_obs + 1, 0);
  {
    for (int j = 0; j < L_obs; ++j) lm_ptr[j + 1] = lm_ptr[j] + (uptr[order[j] + 1] - uptr[order[j]]);
    hostpar::for_range(L_obs, hostpar::threads_for(n), [&](long j0, long j1) {
      for (long j = j0; j < j1; ++j) {
        const int uj = order[j];
        int k2 = lm_ptr[j];
        for (int e = uptr[uj]; e < uptr[uj + 1]; ++e, ++k2) {
          const int k = track[e];
          o_cam[k2] = obs_cam[k]; o_src[k2] = k; o_lm[k2] = (int)j;
          o_u[k2] = obs_uv[2 * (size_t)k]; o_v[k2] = obs_uv[2 * (size_t)k + 1];
          o_free[k2] = h->cam2free[obs_cam[k]];
        }
      }
    });
  }
  h->obs_src.assign(o_src.begin(), o_src.end());
  lap("observation arrays");
  // The per-observation arrays are the bulk of what goes to the device.  Their allocation and their (blocking, pageable)
  // copies run on a helper thread from here on, beside the construction of the tables below; joined after dev_commit.
  // The thread writes six pointers of h->d that nobody reads before the join, and nothing else of the handle.
  struct EarlyUpload {
    std::thread th;
    void* base = nullptr;
    hipError_t err = hipSuccess;
    bool kept = false;
    ~EarlyUpload() {
      if (th.joinable()) th.join();
      if (base && !kept) (void)hipFree(base);
    }
  } early;
  {
    h->pending.clear();
    queue_upload(h, &h->d.o_cam, o_cam); queue_upload(h, &h->d.o_u, o_u); queue_upload(h, &h->d.o_v, o_v);
    queue_upload(h, &h->d.o_src, o_src); queue_upload(h, &h->d.o_lm, o_lm); queue_upload(h, &h->d.o_free, o_free);
    auto list = std::make_shared<std::vector<PendingDev>>();
    list->swap(h->pending);
    EarlyUpload* eu = &early;
    const int dev = h->device;
    auto work = [list, eu, dev] {
      if ((eu->err = hipSetDevice(dev)) != hipSuccess) return;
      eu->err = commit_list(*list, &eu->base);
    };
    try {
      early.th = std::thread(work);
    } catch (const std::system_error&) {  // no thread to be had: the copies run here, before the tables
      work();
    }
  }
  // ---- block pattern of the reduced camera system (upper triangle over free cameras)
  std::vector<int64_t> keys;
  for (int f = 0; f < Cf; ++f) keys.push_back((int64_t)f * Cf + f);
  for (int a : h->prior.cams)  // the prior couples its cameras with each other, landmarks in common or not
    for (int b : h->prior.cams) {
      const int fa = h->cam2free[a], fb = h->cam2free[b];
      if (fa < fb) keys.push_back((int64_t)fa * Cf + fb);
    }
  {
    // pairs of free cameras that share a landmark: landmark ranges on host threads, each with its own key list (runs of
    // landmarks with one signature contribute once), then one sorted, duplicate-free list: whatever the cut, the same list
    const int T = hostpar::threads_for(n, 65536);
    std::vector<std::vector<int64_t>> part((size_t)T);
    hostpar::run(T, [&](int t, int nt) {
      std::vector<int64_t>& out = part[(size_t)t];
      std::vector<int> prev, cur;
      for (long j = hostpar::part_begin(L_obs, t, nt); j < hostpar::part_begin(L_obs, t + 1, nt); ++j) {
        cur.clear();
        for (int k = lm_ptr[j]; k < lm_ptr[j + 1]; ++k) {
          const int f = o_free[k];
          if (f >= 0 && (cur.empty() || cur.back() != f)) cur.push_back(f);
        }
        if (cur == prev) continue;
        for (size_t a = 0; a < cur.size(); ++a)
          for (size_t b = a + 1; b < cur.size(); ++b) out.push_back((int64_t)cur[a] * Cf + cur[b]);
        prev.swap(cur);
        if (out.size() > ((size_t)64 << 20) / (size_t)nt) {  // keep the key lists bounded on unstructured scenes
          std::sort(out.begin(), out.end());
          out.erase(std::unique(out.begin(), out.end()), out.end());
        }
      }
      std::sort(out.begin(), out.end());
      out.erase(std::unique(out.begin(), out.end()), out.end());
    });
    for (auto& v : part) keys.insert(keys.end(), v.begin(), v.end());
  }
  std::sort(keys.begin(), keys.end());
  keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  h->n_global = n;
  if (h->world > 1) {
    // Every rank must use the SAME block pattern (the all-reduce sums S slot by slot): take the union of
    // the ranks' patterns.  Also sum the observation counts (for the RMS reprojection error).
    double cnt[2] = {(double)keys.size(), (double)n};
    DevTmp t_cnt, t_mine, t_all;
    int rc_a = hipMalloc(&t_cnt.p, 4 * sizeof(double)) == hipSuccess ? BA_OK : fail(h, BA_ERR_NOMEM, "set_problem: device allocation failed");
    if ((rc_a = agree_status(h, rc_a, "set_problem"))) return rc_a;
    double* d_cnt = static_cast<double*>(t_cnt.p);
    HIP_TRY(h, hipMemcpy(d_cnt, cnt, 2 * sizeof(double), hipMemcpyHostToDevice));
    NCCL_TRY(h, h->rccl->GroupStart());
    NCCL_TRY(h, h->rccl->AllReduce(d_cnt, d_cnt + 2, 1, ncclDouble, ncclMax, h->comm, h->stream));
    NCCL_TRY(h, h->rccl->AllReduce(d_cnt + 1, d_cnt + 3, 1, ncclDouble, ncclSum, h->comm, h->stream));
    NCCL_TRY(h, h->rccl->GroupEnd());
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(cnt, d_cnt + 2, 2 * sizeof(double), hipMemcpyDeviceToHost));
    const size_t kmax = (size_t)cnt[0];
    h->n_global = (long)cnt[1];
    std::vector<int64_t> mine(kmax, -1), all((size_t)h->world * kmax);
    std::copy(keys.begin(), keys.end(), mine.begin());
    rc_a = (hipMalloc(&t_mine.p, kmax * sizeof(int64_t)) == hipSuccess &&
            hipMalloc(&t_all.p, all.size() * sizeof(int64_t)) == hipSuccess)
               ? BA_OK : fail(h, BA_ERR_NOMEM, "set_problem: device allocation failed");
    if ((rc_a = agree_status(h, rc_a, "set_problem"))) return rc_a;
    int64_t* d_mine = static_cast<int64_t*>(t_mine.p);
    int64_t* d_all = static_cast<int64_t*>(t_all.p);
    HIP_TRY(h, hipMemcpy(d_mine, mine.data(), kmax * sizeof(int64_t), hipMemcpyHostToDevice));
    NCCL_TRY(h, h->rccl->AllGather(d_mine, d_all, kmax, ncclInt64, h->comm, h->stream));
    HIP_TRY(h, hipStreamSynchronize(h->stream));
    HIP_TRY(h, hipMemcpy(all.data(), d_all, all.size() * sizeof(int64_t), hipMemcpyDeviceToHost));
    keys.clear();
    for (int64_t k : all)
      if (k >= 0) keys.push_back(k);
    std::sort(keys.begin(), keys.end());
    keys.erase(std::unique(keys.begin(), keys.end()), keys.end());
  }
  const int nu = (int)keys.size();
  // S upper blocks, rhs and Ugc share one allocation (the all-reduce buffer); the mirrored lower blocks
  // follow after a gap padded to whole blocks, so that "slot * 36" still addresses them
  const int gap_blocks = (int)((6L * Cf + 27L * C + 35) / 36);
  std::vector<int> urow_ptr(Cf + 1, 0), ucol(nu), urow(nu), diag_slot(Cf), low_of(nu, -1);
  for (int s = 0; s < nu; ++s) {
    urow[s] = (int)(keys[s] / Cf);
    ucol[s] = (int)(keys[s] % Cf);
    urow_ptr[urow[s] + 1]++;
    if (urow[s] == ucol[s]) diag_slot[urow[s]] = s;
  }
  for (int f = 0; f < Cf; ++f) urow_ptr[f + 1] += urow_ptr[f];
  int nl = 0;
  std::vector<int> brow_cnt(Cf + 1, 0);
  for (int s = 0; s < nu; ++s) {
    brow_cnt[urow[s] + 1]++;
    if (urow[s] != ucol[s]) { low_of[s] = nu + gap_blocks + nl++; brow_cnt[ucol[s] + 1]++; }
  }
  std::vector<int> brow_ptr(Cf + 1, 0);
  for (int f = 0; f < Cf; ++f) brow_ptr[f + 1] = brow_ptr[f] + brow_cnt[f + 1];
  const int nb = brow_ptr[Cf];
  std::vector<int> bcol(nb), bslot(nb);
  {
    // lower entries of row i (cols j < i) come from upper slots (j, i): visiting upper slots in
    // (row, col) order emits them in ascending j, ahead of the row's own upper entries
    std::vector<int> fill(brow_ptr.begin(), brow_ptr.end() - 1);
    for (int s = 0; s < nu; ++s)
      if (urow[s] != ucol[s]) { const int i = ucol[s]; bcol[fill[i]] = urow[s]; bslot[fill[i]++] = low_of[s]; }
    for (int s = 0; s < nu; ++s) { const int i = urow[s]; bcol[fill[i]] = ucol[s]; bslot[fill[i]++] = s; }
  }
  h->urow_ptr = urow_ptr;
  h->ucol = ucol;
  std::vector<int> sm_slot(nu);
  std::iota(sm_slot.begin(), sm_slot.end(), 0);
  std::sort(sm_slot.begin(), sm_slot.end(), [&](int a, int b) {
    const int da = ucol[a] - urow[a], db = ucol[b] - urow[b];
    return da != db ? da < db : urow[a] < urow[b];
  });

  lap("block pattern");
  // ---- launch geometry
  {
    std::vector<int> lens(L_obs);
    for (int j = 0; j < L_obs; ++j) lens[j] = lm_ptr[j + 1] - lm_ptr[j];
    std::vector<int> tmp(lens);
    const size_t q = (size_t)(0.9 * (tmp.size() - 1));
    std::nth_element(tmp.begin(), tmp.begin() + q, tmp.end());
    const int len90 = std::max(1, tmp[q]);
    h->g_lin = env_int("BA_G_LIN", pow2_at_least(len90, 8, 64));
    h->g_schur = env_int("BA_G_SCHUR", pow2_at_least(len90 * (len90 + 1) / 2, 8, 64));
    // waves per launch: enough to hide latency, few enough that a lane sees long runs of one camera
    const long waves_lin = 256L * env_int("BA_WAVES_LIN", 8), waves_schur = 256L * env_int("BA_WAVES_SCHUR", 4);
    const long gpw_lin = 64 / h->g_lin, gpw_schur = 64 / h->g_schur;
    h->lpg_lin = env_int("BA_LPG_LIN", (int)std::max<long>(1, (L_obs + waves_lin * gpw_lin - 1) / (waves_lin * gpw_lin)));
    h->lpg_schur = env_int("BA_LPG_SCHUR",
                           (int)std::max<long>(1, (L_obs + waves_schur * gpw_schur - 1) / (waves_schur * gpw_schur)));
    h->lpg_back = env_int("BA_LPG_BACK", 1);
  }
  // ---- k_linearize_flat chunks: whole tracks, <= 64 observations and <= LINF_MAXLM landmarks each
  std::vector<int> lin_lm(1, 0), lin_k(1, 0);
  h->lin_cpw = 0;
  h->flat_chunks = 0;
  h->use_flat_backsub = env_int("BA_BACKSUB_FLAT", 1) != 0;
  {
    int maxlen = 0;
    for (int j = 0; j < L_obs; ++j) maxlen = std::max(maxlen, lm_ptr[j + 1] - lm_ptr[j]);
    if (maxlen <= LINF_MAXLM && L_obs > 0) {
      lin_lm.clear(); lin_k.clear();
      for (int j = 0; j < L_obs;) {
        lin_lm.push_back(j);
        lin_k.push_back(lm_ptr[j]);
        int j1 = j + 1;
        while (j1 < L_obs && j1 - j < LINF_MAXLM && lm_ptr[j1 + 1] - lm_ptr[j] <= 64) ++j1;
        j = j1;
      }
      lin_lm.push_back(L_obs);
      lin_k.push_back(lm_ptr[L_obs]);
      h->flat_chunks = (int)lin_lm.size() - 1;
      if (env_int("BA_LIN_FLAT", 1) != 0 && C <= LINF_MAXC) h->lin_cpw = 1;
    }
  }
  // ---- k_linearize_win tables: per chunk the row of every observation in camera order (a permutation of the chunk's lanes)
  // and their camera groups; per workgroup (LINF_WAVES chunks) the cameras that get a slot in its LDS copy of U | gc
  std::vector<unsigned char> lin_perm((size_t)std::max(n, 1), 0);
  std::vector<int> lin_gptr(1, 0), lin_gbeg, lin_gcam, lin_gslot, lin_wincam;
  // (with <= LINF_MAXC cameras k_linearize_flat, whose LDS copy holds every camera, is the faster one: 17.5 vs 19.4 us
  // on BASELINE configs[1]; beyond, this kernel against k_linearize<G>: 58.6 vs 70.6 us on configs[2], 604 vs 840 on [3])
  // Round 3: k_linearize_rows takes every problem that has chunk tables (any number of cameras): no atomics, the same
  // bits in every run.  BA_LIN_ROWS=0 brings back round 2's kernels (k_linearize_flat / k_linearize_win: LDS and global
  // atomics) for timing comparisons.
  h->lin_rows = h->flat_chunks > 0 && env_int("BA_LIN_ROWS", 1) != 0;
  if (h->lin_rows) h->lin_cpw = 0;
  h->lin_win = !h->lin_rows && h->flat_chunks > 0 && env_int("BA_LIN_WIN", h->lin_cpw > 0 ? 0 : 1) != 0;
  std::vector<int> ucr_ptr(1, 0), ucr_row;
  int rows_nslab = 0;
  if (h->lin_win || h->lin_rows) {
    const int nwg = (h->flat_chunks + LINF_WAVES - 1) / LINF_WAVES;
    if (h->lin_win) lin_wincam.assign((size_t)nwg * LINW_WIN, -1);
    // workgroups are independent: ranges of them on host threads, each with its own piece of the group tables
    struct Piece { std::vector<int> gbeg, gcam, gslot, gend; };
    const int T = hostpar::threads_for(n, 16384);
    std::vector<Piece> piece((size_t)T);
    hostpar::run(T, [&](int t, int nt) {
      Piece& pc = piece[(size_t)t];
      std::vector<int> idx;
      for (int wg = (int)hostpar::part_begin(nwg, t, nt); wg < (int)hostpar::part_begin(nwg, t + 1, nt); ++wg) {
        int* win = h->lin_win ? lin_wincam.data() + (size_t)wg * LINW_WIN : nullptr;
        int nwin = 0;
        for (int c = wg * LINF_WAVES; c < std::min(h->flat_chunks, (wg + 1) * LINF_WAVES); ++c) {
          const int kb = lin_k[c], ke = lin_k[c + 1];
          idx.clear();
          for (int k = kb; k < ke; ++k)
            if (o_free[k] >= 0) idx.push_back(k);
          std::stable_sort(idx.begin(), idx.end(), [&](int a, int b) { return o_cam[a] < o_cam[b]; });
          {  // observations of fixed cameras take the rows after the free ones (their products are zero, nobody sums them)
            size_t tail = idx.size();
            for (int k = kb; k < ke; ++k)
              if (o_free[k] < 0) lin_perm[k] = (unsigned char)tail++;
          }
          for (size_t p2 = 0; p2 < idx.size(); ++p2) {
            lin_perm[idx[p2]] = (unsigned char)p2;
            if (p2 == 0 || o_cam[idx[p2]] != o_cam[idx[p2 - 1]]) {
              const int cam = o_cam[idx[p2]];
              int slot = -1;
              if (win) {
                for (int s2 = 0; s2 < nwin; ++s2)
                  if (win[s2] == cam) slot = s2;
                if (slot < 0 && nwin < LINW_WIN) { slot = nwin; win[nwin++] = cam; }
              }
              pc.gbeg.push_back((int)p2); pc.gcam.push_back(cam); pc.gslot.push_back(slot);
            }
          }
          pc.gbeg.push_back((int)idx.size()); pc.gcam.push_back(-1); pc.gslot.push_back(-1);  // sentinel
          pc.gend.push_back((int)pc.gbeg.size());
        }
      }
    });
    size_t total = 0;
    for (const Piece& pc : piece) total += pc.gbeg.size();
    lin_gbeg.reserve(total); lin_gcam.reserve(total); lin_gslot.reserve(total); lin_gptr.reserve((size_t)h->flat_chunks + 1);
    for (const Piece& pc : piece) {
      const int base = (int)lin_gbeg.size();
      lin_gbeg.insert(lin_gbeg.end(), pc.gbeg.begin(), pc.gbeg.end());
      lin_gcam.insert(lin_gcam.end(), pc.gcam.begin(), pc.gcam.end());
      lin_gslot.insert(lin_gslot.end(), pc.gslot.begin(), pc.gslot.end());
      for (int e : pc.gend) lin_gptr.push_back(base + e);
    }
  }
  if (h->lin_rows) {
    // inverted index camera -> rows of d.Urows (= entries of the group tables), ascending, cut into `rows_nslab` segments
    // per camera: a wave of the fold sums one segment (fold_u_rows), the gather sums the segments
    std::vector<int> per_cam(C + 1, 0);
    for (int cam : lin_gcam)
      if (cam >= 0) per_cam[cam + 1]++;
    int max_rows = 0;
    for (int c = 0; c < C; ++c) { max_rows = std::max(max_rows, per_cam[c + 1]); per_cam[c + 1] += per_cam[c]; }
    rows_nslab = std::max(1, std::min(64, (max_rows + 23) / 24));
    if (rows_nslab > 1) rows_nslab = (rows_nslab + 3) & ~3;
    rows_nslab = env_int("BA_NSLAB_U", rows_nslab);
    ucr_row.resize(per_cam[C]);
    {
      std::vector<int> fill(per_cam.begin(), per_cam.end() - 1);
      for (size_t t = 0; t < lin_gcam.size(); ++t)
        if (lin_gcam[t] >= 0) ucr_row[fill[lin_gcam[t]]++] = (int)t;
    }
    ucr_ptr.assign((size_t)C * rows_nslab + 1, 0);
    for (int c = 0; c < C; ++c) {
      const long cnt = per_cam[c + 1] - per_cam[c];
      for (int s2 = 0; s2 < rows_nslab; ++s2) ucr_ptr[(size_t)c * rows_nslab + s2] = per_cam[c] + (int)(cnt * s2 / rows_nslab);
    }
    ucr_ptr[(size_t)C * rows_nslab] = per_cam[C];
  }
  lap("linearize tables");
  // ---- k_schur_staged tables: chunks (consecutive landmarks, <= SCH_CAP observations, <= SCH_MAXLM
  // landmarks), the steps at which a wave must write its accumulators out (some lane's camera pair
  // changes), the contribution rows those write-outs fill, and the inverted index S block -> rows.
  std::vector<int> sch_lm(1, 0), sch_k(1, 0), sch_blk_row(1, 0), ctr_ptr(nu + 1, 0), ctr_row;
  std::vector<unsigned> sch_flush;
  long contrib_rows = 0;
  h->schur_cpb = 0;
  {
    int maxlen = 0;
    for (int j = 0; j < L_obs; ++j) maxlen = std::max(maxlen, lm_ptr[j + 1] - lm_ptr[j]);
    if (env_int("BA_SCHUR_STAGED", 1) != 0 && maxlen <= SCH_MAXLEN && L_obs > 0) {
      const int G = pow2_at_least(maxlen * (maxlen + 1) / 2, 8, 64), GPW = 64 / G;
      sch_lm.clear(); sch_k.clear(); sch_blk_row.clear();
      for (int j = 0; j < L_obs;) {
        sch_lm.push_back(j);
        sch_k.push_back(lm_ptr[j]);
        int j1 = j + 1;
        while (j1 < L_obs && j1 - j < SCH_MAXLM && lm_ptr[j1 + 1] - lm_ptr[j] <= SCH_CAP) ++j1;
        j = j1;
      }
      sch_lm.push_back(L_obs);
      sch_k.push_back(lm_ptr[L_obs]);
      const int n_chunks = (int)sch_lm.size() - 1;
      const int want_blocks = 256 * env_int("BA_SCHUR_BLOCKS_PER_CU", 8);
      const int cpb = std::min(SCH_MAXCPB, env_int("BA_CPB", std::max(1, (n_chunks + want_blocks - 1) / want_blocks)));
      const int n_blocks = (n_chunks + cpb - 1) / cpb;
      sch_flush.assign(n_chunks, 0u);
      // pair (a, b) of lane gl: gl = b (b + 1) / 2 + a
      int pa[64], pb[64];
      for (int b2 = 0, p = 0; p < 64; ++b2)
        for (int a2 = 0; a2 <= b2 && p < 64; ++a2, ++p) { pa[p] = a2; pb[p] = b2; }
      struct Cur { int fa, fb; bool sym; };
      auto slot_of = [&](int fa, int fb) {
        const int* lo = ucol.data() + urow_ptr[fa];
        const int* hi = ucol.data() + urow_ptr[fa + 1];
        return (int)(std::lower_bound(lo, hi, fb) - ucol.data());
      };
      // blocks are independent but for the number of their first contribution row: ranges of blocks on host threads,
      // rows counted from the start of the range, shifted once the ranges before it are known
      struct Piece { std::vector<std::pair<int, int>> contributors; std::vector<long> blk_row; long rows = 0; };
      const int T = hostpar::threads_for(n, 16384);
      std::vector<Piece> piece((size_t)T);
      hostpar::run(T, [&](int t, int nt) {
        Piece& pc = piece[(size_t)t];
        long rows = 0;
        for (int blk = (int)hostpar::part_begin(n_blocks, t, nt); blk < (int)hostpar::part_begin(n_blocks, t + 1, nt); ++blk) {
          Cur cur[64];
          for (auto& c : cur) c = Cur{-1, -1, false};
          pc.blk_row.push_back(rows);
          auto write_out = [&]() {
            for (int lane = 0; lane < 64; ++lane)
              if (cur[lane].fa >= 0) {
                pc.contributors.emplace_back(slot_of(cur[lane].fa, cur[lane].fb), (int)(rows + lane));
                cur[lane] = Cur{-1, -1, false};
              }
            rows += 64;
          };
          for (int ci = blk * cpb; ci < std::min(n_chunks, (blk + 1) * cpb); ++ci) {
            const int jc = sch_lm[ci], nc = sch_lm[ci + 1] - jc;
            for (int l0 = 0, step = 0; l0 < nc; l0 += GPW, ++step) {
              Cur nxt[64];
              bool need = false;
              for (int lane = 0; lane < 64; ++lane) {
                nxt[lane] = Cur{-1, -1, false};
                const int grp = lane / G, gl = lane % G, li = l0 + grp;
                if (li >= nc) continue;
                const int kb = lm_ptr[jc + li], len = lm_ptr[jc + li + 1] - kb;
                if (gl >= len * (len + 1) / 2) continue;
                const int fa = o_free[kb + pa[gl]], fb = o_free[kb + pb[gl]];
                if (fa < 0 || fb < 0) continue;
                nxt[lane] = Cur{fa, fb, fa == fb && pa[gl] != pb[gl]};
                if (cur[lane].fa >= 0 && (cur[lane].fa != fa || cur[lane].fb != fb || cur[lane].sym != nxt[lane].sym))
                  need = true;
              }
              if (need) { sch_flush[ci] |= 1u << step; write_out(); }
              for (int lane = 0; lane < 64; ++lane)
                if (nxt[lane].fa >= 0) cur[lane] = nxt[lane];
            }
          }
          write_out();  // the kernel always ends with one
          if (rows > ((long)1 << 31)) break;  // too many rows for the staged path (decided below): no point in going on
        }
        pc.rows = rows;
      });
      std::vector<std::pair<int, int>> contributors;  // (upper slot, contribution row)
      {
        std::vector<long> base((size_t)T + 1, 0);
        size_t n_contrib = 0;
        for (int t = 0; t < T; ++t) { base[(size_t)t + 1] = base[(size_t)t] + piece[(size_t)t].rows; n_contrib += piece[(size_t)t].contributors.size(); }
        contrib_rows = base[(size_t)T];
        if (contrib_rows <= ((long)1 << 31) - 128) {
          contributors.resize(n_contrib);
          std::vector<size_t> cbase((size_t)T + 1, 0);
          for (int t = 0; t < T; ++t) cbase[(size_t)t + 1] = cbase[(size_t)t] + piece[(size_t)t].contributors.size();
          sch_blk_row.reserve((size_t)n_blocks);
          for (int t = 0; t < T; ++t)
            for (long r : piece[(size_t)t].blk_row) sch_blk_row.push_back((int)(base[(size_t)t] + r));
          hostpar::run(T, [&](int t, int) {
            const Piece& pc = piece[(size_t)t];
            const int shift = (int)base[(size_t)t];
            for (size_t i = 0; i < pc.contributors.size(); ++i)
              contributors[cbase[(size_t)t] + i] = {pc.contributors[i].first, pc.contributors[i].second + shift};
          });
        }
      }
      const long max_rows = ((long)env_int("BA_CONTRIB_MAX_MB", 16384) << 20) / (CTR_ROW * (long)sizeof(double));
      if (contrib_rows <= std::min<long>(max_rows, ((long)1 << 31) - 128)) {
        h->schur_cpb = cpb;
        h->contrib_rows = contrib_rows;
        h->gather_wave = env_int("BA_GATHER_WAVE", (long)contributors.size() < 64L * std::max(nu, 1) ? 1 : 0) != 0;
        h->g_schur = G;
        h->d.n_chunks = n_chunks;
        for (auto& c : contributors) ctr_ptr[c.first + 1]++;
        for (int s2 = 0; s2 < nu; ++s2) ctr_ptr[s2 + 1] += ctr_ptr[s2];
        ctr_row.resize(contributors.size());
        std::vector<int> fill(ctr_ptr.begin(), ctr_ptr.end() - 1);
        for (auto& c : contributors) ctr_row[fill[c.first]++] = c.second;  // rows were generated in ascending order
      } else {
        contrib_rows = 0;  // too many distinct camera pairs per landmark range: atomics path
        sch_lm.assign(1, 0); sch_k.assign(1, 0); sch_blk_row.assign(1, 0); sch_flush.clear();
      }
    }
  }
  lap("schur tables");
  // W is not stored when both of its readers can rebuild it (the staged Schur kernel and the flat back-substitution)
  // and the problem is large enough for the traffic to matter: measured per LM iteration, stored / rebuilt: config 2
  // (0.2 M observations) 135 / 137 us, config 3 (1 M) 395 / 398 us (round 3: 270 / 266 us; 1024 windows of 1000
  // observations with tracks of ten: 1.75 / 1.89 ms per solve: the switch stays at 2 M),
  // config 4 (10 M) 1.89 / 1.76 ms (round 3: 1.46 / 1.28 ms) (the linearisation
  // loses its 96 B per observation of stores, the back-substitution its loads; the Schur kernel, which is not bound by
  // memory, pays 8-14 % for the arithmetic).  Rebuilding also frees 96 B per observation (0.96 GB at config 4).
  h->matrix_free = h->schur_cpb > 0 && h->flat_chunks > 0 && h->use_flat_backsub &&
                   env_int("BA_MATRIX_FREE", n >= 2000000 ? 1 : 0) != 0;
  // privatised slabs: as many copies as fit a fixed budget, so that few atomics share an address
  {
    const size_t u_bytes = (size_t)C * 27 * sizeof(double);
    const size_t s_bytes = ((size_t)nu * 36 + (size_t)6 * Cf) * sizeof(double);
    h->d.nslab_u = env_int("BA_NSLAB_U", (int)std::min<size_t>(32, std::max<size_t>(1, ((size_t)8 << 20) / u_bytes)));
    h->d.nslab_s = env_int("BA_NSLAB_S", (int)std::min<size_t>(16, std::max<size_t>(1, ((size_t)32 << 20) / s_bytes)));
    if (h->schur_cpb > 0) h->d.nslab_s = 0;  // the staged path writes contribution rows instead
    if (h->lin_rows) h->d.nslab_u = rows_nslab;  // the "slabs" are the segment sums of fold_u_rows
    h->d.s_slab_stride = (long)nu * 36 + 6L * Cf;
  }
  // one-workgroup reduced solve for small camera systems (S in registers, vectors + preconditioner +
  // SpMV slots in LDS)
  std::vector<int> sm_gidx(nb);
  {
    std::vector<int> inv_sm(nu), up_of_low(nl, -1);
    for (int t = 0; t < nu; ++t) inv_sm[sm_slot[t]] = t;
    for (int s2 = 0; s2 < nu; ++s2)
      if (low_of[s2] >= 0) up_of_low[low_of[s2] - nu - gap_blocks] = s2;
    for (int e = 0; e < nb; ++e) {
      const int sl = bslot[e];
      sm_gidx[e] = sl < nu ? inv_sm[sl] * SMALL_CSTRIDE : inv_sm[up_of_low[sl - nu - gap_blocks]] * SMALL_CSTRIDE + 6;
    }
    h->small_lds = ((size_t)2 * SMALL_ZSTRIDE * Cf + 16 + 2 * SMALL_COARSE_MAX + (size_t)(nu + 1) * SMALL_CSTRIDE) * sizeof(double) +
                   (size_t)nb * sizeof(int);
    h->small_solve = env_int("BA_SMALL_SOLVE", 1) && 6 * Cf <= SMALL_MAX_ROWS &&
                     nu <= SMALL_MAX_NB * SMALL_THREADS && h->small_lds <= 160 * 1024;
    if (h->small_solve && h->small_lds > 64 * 1024) {
      HIP_TRY(h, hipFuncSetAttribute((const void*)k_solve_small<1, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->small_lds));
      HIP_TRY(h, hipFuncSetAttribute((const void*)k_solve_small<1, true>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->small_lds));
      HIP_TRY(h, hipFuncSetAttribute((const void*)k_solve_small<2, false>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->small_lds));
    }
    // coarse space (ba_coarse.h): opt-in (BA_PCG_COARSE=1).  At config 2 it takes the PCG iterations of a solve from 110
    // to 66 and k_solve_small from 40 to 34 us, but an iteration with it costs 2.7 us against 1.5 (two more barriers and
    // the coarse product), and the LM iteration as a whole is no faster (132 against 129 us): DESIGN.md section 5
    h->coarse = h->small_solve && nu <= SMALL_THREADS && h->world == 1 && h->prior.cams.empty() && h->schur_cpb > 0 &&
                Cf <= 100 && env_int("BA_PCG_COARSE", 0) != 0;
    h->d.c_ag = h->d.c_nc = 0;
    if (h->coarse) {
      const int ag = Cf <= 50 ? 5 : 10, nc = 6 * ((Cf + ag - 1) / ag);  // nc <= 60
      h->d.c_ag = ag;
      h->d.c_nc = nc;
      h->coarse_lds = (size_t)2 * nc * (nc + 1) * sizeof(double);
      int rc2;
      if ((rc2 = dev_alloc(h, &h->csum[0], (size_t)2 * nc * nc)) || (rc2 = dev_alloc(h, &h->cinv[0], (size_t)nc * nc + 1)) ||
          (rc2 = dev_alloc(h, &h->cinv[1], (size_t)nc * nc + 1)))
        return rc2;
      h->csum[1] = h->csum[0] + (size_t)nc * nc;
      if (h->coarse_lds > 64 * 1024)
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_coarse_build, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->coarse_lds));
      if (!h->side) {
        HIP_TRY(h, hipStreamCreateWithFlags(&h->side, hipStreamNonBlocking));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_coarse[0], hipEventDisableTiming));
        HIP_TRY(h, hipEventCreateWithFlags(&h->ev_coarse[1], hipEventDisableTiming));
      }
    }
  }

  // connected components of the camera graph -> dense Cholesky per component when all are small
  std::vector<int> comp_ptr(1, 0), comp_cams(Cf), comp_local(Cf), comp_slot_ptr(1, 0), comp_slots(nu);
  std::vector<int> free_comp;  // component of every free camera
  {
    std::vector<int> parent(Cf);
    std::iota(parent.begin(), parent.end(), 0);
    auto find = [&](int a) { while (parent[a] != a) { parent[a] = parent[parent[a]]; a = parent[a]; } return a; };
    for (int s2 = 0; s2 < nu; ++s2) {
      const int a = find(urow[s2]), b = find(ucol[s2]);
      if (a != b) parent[std::max(a, b)] = std::min(a, b);
    }
    std::vector<int> comp_id(Cf, -1), count;
    for (int f = 0; f < Cf; ++f) {  // roots are the smallest camera of a component: ids ascend with the cameras
      const int r = find(f);
      if (comp_id[r] < 0) { comp_id[r] = (int)count.size(); count.push_back(0); }
      comp_id[f] = comp_id[r];
      comp_local[f] = count[comp_id[f]]++;
    }
    const int n_comp = (int)count.size();
    int max_cams = 0;
    for (int c = 0; c < n_comp; ++c) { comp_ptr.push_back(comp_ptr.back() + count[c]); max_cams = std::max(max_cams, count[c]); }
    for (int f = 0; f < Cf; ++f) comp_cams[comp_ptr[comp_id[f]] + comp_local[f]] = f;
    std::vector<int> scount(n_comp, 0);
    for (int s2 = 0; s2 < nu; ++s2) scount[comp_id[urow[s2]]]++;
    for (int c = 0; c < n_comp; ++c) comp_slot_ptr.push_back(comp_slot_ptr.back() + scount[c]);
    std::vector<int> fill(comp_slot_ptr.begin(), comp_slot_ptr.end() - 1);
    for (int s2 = 0; s2 < nu; ++s2) comp_slots[fill[comp_id[urow[s2]]]++] = s2;
    h->d.n_comp = n_comp;
    free_comp = comp_id;
    const int n = 6 * max_cams;
    h->comp_lds = comp_lds_bytes(n);
    const int want = h->opt.linear_solver;
    // AUTO: several small components (batched windows) -> Cholesky per component; one small system -> the
    // one-workgroup PCG; up to DENSE_MAX_CAMS cameras -> whole-GPU dense Cholesky; beyond -> multi-kernel PCG.
    // DENSE_CHOLESKY: per component when they are small, else whole-GPU.  PCG: never a direct solver.
    constexpr int DENSE_MAX_CAMS = 2048;
    const bool comps_small = Cf > 0 && max_cams <= COMP_MAX_CAMS && env_int("BA_COMP_SOLVE", 1);
    h->comp_solve = comps_small && (want == BA_SOLVER_DENSE_CHOLESKY || (want == BA_SOLVER_AUTO && n_comp > 1));
    // independent components get independent LM sequences when every kernel of the iteration has its per-component
    // form (staged Schur kernel, flat back-substitution, Cholesky per component), on one GPU, without a pose prior
    h->comp_lm = h->comp_solve && n_comp > 1 && h->schur_cpb > 0 && h->flat_chunks > 0 && h->use_flat_backsub &&
                 h->world == 1 && h->prior.cams.empty() && env_int("BA_COMP_LM", 1) != 0;
    if (want == BA_SOLVER_PCG || h->comp_solve) h->dense_solve = false;
    else if (want == BA_SOLVER_DENSE_CHOLESKY) h->dense_solve = true;
    else h->dense_solve = !h->small_solve && env_int("BA_DENSE_SOLVE", 1);
    if (h->dense_solve && Cf > DENSE_MAX_CAMS) {
      if (want == BA_SOLVER_DENSE_CHOLESKY)
        return fail(h, BA_ERR_INVALID, "dense Cholesky is limited to %d free cameras (here: %d)", DENSE_MAX_CAMS, Cf);
      h->dense_solve = false;
    }
    // banded + arrow shape of the pattern in the camera order: for every arrow size na, the half-bandwidth of the
    // blocks that lie outside the last na rows; the cheapest window wins
    h->banded_solve = false;
    h->bseg_P = 0;
    h->sharded = false;
    // best (bw, na) for n cameras whose upper-triangle blocks are `pairs` (r <= c); false if nothing fits the LDS
    auto choose_shape = [](int n, const std::vector<std::pair<int, int>>& pairs, int* bw_out, int* na_out) {
      std::vector<int> reach(n, 0);  // reach[c] = max (c - r) over the blocks of column c, then prefix maximum
      for (const auto& rc : pairs) reach[rc.second] = std::max(reach[rc.second], rc.second - rc.first);
      for (int c = 1; c < n; ++c) reach[c] = std::max(reach[c], reach[c - 1]);
      int best_na = -1, best_bw = 0;
      long best_cost = 0;
      for (int na = 0; na <= std::min(16, n - 1); ++na) {
        const int bw = std::min(reach[n - na - 1], n - na - 1);
        const long width = bw + 1 + na, nrow = width + 1;
        if (nrow > BND_MAX_NROW || width * nrow > BND_MAX_LDS_BLOCKS) continue;
        const long cost = width * width;
        if (best_na < 0 || cost < best_cost) { best_na = na; best_bw = bw; best_cost = cost; }
      }
      *bw_out = best_bw; *na_out = best_na;
      return best_na >= 0;
    };
    if (Cf > 0 && !h->comp_solve && env_int("BA_BANDED_SOLVE", 1)) {
      std::vector<std::pair<int, int>> pairs(nu);
      for (int s2 = 0; s2 < nu; ++s2) pairs[s2] = {urow[s2], ucol[s2]};
      int best_bw = 0, best_na = -1;
      const bool feasible = choose_shape(Cf, pairs, &best_bw, &best_na);
      if (want == BA_SOLVER_BANDED_CHOLESKY) {
        if (!feasible) return fail(h, BA_ERR_INVALID, "the reduced system is not banded enough for the banded Cholesky solver");
        h->banded_solve = true;
      } else if (want == BA_SOLVER_AUTO && feasible && !h->small_solve) {
        h->banded_solve = 4L * (best_bw + 1 + best_na) <= Cf;  // else the system is as good as dense
      }
      if (h->banded_solve) {
        h->bshape = BandedShape{Cf, Cf - best_na, best_bw, best_na, best_bw + 1 + best_na + 1};
        h->dense_solve = false;
        h->small_solve = false;
        // ---- two-level plan (ba_banded.h): ring bandwidth, P segments, where every block goes
        int bwr = 0;
        for (int s2 = 0; s2 < nu; ++s2) bwr = std::max(bwr, std::min(ucol[s2] - urow[s2], Cf - (ucol[s2] - urow[s2])));
        int P = env_int("BA_BSEG_P", (int)std::lround(std::sqrt((double)Cf / (2.0 * std::max(bwr, 1)))));
        P = std::min(P & ~1, 256);
        // Several ranks with BA_SHARDED_SOLVE=1 (opt-in: DESIGN.md section 7): the segments are dealt out to the ranks in
        // contiguous runs, so their number is a multiple of the ranks (and even: the parity rule of k_bseg_corner_add)
        const bool want_sharded = h->world > 1 && env_int("BA_SHARDED_SOLVE", 0) != 0;
        if (want_sharded) {
          int per = std::max(1, (P + h->world / 2) / h->world);
          if ((per * h->world) & 1) ++per;
          P = per * h->world;
        }
        bool two = env_int("BA_BSEG", 1) && bwr >= 1 && P >= 4 && Cf / P >= 3 * bwr + 1 && 3 * bwr + 2 <= BND_MAX_NROW &&
                   (long)(3 * bwr + 1) * (3 * bwr + 2) <= BND_MAX_LDS_BLOCKS;
        if (two) {
          const int bw = bwr, na_s = 2 * bw, nrow_s = 3 * bw + 2, npair = na_s * (na_s + 1) / 2;
          const long cs_s = (long)nrow_s * 36;
          std::vector<int> sp(P + 1);
          for (int q = 0; q <= P; ++q) sp[q] = (int)((long)q * Cf / P);
          std::vector<int> seg_of(Cf), loc(Cf);  // separator camera: seg_of = -1 - p, loc = t; interior: seg_of = p, loc = column
          h->h_segs.assign(P, BandedSeg{});
          long off = 0;
          for (int q = 0; q < P; ++q) {
            const int n_int = sp[q + 1] - sp[q] - bw;
            h->h_segs[q] = BandedSeg{off, n_int, sp[q] + bw};
            off += (long)(n_int + na_s) * cs_s;
            for (int f = sp[q]; f < sp[q + 1]; ++f) {
              if (f < sp[q] + bw) { seg_of[f] = -1 - q; loc[f] = f - sp[q]; }
              else { seg_of[f] = q; loc[f] = f - sp[q] - bw; }
            }
          }
          const long base2 = off;
          // level-2 pattern: every pair inside sep p U sep p + 1
          const int n2 = P * bw;
          std::vector<std::pair<int, int>> pairs2;
          for (int q = 0; q < P; ++q) {
            std::vector<int> cams;
            for (int t = 0; t < bw; ++t) cams.push_back(q * bw + t);
            for (int t = 0; t < bw; ++t) cams.push_back(((q + 1) % P) * bw + t);
            for (int a : cams) for (int b2 : cams) if (a <= b2) pairs2.push_back({a, b2});
          }
          int bw2 = 0, na2 = -1;
          two = choose_shape(n2, pairs2, &bw2, &na2);
          if (two) {
            const BandedShape sh2{n2, n2 - na2, bw2, na2, bw2 + 1 + na2 + 1};
            const long cs2 = (long)sh2.nrow * 36;
            auto l2_of = [&](int f) { return (-1 - seg_of[f]) * bw + loc[f]; };
            auto dst2 = [&](int i2, int k2) {  // block (i2, k2), i2 >= k2, of the level-2 system
              const int slot = i2 == k2 ? bnd_diag_slot(sh2, k2) : bnd_slot(sh2, i2, k2);
              if (k2 < sh2.Cb && i2 < sh2.Cb && i2 - k2 > sh2.bw) two = false;
              return base2 + k2 * cs2 + (long)slot * 36;
            };
            h->h_dst.assign(nu, 0); h->h_tr.assign(nu, 0); h->h_rdst.assign(Cf, 0);
            for (int s2 = 0; s2 < nu && two; ++s2) {
              const int r = urow[s2], c = ucol[s2];
              const bool ri = seg_of[r] >= 0, ci = seg_of[c] >= 0;
              if (ri && ci) {
                if (seg_of[r] != seg_of[c] || c - r > bw) { two = false; break; }
                h->h_dst[s2] = h->h_segs[seg_of[r]].base + loc[r] * cs_s + (long)(c - r) * 36;
                h->h_tr[s2] = 1;
              } else if (ri) {  // c is a camera of the right separator of r's segment
                const int q = seg_of[r], pc = -1 - seg_of[c];
                if (pc != (q + 1) % P) { two = false; break; }
                h->h_dst[s2] = h->h_segs[q].base + loc[r] * cs_s + (long)(bw + 1 + bw + loc[c]) * 36;
                h->h_tr[s2] = 1;
              } else if (ci) {  // r is a separator camera: the left one of c's segment, or (wrap) the right one
                const int q = seg_of[c], pr = -1 - seg_of[r];
                int a;
                if (pr == q) a = loc[r];
                else if (pr == (q + 1) % P) a = bw + loc[r];
                else { two = false; break; }
                h->h_dst[s2] = h->h_segs[q].base + loc[c] * cs_s + (long)(bw + 1 + a) * 36;
                h->h_tr[s2] = 0;
              } else {
                h->h_dst[s2] = dst2(l2_of(c), l2_of(r));
                h->h_tr[s2] = 1;
              }
            }
            for (int f = 0; f < Cf && two; ++f)
              h->h_rdst[f] = seg_of[f] >= 0 ? h->h_segs[seg_of[f]].base + loc[f] * cs_s + (long)(nrow_s - 1) * 36
                                            : base2 + l2_of(f) * cs2 + (long)(sh2.nrow - 1) * 36;
            h->h_l2.assign((size_t)P * na_s, 0);
            h->h_cdst.assign((size_t)P * npair, 0); h->h_ctr.assign((size_t)P * npair, 0); h->h_crdst.assign((size_t)P * na_s, 0);
            for (int q = 0; q < P && two; ++q) {
              for (int a = 0; a < na_s; ++a) {
                h->h_l2[(size_t)q * na_s + a] = a < bw ? q * bw + a : ((q + 1) % P) * bw + (a - bw);
                h->h_crdst[(size_t)q * na_s + a] = base2 + h->h_l2[(size_t)q * na_s + a] * cs2 + (long)(sh2.nrow - 1) * 36;
              }
              for (int a = 0; a < na_s; ++a)
                for (int a2 = 0; a2 <= a; ++a2) {
                  const int i2 = h->h_l2[(size_t)q * na_s + a], k2 = h->h_l2[(size_t)q * na_s + a2];
                  const size_t qi = (size_t)q * npair + (size_t)a * (a + 1) / 2 + a2;
                  if (i2 >= k2) { h->h_cdst[qi] = dst2(i2, k2); h->h_ctr[qi] = 0; }
                  else { h->h_cdst[qi] = dst2(k2, i2); h->h_ctr[qi] = 1; }
                }
            }
            if (two) {
              h->bseg_P = P; h->bseg_bw = bw; h->bseg_base2 = base2; h->bseg_total = base2 + (long)n2 * cs2;
              if (want_sharded && P % h->world == 0 && P <= 256) {
                const int per = P / h->world;
                h->sharded = true;
                h->sh_q0 = h->rank * per;
                h->sh_q1 = h->sh_q0 + per;
                h->h_cam_owner.assign(Cf, 0);
                for (int f = 0; f < Cf; ++f) h->h_cam_owner[f] = (seg_of[f] >= 0 ? seg_of[f] : -1 - seg_of[f]) / per;
              }
              h->bseg_lds = (size_t)(bw + 1 + na_s) * nrow_s * 36 * sizeof(double) + 1024;
              h->bshape = sh2;
              HIP_TRY(h, hipFuncSetAttribute((const void*)k_bseg_factor, hipFuncAttributeMaxDynamicSharedMemorySize,
                                             (int)h->bseg_lds));
            }
          }
        }
        h->banded_lds = (size_t)(h->bshape.bw + 1 + h->bshape.na) * h->bshape.nrow * 36 * sizeof(double) + 1024;  // + pair table
        HIP_TRY(h, hipFuncSetAttribute((const void*)k_banded_factor, hipFuncAttributeMaxDynamicSharedMemorySize,
                                       (int)h->banded_lds));
      }
    } else if (want == BA_SOLVER_BANDED_CHOLESKY) {
      return fail(h, BA_ERR_INVALID, "the banded Cholesky solver needs one connected problem with free cameras");
    }
    // ---- block cyclic reduction (ba_bcr.h): super-blocks of >= bwr consecutive cameras make S block tridiagonal (cyclic)
    h->bcr_solve = false;
    if (Cf > 0 && !h->comp_solve && (want == BA_SOLVER_CYCLIC_REDUCTION || (want == BA_SOLVER_AUTO && h->banded_solve)) &&
        env_int("BA_BCR", 1) && !h->sharded) {
      int bwr = 0;
      for (int s2 = 0; s2 < nu; ++s2) bwr = std::max(bwr, std::min(ucol[s2] - urow[s2], Cf - (ucol[s2] - urow[s2])));
      // Two ways to cut the cameras into super-blocks, the first that keeps S block tridiagonal wins:
      //   0: blocks of exactly g = bwr cameras, the FIRST one short when g does not divide Cf (the smallest block size
      //      there is: m = 6 g; 999 free cameras with tracks of five: 250 blocks of m = 24 instead of 249 of m = 30, four
      //      column steps per factorisation instead of five and 36 % fewer entries per block).  With camera 0 fixed the
      //      loop closure reaches one camera less far than the band (camera 0 sits in the gap), which is what lets a
      //      short first block keep the closure between the first and the last block only;
      //   1: N = Cf / g blocks of nearly equal size (round 2's cut).
      const int g = std::max(bwr, 1);
      for (int layout = (Cf % g != 0 && env_int("BA_BCR_SHORT_FIRST", 1)) ? 0 : 1; layout < 2 && !h->bcr_solve; ++layout) {
      const int N = layout == 0 ? (Cf + g - 1) / g : Cf / g;
      bool ok = N >= 3;
      int maxsz = 0;
      std::vector<int> start(std::max(N, 1) + 1, 0), sb_of(Cf, 0);
      if (ok) {
        if (layout == 0) {
          const int c0 = Cf - (N - 1) * g;
          for (int i2 = 1; i2 <= N; ++i2) start[i2] = c0 + (i2 - 1) * g;
        } else {
          for (int i2 = 0; i2 <= N; ++i2) start[i2] = (int)((long)i2 * Cf / N);
        }
        for (int i2 = 0; i2 < N; ++i2) {
          maxsz = std::max(maxsz, start[i2 + 1] - start[i2]);
          for (int f = start[i2]; f < start[i2 + 1]; ++f) sb_of[f] = i2;
        }
        ok = 6 * maxsz <= BCR_MAX_M;
      }
      if (want == BA_SOLVER_AUTO && N < 8) ok = false;  // few blocks: the other solvers are as fast
      if (ok) {
        const int m = 6 * maxsz;
        const long mm = (long)m * m;
        BcrPlan pl{};
        pl.m = m; pl.N = N;
        long off = 0;
        int n_l = N, nl = 0;
        // level-0 arrays first and contiguous: one memset (in ba_set_problem) clears them
        std::vector<int> ns;  // block counts: elimination levels while more than the tail's share is left, then the tail
        const int n_tail = std::max(2, BCR_TAIL_ROWS / m);
        for (int t = N;; t = (t + 1) / 2) { ns.push_back(t); if (t <= n_tail) break; }
        ok = (int)ns.size() <= BCR_MAX_LEVELS;
        if (ok) {
          for (size_t l = 0; l < ns.size(); ++l) {
            BcrLevel& lv = pl.lv[l];
            lv.n = ns[l]; lv.n_elim = l + 1 < ns.size() ? ns[l] / 2 : 0;
            lv.D = off; off += lv.n * mm;
            lv.E = off; off += lv.n * mm;
            lv.r = off; off += (long)lv.n * m;
            if (l == 0) h->bcr_zero = (size_t)off;
          }
          for (size_t l = 0; l + 1 < ns.size(); ++l) {
            BcrLevel& lv = pl.lv[l];
            lv.LiT = off; off += lv.n_elim * mm;
            lv.Fa = off; off += lv.n_elim * mm;
            lv.Fb = off; off += lv.n_elim * mm;
            lv.y = off; off += (long)lv.n_elim * m;
          }
          pl.X = off; off += (long)N * m;
          pl.n_levels = (int)ns.size() - 1;
          (void)n_l; (void)nl;
          // where every upper block of S goes
          h->h_bdst.assign(nu, 0); h->h_bdst2.assign(nu, 0); h->h_bkind.assign(nu, 0); h->h_brdst.assign(Cf, 0);
          for (int s2 = 0; s2 < nu && ok; ++s2) {
            const int r = urow[s2], c = ucol[s2], R = sb_of[r], Cc = sb_of[c];
            const long lr = 6L * (r - start[R]), lc = 6L * (c - start[Cc]);
            if (r == c) { h->h_bkind[s2] = 0; h->h_bdst[s2] = pl.lv[0].D + R * mm + lr * m + lr; }
            else if (R == Cc) {
              h->h_bkind[s2] = 1;
              h->h_bdst[s2] = pl.lv[0].D + R * mm + lr * m + lc;
              h->h_bdst2[s2] = pl.lv[0].D + R * mm + lc * m + lr;
            } else if (Cc == R + 1) { h->h_bkind[s2] = 2; h->h_bdst[s2] = pl.lv[0].E + R * mm + lr * m + lc; }
            else if (R == 0 && Cc == N - 1) { h->h_bkind[s2] = 3; h->h_bdst[s2] = pl.lv[0].E + (long)(N - 1) * mm + lc * m + lr; }
            else ok = false;  // a coupling beyond the neighbouring super-block
          }
        }
        if (ok) {
          for (int f = 0; f < Cf; ++f) h->h_brdst[f] = pl.lv[0].r + (long)sb_of[f] * m + 6L * (f - start[sb_of[f]]);
          h->h_bpad.clear(); h->h_bcam0.assign(N, 0); h->h_bcnt.assign(N, 0);
          for (int i2 = 0; i2 < N; ++i2) {
            h->h_bcam0[i2] = start[i2]; h->h_bcnt[i2] = start[i2 + 1] - start[i2];
            for (int e = 6 * h->h_bcnt[i2]; e < m; ++e) h->h_bpad.push_back(pl.lv[0].D + i2 * mm + (long)e * m + e);
          }
          h->bcr_npad = (int)h->h_bpad.size();
          h->bcr_plan = pl;
          h->bcr_solve = true;
          h->bcr_lds_factor = ((size_t)(4 * m + 1) * (m + 1) + BCR_BCAST) * sizeof(double);
          h->bcr_lds_update = ((size_t)4 * m * (m + 1) + 2 * m) * sizeof(double);
          HIP_TRY(h, hipFuncSetAttribute((const void*)k_bcr_factor, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bcr_lds_factor));
          HIP_TRY(h, hipFuncSetAttribute((const void*)k_bcr_update, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bcr_lds_update));
          h->bcr_lds_level = ((size_t)(4 * m + 4 * m + 1) * (m + 1) + 2 * m + BCR_BCAST + (size_t)m * (m + 1)) * sizeof(double);
          h->bcr_fused = env_int("BA_BCR_FUSED", 1) != 0 && h->bcr_lds_level <= 160 * 1024;
          if (h->bcr_fused)
            HIP_TRY(h, hipFuncSetAttribute((const void*)k_bcr_level, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bcr_lds_level));
          {
            // the top back-substitution levels go into the tail's launch while they have at most 8 blocks each
            BcrTop top{};
            if (bcr_wave_backsolve_ok(m) && env_int("BA_BCR_TOP", 1) != 0)
              for (int l = pl.n_levels - 1; l >= 0 && top.n < BCR_TOP_MAX && pl.lv[l].n_elim <= 8 && pl.lv[l].n <= 32; --l) {
                top.level[top.n] = l;
                top.lv[top.n] = pl.lv[l];
                top.xshift = l;
                top.nx = pl.lv[l].n;
                ++top.n;
              }
            h->bcr_top = top;
            const size_t nd = (size_t)pl.lv[pl.n_levels].n * m;
            h->bcr_lds_tail = ((nd + 1) * (nd + 1) + (nd / 6) * 36 + nd + BCR_BCAST + 2 + (size_t)top.nx * m) * sizeof(double);
            if (pl.lv[pl.n_levels].n == 2 && m == 24)  // the blocked tail: T1 | T2 | P2 | x | broadcast buffers | x of the folded levels
              h->bcr_lds_tail = ((size_t)(2 * (4 * m + 1) + m) * (m + 1) + nd + BCR_BCAST + 2 + (size_t)top.nx * m) * sizeof(double);
            HIP_TRY(h, hipFuncSetAttribute((const void*)k_bcr_tail, hipFuncAttributeMaxDynamicSharedMemorySize, (int)h->bcr_lds_tail));
          }
          h->banded_solve = false; h->bseg_P = 0; h->dense_solve = false; h->small_solve = false;
        }
      }
      }  // layouts
      if (!h->bcr_solve && want == BA_SOLVER_CYCLIC_REDUCTION)
        return fail(h, BA_ERR_INVALID, "cyclic reduction needs a trajectory: cameras that share landmarks only with their "
                                       "neighbours (at most %d apart), at least 3 groups of them", BCR_MAX_M / 6 - 1);
    } else if (want == BA_SOLVER_CYCLIC_REDUCTION) {
      return fail(h, BA_ERR_INVALID, "cyclic reduction needs one connected problem with free cameras");
    }
    if (h->comp_solve || h->dense_solve) h->small_solve = false;
    if (!h->small_solve) h->coarse = false;
    h->dense_nt = (6 * Cf + DT - 1) / DT;
    if (h->comp_solve && h->comp_lds > 32 * 1024)
      HIP_TRY(h, hipFuncSetAttribute((const void*)k_solve_components, hipFuncAttributeMaxDynamicSharedMemorySize,
                                     (int)h->comp_lds));
  }

  lap("solver plans");
  // ---- upload structure, allocate state
  std::vector<int> fixed_cams;
  for (int c = 0; c < C; ++c)
    if (h->cam2free[c] < 0) fixed_cams.push_back(c);
  std::vector<int> sm_desc((size_t)4 * std::max(nu, 1), 0);
  for (int t = 0; t < nu; ++t) {
    const int s2 = sm_slot[t];
    sm_desc[4 * (size_t)t] = s2; sm_desc[4 * (size_t)t + 1] = urow[s2]; sm_desc[4 * (size_t)t + 2] = ucol[s2];
    sm_desc[4 * (size_t)t + 3] = h->free2cam[urow[s2]] * 27;
  }
  DeviceProblem& d = h->d;
  const int nslab_u = d.nslab_u, nslab_s = d.nslab_s;
  d.C = C; d.Cf = Cf; d.L = L; d.L_obs = L_obs; d.n = n;
  d.K = Intrinsics{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6],
                   (intr[4] != 0.0 || intr[5] != 0.0 || intr[6] != 0.0) ? 1 : 0};
  d.loss = RobustLoss{h->opt.loss, h->opt.f_scale * h->opt.f_scale, 1.0 / (h->opt.f_scale * h->opt.f_scale)};
  d.nu = nu; d.nl = nl; d.nb = nb;
  d.n_lin_chunks = h->flat_chunks;
  d.lin_flat = h->lin_cpw > 0 ? 1 : 0;
  if (h->lin_cpw > 0) {
    const int lds = (int)linf_lds_bytes(C);
    HIP_TRY(h, hipFuncSetAttribute((const void*)k_linearize_flat<false>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    HIP_TRY(h, hipFuncSetAttribute((const void*)k_linearize_flat<true>, hipFuncAttributeMaxDynamicSharedMemorySize, lds));
  }
  int rc;
  h->pending.clear();
  queue_upload(h, &d.lm_ptr, lm_ptr); queue_upload(h, &d.cam2free, h->cam2free);
  queue_upload(h, &d.free2cam, h->free2cam); queue_upload(h, &d.fixed_cams, fixed_cams); queue_upload(h, &d.urow_ptr, urow_ptr);
  queue_upload(h, &d.ucol, ucol); queue_upload(h, &d.urow, urow);
  queue_upload(h, &d.diag_slot, diag_slot); queue_upload(h, &d.low_of, low_of);
  queue_upload(h, &d.brow_ptr, brow_ptr); queue_upload(h, &d.bcol, bcol);
  queue_upload(h, &d.bslot, bslot); queue_upload(h, &d.sm_slot, sm_slot);
  queue_upload(h, &d.sm_desc, sm_desc);
  queue_upload(h, &d.sm_gidx, sm_gidx);
  queue_upload(h, &d.comp_ptr, comp_ptr); queue_upload(h, &d.comp_cams, comp_cams);
  queue_upload(h, &d.comp_local, comp_local); queue_upload(h, &d.comp_slot_ptr, comp_slot_ptr);
  queue_upload(h, &d.comp_slots, comp_slots);
  queue_upload(h, &d.lin_lm, lin_lm); queue_upload(h, &d.lin_k, lin_k);
  queue_upload(h, &d.lin_perm, lin_perm); queue_upload(h, &d.lin_gptr, lin_gptr);
  queue_upload(h, &d.lin_gbeg, lin_gbeg); queue_upload(h, &d.lin_gcam, lin_gcam);
  queue_upload(h, &d.lin_gslot, lin_gslot); queue_upload(h, &d.lin_wincam, lin_wincam);
  queue_upload(h, &d.ucr_ptr, ucr_ptr); queue_upload(h, &d.ucr_row, ucr_row);
  queue_upload(h, &d.sch_lm, sch_lm); queue_upload(h, &d.sch_k, sch_k);
  queue_upload(h, &d.sch_flush, sch_flush); queue_upload(h, &d.sch_blk_row, sch_blk_row);
  queue_upload(h, &d.ctr_ptr, ctr_ptr); queue_upload(h, &d.ctr_row, ctr_row);
  queue_alloc(h, &d.contrib, (size_t)contrib_rows * CTR_ROW);
  d.W = nullptr;
  if (!h->matrix_free) queue_alloc(h, &d.W, (size_t)W_ROWS * n);
  d.n_urows = h->lin_rows ? (long)lin_gcam.size() : 0;
  d.n_bs_blocks = h->flat_chunks > 0 ? grid_for(h->flat_chunks, 4) : 0;
  d.n_pose_blocks = grid_for(C);
  queue_alloc(h, &d.Urows, (size_t)27 * d.n_urows); queue_alloc(h, &d.lin_part, (size_t)4 * h->flat_chunks);
  queue_alloc(h, &d.bs_part, (size_t)8 * d.n_bs_blocks); queue_alloc(h, &d.pose_part, (size_t)5 * d.n_pose_blocks);
  queue_alloc(h, &d.V, (size_t)6 * L);
  queue_alloc(h, &d.gp, (size_t)3 * L); queue_alloc(h, &d.zmin, (size_t)L);
  queue_alloc(h, &d.Ugc_part, (size_t)27 * C);
  queue_alloc(h, &d.S, (size_t)36 * (nu + gap_blocks + nl));
  queue_alloc(h, &h->scal_fold, (size_t)SC_COUNT + FOLD_MAX_RANKS);
  queue_alloc(h, &d.Ugc_slab, (size_t)nslab_u * 27 * C + (size_t)SC_COUNT * SCAL_SLABS + (size_t)nslab_s * d.s_slab_stride);
  queue_alloc(h, &d.ctrl, 1);
  queue_alloc(h, &d.Minv, (size_t)36 * Cf);
  queue_alloc(h, &d.dc, (size_t)6 * Cf); queue_alloc(h, &d.r, (size_t)6 * Cf);
  queue_alloc(h, &d.z, (size_t)6 * Cf); queue_alloc(h, &d.p, (size_t)6 * Cf);
  queue_alloc(h, &d.q, (size_t)6 * Cf);
  queue_alloc(h, &d.poses_buf[0], (size_t)7 * C); queue_alloc(h, &d.poses_buf[1], (size_t)7 * C);
  queue_alloc(h, &h->poses_init, (size_t)7 * C); queue_alloc(h, &d.pts_buf[0], (size_t)3 * L);
  queue_alloc(h, &d.pts_buf[1], (size_t)3 * L); queue_alloc(h, &h->pts_init, (size_t)3 * L);
  if ((rc = dev_commit(h))) return rc;
  if (early.th.joinable()) early.th.join();
  if (early.base) { h->allocs.push_back(early.base); early.kept = true; }
  HIP_TRY(h, early.err);
  lap("uploads + state allocation");
  h->banded_B = nullptr;
  if (h->banded_solve && h->bseg_P > 0) {
    if ((rc = dev_alloc(h, &h->banded_B, (size_t)h->bseg_total)) || (rc = dev_alloc(h, &h->bseg_x2, (size_t)6 * h->bshape.Cf)) ||
        (rc = dev_upload(h, &h->bseg_segs, h->h_segs)) || (rc = dev_upload(h, &h->bseg_dst, h->h_dst)) ||
        (rc = dev_upload(h, &h->bseg_tr, h->h_tr)) || (rc = dev_upload(h, &h->bseg_rdst, h->h_rdst)) ||
        (rc = dev_upload(h, &h->bseg_cdst, h->h_cdst)) || (rc = dev_upload(h, &h->bseg_ctr, h->h_ctr)) ||
        (rc = dev_upload(h, &h->bseg_crdst, h->h_crdst)) || (rc = dev_upload(h, &h->bseg_l2, h->h_l2)))
      return rc;
    h->cam_owner = nullptr;
    if (h->sharded && (rc = dev_upload(h, &h->cam_owner, h->h_cam_owner))) return rc;
  } else if (h->banded_solve && (rc = dev_alloc(h, &h->banded_B, (size_t)Cf * h->bshape.nrow * 36))) {
    return rc;
  }
  h->bcr_B = nullptr;
  if (h->bcr_solve) {
    const BcrPlan& pl = h->bcr_plan;
    const size_t total = (size_t)pl.X + (size_t)pl.N * pl.m;
    if ((rc = dev_alloc(h, &h->bcr_B, total)) || (rc = dev_upload(h, &h->bcr_dst, h->h_bdst)) ||
        (rc = dev_upload(h, &h->bcr_dst2, h->h_bdst2)) || (rc = dev_upload(h, &h->bcr_kind, h->h_bkind)) ||
        (rc = dev_upload(h, &h->bcr_rdst, h->h_brdst)) || (rc = dev_upload(h, &h->bcr_pad, h->h_bpad)) ||
        (rc = dev_upload(h, &h->bcr_cam0, h->h_bcam0)) || (rc = dev_upload(h, &h->bcr_cnt, h->h_bcnt)))
      return rc;
    // the level-0 arrays are cleared ONCE: k_bcr_assemble writes the same entries in every solve (the block pattern is
    // fixed) and nothing else writes there, so what it does not write stays zero
    HIP_TRY(h, hipMemset(h->bcr_B, 0, h->bcr_zero * sizeof(double)));
    // the padded diagonal entries are 1 in every solve: written here once, for the gather that fills the arrays directly
    if (!h->h_bpad.empty()) {
      std::vector<double> level0(h->bcr_zero, 0.0);
      for (long p : h->h_bpad) level0[(size_t)p] = 1.0;
      HIP_TRY(h, hipMemcpy(h->bcr_B, level0.data(), level0.size() * sizeof(double), hipMemcpyHostToDevice));
    }
    h->bcr_direct = h->world == 1 && h->prior.cams.empty() && env_int("BA_BCR_DIRECT", 1) != 0;
  }
  d.clm = CompLm{};
  if (h->comp_lm) {
    std::vector<int> lm_comp(L_obs, 0);
    for (int j = 0; j < L_obs; ++j)
      for (int k = lm_ptr[j]; k < lm_ptr[j + 1]; ++k)
        if (o_free[k] >= 0) { lm_comp[j] = free_comp[o_free[k]]; break; }
    const size_t nc = (size_t)d.n_comp;
    if ((rc = dev_alloc(h, &d.clm.lam, nc)) || (rc = dev_alloc(h, &d.clm.nu, nc)) || (rc = dev_alloc(h, &d.clm.cost, nc)) ||
        (rc = dev_alloc(h, &d.clm.nbehind, nc)) || (rc = dev_alloc(h, &d.clm.stat, nc * CL_STRIDE)) ||
        (rc = dev_alloc(h, &d.clm.accept, nc)) || (rc = dev_alloc(h, &d.clm.done, nc)) ||
        (rc = dev_alloc(h, &d.clm.lam_lm, (size_t)L_obs)) || (rc = dev_upload(h, &d.clm.lm_comp, lm_comp)) ||
        (rc = dev_upload(h, &d.clm.free_comp, free_comp)))
      return rc;
    // per chunk of the flat kernels: the distinct components of its landmarks, in order of first appearance; and per
    // component the entries that belong to it, ascending: the fixed order of the per-component sums
    std::vector<int> cp_ptr(1, 0), cp_comp, cc_ptr(nc + 1, 0), cc_ent;
    for (int ch = 0; ch < h->flat_chunks; ++ch) {
      const size_t first = cp_comp.size();
      for (int j = lin_lm[ch]; j < lin_lm[ch + 1]; ++j) {
        bool seen = false;
        for (size_t e = first; e < cp_comp.size() && !seen; ++e) seen = cp_comp[e] == lm_comp[j];
        if (!seen) cp_comp.push_back(lm_comp[j]);
      }
      cp_ptr.push_back((int)cp_comp.size());
    }
    for (int cm : cp_comp) cc_ptr[(size_t)cm + 1]++;
    for (size_t w = 0; w < nc; ++w) cc_ptr[w + 1] += cc_ptr[w];
    cc_ent.resize(cp_comp.size());
    {
      std::vector<int> fill(cc_ptr.begin(), cc_ptr.end() - 1);
      for (size_t e = 0; e < cp_comp.size(); ++e) cc_ent[(size_t)fill[(size_t)cp_comp[e]]++] = (int)e;
    }
    if ((rc = dev_upload(h, &d.clm.cp_ptr, cp_ptr)) || (rc = dev_upload(h, &d.clm.cp_comp, cp_comp)) ||
        (rc = dev_upload(h, &d.clm.cc_ptr, cc_ptr)) || (rc = dev_upload(h, &d.clm.cc_ent, cc_ent)) ||
        (rc = dev_alloc(h, &d.clm.cp_val, (size_t)3 * cp_comp.size())) || (rc = dev_alloc(h, &d.clm.pose_val, (size_t)Cf)))
      return rc;
    HIP_TRY(h, hipMemset(d.clm.cp_val, 0, std::max<size_t>(1, 3 * cp_comp.size()) * sizeof(double)));
    HIP_TRY(h, hipMemset(d.clm.pose_val, 0, (size_t)Cf * sizeof(double)));
  }
  d.prior = PriorDev{};
  if (!h->prior.cams.empty()) {
    const int np = (int)h->prior.cams.size();
    std::vector<int> fr(np), slot((size_t)np * np, -1);
    for (int a = 0; a < np; ++a) fr[a] = h->cam2free[h->prior.cams[a]];
    for (int a = 0; a < np; ++a)
      for (int b = 0; b < np; ++b)
        if (fr[a] <= fr[b] && (a == b || fr[a] != fr[b]))
          slot[(size_t)a * np + b] =
              (int)(std::lower_bound(keys.begin(), keys.end(), (int64_t)fr[a] * Cf + fr[b]) - keys.begin());
    if ((rc = dev_upload(h, &d.prior.cam, h->prior.cams)) || (rc = dev_upload(h, &d.prior.fr, fr)) ||
        (rc = dev_upload(h, &d.prior.slot, slot)) || (rc = dev_upload(h, &d.prior.ref, h->prior.ref)) ||
        (rc = dev_upload(h, &d.prior.H, h->prior.H)) || (rc = dev_upload(h, &d.prior.g, h->prior.g)))
      return rc;
    d.prior.n = np;
    d.prior.c0 = h->prior.c0;
  }
  h->dense_A = nullptr;
  if (h->dense_solve) {
    const size_t ld = (size_t)h->dense_nt * DT;
    if ((rc = dev_alloc(h, &h->dense_A, (size_t)(h->dense_nt + 1) * DT * ld))) return rc;
    HIP_TRY(h, hipFuncSetAttribute((const void*)k_dense_trsm, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((DT * DT + DT) * sizeof(double))));
    HIP_TRY(h, hipFuncSetAttribute((const void*)k_dense_backsolve, hipFuncAttributeMaxDynamicSharedMemorySize,
                                   (int)((DPACK + 2 * DT) * sizeof(double))));
  }
  lap("solver buffers");
  d.scal = d.Ugc_slab + (size_t)nslab_u * 27 * C;
  d.S_slab = d.scal + (size_t)SC_COUNT * SCAL_SLABS;
  HIP_TRY(h, hipMemset(d.ctrl, 0, sizeof(LmCtrl)));
  d.rhs = d.S + (size_t)36 * nu;
  d.Ugc = d.rhs + (size_t)6 * Cf;
  HIP_TRY(h, hipMemset(d.S, 0, (size_t)36 * (nu + gap_blocks + nl) * sizeof(double)));
  HIP_TRY(h, hipMemset(d.V, 0, (size_t)6 * L * sizeof(double)));
  HIP_TRY(h, hipMemset(d.gp, 0, (size_t)3 * L * sizeof(double)));
  HIP_TRY(h, hipMemset(d.dc, 0, (size_t)6 * Cf * sizeof(double)));
  h->stage_bytes = ((size_t)7 * C + (size_t)3 * L) * sizeof(double);
  HIP_TRY(h, hipHostMalloc((void**)&h->stage, h->stage_bytes));
  HIP_TRY(h, hipDeviceSynchronize());
  lap("memsets + stage + sync");
  h->has_problem = true;
  h->setup_ms = ms_since(t0);
  return BA_OK;
}
}  // namespace

extern "C" {

int ba_upload_guess(ba_handle* h, const double* poses, const double* points) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_problem) return fail(h, BA_ERR_STATE, "upload_guess before set_problem");  // (no problem: no collective either)
  if (!poses || !points) return agree_status(h, fail(h, BA_ERR_INVALID, "null poses/points"), "upload_guess");
  HIP_TRY(h, hipSetDevice(h->device));
  h->agree_failed = false;
  // multi-GPU: the points are rank-local; a non-finite guess on one rank must stop every rank before the solve
  int rc = agree_status(h, stage_state(h, poses, points, h->poses_init, h->pts_init), "upload_guess");
  if (rc) return rc;
  h->has_guess = true;
  h->has_estimate = false;
  return BA_OK;
}

int ba_solve_resident(ba_handle* h, ba_report* report) {
  if (!h) return BA_ERR_INVALID;
  h->agree_failed = false;
  int rc = agree_status(h, h->has_guess ? BA_OK : fail(h, BA_ERR_STATE, "solve before upload_guess"), "solve");
  if (rc) return rc;
  ba_report local;
  HIP_TRY(h, hipSetDevice(h->device));
  return lm_solve(h, report ? report : &local);
}

int ba_download_estimate(ba_handle* h, double* poses, double* points) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_estimate) return fail(h, BA_ERR_STATE, "download before a successful solve");
  if (!poses || !points) return fail(h, BA_ERR_INVALID, "null poses/points");
  HIP_TRY(h, hipSetDevice(h->device));
  return fetch_state(h, h->poses_cur, h->pts_cur, poses, points);
}

int ba_solve(ba_handle* h, double* poses, double* points, ba_report* report) {
  int rc = ba_upload_guess(h, poses, points);
  if (rc) return rc;
  if ((rc = ba_solve_resident(h, report))) return rc;
  return ba_download_estimate(h, poses, points);
}

int ba_solve_batched(ba_handle* h, int32_t B, const int32_t* cam_off, const int32_t* lm_off, const int32_t* obs_off,
                     const int32_t* obs_cam, const int32_t* obs_pt, const double* obs_uv, const double* intr,
                     double* poses, double* points, double* cost0, double* cost, ba_report* report) {
  if (!h) return BA_ERR_INVALID;
  if (!h->prior.cams.empty()) return fail(h, BA_ERR_STATE, "solve_batched: a pose prior is set (ba_set_pose_prior with n_prior = 0 removes it)");
  if (h->world > 1)  // windows are independent: ranks take disjoint sets of windows, each with a handle of its own
    return fail(h, BA_ERR_STATE, "solve_batched needs no communicator: use a handle without ba_comm_init per rank");
  if (B <= 0 || !cam_off || !lm_off || !obs_off || !obs_cam || !obs_pt || !obs_uv || !intr || !poses || !points)
    return fail(h, BA_ERR_INVALID, "solve_batched: empty batch or null pointer");
  if (cam_off[0] != 0 || lm_off[0] != 0 || obs_off[0] != 0)
    return fail(h, BA_ERR_INVALID, "solve_batched: offset arrays must start at 0");
  for (int b = 0; b < B; ++b)
    if (cam_off[b + 1] <= cam_off[b] || lm_off[b + 1] <= lm_off[b] || obs_off[b + 1] <= obs_off[b])
      return fail(h, BA_ERR_INVALID, "solve_batched: problem %d is empty or its offsets decrease", b);
  const int C = cam_off[B], L = lm_off[B], n = obs_off[B];
  // One block-diagonal problem: global indices, the first camera of every window fixed.  The windows
  // share no camera and no landmark, so the camera graph falls into >= B components and the reduced
  // system is solved by one dense Cholesky per component (k_solve_components).
  std::vector<int32_t> gcam(n), gpt(n);
  std::vector<uint8_t> fixed(C, 0);
  for (int b = 0; b < B; ++b) {
    fixed[cam_off[b]] = 1;
    const int nc = cam_off[b + 1] - cam_off[b], nl = lm_off[b + 1] - lm_off[b];
    for (int k = obs_off[b]; k < obs_off[b + 1]; ++k) {
      if (obs_cam[k] < 0 || obs_cam[k] >= nc || obs_pt[k] < 0 || obs_pt[k] >= nl)
        return fail(h, BA_ERR_INVALID, "solve_batched: observation %d of problem %d has an index out of range", k - obs_off[b], b);
      gcam[k] = cam_off[b] + obs_cam[k];
      gpt[k] = lm_off[b] + obs_pt[k];
    }
  }
  int rc = ba_set_problem(h, C, L, n, gcam.data(), gpt.data(), obs_uv, intr, fixed.data());
  if (rc) return rc;
  std::vector<double> res;
  auto per_problem = [&](double* out) -> int {
    res.resize((size_t)2 * n);
    int rc2 = ba_evaluate(h, poses, points, res.data(), nullptr, nullptr);
    if (rc2) return rc2;
    for (int b = 0; b < B; ++b) {
      double acc = 0.0;
      for (size_t k = (size_t)2 * obs_off[b]; k < (size_t)2 * obs_off[b + 1]; ++k) acc += res[k] * res[k];
      out[b] = 0.5 * acc;
    }
    return BA_OK;
  };
  if (cost0 && (rc = per_problem(cost0))) return rc;
  if ((rc = ba_solve(h, poses, points, report))) return rc;
  if (cost && (rc = per_problem(cost))) return rc;
  return BA_OK;
}

int ba_evaluate(ba_handle* h, const double* poses, const double* points, double* residuals, double* cost,
                int32_t* n_behind) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_problem) return fail(h, BA_ERR_STATE, "evaluate before set_problem");
  if (!poses || !points) return fail(h, BA_ERR_INVALID, "null poses/points");
  HIP_TRY(h, hipSetDevice(h->device));
  begin_launches(h);
  int rc = stage_state(h, poses, points, h->d.poses_buf[1], h->d.pts_buf[1]);
  if (rc) return rc;
  h->has_estimate = false;  // the trial buffers were clobbered
  double* d_res = nullptr;
  if (residuals) HIP_TRY(h, hipMalloc((void**)&d_res, (size_t)2 * h->n * sizeof(double)));
  HIP_TRY(h, hipMemsetAsync(h->d.scal, 0, SCAL_BYTES, h->stream));
  launch_cost(h, h->d.poses_buf[1], h->d.pts_buf[1], SC_COST_T, SC_NBEHIND_T, d_res);
  if (residuals)
    HIP_TRY(h, hipMemcpyAsync(residuals, d_res, (size_t)2 * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  rc = fetch_scalars(h);
  if (d_res) (void)hipFree(d_res);
  if (rc || (rc = check_launches(h, "evaluate"))) return rc;
  if (cost) *cost = h->scal_host[SC_COST_T];
  if (n_behind) *n_behind = (int32_t)h->scal_host[SC_NBEHIND_T];
  if (h->scal_host[SC_ZEROZ] > 0.0)
    return fail(h, BA_ERR_NUMERICAL, "cannot project points with zero Z coordinates (division by zero)");
  return BA_OK;
}

int ba_triangulate(ba_handle* h, const double* poses, int32_t refine_iters, double min_depth, double* points,
                   double* reproj_err, uint8_t* valid) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_problem) return fail(h, BA_ERR_STATE, "triangulate before set_problem");
  if (!poses || !points) return fail(h, BA_ERR_INVALID, "null poses/points");
  if (refine_iters < 0 || !std::isfinite(min_depth)) return fail(h, BA_ERR_INVALID, "refine_iters < 0 or min_depth not finite");
  HIP_TRY(h, hipSetDevice(h->device));
  begin_launches(h);
  // poses as in stage_state (every pose is data here: normalise all quaternions)
  double* sp = h->stage;
  for (int c = 0; c < h->C; ++c) {
    const double* p = poses + 7 * c;
    for (int i = 0; i < 7; ++i)
      if (!std::isfinite(p[i])) return fail(h, BA_ERR_INVALID, "pose %d has a non-finite entry", c);
    const double nq = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    if (nq == 0.0) return fail(h, BA_ERR_INVALID, "pose %d has a zero quaternion", c);
    for (int i = 0; i < 3; ++i) sp[7 * c + i] = p[i];
    for (int i = 3; i < 7; ++i) sp[7 * c + i] = p[i] / nq;
  }
  h->has_estimate = false;  // the trial buffers are used as scratch
  double* d_poses = h->d.poses_buf[1];
  double* d_pts = h->d.pts_buf[1];
  const int Lo = h->L_obs;
  double* d_err = nullptr;
  unsigned char* d_valid = nullptr;
  hipEvent_t ev_a = nullptr, ev_b = nullptr;
  HIP_TRY(h, hipMemcpyAsync(d_poses, sp, (size_t)h->C * 7 * sizeof(double), hipMemcpyHostToDevice, h->stream));
  if (Lo > 0) {
    HIP_TRY(h, hipMalloc((void**)&d_err, (size_t)Lo * sizeof(double)));
    if (hipMalloc((void**)&d_valid, (size_t)Lo) != hipSuccess) {
      (void)hipFree(d_err);
      return fail(h, BA_ERR_NOMEM, "triangulate: device allocation failed");
    }
    const int grid = grid_for(Lo, 64);
    const int force_jacobi = env_int("BA_TRI_JACOBI", 0);
    if (h->opt.profile_kernels) {  // outside a solve: the span is read right here (class "other")
      ev_a = next_event(h); ev_b = next_event(h);
      hipExtLaunchKernelGGL(k_triangulate, dim3(grid), dim3(64), 0, h->stream, ev_a, ev_b, 0, h->d, (const double*)d_poses,
                            (int)refine_iters, min_depth, force_jacobi, d_pts, d_err, d_valid);
    } else {
      hipLaunchKernelGGL(k_triangulate, dim3(grid), dim3(64), 0, h->stream, h->d, (const double*)d_poses,
                         (int)refine_iters, min_depth, force_jacobi, d_pts, d_err, d_valid);
    }
  }
  std::vector<double> err((size_t)Lo);
  std::vector<unsigned char> ok((size_t)Lo);
  double* sx = h->stage + (size_t)h->C * 7;
  hipError_t e = hipSuccess;
  if (Lo > 0) {
    e = hipMemcpyAsync(sx, d_pts, (size_t)Lo * 3 * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(err.data(), d_err, (size_t)Lo * sizeof(double), hipMemcpyDeviceToHost, h->stream);
    if (e == hipSuccess) e = hipMemcpyAsync(ok.data(), d_valid, (size_t)Lo, hipMemcpyDeviceToHost, h->stream);
  }
  if (e == hipSuccess) e = hipStreamSynchronize(h->stream);
  if (ev_a) {
    float ms = 0.f;
    if (e == hipSuccess && hipEventElapsedTime(&ms, ev_a, ev_b) == hipSuccess) { h->ktime_ms[KT_OTHER] += ms; h->kcount[KT_OTHER] += 1; }
    h->ev_used = 0;
  }
  if (d_err) (void)hipFree(d_err);
  if (d_valid) (void)hipFree(d_valid);
  if (e != hipSuccess) return fail(h, BA_ERR_DEVICE, "triangulate: %s", hipGetErrorString(e));
  note_launch(h);
  { const int rc_l = check_launches(h, "triangulate"); if (rc_l) return rc_l; }
  const double nan = std::numeric_limits<double>::quiet_NaN();
  if (reproj_err) for (int j = 0; j < h->L; ++j) reproj_err[j] = nan;
  if (valid) std::memset(valid, 0, (size_t)h->L);
  for (int j = 0; j < Lo; ++j) {
    const size_t u = (size_t)h->lm_internal_to_user[j];
    if (reproj_err) reproj_err[u] = err[j];
    if (valid) valid[u] = ok[j];
    if (ok[j]) std::memcpy(points + 3 * u, sx + 3 * (size_t)j, 3 * sizeof(double));  // others stay untouched
  }
  return BA_OK;
}

int ba_jacobian_blocks(ba_handle* h, const double* poses, const double* points, double* Jc, double* Jp) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_problem) return fail(h, BA_ERR_STATE, "jacobian_blocks before set_problem");
  if (!poses || !points || !Jc || !Jp) return fail(h, BA_ERR_INVALID, "null argument");
  HIP_TRY(h, hipSetDevice(h->device));
  begin_launches(h);
  int rc = stage_state(h, poses, points, h->d.poses_buf[1], h->d.pts_buf[1]);
  if (rc) return rc;
  h->has_estimate = false;
  DevTmp t_jc, t_jp;
  HIP_TRY(h, hipMalloc(&t_jc.p, (size_t)12 * h->n * sizeof(double)));
  HIP_TRY(h, hipMalloc(&t_jp.p, (size_t)6 * h->n * sizeof(double)));
  double *dJc = static_cast<double*>(t_jc.p), *dJp = static_cast<double*>(t_jp.p);
  k_export_jac<<<std::min(grid_for(h->n), 4096), 256, 0, h->stream>>>(h->d, h->d.poses_buf[1], h->d.pts_buf[1], dJc, dJp); note_launch(h);
  HIP_TRY(h, hipMemcpyAsync(Jc, dJc, (size_t)12 * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  HIP_TRY(h, hipMemcpyAsync(Jp, dJp, (size_t)6 * h->n * sizeof(double), hipMemcpyDeviceToHost, h->stream));
  const hipError_t e_sync = hipStreamSynchronize(h->stream);
  if (e_sync != hipSuccess) return fail(h, BA_ERR_DEVICE, "jacobian_blocks: %s", hipGetErrorString(e_sync));
  return check_launches(h, "jacobian_blocks");
}

int ba_debug_linearization(ba_handle* h, const double* poses, const double* points, double lambda, double* U,
                           double* V, double* W, double* gc, double* gp, double* S, double* rhs, double* dc,
                           double* dp) {
  if (!h) return BA_ERR_INVALID;
  if (!h->has_problem) return fail(h, BA_ERR_STATE, "debug_linearization before set_problem");
  if (!poses || !points) return fail(h, BA_ERR_INVALID, "null poses/points");
  HIP_TRY(h, hipSetDevice(h->device));
  begin_launches(h);
  double* P0 = h->d.poses_buf[0];
  double* X0 = h->d.pts_buf[0];
  double* P1 = h->d.poses_buf[1];
  double* X1 = h->d.pts_buf[1];
  int rc = stage_state(h, poses, points, P0, X0);
  if (rc) return rc;
  h->has_estimate = false;
  const DeviceProblem& d = h->d;
  const int C = h->C, Cf = h->Cf, L = h->L, n = h->n;
  if ((rc = clear_accumulators(h, true))) return rc;
  enqueue_linearize(h, P0, X0);
  int pcg_it = 0;
  if ((rc = do_reduced_solve(h, lambda, &pcg_it))) return rc;
  HIP_TRY(h, hipMemcpyAsync(X1, X0, (size_t)3 * L * sizeof(double), hipMemcpyDeviceToDevice, h->stream));
  enqueue_trial(h, lambda, P0, X0, P1, X1);
  if ((rc = fetch_scalars(h)) || (rc = check_launches(h, "debug_linearization"))) return rc;
  auto fetch = [&](std::vector<double>& dst, const double* src, size_t count) -> int {
    dst.resize(count);
    HIP_TRY(h, hipMemcpy(dst.data(), src, count * sizeof(double), hipMemcpyDeviceToHost));
    return BA_OK;
  };
  std::vector<double> t;
  if (U || gc) {
    if ((rc = fetch(t, d.Ugc, (size_t)27 * C))) return rc;
    for (int c = 0; c < C; ++c) {
      if (U)
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j)
            U[36 * (size_t)c + 6 * i + j] = t[27 * (size_t)c + (i <= j ? sym6_index(i, j) : sym6_index(j, i))];
      if (gc)
        for (int i = 0; i < 6; ++i) gc[6 * (size_t)c + i] = t[27 * (size_t)c + 21 + i];
    }
  }
  if (V) {
    if ((rc = fetch(t, d.V, (size_t)6 * L))) return rc;
    static const int m[9] = {0, 1, 2, 1, 3, 4, 2, 4, 5};
    for (int j = 0; j < L; ++j)
      for (int e = 0; e < 9; ++e)
        V[9 * (size_t)h->lm_internal_to_user[j] + e] = j < h->L_obs ? t[(size_t)m[e] * L + j] : 0.0;
  }
  if (gp) {
    if ((rc = fetch(t, d.gp, (size_t)3 * L))) return rc;
    for (int j = 0; j < L; ++j)
      for (int e = 0; e < 3; ++e) gp[3 * (size_t)h->lm_internal_to_user[j] + e] = j < h->L_obs ? t[(size_t)e * L + j] : 0.0;
  }
  if (W) {
    DevTmp t_w;
    const double* w_src = d.W;
    if (!w_src) {  // matrix-free path: no stored factors, compute them for the export
      HIP_TRY(h, hipMalloc(&t_w.p, (size_t)W_ROWS * n * sizeof(double)));
      k_export_w<<<std::min(grid_for(n), 4096), 256, 0, h->stream>>>(d, (const double*)P0, (const double*)X0, static_cast<double*>(t_w.p));
      note_launch(h);
      HIP_TRY(h, hipStreamSynchronize(h->stream));
      w_src = static_cast<const double*>(t_w.p);
    }
    if ((rc = fetch(t, w_src, (size_t)W_ROWS * n))) return rc;
    for (int k = 0; k < n; ++k) {  // the device keeps W factored (ba_kernels.h: w_from_j): rebuild the 6 x 3 block
      double j[W_ROWS];
      for (int e = 0; e < W_ROWS; ++e) j[e] = t[(size_t)e * n + k];
      double* w = W + 18 * (size_t)h->obs_src[k];
      for (int i = 0; i < 3; ++i)
        for (int c = 0; c < 3; ++c) {
          w[3 * i + c] = -(j[i] * j[c] + j[3 + i] * j[3 + c]);
          w[9 + 3 * i + c] = j[6 + i] * j[c] + j[9 + i] * j[3 + c];
        }
    }
  }
  if (S) {
    if ((rc = fetch(t, d.S, (size_t)36 * d.nu))) return rc;
    if (h->small_solve || h->comp_solve || h->dense_solve || h->banded_solve || h->bcr_solve) {  // these solvers apply the damping on the fly; add it for the export
      std::vector<double> ugc, ud((size_t)6 * Cf);
      if ((rc = fetch(ugc, d.Ugc, (size_t)27 * C))) return rc;
      for (int f = 0; f < Cf; ++f)
        for (int i = 0; i < 6; ++i) ud[6 * (size_t)f + i] = ugc[27 * (size_t)h->free2cam[f] + sym6_index(i, i)];
      for (int f = 0; f < Cf; ++f) {
        int sl = -1;
        for (int s2 = h->urow_ptr[f]; s2 < h->urow_ptr[f + 1]; ++s2)
          if (h->ucol[s2] == f) sl = s2;
        for (int i = 0; i < 6; ++i) t[36 * (size_t)sl + 7 * i] += lambda * std::max(ud[6 * (size_t)f + i], h->opt.min_diag);
      }
    }
    const size_t N = (size_t)6 * C;
    std::fill(S, S + N * N, 0.0);
    for (int c = 0; c < C; ++c)
      if (h->fixed[c])
        for (int i = 0; i < 6; ++i) S[((size_t)6 * c + i) * N + 6 * c + i] = 1.0;
    for (int f = 0; f < Cf; ++f)
      for (int s = h->urow_ptr[f]; s < h->urow_ptr[f + 1]; ++s) {
        const int ci = h->free2cam[f], cj = h->free2cam[h->ucol[s]];
        for (int i = 0; i < 6; ++i)
          for (int j = 0; j < 6; ++j) {
            const double v = t[36 * (size_t)s + 6 * i + j];
            S[((size_t)6 * ci + i) * N + 6 * cj + j] = v;
            S[((size_t)6 * cj + j) * N + 6 * ci + i] = v;
          }
      }
  }
  if (rhs) {
    if ((rc = fetch(t, d.rhs, (size_t)6 * Cf))) return rc;
    std::fill(rhs, rhs + (size_t)6 * C, 0.0);
    for (int f = 0; f < Cf; ++f)
      for (int i = 0; i < 6; ++i) rhs[6 * (size_t)h->free2cam[f] + i] = t[6 * (size_t)f + i];
  }
  if (dc) {
    if ((rc = fetch(t, d.dc, (size_t)6 * Cf))) return rc;
    std::fill(dc, dc + (size_t)6 * C, 0.0);
    for (int f = 0; f < Cf; ++f)
      for (int i = 0; i < 6; ++i) dc[6 * (size_t)h->free2cam[f] + i] = t[6 * (size_t)f + i];
  }
  if (dp) {
    std::vector<double> a, b;
    if ((rc = fetch(a, X1, (size_t)3 * L)) || (rc = fetch(b, X0, (size_t)3 * L))) return rc;
    for (int j = 0; j < L; ++j)
      for (int e = 0; e < 3; ++e)
        dp[3 * (size_t)h->lm_internal_to_user[j] + e] = j < h->L_obs ? a[3 * (size_t)j + e] - b[3 * (size_t)j + e] : 0.0;
  }
  return BA_OK;
}

int ba_evaluate_static(int32_t device, int32_t C, int32_t L, int32_t n, const int32_t* obs_cam, const int32_t* obs_pt,
                       const double* obs_uv, const double* intr, const double* poses, const double* points,
                       double* residuals, double* cost, int32_t* n_behind, char* err, int32_t err_len) {
  auto bad = [&](int code, const char* fmt, ...) {
    if (err && err_len > 0) {
      va_list ap;
      va_start(ap, fmt);
      vsnprintf(err, (size_t)err_len, fmt, ap);
      va_end(ap);
    }
    return code;
  };
  if (!obs_cam || !obs_pt || !obs_uv || !intr || !poses || !points)
    return bad(BA_ERR_INVALID, "evaluate_static: null pointer");
  if (C <= 0 || L <= 0 || n <= 0) return bad(BA_ERR_INVALID, "evaluate_static: empty problem (C=%d L=%d n=%d)", C, L, n);
  if (C > ST_MAX_CAMS || L > ST_MAX_PTS || n > ST_MAX_OBS)
    return bad(BA_ERR_INVALID,
               "evaluate_static: C=%d L=%d n=%d exceeds the static scratch (at most %ld cameras, %ld landmarks, %ld "
               "observations): use ba_evaluate", C, L, n, ST_MAX_CAMS, ST_MAX_PTS, ST_MAX_OBS);
  for (int i = 0; i < 7; ++i)
    if (!std::isfinite(intr[i])) return bad(BA_ERR_INVALID, "intrinsics[%d] is not finite", i);
  if (!(intr[0] > 0) || !(intr[1] > 0)) return bad(BA_ERR_INVALID, "focal lengths must be positive");
  for (int k = 0; k < n; ++k) {
    if (obs_cam[k] < 0 || obs_cam[k] >= C) return bad(BA_ERR_INVALID, "obs_cam[%d]=%d out of range", k, obs_cam[k]);
    if (obs_pt[k] < 0 || obs_pt[k] >= L) return bad(BA_ERR_INVALID, "obs_pt[%d]=%d out of range", k, obs_pt[k]);
    if (!std::isfinite(obs_uv[2 * (size_t)k]) || !std::isfinite(obs_uv[2 * (size_t)k + 1]))
      return bad(BA_ERR_INVALID, "observation %d is not finite", k);
  }
  for (size_t i = 0; i < (size_t)3 * L; ++i)
    if (!std::isfinite(points[i])) return bad(BA_ERR_INVALID, "landmark %d has a non-finite coordinate", (int)(i / 3));
#define ST_TRY(expr)                                                                                      \
  do {                                                                                                    \
    hipError_t e_ = (expr);                                                                               \
    if (e_ != hipSuccess) {                                                                               \
      (void)hipGetLastError();                                                                            \
      return bad(BA_ERR_DEVICE, "%s failed: %s", #expr, hipGetErrorString(e_));                           \
    }                                                                                                     \
  } while (0)
  int ndev = 0;
  if (hipGetDeviceCount(&ndev) != hipSuccess || ndev == 0) return bad(BA_ERR_DEVICE, "no HIP device (there is no CPU fallback)");
  int caller_device = 0;
  ST_TRY(hipGetDevice(&caller_device));
  if (device < 0) device = caller_device;
  if (device >= ndev || device >= ST_MAX_DEVICES) return bad(BA_ERR_INVALID, "device %d does not exist", device);
  std::lock_guard<std::mutex> lock(static_scratch_mutex);
  // poses exactly as stage_state prepares them for ba_evaluate: pose 0 is the fixed one (staged in a host array that
  // the lock above protects: 56 KB are too much for a caller's stack)
  static double sp[7 * ST_MAX_CAMS];
  for (int c = 0; c < C; ++c) {
    const double* p = poses + 7 * c;
    for (int i = 0; i < 7; ++i)
      if (!std::isfinite(p[i])) return bad(BA_ERR_INVALID, "pose %d has a non-finite entry", c);
    const double nq = std::sqrt(p[3] * p[3] + p[4] * p[4] + p[5] * p[5] + p[6] * p[6]);
    if (nq == 0.0) return bad(BA_ERR_INVALID, "pose %d has a zero quaternion", c);
    const bool renorm = c != 0 || std::fabs(nq - 1.0) > 1e-6;
    for (int i = 0; i < 3; ++i) sp[7 * c + i] = p[i];
    for (int i = 3; i < 7; ++i) sp[7 * c + i] = renorm ? p[i] / nq : p[i];
  }
  // the caller's current device is put back on every way out
  struct DeviceGuard {
    int prev, cur;
    ~DeviceGuard() { if (cur != prev) (void)hipSetDevice(prev); }
  } guard{caller_device, caller_device};
  if (device != caller_device) {
    ST_TRY(hipSetDevice(device));
    guard.cur = device;
  }
  (void)hipGetLastError();  // an error left by an earlier call on this thread is not a refused launch of this one
  if (!static_scratch_ptr[device]) {  // once per device, with that device current: the address of ITS copy
    double* ptr = nullptr;
    ST_TRY(hipGetSymbolAddress((void**)&ptr, HIP_SYMBOL(scratch)));
    if (!ptr) return bad(BA_ERR_DEVICE, "hipGetSymbolAddress returned a null address for the static scratch");
    static_scratch_ptr[device] = ptr;
  }
  double* const base = static_scratch_ptr[device];
  double* d_partial = base + ST_OFF_PARTIAL;
  double* d_out = base + ST_OFF_OUT;
  double* d_poses = base + ST_OFF_POSES;
  double* d_pts = base + ST_OFF_PTS;
  double* d_uv = base + ST_OFF_UV;
  double* d_res = base + ST_OFF_RES;
  int* d_cam = reinterpret_cast<int*>(base + ST_OFF_CAM);
  int* d_pt = reinterpret_cast<int*>(base + ST_OFF_PT);
  // the array keeps the last call's contents: every region read below is overwritten first, for exactly the
  // extent this call uses (partial sums: one slot per block of THIS launch; nothing needs a clear)
  ST_TRY(hipMemcpy(d_poses, sp, (size_t)7 * C * sizeof(double), hipMemcpyHostToDevice));
  ST_TRY(hipMemcpy(d_pts, points, (size_t)3 * L * sizeof(double), hipMemcpyHostToDevice));
  ST_TRY(hipMemcpy(d_uv, obs_uv, (size_t)2 * n * sizeof(double), hipMemcpyHostToDevice));
  ST_TRY(hipMemcpy(d_cam, obs_cam, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  ST_TRY(hipMemcpy(d_pt, obs_pt, (size_t)n * sizeof(int), hipMemcpyHostToDevice));
  const Intrinsics K{intr[0], intr[1], intr[2], intr[3], intr[4], intr[5], intr[6],
                     (intr[4] != 0.0 || intr[5] != 0.0 || intr[6] != 0.0) ? 1 : 0};
  const int blocks = (n + 255) / 256;
  hipLaunchKernelGGL(k_residuals_static, dim3(blocks), dim3(256), 0, nullptr, K, (int)n, (const int*)d_cam,
                     (const int*)d_pt, (const double*)d_uv, (const double*)d_poses, (const double*)d_pts, d_res, d_partial);
  ST_TRY(hipGetLastError());
  hipLaunchKernelGGL(k_reduce_static, dim3(1), dim3(256), 0, nullptr, (const double*)d_partial, blocks, d_out);
  ST_TRY(hipGetLastError());
  double out[3] = {0.0, 0.0, 0.0};
  ST_TRY(hipMemcpy(out, d_out, sizeof out, hipMemcpyDeviceToHost));
  if (residuals) ST_TRY(hipMemcpy(residuals, d_res, (size_t)2 * n * sizeof(double), hipMemcpyDeviceToHost));
#undef ST_TRY
  if (cost) *cost = out[0];
  if (n_behind) *n_behind = (int32_t)out[1];
  if (out[2] > 0.0) return bad(BA_ERR_NUMERICAL, "cannot project points with zero Z coordinates (division by zero)");
  return BA_OK;
}

}  // extern "C"
