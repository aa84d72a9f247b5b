// Layer (4) of the C ABI: the sharded ALS driver INSIDE the library -- one host thread per device, RCCL between them.
//
// What the reference's host calls per half-iteration is one function (`als_implicit_double(c_ui, X, Y, XtX, ...)`,
// R/model_WRMF.R:111-147 -> src/RcppExports.cpp:371-415) that "internally drives" its OpenMP threads; SURVEY.md 8(b) asks the same
// of the replacement: callable from ONE host thread, driving 1..8 GPUs.  Through round 5 the multi-GPU driver existed only
// above the ABI (rsparse_amd/engine.py: Layout / ShardedALS over torch.distributed) and an R caller got one GPU.  This file is
// that driver in C++ behind `rsparse_hip_ctx_*`:
//   * ownership: contiguous, nnz-balanced blocks of users and of items per rank (`balanced_bounds`), full factor replicas on
//     every device, storage sub-block-major so that sub-block j of all ranks is ONE contiguous slab (`Lay`:
//     the same formulas as engine.Layout);
//   * per half-iteration and rank: Gramian partial over the rank's own rows of the fixed side -> ONE all-gather of
//     [k x k partial, sum(F^2), max |F|] as doubles, summed in rank order by every rank (deterministic) + the fp32-rounded
//     ridge (R/model_WRMF.R:476) -> solve sub-block j with the single-GPU kernels of layer (2), in-place all-gather of its slab
//     on a second stream while sub-block j + 1 is solved -> one all-reduce of [regulariser, row part of the loss];
//   * collectives behind a three-function table (`Comm`): RCCL (ncclCommInitAll: one communicator per device, each used from
//     its rank's thread; the library is dlopen'ed -- a single-GPU user needs no librccl), or SHARED: all "ranks" are threads
//     with streams of their own on ONE device and a collective is a host barrier + device copies.  SHARED is what the
//     `-m gpu` tests run at 2 / 4 / 8 ranks on the one GPU of a test box (the sharding, the sub-block storage, the exchange
//     points and the summation orders are the production code; only the transport differs).
// Every rank thread is persistent: the per-thread workspaces / side streams of layers (1)-(2) (thread_local since this round)
// then belong to one device for the life of the context.
#include <hip/hip_runtime.h>
#include <rccl/rccl.h>   // types and enums; the functions are resolved with dlsym (see RcclApi)

#include <dlfcn.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <cmath>
#include <condition_variable>
#include <cstring>
#include <functional>
#include <memory>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "../../include/rsparse_wrmf_hip.h"
#include "wrmf_internal.h"

namespace rsparse_hip {
namespace {

// ---- who owns which rows, and where they are stored (engine.Layout) ----
struct Lay {
  int64_t n = 0;
  int ws = 1, n_sub = 1;
  std::vector<int64_t> b0, b1;   // block of rank r = [b0[r], b1[r])
  int64_t Bs = 1, rows = 0;
  void init(int64_t n_, const std::vector<int64_t>& edges, int n_sub_) {
    n = n_; ws = (int)edges.size() - 1; n_sub = std::max(1, n_sub_);
    b0.assign(edges.begin(), edges.end() - 1);
    b1.assign(edges.begin() + 1, edges.end());
    int64_t biggest = 1;
    for (int r = 0; r < ws; r++) biggest = std::max(biggest, b1[r] - b0[r]);
    Bs = (biggest + n_sub - 1) / n_sub;
    rows = (int64_t)ws * n_sub * Bs;
  }
  void sub_rows(int r, int j, int64_t& c0, int64_t& c1) const {   // local rows of sub-block j of rank r
    const int64_t nr = b1[r] - b0[r];
    c0 = std::min(nr, (int64_t)j * Bs);
    c1 = std::min(nr, (int64_t)(j + 1) * Bs);
  }
  int64_t sub_start(int r, int j) const { return (int64_t)j * ws * Bs + (int64_t)r * Bs; }
  int64_t slab_start(int j) const { return (int64_t)j * ws * Bs; }
  int64_t to_storage(int64_t g) const {
    // the LAST block whose start is <= g (empty blocks share their start with the next one)
    const int r = (int)(std::upper_bound(b0.begin(), b0.end(), g) - b0.begin()) - 1;
    const int64_t loc = g - b0[r], j = loc / Bs;
    return j * ((int64_t)ws * Bs) + (int64_t)r * Bs + (loc - j * Bs);
  }
};

// contiguous blocks balanced by non-zeros: rank r starts where the prefix sum of the row lengths first reaches r / ws of the
// total (engine.balanced_bounds)
std::vector<int64_t> balanced_edges(const int32_t* p, int64_t n, int ws) {
  std::vector<int64_t> e((size_t)ws + 1, n);
  e[0] = 0;
  if (ws <= 1 || n == 0) return e;
  const int64_t total = (int64_t)p[n] - (int64_t)p[0];
  for (int r = 1; r < ws; r++) {
    if (total > 0) {
      const int64_t target = total * r / ws + (int64_t)p[0];
      // first row i with prefix (p[i + 1]) >= target, + 1
      const int32_t* it = std::lower_bound(p + 1, p + n + 1, target, [](int32_t a, int64_t t) { return (int64_t)a < t; });
      e[(size_t)r] = std::min<int64_t>(n, (int64_t)(it - (p + 1)) + 1);
    } else {
      e[(size_t)r] = n * r / ws;
    }
  }
  for (int r = 1; r <= ws; r++) e[(size_t)r] = std::max(e[(size_t)r], e[(size_t)r - 1]);
  return e;
}

struct HostBarrier {
  std::mutex m;
  std::condition_variable cv;
  int n = 1, count = 0;
  uint64_t gen = 0;
  void wait() {
    std::unique_lock<std::mutex> lk(m);
    const uint64_t g = gen;
    if (++count == n) { count = 0; gen++; cv.notify_all(); }
    else cv.wait(lk, [&] { return gen != g; });
  }
};

// ---- the three collectives of a half-iteration ----
struct Comm {
  virtual ~Comm() {}
  // recv[q * n .. ) = rank q's send (doubles); stream-ordered on s
  virtual int all_gather_f64(int rank, const double* send, double* recv, size_t n, hipStream_t s) = 0;
  // buf = ws slices of `count` floats; rank r's own slice is buf + r * count and holds its contribution
  virtual int all_gather_inplace_f32(int rank, float* buf, size_t count, hipStream_t s) = 0;
  virtual int all_reduce_sum_f64(int rank, double* buf, size_t n, hipStream_t s) = 0;
  virtual const char* name() const = 0;
};

struct RcclApi {
  void* lib = nullptr;
  ncclResult_t (*CommInitAll)(ncclComm_t*, int, const int*) = nullptr;
  ncclResult_t (*CommDestroy)(ncclComm_t) = nullptr;
  ncclResult_t (*AllGather)(const void*, void*, size_t, ncclDataType_t, ncclComm_t, hipStream_t) = nullptr;
  ncclResult_t (*AllReduce)(const void*, void*, size_t, ncclDataType_t, ncclRedOp_t, ncclComm_t, hipStream_t) = nullptr;
  const char* (*GetErrorString)(ncclResult_t) = nullptr;
  std::string load() {
    for (const char* nm : {"librccl.so.1", "librccl.so", "/opt/rocm/lib/librccl.so.1"}) {
      lib = dlopen(nm, RTLD_NOW | RTLD_LOCAL);
      if (lib) break;
    }
    if (!lib) return std::string("cannot load librccl: ") + (dlerror() ? dlerror() : "?");
    auto sym = [&](const char* s) { return dlsym(lib, s); };
    CommInitAll = reinterpret_cast<decltype(CommInitAll)>(sym("ncclCommInitAll"));
    CommDestroy = reinterpret_cast<decltype(CommDestroy)>(sym("ncclCommDestroy"));
    AllGather = reinterpret_cast<decltype(AllGather)>(sym("ncclAllGather"));
    AllReduce = reinterpret_cast<decltype(AllReduce)>(sym("ncclAllReduce"));
    GetErrorString = reinterpret_cast<decltype(GetErrorString)>(sym("ncclGetErrorString"));
    if (!CommInitAll || !CommDestroy || !AllGather || !AllReduce) return "librccl lacks ncclCommInitAll / ncclAllGather / ncclAllReduce";
    return "";
  }
};

struct RcclComm : Comm {
  RcclApi api;
  std::vector<ncclComm_t> comms;
  std::string init(const std::vector<int>& devs) {
    std::string e = api.load();
    if (!e.empty()) return e;
    comms.assign(devs.size(), nullptr);
    const ncclResult_t r = api.CommInitAll(comms.data(), (int)devs.size(), devs.data());
    if (r != ncclSuccess) return std::string("ncclCommInitAll: ") + (api.GetErrorString ? api.GetErrorString(r) : "error");
    return "";
  }
  ~RcclComm() override {
    for (ncclComm_t c : comms)
      if (c) (void)api.CommDestroy(c);
  }
  int all_gather_f64(int rank, const double* send, double* recv, size_t n, hipStream_t s) override {
    return api.AllGather(send, recv, n, ncclDouble, comms[(size_t)rank], s) == ncclSuccess ? 0 : 1;
  }
  int all_gather_inplace_f32(int rank, float* buf, size_t count, hipStream_t s) override {
    return api.AllGather(buf + (size_t)rank * count, buf, count, ncclFloat, comms[(size_t)rank], s) == ncclSuccess ? 0 : 1;
  }
  int all_reduce_sum_f64(int rank, double* buf, size_t n, hipStream_t s) override {
    return api.AllReduce(buf, buf, n, ncclDouble, ncclSum, comms[(size_t)rank], s) == ncclSuccess ? 0 : 1;
  }
  const char* name() const override { return "rccl"; }
};

// All ranks on ONE device (or on devices with peer access): a collective is "everybody's data is ready" (stream sync + host
// barrier), device-to-device copies out of the peers' buffers, and "everybody is done reading" (sync + barrier).
struct SharedComm : Comm {
  int ws;
  HostBarrier bar;
  std::vector<const void*> ptr;
  std::vector<std::vector<double>> host;
  explicit SharedComm(int n) : ws(n), ptr((size_t)n, nullptr), host((size_t)n) { bar.n = n; }
  int all_gather_f64(int rank, const double* send, double* recv, size_t n, hipStream_t s) override {
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    ptr[(size_t)rank] = send;
    bar.wait();
    int bad = 0;
    for (int q = 0; q < ws; q++)
      bad |= hipMemcpyAsync(recv + (size_t)q * n, ptr[(size_t)q], n * sizeof(double), hipMemcpyDeviceToDevice, s) != hipSuccess;
    bad |= hipStreamSynchronize(s) != hipSuccess;
    bar.wait();
    return bad;
  }
  int all_gather_inplace_f32(int rank, float* buf, size_t count, hipStream_t s) override {
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    ptr[(size_t)rank] = buf;
    bar.wait();
    int bad = 0;
    for (int q = 0; q < ws; q++)
      if (q != rank && count > 0)
        bad |= hipMemcpyAsync(buf + (size_t)q * count, static_cast<const float*>(ptr[(size_t)q]) + (size_t)q * count,
                              count * sizeof(float), hipMemcpyDeviceToDevice, s) != hipSuccess;
    bad |= hipStreamSynchronize(s) != hipSuccess;
    bar.wait();
    return bad;
  }
  int all_reduce_sum_f64(int rank, double* buf, size_t n, hipStream_t s) override {
    host[(size_t)rank].resize(n);
    if (hipMemcpyAsync(host[(size_t)rank].data(), buf, n * sizeof(double), hipMemcpyDeviceToHost, s) != hipSuccess) return 1;
    if (hipStreamSynchronize(s) != hipSuccess) return 1;
    bar.wait();
    std::vector<double> sum(n, 0.0);
    for (int q = 0; q < ws; q++)   // rank order: the same bits on every rank
      for (size_t e = 0; e < n; e++) sum[e] += host[(size_t)q][e];
    int bad = hipMemcpyAsync(buf, sum.data(), n * sizeof(double), hipMemcpyHostToDevice, s) != hipSuccess;
    bad |= hipStreamSynchronize(s) != hipSuccess;
    bar.wait();
    return bad;
  }
  const char* name() const override { return "shared"; }
};

struct Sub {
  int64_t c0 = 0, c1 = 0;   // local rows of the block
  rsparse_hip_csc* h = nullptr;
};

struct RankState {
  int rank = 0, device = 0;
  hipStream_t cs = nullptr, ms = nullptr;   // compute, exchange
  hipEvent_t solved = nullptr, gathered = nullptr;
  float *U = nullptr, *V = nullptr;         // replicas, storage order
  std::vector<Sub> sub_items, sub_users;
  float *G = nullptr, *Gpart = nullptr, *absmax = nullptr;
  double *red_in = nullptr, *red_all = nullptr, *scal = nullptr, *loss_sub = nullptr;
  float *cnt_user = nullptr, *cnt_item = nullptr;   // explicit feedback: non-zeros per row of the rank's own block (dynamic lambda)
  double h_scal[4] = {0, 0, 0, 0};
  double solve_ms = 0, comm_ms = 0;
  int rc = 0;
  std::string err;
  // worker thread
  std::thread th;
  std::mutex m;
  std::condition_variable cv;
  std::function<void(RankState&)> job;
  bool has_job = false, quit = false, done = false;
};

}  // namespace
}  // namespace rsparse_hip

using namespace rsparse_hip;

struct rsparse_hip_ctx {
  int ws = 1, kind = 0;
  std::vector<int> devs;
  std::unique_ptr<Comm> comm;
  std::vector<std::unique_ptr<RankState>> ranks;
  Lay lay_u, lay_i;
  int64_t n_user = 0, n_item = 0, nnz = 0;
  int k = 0;
  bool have_matrix = false, have_factors = false;
  int n_sub_max = 1;
};

namespace rsparse_hip {
namespace {

#define CTX_HIP(rs, call)                                                                        \
  do {                                                                                           \
    const hipError_t e_ = (call);                                                                \
    if (e_ != hipSuccess) {                                                                      \
      (rs).rc = RSPARSE_HIP_ERR_RUNTIME;                                                         \
      (rs).err = std::string(#call) + ": " + hipGetErrorString(e_);                              \
      return;                                                                                    \
    }                                                                                            \
  } while (0)
#define CTX_API(rs, call)                                                                        \
  do {                                                                                           \
    const int c_ = (call);                                                                       \
    if (c_ != RSPARSE_HIP_OK) {                                                                  \
      (rs).rc = c_;                                                                              \
      (rs).err = std::string(#call) + ": " + rsparse_hip_last_error();                           \
      return;                                                                                    \
    }                                                                                            \
  } while (0)

void worker(RankState* rs) {
  (void)hipSetDevice(rs->device);
  for (;;) {
    std::function<void(RankState&)> job;
    {
      std::unique_lock<std::mutex> lk(rs->m);
      rs->cv.wait(lk, [&] { return rs->has_job || rs->quit; });
      if (rs->quit) return;
      job = rs->job;
      rs->has_job = false;
    }
    job(*rs);
    {
      std::lock_guard<std::mutex> lk(rs->m);
      rs->done = true;
    }
    rs->cv.notify_all();
  }
}

// run `f` on every rank's thread, wait for all; the first failure is reported through the caller's thread-local error text
int run_all(rsparse_hip_ctx* ctx, const std::function<void(RankState&)>& f) {
  for (auto& r : ctx->ranks) {
    std::lock_guard<std::mutex> lk(r->m);
    r->rc = 0; r->err.clear();
    r->job = f; r->has_job = true; r->done = false;
  }
  for (auto& r : ctx->ranks) r->cv.notify_all();
  for (auto& r : ctx->ranks) {
    std::unique_lock<std::mutex> lk(r->m);
    r->cv.wait(lk, [&] { return r->done; });
  }
  for (auto& r : ctx->ranks)
    if (r->rc) return capi_fail(r->rc, "rank " + std::to_string(r->rank) + ": " + r->err);
  return RSPARSE_HIP_OK;
}

void free_subs(std::vector<Sub>& v) {
  for (Sub& s : v)
    if (s.h) (void)rsparse_hip_csc_destroy(s.h);
  v.clear();
}

void free_rank_buffers(RankState& rs, bool matrix, bool factors) {
  if (matrix) {
    free_subs(rs.sub_items);
    free_subs(rs.sub_users);
    if (rs.cnt_user) (void)hipFree(rs.cnt_user);
    if (rs.cnt_item) (void)hipFree(rs.cnt_item);
    rs.cnt_user = rs.cnt_item = nullptr;
  }
  if (factors) {
    for (void* p : {(void*)rs.U, (void*)rs.V, (void*)rs.G, (void*)rs.Gpart, (void*)rs.absmax, (void*)rs.red_in, (void*)rs.red_all,
                    (void*)rs.scal, (void*)rs.loss_sub})
      if (p) (void)hipFree(p);
    rs.U = rs.V = rs.G = rs.Gpart = rs.absmax = nullptr;
    rs.red_in = rs.red_all = rs.scal = rs.loss_sub = nullptr;
  }
}

// the rank's blocks of one orientation: columns [b0, b1) of the host CSC (p, i, x), row ids translated to the fixed side's
// storage rows, one handle per sub-block
void make_subs(RankState& rs, const Lay& lay_solved, const Lay& lay_fixed, const int32_t* p, const int32_t* i, const double* x,
               std::vector<Sub>& out) {
  const int r = rs.rank;
  const int64_t g0 = lay_solved.b0[(size_t)r];
  for (int j = 0; j < lay_solved.n_sub; j++) {
    Sub s;
    lay_solved.sub_rows(r, j, s.c0, s.c1);
    if (s.c1 > s.c0) {
      const int64_t lo = p[g0 + s.c0], hi = p[g0 + s.c1];
      std::vector<int32_t> sp((size_t)(s.c1 - s.c0) + 1), si((size_t)(hi - lo));
      for (int64_t c = s.c0; c <= s.c1; c++) sp[(size_t)(c - s.c0)] = (int32_t)((int64_t)p[g0 + c] - lo);
      for (int64_t e = lo; e < hi; e++) si[(size_t)(e - lo)] = (int32_t)lay_fixed.to_storage(i[e]);
      CTX_API(rs, rsparse_hip_csc_create_host((int)lay_fixed.rows, (int)(s.c1 - s.c0), sp.data(), si.data(), x + lo, &s.h));
    }
    out.push_back(s);
  }
}

void upload_counts(RankState& rs, const Lay& lay, const int32_t* p, float** d_out) {
  // non-zeros per row of the rank's own block, in block order (the explicit regulariser's weights, wrmf_explicit.hpp:160-170)
  const int64_t g0 = lay.b0[(size_t)rs.rank], g1 = lay.b1[(size_t)rs.rank];
  std::vector<float> c((size_t)std::max<int64_t>(g1 - g0, 1), 0.f);
  for (int64_t g = g0; g < g1; g++) c[(size_t)(g - g0)] = (float)(p[g + 1] - p[g]);
  CTX_HIP(rs, hipMalloc(d_out, c.size() * sizeof(float)));
  CTX_HIP(rs, hipMemcpy(*d_out, c.data(), c.size() * sizeof(float), hipMemcpyHostToDevice));
}

double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

}  // namespace
}  // namespace rsparse_hip

extern "C" {

int rsparse_hip_ctx_create(int n_ranks, const int* device_ids, int comm_kind, rsparse_hip_ctx** out) {
  if (!out) return capi_fail(RSPARSE_HIP_ERR_INVALID, "out is NULL");
  *out = nullptr;
  if (n_ranks < 1 || n_ranks > 64) return capi_fail(RSPARSE_HIP_ERR_INVALID, "n_ranks must be in 1..64");
  if (comm_kind != RSPARSE_HIP_COMM_RCCL && comm_kind != RSPARSE_HIP_COMM_SHARED)
    return capi_fail(RSPARSE_HIP_ERR_INVALID, "comm_kind must be RSPARSE_HIP_COMM_RCCL or RSPARSE_HIP_COMM_SHARED");
  int n_dev = 0;
  if (hipGetDeviceCount(&n_dev) != hipSuccess || n_dev < 1) return capi_fail(RSPARSE_HIP_ERR_RUNTIME, "no HIP device");
  std::unique_ptr<rsparse_hip_ctx> ctx(new rsparse_hip_ctx);
  ctx->ws = n_ranks;
  ctx->kind = comm_kind;
  for (int r = 0; r < n_ranks; r++) {
    const int d = device_ids ? device_ids[r] : (comm_kind == RSPARSE_HIP_COMM_SHARED ? 0 : r);
    if (d < 0 || d >= n_dev) return capi_fail(RSPARSE_HIP_ERR_INVALID, "device id out of range (RCCL: one device per rank)");
    ctx->devs.push_back(d);
  }
  if (comm_kind == RSPARSE_HIP_COMM_RCCL) {
    for (int r = 0; r < n_ranks; r++)
      for (int q = 0; q < r; q++)
        if (ctx->devs[(size_t)r] == ctx->devs[(size_t)q])
          return capi_fail(RSPARSE_HIP_ERR_INVALID, "RCCL needs a different device for every rank (use RSPARSE_HIP_COMM_SHARED to run several ranks on one)");
    if (n_ranks > 1) {
      std::unique_ptr<RcclComm> c(new RcclComm);
      const std::string e = c->init(ctx->devs);
      if (!e.empty()) return capi_fail(RSPARSE_HIP_ERR_RUNTIME, e);
      ctx->comm = std::move(c);
    }
  }
  if (!ctx->comm) ctx->comm.reset(new SharedComm(n_ranks));   // (one rank: every collective is a copy onto itself)
  for (int r = 0; r < n_ranks; r++) {
    std::unique_ptr<RankState> rs(new RankState);
    rs->rank = r;
    rs->device = ctx->devs[(size_t)r];
    ctx->ranks.push_back(std::move(rs));
  }
  for (auto& r : ctx->ranks) r->th = std::thread(worker, r.get());
  const int rc = run_all(ctx.get(), [](RankState& rs) {
    CTX_HIP(rs, hipStreamCreateWithFlags(&rs.cs, hipStreamNonBlocking));
    CTX_HIP(rs, hipStreamCreateWithFlags(&rs.ms, hipStreamNonBlocking));
    CTX_HIP(rs, hipEventCreateWithFlags(&rs.solved, hipEventDisableTiming));
    CTX_HIP(rs, hipEventCreateWithFlags(&rs.gathered, hipEventDisableTiming));
  });
  if (rc) {
    rsparse_hip_ctx_destroy(ctx.release());
    return rc;
  }
  *out = ctx.release();
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_destroy(rsparse_hip_ctx* ctx) {
  if (!ctx) return RSPARSE_HIP_OK;
  (void)run_all(ctx, [](RankState& rs) {
    (void)hipDeviceSynchronize();
    free_rank_buffers(rs, true, true);
    if (rs.cs) (void)hipStreamDestroy(rs.cs);
    if (rs.ms) (void)hipStreamDestroy(rs.ms);
    if (rs.solved) (void)hipEventDestroy(rs.solved);
    if (rs.gathered) (void)hipEventDestroy(rs.gathered);
  });
  for (auto& r : ctx->ranks) {
    {
      std::lock_guard<std::mutex> lk(r->m);
      r->quit = true;
    }
    r->cv.notify_all();
    if (r->th.joinable()) r->th.join();
  }
  ctx->comm.reset();
  delete ctx;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_set_matrix(rsparse_hip_ctx* ctx, int n_user, int n_item, const int32_t* ui_p, const int32_t* ui_i,
                               const double* ui_x, const int32_t* iu_p, const int32_t* iu_i, const double* iu_x,
                               int n_sub_users, int n_sub_items) {
  if (!ctx) return capi_fail(RSPARSE_HIP_ERR_INVALID, "ctx is NULL");
  if (n_user < 0 || n_item < 0 || !ui_p || !iu_p) return capi_fail(RSPARSE_HIP_ERR_INVALID, "bad matrix arguments");
  const int64_t nnz = (int64_t)ui_p[n_item] - ui_p[0];
  if (nnz != (int64_t)iu_p[n_user] - iu_p[0]) return capi_fail(RSPARSE_HIP_ERR_INVALID, "the two orientations hold different numbers of non-zeros");
  if (nnz > 0 && (!ui_i || !ui_x || !iu_i || !iu_x)) return capi_fail(RSPARSE_HIP_ERR_INVALID, "index / value arrays are NULL");
  for (int64_t e = 0; e < nnz; e++) {
    if (ui_i[e] < 0 || ui_i[e] >= n_user) return capi_fail(RSPARSE_HIP_ERR_INVALID, "row index of c_ui outside [0, n_user)");
    if (iu_i[e] < 0 || iu_i[e] >= n_item) return capi_fail(RSPARSE_HIP_ERR_INVALID, "row index of c_iu outside [0, n_item)");
  }
  const int ws = ctx->ws;
  const int dflt = ws <= 1 ? 1 : 4;   // sub-blocks per rank and half-iteration (engine.default_subblocks)
  ctx->lay_u.init(n_user, balanced_edges(iu_p, n_user, ws), n_sub_users > 0 ? n_sub_users : dflt);
  ctx->lay_i.init(n_item, balanced_edges(ui_p, n_item, ws), n_sub_items > 0 ? n_sub_items : dflt);
  ctx->n_user = n_user; ctx->n_item = n_item; ctx->nnz = nnz;
  ctx->n_sub_max = std::max(ctx->lay_u.n_sub, ctx->lay_i.n_sub);
  ctx->have_matrix = false;
  const Lay& lu = ctx->lay_u;
  const Lay& li = ctx->lay_i;
  const int rc = run_all(ctx, [&](RankState& rs) {
    free_rank_buffers(rs, true, false);
    // item half: columns = my items (c_ui), rows = users -> the user layout's storage rows; user half likewise
    make_subs(rs, li, lu, ui_p, ui_i, ui_x, rs.sub_items);
    if (rs.rc) return;
    make_subs(rs, lu, li, iu_p, iu_i, iu_x, rs.sub_users);
    if (rs.rc) return;
    upload_counts(rs, lu, iu_p, &rs.cnt_user);
    if (rs.rc) return;
    upload_counts(rs, li, ui_p, &rs.cnt_item);
  });
  if (rc) return rc;
  ctx->have_matrix = true;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_set_factors(rsparse_hip_ctx* ctx, int rank, const float* U, const float* V) {
  if (!ctx || !ctx->have_matrix) return capi_fail(RSPARSE_HIP_ERR_INVALID, "set the matrix first");
  if (rank < 1 || rank > RSPARSE_HIP_MAX_RANK) return capi_fail(RSPARSE_HIP_ERR_UNSUPPORTED, "rank outside 1..256");
  if (!U || !V) return capi_fail(RSPARSE_HIP_ERR_INVALID, "U or V is NULL");
  const int k = rank;
  const Lay& lu = ctx->lay_u;
  const Lay& li = ctx->lay_i;
  // storage-order images on the host, once; every rank uploads them
  std::vector<float> Us((size_t)lu.rows * k, 0.f), Vs((size_t)li.rows * k, 0.f);
  for (int64_t g = 0; g < ctx->n_user; g++) std::memcpy(&Us[(size_t)lu.to_storage(g) * k], U + (size_t)g * k, (size_t)k * sizeof(float));
  for (int64_t g = 0; g < ctx->n_item; g++) std::memcpy(&Vs[(size_t)li.to_storage(g) * k], V + (size_t)g * k, (size_t)k * sizeof(float));
  const int ws = ctx->ws, nsm = ctx->n_sub_max;
  const bool realloc_ = ctx->k != k || !ctx->have_factors;
  const int rc = run_all(ctx, [&](RankState& rs) {
    if (realloc_) {
      free_rank_buffers(rs, false, true);
      const size_t kk = (size_t)k * k;
      CTX_HIP(rs, hipMalloc(&rs.U, std::max<size_t>(Us.size(), 1) * sizeof(float)));
      CTX_HIP(rs, hipMalloc(&rs.V, std::max<size_t>(Vs.size(), 1) * sizeof(float)));
      CTX_HIP(rs, hipMalloc(&rs.G, kk * sizeof(float)));
      CTX_HIP(rs, hipMalloc(&rs.Gpart, kk * sizeof(float)));
      CTX_HIP(rs, hipMalloc(&rs.absmax, sizeof(float)));
      CTX_HIP(rs, hipMalloc(&rs.red_in, (kk + 2) * sizeof(double)));
      CTX_HIP(rs, hipMalloc(&rs.red_all, (size_t)ws * (kk + 2) * sizeof(double)));
      CTX_HIP(rs, hipMalloc(&rs.scal, 4 * sizeof(double)));
      CTX_HIP(rs, hipMalloc(&rs.loss_sub, (size_t)std::max(nsm, 1) * sizeof(double)));
    }
    CTX_HIP(rs, hipMemcpy(rs.U, Us.data(), Us.size() * sizeof(float), hipMemcpyHostToDevice));
    CTX_HIP(rs, hipMemcpy(rs.V, Vs.data(), Vs.size() * sizeof(float), hipMemcpyHostToDevice));
  });
  if (rc) return rc;
  ctx->k = k;
  ctx->have_factors = true;
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_get_factors(rsparse_hip_ctx* ctx, float* U, float* V) {
  if (!ctx || !ctx->have_factors) return capi_fail(RSPARSE_HIP_ERR_INVALID, "no factors in the context");
  const int k = ctx->k;
  const Lay& lu = ctx->lay_u;
  const Lay& li = ctx->lay_i;
  std::vector<float> Us(U ? (size_t)lu.rows * k : 0), Vs(V ? (size_t)li.rows * k : 0);
  const int rc = run_all(ctx, [&](RankState& rs) {
    if (rs.rank != 0) return;   // every replica holds the same bits
    CTX_HIP(rs, hipStreamSynchronize(rs.cs));
    if (U) CTX_HIP(rs, hipMemcpy(Us.data(), rs.U, Us.size() * sizeof(float), hipMemcpyDeviceToHost));
    if (V) CTX_HIP(rs, hipMemcpy(Vs.data(), rs.V, Vs.size() * sizeof(float), hipMemcpyDeviceToHost));
  });
  if (rc) return rc;
  if (U) for (int64_t g = 0; g < ctx->n_user; g++) std::memcpy(U + (size_t)g * k, &Us[(size_t)lu.to_storage(g) * k], (size_t)k * sizeof(float));
  if (V) for (int64_t g = 0; g < ctx->n_item; g++) std::memcpy(V + (size_t)g * k, &Vs[(size_t)li.to_storage(g) * k], (size_t)k * sizeof(float));
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_half_iteration(rsparse_hip_ctx* ctx, int side, int implicit, double lambda, unsigned solver, unsigned cg_steps,
                                   int dynamic_lambda, double* loss_out) {
  if (!ctx || !ctx->have_matrix || !ctx->have_factors) return capi_fail(RSPARSE_HIP_ERR_INVALID, "the context needs a matrix and factors");
  if (side != RSPARSE_HIP_SIDE_ITEMS && side != RSPARSE_HIP_SIDE_USERS) return capi_fail(RSPARSE_HIP_ERR_INVALID, "side must be 0 (items) or 1 (users)");
  const int k = ctx->k, ws = ctx->ws;
  const Lay& layF = side == RSPARSE_HIP_SIDE_ITEMS ? ctx->lay_u : ctx->lay_i;   // fixed side
  const Lay& layS = side == RSPARSE_HIP_SIDE_ITEMS ? ctx->lay_i : ctx->lay_u;   // solved side
  Comm* comm = ctx->comm.get();
  const float ridge = (float)lambda;   // fl(diag(lambda)), R/model_WRMF.R:476
  const int rc = run_all(ctx, [&](RankState& rs) {
    float* F = side == RSPARSE_HIP_SIDE_ITEMS ? rs.U : rs.V;
    float* S = side == RSPARSE_HIP_SIDE_ITEMS ? rs.V : rs.U;
    const std::vector<Sub>& subs = side == RSPARSE_HIP_SIDE_ITEMS ? rs.sub_items : rs.sub_users;
    const float* cntF = side == RSPARSE_HIP_SIDE_ITEMS ? rs.cnt_user : rs.cnt_item;
    hipStream_t cs = rs.cs, ms = rs.ms;
    const int r = rs.rank;
    const size_t kk = (size_t)k * k;
    const double t0 = now_ms();
    double t_comm = 0.0;
    // ---- Gramian of the fixed side: own rows, one exchange ----
    if (implicit) {
      CTX_HIP(rs, hipMemsetAsync(rs.absmax, 0, sizeof(float), cs));
      int pieces = 0;
      for (int j = 0; j < layF.n_sub; j++) {
        int64_t c0, c1;
        layF.sub_rows(r, j, c0, c1);
        pieces += c1 > c0;
      }
      if (ws == 1 && pieces == 1 && layF.n_sub == 1) {   // one rank, one block: the single-GPU call, bit for bit
        CTX_API(rs, rsparse_hip_gramian_absmax_device(F, k, layF.b1[0] - layF.b0[0], lambda, rs.G, rs.scal, rs.absmax, cs));
      } else {
        CTX_HIP(rs, hipMemsetAsync(rs.red_in, 0, (kk + 2) * sizeof(double), cs));
        for (int j = 0; j < layF.n_sub; j++) {
          int64_t c0, c1;
          layF.sub_rows(r, j, c0, c1);
          if (c1 <= c0) continue;
          const float* Fb = F + (size_t)layF.sub_start(r, j) * k;
          CTX_API(rs, rsparse_hip_gramian_absmax_device(Fb, k, c1 - c0, 0.0, rs.Gpart, rs.scal + 2, rs.absmax, cs));
          CTX_HIP(rs, launch_ctx_accumulate(rs.Gpart, rs.scal + 2, rs.red_in, k, cs));
        }
        CTX_HIP(rs, launch_ctx_put_absmax(rs.absmax, rs.red_in, k, cs));
        const double tc = now_ms();
        if (comm->all_gather_f64(r, rs.red_in, rs.red_all, kk + 2, cs)) { rs.rc = RSPARSE_HIP_ERR_RUNTIME; rs.err = "Gramian exchange failed"; return; }
        t_comm += now_ms() - tc;
        CTX_HIP(rs, launch_ctx_reduce(rs.red_all, ws, k, ridge, rs.G, rs.scal, rs.absmax, cs));
      }
    }
    // ---- solve sub-block j, exchange its slab on the second stream while sub-block j + 1 is solved ----
    CTX_HIP(rs, hipMemsetAsync(rs.loss_sub, 0, (size_t)layS.n_sub * sizeof(double), cs));
    for (int j = 0; j < layS.n_sub; j++) {
      const Sub& sb = subs[(size_t)j];
      if (sb.c1 > sb.c0) {
        float* Sb = S + (size_t)layS.sub_start(r, j) * k;
        if (implicit)
          CTX_API(rs, rsparse_hip_als_implicit_device(sb.h, F, Sb, rs.G, k, lambda, solver, cg_steps, rs.absmax, rs.loss_sub + j, cs));
        else
          CTX_API(rs, rsparse_hip_als_explicit_device(sb.h, F, Sb, k, lambda, solver, cg_steps, dynamic_lambda, rs.loss_sub + j, cs));
      }
      if (ws > 1) {
        CTX_HIP(rs, hipEventRecord(rs.solved, cs));
        CTX_HIP(rs, hipStreamWaitEvent(ms, rs.solved, 0));
        const double tc = now_ms();
        if (comm->all_gather_inplace_f32(r, S + (size_t)layS.slab_start(j) * k, (size_t)layS.Bs * k, ms)) {
          rs.rc = RSPARSE_HIP_ERR_RUNTIME; rs.err = "slab exchange failed"; return;
        }
        t_comm += now_ms() - tc;
      }
    }
    if (ws > 1) {
      CTX_HIP(rs, hipEventRecord(rs.gathered, ms));
      CTX_HIP(rs, hipStreamWaitEvent(cs, rs.gathered, 0));
    }
    // ---- loss: row part of the sub-blocks + the regulariser on the fixed side (wrmf_implicit.hpp:286-301, wrmf_explicit.hpp:146-173) ----
    CTX_HIP(rs, launch_ctx_sum(rs.loss_sub, layS.n_sub, rs.scal + 1, cs));
    int n_red = 1;
    double* red = rs.scal + 1;
    if (!implicit) {   // implicit: scal[0] = sum(F^2) over ALL ranks came out of the Gramian exchange
      CTX_HIP(rs, hipMemsetAsync(rs.scal, 0, sizeof(double), cs));
      if (lambda > 0) {
        for (int j = 0; j < layF.n_sub; j++) {
          int64_t c0, c1;
          layF.sub_rows(r, j, c0, c1);
          if (c1 <= c0) continue;
          const float* Fb = F + (size_t)layF.sub_start(r, j) * k;
          CTX_API(rs, rsparse_hip_weighted_sumsq_device(Fb, k, c1 - c0, dynamic_lambda ? cntF + c0 : nullptr, rs.scal + 2, cs));
          CTX_HIP(rs, launch_ctx_add(rs.scal + 2, rs.scal, cs));
        }
      }
      n_red = 2;
      red = rs.scal;
    }
    if (ws > 1) {
      const double tc = now_ms();
      if (comm->all_reduce_sum_f64(r, red, (size_t)n_red, cs)) { rs.rc = RSPARSE_HIP_ERR_RUNTIME; rs.err = "loss all-reduce failed"; return; }
      t_comm += now_ms() - tc;
    }
    CTX_HIP(rs, hipMemcpyAsync(rs.h_scal, rs.scal, 2 * sizeof(double), hipMemcpyDeviceToHost, cs));
    CTX_HIP(rs, hipStreamSynchronize(cs));
    rs.comm_ms = t_comm;
    rs.solve_ms = now_ms() - t0;
  });
  if (rc) return rc;
  if (loss_out) {
    const RankState& r0 = *ctx->ranks[0];
    *loss_out = ctx->nnz > 0 ? (r0.h_scal[1] + lambda * r0.h_scal[0]) / (double)ctx->nnz : 0.0;
  }
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_take_numeric_failures(rsparse_hip_ctx* ctx, int64_t* unresolved_out, int64_t* fallback_out) {
  if (!ctx || !unresolved_out) return capi_fail(RSPARSE_HIP_ERR_INVALID, "NULL argument");
  std::vector<int64_t> bad((size_t)ctx->ws, 0), fell((size_t)ctx->ws, 0);
  const int rc = run_all(ctx, [&](RankState& rs) {
    CTX_API(rs, rsparse_hip_take_numeric_failures(&bad[(size_t)rs.rank], &fell[(size_t)rs.rank]));
  });
  if (rc) return rc;
  *unresolved_out = 0;
  if (fallback_out) *fallback_out = 0;
  for (int r = 0; r < ctx->ws; r++) {
    *unresolved_out += bad[(size_t)r];
    if (fallback_out) *fallback_out += fell[(size_t)r];
  }
  return RSPARSE_HIP_OK;
}

int rsparse_hip_ctx_info(const rsparse_hip_ctx* ctx, int64_t info_out[16], double times_out[2]) {
  if (!ctx || !info_out) return capi_fail(RSPARSE_HIP_ERR_INVALID, "NULL argument");
  for (int i = 0; i < 16; i++) info_out[i] = 0;
  info_out[0] = ctx->ws;
  info_out[1] = ctx->kind;
  info_out[2] = ctx->n_user;
  info_out[3] = ctx->n_item;
  info_out[4] = ctx->nnz;
  info_out[5] = ctx->k;
  info_out[6] = ctx->lay_u.n_sub;
  info_out[7] = ctx->lay_i.n_sub;
  info_out[8] = ctx->lay_u.Bs;
  info_out[9] = ctx->lay_i.Bs;
  info_out[10] = ctx->have_matrix ? ctx->lay_u.b1[0] - ctx->lay_u.b0[0] : 0;   // users of rank 0
  info_out[11] = ctx->have_matrix ? ctx->lay_i.b1[0] - ctx->lay_i.b0[0] : 0;   // items of rank 0
  info_out[12] = ctx->comm && std::string(ctx->comm->name()) == "rccl";
  if (times_out) {
    times_out[0] = times_out[1] = 0.0;
    for (const auto& r : ctx->ranks) {   // the slowest rank of the last half-iteration: wall time, time inside collectives
      times_out[0] = std::max(times_out[0], r->solve_ms);
      times_out[1] = std::max(times_out[1], r->comm_ms);
    }
  }
  return RSPARSE_HIP_OK;
}

}  // extern "C"
