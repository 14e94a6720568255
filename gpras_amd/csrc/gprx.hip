// libgprx: C ABI (include/gprx.h) over the gfx950 kernels in this directory.
// Host orchestration only: parameter transforms and priors (scalar math), buffer ownership,
// launch sequences.  No CPU fallback exists for any device stage.
#include "../../include/gprx.h"

#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <limits>
#include <map>
#include <mutex>
#include <string>
#include <thread>
#include <vector>

#include "comm.h"
#include "gemm_f64.h"
#include "gprx_common.h"
#include "grad.h"
#include "kmat.h"
#include "kmeans.h"
#include "metrics.h"
#include "pca.h"
#include "potrf.h"
#include "potrf_cell.h"
#include "potrf_dag.h"
#include "sgpr.h"
#include "sgpr_asm.h"
#include "sgpr_fused.h"
#include "solve.h"

using namespace gprx;

namespace {

thread_local std::string g_err;

struct Buf {
  double* p = nullptr;
  size_t bytes = 0;
  bool borrowed = false;  // view into the batch arena (gprx_select_slot): never freed through this Buf
};

}  // namespace

struct Theta {
  double variance, noise;
  std::vector<double> ls;
  double w_var, w_noise;
  std::vector<double> w_len;
};

struct gprx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t n = 0, m = 0, np = 0, mp = 0;
  int d = 0, kid = 0, ard = 0, nlen = 1, ntheta = 3, n_units = 0;
  int dist_form = 0;  // GPRX_DIST_DIFFERENCE / GPRX_DIST_EXPANDED
  DagPlan dag;        // task list and state words of the tile-DAG factorisation of a lone matrix (potrf_dag.h), created on first use
  bool dag_used = false;  // the current single factorisation ran through it: its abort word travels with the results
  PotrfTuning tune;   // schedule knobs of this handle: the process defaults at creation, then gprx_set_handle_tuning
  int predict_path = 0;
  std::string err;
  // data
  Buf X, Y, Z, invls, alpha, red, Kmat, invD, Xinv, Tmp, partial, xs, Ks, pred;
  // sparse path
  Buf P, Am, Qm, Bm, invDL, invDB, SM, WP, WHP, WHQ, vecs, dZ, dstage, splitws;
  std::vector<double> yy;  // y.y per unit
  double elbo_trAAT = 0.0;
  int* info = nullptr;
  double* pin = nullptr;  // pinned host staging: [0..63] lengthscales up, [64..71] reductions down, [72] info (as int),
                          // [74] variance, [75] noise (graph replay reads them through the device block `gparams`)
  double* gparams = nullptr;            // device {variance, noise} for captured kernel-matrix builds
  std::map<int, hipGraphExec_t> graphs;  // unit -> captured single-stream exact factorisation
  std::map<std::pair<int, int>, hipGraphExec_t> sgraphs;  // (cells, with gradient) -> captured sparse batch evaluation
  bool sgraph_off = false;                                // a capture failed once: this handle stays on eager launches
  int sgpr_fused = 1;                                     // M <= 64: the five-launch evaluation of sgpr_fused.h ("sgpr_fused" tuning key)
  Buf adam_dev;                                           // device state of the resident Adam loop (sgpr_adam_resident)
  double* adam_pin = nullptr;                             // pinned: stop flags of the cells + the error word, read every few steps
  size_t adam_pin_bytes = 0;
  unsigned long long* sf_stamps = nullptr;                // development aid (gprx_sf_stamps): phase stamps of the fused sparse kernels
  static constexpr int SF_MAX_GROUPS = 2;
  hipStream_t sf_streams[SF_MAX_GROUPS - 1] = {};          // extra streams of the resident Adam loop: large batches run as groups of cells (sf_group_plan)
  hipEvent_t sf_evs[SF_MAX_GROUPS] = {};
  bool sparse_view = false;                               // the current single-model factorisation lives in cell block 0 of `sarena`
  // current factorisation
  bool factorized = false;
  bool have_linv = false;  // Xinv holds L^-1 of the current factorisation (exact path)
  int cur_unit = -1;
  double variance = 1.0, noise = 1.0;
  std::vector<double> ls;
  double timings[4] = {0, 0, 0, 0};
  bool profiling = false;
  PotrfProfile prof;
  PotrfStreams pstreams;
  double prof_out[8] = {0, 0, 0, 0, 0, 0, 0, 0};
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
  // batched exact factorisations (gprx_factorize_batch): `arena_slots` cell blocks, `cell_stride` doubles apart, each
  // [K (np + 64) x np | invD np x 64 | staged diagonal blocks np x 128 | alpha np]; parameter / result tables, one row per cell
  Buf arena, cellpar, cellres, garena, gpartial;  // garena: per cell [L^-1 | K^-1] for batched gradients
  Buf apart;                                      // row-chunk partial sums of alpha_from_inverse
  Buf twork;                                      // work vector of the lone backward solve (np doubles)
  hipEvent_t kev[2] = {nullptr, nullptr};         // profiling: events around the kernel-build launch
  double kmat_ms = 0.0, kmat_bytes = 0.0;
  std::vector<Theta> batch_thetas;                // gprx_factorize_batch: decoded parameter sets of the last call (buffers reused)
  std::vector<double> batch_lml;
  hipEvent_t wev = nullptr;                       // completion event of wait_stream
  hipEvent_t cev[2] = {nullptr, nullptr};         // profiling: events around the one-workgroup-per-cell kernel's launch
  bool cev_recorded = false;
  double cell_ms = 0.0, cell_flops = 0.0, cell_cells = 0.0;
  Buf sarena;                                   // batched sparse models: one cell block per slot (sgpr_batch_layout)
  int sarena_slots = 0;
  double* spin = nullptr;  // pinned staging of the sparse batch: parameters up, reductions / gradients down
  size_t spin_doubles = 0;
  double* bpin = nullptr;  // pinned: [slots][CELL_PAR] parameters up, then [slots][CELL_RES] results down
  int arena_slots = 0;
  int64_t cell_stride = 0, off_invd = 0, off_stage = 0, off_alpha = 0;
  std::vector<Theta> slot_theta;
  std::vector<int> slot_unit;
  std::vector<char> slot_ok;
  double batch_ms = 0.0;  // device time of the last batch (events around the whole batch)
  hipEvent_t bev[2] = {nullptr, nullptr};
  hipEvent_t stagger_evt = nullptr;
};

struct gprx_comm_ctx {
  int device = 0, rank = 0, world = 1;
  ncclComm_t comm = nullptr;
  hipStream_t stream = nullptr;
  double* scratch = nullptr;  // device staging of the host-buffer entry points
  size_t scratch_bytes = 0;
  std::string err;
};

struct gprx_pca_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  int64_t cells = 0, cells_p = 0;  // cells_p: leading dimension of the device copies (multiple of 16, zero padded)
  int k = 0, depth = 0;
  Buf mu, wfwd, wrev, elev, E, base, xm, xs;  // per-cell parameters expanded to all cells; E: (k, cells_p)
  Buf dX, dZ, ws, dMean, dVar, dFull, dVfull;
  std::string err;
};

namespace {

// kernel-matrix and trace launches evaluate r2 in the handle's distance form (gprx_set_distance_form)
template <class Args>
Args with_form(Args a, gprx_handle h) {
  a.form = h->dist_form;
  return a;
}

int fail(gprx_handle h, int code, const std::string& msg) {
  if (h) h->err = msg;
  g_err = msg;
  return code;
}

#define HIPCHK(h, expr)                                                                            \
  do {                                                                                             \
    hipError_t e_ = (expr);                                                                        \
    if (e_ != hipSuccess) {                                                                        \
      return fail(h, e_ == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP,                          \
                  std::string(#expr) + ": " + hipGetErrorString(e_));                              \
    }                                                                                              \
  } while (0)

int& predict_path_tuning() {
  static int v = 0;  // 0: choose, 1: always through L^-1, 2: always forward substitution
  return v;
}

// The legacy (NULL) stream is never used.  A legacy-stream call (hipMemcpy, hipMemset, a launch on stream 0, hipDeviceSynchronize)
// from one host thread is refused while ANOTHER thread captures a graph ("operation would make the legacy stream depend on a
// capturing blocking stream") and invalidates that capture -- in every capture mode of this runtime.  Synchronous copies and
// the handle-less entry points go through one non-blocking utility stream per device instead.
hipStream_t util_stream() {
  static std::mutex m;
  static std::map<int, hipStream_t> streams;
  int dev = 0;
  if (hipGetDevice(&dev) != hipSuccess) return nullptr;
  std::lock_guard<std::mutex> lock(m);
  auto it = streams.find(dev);
  if (it != streams.end()) return it->second;
  hipStream_t st = nullptr;
  if (hipStreamCreateWithFlags(&st, hipStreamNonBlocking) != hipSuccess) return nullptr;
  streams.emplace(dev, st);
  return st;
}
hipError_t copy_sync(void* dst, const void* src, size_t bytes, hipMemcpyKind kind) {
  hipStream_t st = util_stream();
  if (!st) return hipErrorInvalidValue;
  hipError_t e = hipMemcpyAsync(dst, src, bytes, kind, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}
hipError_t memset_sync(void* dst, int value, size_t bytes) {
  hipStream_t st = util_stream();
  if (!st) return hipErrorInvalidValue;
  hipError_t e = hipMemsetAsync(dst, value, bytes, st);
  return e != hipSuccess ? e : hipStreamSynchronize(st);
}

int ensure(gprx_handle h, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return GPRX_OK;
  if (b.p && !b.borrowed) HIPCHK(h, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  b.borrowed = false;
  HIPCHK(h, hipMalloc((void**)&b.p, bytes));
  b.bytes = bytes;
  return GPRX_OK;
}

int ensure_zeroed(gprx_handle h, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return GPRX_OK;
  int rc = ensure(h, b, bytes);
  if (rc) return rc;
  HIPCHK(h, hipMemsetAsync(b.p, 0, bytes, h->stream));
  return GPRX_OK;
}

// ---- scalar transforms (gpflow positive() / LogNormal(0,1) priors; see oracle/transforms.py) ------
// (round 5: the portable forms of px_math.h -- the device-resident Adam loop evaluates the same functions inside a kernel and must get the
// same bits as this host code)
double softplus(double w) { return px_softplus(w); }
double sigmoid(double w) { return px_sigmoid(w); }
double ln_logpdf(double u) { return px_ln_logpdf(u); }
double ln_dlogpdf(double u) { return px_ln_dlogpdf(u); }

__global__ void set_rhs_rows_kernel(double* dst, int64_t ld, const double* y, int n, int np, int rows) {
  const int64_t total = (int64_t)rows * np;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / np), c = (int)(e % np);
    dst[(int64_t)r * ld + c] = (r == 0 && c < n) ? y[c] : 0.0;
  }
}

__global__ void copy_row_kernel(const double* src, double* dst, int n) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[i] = src[i];
}

// back-to-back MFMA issue, 4 independent accumulators per wave, one wave per SIMD
__global__ __launch_bounds__(256) void mfma_f64_peak_kernel(double* out, int iters) {
  d4 c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  const double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  const d4 s = c0 + c1 + c2 + c3;
  if (s.x == 123.456) out[0] = s.y;
}

void decode_theta_into(gprx_handle h, const double* theta, Theta& t) {
  t.w_var = theta[0];
  t.w_noise = theta[1 + h->nlen];
  t.variance = softplus(t.w_var);
  t.noise = NOISE_LOWER + softplus(t.w_noise);
  t.w_len.assign(theta + 1, theta + 1 + h->nlen);
  t.ls.resize(h->d);
  if (h->ard) {
    for (int k = 0; k < h->d; ++k) t.ls[k] = softplus(t.w_len[k]);
  } else {
    const double l = softplus(t.w_len[0]);  // (one shared lengthscale: one softplus, not d)
    for (int k = 0; k < h->d; ++k) t.ls[k] = l;
  }
}
Theta decode_theta(gprx_handle h, const double* theta) {
  Theta t;
  decode_theta_into(h, theta, t);
  return t;
}

// Wait for everything enqueued on `st`.  hipStreamSynchronize blocks in the driver and returns 60-100 us after the GPU has finished
// (tools/batch_overhead.py: 150 us of host time around a 1.1 ms batch of 512 cells of N = 512 -- 12 % of the call, 3 % at N = 1024, 4 % of a
// lone N = 4096 fit); a completion event polled from the calling thread returns within a few us.  Pure spinning for the first 2 ms, then the
// poll yields the core between queries, and after 200 ms (the long batched steps, where the wake-up latency is noise) it hands over to
// hipStreamSynchronize.  GPRX_WAIT_BLOCKING=1 restores the blocking wait.
int& wait_handover_us() {
  static int v = 200000;  // gprx_set_tuning("wait_handover_us", ...): tests lower it to 0 to force the hand-over path
  return v;
}
hipError_t wait_stream(gprx_handle h, hipStream_t st) {
  static const bool blocking = getenv("GPRX_WAIT_BLOCKING") && atoi(getenv("GPRX_WAIT_BLOCKING")) != 0;
  if (blocking) return hipStreamSynchronize(st);
  if (!h->wev) {
    hipError_t e = hipEventCreateWithFlags(&h->wev, hipEventDisableTiming);
    if (e != hipSuccess) return e;
  }
  hipError_t e = hipEventRecord(h->wev, st);
  if (e != hipSuccess) return e;
  const auto t0 = std::chrono::steady_clock::now();
  for (int spins = 0;; ++spins) {
    e = hipEventQuery(h->wev);
    if (e != hipErrorNotReady) {
      // (a poll that found the event pending may have left hipErrorNotReady behind as the thread's "last error": the launch helpers end
      // with hipGetLastError() and must not trip over it)
      if (spins > 0) (void)hipGetLastError();
      return e;
    }
    if ((spins & 63) == 63) {
      const double us = std::chrono::duration<double, std::micro>(std::chrono::steady_clock::now() - t0).count();
      if (us > (double)wait_handover_us()) {
        // (the polls above left hipErrorNotReady as the thread's last error -- it is sticky across later successful calls on
        // this runtime -- and the next launch helper ending in hipGetLastError() would report it for a good call)
        (void)hipGetLastError();
        return hipStreamSynchronize(st);
      }
      if (us > 2000.0) std::this_thread::yield();
    }
  }
}

double log_prior(gprx_handle h, const Theta& t, int mask) {
  double lp = 0.0;
  if (mask & GPRX_TRAIN_VARIANCE) lp += ln_logpdf(t.variance);
  if (mask & GPRX_TRAIN_LENGTHSCALE)
    for (int k = 0; k < h->nlen; ++k) lp += ln_logpdf(t.ls[k]);
  if (mask & GPRX_TRAIN_NOISE) lp += ln_logpdf(t.noise);
  return lp;
}

int upload_inv_ls(gprx_handle h, const Theta& t) {  // uploads the lengthscales (kernels divide by them)
  if (h->d <= 64) {
    // pinned staging: truly asynchronous (the previous use of the staging area was synchronised by the
    // previous call's final stream synchronisation)
    std::memcpy(h->pin, t.ls.data(), sizeof(double) * h->d);
    HIPCHK(h, hipMemcpyAsync(h->invls.p, h->pin, sizeof(double) * h->d, hipMemcpyHostToDevice, h->stream));
    return GPRX_OK;
  }
  HIPCHK(h, hipMemcpyAsync(h->invls.p, t.ls.data(), sizeof(double) * h->d, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GPRX_OK;
}

// ---- exact GP ------------------------------------------------------------------------------------
// K = k(X,X) + s I (lower tiles) with y appended as row np; potrf gives L and beta = L^-1 y in that
// row; alpha by the backward solve; red[0] = sum log diag L, red[1] = |beta|^2.
int ensure_lookahead(gprx_handle h) {
  if (h->pstreams.aux) return GPRX_OK;
  HIPCHK(h, h->pstreams.init());
  return GPRX_OK;
}

// "dag" (gprx_set_tuning) = 1: a lone matrix takes the tile-DAG factorisation (potrf_dag.h: one persistent launch, the dependent
// chain in one workgroup, tile tasks ordered by version counters).  Opt-in: measured on MI355X at N = 4096 it reaches 2.33 ms
// against 2.16 ms for the launch-per-panel schedule (every row block has a dependent TRSM -> update pair
// per column, each costing two or more ~2 us memory hops, as much as the 1.7 us of MFMA work in a 64^3 tile).
bool use_dag(const PotrfTuning& tune, int np) { return tune.dag > 0 && np >= NB; }

// Batched cells: the one-workgroup-per-cell factorisation (potrf_cell.h) for matrices of at most 1024 rows once the batch has
// enough cells ("cell_kernel": 1 always, -1 never).  Equal to the batched launch sequence to rounding, not bit for bit (a tile's
// whole update is one sum there); larger matrices or fewer cells keep the launch sequence (bit-identical to single calls).
bool use_cell_kernel(const PotrfTuning& tune, int np, int cells) {
  if (tune.cell_kernel < 0) return false;
  if (tune.cell_kernel > 0) return true;
  // measured crossovers (tools/batch_n1024.py, fits/s cell kernel vs launch sequence): N = 256: 137 k vs 118 k at 32 cells (76 k vs 92 k
  // at 16); N = 512: tie at 128 cells, 262 k vs 223 k at 256; N = 1024: 64.3 k vs 62.5 k at 256 cells, 71 k vs 67.8 k at 512, 38 k vs
  // 53 k at 128 -- one workgroup per cell needs a cell for every CU before it beats launches that spread one cell over many
  if (np <= 256) return cells >= 32;
  if (np <= 512) return cells >= 160;
  return np <= 1024 && cells >= 256;
}

// with_alpha = false: the backward substitution is left out -- the caller goes on to the gradient, which forms alpha from the
// explicit inverse it builds anyway (alpha_from_inverse)
int exact_factorize_enqueue(gprx_handle h, int unit, const Theta& t, bool lookahead = true, bool capture = false, bool with_alpha = true) {
  const int np = (int)h->np;
  const int64_t ld = h->np;
  int rc;
  if ((rc = ensure(h, h->Kmat, sizeof(double) * (h->np + NB) * ld))) return rc;
  if ((rc = ensure(h, h->invD, sizeof(double) * h->np * NB))) return rc;
  if ((rc = ensure(h, h->alpha, sizeof(double) * h->np))) return rc;
  if ((rc = ensure(h, h->twork, sizeof(double) * h->np))) return rc;
  if ((rc = ensure(h, h->dstage, sizeof(double) * h->np * STAGE_LD))) return rc;
  if (lookahead && (rc = ensure_lookahead(h))) return rc;
  if (h->Kmat.borrowed && h->arena.p && h->cell_stride > 0) {
    // Kmat / invD / alpha are views into a slot of the last batch (gprx_select_slot): this call overwrites that slot's
    // factorisation, so the slot no longer holds what slot_theta / slot_unit say
    const int64_t slot = (h->Kmat.p - h->arena.p) / h->cell_stride;
    if (slot >= 0 && slot < (int64_t)h->slot_ok.size()) {
      h->slot_ok[slot] = 0;
      h->slot_unit[slot] = -1;
    }
  }
  hipStream_t st = h->stream;
  if (capture) {
    // replayable form: every theta-dependent value travels pinned host -> device inside the graph
    HIPCHK(h, hipMemcpyAsync(h->invls.p, h->pin, sizeof(double) * h->d, hipMemcpyHostToDevice, st));
    HIPCHK(h, hipMemcpyAsync(h->gparams, h->pin + 74, sizeof(double) * 2, hipMemcpyHostToDevice, st));
  } else {
    if ((rc = upload_inv_ls(h, t))) return rc;
    HIPCHK(h, hipEventRecord(h->ev[0], st));
  }
  KmatArgs ka{h->X.p, h->X.p, h->invls.p, h->Kmat.p, ld, (int)h->n, (int)h->n, h->d, np, np, t.variance, t.noise, 1, 1.0,
              capture ? h->gparams : nullptr, 0};
  if (h->profiling && !capture) {
    if (!h->kev[0]) {
      HIPCHK(h, hipEventCreate(&h->kev[0]));
      HIPCHK(h, hipEventCreate(&h->kev[1]));
    }
    HIPCHK(h, hipEventRecord(h->kev[0], st));
  }
  HIPCHK(h, launch_kmat(st, h->kid, with_form(ka, h)));
  if (h->profiling && !capture) {
    HIPCHK(h, hipEventRecord(h->kev[1], st));
    h->kmat_bytes = 8.0 * KM_T * KM_T * (double)(np / KM_T) * (np / KM_T + 1) / 2;  // the lower 64 x 64 tiles
  }
  hipLaunchKernelGGL(set_rhs_rows_kernel, dim3(64), dim3(256), 0, st, h->Kmat.p + (int64_t)np * ld, ld, h->Y.p + (int64_t)unit * h->np,
                     (int)h->n, np, NB);
  if (!capture) HIPCHK(h, hipEventRecord(h->ev[1], st));
  HIPCHK(h, hipMemsetAsync(h->info, 0, sizeof(int), st));
  if (h->profiling) h->prof.reset();
  h->dag_used = false;
  if (use_dag(h->tune, np) && !capture && !h->profiling) {
    HIPCHK(h, potrf_dag(st, h->Kmat.p, ld, np, NB, h->invD.p, h->info, h->dag));
    h->dag_used = true;
  } else {
    HIPCHK(h, potrf_lower(st, h->Kmat.p, ld, np, NB, h->invD.p, h->info, h->dstage.p, h->profiling ? &h->prof : nullptr,
                          lookahead ? &h->pstreams : nullptr, 1, 0, 0, &h->tune));
  }
  if (!capture) HIPCHK(h, hipEventRecord(h->ev[2], st));
  const double* beta = h->Kmat.p + (int64_t)np * ld;
  // (alpha = L^-T beta: beta is copied into a work vector that the solve uses up, alpha receives the solution -- two block steps per launch)
  if (with_alpha) hipLaunchKernelGGL(copy_row_kernel, dim3((np + 255) / 256), dim3(256), 0, st, beta, h->twork.p, np);
  hipLaunchKernelGGL(logdet_quad_kernel, dim3(1), dim3(256), 0, st, (const double*)h->Kmat.p, ld, beta, np, h->red.p, (int64_t)0, 0);
  if (with_alpha) HIPCHK(h, trsv_lower(st, h->Kmat.p, ld, h->invD.p, h->alpha.p, np, true, 1, 0, h->twork.p));
  if (!capture) HIPCHK(h, hipEventRecord(h->ev[3], st));
  HIPCHK(h, hipMemcpyAsync(h->pin + 64, h->red.p, sizeof(double) * 2, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(h->pin + 72, h->info, sizeof(int), hipMemcpyDeviceToHost, st));
  if (h->dag_used) HIPCHK(h, hipMemcpyAsync(h->pin + 73, h->dag.state + DAG_ABORT, sizeof(int), hipMemcpyDeviceToHost, st));
  h->factorized = false;
  h->cur_unit = unit;
  h->variance = t.variance;
  h->noise = t.noise;
  h->ls = t.ls;
  return GPRX_OK;
}

// one stream capture at a time in the process (two concurrent thread-local captures on different handles disturbed each other)
std::mutex& capture_mutex() {
  static std::mutex m;
  return m;
}

// Throughput mode (gprx_factorize_many with several cells): the ~250 launches of one single-stream fit are
// captured once per (handle, unit) into a hipGraph and replayed; only the pinned parameter block changes.
// Measured on MI355X with 16 cells of N = 4096 in flight: 784 fits/s eager, 800 fits/s replayed -- the limit is
// the device (4 hardware queues, each cell ~1.8x slower under 4-way sharing), the replay mainly frees the host.
int exact_factorize_replay(gprx_handle h, int unit, const Theta& t) {
  static const bool no_graph = getenv("GPRX_NO_GRAPH") != nullptr;  // escape hatch: eager launches
  if (h->d > 64 || h->profiling || no_graph) return exact_factorize_enqueue(h, unit, t, false);
  auto it = h->graphs.find(unit);
  if (it == h->graphs.end()) {
    // buffers must exist before capture: a first eager pass allocates them (and is a valid fit by itself)
    if (!h->Kmat.p || !h->invD.p || !h->alpha.p || !h->dstage.p) return exact_factorize_enqueue(h, unit, t, false);
    std::memcpy(h->pin, t.ls.data(), sizeof(double) * h->d);
    h->pin[74] = t.variance;
    h->pin[75] = t.noise;
    std::lock_guard<std::mutex> lock(capture_mutex());  // (captures are serialised over the process, see sgpr_objective_batch)
    hipGraph_t graph = nullptr;
    HIPCHK(h, hipStreamBeginCapture(h->stream, hipStreamCaptureModeRelaxed));
    const int rc = exact_factorize_enqueue(h, unit, t, false, true);
    hipError_t e = hipStreamEndCapture(h->stream, &graph);
    if (rc) {
      if (graph) hipGraphDestroy(graph);
      return rc;
    }
    HIPCHK(h, e);
    hipGraphExec_t exec = nullptr;
    e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
    hipGraphDestroy(graph);
    HIPCHK(h, e);
    it = h->graphs.emplace(unit, exec).first;
  }
  std::memcpy(h->pin, t.ls.data(), sizeof(double) * h->d);
  h->pin[74] = t.variance;
  h->pin[75] = t.noise;
  HIPCHK(h, hipGraphLaunch(it->second, h->stream));
  h->dag_used = false;  // a replayed fit is always the launch-per-panel schedule ("dag" applies to eager single factorisations only)
  h->factorized = false;
  h->cur_unit = unit;
  h->variance = t.variance;
  h->noise = t.noise;
  h->ls = t.ls;
  return GPRX_OK;
}

void summarize_profile(gprx_handle h) {
  double gemm_ms = 0.0, gemm_flops = 0.0, panel_ms = 0.0;
  for (auto& mk : h->prof.gemm_marks) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->prof.pool[mk.first], h->prof.pool[mk.first + 1]);
    gemm_ms += ms;
    gemm_flops += mk.second;
  }
  for (auto idx : h->prof.panel_marks) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->prof.pool[idx], h->prof.pool[idx + 1]);
    panel_ms += ms;
  }
  h->prof_out[0] = gemm_ms;
  h->prof_out[1] = (double)h->prof.gemm_marks.size();
  h->prof_out[2] = gemm_flops;
  h->prof_out[3] = panel_ms;
  h->prof_out[4] = (double)h->prof.panel_marks.size();
  double strip_ms = 0.0, strip_flops = 0.0;
  for (auto& mk : h->prof.strip_marks) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->prof.pool[mk.first], h->prof.pool[mk.first + 1]);
    strip_ms += ms;
    strip_flops += mk.second;
  }
  if (h->kev[0]) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->kev[0], h->kev[1]) == hipSuccess) h->kmat_ms = ms;
  }
  h->cell_ms = 0.0;
  if (h->cev_recorded) {
    float ms = 0.f;
    if (hipEventElapsedTime(&ms, h->cev[0], h->cev[1]) == hipSuccess) h->cell_ms = ms;
    h->cev_recorded = false;
  }
  h->prof_out[5] = strip_ms;
  h->prof_out[6] = (double)h->prof.strip_marks.size();
  h->prof_out[7] = strip_flops;
}

int exact_factorize_finish(gprx_handle h, double* lml_out) {
  HIPCHK(h, wait_stream(h, h->stream));
  const double* red = h->pin + 64;
  int info = 0;
  std::memcpy(&info, h->pin + 72, sizeof(int));
  if (h->profiling) summarize_profile(h);
  if (h->dag_used) {
    int gave_up = 0;
    std::memcpy(&gave_up, h->pin + 73, sizeof(int));
    if (gave_up != 0) {
      h->factorized = false;
      return fail(h, GPRX_EHIP, "tile-DAG factorisation: a dependency wait timed out (scheduler gave up, code " + std::to_string(gave_up) + ")");
    }
  }
  if (info != 0) {
    h->factorized = false;
    char msg[128];
    snprintf(msg, sizeof msg, "matrix not positive definite: pivot %d", info);
    return fail(h, GPRX_ENOTPD, msg);
  }
  h->factorized = true;
  h->have_linv = false;
  if (lml_out) *lml_out = -0.5 * red[1] - red[0] - 0.5 * (double)h->n * PX_LOG_2PI;
  return GPRX_OK;
}

int exact_factorize(gprx_handle h, int unit, const Theta& t, double* lml_out) {
  int rc = exact_factorize_enqueue(h, unit, t);
  if (rc) return rc;
  return exact_factorize_finish(h, lml_out);
}

// ---- batched exact factorisations -------------------------------------------------------------------------
// Independent cells (one unit and one hyperparameter vector each, all on this handle's x) factorised by the
// SAME launches: every kernel of the single-cell schedule carries the cell index in a grid dimension, so one
// panel launch is cells x (rows / 128) workgroups and one trailing update is cells x tiles -- the chip is full
// although a single N = 4096 panel occupies 33 of 256 CUs.  Same kernels, same per-element operation order:
// the results are bit-identical to gprx_factorize on each cell.
constexpr int CELL_RES = 4;  // per cell: [0] sum log diag L, [1] |L^-1 y|^2, [2] info (int), [3] unused

__global__ void set_rhs_rows_batch_kernel(double* dst, int64_t ld, const double* ybase, const double* cell_par, int n, int np, int rows,
                                          int64_t cs) {
  const int cell = blockIdx.y;
  const double* y = ybase + (int64_t)cell_par[(int64_t)cell * CELL_PAR + 2] * np;
  dst += (int64_t)cell * cs;
  const int64_t total = (int64_t)rows * np;
  for (int64_t e = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; e < total; e += (int64_t)gridDim.x * blockDim.x) {
    const int r = (int)(e / np), c = (int)(e % np);
    dst[(int64_t)r * ld + c] = (r == 0 && c < n) ? y[c] : 0.0;
  }
}

__global__ void copy_row_batch_kernel(const double* src, double* dst, int n, int64_t cs) {
  const int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i < n) dst[(int64_t)blockIdx.y * cs + i] = src[(int64_t)blockIdx.y * cs + i];
}

void drop_graphs(gprx_handle h) {
  for (auto& kv : h->graphs) hipGraphExecDestroy(kv.second);
  h->graphs.clear();
  for (auto& kv : h->sgraphs)
    if (kv.second) hipGraphExecDestroy(kv.second);
  h->sgraphs.clear();
}

// views of the single-cell buffers into the arena are invalid once it moves
void drop_arena_views(gprx_handle h) {
  for (Buf* b : {&h->Kmat, &h->invD, &h->alpha})
    if (b->borrowed) {
      b->p = nullptr;
      b->bytes = 0;
      b->borrowed = false;
      h->factorized = false;
      h->have_linv = false;
    }
  drop_graphs(h);
}

int ensure_arena(gprx_handle h, int slots) {
  if (h->arena_slots >= slots) return GPRX_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  drop_arena_views(h);
  const int64_t np = h->np;
  h->off_invd = (np + NB) * np;
  h->off_stage = h->off_invd + np * NB;
  h->off_alpha = h->off_stage + np * STAGE_LD;
  h->cell_stride = round_up(h->off_alpha + np, 64);
  for (Buf* b : {&h->arena, &h->cellpar, &h->cellres}) {
    if (b->p) HIPCHK(h, hipFree(b->p));
    b->p = nullptr;
    b->bytes = 0;
  }
  if (h->bpin) HIPCHK(h, hipHostFree(h->bpin));
  h->bpin = nullptr;
  h->arena_slots = 0;
  int rc;
  if ((rc = ensure(h, h->arena, sizeof(double) * (size_t)h->cell_stride * slots))) return rc;
  if ((rc = ensure(h, h->cellpar, sizeof(double) * CELL_PAR * slots))) return rc;
  if ((rc = ensure(h, h->cellres, sizeof(double) * CELL_RES * slots))) return rc;
  HIPCHK(h, hipHostMalloc((void**)&h->bpin, sizeof(double) * (CELL_PAR + CELL_RES) * slots, hipHostMallocDefault));
  if (!h->bev[0]) {
    HIPCHK(h, hipEventCreate(&h->bev[0]));
    HIPCHK(h, hipEventCreate(&h->bev[1]));
  }
  h->arena_slots = slots;
  h->slot_theta.assign(slots, Theta());
  h->slot_unit.assign(slots, -1);
  h->slot_ok.assign(slots, 0);
  return GPRX_OK;
}

int exact_factorize_batch(gprx_handle h, int count, const int* units, const Theta* ts, double* lml_out, int* status_out, bool with_alpha = true) {
  int rc;
  if ((rc = ensure_arena(h, count))) return rc;
  const int np = (int)h->np;
  const int64_t ld = h->np, cs = h->cell_stride;
  hipStream_t st = h->stream;
  double* par = h->bpin;
  double* res = h->bpin + (size_t)CELL_PAR * h->arena_slots;
  for (int c = 0; c < count; ++c) {
    double* row = par + (size_t)c * CELL_PAR;
    std::memset(row, 0, sizeof(double) * CELL_PAR);
    row[0] = ts[c].variance;
    row[1] = ts[c].noise;
    row[2] = (double)units[c];
    for (int k = 0; k < h->d; ++k) row[CELL_PAR_LS + k] = ts[c].ls[k];
    h->slot_ok[c] = 0;
  }
  // Optional (GPRX_BATCH_GROUPS=2): two groups of cells on two streams run the same launch sequence out of phase, so
  // the panel launches of one overlap the MFMA-bound updates of the other.  Measured at N = 4096: +2.5 % at 32 cells,
  // +3 % at 64, nothing at 16, -5 % at 8 -- not worth a default whose per-launch timings depend on what the other
  // stream happens to run.  Per-cell arithmetic is the same either way.
  static const int env_groups = getenv("GPRX_BATCH_GROUPS") ? atoi(getenv("GPRX_BATCH_GROUPS")) : 0;
  const int groups = (h->profiling || count < 2) ? 1 : (env_groups > 1 ? 2 : 1);
  if (groups > 1 && (rc = ensure_lookahead(h))) return rc;
  HIPCHK(h, hipEventRecord(h->bev[0], st));
  HIPCHK(h, hipMemcpyAsync(h->cellpar.p, par, sizeof(double) * CELL_PAR * count, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemsetAsync(h->cellres.p, 0, sizeof(double) * CELL_RES * count, st));
  if (h->profiling) h->prof.reset();
  auto enqueue_group = [&](hipStream_t gs, int c0, int cnt, hipEvent_t wait_evt = nullptr, hipEvent_t record_evt = nullptr) -> int {
    double* K0 = h->arena.p + (int64_t)c0 * cs;
    const double* cpar = h->cellpar.p + (int64_t)c0 * CELL_PAR;
    double* cres = h->cellres.p + (int64_t)c0 * CELL_RES;
    KmatArgs ka{h->X.p, h->X.p, nullptr, K0, ld, (int)h->n, (int)h->n, h->d, np, np, 0.0, 0.0, 1, 1.0, nullptr, 0};
    ka.cell_par = cpar;
    ka.out_stride = cs;
    // (under profiling the launch sequence is instrumented launch by launch; the cell kernel -- ONE launch -- is timed when the handle
    // forces it, "cell_kernel" = 1: gprx_last_cell_kernel)
    const bool cell_kernel = use_cell_kernel(h->tune, np, cnt) && (!h->profiling || h->tune.cell_kernel > 0);
    // (the column-pair cell kernel evaluates K where it consumes it: no build launch, nothing written but the right-hand-side rows)
    const bool cell_builds_k = cell_kernel && potrf_cells_builds_k(h->kid, h->dist_form, np, h->d);
    if (h->profiling) {
      if (!h->kev[0]) {
        HIPCHK(h, hipEventCreate(&h->kev[0]));
        HIPCHK(h, hipEventCreate(&h->kev[1]));
      }
      HIPCHK(h, hipEventRecord(h->kev[0], gs));
    }
    if (!cell_builds_k) HIPCHK(h, launch_kmat(gs, h->kid, with_form(ka, h), cnt));
    if (h->profiling) {
      HIPCHK(h, hipEventRecord(h->kev[1], gs));
      h->kmat_bytes = 8.0 * KM_T * KM_T * (double)(np / KM_T) * (np / KM_T + 1) / 2 * cnt;
    }
    // (the column-pair cell kernel carries the right-hand side as a vector: one row, of which it reads and writes the first np entries)
    // (so does the launch sequence's split panel: potrf_rows_kernel<..., YVEC>)
    const bool rhs_vector = !cell_kernel && potrf_rhs_vector_ok(h->tune, cnt);
    const bool beta_vector = rhs_vector || (cell_kernel && !cell_builds_k && potrf_cells_beta_vector(np, NB, false));
    hipLaunchKernelGGL(set_rhs_rows_batch_kernel, dim3(beta_vector ? 4 : 64, cnt), dim3(256), 0, gs, K0 + (int64_t)np * ld, ld, (const double*)h->Y.p, cpar,
                       (int)h->n, np, beta_vector ? 1 : NB, cs);
    int* info0 = reinterpret_cast<int*>(cres + 2);
    if (wait_evt) HIPCHK(h, hipStreamWaitEvent(gs, wait_evt, 0));
    if (cell_kernel) {
      // small matrices in many cells: one workgroup owns one cell from the first column to the last (potrf_cell.h)
      if (h->profiling) {
        if (!h->cev[0]) {
          HIPCHK(h, hipEventCreate(&h->cev[0]));
          HIPCHK(h, hipEventCreate(&h->cev[1]));
        }
        HIPCHK(h, hipEventRecord(h->cev[0], gs));
      }
      if (cell_builds_k)
        HIPCHK(h, potrf_cells(gs, K0, ld, np, NB, K0 + h->off_invd, info0, cnt, cs, 2 * CELL_RES, 0, h->X.p, cpar, (int)h->n, h->d));
      else
        HIPCHK(h, potrf_cells(gs, K0, ld, np, NB, K0 + h->off_invd, info0, cnt, cs, 2 * CELL_RES));
      if (h->profiling) {
        HIPCHK(h, hipEventRecord(h->cev[1], gs));
        h->cev_recorded = true;
        h->cell_cells = cnt;
        h->cell_flops = (double)np * np * np / 3.0 * cnt;  // algorithmic: N^3 / 3 per cell (the right-hand-side rows' N^2 not counted)
      }
    } else {
      HIPCHK(h, potrf_lower(gs, K0, ld, np, rhs_vector ? 0 : NB, K0 + h->off_invd, info0, K0 + h->off_stage, h->profiling ? &h->prof : nullptr, nullptr, cnt, cs,
                            2 * CELL_RES, &h->tune, 0, record_evt, rhs_vector ? K0 + (int64_t)np * ld : nullptr));
    }
    const double* beta = K0 + (int64_t)np * ld;
    if (with_alpha) hipLaunchKernelGGL(copy_row_batch_kernel, dim3((np + 255) / 256, cnt), dim3(256), 0, gs, beta, K0 + h->off_alpha, np, cs);
    hipLaunchKernelGGL(logdet_quad_kernel, dim3(cnt), dim3(256), 0, gs, (const double*)K0, ld, beta, np, cres, cs, CELL_RES);
    if (with_alpha) HIPCHK(h, trsv_lower(gs, K0, ld, K0 + h->off_invd, K0 + h->off_alpha, np, true, cnt, cs));  // (else: exact_gradient_batch)
    return GPRX_OK;
  };
  if (groups == 1) {
    if ((rc = enqueue_group(st, 0, count))) return rc;
  } else {
    const int first = (count + 1) / 2;
    hipStream_t aux = h->pstreams.aux;
    HIPCHK(h, hipEventRecord(h->pstreams.block_done, st));  // parameter table and cleared results are on the main stream
    HIPCHK(h, hipStreamWaitEvent(aux, h->pstreams.block_done, 0));
    // GPRX_BATCH_STAGGER=1: the second group's factorisation waits until the first group has left its first in-block phase, so the
    // HBM-bound in-block kernels of one group run beside the MFMA-bound bulk updates of the other
    static const bool stagger = getenv("GPRX_BATCH_STAGGER") && atoi(getenv("GPRX_BATCH_STAGGER")) > 0;
    if (stagger && !h->stagger_evt) HIPCHK(h, hipEventCreateWithFlags(&h->stagger_evt, hipEventDisableTiming));
    if ((rc = enqueue_group(st, 0, first, nullptr, stagger ? h->stagger_evt : nullptr))) return rc;
    if ((rc = enqueue_group(aux, first, count - first, stagger ? h->stagger_evt : nullptr, nullptr))) return rc;
    HIPCHK(h, hipEventRecord(h->pstreams.tail_done, aux));
    HIPCHK(h, hipStreamWaitEvent(st, h->pstreams.tail_done, 0));
  }
  HIPCHK(h, hipMemcpyAsync(res, h->cellres.p, sizeof(double) * CELL_RES * count, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipEventRecord(h->bev[1], st));
  HIPCHK(h, wait_stream(h, st));
  float ms = 0.f;
  hipEventElapsedTime(&ms, h->bev[0], h->bev[1]);
  h->batch_ms = ms;
  if (h->profiling) summarize_profile(h);
  int first_error = GPRX_OK;
  for (int c = 0; c < count; ++c) {
    int info = 0;
    std::memcpy(&info, res + (size_t)c * CELL_RES + 2, sizeof(int));
    h->slot_theta[c] = ts[c];
    h->slot_unit[c] = units[c];
    h->slot_ok[c] = info == 0;
    if (status_out) status_out[c] = info == 0 ? GPRX_OK : GPRX_ENOTPD;
    if (info != 0) {
      if (!first_error) {
        char msg[160];
        snprintf(msg, sizeof msg, "cell %d: matrix not positive definite: pivot %d", c, info);
        first_error = fail(h, GPRX_ENOTPD, msg);
      }
      if (lml_out) lml_out[c] = std::numeric_limits<double>::quiet_NaN();
      continue;
    }
    const double* r = res + (size_t)c * CELL_RES;
    if (lml_out) lml_out[c] = -0.5 * r[1] - r[0] - 0.5 * (double)h->n * PX_LOG_2PI;
  }
  // a single-cell view into a slot of this batch is stale now
  for (Buf* b : {&h->Kmat, &h->invD, &h->alpha})
    if (b->borrowed) {
      h->factorized = false;
      h->have_linv = false;
    }
  return first_error;
}

// make slot `slot` of the last batch the handle's current factorisation (predict / gradient work on it)
int select_slot(gprx_handle h, int slot) {
  if (slot < 0 || slot >= h->arena_slots || h->slot_unit[slot] < 0) return fail(h, GPRX_EINVAL, "slot holds no factorisation");
  if (!h->slot_ok[slot]) return fail(h, GPRX_ESTATE, "the factorisation of this slot failed");
  double* base = h->arena.p + (int64_t)slot * h->cell_stride;
  auto view = [&](Buf& b, double* p, size_t bytes) {
    if (b.p && !b.borrowed) hipFree(b.p);
    b.p = p;
    b.bytes = bytes;
    b.borrowed = true;
  };
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (h->Kmat.p != base) drop_graphs(h);
  view(h->Kmat, base, sizeof(double) * (h->np + NB) * h->np);
  view(h->invD, base + h->off_invd, sizeof(double) * h->np * NB);
  view(h->alpha, base + h->off_alpha, sizeof(double) * h->np);
  const Theta& t = h->slot_theta[slot];
  int rc;
  if ((rc = upload_inv_ls(h, t))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  h->cur_unit = h->slot_unit[slot];
  h->variance = t.variance;
  h->noise = t.noise;
  h->ls = t.ls;
  h->factorized = true;
  h->have_linv = false;
  return GPRX_OK;
}

// K^-1 = X^T X (X = L^-1) as the TN product on the LDS-DMA kernel (default), or GPRX_KINV_TN=0: X transposed in place and the NT
// product of L^-T with itself (one more pass over X: 5.9 ms per 128 cells of N = 4096); the same sums in the same k order.
bool kinv_tn() {
  static const bool v = !(getenv("GPRX_KINV_TN") && atoi(getenv("GPRX_KINV_TN")) == 0);
  return v;
}

// Gradients of the LML for every cell of the batch just factorised (slots 0 .. count-1): the single-cell stages
// (L^-1 by bottom-up doubling, K^-1 = L^-T L^-1 on the lower tiles, one trace pass for all 2 + d derivatives) with
// the cell index in the grid.  g: count x ntheta, constrained parameters (variance, lengthscales, noise); rows of
// failed cells are left untouched.
int exact_gradient_batch(gprx_handle h, int count, double* g, bool form_alpha = false) {
  const int np = (int)h->np;
  const int64_t ld = h->np, cs = h->cell_stride, gs = 2 * (int64_t)h->np * h->np;
  int rc;
  if ((rc = ensure(h, h->garena, sizeof(double) * (size_t)gs * count))) return rc;
  const int tiles = np / KM_T;
  const int width = 2 + h->d;
  const int64_t ps = (int64_t)tiles * tiles * width;  // partials per cell; the count x width sums follow all partials
  if ((rc = ensure(h, h->gpartial, sizeof(double) * (size_t)(ps + width) * count))) return rc;
  hipStream_t st = h->stream;
  double* X0 = h->garena.p;
  double* T0 = h->garena.p + (int64_t)np * ld;
  double* K0 = h->arena.p;
  // (X needs no zeroing: scatter_inv_diag writes the diagonal blocks whole -- zeros above the diagonal included --, every tile
  // below them is written with beta = 0 before it is read, and the triangular K ranges of the products never reach a tile above
  // the diagonal; the 128 memsets of 134 MB were 1.5 % of a batched evaluation.  "poison_workspace" = 1 fills X with NaN patterns
  // instead, for the test that proves it.)
  if (h->tune.poison_workspace)
    for (int c = 0; c < count; ++c) HIPCHK(h, hipMemsetAsync(X0 + (int64_t)c * gs, 0xff, sizeof(double) * h->np * ld, st));
  // 64 x 64 tiles throughout: with many cells per launch they beat the 128 x 128 tiles on these triangular products
  // (measured at 32 cells of N = 4096: 52.8 ms against 60.7 ms per batched objective + gradient)
  const int tile = h->tune.update_tile ? h->tune.update_tile : 64;
  HIPCHK(h, trtri_lower(st, K0, ld, K0 + h->off_invd, X0, ld, T0, ld, np, count, cs, gs, tile));
  if (form_alpha) {  // the factorisation left the backward substitution out: alpha = X^T beta (T is free until the next product)
    if ((rc = ensure(h, h->apart, sizeof(double) * (size_t)count * ((np + ALPHA_CHUNK - 1) / ALPHA_CHUNK) * np))) return rc;
    HIPCHK(h, alpha_from_inverse(st, X0, ld, K0 + (int64_t)np * ld, h->apart.p, K0 + h->off_alpha, np, count, gs, cs, cs));
  }
  // K^-1 = L^-T L^-1 on the lower tiles, as an NT product of Xt = L^-T with itself (transposed in place; same sums in the same
  // k order as the TN form it replaces, so the values are unchanged)
  if (kinv_tn()) {
    HIPCHK(h, launch_gemm(st, 1, 0, np, np, np, 1.0, X0, ld, X0, ld, 0.0, T0, ld, GEMM_C_LOWER | GEMM_A_UPPER | GEMM_B_LOWER, 64, count, gs, gs, gs));
  } else {
    HIPCHK(h, transpose_inplace(st, X0, ld, np, count, gs));
    HIPCHK(h, launch_gemm(st, 0, 1, np, np, np, 1.0, X0, ld, X0, ld, 0.0, T0, ld, GEMM_C_LOWER | GEMM_A_UPPER | GEMM_B_LOWER, tile, count, gs, gs,
                          gs));
  }
  TraceArgs ta{h->X.p, h->X.p, nullptr, T0, ld, K0 + h->off_alpha, K0 + h->off_alpha, -1.0, 1.0, (int)h->n, (int)h->n, h->d, 0.0, 1, h->gpartial.p,
               nullptr, 0, tiles};
  ta.cell_par = h->cellpar.p;
  ta.iso = h->ard ? 0 : 1;
  ta.w_stride = gs;
  ta.uv_stride = cs;
  ta.partial_stride = ps;
  HIPCHK(h, launch_trace(st, h->kid, with_form(ta, h), tiles * tiles, count));
  double* sums0 = h->gpartial.p + ps * count;
  hipLaunchKernelGGL(trace_final, dim3(width, count), dim3(64), 0, st, (const double*)h->gpartial.p, tiles * tiles, width, sums0, ps);
  std::vector<double> host((size_t)width * count);
  HIPCHK(h, hipMemcpyAsync(host.data(), sums0, sizeof(double) * width * count, hipMemcpyDeviceToHost, st));
  HIPCHK(h, wait_stream(h, st));
  for (int c = 0; c < count; ++c) {
    if (!h->slot_ok[c]) continue;
    const double* hs = host.data() + (size_t)c * width;
    double* gc = g + (size_t)c * h->ntheta;
    gc[0] = 0.5 * hs[0];
    if (h->ard) {
      for (int k = 0; k < h->d; ++k) gc[1 + k] = 0.5 * hs[2 + k];
    } else {
      double sum = 0.0;
      for (int k = 0; k < h->d; ++k) sum += hs[2 + k];
      gc[1] = 0.5 * sum;
    }
    gc[1 + h->nlen] = 0.5 * hs[1];
  }
  return GPRX_OK;
}

// gradient of the LML w.r.t. constrained (variance, lengthscales[nlen], noise) -> g[0 .. nlen+1]
// Enqueue the gradient's launches behind the factorisation on the handle's stream; `host` (2 + d doubles, alive until the
// stream has been synchronised) receives the trace sums.  form_alpha: the factorisation left the backward substitution out.
int exact_gradient_enqueue(gprx_handle h, const Theta& t, double* host, bool form_alpha) {
  const int np = (int)h->np;
  const int64_t ld = h->np;
  int rc;
  if ((rc = ensure(h, h->Xinv, sizeof(double) * h->np * ld))) return rc;
  if ((rc = ensure(h, h->Tmp, sizeof(double) * h->np * ld))) return rc;
  hipStream_t st = h->stream;
  if (h->tune.poison_workspace) HIPCHK(h, hipMemsetAsync(h->Xinv.p, 0xff, sizeof(double) * h->np * ld, st));  // (see exact_gradient_batch)
  HIPCHK(h, trtri_lower(st, h->Kmat.p, ld, h->invD.p, h->Xinv.p, ld, h->Tmp.p, ld, np));
  if (form_alpha) {
    if ((rc = ensure(h, h->apart, sizeof(double) * (size_t)((np + ALPHA_CHUNK - 1) / ALPHA_CHUNK) * np))) return rc;
    HIPCHK(h, alpha_from_inverse(st, h->Xinv.p, ld, h->Kmat.p + (int64_t)np * ld, h->apart.p, h->alpha.p, np));
  }
  // K^-1 = X^T X on the lower tiles, into Tmp
  h->have_linv = false;  // (Xinv is not zeroed above its diagonal, or holds L^-T: a later predict forms L^-1 again)
  if (kinv_tn()) {
    HIPCHK(h, launch_gemm(st, 1, 0, np, np, np, 1.0, h->Xinv.p, ld, h->Xinv.p, ld, 0.0, h->Tmp.p, ld, GEMM_C_LOWER | GEMM_A_UPPER | GEMM_B_LOWER, 64));
  } else {
    HIPCHK(h, transpose_inplace(st, h->Xinv.p, ld, np));
    HIPCHK(h, launch_gemm(st, 0, 1, np, np, np, 1.0, h->Xinv.p, ld, h->Xinv.p, ld, 0.0, h->Tmp.p, ld,
                          GEMM_C_LOWER | GEMM_A_UPPER | GEMM_B_LOWER, h->tune.update_tile));
  }
  const int tiles = np / KM_T;
  const int width = 2 + h->d;
  if ((rc = ensure(h, h->partial, sizeof(double) * ((size_t)tiles * tiles * width + width)))) return rc;
  TraceArgs ta{h->X.p, h->X.p, h->invls.p, h->Tmp.p, ld, h->alpha.p, h->alpha.p, -1.0, 1.0, (int)h->n, (int)h->n, h->d, t.variance, 1, h->partial.p, nullptr, 0, tiles};
  ta.iso = h->ard ? 0 : 1;
  HIPCHK(h, launch_trace(st, h->kid, with_form(ta, h), tiles * tiles));
  double* sums = h->partial.p + (size_t)tiles * tiles * width;
  hipLaunchKernelGGL(trace_final, dim3(width), dim3(64), 0, st, h->partial.p, tiles * tiles, width, sums);
  HIPCHK(h, hipMemcpyAsync(host, sums, sizeof(double) * width, hipMemcpyDeviceToHost, st));
  return GPRX_OK;
}
// gradient of the LML w.r.t. constrained (variance, lengthscales[nlen], noise) -> g[0 .. nlen+1], from the synchronised trace sums
void exact_gradient_collect(gprx_handle h, const double* host, double* g) {
  g[0] = 0.5 * host[0];
  if (h->ard) {
    for (int k = 0; k < h->d; ++k) g[1 + k] = 0.5 * host[2 + k];
  } else {
    double s = 0.0;
    for (int k = 0; k < h->d; ++k) s += host[2 + k];
    g[1] = 0.5 * s;
  }
  g[1 + h->nlen] = 0.5 * host[1];
}
int exact_gradient(gprx_handle h, const Theta& t, double* g) {
  std::vector<double> host(2 + h->d);
  int rc;
  if ((rc = exact_gradient_enqueue(h, t, host.data(), false))) return rc;
  HIPCHK(h, wait_stream(h, h->stream));
  exact_gradient_collect(h, host.data(), g);
  return GPRX_OK;
}

// ---- sparse GP (SGPR) -----------------------------------------------------------------------------
// Device restatement of gpflow SGPR._common_calculation / elbo / predict_f (oracle/sgpr.py) with
//   P = Kuf (mp x np), Q = Kuu + jitter I -> L, A' = L^-1 P (unscaled: A = A' / sqrt(s)),
//   B = I + A' A'^T / s -> LB, c = LB^-1 A' y / s (carried through the Cholesky as an appended row).
// SM holds nine mp x mp scratch matrices.
constexpr int SPLITK_CHUNK = 256;
constexpr int SGPR_PRED_TILE = 4096;  // test points per pass of the batched sparse predict
enum { SM_BFULL = 0, SM_LINV, SM_LBINV, SM_QINV, SM_SINV, SM_R, SM_T1, SM_T2, SM_W, SM_GQ, SM_COUNT };

double* sm(gprx_handle h, int slot) { return h->SM.p + (size_t)slot * h->mp * h->mp; }

int sgpr_alloc(gprx_handle h) {
  const size_t mp = h->mp, np = h->np;
  int rc;
  if ((rc = ensure(h, h->Z, sizeof(double) * h->m * h->d))) return rc;
  if ((rc = ensure(h, h->P, sizeof(double) * mp * np))) return rc;
  if ((rc = ensure(h, h->Am, sizeof(double) * mp * np))) return rc;
  if ((rc = ensure(h, h->Qm, sizeof(double) * mp * mp))) return rc;
  if ((rc = ensure(h, h->Bm, sizeof(double) * (mp + NB) * mp))) return rc;
  if ((rc = ensure(h, h->invDL, sizeof(double) * mp * NB))) return rc;
  if ((rc = ensure(h, h->invDB, sizeof(double) * mp * NB))) return rc;
  if ((rc = ensure_zeroed(h, h->vecs, sizeof(double) * (4 * mp + np)))) return rc;
  if ((rc = ensure(h, h->dstage, sizeof(double) * std::max(mp, np) * STAGE_LD))) return rc;
  // split-K slabs: K = np in slices of SPLITK_CHUNK, outputs up to mp x mp
  if ((rc = ensure(h, h->splitws, sizeof(double) * ((np + SPLITK_CHUNK - 1) / SPLITK_CHUNK) * mp * mp))) return rc;
  return GPRX_OK;
}

int sgpr_factorize(gprx_handle h, int unit, const Theta& t, const double* z, double* elbo_out) {
  if (!z) return fail(h, GPRX_EINVAL, "z (inducing inputs) is null for a sparse model");
  for (int64_t e = 0; e < h->m * h->d; ++e)
    if (!std::isfinite(z[e])) return fail(h, GPRX_EINVAL, "z is not finite");
  int rc;
  if ((rc = sgpr_alloc(h))) return rc;
  const int mp = (int)h->mp, np = (int)h->np, m = (int)h->m, n = (int)h->n;
  hipStream_t st = h->stream;
  HIPCHK(h, hipMemcpyAsync(h->Z.p, z, sizeof(double) * h->m * h->d, hipMemcpyHostToDevice, st));
  if ((rc = upload_inv_ls(h, t))) return rc;
  const double s = t.noise;
  HIPCHK(h, hipEventRecord(h->ev[0], st));
  KmatArgs kp{h->Z.p, h->X.p, h->invls.p, h->P.p, np, m, n, h->d, mp, np, t.variance, 0.0, 0, 0.0, nullptr, 0};
  HIPCHK(h, launch_kmat(st, h->kid, with_form(kp, h)));
  KmatArgs kq{h->Z.p, h->Z.p, h->invls.p, h->Qm.p, mp, m, m, h->d, mp, mp, t.variance, JITTER, 2, 1.0, nullptr, 0};
  HIPCHK(h, launch_kmat(st, h->kid, with_form(kq, h)));
  HIPCHK(h, hipEventRecord(h->ev[1], st));
  HIPCHK(h, hipMemsetAsync(h->info, 0, sizeof(int), st));
  HIPCHK(h, potrf_lower(st, h->Qm.p, mp, mp, 0, h->invDL.p, h->info, h->dstage.p, nullptr, nullptr, 1, 0, 0, &h->tune));
  HIPCHK(h, hipMemcpyAsync(h->Am.p, h->P.p, sizeof(double) * (size_t)mp * np, hipMemcpyDeviceToDevice, st));
  HIPCHK(h, trsm_lower_left(st, h->Qm.p, mp, h->invDL.p, h->Am.p, np, mp, np));
  // B = I + A' A'^T / s (all of it: the gradient needs the symmetric matrix)
  if (mp <= 512 && np >= 4 * SPLITK_CHUNK) {
    // (mp/64)^2 output tiles against K = np: cut K over workgroups, reduce the slabs in a fixed order
    HIPCHK(h, launch_gemm_splitk(st, 0, 1, mp, mp, np, 1.0 / s, h->Am.p, np, h->Am.p, np, 0.0, h->Bm.p, mp, h->splitws.p, SPLITK_CHUNK));
  } else {
    HIPCHK(h, launch_gemm(st, 0, 1, mp, mp, np, 1.0 / s, h->Am.p, np, h->Am.p, np, 0.0, h->Bm.p, mp, 0));
  }
  hipLaunchKernelGGL(add_diag_kernel, dim3((mp + 255) / 256), dim3(256), 0, st, h->Bm.p, (int64_t)mp, mp, 1.0, (int64_t)0);
  hipLaunchKernelGGL(diag_sum_kernel, dim3(1), dim3(256), 0, st, (const double*)h->Bm.p, (int64_t)mp, mp, 1.0, h->red.p + 2, (int64_t)0);
  if ((rc = ensure(h, h->SM, sizeof(double) * (size_t)SM_COUNT * mp * mp))) return rc;
  HIPCHK(h, hipMemcpyAsync(sm(h, SM_BFULL), h->Bm.p, sizeof(double) * (size_t)mp * mp, hipMemcpyDeviceToDevice, st));
  // appended row: A' y / s  -> comes out of the Cholesky as c
  double* crow = h->Bm.p + (size_t)mp * mp;
  HIPCHK(h, hipMemsetAsync(crow, 0, sizeof(double) * (size_t)NB * mp, st));
  const double* yu = h->Y.p + (size_t)unit * h->np;
  if (np >= 4 * SPLITK_CHUNK) {
    HIPCHK(h, launch_gemm_splitk(st, 0, 0, mp, 1, np, 1.0 / s, h->Am.p, np, yu, 1, 0.0, crow, 1, h->splitws.p, SPLITK_CHUNK));
  } else {
    HIPCHK(h, launch_gemm(st, 0, 0, mp, 1, np, 1.0 / s, h->Am.p, np, yu, 1, 0.0, crow, 1, 0, 64));
  }
  HIPCHK(h, potrf_lower(st, h->Bm.p, mp, mp, NB, h->invDB.p, h->info, h->dstage.p, nullptr, nullptr, 1, 0, 0, &h->tune));
  HIPCHK(h, hipEventRecord(h->ev[2], st));
  hipLaunchKernelGGL(logdet_quad_kernel, dim3(1), dim3(256), 0, st, (const double*)h->Bm.p, (int64_t)mp, (const double*)crow, mp, h->red.p, (int64_t)0, 0);
  HIPCHK(h, hipEventRecord(h->ev[3], st));
  double red[3];
  int info = 0;
  HIPCHK(h, hipMemcpyAsync(red, h->red.p, sizeof(red), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&info, h->info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(h, wait_stream(h, st));
  if (info != 0) {
    h->factorized = false;
    char msg[128];
    snprintf(msg, sizeof msg, "Kuu or B not positive definite: pivot %d", info);
    return fail(h, GPRX_ENOTPD, msg);
  }
  h->factorized = true;
  h->sparse_view = false;
  h->cur_unit = unit;
  h->variance = t.variance;
  h->noise = s;
  h->ls = t.ls;
  h->elbo_trAAT = red[2];
  const double nn = (double)h->n;
  if (elbo_out)
    *elbo_out = sgpr_asm_elbo(nn, h->yy[unit], t.variance, s, red);
  return GPRX_OK;
}

// derivatives of the ELBO w.r.t. constrained (variance, lengthscales[nlen], noise) -> g, and Z -> gz (m x d, host)
int sgpr_gradient(gprx_handle h, int unit, const Theta& t, double* g, double* gz) {
  const int mp = (int)h->mp, np = (int)h->np, m = (int)h->m, n = (int)h->n, d = h->d;
  const size_t mm = (size_t)mp * mp;
  const double s = t.noise;
  hipStream_t st = h->stream;
  int rc;
  if ((rc = ensure(h, h->WP, sizeof(double) * (size_t)mp * np))) return rc;
  if ((rc = ensure_zeroed(h, h->WHP, sizeof(double) * (size_t)mp * np))) return rc;
  if ((rc = ensure_zeroed(h, h->WHQ, sizeof(double) * mm))) return rc;
  if ((rc = ensure(h, h->dZ, sizeof(double) * (size_t)m * d))) return rc;
  const int tiles_m = mp / KM_T, tiles_n = np / KM_T, width = 2 + d;
  const size_t part_p = (size_t)tiles_m * tiles_n * width, part_q = (size_t)tiles_m * tiles_m * width;
  if ((rc = ensure(h, h->partial, sizeof(double) * (part_p + part_q + 2 * width)))) return rc;
  double *Linv = sm(h, SM_LINV), *LBinv = sm(h, SM_LBINV), *Qinv = sm(h, SM_QINV), *Sinv = sm(h, SM_SINV), *R = sm(h, SM_R),
         *T1 = sm(h, SM_T1), *T2 = sm(h, SM_T2), *W = sm(h, SM_W), *GQ = sm(h, SM_GQ), *Bfull = sm(h, SM_BFULL);
  double* cvec = h->Bm.p + mm;  // row mp of Bm
  double* mvec = h->vecs.p;
  double* qvec = h->vecs.p + 4 * mp;
  const double* yu = h->Y.p + (size_t)unit * h->np;

  HIPCHK(h, hipMemsetAsync(Linv, 0, sizeof(double) * mm, st));
  HIPCHK(h, trtri_lower(st, h->Qm.p, mp, h->invDL.p, Linv, mp, T1, mp, mp));
  HIPCHK(h, hipMemsetAsync(LBinv, 0, sizeof(double) * mm, st));
  HIPCHK(h, trtri_lower(st, h->Bm.p, mp, h->invDB.p, LBinv, mp, T1, mp, mp));
  HIPCHK(h, launch_gemm(st, 1, 0, mp, mp, mp, 1.0, Linv, mp, Linv, mp, 0.0, Qinv, mp, GEMM_A_UPPER | GEMM_B_LOWER));
  HIPCHK(h, launch_gemm(st, 0, 0, mp, mp, mp, 1.0, LBinv, mp, Linv, mp, 0.0, R, mp, GEMM_A_LOWER | GEMM_B_LOWER));
  HIPCHK(h, launch_gemm(st, 1, 0, mp, mp, mp, 1.0, R, mp, R, mp, 0.0, Sinv, mp, GEMM_A_UPPER | GEMM_B_LOWER));
  HIPCHK(h, launch_gemm(st, 0, 0, mp, mp, mp, 1.0, Bfull, mp, Linv, mp, 0.0, T2, mp, GEMM_B_LOWER));
  HIPCHK(h, launch_gemm(st, 1, 0, mp, mp, mp, 1.0, Linv, mp, T2, mp, 0.0, T1, mp, GEMM_A_UPPER));
  // m = L^-T LB^-T c
  hipLaunchKernelGGL(copy_row_kernel, dim3((mp + 255) / 256), dim3(256), 0, st, (const double*)cvec, mvec, mp);
  HIPCHK(h, trsv_lower(st, h->Bm.p, mp, h->invDB.p, mvec, mp, true));
  HIPCHK(h, trsv_lower(st, h->Qm.p, mp, h->invDL.p, mvec, mp, true));
  hipLaunchKernelGGL(sgpr_combine_kernel, dim3((mp * mp + 255) / 256), dim3(256), 0, st, (const double*)Qinv, (const double*)Sinv,
                     (const double*)T1, (const double*)mvec, mp, W, GQ, (int64_t)0);
  HIPCHK(h, launch_gemm(st, 0, 0, mp, np, mp, 1.0, W, mp, h->P.p, np, 0.0, h->WP.p, np, 0));
  // contractions with the kernel derivatives
  double* partP = h->partial.p;
  double* partQ = partP + part_p;
  double* sums = partQ + part_q;
  // (one shared lengthscale -- the reference's default kernels: the per-dimension pass collapses to one FMA per element, round 4 for the
  // sparse model too: sum_k ds_k^2 is the r2 of the first pass whatever d is and whether or not w v h is stored for dz_kernel)
  TraceArgs tp{h->Z.p, h->X.p, h->invls.p, h->WP.p, np, mvec, yu, 1.0 / s, 1.0 / s, m, n, d, t.variance, 0, partP, h->WHP.p, np, tiles_n};
  tp.iso = h->ard ? 0 : 1;
  HIPCHK(h, launch_trace(st, h->kid, with_form(tp, h), tiles_m * tiles_n));
  TraceArgs tq{h->Z.p, h->Z.p, h->invls.p, GQ, mp, nullptr, nullptr, 1.0, 0.0, m, m, d, t.variance, 0, partQ, h->WHQ.p, mp, tiles_m};
  tq.iso = h->ard ? 0 : 1;
  HIPCHK(h, launch_trace(st, h->kid, with_form(tq, h), tiles_m * tiles_m));
  hipLaunchKernelGGL(trace_final, dim3(width), dim3(64), 0, st, (const double*)partP, tiles_m * tiles_n, width, sums);
  hipLaunchKernelGGL(trace_final, dim3(width), dim3(64), 0, st, (const double*)partQ, tiles_m * tiles_m, width, sums + width);
  hipLaunchKernelGGL(dz_kernel, dim3(dz_grid(m, d)), dim3(256), 0, st, (const double*)h->Z.p, (const double*)h->X.p, (const double*)h->WHP.p,
                     (int64_t)np, (const double*)h->WHQ.p, (int64_t)mp, (const double*)h->invls.p, m, n, d, h->dZ.p);
  // noise terms: |y - P^T m|^2 and tr(B^-1) = |LB^-1|_F^2
  HIPCHK(h, launch_gemm(st, 1, 0, np, 1, mp, 1.0, h->P.p, np, mvec, 1, 0.0, qvec, 1, 0, 64));
  hipLaunchKernelGGL(resid_sumsq_kernel, dim3(1), dim3(256), 0, st, yu, (const double*)qvec, n, h->red.p + 4, (int64_t)0, (int64_t)0);
  {
    const int nb = mp < 64 ? mp : 64;
    double* part = h->vecs.p + 2 * mp;  // scratch inside the vector block
    hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb), dim3(256), 0, st, (const double*)LBinv, (int64_t)mp, mp, mp, part, (int64_t)0);
    hipLaunchKernelGGL(sum_partials_kernel, dim3(1), dim3(64), 0, st, (const double*)part, nb, h->red.p + 3, (int64_t)0);
  }
  std::vector<double> hs(2 * width), hz((size_t)m * d);
  double red[5];
  HIPCHK(h, hipMemcpyAsync(hs.data(), sums, sizeof(double) * 2 * width, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(red, h->red.p, sizeof(red), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(hz.data(), h->dZ.p, sizeof(double) * m * d, hipMemcpyDeviceToHost, st));
  HIPCHK(h, wait_stream(h, st));
  const double nn = (double)h->n;
  const double tr_sinv_pp = s * ((double)mp - red[3]);
  const double tr_qinv_pp = s * h->elbo_trAAT;
  g[0] = -nn / (2.0 * s) + hs[0] + hs[width];
  if (h->ard) {
    for (int k = 0; k < d; ++k) g[1 + k] = hs[2 + k] + hs[width + 2 + k];
  } else {
    double acc = 0.0;
    for (int k = 0; k < d; ++k) acc += hs[2 + k] + hs[width + 2 + k];
    g[1] = acc;
  }
  g[1 + h->nlen] = (tr_sinv_pp - tr_qinv_pp + red[4] + nn * t.variance) / (2.0 * s * s) - nn / (2.0 * s);
  if (gz) std::memcpy(gz, hz.data(), sizeof(double) * m * d);
  return GPRX_OK;
}

// ---- batched sparse models ------------------------------------------------------------------------------------
// The reference fits its per-mode SGPR models one after the other (gpr.py:272-274); every evaluation is ~45 tiny
// dependent launches (M = 50 inducing points: every M x M matrix is one 64 x 64 tile), i.e. pure launch latency.
// Here `count` cells (unit, theta, Z) on the handle's x go through the SAME launch sequence, the cell index in a grid
// dimension of every kernel: each cell owns one block of `ss` doubles holding all its matrices at fixed offsets, so
// a kernel adds blockIdx * ss to its per-cell pointers; hyperparameters (and 1 / s for the GEMM scalings) come from the
// cell-parameter table.  Same kernels, same per-element operation order as sgpr_factorize / sgpr_gradient:
// bit-identical values.
struct SgprLayout {
  int64_t oZ, oY, oP, oAm, oQm, oBm, oInvDL, oInvDB, oSM, oWP, oWHP, oWHQ, oVecs, odZ, oStage, oPart, oWs, oRed, oKs, oPred, ss;
  int64_t oFU, oFP2;  // fused evaluation (sgpr_fused.h): u partials of the chunks, pass-2 partial blocks
  int64_t part_p, part_q;
  int width, nsplit, p2w;
};

SgprLayout sgpr_batch_layout(gprx_handle h) {
  const int64_t mp = h->mp, np = h->np, m = h->m, d = h->d;
  SgprLayout L{};
  L.width = 2 + (int)d;
  L.nsplit = (int)((np + SPLITK_CHUNK - 1) / SPLITK_CHUNK);
  L.part_p = (mp / KM_T) * (np / KM_T) * L.width;
  L.part_q = (mp / KM_T) * (mp / KM_T) * L.width;
  int64_t o = 0;
  auto take = [&](int64_t doubles) {
    const int64_t at = o;
    o += round_up(doubles, 64);
    return at;
  };
  // (the five-launch evaluation of sgpr_fused.h never stores Kuf, A', W Kuf or the weighted derivative: the four M x N matrices and the
  // trace partials of the launch sequence shrink to nothing -- 8.4 of 9.9 MB per cell at M = 50, N = 4096, which a fit allocated and cleared)
  const bool fused = mp == NB && h->sgpr_fused != 0;
  const int64_t big = fused ? 0 : mp * np;
  L.oZ = take(m * d);
  L.oY = take(np);
  L.oP = take(big);
  L.oAm = take(big);
  L.oQm = take(mp * mp);
  L.oBm = take((mp + NB) * mp);
  L.oInvDL = take(mp * NB);
  L.oInvDB = take(mp * NB);
  L.oSM = take((int64_t)SM_COUNT * mp * mp);
  L.oWP = take(big);
  L.oWHP = take(big);
  L.oWHQ = take(mp * mp);
  L.oVecs = take(4 * mp + np);
  L.odZ = take(m * d);
  L.oStage = take(mp * STAGE_LD);
  L.oPart = take(fused ? 0 : L.part_p + L.part_q + 2 * L.width);
  L.oWs = take((int64_t)L.nsplit * mp * mp);
  L.oRed = take(8);
  L.oKs = take(mp * SGPR_PRED_TILE);                                           // batched predict: Kus tile of this cell
  L.oPred = take(((mp + 255) / 256) * (int64_t)SGPR_PRED_TILE);                // its column-reduction partials
  L.p2w = (int)round_up(SF_P2_HEAD + NB * d, 2);
  L.oFU = take((int64_t)L.nsplit * NB);
  L.oFP2 = take((int64_t)(L.nsplit + 1) * L.p2w);
  L.ss = o;
  return L;
}

int ensure_sarena(gprx_handle h, int slots, const SgprLayout& L) {
  if (h->sarena_slots >= slots) return GPRX_OK;
  HIPCHK(h, hipStreamSynchronize(h->stream));
  drop_graphs(h);  // captured evaluations hold the addresses of the buffers released below
  if (h->sparse_view) {  // the single-model factorisation lived in cell block 0 of the arena released below
    h->sparse_view = false;
    h->factorized = false;
  }
  if (h->sarena.p) HIPCHK(h, hipFree(h->sarena.p));
  h->sarena.p = nullptr;
  h->sarena.bytes = 0;
  h->sarena_slots = 0;
  int rc;
  if ((rc = ensure(h, h->sarena, sizeof(double) * (size_t)L.ss * slots))) return rc;
  HIPCHK(h, hipMemsetAsync(h->sarena.p, 0, sizeof(double) * (size_t)L.ss * slots, h->stream));  // padding of every matrix stays zero
  if ((rc = ensure(h, h->cellpar, sizeof(double) * CELL_PAR * slots))) return rc;
  if ((rc = ensure(h, h->cellres, sizeof(double) * CELL_RES * slots))) return rc;
  const size_t need = (size_t)slots * (CELL_PAR + CELL_RES + 8 + 2 * L.width + 2 * h->m * h->d);
  if (h->spin_doubles < need) {
    if (h->spin) HIPCHK(h, hipHostFree(h->spin));
    h->spin = nullptr;
    HIPCHK(h, hipHostMalloc((void**)&h->spin, sizeof(double) * need, hipHostMallocDefault));
    h->spin_doubles = need;
  }
  h->sarena_slots = slots;
  return GPRX_OK;
}

// Pinned staging block of the sparse batch (h->spin), offsets in doubles for `count` cells.
struct SgprStage {
  size_t par, res, red, sum, dz, z;
};
SgprStage sgpr_stage(gprx_handle h, int count, const SgprLayout& L) {
  SgprStage s{};
  s.par = 0;
  s.res = s.par + (size_t)count * CELL_PAR;
  s.red = s.res + (size_t)count * CELL_RES;
  s.sum = s.red + (size_t)count * 8;
  s.dz = s.sum + (size_t)count * 2 * L.width;
  s.z = s.dz + (size_t)count * h->m * h->d;
  return s;
}

// M <= 64: the five launches of sgpr_fused.h (prep, pass 1, mid, pass 2, final) instead of the 21 below; same staging block, same host
// tail.  "sgpr_fused" = 0 (gprx_set_tuning) keeps the launch sequence -- which larger M always takes.
int& sgpr_fused_tuning() {
  static int v = 1;
  return v;
}
static_assert(SF_CHUNK == SPLITK_CHUNK, "the fused evaluation stores its slabs in the split-K workspace of the cell block");

SfParams sgpr_fused_params(gprx_handle h, const SgprLayout& L, bool want_grad) {
  const int64_t mm = (int64_t)h->mp * h->mp;
  SfParams p{};
  p.X = h->X.p;
  p.Y = h->Y.p;
  p.arena = h->sarena.p;
  p.ss = L.ss;
  p.cpar = h->cellpar.p;
  p.n = (int)h->n;
  p.np = (int)h->np;
  p.m = (int)h->m;
  p.d = h->d;
  p.nchunks = L.nsplit;
  p.oZ = L.oZ;
  p.oL = L.oQm;
  p.oLinv = L.oInvDL;
  p.oLB = L.oBm;
  p.oLBinv = L.oInvDB;
  p.oW = L.oSM + SM_W * mm;
  p.oGQ = L.oSM + SM_GQ * mm;
  p.oM = L.oVecs;
  p.oSlab = L.oWs;
  p.oU = L.oFU;
  p.oP2 = L.oFP2;
  p.oRed = L.oRed;
  p.p2w = L.p2w;
  p.cellres = h->cellres.p;
  p.cellres_stride = CELL_RES;
  p.want_grad = want_grad ? 1 : 0;
  p.store_factors = 1;
  p.stamps = h->sf_stamps;
  return p;
}

// The resident Adam loop on large batches: TWO groups of cells on two streams.  Two of the four launches of a step (mid, Adam + prep) are
// one workgroup per cell around a 64 x 64 chain: with all cells in lock step the chip idles through them (16 of 256 CUs busy for half of a
// 16-cell step), and a pass over more than 16 cells takes a second round of 256 workgroups.  Two groups that start one launch apart keep
// that distance: one group's single-workgroup launches run beside the other's streamed passes.  Measured per lock-step step (N = 4096,
// d = 10, M = 50): 17 cells 183 -> 128 us, 28 cells 187 -> 131, 36 cells 244 -> 205, 50 cells 303 -> 223; 32 cells 188 -> 187 (a pass workgroup fills its CU --
// 512 threads x 256 registers -- and a group of 16 cells occupies all 256: the other group's single workgroups find no CU until the
// round ends); three and more groups LOSE (24 cells as three groups 252 us, 32 as three 258, 50 as five 307: streams beyond the second
// do not run beside the first two on this runtime).  Every cell's arithmetic is untouched: same bits (tools/sgpr_groups_probe.py).
// From `sf_groups_from()` cells on at 16 chunks per cell ("sgpr_groups_from", 0: never; sf_group_count).  Host-driven evaluations stay one group: the cross-stream edges cost a
// single call more than the overlap returns (16 cells 154 -> 210 us per call, 50 cells 360 -> 343).
int& sf_groups_from() {
  static int v = [] {
    const char* e = getenv("GPRX_SF_GROUPS_FROM");
    return e ? atoi(e) : 17;
  }();
  return v;
}
// (the threshold is stated in cells at N = 4096, i.e. 16 chunks of 256 columns per cell; what counts is whether a pass -- cells x chunks
// workgroups -- needs more than one round of the 256 CUs: N = 8192 splits from 9 cells on, N = 2048 from 33)
int sf_group_count(int count, int nchunks) {
  if (sf_groups_from() <= 0 || count < 2) return 1;
  return (int64_t)count * nchunks > (int64_t)(sf_groups_from() - 1) * 16 ? 2 : 1;
}
int sf_group_streams(gprx_handle h, int ngroups) {
  for (int g = 0; g + 1 < ngroups; ++g)
    if (!h->sf_streams[g]) HIPCHK(h, hipStreamCreateWithFlags(&h->sf_streams[g], hipStreamNonBlocking));
  for (int g = 0; g < ngroups; ++g)
    if (!h->sf_evs[g]) HIPCHK(h, hipEventCreateWithFlags(&h->sf_evs[g], hipEventDisableTiming));
  return GPRX_OK;
}
// the parameter block of the cells [cell0, ...) of a batch: every per-cell base pointer moved (the kernels index cells from 0)
SfParams sf_params_from(SfParams p, int cell0) {
  p.arena += (int64_t)cell0 * p.ss;
  p.cpar += (int64_t)cell0 * CELL_PAR;
  p.cellres += (int64_t)cell0 * p.cellres_stride;
  if (p.active) p.active += cell0;
  if (cell0 != 0) p.stamps = nullptr;
  return p;
}

int sgpr_fused_enqueue(gprx_handle h, int count, const SgprLayout& L, bool want_grad) {
  hipStream_t st = h->stream;
  const SgprStage sg = sgpr_stage(h, count, L);
  const SfParams p = sgpr_fused_params(h, L, want_grad);
  const int iso = (h->ard || h->dist_form) ? 0 : 1;
  HIPCHK(h, sf_launch_prep(st, h->kid, h->dist_form, p, count, h->spin + sg.par, h->spin + sg.z, h->cellpar.p));
  HIPCHK(h, sf_launch_pass1(st, h->kid, h->dist_form, p, count));
  HIPCHK(h, sf_launch_mid(st, p, count));
  if (want_grad) HIPCHK(h, sf_launch_pass2(st, h->kid, h->dist_form, iso, p, count));
  HIPCHK(h, sf_launch_final(st, iso, p, count, h->spin + sg.res, h->spin + sg.red, h->spin + sg.sum, h->spin + sg.dz));
  return GPRX_OK;
}

// Device part of one batched evaluation: everything between the staged inputs (parameter table and Z in pinned memory) and
// the staged outputs (pivot status, reductions, trace sums, dZ in pinned memory).  Nothing here depends on the VALUES of the
// parameters -- they travel through the cell-parameter table -- so the sequence is captured once per (cells, gradient) into a
// hipGraph and replayed (sgpr_objective_batch): ~45 launches whose enqueue cost, not their device time, bounded a step.
int sgpr_batch_enqueue(gprx_handle h, int count, const SgprLayout& L, bool want_grad) {
  const int mp = (int)h->mp, np = (int)h->np, m = (int)h->m, n = (int)h->n, d = h->d;
  const int64_t ss = L.ss, mm = (int64_t)mp * mp;
  const size_t pitch = sizeof(double) * (size_t)ss;
  hipStream_t st = h->stream;
  double* A0 = h->sarena.p;
  if (mp == NB && h->sgpr_fused) return sgpr_fused_enqueue(h, count, L, want_grad);
  const SgprStage sg = sgpr_stage(h, count, L);
  // (Tried: the independent branches of the evaluation -- Kuf beside Kuu's factorisation; R, Sinv / T2, T1 / Qinv, m; the
  // two contractions and the noise terms -- on side streams, i.e. parallel branches of the captured graph.  The dependent chain
  // drops from 34 to 20 launches, but every cross-branch edge costs more than an in-order kernel boundary on this runtime:
  // 16 cells 0.427 ms against 0.400 ms serial.  One stream it is.)
  const double* cpar = h->cellpar.p;
  const double* inv_s = cpar + 3;  // alpha table: 1 / s, CELL_PAR apart
  static_assert(CELL_RES <= 256 && CELL_PAR <= 256, "sgpr_stage_in_kernel moves them with its first workgroup");
  hipLaunchKernelGGL(sgpr_stage_in_kernel, dim3((std::max(np, m * d) + 255) / 256, count), dim3(256), 0, st, (const double*)h->Y.p, np,
                     (const double*)(h->spin + sg.par), CELL_PAR, h->cellpar.p, (const double*)(h->spin + sg.z), m * d, A0 + L.oZ, A0 + L.oY, ss,
                     h->cellres.p, CELL_RES);
  // ---- factorisation (sgpr_factorize) ----
  KmatArgs kp{A0 + L.oZ, h->X.p, nullptr, A0 + L.oP, np, m, n, d, mp, np, 0.0, 0.0, 0, 0.0, nullptr, 0};
  kp.cell_par = cpar;
  kp.out_stride = ss;
  kp.a_stride = ss;
  kp.diag_const = 1;
  KmatArgs kq{A0 + L.oZ, A0 + L.oZ, nullptr, A0 + L.oQm, mp, m, m, d, mp, mp, 0.0, JITTER, 2, 1.0, nullptr, 0};
  kq.cell_par = cpar;
  kq.out_stride = ss;
  kq.a_stride = ss;
  kq.b_stride = ss;
  kq.diag_const = 1;
  HIPCHK(h, launch_kmat_pair(st, h->kid, with_form(kp, h), with_form(kq, h), count));  // Kuf and Kuu in one launch
  int* info0 = reinterpret_cast<int*>(h->cellres.p + 2);
  HIPCHK(h, potrf_lower(st, A0 + L.oQm, mp, mp, 0, A0 + L.oInvDL, info0, A0 + L.oStage, nullptr, nullptr, count, ss, 2 * CELL_RES, &h->tune));
  const bool one_block = mp == NB;  // M <= 64 (the reference's default is 50): every M x M matrix is one 64 x 64 tile
  if (one_block) {
    HIPCHK(h, trsm_lower_left(st, A0 + L.oQm, mp, A0 + L.oInvDL, A0 + L.oAm, np, mp, np, count, ss, A0 + L.oP));  // A = L^-1 P straight from P
  } else {
    HIPCHK(h, hipMemcpy2DAsync(A0 + L.oAm, pitch, A0 + L.oP, pitch, sizeof(double) * (size_t)mp * np, count, hipMemcpyDeviceToDevice, st));
    HIPCHK(h, trsm_lower_left(st, A0 + L.oQm, mp, A0 + L.oInvDL, A0 + L.oAm, np, mp, np, count, ss));
  }
  if (mp <= 512 && np >= 4 * SPLITK_CHUNK) {
    HIPCHK(h, launch_gemm_splitk(st, 0, 1, mp, mp, np, 0.0, A0 + L.oAm, np, A0 + L.oAm, np, 0.0, A0 + L.oBm, mp, A0 + L.oWs, SPLITK_CHUNK, count, ss, ss,
                                 ss, ss, inv_s, CELL_PAR));
  } else {
    HIPCHK(h, launch_gemm(st, 0, 1, mp, mp, np, 0.0, A0 + L.oAm, np, A0 + L.oAm, np, 0.0, A0 + L.oBm, mp, 0, 0, 1, 0, 0, 0, count, ss, ss, ss, inv_s,
                          CELL_PAR));
  }
  double* SM0 = A0 + L.oSM;
  auto smb = [&](int slot) { return SM0 + (size_t)slot * mm; };
  double* crow = A0 + L.oBm + mm;
  if (mp <= 128) {
    hipLaunchKernelGGL(sgpr_b_finish_kernel, dim3(count), dim3(256), 0, st, A0 + L.oBm, mp, A0 + L.oRed + 2, smb(SM_BFULL), ss);
  } else {
    hipLaunchKernelGGL(add_diag_kernel, dim3((mp + 255) / 256, count), dim3(256), 0, st, A0 + L.oBm, (int64_t)mp, mp, 1.0, ss);
    hipLaunchKernelGGL(diag_sum_kernel, dim3(1, count), dim3(256), 0, st, (const double*)(A0 + L.oBm), (int64_t)mp, mp, 1.0, A0 + L.oRed + 2, ss);
    HIPCHK(h, hipMemcpy2DAsync(smb(SM_BFULL), pitch, A0 + L.oBm, pitch, sizeof(double) * (size_t)mm, count, hipMemcpyDeviceToDevice, st));
    HIPCHK(h, hipMemset2DAsync(crow, pitch, 0, sizeof(double) * (size_t)NB * mp, count, st));
  }
  if (np >= 4 * SPLITK_CHUNK) {
    HIPCHK(h, launch_gemm_splitk(st, 0, 0, mp, 1, np, 0.0, A0 + L.oAm, np, A0 + L.oY, 1, 0.0, crow, 1, A0 + L.oWs, SPLITK_CHUNK, count, ss, ss, ss, ss,
                                 inv_s, CELL_PAR));
  } else {
    HIPCHK(h, launch_gemm(st, 0, 0, mp, 1, np, 0.0, A0 + L.oAm, np, A0 + L.oY, 1, 0.0, crow, 1, 0, 64, 1, 0, 0, 0, count, ss, ss, ss, inv_s, CELL_PAR));
  }
  HIPCHK(h, potrf_lower(st, A0 + L.oBm, mp, mp, NB, A0 + L.oInvDB, info0, A0 + L.oStage, nullptr, nullptr, count, ss, 2 * CELL_RES, &h->tune));
  const bool fused_small = one_block && want_grad;  // sgpr_small_kernel: the M x M algebra of the gradient, and these two reductions with it
  if (!fused_small)
    hipLaunchKernelGGL(logdet_quad_kernel, dim3(count), dim3(256), 0, st, (const double*)(A0 + L.oBm), (int64_t)mp, (const double*)crow, mp,
                       A0 + L.oRed, ss, (int)ss);
  // ---- gradient (sgpr_gradient) ----
  // one block: L^-1 and LB^-1 ARE the inverses of the diagonal blocks that the factorisations left behind (trtri_lower would
  // clear a matrix and copy them into it)
  double *Linv = one_block ? A0 + L.oInvDL : smb(SM_LINV), *LBinv = one_block ? A0 + L.oInvDB : smb(SM_LBINV), *Qinv = smb(SM_QINV), *Sinv = smb(SM_SINV), *R = smb(SM_R), *T1 = smb(SM_T1),
         *T2 = smb(SM_T2), *W = smb(SM_W), *GQ = smb(SM_GQ), *Bfull = smb(SM_BFULL);
  double* mvec = A0 + L.oVecs;
  double* qvec = A0 + L.oVecs + 4 * mp;
  double* partP = A0 + L.oPart;
  double* partQ = partP + L.part_p;
  double* sums = partQ + L.part_q;
  const int tiles_m = mp / KM_T, tiles_n = np / KM_T, width = L.width;
  if (want_grad) {
    auto gemm_mm = [&](hipStream_t sx, int ta, int tb, const double* A, const double* B, double* C, int flags) {
      return launch_gemm(sx, ta, tb, mp, mp, mp, 1.0, A, mp, B, mp, 0.0, C, mp, flags, 0, 1, 0, 0, 0, count, ss, ss, ss);
    };
    if (fused_small) {
      hipLaunchKernelGGL(sgpr_small_kernel, dim3(count), dim3(256), 0, st, (const double*)(A0 + L.oBm), (const double*)crow,
                         (const double*)(A0 + L.oInvDL), (const double*)(A0 + L.oInvDB), (const double*)Bfull, A0 + L.oRed, mvec, W, GQ, ss);
    } else {
      if (!one_block) {
        HIPCHK(h, hipMemset2DAsync(Linv, pitch, 0, sizeof(double) * (size_t)mm, count, st));
        HIPCHK(h, trtri_lower(st, A0 + L.oQm, mp, A0 + L.oInvDL, Linv, mp, T1, mp, mp, count, ss, ss));
        HIPCHK(h, hipMemset2DAsync(LBinv, pitch, 0, sizeof(double) * (size_t)mm, count, st));
        HIPCHK(h, trtri_lower(st, A0 + L.oBm, mp, A0 + L.oInvDB, LBinv, mp, T1, mp, mp, count, ss, ss));
      }
      HIPCHK(h, gemm_mm(st, 0, 0, LBinv, Linv, R, GEMM_A_LOWER | GEMM_B_LOWER));
      HIPCHK(h, gemm_mm(st, 1, 0, R, R, Sinv, GEMM_A_UPPER | GEMM_B_LOWER));
      HIPCHK(h, gemm_mm(st, 0, 0, Bfull, Linv, T2, GEMM_B_LOWER));
      HIPCHK(h, gemm_mm(st, 1, 0, Linv, T2, T1, GEMM_A_UPPER));
      // m = L^-T LB^-T c
      HIPCHK(h, gemm_mm(st, 1, 0, Linv, Linv, Qinv, GEMM_A_UPPER | GEMM_B_LOWER));
      hipLaunchKernelGGL(copy_row_batch_kernel, dim3((mp + 255) / 256, count), dim3(256), 0, st, (const double*)crow, mvec, mp, ss);
      HIPCHK(h, trsv_lower(st, A0 + L.oBm, mp, A0 + L.oInvDB, mvec, mp, true, count, ss));
      HIPCHK(h, trsv_lower(st, A0 + L.oQm, mp, A0 + L.oInvDL, mvec, mp, true, count, ss));
      hipLaunchKernelGGL(sgpr_combine_kernel, dim3((mp * mp + 255) / 256, count), dim3(256), 0, st, (const double*)Qinv, (const double*)Sinv,
                         (const double*)T1, (const double*)mvec, mp, W, GQ, ss);
    }
    HIPCHK(h, launch_gemm(st, 0, 0, mp, np, mp, 1.0, W, mp, A0 + L.oP, np, 0.0, A0 + L.oWP, np, 0, 0, 1, 0, 0, 0, count, ss, ss, ss));
    TraceArgs tp{A0 + L.oZ, h->X.p, nullptr, A0 + L.oWP, np, mvec, A0 + L.oY, 0.0, 0.0, m, n, d, 0.0, 0, partP, A0 + L.oWHP, np, tiles_n};
    tp.cell_par = cpar;
    tp.w_stride = ss;
    tp.uv_stride = ss;
    tp.partial_stride = ss;
    tp.a_stride = ss;
    tp.wh_stride = ss;
    tp.scale_inv_noise = 1;
    tp.iso = h->ard ? 0 : 1;
    TraceArgs tq{A0 + L.oZ, A0 + L.oZ, nullptr, GQ, mp, nullptr, nullptr, 1.0, 0.0, m, m, d, 0.0, 0, partQ, A0 + L.oWHQ, mp, tiles_m};
    tq.cell_par = cpar;
    tq.w_stride = ss;
    tq.partial_stride = ss;
    tq.a_stride = ss;
    tq.b_stride = ss;
    tq.wh_stride = ss;
    tq.iso = h->ard ? 0 : 1;
    HIPCHK(h, launch_trace_pair(st, h->kid, with_form(tp, h), tiles_m * tiles_n, with_form(tq, h), tiles_m * tiles_m, count));
    HIPCHK(h, launch_gemm(st, 1, 0, np, 1, mp, 1.0, A0 + L.oP, np, mvec, 1, 0.0, qvec, 1, 0, 64, 1, 0, 0, 0, count, ss, ss, ss));
    hipLaunchKernelGGL(resid_sumsq_kernel, dim3(1, count), dim3(256), 0, st, (const double*)(A0 + L.oY), (const double*)qvec, n, A0 + L.oRed + 4, ss,
                       ss);
    if (fused_small) {
      // (|LB^-1|_F^2 came out of sgpr_small_kernel)
    } else {
      const int nb = mp < 64 ? mp : 64;
      double* part = A0 + L.oVecs + 2 * mp;
      hipLaunchKernelGGL(sumsq_partial_kernel, dim3(nb, count), dim3(256), 0, st, (const double*)LBinv, (int64_t)mp, mp, mp, part, ss);
      hipLaunchKernelGGL(sum_partials_kernel, dim3(1, count), dim3(64), 0, st, (const double*)part, nb, A0 + L.oRed + 3, ss);
    }
    hipLaunchKernelGGL(dz_kernel, dim3(dz_grid(m, d), count), dim3(256), 0, st, (const double*)(A0 + L.oZ), (const double*)h->X.p,
                       (const double*)(A0 + L.oWHP), (int64_t)np, (const double*)(A0 + L.oWHQ), (int64_t)mp, (const double*)nullptr, m, n, d,
                       A0 + L.odZ, ss, cpar);
  }
  // ---- results: reductions, pivot status, trace sums, dZ -> pinned memory ----
  hipLaunchKernelGGL(sgpr_stage_out_kernel, dim3(count), dim3(256), 0, st, (const double*)h->cellres.p, CELL_RES, h->spin + sg.res,
                     (const double*)(A0 + L.oRed), h->spin + sg.red, want_grad ? (const double*)partP : nullptr, tiles_m * tiles_n,
                     (const double*)partQ, tiles_m * tiles_m, width, h->spin + sg.sum, want_grad ? (const double*)(A0 + L.odZ) : nullptr, m * d,
                     h->spin + sg.dz, ss);
  HIPCHK(h, hipGetLastError());
  return GPRX_OK;
}

// elbo_out[c] (NaN if a Cholesky failed), g: count x ntheta constrained-parameter derivatives, gz: count x m x d (host);
// g / gz may be null (loss only).  status[c]: GPRX_OK / GPRX_ENOTPD.
int sgpr_objective_batch(gprx_handle h, int count, const int* units, const Theta* ts, const double* zs, double* elbo_out, double* g, double* gz,
                         int* status) {
  const SgprLayout L = sgpr_batch_layout(h);
  int rc;
  if ((rc = ensure_sarena(h, count, L))) return rc;
  const int mp = (int)h->mp, m = (int)h->m, d = h->d;
  const int width = L.width;
  hipStream_t st = h->stream;
  const SgprStage sg = sgpr_stage(h, count, L);
  double* par = h->spin + sg.par;
  for (int c = 0; c < count; ++c) {
    double* row = par + (size_t)c * CELL_PAR;
    std::memset(row, 0, sizeof(double) * CELL_PAR);
    row[0] = ts[c].variance;
    row[1] = ts[c].noise;
    row[2] = (double)units[c];
    row[3] = 1.0 / ts[c].noise;
    for (int k = 0; k < d; ++k) row[CELL_PAR_LS + k] = ts[c].ls[k];
  }
  std::memcpy(h->spin + sg.z, zs, sizeof(double) * (size_t)count * m * d);
  static const bool no_graph = getenv("GPRX_NO_GRAPH") != nullptr;  // escape hatch: eager launches
  const bool want_grad = g != nullptr;
  bool replayed = false;
  // (the five launches of the fused evaluation go out eagerly: replaying them from a graph starts the first kernel later than a direct
  // launch does -- 178.8 against 172.5 us per 16-cell evaluation, MI355X_MICROARCH.md "graph-replay-floor")
  const bool five_launches = mp == NB && h->sgpr_fused != 0;
  if (!no_graph && !h->sgraph_off && !h->profiling && !five_launches) {
    const std::pair<int, int> key(count, want_grad ? 1 : 0);
    auto it = h->sgraphs.find(key);
    if (it == h->sgraphs.end()) {
      // first evaluation of this shape: eager (every kernel's code object gets loaded outside a capture); the capture happens
      // on the second one
      h->sgraphs.emplace(key, nullptr);
    } else if (it->second == nullptr) {
      // Relaxed capture mode: another host thread may call a legacy-stream API (hipMemset / hipMemcpy of another handle's set-up)
      // while this capture runs; in the global and thread-local modes the runtime refuses that call ("operation would make the
      // legacy stream depend on a capturing blocking stream") AND invalidates this capture.  The handle's streams are
      // non-blocking, so no implicit dependency on the legacy stream exists that the capture could miss.  Captures are
      // serialised over the process as well; one that fails anyway is abandoned and the handle stays on eager launches.
      std::lock_guard<std::mutex> lock(capture_mutex());
      hipGraph_t graph = nullptr;
      hipGraphExec_t exec = nullptr;
      hipError_t e = hipStreamBeginCapture(st, hipStreamCaptureModeRelaxed);
      if (e == hipSuccess) {
        const int crc = sgpr_batch_enqueue(h, count, L, want_grad);
        e = hipStreamEndCapture(st, &graph);
        if (!crc && e == hipSuccess && graph) e = hipGraphInstantiate(&exec, graph, nullptr, nullptr, 0);
        if (crc) e = hipErrorUnknown;
      }
      if (graph) hipGraphDestroy(graph);
      if (e != hipSuccess || !exec) {
        (void)hipGetLastError();
        h->err.clear();
        h->sgraph_off = true;
        h->sgraphs.erase(it);
      } else {
        it->second = exec;
      }
    }
    it = h->sgraphs.find(key);
    if (!h->sgraph_off && it != h->sgraphs.end() && it->second != nullptr) {
      HIPCHK(h, hipGraphLaunch(it->second, st));
      replayed = true;
    }
  }
  if (!replayed && (rc = sgpr_batch_enqueue(h, count, L, want_grad))) return rc;
  const double* hres = h->spin + sg.res;
  const double* hred = h->spin + sg.red;
  const double* hsum = h->spin + sg.sum;
  const double* hdz = h->spin + sg.dz;
  HIPCHK(h, wait_stream(h, st));
  h->factorized = false;  // the single-model state of the handle is untouched but no longer "the last evaluation"
  int first_error = GPRX_OK;
  const double nn = (double)h->n;
  for (int c = 0; c < count; ++c) {
    int info = 0;
    std::memcpy(&info, hres + (size_t)c * CELL_RES + 2, sizeof(int));
    if (status) status[c] = info == 0 ? GPRX_OK : GPRX_ENOTPD;
    if (info != 0) {
      if (!first_error) {
        char msg[160];
        snprintf(msg, sizeof msg, "cell %d: Kuu or B not positive definite: pivot %d", c, info);
        first_error = fail(h, GPRX_ENOTPD, msg);
      }
      elbo_out[c] = std::numeric_limits<double>::quiet_NaN();
      continue;
    }
    const double* red = hred + (size_t)c * 8;
    const double s = ts[c].noise, v = ts[c].variance;
    elbo_out[c] = sgpr_asm_elbo(nn, h->yy[units[c]], v, s, red);  // (sgpr_asm.h: the resident Adam loop forms the same sums on the device)
    if (!g) continue;
    const double* hs = hsum + (size_t)c * 2 * width;
    double* gc = g + (size_t)c * h->ntheta;
    for (int k = 0; k < h->ntheta; ++k) gc[k] = sgpr_asm_dparam(k, h->nlen, h->ard, d, width, nn, mp, v, s, red, hs);
    if (gz) std::memcpy(gz + (size_t)c * m * d, hdz + (size_t)c * m * d, sizeof(double) * m * d);
  }
  return first_error;
}

// SGPR.predict_y for every cell of the batch that sgpr_objective_batch has just factorised (its cell blocks hold L, invDL, LB,
// invDB and c): the nine small launches of one model's predict serve all cells -- Kus per cell (own Z and hyperparameters),
// tmp1 = L^-1 Kus, tmp2 = LB^-1 tmp1, mean = tmp2^T c, var = v + colsum(tmp2^2) - colsum(tmp1^2) (+ s).  Same kernels and
// operation order as gprx_predict_dev on each cell: bit-identical values.  means / vars: (count, ns) device, row-major.
int sgpr_predict_batch(gprx_handle h, int count, const double* xs_dev, int64_t ns, double* means_dev, double* vars_dev, int include_noise) {
  const SgprLayout L = sgpr_batch_layout(h);
  const int mp = (int)h->mp, m = (int)h->m;
  const int64_t ss = L.ss;
  hipStream_t st = h->stream;
  double* A0 = h->sarena.p;
  const double* cpar = h->cellpar.p;
  const int rows_per_chunk = 256;
  const int nchunks = (mp + rows_per_chunk - 1) / rows_per_chunk;
  const int tile = SGPR_PRED_TILE;
  const double* cvec = A0 + L.oBm + (int64_t)mp * mp;
  for (int64_t t0 = 0; t0 < ns; t0 += tile) {
    const int ts = (int)std::min<int64_t>(tile, ns - t0);
    const int tsp = (int)round_up(ts, NB);
    KmatArgs ka{A0 + L.oZ, xs_dev + t0 * h->d, nullptr, A0 + L.oKs, tile, m, ts, h->d, mp, tsp, 0.0, 0.0, 0, 0.0, nullptr, 0};
    ka.cell_par = cpar;
    ka.out_stride = ss;
    ka.a_stride = ss;
    ka.diag_const = 1;
    HIPCHK(h, launch_kmat(st, h->kid, with_form(ka, h), count));
    const dim3 pgrid((ts + 255) / 256, nchunks, count), fgrid((ts + 255) / 256, count);
    HIPCHK(h, trsm_lower_left(st, A0 + L.oQm, mp, A0 + L.oInvDL, A0 + L.oKs, tile, mp, tsp, count, ss));
    hipLaunchKernelGGL(colreduce_partial, pgrid, dim3(256), 0, st, (const double*)(A0 + L.oKs), (int64_t)tile, (const double*)nullptr, mp, ts,
                       rows_per_chunk, A0 + L.oPred, ss, (int64_t)0, ss);
    // var = (v [+ s]) - colsum(tmp1^2): per-cell base from the parameter table ([0] variance, [1] noise)
    hipLaunchKernelGGL(colreduce_final, fgrid, dim3(256), 0, st, (const double*)(A0 + L.oPred), nchunks, ts, 0.0, -1.0, 0, vars_dev + t0, ss, ns, cpar,
                       include_noise ? cpar + 1 : (const double*)nullptr, CELL_PAR);
    HIPCHK(h, trsm_lower_left(st, A0 + L.oBm, mp, A0 + L.oInvDB, A0 + L.oKs, tile, mp, tsp, count, ss));
    hipLaunchKernelGGL(colreduce_partial, pgrid, dim3(256), 0, st, (const double*)(A0 + L.oKs), (int64_t)tile, cvec, mp, ts, rows_per_chunk,
                       A0 + L.oPred, ss, ss, ss);
    hipLaunchKernelGGL(colreduce_final, fgrid, dim3(256), 0, st, (const double*)(A0 + L.oPred), nchunks, ts, 0.0, 1.0, 0, means_dev + t0, ss, ns);
    hipLaunchKernelGGL(colreduce_partial, pgrid, dim3(256), 0, st, (const double*)(A0 + L.oKs), (int64_t)tile, (const double*)nullptr, mp, ts,
                       rows_per_chunk, A0 + L.oPred, ss, (int64_t)0, ss);
    hipLaunchKernelGGL(colreduce_final, fgrid, dim3(256), 0, st, (const double*)(A0 + L.oPred), nchunks, ts, 0.0, 1.0, 1, vars_dev + t0, ss, ns);
  }
  HIPCHK(h, hipGetLastError());
  return GPRX_OK;
}

// gprx_adam_batch for sparse models with M <= 64: the loop RESIDENT on the device.  A step is FOUR launches (sgpr_fused.h: pass 1, mid,
// pass 2, and sf_adam_prep_kernel = partial sums + loss + gradient + Keras's update + the stop rule of gpr.py:160-171, then Kuu, L, L^-1
// of the updated variables with the positive transforms evaluated on the device); cells that have stopped return at once from every launch.  The host enqueues `check_every` steps, then reads
// the stop flags (count + 1 ints through pinned memory) -- no gradient, loss or parameter crosses the host link during the run (round 4:
// every step synchronised, downloaded the gradients, updated on the host and uploaded).  Same variables as the host-stepped loop, bit
// for bit (sgpr_asm.h, px_math.h; tests/test_gpu_gpras.py).  A cell whose Kuu or B stops being positive definite ends the call with
// GPRX_ENOTPD at the next check; the other cells may then be up to check_every - 1 steps past that evaluation.
int sgpr_adam_resident(gprx_handle h, int count, const int* units, double* theta, double* z, int mask, int max_iter, int* n_evals, int* batches) {
  const SgprLayout L = sgpr_batch_layout(h);
  int rc;
  if ((rc = ensure_sarena(h, count, L))) return rc;
  hipStream_t st = h->stream;
  const int nt = h->ntheta;
  const int64_t nz = h->m * h->d, gw = nt + nz;
  // ---- device state: doubles first, then ints ----
  const size_t n_dbl = (size_t)count * nt + 2 * (size_t)count * gw + 2 * (size_t)count + (size_t)max_iter + 1 + (size_t)h->n_units;
  const size_t n_int = 5 * (size_t)count + gprx_ctx::SF_MAX_GROUPS;  // (one error word per group of cells)
  if ((rc = ensure(h, h->adam_dev, sizeof(double) * n_dbl + sizeof(int) * n_int))) return rc;
  const size_t pin_need = sizeof(int) * ((size_t)count + gprx_ctx::SF_MAX_GROUPS);
  if (h->adam_pin_bytes < pin_need) {
    if (h->adam_pin) HIPCHK(h, hipHostFree(h->adam_pin));
    h->adam_pin = nullptr;
    HIPCHK(h, hipHostMalloc((void**)&h->adam_pin, pin_need, hipHostMallocDefault));
    h->adam_pin_bytes = pin_need;
  }
  SfAdam ad{};
  double* dp = h->adam_dev.p;
  ad.theta = dp;                 dp += (size_t)count * nt;
  ad.mom = dp;                   dp += (size_t)count * gw;
  ad.vel = dp;                   dp += (size_t)count * gw;
  ad.best = dp;                  dp += count;
  ad.loss = dp;                  dp += count;
  double* d_alpha = dp;          dp += (size_t)max_iter + 1;
  double* d_yy = dp;             dp += h->n_units;
  int* ip = reinterpret_cast<int*>(dp);
  ad.stale = ip;                 ip += count;
  ad.active = ip;                ip += count;
  ad.n_evals = ip;               ip += count;
  ad.tstep = ip;                 ip += count;
  int* d_units = ip;             ip += count;
  ad.error = ip;
  ad.units = d_units;
  ad.alpha = d_alpha;
  ad.yy = d_yy;
  ad.nt = nt;
  ad.nlen = h->nlen;
  ad.ard = h->ard;
  ad.mask = mask;
  ad.max_iter = max_iter;
  // ---- initial state (host vectors live until the synchronisation below) ----
  std::vector<double> hd(n_dbl, 0.0);
  std::vector<int> hi(n_int, 0);
  std::memcpy(hd.data(), theta, sizeof(double) * (size_t)count * nt);
  {
    double* hbest = hd.data() + (ad.best - h->adam_dev.p);
    for (int c = 0; c < count; ++c) hbest[c] = std::numeric_limits<double>::infinity();
    double* halpha = hd.data() + (d_alpha - h->adam_dev.p);
    for (int t = 1; t <= max_iter; ++t)
      halpha[t] = ADAM_LR * std::sqrt(1.0 - std::pow(ADAM_BETA2, (double)t)) / (1.0 - std::pow(ADAM_BETA1, (double)t));  // gprx_adam_batch's expression
    std::memcpy(hd.data() + (d_yy - h->adam_dev.p), h->yy.data(), sizeof(double) * h->n_units);
    for (int c = 0; c < count; ++c) {
      hi[(size_t)count + c] = 1;          // active
      hi[4 * (size_t)count + c] = units[c];
    }
  }
  HIPCHK(h, hipMemcpyAsync(h->adam_dev.p, hd.data(), sizeof(double) * n_dbl, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpyAsync(ad.stale, hi.data(), sizeof(int) * n_int, hipMemcpyHostToDevice, st));
  HIPCHK(h, hipMemcpy2DAsync(h->sarena.p + L.oZ, sizeof(double) * (size_t)L.ss, z, sizeof(double) * (size_t)nz, sizeof(double) * (size_t)nz, count,
                             hipMemcpyHostToDevice, st));
  HIPCHK(h, hipStreamSynchronize(st));
  SfParams p = sgpr_fused_params(h, L, true);
  p.active = ad.active;
  p.store_factors = 0;  // (nobody predicts from the cell blocks of a running optimisation)
  const int iso = (h->ard || h->dist_form) ? 0 : 1;
  static const int check_every = [] {
    const char* e = getenv("GPRX_ADAM_CHECK_EVERY");
    const int v = e ? atoi(e) : 0;
    return v > 0 ? v : 25;
  }();
  int* flags = reinterpret_cast<int*>(h->adam_pin);
  int error_cell = 0;
  h->factorized = false;  // the cell blocks are overwritten
  h->sparse_view = false;
  // (large batches: two groups of cells on two streams, one launch apart -- sf_group_count)
  constexpr int MAXG = gprx_ctx::SF_MAX_GROUPS;
  const int ngroups = sf_group_count(count, L.nsplit);
  if (ngroups > 1 && (rc = sf_group_streams(h, ngroups))) return rc;
  SfParams pg[MAXG];
  SfAdam adg[MAXG];
  int cells_g[MAXG], cell0_g[MAXG];
  hipStream_t sg_[MAXG];
  for (int g = 0, cell0 = 0; g < ngroups; ++g) {
    const int cells = count / ngroups + (g < count % ngroups ? 1 : 0);
    sg_[g] = g == 0 ? st : h->sf_streams[g - 1];
    cell0_g[g] = cell0;
    cells_g[g] = cells;
    pg[g] = sf_params_from(p, cell0);
    adg[g] = ad;
    adg[g].theta += (int64_t)cell0 * nt;
    adg[g].mom += (int64_t)cell0 * gw;
    adg[g].vel += (int64_t)cell0 * gw;
    adg[g].best += cell0;
    adg[g].loss += cell0;
    adg[g].stale += cell0;
    adg[g].active += cell0;
    adg[g].n_evals += cell0;
    adg[g].tstep += cell0;
    adg[g].units += cell0;
    adg[g].error += g;
    cell0 += cells;
  }
  for (int g = 0; g < ngroups; ++g)
    HIPCHK(h, sf_launch_prep(sg_[g], h->kid, h->dist_form, pg[g], cells_g[g], nullptr, nullptr, h->cellpar.p + (size_t)cell0_g[g] * CELL_PAR, &adg[g]));  // opens step 1
  for (int done = 0; done < max_iter;) {
    const int k = std::min(check_every, max_iter - done);
    for (int i = 0; i < k; ++i) {
      for (int g = 0; g < ngroups; ++g) {
        // (a group starts one launch behind the group before it; groups that start together stay in lock step and gain nothing)
        const bool first = done == 0 && i == 0;
        if (first && g > 0) HIPCHK(h, hipStreamWaitEvent(sg_[g], h->sf_evs[g - 1], 0));
        HIPCHK(h, sf_launch_pass1(sg_[g], h->kid, h->dist_form, pg[g], cells_g[g]));
        if (first && g + 1 < ngroups) HIPCHK(h, hipEventRecord(h->sf_evs[g], sg_[g]));
        HIPCHK(h, sf_launch_mid(sg_[g], pg[g], cells_g[g]));
        HIPCHK(h, sf_launch_pass2(sg_[g], h->kid, h->dist_form, iso, pg[g], cells_g[g]));
        // closes this step, opens the next
        HIPCHK(h, sf_launch_adam_prep(sg_[g], h->kid, h->dist_form, iso, pg[g], cells_g[g], adg[g], h->cellpar.p + (size_t)cell0_g[g] * CELL_PAR));
      }
    }
    done += k;
    for (int g = 1; g < ngroups; ++g) {
      HIPCHK(h, hipEventRecord(h->sf_evs[g], sg_[g]));
      HIPCHK(h, hipStreamWaitEvent(st, h->sf_evs[g], 0));
    }
    HIPCHK(h, hipMemcpyAsync(flags, ad.active, sizeof(int) * count, hipMemcpyDeviceToHost, st));
    HIPCHK(h, hipMemcpyAsync(flags + count, ad.error, sizeof(int) * ngroups, hipMemcpyDeviceToHost, st));
    HIPCHK(h, wait_stream(h, st));
    error_cell = 0;
    for (int g = ngroups - 1; g >= 0; --g)
      if (flags[count + g] != 0) error_cell = cell0_g[g] + flags[count + g];
    bool any = false;
    for (int c = 0; c < count; ++c) any = any || flags[c] != 0;
    if (error_cell != 0 || !any) break;
  }
  // ---- results ----
  for (int g = 1; g < ngroups; ++g) {  // (nothing of the other group's stream may outlive the call: max_iter = 0 enqueued its prep launch only)
    HIPCHK(h, hipEventRecord(h->sf_evs[g], sg_[g]));
    HIPCHK(h, hipStreamWaitEvent(st, h->sf_evs[g], 0));
  }
  HIPCHK(h, hipMemcpyAsync(theta, ad.theta, sizeof(double) * (size_t)count * nt, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpy2DAsync(z, sizeof(double) * (size_t)nz, h->sarena.p + L.oZ, sizeof(double) * (size_t)L.ss, sizeof(double) * (size_t)nz, count,
                             hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(n_evals, ad.n_evals, sizeof(int) * count, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (batches) {
    int mx = 0;
    for (int c = 0; c < count; ++c) mx = std::max(mx, n_evals[c]);
    *batches = mx;
  }
  if (error_cell != 0) {
    char msg[160];
    snprintf(msg, sizeof msg, "cell %d: Kuu or B not positive definite", error_cell - 1);
    return fail(h, GPRX_ENOTPD, msg);
  }
  return GPRX_OK;
}

int check_handle(gprx_handle h) {
  if (!h) return fail(nullptr, GPRX_EINVAL, "null handle");
  hipError_t e = hipSetDevice(h->device);
  if (e != hipSuccess) return fail(h, GPRX_EHIP, std::string("hipSetDevice: ") + hipGetErrorString(e));
  return GPRX_OK;
}

}  // namespace

// ======================================================================================================
extern "C" {

int gprx_version(void) { return GPRX_VERSION; }

const char* gprx_last_error(gprx_handle h) { return h ? h->err.c_str() : g_err.c_str(); }

int gprx_device_count(int* count) {
  if (!count) return fail(nullptr, GPRX_EINVAL, "count is null");
  HIPCHK(nullptr, hipGetDeviceCount(count));
  return GPRX_OK;
}

int gprx_create(int device, int64_t n, int d, int64_t m, int kernel_id, int ard, gprx_handle* out) {
  if (!out) return fail(nullptr, GPRX_EINVAL, "out is null");
  *out = nullptr;
  if (n <= 0 || d <= 0 || m < 0) return fail(nullptr, GPRX_EINVAL, "n, d must be positive and m non-negative");
  if (kernel_id < 0 || kernel_id > 4) return fail(nullptr, GPRX_EINVAL, "unknown kernel id");
  if (n > (1 << 30) || m > (1 << 30)) return fail(nullptr, GPRX_EINVAL, "n or m too large");
  HIPCHK(nullptr, hipSetDevice(device));
  gprx_handle h = new gprx_ctx();
  h->device = device;
  h->n = n;
  h->m = m;
  h->d = d;
  h->kid = kernel_id;
  h->ard = ard ? 1 : 0;
  h->nlen = ard ? d : 1;
  h->ntheta = 2 + h->nlen;
  h->np = round_up(n, NB);
  h->mp = round_up(m, NB);
  h->tune = potrf_tuning();  // a private copy: later gprx_set_tuning calls (process defaults) do not reach this handle
  h->predict_path = predict_path_tuning();
  h->sgpr_fused = sgpr_fused_tuning();
  // normal priority on purpose: measured on MI355X, raised/lowered stream priorities do nothing for a single
  // cell and cut the throughput of several concurrent cells by up to 2x
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(nullptr, GPRX_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  h->own_stream = true;
  for (auto& ev : h->ev) hipEventCreate(&ev);
  // the look-ahead stream is created on first use (ensure_lookahead): HIP binds streams to its 4 hardware
  // queues round-robin at creation, so an unused second stream per handle would leave the main streams of
  // many concurrent cells on half of the queues (measured: 584 instead of 797 fits/s with 12 cells)
  int rc;
  if ((rc = ensure(h, h->invls, sizeof(double) * d)) || (rc = ensure(h, h->red, sizeof(double) * 16))) {
    gprx_destroy(h);
    return rc;
  }
  e = hipMalloc((void**)&h->info, sizeof(int));
  if (e == hipSuccess) e = hipHostMalloc((void**)&h->pin, sizeof(double) * 80, hipHostMallocDefault);
  if (e == hipSuccess) e = hipMalloc((void**)&h->gparams, sizeof(double) * 2);
  if (e != hipSuccess) {
    gprx_destroy(h);
    return fail(nullptr, GPRX_ENOMEM, "hipMalloc(info) / hipHostMalloc(staging)");
  }
  *out = h;
  return GPRX_OK;
}

int gprx_destroy(gprx_handle h) {
  if (!h) return GPRX_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (Buf* b : {&h->X, &h->Y, &h->Z, &h->invls, &h->alpha, &h->red, &h->Kmat, &h->invD, &h->Xinv, &h->Tmp, &h->partial, &h->xs, &h->Ks,
                 &h->pred, &h->P, &h->Am, &h->Qm, &h->Bm, &h->invDL, &h->invDB, &h->SM, &h->WP, &h->WHP, &h->WHQ, &h->vecs, &h->dZ,
                 &h->dstage, &h->splitws, &h->arena, &h->cellpar, &h->cellres, &h->garena, &h->gpartial, &h->apart, &h->twork, &h->sarena})
    if (b->p && !b->borrowed) hipFree(b->p);
  if (h->bpin) hipHostFree(h->bpin);
  if (h->spin) hipHostFree(h->spin);
  if (h->sf_stamps) hipFree(h->sf_stamps);
  for (auto& ev : h->sf_evs)
    if (ev) hipEventDestroy(ev);
  for (auto& sx : h->sf_streams)
    if (sx) hipStreamDestroy(sx);
  if (h->adam_dev.p) hipFree(h->adam_dev.p);
  if (h->adam_pin) hipHostFree(h->adam_pin);
  for (auto& ev : h->bev)
    if (ev) hipEventDestroy(ev);
  if (h->stagger_evt) hipEventDestroy(h->stagger_evt);
  for (auto& ev : h->kev)
    if (ev) hipEventDestroy(ev);
  for (auto& ev : h->cev)
    if (ev) hipEventDestroy(ev);
  if (h->wev) hipEventDestroy(h->wev);
  if (h->info) hipFree(h->info);
  if (h->pin) hipHostFree(h->pin);
  if (h->gparams) hipFree(h->gparams);
  drop_graphs(h);
  for (auto& ev : h->ev)
    if (ev) hipEventDestroy(ev);
  h->pstreams.destroy();
  h->dag.destroy();
  if (h->own_stream && h->stream) hipStreamDestroy(h->stream);
  delete h;
  return GPRX_OK;
}

int gprx_set_stream(gprx_handle h, void* hip_stream) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (h->own_stream && h->stream) {
    hipStreamSynchronize(h->stream);
    hipStreamDestroy(h->stream);
  }
  h->stream = (hipStream_t)hip_stream;
  h->own_stream = false;
  return GPRX_OK;
}

int gprx_set_distance_form(gprx_handle h, int form) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (form != GPRX_DIST_DIFFERENCE && form != GPRX_DIST_EXPANDED) return fail(h, GPRX_EINVAL, "unknown distance form");
  if (form != h->dist_form) {
    HIPCHK(h, hipStreamSynchronize(h->stream));
    h->dist_form = form;
    h->factorized = false;  // resident factorisations (and captured graphs) were built with the other form
    h->have_linv = false;
    drop_graphs(h);
    std::fill(h->slot_ok.begin(), h->slot_ok.end(), 0);
  }
  return GPRX_OK;
}

int gprx_synchronize(gprx_handle h) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  HIPCHK(h, wait_stream(h, h->stream));
  return GPRX_OK;
}

int gprx_set_data(gprx_handle h, const double* x, const double* y, int n_units) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (!x || !y || n_units <= 0) return fail(h, GPRX_EINVAL, "x, y must be non-null and n_units positive");
  if ((rc = ensure(h, h->X, sizeof(double) * h->n * h->d))) return rc;
  if ((rc = ensure(h, h->Y, sizeof(double) * h->np * n_units))) return rc;
  // unit-major copy of y, zero padded to np
  std::vector<double> yt((size_t)h->np * n_units, 0.0);
  for (int64_t i = 0; i < h->n; ++i)
    for (int u = 0; u < n_units; ++u) yt[(size_t)u * h->np + i] = y[i * n_units + u];
  HIPCHK(h, copy_sync(h->X.p, x, sizeof(double) * h->n * h->d, hipMemcpyHostToDevice));
  HIPCHK(h, copy_sync(h->Y.p, yt.data(), sizeof(double) * yt.size(), hipMemcpyHostToDevice));
  h->n_units = n_units;
  h->factorized = false;
  h->have_linv = false;
  std::fill(h->slot_ok.begin(), h->slot_ok.end(), 0);  // resident batch slots were factorised from the old data
  std::fill(h->slot_unit.begin(), h->slot_unit.end(), -1);
  drop_graphs(h);
  h->yy.assign(n_units, 0.0);
  for (int u = 0; u < n_units; ++u) {
    double acc = 0.0;
    for (int64_t i = 0; i < h->n; ++i) acc += y[i * n_units + u] * y[i * n_units + u];
    h->yy[u] = acc;
  }
  return GPRX_OK;
}

// priors and softplus chain rule on the hyperparameter part of a gradient; loss = -(value + log prior)
void chain_rule(gprx_handle h, const Theta& t, int mask, const double* g, double* grad) {
  grad[0] = sgpr_asm_chain(g[0], t.variance, t.w_var, (mask & GPRX_TRAIN_VARIANCE) != 0);
  for (int k = 0; k < h->nlen; ++k)
    grad[1 + k] = sgpr_asm_chain(g[1 + k], t.ls[k], t.w_len[k], (mask & GPRX_TRAIN_LENGTHSCALE) != 0);
  grad[1 + h->nlen] = sgpr_asm_chain(g[1 + h->nlen], t.noise, t.w_noise, (mask & GPRX_TRAIN_NOISE) != 0);
}

static int objective_impl(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss, double* grad) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (!theta) return fail(h, GPRX_EINVAL, "theta is null");
  if (unit < 0 || unit >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
  for (int k = 0; k < h->ntheta; ++k)
    if (!std::isfinite(theta[k])) return fail(h, GPRX_EINVAL, "theta is not finite");
  const Theta t = decode_theta(h, theta);
  const bool sparse = h->m != 0;
  double value = 0.0;  // LML (exact) or ELBO (sparse)
  if (sparse && h->mp == NB && h->sgpr_fused && h->d <= CELL_PAR - CELL_PAR_LS && !h->profiling) {
    // M <= 64: one model is a batch of one cell through the five launches of sgpr_fused.h -- the same kernels, the same summation
    // order as any batch (a model evaluated alone and inside a batch agree bit for bit); the factorisation stays in cell block 0,
    // where gprx_predict reads it (sparse_view)
    if (!z) return fail(h, GPRX_EINVAL, "z (inducing inputs) is null for a sparse model");
    const int64_t nz = h->m * h->d;
    for (int64_t e = 0; e < nz; ++e)
      if (!std::isfinite(z[e])) return fail(h, GPRX_EINVAL, "z is not finite");
    std::vector<double> g(grad ? h->ntheta : 0), gzv(grad ? nz : 0);
    int st = GPRX_OK;
    if ((rc = sgpr_objective_batch(h, 1, &unit, &t, z, &value, grad ? g.data() : nullptr, grad ? gzv.data() : nullptr, &st))) return rc;
    if (loss) *loss = -(value + log_prior(h, t, mask));
    if (grad) {
      chain_rule(h, t, mask, g.data(), grad);
      double* gz = grad + h->ntheta;
      for (int64_t e = 0; e < nz; ++e) gz[e] = (mask & GPRX_TRAIN_Z) ? -gzv[e] : 0.0;
    }
    for (auto& ev : h->ev) HIPCHK(h, hipEventRecord(ev, h->stream));
    for (double& tm : h->timings) tm = 0.0;
    h->factorized = true;
    h->sparse_view = true;
    h->cur_unit = unit;
    h->variance = t.variance;
    h->noise = t.noise;
    h->ls = t.ls;
    return GPRX_OK;
  }
  // exact model with gradient: ONE stream synchronisation for both halves, and alpha from the inverse the gradient builds (the
  // 64 dependent launches of the backward substitution drop out of the evaluation); a non-PD matrix is reported by the
  // factorisation's status as before (the gradient launches behind it are then wasted, not wrong: nothing is read back)
  static const bool fused_eval = !(getenv("GPRX_FUSED_EVAL") && atoi(getenv("GPRX_FUSED_EVAL")) == 0);
  const bool fused = !sparse && grad && fused_eval && !h->profiling;
  std::vector<double> ghost(fused ? 2 + h->d : 0);
  if (sparse) {
    if ((rc = sgpr_factorize(h, unit, t, z, &value))) return rc;
  } else if (fused) {
    if ((rc = exact_factorize_enqueue(h, unit, t, true, false, false))) return rc;
    if ((rc = exact_gradient_enqueue(h, t, ghost.data(), true))) {
      hipStreamSynchronize(h->stream);
      return rc;
    }
    if ((rc = exact_factorize_finish(h, &value))) return rc;
  } else {
    if ((rc = exact_factorize(h, unit, t, &value))) return rc;
  }
  const double lp = log_prior(h, t, mask);
  if (loss) *loss = -(value + lp);
  if (grad) {
    std::vector<double> g(h->ntheta, 0.0);
    double* gz = sparse ? grad + h->ntheta : nullptr;
    if (sparse) {
      if ((rc = sgpr_gradient(h, unit, t, g.data(), gz))) return rc;
    } else if (fused) {
      exact_gradient_collect(h, ghost.data(), g.data());
    } else {
      if ((rc = exact_gradient(h, t, g.data()))) return rc;
    }
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    chain_rule(h, t, mask, g.data(), grad);
    if (sparse) {
      const int64_t nz = h->m * h->d;
      if (mask & GPRX_TRAIN_Z) {
        for (int64_t e = 0; e < nz; ++e) gz[e] = -gz[e];
      } else {
        for (int64_t e = 0; e < nz; ++e) gz[e] = 0.0;
      }
    }
  } else {
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
  }
  HIPCHK(h, wait_stream(h, h->stream));
  for (int s = 0; s < 4; ++s) {
    float ms = 0.f;
    hipEventElapsedTime(&ms, h->ev[s], h->ev[s + 1]);
    h->timings[s] = ms;
  }
  return GPRX_OK;
}

int gprx_objective(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss, double* grad) {
  return objective_impl(h, unit, theta, z, mask, loss, grad);
}

int gprx_factorize(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss) {
  return objective_impl(h, unit, theta, z, mask, loss, nullptr);
}

int gprx_factorize_many(int count, gprx_handle* handles, const int* units, const double* thetas, int mask, double* losses) {
  if (count < 0 || !handles || !units || !thetas) return fail(nullptr, GPRX_EINVAL, "null argument");
  std::vector<Theta> ts(count);
  // enqueue every cell's work first (nothing blocks), then wait for each: the cells overlap on the device.
  // With several cells in flight each one runs on its single stream (no look-ahead stream): measured, 12
  // cells reach 2.2x the single-cell rate that way and only 1.5x with two streams per cell.
  for (int i = 0; i < count; ++i) {
    gprx_handle h = handles[i];
    int rc;
    if ((rc = check_handle(h))) return rc;
    if (h->m != 0) return fail(h, GPRX_EINVAL, "gprx_factorize_many: exact models only");
    if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range");
    for (int k = 0; k < h->ntheta; ++k)
      if (!std::isfinite(thetas[(int64_t)i * h->ntheta + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
    ts[i] = decode_theta(h, thetas + (int64_t)i * h->ntheta);
    if ((rc = (count == 1) ? exact_factorize_enqueue(h, units[i], ts[i], true) : exact_factorize_replay(h, units[i], ts[i]))) return rc;
  }
  int first_error = GPRX_OK;
  for (int i = 0; i < count; ++i) {
    gprx_handle h = handles[i];
    hipSetDevice(h->device);
    double lml = 0.0;
    const int rc = exact_factorize_finish(h, &lml);
    if (rc && !first_error) first_error = rc;
    if (losses) losses[i] = rc ? std::numeric_limits<double>::quiet_NaN() : -(lml + log_prior(h, ts[i], mask));
  }
  return first_error;
}

int gprx_adam_batch(gprx_handle h, int count, const int* units, double* theta, double* z, int mask, int max_iter, int* n_evals, int* batches) {
#pragma clang fp contract(off)
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count <= 0 || !units || !theta || !n_evals || max_iter < 0) return fail(h, GPRX_EINVAL, "null argument");
  if (h->m != 0 && !z) return fail(h, GPRX_EINVAL, "z (inducing inputs) is null for a sparse model");
  const int nt = h->ntheta;
  const int64_t nz = h->m * h->d, gw = nt + nz;
  if (batches) *batches = 0;
  for (int i = 0; i < count; ++i) n_evals[i] = 0;
  // trainable elements of a cell's gradient row [d theta | d Z] (theta: [variance, lengthscales..., noise])
  std::vector<char> train((size_t)gw, 0);
  train[0] = (mask & GPRX_TRAIN_VARIANCE) != 0;
  for (int k = 1; k < nt - 1; ++k) train[k] = (mask & GPRX_TRAIN_LENGTHSCALE) != 0;
  train[nt - 1] = (mask & GPRX_TRAIN_NOISE) != 0;
  for (int64_t e = 0; e < nz; ++e) train[nt + e] = (mask & GPRX_TRAIN_Z) != 0;
  bool any = false;
  for (char t : train) any = any || t;
  if (!any) return GPRX_OK;  // nothing trainable: no step can change anything (optimizers._optimize_adam returns at once)
  static const bool adam_on_host = getenv("GPRX_ADAM_HOST") && atoi(getenv("GPRX_ADAM_HOST")) != 0;  // escape hatch: the host-stepped loop
  if (h->m != 0 && h->mp == NB && h->sgpr_fused && h->d <= CELL_PAR - CELL_PAR_LS && !adam_on_host && max_iter > 0) {
    for (int i = 0; i < count; ++i) {
      if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
      for (int k = 0; k < nt; ++k)
        if (!std::isfinite(theta[(size_t)i * nt + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
    }
    for (int64_t e = 0; e < (int64_t)count * nz; ++e)
      if (!std::isfinite(z[e])) return fail(h, GPRX_EINVAL, "z is not finite");
    return sgpr_adam_resident(h, count, units, theta, z, mask, max_iter, n_evals, batches);
  }
  const double lr = ADAM_LR, beta1 = ADAM_BETA1, beta2 = ADAM_BETA2;
  std::vector<double> mom((size_t)count * gw, 0.0), vel((size_t)count * gw, 0.0), best(count, std::numeric_limits<double>::infinity());
  std::vector<int> stale(count, 0), active(count);
  for (int i = 0; i < count; ++i) active[i] = i;
  std::vector<int> a_units(count);
  std::vector<double> a_theta((size_t)count * nt), a_z((size_t)count * nz), losses(count), grads((size_t)count * gw);
  for (int t = 1; t <= max_iter && !active.empty(); ++t) {
    const int na = (int)active.size();
    for (int j = 0; j < na; ++j) {
      const int i = active[j];
      a_units[j] = units[i];
      std::memcpy(&a_theta[(size_t)j * nt], theta + (size_t)i * nt, sizeof(double) * nt);
      if (nz) std::memcpy(&a_z[(size_t)j * nz], z + (size_t)i * nz, sizeof(double) * nz);
    }
    rc = gprx_objective_batch(h, na, a_units.data(), a_theta.data(), nz ? a_z.data() : nullptr, mask, losses.data(), grads.data());
    if (batches) ++*batches;
    for (int j = 0; j < na; ++j) ++n_evals[active[j]];
    if (rc) return rc;  // (GPRX_ENOTPD included: the reference's optimiser dies with the exception of that evaluation)
    const double alpha = lr * std::sqrt(1.0 - std::pow(beta2, (double)t)) / (1.0 - std::pow(beta1, (double)t));
    std::vector<int> next;
    next.reserve(na);
    for (int j = 0; j < na; ++j) {
      const int i = active[j];
      double* mo = &mom[(size_t)i * gw];
      double* ve = &vel[(size_t)i * gw];
      const double* g = &grads[(size_t)j * gw];
      for (int64_t e = 0; e < gw; ++e) {
        if (!train[e]) continue;
        double* x = e < nt ? theta + (size_t)i * nt + e : z + (size_t)i * nz + (e - nt);
        adam_element(g[e], alpha, mo[e], ve[e], *x);  // (sgpr_asm.h: the resident loop's kernel runs the same function)
      }
      if (adam_keep_running(losses[j], best[i], stale[i])) next.push_back(i);
    }
    active.swap(next);
  }
  return GPRX_OK;
}

int gprx_factorize_batch(gprx_handle h, int count, const int* units, const double* thetas, int mask, double* losses, int* status) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count <= 0 || !units || !thetas) return fail(h, GPRX_EINVAL, "count must be positive, units and thetas non-null");
  if (h->m != 0) return fail(h, GPRX_EINVAL, "gprx_factorize_batch: exact models only");
  if (h->d > CELL_PAR - CELL_PAR_LS) return fail(h, GPRX_EINVAL, "gprx_factorize_batch: d <= 64 only");
  // (the handle keeps the decoded parameter sets of the last batch: their lengthscale vectors are reused, no allocation per cell and call --
  // 512 cells of N = 512 spent 68 us here, 5 % of the call)
  std::vector<Theta>& ts = h->batch_thetas;
  if ((int)ts.size() < count) ts.resize(count);
  for (int i = 0; i < count; ++i) {
    if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
    for (int k = 0; k < h->ntheta; ++k)
      if (!std::isfinite(thetas[(int64_t)i * h->ntheta + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
    decode_theta_into(h, thetas + (int64_t)i * h->ntheta, ts[i]);
  }
  std::vector<double>& lml = h->batch_lml;
  if ((int)lml.size() < count) lml.resize(count);
  rc = exact_factorize_batch(h, count, units, ts.data(), lml.data(), status);
  if (rc != GPRX_OK && rc != GPRX_ENOTPD) return rc;
  if (losses)
    for (int i = 0; i < count; ++i) losses[i] = -(lml[i] + log_prior(h, ts[i], mask));  // NaN for a failed cell
  return rc;
}

int gprx_select_slot(gprx_handle h, int slot) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  return select_slot(h, slot);
}

int gprx_last_batch_ms(gprx_handle h, double* ms) {
  if (!h || !ms) return fail(h, GPRX_EINVAL, "null argument");
  *ms = h->batch_ms;
  return GPRX_OK;
}

int gprx_last_timings(gprx_handle h, double* ms4) {
  if (!h || !ms4) return fail(h, GPRX_EINVAL, "null argument");
  for (int s = 0; s < 4; ++s) ms4[s] = h->timings[s];
  return GPRX_OK;
}

int gprx_set_profiling(gprx_handle h, int enabled) {
  if (!h) return fail(h, GPRX_EINVAL, "null handle");
  h->profiling = enabled != 0;
  return GPRX_OK;
}

int gprx_last_profile(gprx_handle h, double* out8) {
  if (!h || !out8) return fail(h, GPRX_EINVAL, "null argument");
  for (int i = 0; i < 8; ++i) out8[i] = h->prof_out[i];
  return GPRX_OK;
}

int gprx_last_kernel_build(gprx_handle h, double* ms, double* bytes) {
  if (!h || !ms || !bytes) return fail(h, GPRX_EINVAL, "null argument");
  *ms = h->kmat_ms;
  *bytes = h->kmat_bytes;
  return GPRX_OK;
}

int gprx_last_cell_kernel(gprx_handle h, double* ms, double* flops, double* cells) {
  if (!h || !ms || !flops || !cells) return fail(h, GPRX_EINVAL, "null argument");
  *ms = h->cell_ms;
  *flops = h->cell_ms > 0.0 ? h->cell_flops : 0.0;
  *cells = h->cell_ms > 0.0 ? h->cell_cells : 0.0;
  return GPRX_OK;
}

int gprx_objective_batch(gprx_handle h, int count, const int* units, const double* theta, const double* z, int mask, double* losses,
                         double* grads) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count < 0 || !units || !theta || !losses) return fail(h, GPRX_EINVAL, "null argument");
  const int64_t gw = h->ntheta + h->m * h->d;
  if (h->m == 0 && count > 1 && h->d <= CELL_PAR - CELL_PAR_LS) {
    // exact models: every stage once for all cells (batched launches), results identical to the loop below
    std::vector<Theta> ts(count);
    for (int i = 0; i < count; ++i) {
      if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
      for (int k = 0; k < h->ntheta; ++k)
        if (!std::isfinite(theta[(int64_t)i * h->ntheta + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
      ts[i] = decode_theta(h, theta + (int64_t)i * h->ntheta);
    }
    std::vector<double> lml(count);
    static const bool fused_eval = !(getenv("GPRX_FUSED_EVAL") && atoi(getenv("GPRX_FUSED_EVAL")) == 0);
    const bool form_alpha = grads && fused_eval;  // (as gprx_objective: alpha from the gradient's inverse, same kernels -> same bits)
    const int frc = exact_factorize_batch(h, count, units, ts.data(), lml.data(), nullptr, !form_alpha);
    if (frc != GPRX_OK && frc != GPRX_ENOTPD) return frc;
    for (int i = 0; i < count; ++i) losses[i] = -(lml[i] + log_prior(h, ts[i], mask));
    if (grads) {
      std::vector<double> g((size_t)count * h->ntheta, 0.0);
      if ((rc = exact_gradient_batch(h, count, g.data(), form_alpha))) {
        if (form_alpha)  // (the slots hold factors without their alpha: nothing may predict from them)
          for (int i = 0; i < count; ++i) {
            h->slot_ok[i] = 0;
            h->slot_unit[i] = -1;
          }
        return rc;
      }
      for (int i = 0; i < count; ++i) {
        double* gi = grads + (int64_t)i * gw;
        if (h->slot_ok[i]) {
          chain_rule(h, ts[i], mask, g.data() + (size_t)i * h->ntheta, gi);
        } else {
          for (int k = 0; k < h->ntheta; ++k) gi[k] = std::numeric_limits<double>::quiet_NaN();
        }
      }
    }
    return frc;
  }
  static const bool no_sparse_batch = getenv("GPRX_NO_SPARSE_BATCH") != nullptr;  // escape hatch: one model after the other
  // (count == 1 too, round 4: the batched evaluation is 21 launches replayed from a graph, the single-model path ~40 launches and a dozen
  // copies -- 0.30 against 0.42 ms per evaluation at N = 4096, M = 50; same values bit for bit, as for every other count)
  if (h->m != 0 && count >= 1 && h->d <= CELL_PAR - CELL_PAR_LS && !no_sparse_batch) {
    if (!z) return fail(h, GPRX_EINVAL, "z (inducing inputs) is null for a sparse model");
    const int64_t nz = h->m * h->d;
    for (int64_t e = 0; e < (int64_t)count * nz; ++e)
      if (!std::isfinite(z[e])) return fail(h, GPRX_EINVAL, "z is not finite");
    std::vector<Theta> ts(count);
    for (int i = 0; i < count; ++i) {
      if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
      for (int k = 0; k < h->ntheta; ++k)
        if (!std::isfinite(theta[(int64_t)i * h->ntheta + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
      ts[i] = decode_theta(h, theta + (int64_t)i * h->ntheta);
    }
    std::vector<double> elbo(count), g(grads ? (size_t)count * h->ntheta : 0), gzv(grads ? (size_t)count * nz : 0);
    std::vector<int> st(count);
    const int frc = sgpr_objective_batch(h, count, units, ts.data(), z, elbo.data(), grads ? g.data() : nullptr, grads ? gzv.data() : nullptr, st.data());
    if (frc != GPRX_OK && frc != GPRX_ENOTPD) return frc;
    for (int i = 0; i < count; ++i) {
      losses[i] = -(elbo[i] + log_prior(h, ts[i], mask));  // NaN for a failed cell
      if (!grads) continue;
      double* gi = grads + (int64_t)i * gw;
      if (st[i] != GPRX_OK) {
        for (int64_t k = 0; k < gw; ++k) gi[k] = std::numeric_limits<double>::quiet_NaN();
        continue;
      }
      chain_rule(h, ts[i], mask, g.data() + (size_t)i * h->ntheta, gi);
      double* gzi = gi + h->ntheta;
      for (int64_t e = 0; e < nz; ++e) gzi[e] = (mask & GPRX_TRAIN_Z) ? -gzv[(size_t)i * nz + e] : 0.0;
    }
    return frc;
  }
  // one cell after the other (a single cell, d > 64, or GPRX_NO_SPARSE_BATCH): same contract as the batched paths -- a cell
  // whose matrix is not positive definite gets NaN, the others are still evaluated, the first failure is returned
  int first_error = GPRX_OK;
  for (int i = 0; i < count; ++i) {
    rc = objective_impl(h, units[i], theta + (int64_t)i * h->ntheta, z ? z + (int64_t)i * h->m * h->d : nullptr, mask, losses + i,
                        grads ? grads + (int64_t)i * gw : nullptr);
    if (rc == GPRX_ENOTPD) {
      losses[i] = std::numeric_limits<double>::quiet_NaN();
      if (grads)
        for (int64_t k = 0; k < gw; ++k) grads[(int64_t)i * gw + k] = std::numeric_limits<double>::quiet_NaN();
      if (!first_error) first_error = rc;
      continue;
    }
    if (rc) return rc;
  }
  return first_error;
}

static constexpr int PRED_TILE = 8192;
// test points per pass of ONE exact model's predict: the N x tile blocks Ks and V = L^-1 Ks stay at the 268 MB each they have at N = 4096,
// so a small model takes more points per pass (N = 1024: 13 passes of 7 692 points with four launches each spent 21 % of the predict outside
// the product; per-point results do not depend on the pass they fall into)
static inline int pred_tile_for(int64_t np) { return np <= 1024 ? 4 * PRED_TILE : (np <= 2048 ? 2 * PRED_TILE : PRED_TILE); }

// what one exact factorisation contributes to a prediction through the explicit inverse: alpha, L^-1, the kernel's
// hyperparameters (a device lengthscale vector, or a row of the cell-parameter table) and the variance offset
struct ExactPredictSrc {
  const double* alpha;
  const double* Xinv;
  const double* ls_dev;
  const double* cell_par;  // row of the batch's cell-parameter table ([0] variance, [8..] lengthscales) or nullptr
  double variance, base;   // base = variance (+ noise for predict_y)
};

// transposed formulation, test points along the rows: Kst = k(Xs, X) (ts x np), mean = Kst alpha (row dots),
// Vt = Kst L^-T as an NT GEMM -- both operands k-contiguous, the GEMM kernel's fastest case; op(B) = L^-T is
// upper triangular, so the K range of a tile ends at its last column and every tile row mixes short and long
// tiles (no tail of long tiles) -- and var = base - row sums of Vt^2.  h->Ks holds 2 x np x tile doubles.
static int exact_predict_inverse(gprx_handle h, const ExactPredictSrc& src, const double* xs_dev, int64_t ns, double* mean_dev, double* var_dev,
                                 int tile) {
  const int np = (int)h->np;
  const int64_t ld = h->np;
  hipStream_t st = h->stream;
  double* Vbuf = h->Ks.p + (size_t)h->np * tile;
  for (int64_t t0 = 0; t0 < ns; t0 += tile) {
    const int ts = (int)std::min<int64_t>(tile, ns - t0);
    const int tsp = (int)round_up(ts, NB);
    KmatArgs ka{xs_dev + t0 * h->d, h->X.p, src.ls_dev, h->Ks.p, ld, ts, (int)h->n, h->d, tsp, np, src.variance, 0.0, 0, 0.0, nullptr, 0};
    if (src.cell_par) {  // hyperparameters of a batch slot: straight from the device table (no upload, no synchronisation)
      ka.cell_par = src.cell_par;
      ka.diag_const = 1;
    }
    HIPCHK(h, launch_kmat(st, h->kid, with_form(ka, h)));
    hipLaunchKernelGGL(rowreduce_kernel, dim3((ts + 3) / 4), dim3(256), 0, st, (const double*)h->Ks.p, ld, src.alpha, ts, np, 0.0, 1.0, mean_dev + t0);
    // Vt is never stored: the GEMM's epilogue leaves the row sums of squares of its tiles (2 slabs per tile column), which
    // the final kernel adds in a fixed order -- 2 x 8 np tile bytes less HBM traffic per tile than storing and re-reading Vt
    // 64 x 64 tiles: operands by LDS-DMA and finer clipping of the triangular K range (measured at N = 4096, 100 000 points:
    // 3.65 M points/s = 61.2 TFLOP/s against 3.35 M with the 128 x 128 register-staged kernel)
    static const int ptile = getenv("GPRX_PREDICT_TILE") ? atoi(getenv("GPRX_PREDICT_TILE")) : 64;
    const int nparts = 2 * ((np + ptile - 1) / ptile);
    HIPCHK(h, launch_gemm(st, 0, 1, tsp, np, np, 1.0, h->Ks.p, ld, src.Xinv, ld, 0.0, Vbuf, ld, GEMM_B_UPPER, ptile, 1, 0, 0, 0, 1, 0, 0, 0, nullptr, 0, 0,
                          Vbuf, (int64_t)tile));
    hipLaunchKernelGGL(rowsq_final_kernel, dim3((ts + 255) / 256), dim3(256), 0, st, (const double*)Vbuf, nparts, (int64_t)tile, ts, src.base, var_dev + t0);
  }
  HIPCHK(h, hipGetLastError());
  return GPRX_OK;
}

int gprx_predict_dev(gprx_handle h, const double* xs_dev, int64_t ns, double* mean_dev, double* var_dev, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (!h->factorized) return fail(h, GPRX_ESTATE, "gprx_predict before a successful gprx_factorize / gprx_objective");
  if (ns < 0 || (ns > 0 && (!xs_dev || !mean_dev || !var_dev))) return fail(h, GPRX_EINVAL, "null argument");
  hipStream_t st = h->stream;
  const int rows_per_chunk = 256;
  if (h->m != 0 && h->sparse_view) return sgpr_predict_batch(h, 1, xs_dev, ns, mean_dev, var_dev, include_noise);  // cell block 0
  if (h->m != 0) {
    // gpflow SGPR.predict_f: tmp1 = L^-1 Kus, tmp2 = LB^-1 tmp1, mean = tmp2^T c,
    // var = v + colsum(tmp2^2) - colsum(tmp1^2) (+ s for predict_y)
    const int mp = (int)h->mp, m = (int)h->m;
    const int tile = (int)std::min<int64_t>(PRED_TILE, round_up(ns, NB));
    if ((rc = ensure(h, h->Ks, sizeof(double) * (size_t)mp * tile))) return rc;
    const int nchunks = (mp + rows_per_chunk - 1) / rows_per_chunk;
    if ((rc = ensure(h, h->pred, sizeof(double) * (size_t)nchunks * tile))) return rc;
    const double base = h->variance + (include_noise ? h->noise : 0.0);
    const double* cvec = h->Bm.p + (size_t)mp * mp;
    for (int64_t t0 = 0; t0 < ns; t0 += tile) {
      const int ts = (int)std::min<int64_t>(tile, ns - t0);
      const int tsp = (int)round_up(ts, NB);
      KmatArgs ka{h->Z.p, xs_dev + t0 * h->d, h->invls.p, h->Ks.p, tile, m, ts, h->d, mp, tsp, h->variance, 0.0, 0, 0.0, nullptr, 0};
      HIPCHK(h, launch_kmat(st, h->kid, with_form(ka, h)));
      dim3 grid((ts + 255) / 256, nchunks);
      HIPCHK(h, trsm_lower_left(st, h->Qm.p, mp, h->invDL.p, h->Ks.p, tile, mp, tsp));
      hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, (const double*)nullptr, mp, ts, rows_per_chunk,
                         h->pred.p);
      hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, base, -1.0, 0, var_dev + t0);
      HIPCHK(h, trsm_lower_left(st, h->Bm.p, mp, h->invDB.p, h->Ks.p, tile, mp, tsp));
      hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, cvec, mp, ts, rows_per_chunk, h->pred.p);
      hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, 0.0, 1.0, 0, mean_dev + t0);
      hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, (const double*)nullptr, mp, ts, rows_per_chunk,
                         h->pred.p);
      hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, 0.0, 1.0, 1, var_dev + t0);
    }
    HIPCHK(h, hipGetLastError());
    return GPRX_OK;
  }
  const int np = (int)h->np;
  const int64_t ld = h->np;
  const int tile = (int)std::min<int64_t>(pred_tile_for(h->np), round_up(ns, NB));
  // Many test points: V = L^-1 Ks as ONE triangular GEMM per tile against the explicit inverse (computed once
  // per factorisation, N^3/3 flops amortised over N* >= 2 N points) instead of the recursive solve's ~2 N/64
  // dependent launches per tile.  Few points: blocked forward substitution on L itself.
  // Measured at N = 4096 (tools/predict_sizes.py): the substitution path costs ~1.7 ms whatever the batch (2 N / 64
  // dependent launches), the inverse path 0.9 ms for L^-1 plus 0.3 us per point -- faster for every batch size; at larger
  // N the N^3 / 3 flops of L^-1 only pay from about N / 2 points on.  predict_path (gprx_set_tuning): 1 / 2 force a path.
  const int forced = h->predict_path;
  const bool use_inverse = forced == 1 || (forced != 2 && (h->have_linv || h->n <= 4096 || 2 * ns >= (int64_t)h->n));
  if ((rc = ensure(h, h->Ks, sizeof(double) * h->np * tile * (use_inverse ? 2 : 1)))) return rc;
  double* Vbuf = h->Ks.p + (use_inverse ? (size_t)h->np * tile : 0);
  if (use_inverse && !h->have_linv) {
    if ((rc = ensure(h, h->Xinv, sizeof(double) * h->np * ld))) return rc;
    if ((rc = ensure(h, h->Tmp, sizeof(double) * h->np * ld))) return rc;
    HIPCHK(h, hipMemsetAsync(h->Xinv.p, 0, sizeof(double) * h->np * ld, st));
    HIPCHK(h, trtri_lower(st, h->Kmat.p, ld, h->invD.p, h->Xinv.p, ld, h->Tmp.p, ld, np));
    h->have_linv = true;
  }
  const double base = h->variance + (include_noise ? h->noise : 0.0);
  if (use_inverse) {
    const ExactPredictSrc src{h->alpha.p, h->Xinv.p, h->invls.p, nullptr, h->variance, base};
    return exact_predict_inverse(h, src, xs_dev, ns, mean_dev, var_dev, tile);
  }
  const int nchunks = (np + rows_per_chunk - 1) / rows_per_chunk;
  if ((rc = ensure(h, h->pred, sizeof(double) * (size_t)nchunks * tile))) return rc;
  for (int64_t t0 = 0; t0 < ns; t0 += tile) {
    const int ts = (int)std::min<int64_t>(tile, ns - t0);
    const int tsp = (int)round_up(ts, NB);
    KmatArgs ka{h->X.p, xs_dev + t0 * h->d, h->invls.p, h->Ks.p, tile, (int)h->n, ts, h->d, np, tsp, h->variance, 0.0, 0, 0.0, nullptr, 0};
    HIPCHK(h, launch_kmat(st, h->kid, with_form(ka, h)));
    dim3 grid((ts + 255) / 256, nchunks);
    hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, h->alpha.p, np, ts, rows_per_chunk, h->pred.p);
    hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, 0.0, 1.0, 0, mean_dev + t0);
    HIPCHK(h, trsm_lower_left(st, h->Kmat.p, ld, h->invD.p, h->Ks.p, tile, np, tsp));
    hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, (const double*)h->Ks.p, (int64_t)tile, (const double*)nullptr, np, ts, rows_per_chunk,
                       h->pred.p);
    hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, base, -1.0, 0, var_dev + t0);
  }
  HIPCHK(h, hipGetLastError());
  return GPRX_OK;
}

int gprx_predict(gprx_handle h, const double* xs, int64_t ns, double* mean, double* var, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (ns < 0 || (ns > 0 && (!xs || !mean || !var))) return fail(h, GPRX_EINVAL, "null argument");
  if (ns == 0) return GPRX_OK;
  if ((rc = ensure(h, h->xs, sizeof(double) * (ns * h->d + 2 * ns)))) return rc;
  double* dxs = h->xs.p;
  double* dmean = dxs + ns * h->d;
  double* dvar = dmean + ns;
  HIPCHK(h, hipMemcpyAsync(dxs, xs, sizeof(double) * ns * h->d, hipMemcpyHostToDevice, h->stream));
  if ((rc = gprx_predict_dev(h, dxs, ns, dmean, dvar, include_noise))) return rc;
  HIPCHK(h, hipMemcpyAsync(mean, dmean, sizeof(double) * ns, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, hipMemcpyAsync(var, dvar, sizeof(double) * ns, hipMemcpyDeviceToHost, h->stream));
  HIPCHK(h, wait_stream(h, h->stream));
  return GPRX_OK;
}

// ---- device memory helpers ---------------------------------------------------------------------
// Core of gprx_predict_batch: the test points are in device memory (xs_dev) and the results go to device memory
// (means_dev / vars_dev: (count, ns) row-major); asynchronous on the handle's stream after the batched factorisation.
static int predict_batch_core(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs_dev, int64_t ns,
                              double* means_dev, double* vars_dev, int include_noise) {
  int rc;
  static const bool no_sparse_batch = getenv("GPRX_NO_SPARSE_BATCH") != nullptr;
  hipStream_t st = h->stream;
  if (h->m == 0 && h->d <= CELL_PAR - CELL_PAR_LS) {
    // exact models: all factorisations by one batched launch sequence, then every slot predicts
    if ((rc = gprx_factorize_batch(h, count, units, thetas, 0, nullptr, nullptr))) return rc;
    if (ns == 0) return GPRX_OK;
    const bool use_inverse = h->predict_path != 2 && (h->n <= 4096 || 2 * ns >= (int64_t)h->n);
    if (!use_inverse) {
      for (int i = 0; i < count; ++i) {
        if ((rc = select_slot(h, i))) return rc;
        if ((rc = gprx_predict_dev(h, xs_dev, ns, means_dev + (int64_t)i * ns, vars_dev + (int64_t)i * ns, include_noise))) return rc;
      }
      return GPRX_OK;
    }
    // L^-1 of every slot by batched launches (trtri_lower with the cell index in its grids); each slot then predicts with
    // its alpha / L^-1 / row of the parameter table -- no per-cell upload or synchronisation
    const int np = (int)h->np;
    const int64_t ld = h->np, cs = h->cell_stride, gs = 2 * (int64_t)h->np * h->np;
    const int tile = (int)std::min<int64_t>(PRED_TILE, round_up(ns, NB));
    if ((rc = ensure(h, h->garena, sizeof(double) * (size_t)gs * count))) return rc;
    if ((rc = ensure(h, h->Ks, sizeof(double) * h->np * tile * 2))) return rc;
    for (int c = 0; c < count; ++c) HIPCHK(h, hipMemsetAsync(h->garena.p + (int64_t)c * gs, 0, sizeof(double) * h->np * ld, st));
    HIPCHK(h, trtri_lower(st, h->arena.p, ld, h->arena.p + h->off_invd, h->garena.p, ld, h->garena.p + (int64_t)np * ld, ld, np, count, cs, gs,
                          h->tune.update_tile ? h->tune.update_tile : 64));
    for (int i = 0; i < count; ++i) {
      const Theta& t = h->slot_theta[i];
      const double* cpar = h->cellpar.p + (int64_t)i * CELL_PAR;
      const ExactPredictSrc src{h->arena.p + (int64_t)i * cs + h->off_alpha, h->garena.p + (int64_t)i * gs, nullptr, cpar, t.variance,
                                t.variance + (include_noise ? t.noise : 0.0)};
      if ((rc = exact_predict_inverse(h, src, xs_dev, ns, means_dev + (int64_t)i * ns, vars_dev + (int64_t)i * ns, tile))) return rc;
    }
    h->have_linv = false;  // the single-cell views of the handle may point into the arena: their cached L^-1 is not this batch's
    return GPRX_OK;
  }
  if (h->m != 0 && count > 1 && h->d <= CELL_PAR - CELL_PAR_LS && !no_sparse_batch) {
    // sparse models (what gpras runs): every cell factorised by ONE batched launch sequence, then one batched predict
    if (!z) return fail(h, GPRX_EINVAL, "z (inducing inputs) is null for a sparse model");
    const int64_t nz = h->m * h->d;
    for (int64_t e = 0; e < (int64_t)count * nz; ++e)
      if (!std::isfinite(z[e])) return fail(h, GPRX_EINVAL, "z is not finite");
    std::vector<Theta> ts(count);
    for (int i = 0; i < count; ++i) {
      if (units[i] < 0 || units[i] >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
      for (int k = 0; k < h->ntheta; ++k)
        if (!std::isfinite(thetas[(int64_t)i * h->ntheta + k])) return fail(h, GPRX_EINVAL, "theta is not finite");
      ts[i] = decode_theta(h, thetas + (int64_t)i * h->ntheta);
    }
    std::vector<double> elbo(count);
    std::vector<int> stv(count);
    if ((rc = sgpr_objective_batch(h, count, units, ts.data(), z, elbo.data(), nullptr, nullptr, stv.data()))) return rc;  // ENOTPD included
    if (ns == 0) return GPRX_OK;
    return sgpr_predict_batch(h, count, xs_dev, ns, means_dev, vars_dev, include_noise);
  }
  for (int i = 0; i < count; ++i) {
    if ((rc = objective_impl(h, units[i], thetas + (int64_t)i * h->ntheta, z ? z + (int64_t)i * h->m * h->d : nullptr, 0, nullptr, nullptr)))
      return rc;
    if ((rc = gprx_predict_dev(h, xs_dev, ns, means_dev + (int64_t)i * ns, vars_dev + (int64_t)i * ns, include_noise))) return rc;
  }
  return GPRX_OK;
}

int gprx_predict_batch_dev(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs_dev, int64_t ns,
                           double* means_dev, double* vars_dev, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count <= 0 || !units || !thetas || ns < 0 || (ns > 0 && (!xs_dev || !means_dev || !vars_dev))) return fail(h, GPRX_EINVAL, "null argument");
  return predict_batch_core(h, count, units, thetas, z, xs_dev, ns, means_dev, vars_dev, include_noise);
}

int gprx_predict_batch(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs, int64_t ns,
                       double* means, double* vars, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count <= 0 || !units || !thetas || ns < 0 || (ns > 0 && (!xs || !means || !vars))) return fail(h, GPRX_EINVAL, "null argument");
  // the test points go up ONCE; the (count, ns) results come back in slabs of cells so that the device staging stays small
  // at configs[3]'s size (100 000 points: 1.6 MB per cell)
  const int slab = (int)std::max<int64_t>(1, std::min<int64_t>(count, ((int64_t)1 << 27) / std::max<int64_t>(2 * ns, 1)));
  if ((rc = ensure(h, h->xs, sizeof(double) * (ns * h->d + 2 * ns * slab + 16)))) return rc;
  double* dxs = h->xs.p;
  double* dmean = dxs + ns * h->d;
  double* dvar = dmean + ns * slab;
  if (ns > 0) HIPCHK(h, hipMemcpyAsync(dxs, xs, sizeof(double) * ns * h->d, hipMemcpyHostToDevice, h->stream));
  for (int c0 = 0; c0 < count; c0 += slab) {
    const int cnt = std::min(slab, count - c0);
    if ((rc = predict_batch_core(h, cnt, units + c0, thetas + (int64_t)c0 * h->ntheta, z ? z + (int64_t)c0 * h->m * h->d : nullptr, dxs, ns, dmean, dvar,
                                 include_noise)))
      return rc;
    if (ns > 0) {
      HIPCHK(h, hipMemcpyAsync(means + (int64_t)c0 * ns, dmean, sizeof(double) * ns * cnt, hipMemcpyDeviceToHost, h->stream));
      HIPCHK(h, hipMemcpyAsync(vars + (int64_t)c0 * ns, dvar, sizeof(double) * ns * cnt, hipMemcpyDeviceToHost, h->stream));
    }
    HIPCHK(h, wait_stream(h, h->stream));
  }
  return GPRX_OK;
}

int gprx_predict_batch_t(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs, int64_t ns,
                         double* means_t, double* vars_t, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count <= 0 || !units || !thetas || ns < 0 || (ns > 0 && (!xs || !means_t || !vars_t))) return fail(h, GPRX_EINVAL, "null argument");
  // as gprx_predict_batch, with every slab of cells transposed on the device before it leaves: the host block is written in its final
  // (ns, count) layout -- whole when all cells fit one slab, by 2-D copies into the slab's columns otherwise
  int slab = (int)std::max<int64_t>(1, std::min<int64_t>(count, ((int64_t)1 << 26) / std::max<int64_t>(2 * ns, 1)));
  if (const char* e = getenv("GPRX_PREDICT_SLAB")) slab = std::max(1, std::min(count, atoi(e)));  // (tests: force the slab-by-slab copies)
  if ((rc = ensure(h, h->xs, sizeof(double) * (ns * h->d + 4 * ns * slab + 16)))) return rc;
  double* dxs = h->xs.p;
  double* dmean = dxs + ns * h->d;
  double* dvar = dmean + ns * slab;
  double* tmean = dvar + ns * slab;
  double* tvar = tmean + ns * slab;
  if (ns > 0) HIPCHK(h, hipMemcpyAsync(dxs, xs, sizeof(double) * ns * h->d, hipMemcpyHostToDevice, h->stream));
  for (int c0 = 0; c0 < count; c0 += slab) {
    const int cnt = std::min(slab, count - c0);
    if ((rc = predict_batch_core(h, cnt, units + c0, thetas + (int64_t)c0 * h->ntheta, z ? z + (int64_t)c0 * h->m * h->d : nullptr, dxs, ns, dmean, dvar,
                                 include_noise)))
      return rc;
    if (ns > 0) {
      const unsigned grid = (unsigned)std::min<int64_t>(((int64_t)cnt * ns + 255) / 256, 4096);
      hipLaunchKernelGGL(transpose_small_kernel, dim3(grid), dim3(256), 0, h->stream, (const double*)dmean, (int64_t)cnt, ns, tmean);
      hipLaunchKernelGGL(transpose_small_kernel, dim3(grid), dim3(256), 0, h->stream, (const double*)dvar, (int64_t)cnt, ns, tvar);
      HIPCHK(h, hipGetLastError());
      if (cnt == count) {
        HIPCHK(h, hipMemcpyAsync(means_t, tmean, sizeof(double) * ns * cnt, hipMemcpyDeviceToHost, h->stream));
        HIPCHK(h, hipMemcpyAsync(vars_t, tvar, sizeof(double) * ns * cnt, hipMemcpyDeviceToHost, h->stream));
      } else {
        HIPCHK(h, hipMemcpy2DAsync(means_t + c0, sizeof(double) * count, tmean, sizeof(double) * cnt, sizeof(double) * cnt, (size_t)ns, hipMemcpyDeviceToHost,
                                   h->stream));
        HIPCHK(h, hipMemcpy2DAsync(vars_t + c0, sizeof(double) * count, tvar, sizeof(double) * cnt, sizeof(double) * cnt, (size_t)ns, hipMemcpyDeviceToHost,
                                   h->stream));
      }
    }
    HIPCHK(h, wait_stream(h, h->stream));
  }
  return GPRX_OK;
}

// ---- EOF projection either side of the GP path ------------------------------
namespace {
int pfail(gprx_pca_handle p, int code, const std::string& msg) {
  if (p) p->err = msg;
  g_err = msg;
  return code;
}
#define PCACHK(p, expr)                                                                                              \
  do {                                                                                                               \
    hipError_t e_ = (expr);                                                                                          \
    if (e_ != hipSuccess)                                                                                            \
      return pfail(p, e_ == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
int pensure(gprx_pca_handle p, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return GPRX_OK;
  if (b.p) PCACHK(p, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  PCACHK(p, hipMalloc((void**)&b.p, bytes));
  b.bytes = bytes;
  return GPRX_OK;
}
int pupload(gprx_pca_handle p, Buf& b, const std::vector<double>& v) {
  int rc = pensure(p, b, sizeof(double) * v.size());
  if (rc) return rc;
  PCACHK(p, copy_sync(b.p, v.data(), sizeof(double) * v.size(), hipMemcpyHostToDevice));
  return GPRX_OK;
}
// device staging per pass of the host-buffer entry points: 1 GiB of x / output (GPRX_PCA_CHUNK_DOUBLES overrides, for tests)
int64_t pca_chunk_doubles() {
  static const int64_t v = getenv("GPRX_PCA_CHUNK_DOUBLES") ? atoll(getenv("GPRX_PCA_CHUNK_DOUBLES")) : ((int64_t)1 << 27);
  return v;
}
}  // namespace

int gprx_pca_create(int device, int64_t n_cells, int k, const unsigned char* dry, const double* elevations, const double* input_mean,
                    const double* weights, const double* eofs, const double* x_mean, const double* x_std, int depth_mode,
                    gprx_pca_handle* out) {
  if (!out) return pfail(nullptr, GPRX_EINVAL, "out is null");
  *out = nullptr;
  if (n_cells <= 0 || k <= 0 || k > 64) return pfail(nullptr, GPRX_EINVAL, "n_cells must be positive and 1 <= k <= 64");
  if (!input_mean || !eofs || !x_mean || !x_std) return pfail(nullptr, GPRX_EINVAL, "input_mean, eofs, x_mean, x_std must be non-null");
  int64_t n_dry = 0;
  if (dry)
    for (int64_t c = 0; c < n_cells; ++c) n_dry += dry[c] != 0;
  if (depth_mode && !elevations) return pfail(nullptr, GPRX_EINVAL, "depth mode needs the cell elevations");
  if (!depth_mode && n_dry > 0 && !elevations) return pfail(nullptr, GPRX_EINVAL, "always-dry cells are filled with their elevations: elevations is null");
  PCACHK(nullptr, hipSetDevice(device));
  gprx_pca_handle p = new gprx_pca_ctx();
  p->device = device;
  p->cells = n_cells;
  p->cells_p = round_up(n_cells, 16);
  p->k = k;
  p->depth = depth_mode ? 1 : 0;
  const int64_t n_wet = n_cells - n_dry, cp = p->cells_p;
  // expand the wet-cell parameters to the full cell axis: dry cells get weight 0 (forward) / 1 (reverse), E = 0 and the fill value
  std::vector<double> mu(cp, 0.0), wf(cp, 0.0), wr(cp, 1.0), el(cp, 0.0), base(cp, 0.0), E((size_t)k * cp, 0.0);
  int64_t j = 0;
  for (int64_t c = 0; c < n_cells; ++c) {
    if (elevations) el[c] = elevations[c];
    if (dry && dry[c]) {
      base[c] = depth_mode ? 0.0 : elevations[c];
      continue;
    }
    mu[c] = input_mean[j];
    wf[c] = weights ? weights[j] : 1.0;
    wr[c] = wf[c];
    base[c] = input_mean[j];
    for (int kk = 0; kk < k; ++kk) E[(size_t)kk * cp + c] = eofs[(size_t)kk * n_wet + j];
    ++j;
  }
  int rc = GPRX_OK;
  hipError_t e = hipStreamCreateWithFlags(&p->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete p;
    return pfail(nullptr, GPRX_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  std::vector<double> xm(x_mean, x_mean + k), xs(x_std, x_std + k);
  if ((rc = pupload(p, p->mu, mu)) || (rc = pupload(p, p->wfwd, wf)) || (rc = pupload(p, p->wrev, wr)) || (rc = pupload(p, p->elev, el)) ||
      (rc = pupload(p, p->base, base)) || (rc = pupload(p, p->E, E)) || (rc = pupload(p, p->xm, xm)) || (rc = pupload(p, p->xs, xs))) {
    gprx_pca_destroy(p);
    return rc;
  }
  *out = p;
  return GPRX_OK;
}

int gprx_pca_destroy(gprx_pca_handle p) {
  if (!p) return GPRX_OK;
  hipSetDevice(p->device);
  if (p->stream) hipStreamSynchronize(p->stream);
  for (Buf* b : {&p->mu, &p->wfwd, &p->wrev, &p->elev, &p->E, &p->base, &p->xm, &p->xs, &p->dX, &p->dZ, &p->ws, &p->dMean, &p->dVar, &p->dFull,
                 &p->dVfull})
    if (b->p) hipFree(b->p);
  if (p->stream) hipStreamDestroy(p->stream);
  delete p;
  return GPRX_OK;
}

// x_dev: (rows, ld) with ld = cells_p (padding columns may hold anything finite: their weight is 0); z_dev: (rows, k)
int gprx_pca_transform_dev(gprx_pca_handle p, const double* x_dev, int64_t rows, double* z_dev) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || (rows > 0 && (!x_dev || !z_dev))) return pfail(p, GPRX_EINVAL, "null argument");
  if (rows == 0) return GPRX_OK;
  if (rows > (1 << 30)) return pfail(p, GPRX_EINVAL, "too many rows in one call");
  PCACHK(p, hipSetDevice(p->device));
  hipStream_t st = p->stream;
  const int64_t cp = p->cells_p;
  // Z = ((g(X) - mu) w) E^T in ONE pass over X: the centring / weighting runs inside the GEMM's operand load
  // (gemm_f64_kernel AXF); M = rows, N = k, K = cells_p cut into slices so that a few thousand workgroups exist
  const int tiles_m = (int)((rows + 63) / 64);
  int nsplit = std::max(1, 2048 / tiles_m);
  int kchunk = (int)round_up((cp + nsplit - 1) / nsplit, 16);
  if (kchunk < 256) kchunk = 256;
  nsplit = (int)((cp + kchunk - 1) / kchunk);
  int rc;
  if ((rc = pensure(p, p->ws, sizeof(double) * (size_t)nsplit * rows * p->k))) return rc;
  PCACHK(p, launch_gemm_splitk_axf(st, (int)rows, p->k, (int)cp, x_dev, cp, p->E.p, cp, z_dev, p->k, p->ws.p, kchunk, p->mu.p, p->wfwd.p,
                                   p->depth ? p->elev.p : nullptr));
  hipLaunchKernelGGL(pca_standardize_kernel, dim3((unsigned)((rows * p->k + 255) / 256)), dim3(256), 0, st, z_dev, rows, p->k,
                     (const double*)p->xm.p, (const double*)p->xs.p);
  PCACHK(p, hipGetLastError());
  return GPRX_OK;
}

int gprx_pca_reverse_dev(gprx_pca_handle p, const double* mean_dev, const double* var_dev, int64_t rows, double* full_dev, double* vfull_dev) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || (rows > 0 && (!mean_dev || !full_dev))) return pfail(p, GPRX_EINVAL, "null argument");
  if ((var_dev == nullptr) != (vfull_dev == nullptr)) return pfail(p, GPRX_EINVAL, "var and var_full must both be given or both be null");
  if (rows == 0) return GPRX_OK;
  PCACHK(p, hipSetDevice(p->device));
  dim3 grid((unsigned)((p->cells + 255) / 256), (unsigned)std::min<int64_t>((rows + PCA_RB - 1) / PCA_RB, 64));
  if (p->k <= 16)
    hipLaunchKernelGGL(pca_reverse_kernel<16>, grid, dim3(256), 0, p->stream, mean_dev, var_dev, rows, p->k, p->cells, (const double*)p->E.p, p->cells_p,
                       (const double*)p->wrev.p, (const double*)p->base.p, (const double*)p->xm.p, (const double*)p->xs.p, full_dev, vfull_dev);
  else
    hipLaunchKernelGGL(pca_reverse_kernel<64>, grid, dim3(256), 0, p->stream, mean_dev, var_dev, rows, p->k, p->cells, (const double*)p->E.p, p->cells_p,
                       (const double*)p->wrev.p, (const double*)p->base.p, (const double*)p->xm.p, (const double*)p->xs.p, full_dev, vfull_dev);
  PCACHK(p, hipGetLastError());
  return GPRX_OK;
}

int gprx_pca_to_depth_dev(gprx_pca_handle p, double* field_dev, int64_t rows, int add_elevations_first) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || (rows > 0 && !field_dev)) return pfail(p, GPRX_EINVAL, "null argument");
  if (rows == 0) return GPRX_OK;
  PCACHK(p, hipSetDevice(p->device));
  hipLaunchKernelGGL(field_to_depth_kernel, dim3(4096), dim3(256), 0, p->stream, field_dev, rows, p->cells, (const double*)p->elev.p,
                     add_elevations_first ? 1 : 0);
  PCACHK(p, hipGetLastError());
  return GPRX_OK;
}

int gprx_pca_sqrt_dev(gprx_pca_handle p, double* field_dev, int64_t count) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (count < 0 || (count > 0 && !field_dev)) return pfail(p, GPRX_EINVAL, "null argument");
  if (count == 0) return GPRX_OK;
  PCACHK(p, hipSetDevice(p->device));
  hipLaunchKernelGGL(field_sqrt_kernel, dim3(4096), dim3(256), 0, p->stream, field_dev, count);
  PCACHK(p, hipGetLastError());
  return GPRX_OK;
}

int gprx_pca_transpose_dev(gprx_pca_handle p, const double* src_dev, int64_t rows, int64_t cols, double* dst_dev) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || cols < 0 || (rows * cols > 0 && (!src_dev || !dst_dev))) return pfail(p, GPRX_EINVAL, "null argument");
  if (rows * cols == 0) return GPRX_OK;
  PCACHK(p, hipSetDevice(p->device));
  hipLaunchKernelGGL(transpose_small_kernel, dim3((unsigned)std::min<int64_t>((rows * cols + 255) / 256, 4096)), dim3(256), 0, p->stream, src_dev, rows,
                     cols, dst_dev);
  PCACHK(p, hipGetLastError());
  return GPRX_OK;
}

int gprx_pca_synchronize(gprx_pca_handle p) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  PCACHK(p, hipStreamSynchronize(p->stream));
  return GPRX_OK;
}

int gprx_pca_transform(gprx_pca_handle p, const double* x, int64_t rows, double* z) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || (rows > 0 && (!x || !z))) return pfail(p, GPRX_EINVAL, "null argument");
  PCACHK(p, hipSetDevice(p->device));
  const int64_t cp = p->cells_p, chunk = std::max<int64_t>(64, pca_chunk_doubles() / cp);
  int rc;
  for (int64_t t0 = 0; t0 < rows; t0 += chunk) {
    const int64_t nr = std::min(chunk, rows - t0);
    if ((rc = pensure(p, p->dX, sizeof(double) * (size_t)nr * cp)) || (rc = pensure(p, p->dZ, sizeof(double) * (size_t)nr * p->k))) return rc;
    if (cp > p->cells)  // padding columns must be finite (their weight is 0, and 0 * NaN is not)
      PCACHK(p, hipMemset2DAsync(p->dX.p + p->cells, sizeof(double) * cp, 0, sizeof(double) * (cp - p->cells), nr, p->stream));
    PCACHK(p, hipMemcpy2DAsync(p->dX.p, sizeof(double) * cp, x + t0 * p->cells, sizeof(double) * p->cells, sizeof(double) * p->cells, nr,
                               hipMemcpyHostToDevice, p->stream));
    if ((rc = gprx_pca_transform_dev(p, p->dX.p, nr, p->dZ.p))) return rc;
    PCACHK(p, hipMemcpyAsync(z + t0 * p->k, p->dZ.p, sizeof(double) * nr * p->k, hipMemcpyDeviceToHost, p->stream));
    PCACHK(p, hipStreamSynchronize(p->stream));
  }
  return GPRX_OK;
}

int gprx_pca_reverse(gprx_pca_handle p, const double* mean, const double* var, int64_t rows, double* full, double* var_full) {
  if (!p) return pfail(p, GPRX_EINVAL, "null handle");
  if (rows < 0 || (rows > 0 && (!mean || !full))) return pfail(p, GPRX_EINVAL, "null argument");
  if ((var == nullptr) != (var_full == nullptr)) return pfail(p, GPRX_EINVAL, "var and var_full must both be given or both be null");
  PCACHK(p, hipSetDevice(p->device));
  const int64_t chunk = std::max<int64_t>(64, pca_chunk_doubles() / p->cells);
  int rc;
  for (int64_t t0 = 0; t0 < rows; t0 += chunk) {
    const int64_t nr = std::min(chunk, rows - t0);
    if ((rc = pensure(p, p->dMean, sizeof(double) * (size_t)nr * p->k)) || (rc = pensure(p, p->dFull, sizeof(double) * (size_t)nr * p->cells))) return rc;
    PCACHK(p, hipMemcpyAsync(p->dMean.p, mean + t0 * p->k, sizeof(double) * nr * p->k, hipMemcpyHostToDevice, p->stream));
    if (var) {
      if ((rc = pensure(p, p->dVar, sizeof(double) * (size_t)nr * p->k)) || (rc = pensure(p, p->dVfull, sizeof(double) * (size_t)nr * p->cells))) return rc;
      PCACHK(p, hipMemcpyAsync(p->dVar.p, var + t0 * p->k, sizeof(double) * nr * p->k, hipMemcpyHostToDevice, p->stream));
    }
    if ((rc = gprx_pca_reverse_dev(p, p->dMean.p, var ? p->dVar.p : nullptr, nr, p->dFull.p, var ? p->dVfull.p : nullptr))) return rc;
    PCACHK(p, hipMemcpyAsync(full + t0 * p->cells, p->dFull.p, sizeof(double) * nr * p->cells, hipMemcpyDeviceToHost, p->stream));
    if (var) PCACHK(p, hipMemcpyAsync(var_full + t0 * p->cells, p->dVfull.p, sizeof(double) * nr * p->cells, hipMemcpyDeviceToHost, p->stream));
    PCACHK(p, hipStreamSynchronize(p->stream));
  }
  return GPRX_OK;
}

const char* gprx_pca_last_error(gprx_pca_handle p) { return p ? p->err.c_str() : g_err.c_str(); }

// ---- fused error metrics over reconstructed fields ----------------------------
int gprx_metrics_dev(int device, const double* x_dev, const double* y_dev, const double* conf_dev, int64_t rows, int64_t cells, int t_tol,
                     double v_tol, double* row_sums_dev, double* cell_sums_dev, int* cell_arg_dev, unsigned long long* matches) {
  if (!x_dev || !y_dev || !row_sums_dev || !cell_sums_dev || !cell_arg_dev || !matches) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (rows <= 0 || cells <= 0 || rows > (1 << 30)) return fail(nullptr, GPRX_EINVAL, "rows and cells must be positive");
  if (t_tol < 0 || t_tol > MET_TMAX) return fail(nullptr, GPRX_EINVAL, "t_tol must be between 0 and 8");
  HIPCHK(nullptr, hipSetDevice(device));
  const int nwg = (int)((cells + 255) / 256);
  unsigned long long* match_partial = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&match_partial, sizeof(unsigned long long) * nwg));
  MetricsArgs a{x_dev, y_dev, conf_dev, rows, cells, v_tol, t_tol, cell_sums_dev, cell_sums_dev + cells, cell_sums_dev + 2 * cells,
                cell_sums_dev + 3 * cells, cell_sums_dev + 4 * cells, cell_arg_dev, cell_arg_dev + cells, match_partial};
  hipLaunchKernelGGL(metrics_cells_kernel, dim3(nwg), dim3(256), 0, util_stream(), a);
  hipLaunchKernelGGL(metrics_rows_kernel, dim3((unsigned)rows), dim3(256), 0, util_stream(), x_dev, y_dev, conf_dev, cells, row_sums_dev);
  std::vector<unsigned long long> hm(nwg);
  hipError_t e = copy_sync(hm.data(), match_partial, sizeof(unsigned long long) * nwg, hipMemcpyDeviceToHost);  // synchronises
  hipFree(match_partial);
  HIPCHK(nullptr, e);
  unsigned long long total = 0;
  for (auto v : hm) total += v;
  *matches = total;
  return GPRX_OK;
}

int gprx_metrics(int device, const double* x, const double* y, const double* conf, int64_t rows, int64_t cells, int t_tol, double v_tol,
                 double* row_sums, double* cell_sums, int* cell_arg, unsigned long long* matches) {
  if (!x || !y || !row_sums || !cell_sums || !cell_arg || !matches) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (rows <= 0 || cells <= 0) return fail(nullptr, GPRX_EINVAL, "rows and cells must be positive");
  HIPCHK(nullptr, hipSetDevice(device));
  const size_t fb = sizeof(double) * (size_t)rows * cells;
  double *dx = nullptr, *dy = nullptr, *dc = nullptr, *drow = nullptr, *dcell = nullptr;
  int* darg = nullptr;
  auto cleanup = [&]() {
    for (void* q : {(void*)dx, (void*)dy, (void*)dc, (void*)drow, (void*)dcell, (void*)darg})
      if (q) hipFree(q);
  };
  hipError_t e = hipMalloc((void**)&dx, fb);
  if (e == hipSuccess) e = hipMalloc((void**)&dy, fb);
  if (e == hipSuccess && conf) e = hipMalloc((void**)&dc, fb);
  if (e == hipSuccess) e = hipMalloc((void**)&drow, sizeof(double) * rows * 4);
  if (e == hipSuccess) e = hipMalloc((void**)&dcell, sizeof(double) * cells * 5);
  if (e == hipSuccess) e = hipMalloc((void**)&darg, sizeof(int) * cells * 2);
  if (e == hipSuccess) e = copy_sync(dx, x, fb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = copy_sync(dy, y, fb, hipMemcpyHostToDevice);
  if (e == hipSuccess && conf) e = copy_sync(dc, conf, fb, hipMemcpyHostToDevice);
  if (e != hipSuccess) {
    cleanup();
    return fail(nullptr, e == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP, std::string("gprx_metrics staging: ") + hipGetErrorString(e));
  }
  int rc = gprx_metrics_dev(device, dx, dy, dc, rows, cells, t_tol, v_tol, drow, dcell, darg, matches);
  if (rc == GPRX_OK) {
    e = copy_sync(row_sums, drow, sizeof(double) * rows * 4, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = copy_sync(cell_sums, dcell, sizeof(double) * cells * 5, hipMemcpyDeviceToHost);
    if (e == hipSuccess) e = copy_sync(cell_arg, darg, sizeof(int) * cells * 2, hipMemcpyDeviceToHost);
    if (e != hipSuccess) rc = fail(nullptr, GPRX_EHIP, std::string("gprx_metrics copy back: ") + hipGetErrorString(e));
  }
  cleanup();
  return rc;
}

// ---- k-means inducing-point initialisation: Lloyd iterations on the device ---------------
int gprx_kmeans_lloyd(int device, const double* x, int64_t n, int d, double* centers, int m, double tol, int max_iter, int32_t* labels,
                      int* n_iter, int* empty) {
  if (!x || !centers || !labels || !n_iter || !empty) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (n <= 0 || d <= 0 || d > 64 || m <= 0 || m > n || max_iter <= 0 || n > (1 << 30)) return fail(nullptr, GPRX_EINVAL, "need 0 < m <= n, 0 < d <= 64, max_iter > 0");
  HIPCHK(nullptr, hipSetDevice(device));
  double *dx = nullptr, *dc[2] = {nullptr, nullptr}, *dstat = nullptr;
  int* dlab = nullptr;
  auto cleanup = [&]() {
    for (void* q : {(void*)dx, (void*)dc[0], (void*)dc[1], (void*)dstat, (void*)dlab})
      if (q) hipFree(q);
  };
  const size_t cb = sizeof(double) * (size_t)m * d;
  hipError_t e = hipMalloc((void**)&dx, sizeof(double) * (size_t)n * d);
  if (e == hipSuccess) e = hipMalloc((void**)&dc[0], cb);
  if (e == hipSuccess) e = hipMalloc((void**)&dc[1], cb);
  if (e == hipSuccess) e = hipMalloc((void**)&dstat, sizeof(double) * (2 + m));
  if (e == hipSuccess) e = hipMalloc((void**)&dlab, sizeof(int) * n);
  if (e == hipSuccess) e = copy_sync(dx, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = copy_sync(dc[0], centers, cb, hipMemcpyHostToDevice);
  if (e == hipSuccess) e = memset_sync(dlab, 0xff, sizeof(int) * n);  // labels_old = -1 (_kmeans_single_lloyd)
  if (e != hipSuccess) {
    cleanup();
    return fail(nullptr, e == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP, std::string("gprx_kmeans_lloyd staging: ") + hipGetErrorString(e));
  }
  std::vector<double> stat(2 + m);
  const dim3 pgrid((unsigned)((n + 255) / 256));
  int cur = 0, it = 0;
  bool strict = false;
  *empty = 0;
  for (it = 0; it < max_iter; ++it) {
    // one iteration of lloyd_iter_chunked_dense: labels from the current centres, then the new centres and their shifts
    // the flags are cleared on the stream the two kernels run on (a non-blocking stream has no ordering with the legacy stream)
    e = hipMemsetAsync(dstat, 0, sizeof(double) * 2, util_stream());
    if (e != hipSuccess) break;
    hipLaunchKernelGGL(kmeans_assign_kernel, pgrid, dim3(256), 0, util_stream(), (const double*)dx, (int)n, d, (const double*)dc[cur], m, dlab, dstat);
    hipLaunchKernelGGL(kmeans_update_kernel, dim3(m), dim3(256), 0, util_stream(), (const double*)dx, (int)n, d, (const int*)dlab, (const double*)dc[cur],
                       dc[cur ^ 1], dstat);
    e = copy_sync(stat.data(), dstat, sizeof(double) * (2 + m), hipMemcpyDeviceToHost);  // synchronises
    if (e != hipSuccess) break;
    if (stat[1] != 0.0) {  // scikit-learn relocates empty clusters to far points; the caller falls back to it
      *empty = 1;
      break;
    }
    cur ^= 1;  // centers, centers_new = centers_new, centers
    if (stat[0] == 0.0) {  // labels equal labels_old: strict convergence
      strict = true;
      ++it;
      break;
    }
    double shift_tot = 0.0;
    for (int j = 0; j < m; ++j) shift_tot += stat[2 + j];
    if (shift_tot <= tol) {
      ++it;
      break;
    }
  }
  if (e == hipSuccess && !*empty && !strict) {
    // rerun the E-step so that the labels match the final centres
    hipLaunchKernelGGL(kmeans_assign_kernel, pgrid, dim3(256), 0, util_stream(), (const double*)dx, (int)n, d, (const double*)dc[cur], m, dlab, dstat);
  }
  if (e == hipSuccess) e = copy_sync(centers, dc[cur], cb, hipMemcpyDeviceToHost);
  if (e == hipSuccess) e = copy_sync(labels, dlab, sizeof(int) * n, hipMemcpyDeviceToHost);
  cleanup();
  HIPCHK(nullptr, e);
  *n_iter = it > max_iter ? max_iter : it;
  return GPRX_OK;
}

// k-means++ seeding on the device (kmeans.h): x (n, d) host, centred as scikit-learn centres it; xsq = row_norms(x, squared=True);
// first_id and uniforms ((m - 1) x trials) are the host's RandomState draws.  indices_out: m chosen point indices.
int gprx_kmeans_pp(int device, const double* x, int64_t n, int d, const double* xsq, int m, int trials, int64_t first_id, const double* uniforms,
                   int64_t* indices_out) {
  if (!x || !xsq || !indices_out || (m > 1 && !uniforms)) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (n <= 0 || d <= 0 || d > 64 || m <= 0 || m > n || trials <= 0 || trials > KPP_MAX_TRIALS || first_id < 0 || first_id >= n || n > (1 << 30))
    return fail(nullptr, GPRX_EINVAL, "need 0 < m <= n, 0 < d <= 64, 0 < trials <= 16, 0 <= first_id < n");
  HIPCHK(nullptr, hipSetDevice(device));
  hipStream_t us = util_stream();
  if (!us) return fail(nullptr, GPRX_EHIP, "no utility stream");
  const int nblocks = (int)((n + 255) / 256);
  double *dx = nullptr, *dsq = nullptr, *dbuf = nullptr, *dpart = nullptr, *duni = nullptr;
  KppState* dst = nullptr;
  long long* didx = nullptr;
  auto cleanup = [&]() {
    for (void* q : {(void*)dx, (void*)dsq, (void*)dbuf, (void*)dpart, (void*)duni, (void*)dst, (void*)didx})
      if (q) hipFree(q);
  };
  const size_t slab = sizeof(double) * (size_t)trials * n;  // one generation of candidate distance arrays
  hipError_t e = hipMalloc((void**)&dx, sizeof(double) * (size_t)n * d);
  if (e == hipSuccess) e = hipMalloc((void**)&dsq, sizeof(double) * n);
  if (e == hipSuccess) e = hipMalloc((void**)&dbuf, 2 * slab);
  if (e == hipSuccess) e = hipMalloc((void**)&dpart, sizeof(double) * (size_t)trials * nblocks);
  if (e == hipSuccess) e = hipMalloc((void**)&duni, sizeof(double) * (size_t)std::max(1, (m - 1) * trials));
  if (e == hipSuccess) e = hipMalloc((void**)&dst, 2 * sizeof(KppState));
  if (e == hipSuccess) e = hipMalloc((void**)&didx, sizeof(long long) * m);
  if (e == hipSuccess) e = hipMemcpyAsync(dx, x, sizeof(double) * (size_t)n * d, hipMemcpyHostToDevice, us);
  if (e == hipSuccess) e = hipMemcpyAsync(dsq, xsq, sizeof(double) * n, hipMemcpyHostToDevice, us);
  if (e == hipSuccess && m > 1) e = hipMemcpyAsync(duni, uniforms, sizeof(double) * (size_t)(m - 1) * trials, hipMemcpyHostToDevice, us);
  if (e != hipSuccess) {
    hipStreamSynchronize(us);
    cleanup();
    return fail(nullptr, e == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP, std::string("gprx_kmeans_pp staging: ") + hipGetErrorString(e));
  }
  double* gen[2] = {dbuf, dbuf + (size_t)trials * n};
  // distances to the first centre: generation 0, one "candidate"
  hipLaunchKernelGGL(kpp_dist_kernel, dim3(nblocks, 1), dim3(256), 0, us, (const double*)dx, (int)n, d, (const double*)dsq, (const KppState*)nullptr,
                     (const double*)nullptr, (int)first_id, gen[0], dpart, nblocks);
  int cur = 0, prev_trials = 1;
  for (int c = 1; c <= m; ++c) {
    // choose among the candidates of centre c - 1 (c == 1: the first centre itself); c < m: candidates of centre c
    const bool more = c < m;
    hipLaunchKernelGGL(kpp_select_kernel, dim3(1), dim3(256), 0, us, (int)n, (const double*)gen[cur], (const double*)dpart, nblocks, prev_trials,
                       c == 1 ? (const KppState*)nullptr : (const KppState*)(dst + ((c - 1) & 1)), (int)first_id, dst + (c & 1),
                       more ? (const double*)(duni + (size_t)(c - 1) * trials) : (const double*)nullptr, trials, didx + (c - 1));
    if (!more) break;
    hipLaunchKernelGGL(kpp_dist_kernel, dim3(nblocks, trials), dim3(256), 0, us, (const double*)dx, (int)n, d, (const double*)dsq,
                       (const KppState*)(dst + (c & 1)), (const double*)gen[cur], (int)first_id, gen[cur ^ 1], dpart, nblocks);
    cur ^= 1;
    prev_trials = trials;
  }
  std::vector<long long> idx(m);
  e = hipMemcpyAsync(idx.data(), didx, sizeof(long long) * m, hipMemcpyDeviceToHost, us);
  hipError_t e2 = hipStreamSynchronize(us);
  cleanup();
  HIPCHK(nullptr, e);
  HIPCHK(nullptr, e2);
  for (int c = 0; c < m; ++c) indices_out[c] = idx[c];
  return GPRX_OK;
}

int gprx_gather_rows(int device, const double* field_dev, int64_t rows, int64_t cells, const int64_t* idx, double* out) {
  if (!field_dev || !idx || !out) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (rows <= 0 || cells <= 0) return fail(nullptr, GPRX_EINVAL, "rows and cells must be positive");
  std::vector<int64_t> wrapped(idx, idx + cells);
  for (int64_t c = 0; c < cells; ++c) {
    if (wrapped[c] < -rows || wrapped[c] >= rows) {
      char msg[160];
      snprintf(msg, sizeof msg, "index %lld is out of bounds for axis 0 with size %lld", (long long)idx[c], (long long)rows);
      return fail(nullptr, GPRX_EINVAL, msg);
    }
    if (wrapped[c] < 0) wrapped[c] += rows;
  }
  HIPCHK(nullptr, hipSetDevice(device));
  int64_t* didx = nullptr;
  double* dout = nullptr;
  hipError_t e = hipMalloc((void**)&didx, sizeof(int64_t) * cells);
  if (e == hipSuccess) e = hipMalloc((void**)&dout, sizeof(double) * cells);
  if (e == hipSuccess) e = copy_sync(didx, wrapped.data(), sizeof(int64_t) * cells, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(gather_rows_kernel, dim3((unsigned)((cells + 255) / 256)), dim3(256), 0, util_stream(), field_dev, cells, (const int64_t*)didx, dout);
    e = copy_sync(out, dout, sizeof(double) * cells, hipMemcpyDeviceToHost);  // synchronises
  }
  if (didx) hipFree(didx);
  if (dout) hipFree(dout);
  HIPCHK(nullptr, e);
  return GPRX_OK;
}

int gprx_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes) {
  if (!free_bytes || !total_bytes) return fail(nullptr, GPRX_EINVAL, "null argument");
  HIPCHK(nullptr, hipSetDevice(device));
  size_t f = 0, t = 0;
  HIPCHK(nullptr, hipMemGetInfo(&f, &t));
  *free_bytes = (int64_t)f;
  *total_bytes = (int64_t)t;
  return GPRX_OK;
}

int gprx_cell_bytes(gprx_handle h, int with_gradient, int64_t* bytes) {
  if (!h || !bytes) return fail(h, GPRX_EINVAL, "null argument");
  if (h->m != 0) {
    *bytes = (int64_t)sizeof(double) * sgpr_batch_layout(h).ss;
    return GPRX_OK;
  }
  const int64_t np = h->np;
  int64_t doubles = round_up((np + NB) * np + np * NB + np * STAGE_LD + np, 64);  // arena cell (ensure_arena)
  if (with_gradient) doubles += 2 * np * np + (np / KM_T) * (np / KM_T) * (2 + h->d) + (2 + h->d);  // garena + trace partials
  *bytes = (int64_t)sizeof(double) * doubles;
  return GPRX_OK;
}

// ---- the one collective of the path: RCCL over xGMI -----------------------------------------
namespace {
int cfail(gprx_comm c, int code, const std::string& msg) {
  if (c) c->err = msg;
  g_err = msg;
  return code;
}
#define COMMHIP(c, expr)                                                                                               \
  do {                                                                                                                 \
    hipError_t e_ = (expr);                                                                                            \
    if (e_ != hipSuccess) return cfail(c, e_ == hipErrorOutOfMemory ? GPRX_ENOMEM : GPRX_EHIP, std::string(#expr) + ": " + hipGetErrorString(e_)); \
  } while (0)
#define COMMNCCL(c, expr)                                                                                              \
  do {                                                                                                                 \
    ncclResult_t r_ = (expr);                                                                                          \
    if (r_ != ncclSuccess) return cfail(c, GPRX_ERCCL, std::string(#expr) + ": " + rccl().GetErrorString(r_));         \
  } while (0)
int comm_scratch(gprx_comm c, size_t bytes) {
  if (c->scratch_bytes >= bytes) return GPRX_OK;
  if (c->scratch) COMMHIP(c, hipFree(c->scratch));
  c->scratch = nullptr;
  c->scratch_bytes = 0;
  COMMHIP(c, hipMalloc((void**)&c->scratch, bytes));
  c->scratch_bytes = bytes;
  return GPRX_OK;
}
}  // namespace

// RCCL is loaded only after this process has initialised HIP and seen its devices: loaded first (measured on the MI355X
// box: ncclGetUniqueId before any HIP call) it left the process with "no ROCm-capable device is detected".
static int comm_runtime_ready() {
  int count = 0;
  COMMHIP(nullptr, hipInit(0));
  COMMHIP(nullptr, hipGetDeviceCount(&count));
  if (count <= 0) return cfail(nullptr, GPRX_EHIP, "no device visible to this process");
  COMMHIP(nullptr, hipFree(nullptr));  // forces the runtime (context of the current device) into existence
  if (!rccl().load()) return cfail(nullptr, GPRX_ERCCL, rccl().error);
  return GPRX_OK;
}

int gprx_comm_runtime_check(int device) {
  COMMHIP(nullptr, hipSetDevice(device));
  return comm_runtime_ready();
}

int gprx_comm_unique_id(unsigned char* id128) {
  if (!id128) return cfail(nullptr, GPRX_EINVAL, "null argument");
  int rc0;
  if ((rc0 = comm_runtime_ready())) return rc0;
  ncclUniqueId id;
  COMMNCCL(nullptr, rccl().GetUniqueId(&id));
  static_assert(sizeof(id) == GPRX_UNIQUE_ID_BYTES, "ncclUniqueId size");
  std::memcpy(id128, &id, sizeof(id));
  return GPRX_OK;
}

int gprx_comm_init(int device, int rank, int world, const unsigned char* id128, gprx_comm* out) {
  if (!out) return cfail(nullptr, GPRX_EINVAL, "out is null");
  *out = nullptr;
  if (!id128 || world <= 0 || rank < 0 || rank >= world) return cfail(nullptr, GPRX_EINVAL, "bad rank / world / id");
  COMMHIP(nullptr, hipSetDevice(device));
  int rc0;
  if ((rc0 = comm_runtime_ready())) return rc0;
  gprx_comm c = new gprx_comm_ctx();
  c->device = device;
  c->rank = rank;
  c->world = world;
  hipError_t e = hipStreamCreateWithFlags(&c->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete c;
    return cfail(nullptr, GPRX_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  ncclUniqueId id;
  std::memcpy(&id, id128, sizeof(id));
  ncclResult_t r = rccl().CommInitRank(&c->comm, world, id, rank);  // collective: every rank of the job calls it
  if (r != ncclSuccess) {
    const std::string msg = std::string("ncclCommInitRank: ") + rccl().GetErrorString(r);
    hipStreamDestroy(c->stream);
    delete c;
    return cfail(nullptr, GPRX_ERCCL, msg);
  }
  *out = c;
  return GPRX_OK;
}

int gprx_comm_destroy(gprx_comm c) {
  if (!c) return GPRX_OK;
  hipSetDevice(c->device);
  if (c->stream) hipStreamSynchronize(c->stream);
  if (c->comm) rccl().CommDestroy(c->comm);
  if (c->scratch) hipFree(c->scratch);
  if (c->stream) hipStreamDestroy(c->stream);
  delete c;
  return GPRX_OK;
}

const char* gprx_comm_last_error(gprx_comm c) { return c ? c->err.c_str() : g_err.c_str(); }

int gprx_comm_rank(gprx_comm c, int* rank, int* world) {
  if (!c || !rank || !world) return cfail(c, GPRX_EINVAL, "null argument");
  *rank = c->rank;
  *world = c->world;
  // what RCCL itself reports for this communicator (ncclCommUserRank / ncclCommCount), so that "did RCCL see N ranks" does not rest
  // on the numbers the caller passed to gprx_comm_init
  if (rccl().CommUserRank) COMMNCCL(c, rccl().CommUserRank(c->comm, rank));
  if (rccl().CommCount) COMMNCCL(c, rccl().CommCount(c->comm, world));
  return GPRX_OK;
}

int gprx_comm_synchronize(gprx_comm c) {
  if (!c) return cfail(c, GPRX_EINVAL, "null communicator");
  COMMHIP(c, hipSetDevice(c->device));
  COMMHIP(c, hipStreamSynchronize(c->stream));
  return GPRX_OK;
}

int gprx_comm_all_gather(gprx_comm c, const double* send_dev, double* recv_dev, int64_t count) {
  if (!c || count < 0 || (count > 0 && (!send_dev || !recv_dev))) return cfail(c, GPRX_EINVAL, "null argument");
  if (count == 0) return GPRX_OK;
  COMMHIP(c, hipSetDevice(c->device));
  COMMNCCL(c, rccl().AllGather(send_dev, recv_dev, (size_t)count, ncclDouble, c->comm, c->stream));
  return GPRX_OK;
}

int gprx_comm_gather(gprx_comm c, const double* send_dev, double* recv_dev, int64_t count, int root) {
  if (!c || count < 0 || root < 0 || root >= c->world || (count > 0 && !send_dev)) return cfail(c, GPRX_EINVAL, "bad argument");
  if (c->rank == root && count > 0 && !recv_dev) return cfail(c, GPRX_EINVAL, "recv_dev is null on the root");
  if (count == 0) return GPRX_OK;
  COMMHIP(c, hipSetDevice(c->device));
  // one group: the root posts world - 1 receives (its own block is a device copy), every other rank one send; inside a
  // node all inbound xGMI links of the root are busy at once
  COMMNCCL(c, rccl().GroupStart());
  ncclResult_t r = ncclSuccess;
  if (c->rank == root) {
    for (int p = 0; p < c->world && r == ncclSuccess; ++p)
      if (p != root) r = rccl().Recv(recv_dev + (int64_t)p * count, (size_t)count, ncclDouble, p, c->comm, c->stream);
  } else {
    r = rccl().Send(send_dev, (size_t)count, ncclDouble, root, c->comm, c->stream);
  }
  const ncclResult_t r2 = rccl().GroupEnd();
  COMMNCCL(c, r);
  COMMNCCL(c, r2);
  if (c->rank == root && recv_dev + (int64_t)root * count != send_dev)
    COMMHIP(c, hipMemcpyAsync(recv_dev + (int64_t)root * count, send_dev, sizeof(double) * count, hipMemcpyDeviceToDevice, c->stream));
  return GPRX_OK;
}

int gprx_comm_all_reduce_max(gprx_comm c, double* buf_dev, int64_t count) {
  if (!c || count < 0 || (count > 0 && !buf_dev)) return cfail(c, GPRX_EINVAL, "null argument");
  if (count == 0) return GPRX_OK;
  COMMHIP(c, hipSetDevice(c->device));
  COMMNCCL(c, rccl().AllReduce(buf_dev, buf_dev, (size_t)count, ncclDouble, ncclMax, c->comm, c->stream));
  return GPRX_OK;
}

int gprx_comm_all_gather_host(gprx_comm c, const double* send, double* recv, int64_t count) {
  if (!c || count < 0 || (count > 0 && (!send || !recv))) return cfail(c, GPRX_EINVAL, "null argument");
  if (count == 0) return GPRX_OK;
  COMMHIP(c, hipSetDevice(c->device));
  int rc;
  if ((rc = comm_scratch(c, sizeof(double) * (size_t)count * (c->world + 1)))) return rc;
  double* dsend = c->scratch;
  double* drecv = c->scratch + count;
  COMMHIP(c, hipMemcpyAsync(dsend, send, sizeof(double) * count, hipMemcpyHostToDevice, c->stream));
  if ((rc = gprx_comm_all_gather(c, dsend, drecv, count))) return rc;
  COMMHIP(c, hipMemcpyAsync(recv, drecv, sizeof(double) * count * c->world, hipMemcpyDeviceToHost, c->stream));
  COMMHIP(c, hipStreamSynchronize(c->stream));
  return GPRX_OK;
}

int gprx_comm_barrier(gprx_comm c) {
  if (!c) return cfail(c, GPRX_EINVAL, "null communicator");
  int rc;
  COMMHIP(c, hipSetDevice(c->device));
  if ((rc = comm_scratch(c, sizeof(double) * (size_t)(c->world + 1)))) return rc;
  COMMHIP(c, hipMemsetAsync(c->scratch, 0, sizeof(double), c->stream));
  if ((rc = gprx_comm_all_reduce_max(c, c->scratch, 1))) return rc;
  COMMHIP(c, hipStreamSynchronize(c->stream));
  return GPRX_OK;
}

int gprx_dev_malloc(int device, int64_t bytes, void** out) {
  if (!out || bytes < 0) return fail(nullptr, GPRX_EINVAL, "bad argument");
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, hipMalloc(out, (size_t)std::max<int64_t>(bytes, 16)));
  return GPRX_OK;
}
int gprx_dev_free(int device, void* ptr) {
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, hipFree(ptr));
  return GPRX_OK;
}
int gprx_memcpy_h2d(int device, void* dst_dev, const void* src_host, int64_t bytes) {
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, copy_sync(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
  return GPRX_OK;
}
int gprx_memcpy_d2h(int device, void* dst_host, const void* src_dev, int64_t bytes) {
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, copy_sync(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
  return GPRX_OK;
}

// ---- building blocks ----------------------------------------------------------------------------
int gprx_kmat(int device, int kernel_id, const double* a_dev, int64_t n1, const double* b_dev, int64_t n2, int d,
              const double* ls_host, double variance, double diag_add, double* out_dev, int64_t ld, int64_t n1p, int64_t n2p,
              int mode) {
  if (!a_dev || !b_dev || !ls_host || !out_dev) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (n1p % NB || n2p % NB || ld % 2 || n1p < n1 || n2p < n2 || ld < n2p) return fail(nullptr, GPRX_EINVAL, "padded sizes must be multiples of 64");
  if (mode < 0 || mode > 6) return fail(nullptr, GPRX_EINVAL, "bad kernel id or mode");
  const int form = (mode & 4) ? GPRX_DIST_EXPANDED : GPRX_DIST_DIFFERENCE;  // mode + 4: gpflow's expanded distance form
  mode &= 3;
  if (kernel_id < 0 || kernel_id > 4 || mode < 0 || mode > 2) return fail(nullptr, GPRX_EINVAL, "bad kernel id or mode");
  HIPCHK(nullptr, hipSetDevice(device));
  double* dinv = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dinv, sizeof(double) * d));
  HIPCHK(nullptr, copy_sync(dinv, ls_host, sizeof(double) * d, hipMemcpyHostToDevice));
  KmatArgs ka{a_dev, b_dev, dinv, out_dev, ld, (int)n1, (int)n2, d, (int)n1p, (int)n2p, variance, diag_add, mode, mode ? 1.0 : 0.0, nullptr, 0};
  ka.form = form;
  hipError_t e = launch_kmat(util_stream(), kernel_id, ka);
  hipError_t e2 = hipStreamSynchronize(util_stream());
  hipFree(dinv);
  HIPCHK(nullptr, e);
  HIPCHK(nullptr, e2);
  return GPRX_OK;
}

// the kernel build's exponential (exp_nonpos_tab) / the gradient passes' (exp_nonpos) on an array: parity test of the function itself
namespace {
__global__ __launch_bounds__(256) void exp_probe_kernel(const double* __restrict__ x, double* __restrict__ out, int64_t n, int which) {
  __shared__ double sTab[64];
  exp_tab_fill(sTab);
  __syncthreads();
  for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
    out[i] = which == 0 ? exp_nonpos_tab(x[i], sTab) : exp_nonpos(x[i]);
}
}  // namespace

int gprx_exp_probe(int device, int which, const double* x, int64_t n, double* out) {
  if (!x || !out || n < 0 || which < 0 || which > 1) return fail(nullptr, GPRX_EINVAL, "bad argument");
  if (n == 0) return GPRX_OK;
  HIPCHK(nullptr, hipSetDevice(device));
  double *dx = nullptr, *dout = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dx, sizeof(double) * n));
  hipError_t e = hipMalloc((void**)&dout, sizeof(double) * n);
  if (e == hipSuccess) e = copy_sync(dx, x, sizeof(double) * n, hipMemcpyHostToDevice);
  if (e == hipSuccess) {
    hipLaunchKernelGGL(exp_probe_kernel, dim3(2048), dim3(256), 0, util_stream(), dx, dout, n, which);
    e = hipGetLastError();
  }
  if (e == hipSuccess) e = copy_sync(out, dout, sizeof(double) * n, hipMemcpyDeviceToHost);
  hipFree(dx);
  if (dout) hipFree(dout);
  HIPCHK(nullptr, e);
  return GPRX_OK;
}

int gprx_gemm(int device, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a_dev, int64_t lda,
              const double* b_dev, int64_t ldb, double beta, double* c_dev, int64_t ldc, int flags, int tile) {
  if (!a_dev || !b_dev || !c_dev) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (k % 16 || lda % 2 || ldb % 2) return fail(nullptr, GPRX_EINVAL, "k must be a multiple of 16, leading dimensions even");
  if (!((ta == 0 && tb == 1) || (ta == 0 && tb == 0) || (ta == 1 && tb == 0))) return fail(nullptr, GPRX_EINVAL, "unsupported transpose pair");
  if (tile != 0 && tile != 64 && tile != 128) return fail(nullptr, GPRX_EINVAL, "tile must be 0, 64 or 128");
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, launch_gemm(util_stream(), ta, tb, (int)m, (int)n, (int)k, alpha, a_dev, lda, b_dev, ldb, beta, c_dev, ldc, flags, tile));
  HIPCHK(nullptr, hipStreamSynchronize(util_stream()));
  return GPRX_OK;
}

int gprx_potrf(int device, double* a_dev, int64_t lda, int64_t np, int64_t extra, double* inv_diag_dev, int* info_host) {
  if (!a_dev || !inv_diag_dev || !info_host) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (np % NB || np <= 0 || extra < 0 || lda < np || lda % 2) return fail(nullptr, GPRX_EINVAL, "np must be a positive multiple of 64");
  HIPCHK(nullptr, hipSetDevice(device));
  int* dinfo = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dinfo, sizeof(int)));
  HIPCHK(nullptr, memset_sync(dinfo, 0, sizeof(int)));
  PotrfStreams ps;
  hipStream_t st = nullptr;
  HIPCHK(nullptr, hipStreamCreateWithFlags(&st, hipStreamNonBlocking));
  HIPCHK(nullptr, ps.init());
  double* dstage = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dstage, sizeof(double) * np * STAGE_LD));
  DagPlan dag;
  const bool by_dag = use_dag(potrf_tuning(), (int)np) && extra % NB == 0;
  hipError_t e = by_dag ? potrf_dag(st, a_dev, lda, (int)np, (int)extra, inv_diag_dev, dinfo, dag)
                        : potrf_lower(st, a_dev, lda, (int)np, (int)extra, inv_diag_dev, dinfo, dstage, nullptr, &ps);
  hipError_t e2 = hipStreamSynchronize(st);
  if (ps.aux && e2 == hipSuccess) e2 = hipStreamSynchronize(ps.aux);
  int gave_up = 0;
  if (by_dag && e == hipSuccess && e2 == hipSuccess) copy_sync(&gave_up, dag.state + DAG_ABORT, sizeof(int), hipMemcpyDeviceToHost);
  dag.destroy();
  hipFree(dstage);
  ps.destroy();
  hipStreamDestroy(st);
  copy_sync(info_host, dinfo, sizeof(int), hipMemcpyDeviceToHost);
  hipFree(dinfo);
  HIPCHK(nullptr, e);
  HIPCHK(nullptr, e2);
  if (gave_up) return fail(nullptr, GPRX_EHIP, "tile-DAG factorisation: a dependency wait timed out");
  return *info_host ? fail(nullptr, GPRX_ENOTPD, "matrix not positive definite") : GPRX_OK;
}

// development aid (GPRX_DAG_STAMPS=1): the stamps of the handle's last tile-DAG factorisation; returns the number of words
extern "C" int gprx_dag_stamps(gprx_handle h, unsigned long long* out, int max_words, int* T, int* grid) {
  if (!h || !h->dag.stamps) return 0;
  const int n = (int)std::min<size_t>(h->dag.stamp_words, (size_t)max_words);
  hipStreamSynchronize(h->stream);
  copy_sync(out, h->dag.stamps, sizeof(unsigned long long) * n, hipMemcpyDeviceToHost);
  if (T) *T = h->dag.T;
  if (grid) *grid = h->dag.grid;
  return n;
}

#ifdef GPRX_CHAIN_STAMPS
extern "C" int gprx_chain_stamps(unsigned long long* out16) {
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(gprx::g_chain_stamps), sizeof(unsigned long long) * 16);
  return 0;
}
#endif

#ifdef GPRX_PANEL_ACC
int gprx_panel_acc(unsigned long long* out8, int reset) {
  if (reset) {
    unsigned long long z[8] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(gprx::g_panel_acc), z, sizeof(z));
  }
  hipMemcpyFromSymbol(out8, HIP_SYMBOL(gprx::g_panel_acc), sizeof(unsigned long long) * 8);
  return 0;
}
#endif
#ifdef GPRX_CELL_ACC
int gprx_cell_acc(unsigned long long* out16, int reset) {
  if (reset) {
    unsigned long long z[16] = {0};
    hipMemcpyToSymbol(HIP_SYMBOL(gprx::g_cell_acc), z, sizeof(z));
  }
  hipMemcpyFromSymbol(out16, HIP_SYMBOL(gprx::g_cell_acc), sizeof(unsigned long long) * 16);
  return 0;
}
#endif
// Development aid: phase stamps (s_memtime, shader clocks) of workgroup (0, 0) of each of the five fused sparse kernels (sgpr_fused.h
// SF_STAMP) for the evaluations that follow enable = 1; out (may be null): SF_STAMP_WORDS words of the last evaluation.
int gprx_sf_stamps(gprx_handle h, int enable, unsigned long long* out) {
  if (!h) return fail(h, GPRX_EINVAL, "null handle");
  HIPCHK(h, hipSetDevice(h->device));
  HIPCHK(h, hipStreamSynchronize(h->stream));
  if (out && h->sf_stamps) HIPCHK(h, copy_sync(out, h->sf_stamps, sizeof(unsigned long long) * SF_STAMP_WORDS, hipMemcpyDeviceToHost));
  if (enable && !h->sf_stamps) {
    HIPCHK(h, hipMalloc((void**)&h->sf_stamps, sizeof(unsigned long long) * SF_STAMP_WORDS));
    HIPCHK(h, memset_sync(h->sf_stamps, 0, sizeof(unsigned long long) * SF_STAMP_WORDS));
    drop_graphs(h);
  } else if (!enable && h->sf_stamps) {
    HIPCHK(h, hipFree(h->sf_stamps));
    h->sf_stamps = nullptr;
    drop_graphs(h);
  }
  return GPRX_OK;
}
#ifdef GPRX_PANEL_STAMPS
int gprx_panel_stamps(unsigned long long* out64) {
  hipMemcpyFromSymbol(out64, HIP_SYMBOL(gprx::g_panel_stamps), sizeof(unsigned long long) * 64);
  return 0;
}
#endif

namespace {
bool apply_tuning(PotrfTuning& t, int& predict_path, int& fused, const std::string& k, int value) {
  if (k == "panel_width" && (value == 0 || value == 64 || value == 128)) t.panel_width = value;
  else if (k == "outer_block" && value >= 0 && value % 128 == 0) t.outer_block = value;
  else if (k == "update_tile" && (value == 0 || value == 64 || value == 128)) t.update_tile = value;
  else if (k == "no_lookahead") t.no_lookahead = value != 0;
  else if (k == "panel_rows" && (value == 0 || value == 128 || value == 256)) t.panel_rows = value;
  else if (k == "panel_occ" && (value == 0 || value == 2 || value == 3)) t.panel_occ = value;
  else if (k == "inblock" && (value == 0 || value == 1)) t.inblock = value;
  else if (k == "split_panel" && value >= -1 && value <= 1) t.split_panel = value;
  else if (k == "dag" && value >= -1 && value <= 1) t.dag = value;
  else if (k == "rhs_vector" && value >= -1 && value <= 1) t.rhs_vector = value;
  else if (k == "rows_inv" && value >= -1 && value <= 1) t.rows_inv = value;
  else if (k == "rows_inv_rt" && value >= 0 && value <= 2) t.rows_inv_rt = value;
  else if (k == "rows_inv_lone" && value >= 0 && value <= 1) t.rows_inv_lone = value;
  else if (k == "cell_kernel" && value >= -1 && value <= 1) t.cell_kernel = value;
  else if (k == "split_updates" && value >= 0 && value <= 1) t.split_updates = value;
  else if (k == "poison_workspace" && value >= 0 && value <= 1) t.poison_workspace = value;
  else if (k == "predict_path" && value >= 0 && value <= 2) predict_path = value;
  else if (k == "sgpr_fused" && value >= 0 && value <= 1) fused = value;
  else if (k == "wait_handover_us" && value >= 0) wait_handover_us() = value;  // (process-wide whichever entry point sets it)
  else if (k == "sgpr_groups_from" && value >= 0) sf_groups_from() = value;    // (process-wide; 0: the resident Adam loop never splits a batch into groups)
  else return false;
  return true;
}
}  // namespace

int gprx_set_tuning(const char* key, int value) {
  if (!key) return fail(nullptr, GPRX_EINVAL, "null key");
  if (!apply_tuning(potrf_tuning(), predict_path_tuning(), sgpr_fused_tuning(), key, value)) return fail(nullptr, GPRX_EINVAL, "unknown tuning key or bad value");
  return GPRX_OK;
}

int gprx_set_handle_tuning(gprx_handle h, const char* key, int value) {
  if (!h || !key) return fail(h, GPRX_EINVAL, "null argument");
  const int fused_before = h->sgpr_fused;
  if (!apply_tuning(h->tune, h->predict_path, h->sgpr_fused, key, value)) return fail(h, GPRX_EINVAL, "unknown tuning key or bad value");
  if (hipSetDevice(h->device) == hipSuccess && hipStreamSynchronize(h->stream) == hipSuccess) drop_graphs(h);  // captured with the old schedule
  if (h->sgpr_fused != fused_before) {  // the cell blocks of the two sparse schedules differ (sgpr_batch_layout): the arena is rebuilt on the next call
    h->sarena_slots = 0;
    h->factorized = h->sparse_view ? false : h->factorized;
    h->sparse_view = false;
  }
  return GPRX_OK;
}

int gprx_mfma_f64_peak(int device, double* tflops) {
  if (!tflops) return fail(nullptr, GPRX_EINVAL, "null argument");
  HIPCHK(nullptr, hipSetDevice(device));
  double* out = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&out, 64));
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  const int iters = 20000, blocks = 256 * 4;
  hipStream_t us = util_stream();
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, us, out, 100);
  hipEventRecord(e0, us);
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, us, out, iters);
  hipEventRecord(e1, us);
  hipError_t e = hipStreamSynchronize(us);
  float ms = 0.f;
  hipEventElapsedTime(&ms, e0, e1);
  hipEventDestroy(e0);
  hipEventDestroy(e1);
  hipFree(out);
  HIPCHK(nullptr, e);
  const double flops = (double)blocks * 4 /*waves*/ * iters * 4 /*mfma*/ * 2048.0;
  *tflops = flops / (ms * 1e-3) / 1e12;
  return GPRX_OK;
}

}  // extern "C"
