// libgprx: C ABI (include/gprx.h) over the gfx950 kernels in this directory.
// Host orchestration only: parameter transforms and priors (scalar math), buffer ownership,
// launch sequences.  No CPU fallback exists for any device stage.
#include "../../include/gprx.h"

#include <hip/hip_runtime.h>

#include <cmath>
#include <cstdio>
#include <cstring>
#include <string>
#include <vector>

#include "gemm_f64.h"
#include "gprx_common.h"
#include "grad.h"
#include "kmat.h"
#include "potrf.h"
#include "solve.h"

using namespace gprx;

namespace {

thread_local std::string g_err;

struct Buf {
  double* p = nullptr;
  size_t bytes = 0;
};

}  // namespace

struct gprx_ctx {
  int device = 0;
  hipStream_t stream = nullptr;
  bool own_stream = false;
  int64_t n = 0, m = 0, np = 0, mp = 0;
  int d = 0, kid = 0, ard = 0, nlen = 1, ntheta = 3, n_units = 0;
  std::string err;
  // data
  Buf X, Y, Z, invls, alpha, red, Kmat, invD, Xinv, Tmp, partial, xs, Ks, pred;
  int* info = nullptr;
  // current factorisation
  bool factorized = false;
  int cur_unit = -1;
  double variance = 1.0, noise = 1.0;
  std::vector<double> ls;
  double timings[4] = {0, 0, 0, 0};
  hipEvent_t ev[5] = {nullptr, nullptr, nullptr, nullptr, nullptr};
};

namespace {

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

int ensure(gprx_handle h, Buf& b, size_t bytes) {
  if (b.bytes >= bytes) return GPRX_OK;
  if (b.p) HIPCHK(h, hipFree(b.p));
  b.p = nullptr;
  b.bytes = 0;
  HIPCHK(h, hipMalloc((void**)&b.p, bytes));
  b.bytes = bytes;
  return GPRX_OK;
}

// ---- scalar transforms (gpflow positive() / LogNormal(0,1) priors; see oracle/transforms.py) ------
double softplus(double w) { return w > 0 ? w + std::log1p(std::exp(-w)) : std::log1p(std::exp(w)); }
double sigmoid(double w) { return 0.5 * (1.0 + std::tanh(0.5 * w)); }
double ln_logpdf(double u) {
  const double lu = std::log(u);
  return -lu - 0.5 * std::log(2.0 * M_PI) - 0.5 * lu * lu;
}
double ln_dlogpdf(double u) { return -(1.0 + std::log(u)) / u; }

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

struct Theta {
  double variance, noise;
  std::vector<double> ls;
  double w_var, w_noise;
  std::vector<double> w_len;
};

Theta decode_theta(gprx_handle h, const double* theta) {
  Theta t;
  t.w_var = theta[0];
  t.w_noise = theta[1 + h->nlen];
  t.variance = softplus(t.w_var);
  t.noise = NOISE_LOWER + softplus(t.w_noise);
  t.w_len.assign(theta + 1, theta + 1 + h->nlen);
  t.ls.resize(h->d);
  for (int k = 0; k < h->d; ++k) t.ls[k] = softplus(t.w_len[h->ard ? k : 0]);
  return t;
}

double log_prior(gprx_handle h, const Theta& t, int mask) {
  double lp = 0.0;
  if (mask & GPRX_TRAIN_VARIANCE) lp += ln_logpdf(t.variance);
  if (mask & GPRX_TRAIN_LENGTHSCALE)
    for (int k = 0; k < h->nlen; ++k) lp += ln_logpdf(t.ls[k]);
  if (mask & GPRX_TRAIN_NOISE) lp += ln_logpdf(t.noise);
  return lp;
}

int upload_inv_ls(gprx_handle h, const Theta& t) {
  std::vector<double> inv(h->d);
  for (int k = 0; k < h->d; ++k) inv[k] = 1.0 / t.ls[k];
  HIPCHK(h, hipMemcpyAsync(h->invls.p, inv.data(), sizeof(double) * h->d, hipMemcpyHostToDevice, h->stream));
  HIPCHK(h, hipStreamSynchronize(h->stream));  // `inv` is a stack temporary
  return GPRX_OK;
}

// ---- exact GP ------------------------------------------------------------------------------------
// K = k(X,X) + s I (lower tiles) with y appended as row np; potrf gives L and beta = L^-1 y in that
// row; alpha by the backward solve; red[0] = sum log diag L, red[1] = |beta|^2.
int exact_factorize(gprx_handle h, int unit, const Theta& t, double* lml_out) {
  const int np = (int)h->np;
  const int64_t ld = h->np;
  int rc;
  if ((rc = ensure(h, h->Kmat, sizeof(double) * (h->np + NB) * ld))) return rc;
  if ((rc = ensure(h, h->invD, sizeof(double) * h->np * NB))) return rc;
  if ((rc = ensure(h, h->alpha, sizeof(double) * h->np))) return rc;
  if ((rc = upload_inv_ls(h, t))) return rc;
  hipStream_t st = h->stream;
  HIPCHK(h, hipEventRecord(h->ev[0], st));
  KmatArgs ka{h->X.p, h->X.p, h->invls.p, h->Kmat.p, ld, (int)h->n, (int)h->n, h->d, np, np, t.variance, t.noise, 1, 1.0, 0};
  HIPCHK(h, launch_kmat(st, h->kid, ka));
  hipLaunchKernelGGL(set_rhs_rows_kernel, dim3(64), dim3(256), 0, st, h->Kmat.p + (int64_t)np * ld, ld, h->Y.p + (int64_t)unit * h->np,
                     (int)h->n, np, NB);
  HIPCHK(h, hipEventRecord(h->ev[1], st));
  HIPCHK(h, hipMemsetAsync(h->info, 0, sizeof(int), st));
  HIPCHK(h, potrf_lower(st, h->Kmat.p, ld, np, NB, h->invD.p, h->info));
  HIPCHK(h, hipEventRecord(h->ev[2], st));
  const double* beta = h->Kmat.p + (int64_t)np * ld;
  hipLaunchKernelGGL(copy_row_kernel, dim3((np + 255) / 256), dim3(256), 0, st, beta, h->alpha.p, np);
  hipLaunchKernelGGL(logdet_quad_kernel, dim3(1), dim3(256), 0, st, h->Kmat.p, ld, beta, np, h->red.p);
  HIPCHK(h, trsv_lower(st, h->Kmat.p, ld, h->invD.p, h->alpha.p, np, true));
  HIPCHK(h, hipEventRecord(h->ev[3], st));
  double red[2];
  int info = 0;
  HIPCHK(h, hipMemcpyAsync(red, h->red.p, sizeof(red), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipMemcpyAsync(&info, h->info, sizeof(int), hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  if (info != 0) {
    h->factorized = false;
    char msg[128];
    snprintf(msg, sizeof msg, "matrix not positive definite: pivot %d", info);
    return fail(h, GPRX_ENOTPD, msg);
  }
  h->factorized = true;
  h->cur_unit = unit;
  h->variance = t.variance;
  h->noise = t.noise;
  h->ls = t.ls;
  if (lml_out) *lml_out = -0.5 * red[1] - red[0] - 0.5 * (double)h->n * std::log(2.0 * M_PI);
  return GPRX_OK;
}

// gradient of the LML w.r.t. constrained (variance, lengthscales[nlen], noise) -> g[0 .. nlen+1]
int exact_gradient(gprx_handle h, const Theta& t, double* g) {
  const int np = (int)h->np;
  const int64_t ld = h->np;
  int rc;
  if ((rc = ensure(h, h->Xinv, sizeof(double) * h->np * ld))) return rc;
  if ((rc = ensure(h, h->Tmp, sizeof(double) * h->np * ld))) return rc;
  hipStream_t st = h->stream;
  HIPCHK(h, hipMemsetAsync(h->Xinv.p, 0, sizeof(double) * h->np * ld, st));
  HIPCHK(h, trtri_lower(st, h->Kmat.p, ld, h->invD.p, h->Xinv.p, ld, h->Tmp.p, ld, np));
  // K^-1 = X^T X, lower tiles, into Tmp
  HIPCHK(h, launch_gemm(st, 1, 0, np, np, np, 1.0, h->Xinv.p, ld, h->Xinv.p, ld, 0.0, h->Tmp.p, ld,
                        GEMM_C_LOWER | GEMM_A_UPPER | GEMM_B_LOWER));
  const int tiles = np / KM_T;
  const int width = 2 + h->d;
  if ((rc = ensure(h, h->partial, sizeof(double) * ((size_t)tiles * tiles * width + width)))) return rc;
  TraceArgs ta{h->X.p, h->X.p, h->invls.p, h->Tmp.p, ld, h->alpha.p, (int)h->n, (int)h->n, h->d, t.variance, 1, h->partial.p, tiles};
  HIPCHK(h, launch_trace(st, h->kid, ta, tiles * tiles));
  double* sums = h->partial.p + (size_t)tiles * tiles * width;
  hipLaunchKernelGGL(trace_final, dim3(width), dim3(64), 0, st, h->partial.p, tiles * tiles, width, sums);
  std::vector<double> host(width);
  HIPCHK(h, hipMemcpyAsync(host.data(), sums, sizeof(double) * width, hipMemcpyDeviceToHost, st));
  HIPCHK(h, hipStreamSynchronize(st));
  g[0] = 0.5 * host[0];
  if (h->ard) {
    for (int k = 0; k < h->d; ++k) g[1 + k] = 0.5 * host[2 + k];
  } else {
    double s = 0.0;
    for (int k = 0; k < h->d; ++k) s += host[2 + k];
    g[1] = 0.5 * s;
  }
  g[1 + h->nlen] = 0.5 * host[1];
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
  hipError_t e = hipStreamCreateWithFlags(&h->stream, hipStreamNonBlocking);
  if (e != hipSuccess) {
    delete h;
    return fail(nullptr, GPRX_EHIP, std::string("hipStreamCreate: ") + hipGetErrorString(e));
  }
  h->own_stream = true;
  for (auto& ev : h->ev) hipEventCreate(&ev);
  int rc;
  if ((rc = ensure(h, h->invls, sizeof(double) * d)) || (rc = ensure(h, h->red, sizeof(double) * 16))) {
    gprx_destroy(h);
    return rc;
  }
  e = hipMalloc((void**)&h->info, sizeof(int));
  if (e != hipSuccess) {
    gprx_destroy(h);
    return fail(nullptr, GPRX_ENOMEM, "hipMalloc(info)");
  }
  *out = h;
  return GPRX_OK;
}

int gprx_destroy(gprx_handle h) {
  if (!h) return GPRX_OK;
  hipSetDevice(h->device);
  if (h->stream) hipStreamSynchronize(h->stream);
  for (Buf* b : {&h->X, &h->Y, &h->Z, &h->invls, &h->alpha, &h->red, &h->Kmat, &h->invD, &h->Xinv, &h->Tmp, &h->partial, &h->xs, &h->Ks,
                 &h->pred})
    if (b->p) hipFree(b->p);
  if (h->info) hipFree(h->info);
  for (auto& ev : h->ev)
    if (ev) hipEventDestroy(ev);
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

int gprx_synchronize(gprx_handle h) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  HIPCHK(h, hipStreamSynchronize(h->stream));
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
  HIPCHK(h, hipMemcpy(h->X.p, x, sizeof(double) * h->n * h->d, hipMemcpyHostToDevice));
  HIPCHK(h, hipMemcpy(h->Y.p, yt.data(), sizeof(double) * yt.size(), hipMemcpyHostToDevice));
  h->n_units = n_units;
  h->factorized = false;
  return GPRX_OK;
}

static int objective_impl(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss, double* grad) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (!theta) return fail(h, GPRX_EINVAL, "theta is null");
  if (unit < 0 || unit >= h->n_units) return fail(h, GPRX_EINVAL, "unit out of range (call gprx_set_data first)");
  for (int k = 0; k < h->ntheta; ++k)
    if (!std::isfinite(theta[k])) return fail(h, GPRX_EINVAL, "theta is not finite");
  const Theta t = decode_theta(h, theta);
  if (h->m != 0) return fail(h, GPRX_EINVAL, "sparse (m > 0) path not built in this library version");
  (void)z;
  double lml = 0.0;
  if ((rc = exact_factorize(h, unit, t, &lml))) return rc;
  const double lp = log_prior(h, t, mask);
  if (loss) *loss = -(lml + lp);
  if (grad) {
    std::vector<double> g(h->ntheta, 0.0);
    if ((rc = exact_gradient(h, t, g.data()))) return rc;
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
    // priors and softplus chain rule; loss = -(LML + log prior)
    grad[0] = (mask & GPRX_TRAIN_VARIANCE) ? -(g[0] + ln_dlogpdf(t.variance)) * sigmoid(t.w_var) : 0.0;
    for (int k = 0; k < h->nlen; ++k)
      grad[1 + k] = (mask & GPRX_TRAIN_LENGTHSCALE) ? -(g[1 + k] + ln_dlogpdf(t.ls[k])) * sigmoid(t.w_len[k]) : 0.0;
    grad[1 + h->nlen] = (mask & GPRX_TRAIN_NOISE) ? -(g[1 + h->nlen] + ln_dlogpdf(t.noise)) * sigmoid(t.w_noise) : 0.0;
  } else {
    HIPCHK(h, hipEventRecord(h->ev[4], h->stream));
  }
  HIPCHK(h, hipStreamSynchronize(h->stream));
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

int gprx_last_timings(gprx_handle h, double* ms4) {
  if (!h || !ms4) return fail(h, GPRX_EINVAL, "null argument");
  for (int s = 0; s < 4; ++s) ms4[s] = h->timings[s];
  return GPRX_OK;
}

int gprx_objective_batch(gprx_handle h, int count, const int* units, const double* theta, const double* z, int mask, double* losses,
                         double* grads) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (count < 0 || !units || !theta || !losses) return fail(h, GPRX_EINVAL, "null argument");
  const int64_t gw = h->ntheta + h->m * h->d;
  for (int i = 0; i < count; ++i) {
    rc = objective_impl(h, units[i], theta + (int64_t)i * h->ntheta, z ? z + (int64_t)i * h->m * h->d : nullptr, mask, losses + i,
                        grads ? grads + (int64_t)i * gw : nullptr);
    if (rc) return rc;
  }
  return GPRX_OK;
}

static constexpr int PRED_TILE = 8192;

int gprx_predict_dev(gprx_handle h, const double* xs_dev, int64_t ns, double* mean_dev, double* var_dev, int include_noise) {
  int rc;
  if ((rc = check_handle(h))) return rc;
  if (!h->factorized) return fail(h, GPRX_ESTATE, "gprx_predict before a successful gprx_factorize / gprx_objective");
  if (ns < 0 || (ns > 0 && (!xs_dev || !mean_dev || !var_dev))) return fail(h, GPRX_EINVAL, "null argument");
  if (h->m != 0) return fail(h, GPRX_EINVAL, "sparse (m > 0) path not built in this library version");
  const int np = (int)h->np;
  const int64_t ld = h->np;
  const int tile = (int)std::min<int64_t>(PRED_TILE, round_up(ns, NB));
  if ((rc = ensure(h, h->Ks, sizeof(double) * h->np * tile))) return rc;
  const int rows_per_chunk = 256;
  const int nchunks = (np + rows_per_chunk - 1) / rows_per_chunk;
  if ((rc = ensure(h, h->pred, sizeof(double) * (size_t)nchunks * tile))) return rc;
  hipStream_t st = h->stream;
  const double base = h->variance + (include_noise ? h->noise : 0.0);
  for (int64_t t0 = 0; t0 < ns; t0 += tile) {
    const int ts = (int)std::min<int64_t>(tile, ns - t0);
    const int tsp = (int)round_up(ts, NB);
    KmatArgs ka{h->X.p, xs_dev + t0 * h->d, h->invls.p, h->Ks.p, tile, (int)h->n, ts, h->d, np, tsp, h->variance, 0.0, 0, 0.0, 0};
    HIPCHK(h, launch_kmat(st, h->kid, ka));
    dim3 grid((ts + 255) / 256, nchunks);
    hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, h->alpha.p, np, ts, rows_per_chunk, h->pred.p);
    hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, 0.0, 1.0, mean_dev + t0);
    HIPCHK(h, trsm_lower_left(st, h->Kmat.p, ld, h->invD.p, h->Ks.p, tile, np, tsp));
    hipLaunchKernelGGL(colreduce_partial, grid, dim3(256), 0, st, h->Ks.p, (int64_t)tile, (const double*)nullptr, np, ts, rows_per_chunk,
                       h->pred.p);
    hipLaunchKernelGGL(colreduce_final, dim3((ts + 255) / 256), dim3(256), 0, st, h->pred.p, nchunks, ts, base, -1.0, var_dev + t0);
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
  HIPCHK(h, hipStreamSynchronize(h->stream));
  return GPRX_OK;
}

// ---- device memory helpers ---------------------------------------------------------------------
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
  HIPCHK(nullptr, hipMemcpy(dst_dev, src_host, (size_t)bytes, hipMemcpyHostToDevice));
  return GPRX_OK;
}
int gprx_memcpy_d2h(int device, void* dst_host, const void* src_dev, int64_t bytes) {
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, hipMemcpy(dst_host, src_dev, (size_t)bytes, hipMemcpyDeviceToHost));
  return GPRX_OK;
}

// ---- building blocks ----------------------------------------------------------------------------
int gprx_kmat(int device, int kernel_id, const double* a_dev, int64_t n1, const double* b_dev, int64_t n2, int d,
              const double* inv_ls_host, double variance, double diag_add, double* out_dev, int64_t ld, int64_t n1p, int64_t n2p,
              int mode) {
  if (!a_dev || !b_dev || !inv_ls_host || !out_dev) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (n1p % NB || n2p % NB || ld % 2 || n1p < n1 || n2p < n2 || ld < n2p) return fail(nullptr, GPRX_EINVAL, "padded sizes must be multiples of 64");
  if (kernel_id < 0 || kernel_id > 4 || mode < 0 || mode > 2) return fail(nullptr, GPRX_EINVAL, "bad kernel id or mode");
  HIPCHK(nullptr, hipSetDevice(device));
  double* dinv = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dinv, sizeof(double) * d));
  HIPCHK(nullptr, hipMemcpy(dinv, inv_ls_host, sizeof(double) * d, hipMemcpyHostToDevice));
  KmatArgs ka{a_dev, b_dev, dinv, out_dev, ld, (int)n1, (int)n2, d, (int)n1p, (int)n2p, variance, diag_add, mode, mode ? 1.0 : 0.0, 0};
  hipError_t e = launch_kmat(nullptr, kernel_id, ka);
  hipError_t e2 = hipDeviceSynchronize();
  hipFree(dinv);
  HIPCHK(nullptr, e);
  HIPCHK(nullptr, e2);
  return GPRX_OK;
}

int gprx_gemm(int device, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a_dev, int64_t lda,
              const double* b_dev, int64_t ldb, double beta, double* c_dev, int64_t ldc, int flags, int tile) {
  if (!a_dev || !b_dev || !c_dev) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (k % 16 || lda % 2 || ldb % 2) return fail(nullptr, GPRX_EINVAL, "k must be a multiple of 16, leading dimensions even");
  if (!((ta == 0 && tb == 1) || (ta == 0 && tb == 0) || (ta == 1 && tb == 0))) return fail(nullptr, GPRX_EINVAL, "unsupported transpose pair");
  if (tile != 0 && tile != 64 && tile != 128) return fail(nullptr, GPRX_EINVAL, "tile must be 0, 64 or 128");
  HIPCHK(nullptr, hipSetDevice(device));
  HIPCHK(nullptr, launch_gemm(nullptr, ta, tb, (int)m, (int)n, (int)k, alpha, a_dev, lda, b_dev, ldb, beta, c_dev, ldc, flags, tile));
  HIPCHK(nullptr, hipDeviceSynchronize());
  return GPRX_OK;
}

int gprx_potrf(int device, double* a_dev, int64_t lda, int64_t np, int64_t extra, double* inv_diag_dev, int* info_host) {
  if (!a_dev || !inv_diag_dev || !info_host) return fail(nullptr, GPRX_EINVAL, "null argument");
  if (np % NB || np <= 0 || extra < 0 || lda < np || lda % 2) return fail(nullptr, GPRX_EINVAL, "np must be a positive multiple of 64");
  HIPCHK(nullptr, hipSetDevice(device));
  int* dinfo = nullptr;
  HIPCHK(nullptr, hipMalloc((void**)&dinfo, sizeof(int)));
  HIPCHK(nullptr, hipMemset(dinfo, 0, sizeof(int)));
  hipError_t e = potrf_lower(nullptr, a_dev, lda, (int)np, (int)extra, inv_diag_dev, dinfo);
  hipError_t e2 = hipDeviceSynchronize();
  hipMemcpy(info_host, dinfo, sizeof(int), hipMemcpyDeviceToHost);
  hipFree(dinfo);
  HIPCHK(nullptr, e);
  HIPCHK(nullptr, e2);
  return *info_host ? fail(nullptr, GPRX_ENOTPD, "matrix not positive definite") : GPRX_OK;
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
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, 100);
  hipEventRecord(e0, nullptr);
  hipLaunchKernelGGL(mfma_f64_peak_kernel, dim3(blocks), dim3(256), 0, nullptr, out, iters);
  hipEventRecord(e1, nullptr);
  hipError_t e = hipDeviceSynchronize();
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
