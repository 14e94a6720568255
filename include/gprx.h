/*
 * gprx.h -- C ABI of the MI355X-native GP regression engine (libgprx.so).
 *
 * Drop-in boundary for the hot path of fema-ffrd/gpras.  The reference has NO FFI or
 * plugin interface for this path: its boundary is the Python class GPRAS
 * (/root/reference/gpras/gpr.py:217-384) whose arithmetic is delegated to gpflow's SGPR.
 * Each entry point below names the reference lines whose work it replaces; the Python
 * shim that a gpras maintainer would bind (ctypes) is gpras_amd/_lib.py and
 * INTEGRATION.md shows the stub.
 *
 * Conventions
 *   - plain pointers and sizes only; all floating point is IEEE binary64 ("f64").
 *   - host matrices are C-contiguous row-major (what numpy hands over after
 *     x.astype(np.float64), gpr.py:265-266, :333).  The caller owns every host buffer;
 *     the library copies during the call and never keeps a host pointer.
 *   - every function returns an int status (GPRX_OK == 0).  No C++ exception crosses the
 *     boundary.  gprx_last_error() returns a message for the last failure on that handle
 *     (or, with a NULL handle, of the calling thread's last handle-less failure).
 *   - a handle is bound to one device and one HIP stream and is NOT thread-safe; different
 *     handles may be driven from different host threads / processes (one per GPU).
 *   - "unconstrained" parameters are the optimiser's variables: softplus^-1 of kernel
 *     variance and lengthscale(s), softplus^-1(noise - 1e-6) for the likelihood variance
 *     (gpflow positive() / Gaussian likelihood lower bound, as used at gpr.py:298-305).
 *     theta = [w_variance, w_lengthscale[0..n_len-1], w_noise],  n_len = ard ? d : 1.
 *   - device-pointer variants (suffix _dev) take pointers obtained from gprx_dev_malloc or
 *     from any allocator of the same HIP runtime (e.g. torch.Tensor.data_ptr()).
 */
#ifndef GPRX_H
#define GPRX_H

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GPRX_VERSION 100 /* 0.1.0 */

/* status codes */
#define GPRX_OK 0
#define GPRX_EINVAL 1  /* bad argument                       -> ValueError / KeyError   */
#define GPRX_ENOTPD 2  /* Cholesky hit a non-positive pivot  -> numpy.linalg.LinAlgError */
#define GPRX_EHIP 3    /* HIP runtime error                  -> RuntimeError            */
#define GPRX_ENOMEM 4  /* device allocation failed           -> MemoryError             */
#define GPRX_ESTATE 5  /* call order violated (e.g. predict before factorize)           */
#define GPRX_ERCCL 6   /* RCCL missing or a collective failed                -> RuntimeError */

/* kernel ids: the five stationary kernels KERNEL_FACTORY (gpr.py:21-37) can construct
 * with kernel(variance=, lengthscales=) at gpr.py:298 */
#define GPRX_KERNEL_RBF 0
#define GPRX_KERNEL_MATERN12 1
#define GPRX_KERNEL_MATERN32 2
#define GPRX_KERNEL_MATERN52 3
#define GPRX_KERNEL_EXPONENTIAL 4

/* trainable mask bits (gpflow.set_trainable at gpr.py:48-49, 115-125, 133-143) */
#define GPRX_TRAIN_VARIANCE 1
#define GPRX_TRAIN_LENGTHSCALE 2
#define GPRX_TRAIN_NOISE 4
#define GPRX_TRAIN_Z 8

typedef struct gprx_ctx* gprx_handle;
typedef struct gprx_pca_ctx* gprx_pca_handle;
typedef struct gprx_comm_ctx* gprx_comm;

/* ---- library / device -------------------------------------------------------------- */
int gprx_version(void);
const char* gprx_last_error(gprx_handle h);
int gprx_device_count(int* count);

/* ---- model handle ------------------------------------------------------------------ */
/* One handle = one training set x (n, d) shared by n_units output columns, i.e. the list
 * self.models that GPRAS._init_models builds (gpr.py:277-308).  m = number of inducing
 * points (SGPR, gpr.py:299); m == 0 selects the exact GP (Z = X specialisation). */
int gprx_create(int device, int64_t n, int d, int64_t m, int kernel_id, int ard, gprx_handle* out);
int gprx_destroy(gprx_handle h);
/* run all work of this handle on an existing HIP stream (e.g. torch's current stream) */
int gprx_set_stream(gprx_handle h, void* hip_stream);
int gprx_synchronize(gprx_handle h);

/* Form of the scaled squared distance r2 inside every kernel evaluation of this handle (kernel-matrix builds and the
 * g / h factors of the gradient passes):
 *   GPRX_DIST_DIFFERENCE (default)  r2 = sum_k ((a_k - b_k) / l_k)^2: no cancellation, r2(a, a) == 0 exactly;
 *   GPRX_DIST_EXPANDED              r2 = |a/l|^2 + |b/l|^2 - 2 (a/l).(b/l): the literal arithmetic of gpflow's
 *                                   square_distance, which the kernels constructed at gpr.py:298 evaluate.
 * The two agree to rounding for RBF / Matern32 / Matern52; for Matern12 / Exponential (not differentiable at r = 0) the
 * expanded form leaves r2 ~ 1e-15 on coincident points, which moves outputs by 1e-9 .. 2.5e-8:
 * a caller that promises 1e-8 against gpflow selects GPRX_DIST_EXPANDED for those two kernels -- the host class
 * gpras_amd.gpr.GPRAS does so by default. */
#define GPRX_DIST_DIFFERENCE 0
#define GPRX_DIST_EXPANDED 1
int gprx_set_distance_form(gprx_handle h, int form);

/* x: (n, d) row-major, y: (n, n_units) row-major -- the arrays GPRAS.fit stores after the
 * float64 cast (gpr.py:265-266). */
int gprx_set_data(gprx_handle h, const double* x, const double* y, int n_units);

/* SGPR.training_loss() and its gradient w.r.t. the unconstrained variables (replaces the
 * GradientTape / gpflow.optimizers.Scipy evaluations at gpr.py:61-62, 95, 127, 153-155,
 * 186-188, 197-203).  LogNormal(0,1) log-priors (gpr.py:303-305) are added for the
 * parameters whose mask bit is set, as gpflow does for trainable parameters.
 *   theta : n_theta = 2 + n_len unconstrained values
 *   z     : (m, d) inducing inputs, ignored (may be NULL) when m == 0
 *   loss  : out, scalar
 *   grad  : out or NULL; n_theta values followed by m*d values for Z; entries of
 *           parameters whose mask bit is clear are written as 0.
 * Leaves the factorisation resident, so gprx_predict may follow for the same (unit, theta, z). */
int gprx_objective(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss, double* grad);

/* Factorise only (kernel build + Cholesky + weights); what SGPR.predict_y recomputes on
 * every call at gpr.py:337.  loss may be NULL; otherwise receives the training loss with
 * priors for the parameters in mask. */
int gprx_factorize(gprx_handle h, int unit, const double* theta, const double* z, int mask, double* loss);

/* Many independent cells on one GPU: factorise `count` exact models (one handle each, all on the caller's
 * thread) by enqueueing every handle's work before waiting for any of them, so the latency-bound panel
 * chains of different cells overlap on the device.  thetas: (count, n_theta); losses: count values.
 * Replaces a Python loop of gprx_factorize calls; each handle is left factorised as by gprx_factorize.  A cell whose matrix
 * is not positive definite gets a NaN loss, the others finish, and the call returns the first such error (GPRX_ENOTPD). */
int gprx_factorize_many(int count, gprx_handle* handles, const int* units, const double* thetas, int mask, double* losses);

/* Batched cells on ONE handle: `count` exact factorisations -- cell i = (units[i], thetas[i]), all on the handle's x,
 * which is what the reference's per-mode loop (gpr.py:272-274, 336-339), its multi-start (_optimize_multi_start, gpr.py:73-109) and its
 * differential-evolution objective (_optimize_differential_evolutions, gpr.py:44-70) evaluate one after the other.  Every kernel of the factorisation
 * is launched once for all cells (cell index in the grid), so small matrices still fill the GPU.  thetas: (count,
 * n_theta); losses (may be NULL): count training losses, NaN for a cell whose matrix is not positive definite; status
 * (may be NULL): per-cell GPRX_OK / GPRX_ENOTPD.  Returns GPRX_ENOTPD if any cell failed (the others are valid).
 * Results are bit-identical to gprx_factorize on each cell -- except where many SMALL matrices take the one-workgroup-per-cell
 * factorisation (default: N <= 256 from 32 cells, N <= 512 from 160, N <= 1024 from 256; tuning key "cell_kernel" = -1 forbids it,
 * 1 forces it): the same tile products with a tile's whole update formed as one sum, equal to the launch sequence to rounding
 * (1e-13 relative on the loss, tests/test_gpu_cells.py); and from 24 cells per launch on (the split panel) the right-hand side y of a
 * cell travels as a VECTOR through the launch sequence instead of a 64-row tile below the matrix (4.5 % fewer flops at N = 4096): the
 * factor and log det are gprx_factorize's bits, beta = L^-1 y and with it y^T K^-1 y are summed in another fixed order (1e-14 relative
 * on the loss, 1e-12 on gradients; tuning key "rhs_vector" = -1 keeps the tile and with it the single call's bits).
 * The factorisations stay resident in slots 0..count-1
 * until the next batch; gprx_select_slot makes one of them current for gprx_predict / gprx_predict_dev. */
int gprx_factorize_batch(gprx_handle h, int count, const int* units, const double* thetas, int mask, double* losses, int* status);
int gprx_select_slot(gprx_handle h, int slot);
/* device time (ms) of the last gprx_factorize_batch, HIP events on the handle's stream around the whole batch */
int gprx_last_batch_ms(gprx_handle h, double* ms);

/* SGPR.predict_y (gpr.py:336-339): predictive mean and variance at xs (ns, d) for the unit
 * factorised last.  include_noise != 0 adds the likelihood variance (predict_y); 0 gives
 * predict_f.  mean/var: ns values each. */
int gprx_predict(gprx_handle h, const double* xs, int64_t ns, double* mean, double* var, int include_noise);
/* same, every pointer is a device pointer; asynchronous on the handle's stream */
int gprx_predict_dev(gprx_handle h, const double* xs_dev, int64_t ns, double* mean_dev, double* var_dev, int include_noise);

/* timings (ms, HIP events on the handle's stream) of the stages of the last
 * gprx_objective / gprx_factorize call: [kernel build, cholesky, solves, gradient]. */
int gprx_last_timings(gprx_handle h, double* ms4);

/* Per-launch timing of the Cholesky's two kernels (exact path), for bench.py's roofline line.
 * With profiling enabled every gprx_factorize / gprx_objective / gprx_factorize_batch brackets each launch with HIP
 * events on the launch stream (this perturbs the run slightly: keep it off inside timed regions).
 * gprx_last_profile: out8 = [main GEMM kernel gemm_f64_kernel<0,1,64,64,0> (bulk trailing updates and in-block updates
 *                           with K > 128) total ms, launches, algorithmic flops (all cells of a batch), panel kernel
 *                           total ms, launches, short-K in-block updates (K = 64 / 128: GEMM with C
 *                           prefetch) total ms, launches, algorithmic flops] of the last exact factorisation(s). */
int gprx_set_profiling(gprx_handle h, int enabled);
int gprx_last_profile(gprx_handle h, double* out8);
/* The kernel-build launch (kmat_kernel: every 64 x 64 tile on or below the diagonal, all cells of a batch in one launch) of the
 * last PROFILED exact factorisation: duration by HIP events around that launch, and the bytes it writes (8 * 64 * 64 * lower tiles
 * * cells) -- bench.py's kernel_build_hbm figure (north_star: "HBM GB/s on the kernel build"; reference: the K(X, X) inside
 * every training_loss, /root/reference/gpras/gpr.py:153-155). */
int gprx_last_kernel_build(gprx_handle h, double* ms, double* bytes);
/* The one-workgroup-per-cell Cholesky launch (potrf_cell.h: the default for many cells of N <= 1024) of the last PROFILED
 * gprx_factorize_batch on a handle whose tuning forces it ("cell_kernel" = 1; without that a profiled batch runs the instrumented
 * launch sequence): duration by HIP events around that ONE launch, its algorithmic flops (N^3 / 3 per cell) and the cell count;
 * zeros when the last profiled call did not run it.  bench.py's other_sizes.N1024_d8_roofline. */
int gprx_last_cell_kernel(gprx_handle h, double* ms, double* flops, double* cells);

/* ---- batched small problems ------------------------------------------------------- */
/* Evaluate loss (+ gradient) for `count` units in one call: units[i] with theta row i
 * (count, n_theta) and z block i (count, m, d).  Replaces the serial loop over
 * self.models at gpr.py:272-274.  losses: count values; grads: (count, n_theta + m*d) or NULL.
 * With count > 1 (and d <= 64) every stage runs once for all cells, the cell index in every launch's grid: exact models
 * -- kernel build, Cholesky, solves, L^-1, K^-1, the trace pass (as gprx_factorize_batch) --, and sparse models.
 * Sparse models with M <= 64 inducing points (the reference's example configuration has 50) take FIVE launches per evaluation
 * whatever the cell count (round 5, csrc/sgpr_fused.h: Kuu and its factor | Kuf tile by tile on MFMA against the register-resident
 * L^-1, never stored | B, its factor and the M x M algebra of the gradient | the contractions with dk/dtheta and dk/dZ | sums in a
 * fixed order); count == 1 takes them too, and so does gprx_objective: a model evaluated alone and inside a batch gives the
 * same bits.  Larger M takes the general launch sequence (Kuf, Kuu, both Cholesky factorisations, A, B (split-K), c, the M x M
 * products of the gradient, both trace passes, dZ: ~45 launches serve all cells), bit-identical to gprx_objective on each cell.
 * The tuning key "sgpr_fused" = 0 sends M <= 64 through that sequence as well (equal to rounding, not bit for bit).  A cell
 * whose matrix is not positive definite gets NaN loss and gradient and the call returns GPRX_ENOTPD after finishing the
 * others.  The single-model state of the handle is left unfactorised (gprx_predict needs a gprx_factorize / gprx_objective). */
int gprx_objective_batch(gprx_handle h, int count, const int* units, const double* theta, const double* z, int mask,
                         double* losses, double* grads);

/* Sizing of batched calls: free / total device memory, and the device bytes ONE cell of a batched call on this handle
 * occupies (exact models: kernel matrix + staging + alpha, plus L^-1 and K^-1 when gradients are asked for; sparse models:
 * the cell block of gprx_objective_batch).  A caller keeps count * cell_bytes below the free memory; a batch that does
 * not fit returns GPRX_ENOMEM and may be retried with fewer cells (results do not depend on the batch composition). */
int gprx_mem_info(int device, int64_t* free_bytes, int64_t* total_bytes);
int gprx_cell_bytes(gprx_handle h, int with_gradient, int64_t* bytes);

/* ---- multi-GPU: independent units sharded over ranks, ONE gather at the end ------------------- */
/* The reference has no multi-device code; its unit loops (gpr.py:272-274, 336-339; restarts :87; CV configurations,
 * cross_validation.py:61) are serial and share nothing but x.  One process per GPU: unit u -> rank u mod world, no
 * communication during compute, and these calls for the single collective at the end.  RCCL is loaded at run time
 * (dlopen), communicators are bound to one device and one private stream; every buffer is DEVICE memory and the calls are
 * asynchronous on that stream (gprx_comm_synchronize waits), so results never bounce through the host.
 *   gprx_comm_unique_id : rank 0 creates the 128-byte id (ncclGetUniqueId); the launcher distributes it (any side channel:
 *                         torch.distributed's store, a file, MPI) -- the id is the only out-of-band datum.
 *   gprx_comm_init      : collective over all ranks (ncclCommInitRank).
 *   gprx_comm_all_gather: recv_dev (world * count) <- every rank's send_dev (count), rank-major (ncclAllGather).
 *   gprx_comm_gather    : to `root` only, as grouped ncclSend / ncclRecv (all inbound xGMI links of the root at once);
 *                         recv_dev is read on the root only.
 *   gprx_comm_all_reduce_max: element-wise maximum in place (the slowest rank's time of a benchmark).
 *   gprx_comm_all_gather_host: the same all-gather for small HOST buffers (fitted parameters), staged through the device;
 *                         synchronous.   gprx_comm_barrier: all ranks have arrived (one-element all-reduce + wait). */
#define GPRX_UNIQUE_ID_BYTES 128
/* Everything gprx_comm_init needs short of the collective itself: the device is selectable, HIP is initialised, RCCL is loaded
 * and its symbols resolve.  Ranks agree on this through their launcher BEFORE anyone calls gprx_comm_init (a rank that fails
 * here would otherwise leave the others blocked inside ncclCommInitRank). */
int gprx_comm_runtime_check(int device);
int gprx_comm_unique_id(unsigned char* id128);
int gprx_comm_init(int device, int rank, int world, const unsigned char* id128, gprx_comm* out);
int gprx_comm_destroy(gprx_comm c);
const char* gprx_comm_last_error(gprx_comm c);
/* rank / world as RCCL itself reports them for the communicator (ncclCommUserRank / ncclCommCount), not the caller's numbers */
int gprx_comm_rank(gprx_comm c, int* rank, int* world);
int gprx_comm_all_gather(gprx_comm c, const double* send_dev, double* recv_dev, int64_t count);
int gprx_comm_gather(gprx_comm c, const double* send_dev, double* recv_dev, int64_t count, int root);
int gprx_comm_all_reduce_max(gprx_comm c, double* buf_dev, int64_t count);
int gprx_comm_all_gather_host(gprx_comm c, const double* send, double* recv, int64_t count);
int gprx_comm_barrier(gprx_comm c);
int gprx_comm_synchronize(gprx_comm c);

/* ---- device memory helpers (for callers that keep inputs resident in HBM) ----------- */
int gprx_dev_malloc(int device, int64_t bytes, void** out);
int gprx_dev_free(int device, void* ptr);
int gprx_memcpy_h2d(int device, void* dst_dev, const void* src_host, int64_t bytes);
int gprx_memcpy_d2h(int device, void* dst_host, const void* src_dev, int64_t bytes);

/* ---- building blocks (exported for parity tests, profiling and bench.py) ----------- */
/* All matrices row-major f64 in device memory, leading dimension in elements.  These run on
 * the library's non-blocking utility stream of `device` (one per device; the library never touches the legacy NULL stream) and
 * synchronise that stream before returning unless noted. */

/* out[i, j] = variance * g(r(a_i, b_j)) + (i == j ? diag_add : 0)
 * a: (n1, d), b: (n2, d) device, ls: d host values (lengthscale per dimension; inputs are divided by it).
 * mode 0: all of the (n1p, n2p) padded rectangle, zero padding; mode 1: a == b, only tiles on or
 * below the diagonal are written (what the Cholesky reads); mode 2: a == b, all tiles.  Modes 1
 * and 2 pad (i >= n1 or j >= n2) with the identity.  mode + 4: the same with r2 in gpflow's expanded form
 * (GPRX_DIST_EXPANDED, see gprx_set_distance_form). */
int gprx_kmat(int device, int kernel_id, const double* a_dev, int64_t n1, const double* b_dev, int64_t n2, int d,
              const double* ls_host, double variance, double diag_add, double* out_dev, int64_t ld, int64_t n1p,
              int64_t n2p, int mode);

/* C = alpha * op(A) op(B) + beta * C;  ta/tb: 0 = as stored, 1 = transposed.  Supported:
 * (ta,tb) in {(0,1), (0,0), (1,0)}.  k must be a multiple of 16.  flags: GPRX_GEMM_* */
#define GPRX_GEMM_C_LOWER 1  /* compute only tiles touching the lower triangle of C            */
#define GPRX_GEMM_A_LOWER 2  /* op(A)[i,k] == 0 for k > i  (skip zero tiles)                  */
#define GPRX_GEMM_A_UPPER 4  /* op(A)[i,k] == 0 for k < i                                     */
#define GPRX_GEMM_B_LOWER 8  /* op(B)[k,j] == 0 for k < j                                     */
#define GPRX_GEMM_B_UPPER 16 /* op(B)[k,j] == 0 for k > j                                     */
int gprx_gemm(int device, int ta, int tb, int64_t m, int64_t n, int64_t k, double alpha, const double* a_dev, int64_t lda,
              const double* b_dev, int64_t ldb, double beta, double* c_dev, int64_t ldc, int flags, int tile);

/* In-place lower Cholesky of the (np, np) matrix (np multiple of 64) with `extra` right-hand
 * side rows stored below it (rows np .. np+extra-1, each of length np): on return the lower
 * triangle holds L, the extra rows hold (L^-1 rhs)^T, inv_diag (np/64 blocks of 64x64) holds
 * the inverses of the diagonal blocks.  info_host: 0, or 1-based index of the failing pivot. */
int gprx_potrf(int device, double* a_dev, int64_t lda, int64_t np, int64_t extra, double* inv_diag_dev, int* info_host);

/* The whole predict loop of gpr.py:336-339 in one call: cell i = (units[i], thetas[i], z block i for sparse models) is
 * factorised (exact models: all cells by one batched launch sequence) and predicts at the shared xs (ns, d);
 * means / vars: (count, ns) row-major (the transpose of GPRAS.predict's (ns, K)). */
int gprx_predict_batch(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs, int64_t ns,
                       double* means, double* vars, int include_noise);

/* The same with the results in the layout GPRAS.predict returns (gpr.py:340-342: the modes' columns concatenated): means_t / vars_t
 * (ns, count) row-major -- every slab of cells is transposed on the device before it is copied out.  Same values. */
int gprx_predict_batch_t(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs, int64_t ns,
                         double* means_t, double* vars_t, int include_noise);

/* The same with the test points and the results in DEVICE memory (means_dev / vars_dev: (count, ns) row-major), asynchronous
 * on the handle's stream once the batched factorisation has returned: predictions that stay in HBM for the reverse
 * projection and the metrics (production/analysis/pipeline.py:260-288). */
int gprx_predict_batch_dev(gprx_handle h, int count, const int* units, const double* thetas, const double* z, const double* xs_dev, int64_t ns,
                           double* means_dev, double* vars_dev, int include_noise);

/* The reference's Adam driver (gpr.py:147-173: tf.keras.optimizers.Adam() defaults, at most max_iter steps, early stop once the
 * relative improvement of the loss stayed <= 1e-5 for more than 50 consecutive steps) for `count` cells in lock step.
 * Sparse models with M <= 64 (round 5): the loop is RESIDENT ON THE DEVICE -- variables, moments, best loss and patience counters
 * live in device memory, a step is four launches (the last one forms loss and gradient, applies the update and the stop rule and
 * opens the next step with Kuu of the new variables; softplus, its derivative and the LogNormal priors are evaluated inside the
 * kernel), cells that have stopped return at once from every launch, and the host only reads the stop flags every 25 steps
 * (GPRX_ADAM_CHECK_EVERY): nothing else crosses the host link between the call's first upload and its last download.
 * GPRX_ADAM_HOST=1 selects the host-stepped loop below instead; both give the same variables bit for bit (the scalar tail of an
 * evaluation and the update are ONE source for host and device, csrc/sgpr_asm.h, on exp / log written out in IEEE operations,
 * csrc/px_math.h).  Other models: every step
 * is ONE batched evaluation (gprx_objective_batch) of the cells still running, the update happens here on the host side of the
 * library -- no per-step round trip through the caller's language.  theta (count, n_theta) and z (count, m, d; NULL for exact
 * models) are the optimiser's variables, updated in place (elements outside `mask` stay as they are); n_evals[i] receives the
 * number of evaluations cell i took part in; batches (optional) the number of batched evaluations.  The arithmetic per element
 * is that of the NumPy statement of the update (one rounding per operation, no contraction): the result equals
 * gpras_amd.optimizers._optimize_adam on each cell bit for bit.  A cell whose matrix stops being positive definite ends the
 * call with GPRX_ENOTPD (gpr.py: the exception leaves the optimiser); theta / z hold the state of that step (resident loop: the
 * failing cell is as it was before the failing evaluation, the others may be up to 24 steps further: the flags are read every 25). */
int gprx_adam_batch(gprx_handle h, int count, const int* units, double* theta, double* z, int mask, int max_iter, int* n_evals, int* batches);

/* ---- EOF (PCA) projection either side of the GP path: SURVEY.md section 8(f) row N1 ------------------- */
/* One projector = the fitted state of a reference PreProcessor (gpras/preprocess.py:868-927): `dry` (n_cells bytes, 1 =
 * always-dry cell, may be NULL = none), `elevations` (n_cells, needed for depth mode and for filling dry cells in wse mode),
 * and, over the n_wet = n_cells - sum(dry) wet cells in ascending cell order: `input_mean` (n_wet), `weights` (n_wet or
 * NULL = unweighted), `eofs` (k, n_wet) row-major; `x_mean`, `x_std` (k).  depth_mode != 0: inputs are water-surface
 * elevations converted with max(x - elevation, 0) (wse_2_depth, preprocess.py:1040-1044).  1 <= k <= 64. */
int gprx_pca_create(int device, int64_t n_cells, int k, const unsigned char* dry, const double* elevations, const double* input_mean,
                    const double* weights, const double* eofs, const double* x_mean, const double* x_std, int depth_mode,
                    gprx_pca_handle* out);
int gprx_pca_destroy(gprx_pca_handle p);
const char* gprx_pca_last_error(gprx_pca_handle p);
/* PreProcessor.transform (preprocess.py:1009-1038): x (rows, n_cells) -> z (rows, k), host buffers. */
int gprx_pca_transform(gprx_pca_handle p, const double* x, int64_t rows, double* z);
/* PreProcessor.reverse_transform (preprocess.py:1052-1085) with _linear_transform_for_var (:1087-1094): mean (rows, k)
 * [, var (rows, k)] -> full (rows, n_cells) [, var_full (rows, n_cells)]; var and var_full both NULL or both given. */
int gprx_pca_reverse(gprx_pca_handle p, const double* mean, const double* var, int64_t rows, double* full, double* var_full);
/* device-resident forms, asynchronous on the projector's stream (gprx_pca_synchronize waits).  x_dev: (rows, ld) with
 * ld = n_cells rounded up to a multiple of 16 (padding columns: any finite values); outputs as above with ld = k / n_cells. */
int gprx_pca_transform_dev(gprx_pca_handle p, const double* x_dev, int64_t rows, double* z_dev);
int gprx_pca_reverse_dev(gprx_pca_handle p, const double* mean_dev, const double* var_dev, int64_t rows, double* full_dev, double* vfull_dev);
int gprx_pca_synchronize(gprx_pca_handle p);
/* What production/analysis/pipeline.py:262-277 and :286 do to the reconstructed fields before the metrics, in place on the
 * device and on the projector's stream:  to_depth -- field (rows, n_cells): add_elevations_first != 0 ("depth" models:
 * y += elevations, then wse_2_depth) computes max((y + e) - e, 0), else max(y - e, 0) (PreProcessor.wse_2_depth,
 * preprocess.py:1040-1044; also for the truth field);  sqrt -- conf = sqrt(var) over `count` values;  transpose -- dst (cols,
 * rows) = src (rows, cols)^T, e.g. the (modes, points) block of gprx_predict_batch_dev into reverse's (points, modes). */
int gprx_pca_to_depth_dev(gprx_pca_handle p, double* field_dev, int64_t rows, int add_elevations_first);
int gprx_pca_sqrt_dev(gprx_pca_handle p, double* field_dev, int64_t count);
int gprx_pca_transpose_dev(gprx_pca_handle p, const double* src_dev, int64_t rows, int64_t cols, double* dst_dev);

/* ---- fused error metrics over two fields: SURVEY.md section 8(f) row N3 (gpras/metrics.py:85-318) ---------- */
/* Two streaming passes over x (truth), y (prediction) and conf (may be NULL), each (rows, cells) row-major, yield every
 * reduction the reference's metric functions need:
 *   row_sums  (rows, 4): per timestep, over cells: sum (x-y), sum (x-y)^2, sum conf, sum |x-y|
 *   cell_sums (5, cells): per cell, over timesteps: sum (x-y), sum (x-y)^2, sum conf, max_t x, max_t y
 *   cell_arg  (2, cells): first timestep of the maximum of x and of y (numpy argmax semantics, metrics.py:35-36)
 *   matches: number of (t, cell) pairs counted by the fidelity index with lag tolerance t_tol (0..8) and value tolerance
 *            v_tol (metrics.py:187-197). */
int gprx_metrics(int device, const double* x, const double* y, const double* conf, int64_t rows, int64_t cells, int t_tol, double v_tol,
                 double* row_sums, double* cell_sums, int* cell_arg, unsigned long long* matches);
/* same with device-resident fields and outputs (matches is a host pointer; the call synchronises) */
int gprx_metrics_dev(int device, const double* x_dev, const double* y_dev, const double* conf_dev, int64_t rows, int64_t cells, int t_tol,
                     double v_tol, double* row_sums_dev, double* cell_sums_dev, int* cell_arg_dev, unsigned long long* matches);

/* ---- k-means inducing-point initialisation: SURVEY.md section 8(f) row N4 (gpras/gpr.py:312-315) ---------------------- */
/* The Lloyd iterations of KMeans(n_clusters=M, random_state=0, n_init="auto").fit(x) (scikit-learn's
 * _kmeans_single_lloyd) on the device.  x: (n, d) host, already centred (KMeans subtracts the column means); centers: (m, d)
 * host, in: the k-means++ seeding (host: sklearn.cluster.kmeans_plusplus on RandomState(0), as KMeans draws it), out: the
 * final centres (still centred); tol: absolute (mean(var(x)) * 1e-4); labels: n values out; n_iter: iterations run.
 * Stops like scikit-learn: labels repeat (strict convergence) or sum of squared centre shifts <= tol, at most max_iter.
 * *empty = 1 when a cluster lost all members (scikit-learn relocates it; centres / labels are then undefined and the
 * caller falls back to scikit-learn).  d <= 64. */
/* k-means++ seeding of the same KMeans call on the device (sklearn.cluster._kmeans._kmeans_plusplus behind gpr.py:313).  The host
 * keeps the RandomState(0) draws, which do not depend on the data: first_id = random_state.choice(n), uniforms = (m - 1) x trials
 * values of random_state.uniform(size=trials) with trials = 2 + int(log(m)).  x: (n, d) host, already centred; xsq: its squared row
 * norms (sklearn.utils.extmath.row_norms).  indices_out: the m chosen rows of x.  Candidate distances, the running minimum, the
 * potentials and the cumulative-sum search run on the device. */
int gprx_kmeans_pp(int device, const double* x, int64_t n, int d, const double* xsq, int m, int trials, int64_t first_id, const double* uniforms,
                   int64_t* indices_out);
int gprx_kmeans_lloyd(int device, const double* x, int64_t n, int d, double* centers, int m, double tol, int max_iter, int32_t* labels,
                      int* n_iter, int* empty);

/* out[c] = field[idx[c], c] for a device-resident field (rows, cells): the gathers x[x_mts, np.arange(x.shape[1])] that
 * every *_mts function of the reference performs (gpras/metrics.py:119-121, 133-135, 147-151, 167-171, 215-224, ...) when
 * the CALLER supplies the timesteps (x_mts / y_mts), as export_metric_summary does at metrics.py:46-57.  idx: cells host
 * values; negative values count from the end as in numpy; an index outside [-rows, rows) gives GPRX_EINVAL (numpy raises
 * IndexError).  out: cells host values. */
int gprx_gather_rows(int device, const double* field_dev, int64_t rows, int64_t cells, const int64_t* idx, double* out);

/* Tuning of the Cholesky schedule; value 0 restores the default.  gprx_set_tuning changes the PROCESS DEFAULTS: they
 * are read by the handle-less building blocks (gprx_potrf) and COPIED into a handle when it is created, so a handle never
 * sees a later change (handles on different host threads do not share mutable tuning state); gprx_set_handle_tuning changes
 * one handle's copy.  Set process defaults before creating handles, from one thread.  Keys: "panel_width" (64 | 128),
 * "outer_block" (multiple of 128), "update_tile" (64 | 128: workgroup tile of the bulk trailing update), "no_lookahead"
 * (1: single stream), "panel_rows" (128 | 256 rows per panel workgroup), "panel_occ" (2 | 3 workgroups per CU),
 * "inblock" (1: right-looking K = 64 strips inside an outer block instead of recursive halving), "split_panel"
 * (1: always one diagonal workgroup + a rows-only kernel per panel, -1: never; default: from 24 cells per launch on).
 * "cell_kernel" (1: batched cells always take the one-workgroup-per-cell factorisation, -1: never; default by size, see
 * gprx_factorize_batch).
 * "rhs_vector" (-1: batched cells always carry their right-hand side as a 64-row tile below the matrix, as single calls do; default 0:
 * as a vector wherever the split panel runs -- potrf_rows_kernel<..., YVEC>; see gprx_factorize_batch).
 * "rows_inv" (1: the split panel solves the rows below a diagonal block by ONE MFMA tile product against the block's explicit inverse
 * (potrf_rows_inv_kernel) instead of the eight-step substitution: equal to rounding, not bit for bit; "rows_inv_rt" = 1 | 2 sixteen-row
 * tiles per wave, "rows_inv_lone" = 1: a lone matrix takes the split panel too.  Opt-in: measured no faster).
 * "split_updates" (1: ONE matrix's in-block and HEAD updates with K >= 256 are split by columns -- the 64 columns the next panel
 * needs on the main stream, the rest in dyadic pieces on a side stream behind events; bit-identical factor, measured slower).
 * "dag" (1: ONE matrix is factored by the tile-DAG kernel -- a single persistent launch, the dependent chain of diagonal
 * blocks in one workgroup, every other tile task claimed from a queue and ordered by per-tile version counters in device memory;
 * deterministic, within rounding of the default; default 0 = the launch-per-panel schedule, which measured faster on MI355X:
 * DESIGN.md section 3.2b).  Applies to EAGER single factorisations only (gprx_factorize / gprx_objective): the graph replays of
 * gprx_factorize_many and every batched call always run the launch-per-panel schedule.
 * "poison_workspace" (testing, 1: the L^-1 workspace of the gradient starts as NaN patterns instead of whatever it held -- the
 * gradient never depends on its old contents, and no longer zeroes it).
 * Every schedule gives the same factor up to rounding; fused and split panels are bit-identical.
 * "predict_path": 0 choose (default), 1 always the triangular GEMM against L^-1, 2 always blocked forward substitution.
 * "sgpr_fused": 1 (default) sparse models with M <= 64 take the five-launch evaluation and the device-resident Adam loop, 0: the
 * general launch sequence (gprx_objective_batch); a handle's cell blocks are rebuilt when its value changes.
 * "sgpr_groups_from" (process-wide): the device-resident Adam loop runs a batch of at least this many cells (stated at N = 4096, i.e.
 * 16 chunks of 256 training points per cell: the criterion is cells x chunks > 16 (value - 1), a pass that needs a second round of the
 * CUs) as two groups of cells on two streams, one launch apart, so that one group's one-workgroup-per-cell launches overlap the
 * other's streamed passes (default 17; 1: always; 0: never; every cell's values are the same bits either way).
 * "wait_handover_us" (process-wide): microseconds of polling after which a wait hands over to hipStreamSynchronize (default
 * 200 000; tests set 0 to force the hand-over). */
int gprx_set_tuning(const char* key, int value);
int gprx_set_handle_tuning(gprx_handle h, const char* key, int value);

/* The device exponentials on an array of HOST values (parity tests): which = 0: the kernel-matrix build's exp (2^(j/64) table in LDS +
 * degree-5 polynomial, arguments <= 0), which = 1: the degree-13 form the gradient passes use.  x, out: n host doubles. */
int gprx_exp_probe(int device, int which, const double* x, int64_t n, double* out);

/* measured back-to-back v_mfma_f64_16x16x4_f64 rate of the whole chip, TFLOP/s */
int gprx_mfma_f64_peak(int device, double* tflops);

#ifdef __cplusplus
}
#endif
#endif /* GPRX_H */
