// Cholesky schedule for ONE large matrix: block columns with the panel chain confined to the diagonal block.
// MEASURED DEAD END (round 2), kept as a tested option ("large_min" of gprx_set_tuning), off by default: N = 16384 36.1 ms
// against 30.7 ms for potrf_lower with a 64 x 64 TAIL tile, N = 8192 9.1 against 6.6 ms.  rocprofv3 of N = 16384: the
// per-block chain -- 16 panels (19 us each), 8 inversion GEMMs (30 us each), copy, triangular GEMM, HEAD -- takes ~2 ms per
// 1024 columns and the bulk update it should hide behind only 1.5 ms on average, so the second half of the factorisation is
// chain-bound; reserving CUs for the chain (CU-masked streams) changed nothing.
//
// potrf_lower (potrf.h) runs every 64-column panel over ALL rows below it.  At N = 16384 that is 256 dependent panel
// launches of up to 128 workgroups each, interleaved with the bulk update of the previous block on a second stream: every
// panel launch waits for workgroup slots that the bulk GEMM's long-running workgroups free one at a time (measured in
// round 1: 63 us per panel instead of 15; the chain, not the MFMA work, bounded the factorisation: 0.57 of the fp64 peak).
// Here, for each outer block J = columns [C, C + w):
//   1. DIAGONAL BLOCK  A[C:C+w, C:C+w] is factored by potrf_lower itself (w <= 1024: 16 small panels and their in-block
//      updates) -- on the CHAIN stream, which owns a few RESERVED compute units (CU mask), so these small dependent
//      launches never queue behind bulk workgroups;
//   2. X = L11^-1 (trtri_lower: bottom-up doubling, GEMMs on the small block), also on the chain stream;
//   3. ROWS BELOW:  L21 = A21 L11^-T  as ONE triangular NT GEMM  A21 <- S X^T  (S = copy of A21; K clipped by B_UPPER) on the
//      GEMM stream (all other CUs) -- the flops that the in-block recursion spent in K = 64 .. 512 updates over every row now
//      run in the main GEMM kernel at K up to w, and each element of the block column is read and written once;
//   4. HEAD(J) (next block's columns) on the GEMM stream, TAIL(J) (everything right of it) on the tail stream.
// The chain of block J + 1 (steps 1-2) runs on its reserved CUs while TAIL(J) fills the rest of the chip.
// Numerics: step 3 multiplies by the explicit inverse of a w x w triangular block (the solves of this library already use
// the 64 x 64 inverses, and predict uses the full L^-1); the factor agrees with the substitution-based schedule to
// cond(L11) * eps and is NOT bit-identical to it, which is why the schedule is chosen by the matrix size alone (single and
// batched calls on a given size always take the same one).
#pragma once
#include "potrf.h"
#include "solve.h"

namespace gprx {


// dst[r][c] = src[r][c] for an (rows x cols) block, 16-byte accesses (cols even, both leading dimensions even): the copy
// of A21 into the GEMM's input buffer at HBM speed (hipMemcpy2DAsync took the slow pitched-copy path: 126 MB in ~2 ms)
__global__ __launch_bounds__(256) void copy_rows_kernel(const double* __restrict__ src, int64_t lds, double* __restrict__ dst, int64_t ldd, int rows,
                                                        int cols) {
  const int chunks = cols / 2;
  const int64_t total = (int64_t)rows * chunks;
  for (int64_t e = blockIdx.x * (int64_t)256 + threadIdx.x; e < total; e += (int64_t)gridDim.x * 256) {
    const int64_t r = e / chunks;
    const int c = (int)(e - r * chunks);
    *reinterpret_cast<d2*>(dst + r * ldd + 2 * c) = *reinterpret_cast<const d2*>(src + r * lds + 2 * c);
  }
}

struct PotrfLarge {
  hipStream_t chain = nullptr, gemm = nullptr, tail = nullptr;
  hipEvent_t ev_in = nullptr, ev_head = nullptr, ev_diag = nullptr, ev_trsm = nullptr, ev_tail = nullptr, ev_chain = nullptr;
  double* S = nullptr;  // (np + extra) x ob copy of the rows below the diagonal block
  double* X = nullptr;  // ob x ob: L11^-1 (upper part stays zero)
  double* T = nullptr;  // ob x ob: scratch of trtri_lower
  size_t s_doubles = 0;
  int ob = 0;
  int reserved = -1;

  hipError_t init_streams() {
    if (chain) return hipSuccess;
    // reserved CUs for the chain stream: bits 0 .. r-1 of the CU mask (the kernel driver deals consecutive mask bits to the
    // XCDs round-robin, so 16 bits = 2 CUs of every XCD); the two GEMM streams get the complement
    static const int env_r = getenv("GPRX_LARGE_RESERVED_CUS") ? atoi(getenv("GPRX_LARGE_RESERVED_CUS")) : 16;
    reserved = env_r;
    hipError_t e;
    if (reserved > 0 && reserved < 128) {
      uint32_t m_chain[8] = {0, 0, 0, 0, 0, 0, 0, 0}, m_rest[8];
      for (int b = 0; b < reserved; ++b) m_chain[b / 32] |= 1u << (b % 32);
      for (int i = 0; i < 8; ++i) m_rest[i] = ~m_chain[i];
      if ((e = hipExtStreamCreateWithCUMask(&chain, 8, m_chain)) != hipSuccess) return e;
      if ((e = hipExtStreamCreateWithCUMask(&gemm, 8, m_rest)) != hipSuccess) return e;
      if ((e = hipExtStreamCreateWithCUMask(&tail, 8, m_rest)) != hipSuccess) return e;
    } else {
      if ((e = hipStreamCreateWithFlags(&chain, hipStreamNonBlocking)) != hipSuccess) return e;
      if ((e = hipStreamCreateWithFlags(&gemm, hipStreamNonBlocking)) != hipSuccess) return e;
      if ((e = hipStreamCreateWithFlags(&tail, hipStreamNonBlocking)) != hipSuccess) return e;
    }
    for (hipEvent_t* ev : {&ev_in, &ev_head, &ev_diag, &ev_trsm, &ev_tail, &ev_chain})
      if ((e = hipEventCreateWithFlags(ev, hipEventDisableTiming)) != hipSuccess) return e;
    return hipSuccess;
  }
  hipError_t ensure(int total_rows, int block) {
    hipError_t e = init_streams();
    if (e != hipSuccess) return e;
    const size_t need = (size_t)total_rows * block;
    if (block != ob || need > s_doubles) {
      for (double** q : {&S, &X, &T})
        if (*q) {
          hipFree(*q);
          *q = nullptr;
        }
      if ((e = hipMalloc((void**)&S, sizeof(double) * need)) != hipSuccess) return e;
      if ((e = hipMalloc((void**)&X, sizeof(double) * block * block)) != hipSuccess) return e;
      if ((e = hipMalloc((void**)&T, sizeof(double) * block * block)) != hipSuccess) return e;
      if ((e = hipMemset(X, 0, sizeof(double) * block * block)) != hipSuccess) return e;  // the strictly upper 64-blocks are never written
      if ((e = hipMemset(T, 0, sizeof(double) * block * block)) != hipSuccess) return e;
      s_doubles = need;
      ob = block;
    }
    return hipSuccess;
  }
  void destroy() {
    for (double** q : {&S, &X, &T})
      if (*q) {
        hipFree(*q);
        *q = nullptr;
      }
    for (hipStream_t* s : {&chain, &gemm, &tail})
      if (*s) {
        hipStreamDestroy(*s);
        *s = nullptr;
      }
    for (hipEvent_t* ev : {&ev_in, &ev_head, &ev_diag, &ev_trsm, &ev_tail, &ev_chain})
      if (*ev) {
        hipEventDestroy(*ev);
        *ev = nullptr;
      }
    s_doubles = 0;
    ob = 0;
  }
};

// Same contract as potrf_lower for ONE matrix (batch = 1): in place, `extra` right-hand-side rows below the matrix, inv_diag
// and diag_stage as there, info zeroed by the caller.  All work is ordered after what `st` holds on entry, and `st` waits for
// all of it on return.
inline hipError_t potrf_lower_large(hipStream_t st, double* A, int64_t lda, int np, int extra, double* inv_diag, int* info, double* diag_stage,
                                    PotrfLarge& w2, const PotrfTuning& tune) {
  const int total_rows = np + extra;
  const int ob = tune.outer_block ? (tune.outer_block > 1024 ? 1024 : tune.outer_block) : 1024;
  hipError_t e = w2.ensure(total_rows, ob);
  if (e != hipSuccess) return e;
  PotrfTuning sub = tune;
  sub.outer_block = 1024;  // the diagonal block is one outer block of the substitution-based schedule
  sub.no_lookahead = 1;
  hipEventRecord(w2.ev_in, st);
  for (hipStream_t s : {w2.chain, w2.gemm, w2.tail}) hipStreamWaitEvent(s, w2.ev_in, 0);
  bool tail_pending = false;
  for (int C = 0; C < np; C += ob) {
    const int w = (np - C < ob) ? np - C : ob;
    const int R = total_rows - C - w;  // rows below the diagonal block (the right-hand-side rows included)
    double* Ajj = A + (int64_t)C * lda + C;
    double* invd = inv_diag + (int64_t)(C / NB) * NB * NB;
    // 1. diagonal block on the chain stream (after HEAD of the previous block, which wrote these columns)
    if (C > 0) hipStreamWaitEvent(w2.chain, w2.ev_head, 0);
    if ((e = potrf_lower(w2.chain, Ajj, lda, w, 0, invd, info, diag_stage + (int64_t)C * STAGE_LD, nullptr, nullptr, 1, 0, 0, &sub, C)) != hipSuccess) return e;
    if (R <= 0) break;
    // 2. X = L11^-1
    if ((e = trtri_lower(w2.chain, Ajj, lda, invd, w2.X, ob, w2.T, ob, w)) != hipSuccess) return e;
    hipEventRecord(w2.ev_diag, w2.chain);
    // 3. rows below: S <- A21, A21 <- S X^T   (gemm stream: already behind HEAD of the previous block)
    double* A21 = A + (int64_t)(C + w) * lda + C;
    hipStreamWaitEvent(w2.gemm, w2.ev_diag, 0);
    hipLaunchKernelGGL(copy_rows_kernel, dim3(2048), dim3(256), 0, w2.gemm, (const double*)A21, lda, w2.S, (int64_t)w, R, w);
    if ((e = launch_gemm(w2.gemm, 0, 1, R, w, w, 1.0, w2.S, w, w2.X, ob, 0.0, A21, lda, GEMM_B_UPPER, 0)) != hipSuccess) return e;
    hipEventRecord(w2.ev_trsm, w2.gemm);
    const int Rn = C + w;  // first column right of this block
    if (Rn >= np) break;   // (only right-hand-side rows were left)
    const int wn = (np - Rn < ob) ? np - Rn : ob;
    const double* Lpan = A + (int64_t)Rn * lda + C;  // L[Rn:, C:C+w]
    // 4a. HEAD(J): columns [Rn, Rn + wn), rows [Rn, total_rows); TAIL(J-1) also wrote them
    if (tail_pending) hipStreamWaitEvent(w2.gemm, w2.ev_tail, 0);
    if ((e = launch_gemm(w2.gemm, 0, 1, total_rows - Rn, wn, w, -1.0, Lpan, lda, Lpan, lda, 1.0, A + (int64_t)Rn * lda + Rn, lda, GEMM_C_LOWER, 64)) != hipSuccess)
      return e;
    hipEventRecord(w2.ev_head, w2.gemm);
    // 4b. TAIL(J): columns [Rn + wn, np) on the tail stream, once HEAD(J) -- the chain's input -- is through
    const int R2 = Rn + wn;
    if (R2 < np) {
      hipStreamWaitEvent(w2.tail, w2.ev_head, 0);
      const double* Lrow = A + (int64_t)R2 * lda + C;
      if ((e = launch_gemm(w2.tail, 0, 1, total_rows - R2, np - R2, w, -1.0, Lrow, lda, Lrow, lda, 1.0, A + (int64_t)R2 * lda + R2, lda, GEMM_C_LOWER,
                           tune.update_tile)) != hipSuccess)
        return e;
      hipEventRecord(w2.ev_tail, w2.tail);
      tail_pending = true;
    }
  }
  // `st` continues when the three streams are done
  hipEventRecord(w2.ev_chain, w2.chain);
  hipEventRecord(w2.ev_head, w2.gemm);
  hipStreamWaitEvent(st, w2.ev_chain, 0);
  hipStreamWaitEvent(st, w2.ev_head, 0);
  if (tail_pending) hipStreamWaitEvent(st, w2.ev_tail, 0);
  return hipGetLastError();
}

}  // namespace gprx
