// nlsolver_amd/csrc/nlsg_lm_kernels.h — gfx950 kernel of the batched LM engine.
//
// Replaces (reference file:line): LevenbergMarquardt::solve nlsolver.h:3465-3544 with
// Gauss-Newton functors (f = sum r^2, g = 2 J^T r, H = 2 J^T J), math::cholesky /
// forwardsolve_inplace / backsolve_inplace_t / is_diagonal / get_update_with_hessian
// nlsolver.h:251-330.
//
// One persistent 256-thread workgroup per problem runs the whole solve (problems
// converge independently; no lock step). Per evaluation the design matrix A (m x 64 fp64,
// 256 KiB at m = 512) is streamed ONCE from HBM in 64-row blocks:
//   global -> registers (1 KiB coalesced per wave instruction) -> z = A theta by a 32-lane
//   butterfly -> tanh, residual, weight -> scaled Jacobian rows -> LDS (row stride 80 doubles:
//   conflict-free ds_read_b64 for the MFMA operand pattern) -> J^T J on the fp64 matrix
//   cores: 16 tiles of v_mfma_f64_16x16x4_f64 (4 per wave, A operand shared), J^T r on the
//   VALU from the same A operand.
// fp64 MFMA is a k-ordered fma chain (verified on gfx950), so H = 2 * fma-chain over the
// rows in order; the CPU restatement (oracle_lm.c, order = 1) mirrors every sum.
// The damped system is solved in LDS: column-parallel Cholesky (same per-element
// arithmetic as the reference's row order), column-sweep forward/back substitution.
#pragma once

#include "nlsg_common.h"
#include "nlsg_math.h"

namespace nlsg {

typedef double v4d __attribute__((ext_vector_type(4)));

constexpr int kLmN = 64;        // parameters are padded to 64 columns
constexpr int kLmJStride = 80;  // LDS row stride of the Jacobian block (doubles)
constexpr int kLmHStride = 65;  // LDS row stride of the damped matrix (doubles)

struct LmProblem {
  double f, lambda;
  uint64_t iter, fcalls;
  int32_t done, pad;
};

struct LmParams {
  const double *A;   // [batch][m][64]
  const double *y;   // [batch][m]
  double *theta;     // [batch][64]
  LmProblem *prob;   // [batch]
  const double *zero;
  uint64_t batch, m, n, max_iter;
  double lambda0, up, down, f_delta;
};

struct LmQrShared {               // extra LDS of the QR solver
  double Q[64 * kLmHStride];     // the orthogonal factor's transpose, rotated alongside R
  double c[32], s[32];           // Givens coefficients of the current wavefront step
};

struct LmShared {
  double J[64 * kLmJStride];  // scaled Jacobian block
  double H[64 * kLmHStride];  // 2 J^T J (+ lambda I), then its Cholesky factor
  double r[64];               // residuals of the block
  double theta[64], g[64], upd[64], sum[64];
  double gpart[4][64];
  double fpart[8];
  int flag;
};

// f, g, H at sh.theta. Every thread returns f.
__device__ inline double lm_evaluate(const LmParams &p, LmShared &sh, uint64_t pid) {
  const int lane = lane_id();
  const int w = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) >> 6);
  const int half = lane >> 5, lp = lane & 31;
  const double th0 = sh.theta[2 * lp], th1 = sh.theta[2 * lp + 1];
  const double *Ap = p.A + pid * p.m * kLmN;
  const double *yp = p.y + pid * p.m;
  v4d acc[4];
#pragma unroll
  for (int c = 0; c < 4; c++) acc[c] = v4d{0.0, 0.0, 0.0, 0.0};
  double gacc = 0.0, facc = 0.0;
  const uint64_t nblk = (p.m + 63) / 64;
  for (uint64_t blk = 0; blk < nblk; blk++) {
    // ---- stream 64 rows: two rows per wave instruction, 8 instructions per wave
    double2 a[8];
    double yv[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int rb = 2 * (8 * w + k) + half;
      const uint64_t i = blk * 64 + rb;
      const bool in = i < p.m;
      a[k] = *reinterpret_cast<const double2 *>(in ? Ap + i * kLmN + 2 * lp : p.zero);
      yv[k] = *(in ? yp + i : p.zero);
    }
    // z = A theta for the wave's 16 rows: one 32-lane butterfly per load instruction
    double z[8];
#pragma unroll
    for (int k = 0; k < 8; k++) {
      z[k] = a[k].x * th0 + a[k].y * th1;
#pragma unroll
      for (int off = 16; off >= 1; off >>= 1) z[k] = z[k] + __shfl_xor(z[k], off, 64);
    }
    // tanh / residual / weight ONCE per row: lane lp < 8 of each half takes row k = lp
    // (instead of all 32 lanes of the half repeating the same transcendental 8 times)
    double zsel = z[0], ysel = yv[0];
#pragma unroll
    for (int k = 1; k < 8; k++) {
      zsel = (lp == k) ? z[k] : zsel;
      ysel = (lp == k) ? yv[k] : ysel;
    }
    const double tsel = det_tanh(zsel);
    const double rsel = ysel - tsel;
    const double wsel = 1 - tsel * tsel;
    if (lp < 8) sh.r[2 * (8 * w + lp) + half] = rsel;
#pragma unroll
    for (int k = 0; k < 8; k++) {
      const int rb = 2 * (8 * w + k) + half;
      const double r = __shfl(rsel, 32 * half + k, 64);
      const double wgt = __shfl(wsel, 32 * half + k, 64);
      facc = facc + r * r;
      double2 jv;
      jv.x = -(wgt * a[k].x);
      jv.y = -(wgt * a[k].y);
      *reinterpret_cast<double2 *>(&sh.J[rb * kLmJStride + 2 * lp]) = jv;
    }
    __syncthreads();
    // ---- J^T J on the matrix cores, J^T r on the VALU (same A operand)
    const int kk = lane >> 4, cc = lane & 15;
#pragma unroll 4
    for (int ks = 0; ks < 16; ks++) {
      const double *row = &sh.J[(4 * ks + kk) * kLmJStride];
      const double aop = row[16 * w + cc];
      gacc = gacc + aop * sh.r[4 * ks + kk];
#pragma unroll
      for (int c = 0; c < 4; c++)
        acc[c] = __builtin_amdgcn_mfma_f64_16x16x4f64(aop, row[16 * c + cc], acc[c], 0, 0, 0);
    }
    __syncthreads();
  }
  // ---- publish H = 2 J^T J, g = 2 J^T r, f
  {
    const int kk = lane >> 4, cc = lane & 15;
#pragma unroll
    for (int c = 0; c < 4; c++)
#pragma unroll
      for (int rg = 0; rg < 4; rg++)
        sh.H[(16 * w + kk + 4 * rg) * kLmHStride + 16 * c + cc] = 2 * acc[c][rg];
    sh.gpart[kk][16 * w + cc] = gacc;
    if (lp == 0) sh.fpart[2 * w + half] = facc;
  }
  __syncthreads();
  if (threadIdx.x < 64) {
    const int j = threadIdx.x;
    sh.g[j] = 2 * (((sh.gpart[0][j] + sh.gpart[1][j]) + sh.gpart[2][j]) + sh.gpart[3][j]);
  }
  double f = 0.0;
#pragma unroll
  for (int k = 0; k < 8; k++) f = f + sh.fpart[k];
  __syncthreads();
  return f;
}

// get_update_with_hessian (nlsolver.h:310-330) on the n x n leading block of sh.H.
// Runs in ONE wave (lane = matrix row, n <= 64): no workgroup barriers inside; values
// cross lanes through LDS (a wave's DS instructions execute in order) or a lane broadcast.
__device__ inline void lm_solve_cholesky_wave(LmShared &sh, int n) {
  const int t = lane_id();
  const bool row = t < n;
  // is_diagonal (:295-307): any off-diagonal above eps * 1e12 (positive values only)
  bool off = false;
  if (row)
    for (int j = 0; j < n; j++)
      off = off || (j != t && sh.H[t * kLmHStride + j] > 2.220446049250313e-16 * 1e12);
  if (__ballot(off) == 0ull) {
    if (row) sh.upd[t] = sh.g[t] / sh.H[t * kLmHStride + t];
    return;
  }
  // cholesky (:251-269), column by column; every element's sum runs over k in order
  double *Ht = &sh.H[t * kLmHStride];
  for (int j = 0; j < n; j++) {
    const double *Hj = &sh.H[j * kLmHStride];
    double sum = 0;
    if (row && t >= j) {
#pragma unroll 8
      for (int k = 0; k < j; k++) sum += Ht[k] * Hj[k];
    }
    // diagonal first (lane j: sum = sum_k L[j][k]^2), then the column below it
    double d = 0.0;
    if (t == j) d = sqrt(Ht[j] - sum);
    d = __shfl(d, j, 64);
    if (t == j) Ht[j] = d;
    if (row && t > j) Ht[j] = (1.0 / d * (Ht[j] - sum));
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    __builtin_amdgcn_wave_barrier();
  }
  // forwardsolve_inplace (:282-294): column sweep, each row's sum grows in j order
  double sum = 0.0, u = 0.0;
  for (int j = 0; j < n; j++) {
    if (t == j) u = (sh.g[j] - sum) / Ht[j];
    const double uj = __shfl(u, j, 64);
    if (row && t > j) sum += Ht[j] * uj;
  }
  // backsolve_inplace_t (:270-281) with the inner sums taken from j = n-1 down to i+1
  sum = 0.0;
  for (int j = n - 1; j >= 0; j--) {
    if (t == j) u = (u - sum) / Ht[j];
    const double uj = __shfl(u, j, 64);
    if (row && t < j) sum += sh.H[j * kLmHStride + t] * uj;
  }
  if (row) sh.upd[t] = u;
}

// tinyqr::lm on the damped matrix (tinyqr.h:253-310, 437-470): Givens QR in the reference's
// rotation order (column j, rows bottom-up), executed as wavefronts: rotation (j, i) runs at
// step (n-1-i) + 2j, all rotations of a step touch disjoint row pairs, so every element sees
// exactly the sequence of updates the serial loop applies (columns left of j are skipped:
// they only hold annihilated entries that the cleanup pass zeroes and nothing reads).
// sh.H is the working R (row-major, = the transposed-input layout of qr_decomposition).
__device__ inline void lm_solve_qr(LmShared &sh, LmQrShared &qs, int n) {
  const int t = threadIdx.x, lane = lane_id();
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  for (int e = t; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    qs.Q[i * kLmHStride + j] = (i == j) ? 1.0 : 0.0;  // make_identity, tinyqr.h:205-210
  }
  __syncthreads();
  for (int step = 0; step <= 2 * n - 4; step++) {
    const int jlo = step - (n - 2) > 0 ? step - (n - 2) : 0;
    const int jhi = step / 2 < n - 2 ? step / 2 : n - 2;
    const int count = jhi - jlo + 1;
    if (t < count) {  // givens_rotation, tinyqr.h:86-97
      const int j = jlo + t, i = n - 1 - (step - 2 * j);
      const double a = sh.H[(i - 1) * kLmHStride + j], b = sh.H[i * kLmHStride + j];
      double c, sv;
      if (fabs(b) > fabs(a)) {
        const double r = a / b;
        sv = 1.0 / sqrt(r * r + 1.0);
        c = sv * r;
      } else {
        const double r = b / a;
        c = 1.0 / sqrt(r * r + 1.0);
        sv = c * r;
      }
      qs.c[t] = c;
      qs.s[t] = sv;
    }
    __syncthreads();
    for (int r = wid; r < count; r += 4) {  // rotate_matrix on R and Q, tinyqr.h:126-139
      const int j = jlo + r, i = n - 1 - (step - 2 * j);
      const double c = qs.c[r], sv = qs.s[r];
      if (lane >= j && lane < n) {
        double *lo = &sh.H[(i - 1) * kLmHStride + lane], *up = &sh.H[i * kLmHStride + lane];
        const double t1 = *lo, t2 = *up;
        *lo = c * t1 + sv * t2;
        *up = -sv * t1 + c * t2;
      }
      if (lane < n) {
        double *lo = &qs.Q[(i - 1) * kLmHStride + lane], *up = &qs.Q[i * kLmHStride + lane];
        const double t1 = *lo, t2 = *up;
        *lo = c * t1 + sv * t2;
        *up = -sv * t1 + c * t2;
      }
    }
    __syncthreads();
  }
  // cleanup with lm()'s tol = 1e-12 (tinyqr.h:278-282, 465) on the entries back_solve reads
  for (int e = t; e < n * n; e += 256) {
    const int i = e / n, j = e % n;
    if (j >= i && fabs(sh.H[i * kLmHStride + j]) < 1e-12) sh.H[i * kLmHStride + j] = 0.0;
  }
  __syncthreads();
  // back_solve (tinyqr.h:437-459): Q^T y lazily per row, then the triangular sweep with the
  // inner sums taken from j = n-1 down to i+1 (oracle order 1)
  if (t < 64) {
    double ytmp = 0;
    if (t < n)
      for (int j = 0; j < n; j++) ytmp += qs.Q[t * kLmHStride + j] * sh.g[j];
    double temp = 0.0, u = 0.0;
    for (int j = n - 1; j >= 0; j--) {
      if (t == j) u = (ytmp - temp) / sh.H[j * kLmHStride + j];
      const double uj = __shfl(u, j, 64);
      if (t < j) temp += sh.H[t * kLmHStride + j] * uj;
    }
    if (t < n) sh.upd[t] = u;
  }
}

template <bool QR>
__global__ __launch_bounds__(256) void lm_solve_kernel(LmParams p) {
  extern __shared__ __align__(16) unsigned char lm_smem[];
  LmShared &sh = *reinterpret_cast<LmShared *>(lm_smem);
  LmQrShared &qs = *reinterpret_cast<LmQrShared *>(lm_smem + sizeof(LmShared));  // QR only
  const uint64_t pid = blockIdx.x;
  const int t = threadIdx.x;
  const int n = static_cast<int>(p.n);
  if (t < 64) sh.theta[t] = p.theta[pid * kLmN + t];
  __syncthreads();
  double lambda = p.lambda0;
  uint64_t iter = 0, fcalls = 1;
  double cur = lm_evaluate(p, sh, pid);  // g, H, f at x0 (:3513-3516)
  double prev = 0.0;
  for (;;) {
    const double delta = fabs(prev - cur);
    if (iter >= p.max_iter || delta < p.f_delta || isnan(prev)) break;  // :3520-3527
    if (t < n) sh.H[t * kLmHStride + t] += lambda;                      // :3529-3531
    __syncthreads();
    if (QR) {
      lm_solve_qr(sh, qs, n);
      if (t < n) sh.theta[t] = sh.theta[t] - sh.upd[t];  // always accepted, :3534
    } else if (t < 64) {  // wave 0 solves the damped system and applies the step
      lm_solve_cholesky_wave(sh, n);
      if (t < n) sh.theta[t] = sh.theta[t] - sh.upd[t];
    }
    __syncthreads();
    prev = cur;
    cur = lm_evaluate(p, sh, pid);
    fcalls++;
    iter++;
    lambda = cur < prev ? lambda / p.down : lambda * p.up;  // :3541-3542
  }
  if (t < 64) p.theta[pid * kLmN + t] = sh.theta[t];
  if (t == 0) {
    LmProblem *pr = p.prob + pid;
    pr->f = cur;
    pr->lambda = lambda;
    pr->iter = iter;
    pr->fcalls = fcalls;
    pr->done = 1;
  }
}

}  // namespace nlsg
