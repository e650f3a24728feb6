// nlsolver_amd/csrc/nlsg_nm_kernels.h — gfx950 kernel of the batched Nelder-Mead engine.
//
// Replaces (nlsolver.h): NelderMead::solve 2166-2299 and its wrappers 2127-2163, simplex
// ctor 1905-1950, update_centroid 1965-1984, simplex_transform 1986-2007, shrink 2009-2035,
// max_abs_vec 1894-1904, std_err 2037-2052.
//
// One persistent 1024-thread workgroup per start (16 waves: the LDS image allows one workgroup
// per CU, so the waves are what hides latency and shares the vertex rows of a shrink); the (n+1) x n simplex, its scores and the
// work vectors live in LDS (132 KiB at n = 128 of the 160 KiB a CDNA4 CU has), so an
// iteration touches no HBM at all. Past n = 128 (CHUNKS > 1, n <= 128 CHUNKS) the simplex rows
// move to a per-start workspace in global memory (8.4 MB at n = 1024: L2-resident while the
// workgroup — the only reader and writer of it — runs; a workgroup barrier orders its own
// accesses), scores and work vectors stay in LDS; same code, same arithmetic. The method is a chain of data-dependent decisions
// (latency bound): control flow is evaluated by thread 0 and broadcast through
// LDS, vector work is spread over threads (centroid: thread j sums vertex coordinates in the
// reference's vertex order), objectives are evaluated one vertex per wave with the fixed
// lane tree. All reference quirks on the path are reproduced (stale centroid).
#pragma once

#include "nlsg_common.h"

namespace nlsg {

constexpr int kNmThreads = 1024;  // the largest workgroup (n = 128: 16 waves)
// threads of the workgroup of an n-dimensional simplex: one wave per four vertices (the waves
// share the vertex evaluations of the initial scoring and of a shrink; everything else is one
// wave's or n threads' work), so that a batch of small simplexes fills the chip with many small
// workgroups instead of a few mostly idle large ones
__host__ __device__ inline int nm_block_threads(uint64_t n) {
  const uint64_t waves = (n + 1 + 3) / 4;
  return 64 * static_cast<int>(waves < 1 ? 1 : (waves > kNmThreads / 64 ? kNmThreads / 64 : waves));
}

struct NmProblem {
  double f, eps;
  uint64_t iter, fcalls;
};

struct NmParams {
  double *simplex;      // [batch][n + 1][n] workspace, n > 128 only
  double *x;            // [batch][n] in/out
  const double *upper, *lower;  // [n]
  NmProblem *prob;      // [batch]
  uint64_t batch, n, max_iter, no_change_tol, restarts;
  double step, alpha, gamma, rho, sigma, eps, fmul;
  int32_t bounded;
  int32_t seq;  // NLSG_NM_REFERENCE_ORDER: the objective's terms and std_err's two sums in index order; the
                // value caps the waves that take part in a shrink's rescoring (NLSG_NM_SEQ_WAVES, default 16)
  // measurement aid (nlsg_nm_phase_cycles; nullptr otherwise): [batch][kNmPhases] shader-clock
  // cycles the start's decision chain spent per phase, and two counts
  unsigned long long *phase;
};
// phases: 0 scan (std_err, best / worst / second worst, stop tests), 1 centroid, 2 reflection
// (transform, evaluation, decision), 3 expansion or contraction (transform, evaluation, accept),
// 4 shrink + rescoring (all waves, the barriers around it included), 5 -, 6 iterations, 7 shrinks
constexpr int kNmPhases = 8;

struct NmCtl {  // control block in LDS
  double ref_score, exp_score, cont_score, eps, se;
  uint64_t best, worst, second_worst, prev_worst, last_best, no_change, iter, fcalls;
  int stop, shrunk, action;
  int cmd;  // nm_solve_driver_kernel: what the driver wave asks of the others at the next barrier
};

__host__ __device__ inline int nm_chunks(uint64_t n) { return n <= 128 ? 1 : n <= 256 ? 2 : n <= 512 ? 4 : 8; }
__host__ __device__ inline size_t nm_lds_bytes(uint64_t n) {  // the workgroup's LDS image (without term buffers)
  const uint64_t nv = n + 1;
  const uint64_t rows = nm_chunks(n) == 1 ? nv * n : 0;
  return (rows + ((nv + 1) & ~1ull) + 7 * n) * sizeof(double) + sizeof(NmCtl) + 16 +
         kNmPhases * sizeof(unsigned long long);  // (the driver kernel's phase counters)
}

// the point at `pt` in the lane layout of the other engines (element 128 c + 2 lane + k)
template <int CHUNKS>
__device__ inline void nm_load_point(const double *pt, uint64_t n, double (&xv)[CHUNKS][2]) {
  const int lane = lane_id();
#pragma unroll
  for (int c = 0; c < CHUNKS; c++) {
    const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
    xv[c][0] = (e0 < n) ? pt[e0] : 0.0;
    xv[c][1] = (e0 + 1 < n) ? pt[e0 + 1] : 0.0;
  }
}
// objective of the point at `pt` (n <= 128 CHUNKS), evaluated by one wave; all lanes get it.
// seq_buf != nullptr: reference order (the terms added in index order through the wave's buffer)
template <int OBJ, int CHUNKS>
__device__ inline double nm_wave_f(const double *pt, uint64_t n, double fmul, double *seq_buf = nullptr) {
  double xv[CHUNKS][2];
  nm_load_point<CHUNKS>(pt, n, xv);
  if (seq_buf) return fmul * wave_objective_seq_buf<OBJ, CHUNKS>(xv, n, seq_buf);  // (wave-uniform)
  return fmul * wave_objective<OBJ, CHUNKS>(xv, n);
}
// Reference order (NLSG_NM_REFERENCE_ORDER): what separates the device run from the reference's is
// the order of two kinds of sums — the objective's terms and std_err's mean and deviations (the
// centroid already adds the vertices in the reference's order; everything else is per coordinate).
// Ties between vertices that are equal under the reference's sequential sum and one ulp apart under
// the lane tree send the two runs down different branches (Rosenbrock-128D from a constant start
// forks at the 262nd evaluation). With both kinds taken in index order the engine reproduces the
// reference's runs bit for bit: nm_solve_kernel with p.seq, every wave with a term buffer of its own
// behind the workgroup's LDS image (as many waves as have room take part in a shrink's rescoring).
__host__ __device__ inline size_t nm_seq_buffer_bytes(uint64_t n) { return 128ull * nm_chunks(n) * sizeof(double); }
__host__ __device__ inline uint64_t nm_seq_buffers(uint64_t n, uint64_t nwaves, size_t lds_base) {
  const size_t room = 160 * 1024 - ((lds_base + 15) & ~size_t(15)) - 64;  // (64: serial_sum_lds reads ahead)
  const uint64_t fit = room / nm_seq_buffer_bytes(n);
  return fit < nwaves ? fit : nwaves;
}
// dynamic LDS of a launch: the image, and in reference order the term buffers behind it
__host__ __device__ inline size_t nm_launch_lds_bytes(uint64_t n, uint64_t nwaves, bool seq) {
  const size_t base = nm_lds_bytes(n);
  return seq ? ((base + 15) & ~size_t(15)) + nm_seq_buffers(n, nwaves, base) * nm_seq_buffer_bytes(n) + 64 : base;
}

template <int OBJ, int CHUNKS = 1>
__global__ __launch_bounds__(kNmThreads) void nm_solve_kernel(NmParams p) {
  // the workgroup is sized by the host to the simplex (nm_block_threads): 64 .. kNmThreads
  const uint64_t nthreads = blockDim.x, nwaves = blockDim.x >> 6;
  extern __shared__ __align__(16) unsigned char nm_smem[];
  const uint64_t n = p.n, nv = p.n + 1;
  const uint64_t pid = blockIdx.x;
  double *const lds0 = reinterpret_cast<double *>(nm_smem);
  double *S = CHUNKS == 1 ? lds0 : p.simplex + pid * nv * n;  // [nv][n]
  double *scores = CHUNKS == 1 ? lds0 + nv * n : lds0;  // [nv] (padded to even)
  double *centroid = scores + ((nv + 1) & ~1ull);
  double *tr = centroid + n, *te = tr + n, *tc = te + n, *x0 = tc + n, *up = x0 + n, *lo = up + n;
  NmCtl *ctl = reinterpret_cast<NmCtl *>(lo + n);
  const int t = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = lane_id();
  // reference order: term buffers of 128 CHUNKS doubles behind the image (nm_lds_bytes), one per wave
  // as far as the CU's LDS goes; sbuf: this wave's (nullptr: tree order, or no room for this wave)
  const uint64_t seq_waves = p.seq ? nm_seq_buffers(n, static_cast<uint64_t>(p.seq) < nwaves ? p.seq : nwaves, nm_lds_bytes(n)) : 0;
  double *const seq_base = reinterpret_cast<double *>(nm_smem + ((nm_lds_bytes(n) + 15) & ~size_t(15)));
  double *const sbuf = static_cast<uint64_t>(wid) < seq_waves ? seq_base + static_cast<uint64_t>(wid) * (128 * CHUNKS) : nullptr;

  for (uint64_t j = t; j < n; j += nthreads) {
    x0[j] = p.x[pid * n + j];
    up[j] = p.bounded ? p.upper[j] : 0.0;
    lo[j] = p.bounded ? p.lower[j] : 0.0;
  }
  if (t == 0) {
    ctl->eps = p.eps;
    ctl->fcalls = 0;
    ctl->iter = 0;
  }
  __syncthreads();
  uint64_t total_iter = 0;
  double final_f = 0.0;
  // measurement aid (p.phase): what one thread sees of the workgroup's phases, barriers included
  unsigned long long ph[kNmPhases] = {0, 0, 0, 0, 0, 0, 0, 0};
  unsigned long long tk = p.phase ? __builtin_readcyclecounter() : 0;
  auto lap = [&](int k) {
    if (p.phase) {
      const unsigned long long now = __builtin_readcyclecounter();
      ph[k] += now - tk;
      tk = now;
    }
  };

  for (uint64_t run = 0; run <= p.restarts; run++) {
    // ---- simplex ctor (1910-1950) with the effective vertices of SURVEY B1
    double scale = p.step;
    if (p.step < 0) {
      double inf_norm = fabs(x0[0]);  // max_abs_vec, 1894-1904
      for (uint64_t i = 1; i < n; i++) {
        const double a = fabs(x0[i]);
        if (inf_norm < a) inf_norm = a;
      }
      const double a = inf_norm < 1.0 ? 1.0 : inf_norm;
      scale = a < 10 ? a : 10;
    }
    for (uint64_t e = t; e < nv * n; e += nthreads) {
      const uint64_t v = e / n, j = e % n;
      double val = x0[j];
      if (v >= 1 && v < n && j == v) val = val + scale;  // vertex n keeps x (the OOB write)
      if (v == 0 && p.step < 0) {
        const double nn = static_cast<double>(n);
        val = x0[j] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
      }
      S[e] = val;
    }
    for (uint64_t j = t; j < n; j += nthreads) centroid[j] = 0.0;  // :2195
    __syncthreads();
    if (p.seq) {
      if (sbuf)
        for (uint64_t v = wid; v < nv; v += seq_waves) {
          const double f = nm_wave_f<OBJ, CHUNKS>(S + v * n, n, p.fmul, sbuf);
          if (lane == 0) scores[v] = f;
        }
    } else {
      for (uint64_t v = wid; v < nv; v += nwaves) {  // 2184-2186
        const double f = nm_wave_f<OBJ, CHUNKS>(S + v * n, n, p.fmul);
        if (lane == 0) scores[v] = f;
      }
    }
    __syncthreads();
    if (t == 0) {
      ctl->fcalls += nv;
      ctl->eps = ctl->eps * (scores[0] * ctl->eps);  // 2189 (B2)
      ctl->worst = 0;
      ctl->second_worst = 0;
      ctl->prev_worst = 0;
      ctl->last_best = 99999999;
      ctl->no_change = 0;
      ctl->iter = 0;
      ctl->shrunk = 0;
    }
    __syncthreads();

    for (;;) {
      // ---- std_err(scores) and the best / worst / second-worst scan by wave 0
      if (wid == 0) {
        // Two butterfly sweeps, each carrying everything that is ready for it (the shuffles of
        // one level are independent, so their LDS-crossbar latencies overlap):
        //   1: sum of the scores | first index of the minimum | first index of the maximum
        //   2: sum of squared deviations | first index of the maximum of scores[0 .. worst)
        // The serial scan of 2208-2221 (B3) in closed form: best = first index of the minimum,
        // worst = first index of the maximum (an element that lowers the running minimum can
        // never raise the running maximum, so the else-if loses nothing), second = the holder
        // of the running maximum just before `worst` took over. NaN never wins a comparison;
        // a NaN at index 0 freezes all three at 0.
        double acc = 0.0;
        double mnv = __builtin_inf(), mxv = -__builtin_inf();
        uint64_t mni = ~0ull, mxi = ~0ull;
        for (uint64_t i = lane; i < nv; i += 64) {
          const double si = scores[i];
          acc = acc + si;
          argmin_combine(mnv, mni, si, i);
          argmax_combine(mxv, mxi, si, i);
        }
        butterfly_levels<32>([&](auto off) {
          constexpr int o = decltype(off)::value;
          const double oa = lane_xor<o>(acc);
          const double omn = lane_xor<o>(mnv), omx = lane_xor<o>(mxv);
          const uint64_t omni = lane_xor<o>(mni), omxi = lane_xor<o>(mxi);
          acc = acc + oa;
          argmin_combine(mnv, mni, omn, omni);
          argmax_combine(mxv, mxi, omx, omxi);
        });
        if (p.seq) {  // std_err's mean (2037-2052) in index order: every lane walks the scores
          acc = serial_sum_lds(scores, static_cast<int>(nv));
        }
        const double mean = acc / static_cast<double>(nv);
        const bool frozen = isnan(scores[0]);
        const uint64_t worst_i = (frozen || mxi == ~0ull) ? 0 : mxi;
        acc = 0.0;
        double sv = -__builtin_inf();
        uint64_t svi = ~0ull;
        for (uint64_t i = lane; i < nv; i += 64) {
          const double si = scores[i];
          const double d = si - mean;
          acc = acc + d * d;
          if (i < worst_i) argmax_combine(sv, svi, si, i);
        }
        butterfly_levels<32>([&](auto off) {
          constexpr int o = decltype(off)::value;
          const double oa = lane_xor<o>(acc);
          const double osv = lane_xor<o>(sv);
          const uint64_t osvi = lane_xor<o>(svi);
          acc = acc + oa;
          argmax_combine(sv, svi, osv, osvi);
        });
        if (p.seq)  // ... and the squared deviations
          acc = serial_chain_lds(scores, static_cast<int>(nv), 0.0, [mean](double v) {
            const double d = v - mean;
            return d * d;
          });
        const double se = sqrt(acc / static_cast<double>(nv - 1));
        if (lane == 0) {
          const uint64_t best = (frozen || mni == ~0ull) ? 0 : mni;
          const uint64_t worst = worst_i;
          const uint64_t second = (svi == ~0ull) ? 0 : svi;
          ctl->prev_worst = ctl->worst;
          ctl->best = best;
          ctl->worst = worst;
          ctl->second_worst = second;
          ctl->se = se;
          if (ctl->last_best == best) {  // 2223-2230
            ctl->no_change++;
          } else {
            ctl->no_change = 0;
            ctl->last_best = best;
          }
          ctl->stop = (ctl->iter >= p.max_iter || se < ctl->eps ||
                       ctl->no_change >= p.no_change_tol)
                          ? 1
                          : 0;  // 2233-2234
          if (!ctl->stop) ctl->iter++;
        }
      }
      __syncthreads();
      lap(0);
      if (ctl->stop) break;
      const uint64_t best = ctl->best, worst = ctl->worst, second = ctl->second_worst;
      // ---- centroid of all vertices but the worst (1965-1984), only when it can have changed
      if (ctl->prev_worst != worst || ctl->shrunk) {
        for (uint64_t j = t; j < n; j += nthreads) {  // two branch-free runs: loads pipeline
          double c = 0.0;
#pragma unroll 8
          for (uint64_t v = 0; v < worst; v++) c += S[v * n + j];
#pragma unroll 8
          for (uint64_t v = worst + 1; v < nv; v++) c += S[v * n + j];
          centroid[j] = c / static_cast<double>(nv - 1);
        }
      }
      __syncthreads();
      lap(1);
      // ---- reflect (2245): c + alpha (c - p), clamped when bounded
      for (uint64_t j = t; j < n; j += nthreads) {
        double v = centroid[j] + p.alpha * (centroid[j] - S[worst * n + j]);
        if (p.bounded) v = v < lo[j] ? lo[j] : (up[j] < v ? up[j] : v);
        tr[j] = v;
      }
      __syncthreads();
      if (wid == 0) {
        const double rs = nm_wave_f<OBJ, CHUNKS>(tr, n, p.fmul, sbuf);
        if (lane == 0) {
          ctl->ref_score = rs;
          ctl->fcalls++;
          ctl->shrunk = 0;
          // 0 accept reflection, 1 expand, 2 contract
          ctl->action = (rs >= scores[best] && rs < scores[second]) ? 0 : (rs < scores[best] ? 1 : 2);
        }
      }
      __syncthreads();
      const int action = ctl->action;
      const double ref_score = ctl->ref_score;
      lap(2);
      if (action == 0) {  // 2251-2253
        for (uint64_t j = t; j < n; j += nthreads) S[worst * n + j] = tr[j];
        if (t == 0) scores[worst] = ref_score;
      } else if (action == 1) {  // expand, 2255-2265: c + gamma (reflected - c)
        for (uint64_t j = t; j < n; j += nthreads) {
          double v = centroid[j] + p.gamma * (tr[j] - centroid[j]);
          if (p.bounded) v = v < lo[j] ? lo[j] : (up[j] < v ? up[j] : v);
          te[j] = v;
        }
        __syncthreads();
        if (wid == 0) {
          const double es = nm_wave_f<OBJ, CHUNKS>(te, n, p.fmul, sbuf);
          if (lane == 0) {
            ctl->exp_score = es;
            ctl->fcalls++;
          }
        }
        __syncthreads();
        const bool take_exp = ctl->exp_score < ref_score;
        for (uint64_t j = t; j < n; j += nthreads) S[worst * n + j] = take_exp ? te[j] : tr[j];
        if (t == 0) scores[worst] = take_exp ? ctl->exp_score : ref_score;
      } else {  // contraction, 2266-2297 (B4: the reflect transform for both kinds)
        const bool outside = ref_score < scores[worst];
        for (uint64_t j = t; j < n; j += nthreads) {
          const double pt = outside ? tr[j] : S[worst * n + j];
          double v = centroid[j] + p.rho * (centroid[j] - pt);
          if (p.bounded) v = v < lo[j] ? lo[j] : (up[j] < v ? up[j] : v);
          tc[j] = v;
        }
        __syncthreads();
        if (wid == 0) {
          const double cs = nm_wave_f<OBJ, CHUNKS>(tc, n, p.fmul, sbuf);
          if (lane == 0) {
            ctl->cont_score = cs;
            ctl->fcalls++;
          }
        }
        __syncthreads();
        const double cont_score = ctl->cont_score;
        const double worst_score = scores[worst];
        if (cont_score < (outside ? ref_score : worst_score)) {
          __syncthreads();  // every thread has read scores[worst]
          for (uint64_t j = t; j < n; j += nthreads) S[worst * n + j] = tc[j];
          if (t == 0) scores[worst] = cont_score;
        } else {  // shrink (2009-2035) and rescoring (2288-2294)
          __syncthreads();
          // One pass per vertex: the wave that owns row v shrinks it towards the best vertex
          // and scores it from the registers it holds; four rows at a time so that their
          // lane trees overlap (the method spends most iterations here on Rosenbrock-128D:
          // the reference's second-worst rule, SURVEY B3, rarely accepts a reflection).
          if (p.seq) {  // reference order: a row at a time on the waves that have a term buffer
            if (sbuf) {
              double bv[CHUNKS][2];
              nm_load_point<CHUNKS>(S + best * n, n, bv);
              for (uint64_t v = wid; v < nv; v += seq_waves) {
                if (v == best) continue;
                double *row = S + v * n;
                double ov[CHUNKS][2], xv[CHUNKS][2];
                nm_load_point<CHUNKS>(row, n, ov);
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) {
                  const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
                  const bool in0 = e0 < n, in1 = e0 + 1 < n;
                  xv[c][0] = bv[c][0] + p.sigma * (ov[c][0] - bv[c][0]);
                  xv[c][1] = bv[c][1] + p.sigma * (ov[c][1] - bv[c][1]);
                  if (in0) row[e0] = xv[c][0];
                  if (in1) row[e0 + 1] = xv[c][1];
                  if (!in0) xv[c][0] = 0.0;
                  if (!in1) xv[c][1] = 0.0;
                }
                const double f = p.fmul * wave_objective_seq_buf<OBJ, CHUNKS>(xv, n, sbuf);
                if (lane == 0) scores[v] = f;
              }
            }
          } else {
            // ROWS rows at a time per wave (register budget: ROWS x CHUNKS x 2 doubles)
            constexpr int ROWS = CHUNKS == 1 ? 4 : CHUNKS == 2 ? 2 : 1;
            double bv[CHUNKS][2];
            nm_load_point<CHUNKS>(S + best * n, n, bv);
            for (uint64_t v0 = wid; v0 < nv; v0 += ROWS * nwaves) {
              double xv[ROWS][CHUNKS][2];
#pragma unroll
              for (int q = 0; q < ROWS; q++) {
                const uint64_t v = v0 + nwaves * q;
                const bool live = v < nv && v != best;
                double *row = S + (live ? v : best) * n;
                double ov[CHUNKS][2];
                nm_load_point<CHUNKS>(row, n, ov);
#pragma unroll
                for (int c = 0; c < CHUNKS; c++) {
                  const uint64_t e0 = static_cast<uint64_t>(c) * 128 + 2 * static_cast<uint64_t>(lane);
                  const bool in0 = e0 < n, in1 = e0 + 1 < n;
                  xv[q][c][0] = bv[c][0] + p.sigma * (ov[c][0] - bv[c][0]);
                  xv[q][c][1] = bv[c][1] + p.sigma * (ov[c][1] - bv[c][1]);
                  if (live && in0) row[e0] = xv[q][c][0];
                  if (live && in1) row[e0 + 1] = xv[q][c][1];
                  if (!in0) xv[q][c][0] = 0.0;
                  if (!in1) xv[q][c][1] = 0.0;
                }
              }
              double f[ROWS];
#pragma unroll
              for (int q = 0; q < ROWS; q++) f[q] = p.fmul * wave_objective<OBJ, CHUNKS>(xv[q], n);
#pragma unroll
              for (int q = 0; q < ROWS; q++) {
                const uint64_t v = v0 + nwaves * q;
                if (lane == 0 && v < nv && v != best) scores[v] = f[q];
              }
            }
          }
          if (t == 0) {
            ctl->fcalls += nv - 1;
            ctl->shrunk = 1;
          }
          if (p.phase) __syncthreads();  // (uniform) so that the shrink's time is its own
          ph[7] += p.phase ? 1 : 0;
          lap(4);
        }
      }
      __syncthreads();
      lap(3);
    }
    // x = current_simplex.vals[best] (2235); restarts continue from it (2129-2132)
    const uint64_t best = ctl->best;
    for (uint64_t j = t; j < n; j += nthreads) x0[j] = S[best * n + j];
    total_iter += ctl->iter;
    ph[6] += p.phase ? ctl->iter : 0;
    final_f = scores[best];
    __syncthreads();
  }
  for (uint64_t j = t; j < n; j += nthreads) p.x[pid * n + j] = x0[j];
  if (t == 0) {
    NmProblem *pr = p.prob + pid;
    pr->f = final_f;
    pr->eps = ctl->eps;
    pr->iter = total_iter;
    pr->fcalls = ctl->fcalls;
    if (p.phase)
      for (int k = 0; k < kNmPhases; k++) p.phase[pid * kNmPhases + k] = ph[k];
  }
}

// ---- n <= 128: the same method with ONE wave driving it. An iteration of Nelder-Mead is a chain
// of data-dependent decisions, each behind an objective evaluation; in nm_solve_kernel every link
// of that chain is a workgroup phase (decide on one lane, broadcast through LDS, barrier — eight
// barriers of up to sixteen waves per iteration, which is where its time goes: profiles/r03).
// Here wave 0 (the driver) keeps the chain in its registers: the scan's results are wave-uniform
// by construction (butterflies), the centroid and the trial points live in the lane layout (lane l
// owns coordinates 2l, 2l+1: the centroid's sums run over the vertices in the reference's order,
// two chains per lane), trial points are scored from registers, decisions need no broadcast. The
// other waves sleep at a barrier and are woken only for what is parallel: the shrink with its
// n rescorings (and the run's start and end). Same arithmetic per coordinate, same lane trees:
// the bits of nm_solve_kernel and of the oracle.
constexpr int kNmCmdShrink = 1, kNmCmdEnd = 2;

// The shrink (2009-2035) and rescoring (2288-2294) of the rows a wave owns, four at a time: rows
// wid + nwaves q, q = 0 .. 3, then the same 4 nwaves further on. The four objectives are reduced
// together (wave_sum4: the butterflies' own pairs, a third of their instructions — the rescoring
// is bound by vector issue: n objective evaluations of ~60 instructions each per shrink).
// VEC: n even — a lane's pair is 16-byte aligned in every row: one 128-bit LDS access each way
// FULL: n = 128 — every lane holds two coordinates and the row length is a compile-time constant
// (no masks on the loads, the objective's own lane tests fold away)
template <int OBJ, bool VEC, bool FULL = false>
__device__ inline void nm_shrink_rows_impl(double *S, double *scores, uint64_t n64, uint64_t nv64, uint64_t best64,
                                           double sigma, double fmul, int wid, uint64_t nwaves64) {
  using O = Objective<OBJ>;
  const int lane = lane_id();
  // 32-bit indices (n <= 1024: a row offset fits easily): 64-bit products and compares for every
  // row were a third of this loop's scalar instructions
  const uint32_t n = FULL ? 128u : static_cast<uint32_t>(n64), nv = FULL ? 129u : static_cast<uint32_t>(nv64);
  const uint32_t best = static_cast<uint32_t>(best64), nwaves = static_cast<uint32_t>(nwaves64);
  const uint32_t e0 = 2 * static_cast<uint32_t>(lane);
  const bool in0 = FULL || e0 < n, in1 = FULL || e0 + 1 < n;
  auto load_pair = [&](const double *row, double (&v)[1][2]) {
    if constexpr (VEC) {
      const double2 q = *reinterpret_cast<const double2 *>(row + (in0 ? e0 : 0));
      v[0][0] = in0 ? q.x : 0.0;
      v[0][1] = in0 ? q.y : 0.0;
    } else {
      nm_load_point<1>(row, n, v);
    }
  };
  auto store_pair = [&](double *row, double a, double b) {
    if constexpr (VEC) {
      if (in0) *reinterpret_cast<double2 *>(row + e0) = make_double2(a, b);
    } else {
      if (in0) row[e0] = a;
      if (in1) row[e0 + 1] = b;
    }
  };
  double bv[1][2];
  load_pair(S + best * n, bv);
  // a tail of at most one row per wave (n = 128: row 128 of 129) is not worth a pass of four on
  // the wave that would get it: those rows go one each to the LAST waves, scored on their own
  const uint32_t full = nv / (4 * nwaves) * (4 * nwaves);
  const uint32_t tail = nv - full <= nwaves ? nv - full : 0;
  if (tail && static_cast<uint32_t>(wid) + tail >= nwaves) {
    const uint32_t v = full + (nwaves - 1 - static_cast<uint32_t>(wid));
    if (v != best) {
      double *row = S + v * n;
      double ov[1][2];
      load_pair(row, ov);
      double pt[1][2];
      pt[0][0] = bv[0][0] + sigma * (ov[0][0] - bv[0][0]);
      pt[0][1] = bv[0][1] + sigma * (ov[0][1] - bv[0][1]);
      store_pair(row, pt[0][0], pt[0][1]);
      // (coordinates past n are zero in both loaded points, so their transform is +0 already)
      const double fv = fmul * wave_objective<OBJ, 1>(pt, n);
      if (lane == 0) scores[v] = fv;
    }
  }
  const uint32_t end = tail ? full : nv;
  for (uint32_t v0 = static_cast<uint32_t>(wid); v0 < end; v0 += 4 * nwaves) {
    double xv[4][2];
#pragma unroll
    for (int q = 0; q < 4; q++) {
      const uint32_t v = v0 + nwaves * static_cast<uint32_t>(q);
      const bool live = v < end && v != best;
      double *row = S + (live ? v : best) * n;
      double ov[1][2];
      load_pair(row, ov);
      xv[q][0] = bv[0][0] + sigma * (ov[0][0] - bv[0][0]);
      xv[q][1] = bv[0][1] + sigma * (ov[0][1] - bv[0][1]);
      if (live) store_pair(row, xv[q][0], xv[q][1]);
    }
    double f;
    if constexpr (O::kWhole) {  // a whole-vector user objective reduces inside its own body
      double fq[4];
#pragma unroll
      for (int q = 0; q < 4; q++) {
        const double pt[1][2] = {{xv[q][0], xv[q][1]}};
        fq[q] = fmul * wave_objective<OBJ, 1>(pt, n);
      }
      const int g = lane >> 4;
      f = g == 0 ? fq[0] : g == 1 ? fq[1] : g == 2 ? fq[2] : fq[3];
    } else {
      double part[4];
#pragma unroll
      for (int q = 0; q < 4; q++) part[q] = wave_objective_partial<OBJ>(xv[q][0], xv[q][1], n);
      f = fmul * O::finish(wave_sum4(part[0], part[1], part[2], part[3]), n);
    }
    // lanes 16 q .. 16 q + 15 hold row q's value
    const uint32_t v = v0 + nwaves * static_cast<uint32_t>(lane >> 4);
    if ((lane & 15) == 0 && v < end && v != best) scores[v] = f;
  }
}
// Reference order, many rows at once (LDS-resident simplex): a LANE per row. Row r's objective is a
// serial chain over its terms — but 64 rows are 64 independent chains, one v_add_f64 advances them
// all. Lane l walks row 64 wid + l and computes its terms from the row itself; it runs k (l mod 32)
// steps behind lane 0 (k = 1 or 2, whichever makes n - k odd), so that at any step a half-wave reads
// (n - k) l + s: 32 different banks instead of one (a row is n doubles: same column = same bank).
// (A term buffer per wave and one row per wave at a time was LDS-bound: an LDS read at a uniform
// address costs its 64 lane slots whatever the exec mask — 77 K cycles per shrink at n = 128.)
// best: the row that keeps its score (~0: none).
template <int OBJ>
__device__ inline void nm_rescore_lanes(const double *S, double *scores, uint64_t n64, uint64_t nv64, uint64_t best,
                                        double fmul, int wid) {
  using O = Objective<OBJ>;
  const int lane = lane_id();
  const int n = static_cast<int>(n64), nv = static_cast<int>(nv64), nt = static_cast<int>(O::n_terms(n64));
  const int r = 64 * wid + lane;
  const int k = (n & 1) ? 2 : 1;
  const double *row = S + (r < nv ? r : nv - 1) * n;
  double acc = 0.0;
  // the skew spans a half-wave (32 lanes = the 32 banks; lanes l and l + 32 share a bank either way)
  const int kl = k * (lane & 31);
  const int steps = nt + 31 * k;
  // three stretches: the lanes join one by one, all lanes inside their windows (no mask, no clamp:
  // most of the steps when n is large), the lanes leave one by one
  const bool mid = 31 * k < nt;
  const int a_end = mid ? 31 * k : steps, b_end = mid ? nt : steps;
  auto masked = [&](int s) {
    const int e = s - kl;
    if (e >= 0 && e < nt) {
      const double xe = row[e];
      const double xn = O::kChain ? row[e + 1] : 0.0;
      acc = acc + O::term(xe, xn);
    }
  };
  int s = 0;
#pragma unroll 4
  for (; s < a_end; s++) masked(s);
  {
    const double *at = row + (s - kl);  // element e of this lane at step s, one further per step
#pragma unroll 8
    for (; s < b_end; s++) {
      const double xe = at[0];
      const double xn = O::kChain ? at[1] : 0.0;
      acc = acc + O::term(xe, xn);
      at++;
    }
  }
#pragma unroll 4
  for (; s < steps; s++) masked(s);
  if (r < nv && static_cast<uint64_t>(r) != best) scores[r] = fmul * O::finish(acc, n64);
}
// the shrink (2009-2035) and rescoring (2288-2294) in reference order: every wave transforms its rows,
// then (a workgroup barrier later) the first ceil(nv / 64) waves rescore them, a lane per row.
// Called by all waves of the workgroup.
template <int OBJ>
__device__ inline void nm_shrink_rows_seq(double *S, double *scores, uint64_t n, uint64_t nv, uint64_t best,
                                          double sigma, double fmul, int wid, uint64_t nwaves) {
  const int lane = lane_id();
  const uint64_t e0 = 2 * static_cast<uint64_t>(lane);
  const bool in0 = e0 < n, in1 = e0 + 1 < n;
  double bv[1][2];
  nm_load_point<1>(S + best * n, n, bv);
  for (uint64_t v = wid; v < nv; v += nwaves) {
    if (v == best) continue;
    double *row = S + v * n;
    double ov[1][2];
    nm_load_point<1>(row, n, ov);
    if (in0) row[e0] = bv[0][0] + sigma * (ov[0][0] - bv[0][0]);
    if (in1) row[e0 + 1] = bv[0][1] + sigma * (ov[0][1] - bv[0][1]);
  }
  __syncthreads();
  if (64ull * static_cast<uint64_t>(wid) < nv) nm_rescore_lanes<OBJ>(S, scores, n, nv, best, fmul, wid);
}
template <int OBJ>
__device__ inline void nm_shrink_rows(double *S, double *scores, uint64_t n, uint64_t nv, uint64_t best,
                                      double sigma, double fmul, int wid, uint64_t nwaves) {
  if (n == 128)
    nm_shrink_rows_impl<OBJ, true, true>(S, scores, n, nv, best, sigma, fmul, wid, nwaves);
  else if ((n & 1) == 0)
    nm_shrink_rows_impl<OBJ, true>(S, scores, n, nv, best, sigma, fmul, wid, nwaves);
  else
    nm_shrink_rows_impl<OBJ, false>(S, scores, n, nv, best, sigma, fmul, wid, nwaves);
}

template <int OBJ, bool SEQ = false>  // SEQ: NLSG_NM_REFERENCE_ORDER (nm_wave_f above)
__global__ __launch_bounds__(kNmThreads) void nm_solve_driver_kernel(NmParams p) {
  const uint64_t nthreads = blockDim.x, nwaves = blockDim.x >> 6;
  extern __shared__ __align__(16) unsigned char nm_smem[];
  const uint64_t n = p.n, nv = p.n + 1;
  const uint64_t pid = blockIdx.x;
  double *const S = reinterpret_cast<double *>(nm_smem);  // [nv][n]
  double *scores = S + nv * n;                            // [nv] (padded to even)
  double *centroid = scores + ((nv + 1) & ~1ull);         // (layout of nm_lds_bytes; unused vectors stay)
  double *x0 = centroid + 4 * n, *up = x0 + n, *lo = up + n;
  NmCtl *ctl = reinterpret_cast<NmCtl *>(lo + n);
  // phase counters (measurement aid) live in LDS behind the control block: the workgroup's size
  // caps a thread at 128 registers and the driver wave needs them all
  unsigned long long *ph = reinterpret_cast<unsigned long long *>(ctl + 1);
  const int t = threadIdx.x;
  const int wid = __builtin_amdgcn_readfirstlane(t >> 6);
  const int lane = lane_id();
  const uint64_t e0 = 2 * static_cast<uint64_t>(lane), e1 = e0 + 1;
  const bool in0 = e0 < n, in1 = e1 < n;
  // reference order: a term buffer of 128 doubles per wave behind the image (as in nm_solve_kernel)
  const uint64_t seq_waves = SEQ ? nm_seq_buffers(n, static_cast<uint64_t>(p.seq) < nwaves ? p.seq : nwaves, nm_lds_bytes(n)) : 0;
  double *const seq_base = reinterpret_cast<double *>(nm_smem + ((nm_lds_bytes(n) + 15) & ~size_t(15)));
  double *const sbuf = SEQ && static_cast<uint64_t>(wid) < seq_waves ? seq_base + static_cast<uint64_t>(wid) * 128 : nullptr;

  for (uint64_t j = t; j < n; j += nthreads) {
    x0[j] = p.x[pid * n + j];
    up[j] = p.bounded ? p.upper[j] : 0.0;
    lo[j] = p.bounded ? p.lower[j] : 0.0;
  }
  if (t == 0) {
    ctl->eps = p.eps;
    ctl->fcalls = 0;
    ctl->iter = 0;
    for (int k = 0; k < kNmPhases; k++) ph[k] = 0;
  }
  __syncthreads();
  uint64_t total_iter = 0;
  double final_f = 0.0;

  for (uint64_t run = 0; run <= p.restarts; run++) {
    // ---- simplex ctor (1910-1950) with the effective vertices of SURVEY B1
    double scale = p.step;
    if (p.step < 0) {
      double inf_norm = fabs(x0[0]);  // max_abs_vec, 1894-1904
      for (uint64_t i = 1; i < n; i++) {
        const double a = fabs(x0[i]);
        if (inf_norm < a) inf_norm = a;
      }
      const double a = inf_norm < 1.0 ? 1.0 : inf_norm;
      scale = a < 10 ? a : 10;
    }
    for (uint64_t e = t; e < nv * n; e += nthreads) {
      const uint64_t v = e / n, j = e % n;
      double val = x0[j];
      if (v >= 1 && v < n && j == v) val = val + scale;  // vertex n keeps x (the OOB write)
      if (v == 0 && p.step < 0) {
        const double nn = static_cast<double>(n);
        val = x0[j] + ((1.0 - sqrt(nn + 1.0)) / nn * scale);
      }
      S[e] = val;
    }
    __syncthreads();
    if constexpr (SEQ) {
      if (64ull * static_cast<uint64_t>(wid) < nv) nm_rescore_lanes<OBJ>(S, scores, n, nv, ~0ull, p.fmul, wid);
    } else {
      for (uint64_t v = wid; v < nv; v += nwaves) {  // 2184-2186
        const double f = nm_wave_f<OBJ, 1>(S + v * n, n, p.fmul);
        if (lane == 0) scores[v] = f;
      }
    }
    __syncthreads();

    if (wid != 0) {
      // ---- the other waves: asleep at a barrier until the driver has work for all
      for (;;) {
        __syncthreads();  // (A) the driver's request is in ctl
        if (ctl->cmd == kNmCmdEnd) break;
        if constexpr (SEQ)
          nm_shrink_rows_seq<OBJ>(S, scores, n, nv, ctl->best, p.sigma, p.fmul, wid, nwaves);
        else
          nm_shrink_rows<OBJ>(S, scores, n, nv, ctl->best, p.sigma, p.fmul, wid, nwaves);
        __syncthreads();  // (B) every row is shrunk and rescored
      }
    } else {
      // ---- the driver wave. Everything below is wave-uniform except the coordinates.
      double eps = ctl->eps;
      eps = eps * (scores[0] * eps);  // 2189 (B2)
      uint64_t fcalls = ctl->fcalls + nv, iter = 0, no_change = 0;
      // (vertex indices in 32 bits: the lone driver wave pays an issue slot for every scalar
      // instruction of a 64-bit product or compare as well)
      uint32_t worst = 0, prev_worst = 0, best = 0, second = 0, last_best = 99999999u;
      int shrunk = 0;
      double c0 = 0.0, c1 = 0.0;  // the centroid, zeroed at :2195
      unsigned long long tk = p.phase ? __builtin_readcyclecounter() : 0;
      auto lap = [&](int k) {
        if (p.phase) {
          const unsigned long long now = __builtin_readcyclecounter();
          if (lane == 0) ph[k] += now - tk;
          tk = now;
        }
      };
      auto clamp = [&](double v, double l, double u) { return v < l ? l : (u < v ? u : v); };
      // (bounds are read from LDS where a bounded transform needs them: no registers held for them)
      auto clamp2 = [&](double &a, double &b) {
        const uint64_t i0 = in0 ? e0 : 0, i1 = in1 ? e1 : 0;
        a = clamp(a, lo[i0], up[i0]);
        b = clamp(b, lo[i1], up[i1]);
      };
      auto score = [&](double a, double b) {
        const double xv[1][2] = {{in0 ? a : 0.0, in1 ? b : 0.0}};
        if constexpr (SEQ) return p.fmul * wave_objective_seq_buf<OBJ, 1>(xv, n, sbuf);  // (wave 0 always has a buffer)
        return p.fmul * wave_objective<OBJ, 1>(xv, n);
      };
      for (;;) {
        // ---- std_err(scores) and the best / worst / second-worst scan (the closed form of
        // nm_solve_kernel). The extrema travel as VALUES only (one min / max per butterfly level
        // instead of a value-and-index pair with its compound compare); the first index that holds
        // an extremum is then read off three ballots: lane l holds indices l, l + 64, l + 128, so
        // the lowest index is the lowest lane of the lowest round that has a holder. A NaN never
        // equals anything and min / max skip it: NaN never wins, as in the serial scan.
        auto first_holder = [&](double target, const double (&val)[3], uint32_t limit) -> uint32_t {
          uint32_t found = ~0u;
#pragma unroll
          for (int q = 2; q >= 0; q--) {
            const uint32_t i = static_cast<uint32_t>(lane) + 64u * q;
            const uint64_t mask = __ballot(i < limit && val[q] == target);
            if (mask) found = static_cast<uint32_t>(__builtin_ctzll(mask)) + 64u * q;
          }
          return found;
        };
        const uint32_t nv32 = static_cast<uint32_t>(nv);
        double acc = 0.0;
        double mnv = __builtin_inf(), mxv = -__builtin_inf();
        double sc[3] = {0.0, 0.0, 0.0};  // the lane's scores (nv <= 129: at most three)
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const uint32_t i = static_cast<uint32_t>(lane) + 64u * q;
          if (i < nv32) {
            const double si = scores[i];
            sc[q] = si;
            acc = acc + si;
            mnv = __builtin_fmin(mnv, si);  // (one instruction; a NaN never wins, as in the compare-and-keep form)
            mxv = __builtin_fmax(mxv, si);
          }
        }
        butterfly_levels<32>([&](auto off) {
          constexpr int o = decltype(off)::value;
          const double oa = lane_xor<o>(acc);
          const double omn = lane_xor<o>(mnv), omx = lane_xor<o>(mxv);
          acc = acc + oa;
          mnv = __builtin_fmin(mnv, omn);
          mxv = __builtin_fmax(mxv, omx);
        });
        if constexpr (SEQ) {  // std_err's mean (2037-2052) in index order: every lane walks the scores
          acc = serial_sum_lds(scores, static_cast<int>(nv32));
        }
        const double mean = acc / static_cast<double>(nv);
        const bool frozen = isnan(scores[0]);
        const uint32_t mni = first_holder(mnv, sc, nv32), mxi = first_holder(mxv, sc, nv32);
        const uint32_t worst_i = (frozen || mxi == ~0u) ? 0 : mxi;
        acc = 0.0;
        double sv = -__builtin_inf();
#pragma unroll
        for (int q = 0; q < 3; q++) {
          const uint32_t i = static_cast<uint32_t>(lane) + 64u * q;
          if (i < nv32) {
            const double d = sc[q] - mean;
            acc = acc + d * d;
            if (i < worst_i) sv = __builtin_fmax(sv, sc[q]);
          }
        }
        butterfly_levels<32>([&](auto off) {
          constexpr int o = decltype(off)::value;
          const double oa = lane_xor<o>(acc);
          const double osv = lane_xor<o>(sv);
          acc = acc + oa;
          sv = __builtin_fmax(sv, osv);
        });
        const uint32_t svi = first_holder(sv, sc, worst_i);
        if constexpr (SEQ)  // ... and the squared deviations
          acc = serial_chain_lds(scores, static_cast<int>(nv32), 0.0, [mean](double v) {
            const double d = v - mean;
            return d * d;
          });
        const double se = sqrt(acc / static_cast<double>(nv - 1));
        prev_worst = worst;
        best = (frozen || mni == ~0u) ? 0 : mni;
        worst = worst_i;
        second = (svi == ~0u) ? 0 : svi;
        // the butterflies leave the same bits in every lane: make that visible to the compiler
        best = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(best)));
        worst = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(worst)));
        second = static_cast<uint32_t>(__builtin_amdgcn_readfirstlane(static_cast<int>(second)));
        if (last_best == best) {  // 2223-2230
          no_change++;
        } else {
          no_change = 0;
          last_best = best;
        }
        const bool stop = iter >= p.max_iter || se < eps || no_change >= p.no_change_tol;  // 2233-2234
        lap(0);
        if (__builtin_amdgcn_readfirstlane(stop ? 1 : 0)) break;
        iter++;
        // ---- centroid of all vertices but the worst (1965-1984), only when it can have changed:
        // per coordinate the vertices in order, two branch-free runs around the worst one
        if (prev_worst != worst || shrunk) {
          // the lane's two coordinates are adjacent: for an even n one 128-bit LDS read per vertex
          // (conflict-free: the lanes' pairs tile the row), sixteen vertices in flight ahead of
          // their additions; two branch-free runs around the worst vertex. The pair past the
          // point's end is read and never used. 32-bit element indices.
          const uint32_t n32 = static_cast<uint32_t>(n);
          const uint32_t off0 = in0 ? static_cast<uint32_t>(e0) : 0u;
          double a0 = 0.0, a1 = 0.0;
          // STRIDE: the row length when it is known at compile time (128, 64: the offsets of a
          // block's reads are then immediates of the instructions), 0: n32
          auto run = [&](auto vec, auto stride, uint32_t lo_v, uint32_t hi_v) {
            constexpr bool VEC = decltype(vec)::value != 0;
            constexpr uint32_t STRIDE = decltype(stride)::value;
            const uint32_t rs = STRIDE ? STRIDE : n32;
            uint32_t v = lo_v, base = lo_v * rs + off0;
            // C consecutive vertices: their reads in flight together, then their additions in order
            auto block = [&](auto count) {
              constexpr int C = decltype(count)::value;
              double q0[C], q1[C];
#pragma unroll
              for (int u = 0; u < C; u++) {
                if constexpr (VEC) {
                  const double2 pr = *reinterpret_cast<const double2 *>(S + base + static_cast<uint32_t>(u) * rs);
                  q0[u] = pr.x;
                  q1[u] = pr.y;
                } else {
                  q0[u] = S[base + static_cast<uint32_t>(u) * rs];
                  q1[u] = S[base + static_cast<uint32_t>(u) * rs + 1];
                }
              }
#pragma unroll
              for (int u = 0; u < C; u++) {
                a0 += q0[u];
                a1 += q1[u];
              }
              v += C;
              base += C * rs;
            };
            while (v + 16 <= hi_v) block(int_c<16>{});
            // the rest of the run in blocks of 8, 4, 2, 1 by the bits of its length (a tail walked
            // vertex by vertex paid an LDS round trip per vertex)
            const uint32_t rest = hi_v - v;
            if (rest & 8) block(int_c<8>{});
            if (rest & 4) block(int_c<4>{});
            if (rest & 2) block(int_c<2>{});
            if (rest & 1) block(int_c<1>{});
          };
          const uint32_t w32 = static_cast<uint32_t>(worst), nv32c = static_cast<uint32_t>(nv);
          if (n32 == 128) {
            run(int_c<1>{}, int_c<128>{}, 0, w32);
            run(int_c<1>{}, int_c<128>{}, w32 + 1, nv32c);
          } else if (n32 == 64) {
            run(int_c<1>{}, int_c<64>{}, 0, w32);
            run(int_c<1>{}, int_c<64>{}, w32 + 1, nv32c);
          } else if ((n32 & 1) == 0) {
            run(int_c<1>{}, int_c<0>{}, 0, w32);
            run(int_c<1>{}, int_c<0>{}, w32 + 1, nv32c);
          } else {
            run(int_c<0>{}, int_c<0>{}, 0, w32);
            run(int_c<0>{}, int_c<0>{}, w32 + 1, nv32c);
          }
          c0 = a0 / static_cast<double>(nv - 1);
          c1 = a1 / static_cast<double>(nv - 1);
        }
        lap(1);
        // ---- reflect (2245): c + alpha (c - p), clamped when bounded
        double *wrow = S + worst * static_cast<uint32_t>(n);
        const double w0 = in0 ? wrow[e0] : 0.0, w1 = in1 ? wrow[e1] : 0.0;
        double r0 = c0 + p.alpha * (c0 - w0), r1 = c1 + p.alpha * (c1 - w1);
        if (p.bounded) clamp2(r0, r1);
        const double rs = score(r0, r1);
        fcalls++;
        shrunk = 0;
        const double sb = scores[best], ss = scores[second], sw = scores[worst];
        // 0 accept reflection, 1 expand, 2 contract
        const int action = (rs >= sb && rs < ss) ? 0 : (rs < sb ? 1 : 2);
        lap(2);
        if (action == 0) {  // 2251-2253
          if (in0) wrow[e0] = r0;
          if (in1) wrow[e1] = r1;
          if (lane == 0) scores[worst] = rs;
        } else if (action == 1) {  // expand, 2255-2265: c + gamma (reflected - c)
          double x0e = c0 + p.gamma * (r0 - c0), x1e = c1 + p.gamma * (r1 - c1);
          if (p.bounded) clamp2(x0e, x1e);
          const double es = score(x0e, x1e);
          fcalls++;
          const bool take_exp = es < rs;
          if (in0) wrow[e0] = take_exp ? x0e : r0;
          if (in1) wrow[e1] = take_exp ? x1e : r1;
          if (lane == 0) scores[worst] = take_exp ? es : rs;
          lap(3);
        } else {  // contraction, 2266-2297 (B4: the reflect transform for both kinds)
          const bool outside = rs < sw;
          double x0c = c0 + p.rho * (c0 - (outside ? r0 : w0)), x1c = c1 + p.rho * (c1 - (outside ? r1 : w1));
          if (p.bounded) clamp2(x0c, x1c);
          const double cs = score(x0c, x1c);
          fcalls++;
          if (cs < (outside ? rs : sw)) {
            if (in0) wrow[e0] = x0c;
            if (in1) wrow[e1] = x1c;
            if (lane == 0) scores[worst] = cs;
            lap(3);
          } else {  // shrink (2009-2035) and rescoring (2288-2294): every wave takes its rows
            lap(3);
            if (lane == 0) {
              ctl->best = best;
              ctl->cmd = kNmCmdShrink;
            }
            __syncthreads();  // (A)
            if constexpr (SEQ)
              nm_shrink_rows_seq<OBJ>(S, scores, n, nv, best, p.sigma, p.fmul, 0, nwaves);
            else
              nm_shrink_rows<OBJ>(S, scores, n, nv, best, p.sigma, p.fmul, 0, nwaves);
            __syncthreads();  // (B)
            fcalls += nv - 1;
            shrunk = 1;
            if (p.phase && lane == 0) ph[7] += 1;
            lap(4);
          }
        }
      }
      if (lane == 0) {
        ctl->best = best;
        ctl->iter = iter;
        ctl->fcalls = fcalls;
        ctl->eps = eps;
        ctl->cmd = kNmCmdEnd;
      }
      if (p.phase && lane == 0) ph[6] += iter;
      __syncthreads();  // (A) with the end request
    }
    __syncthreads();  // ctl as the driver left it, for every wave
    // x = current_simplex.vals[best] (2235); restarts continue from it (2129-2132)
    const uint64_t best = ctl->best;
    for (uint64_t j = t; j < n; j += nthreads) x0[j] = S[best * n + j];
    total_iter += ctl->iter;
    final_f = scores[best];
    __syncthreads();
  }
  for (uint64_t j = t; j < n; j += nthreads) p.x[pid * n + j] = x0[j];
  if (t == 0) {
    NmProblem *pr = p.prob + pid;
    pr->f = final_f;
    pr->eps = ctl->eps;
    pr->iter = total_iter;
    pr->fcalls = ctl->fcalls;
    if (p.phase)
      for (int k = 0; k < kNmPhases; k++) p.phase[pid * kNmPhases + k] = ph[k];
  }
}

// chunks of 128 coordinates per point: 1 = simplex in LDS, else in the global workspace

}  // namespace nlsg
