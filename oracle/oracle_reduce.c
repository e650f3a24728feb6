/* oracle/oracle_reduce.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE. */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

/* std_err, nlsolver.h:2037-2052, literally (two serial passes, pow(.,2)). */
double orc_std_err_serial(const double *x, size_t n) {
  size_t i = 0;
  double mean_val = 0, result = 0;
  for (; i < n; i++) mean_val += x[i];
  mean_val /= (double)i;
  i = 0;
  for (; i < n; i++) result += pow(x[i] - mean_val, 2);
  result /= (double)(i - 1);
  return sqrt(result);
}

/* The device's 256-thread block tree: thread t adds v[t], v[t+256], ... in
 * that order; 64-lane xor butterfly inside each of the 4 waves; the 4 wave
 * sums are added left to right. */
static double block_tree(const double *v, size_t n, int square_dev, double mean) {
  double th[256];
  for (int t = 0; t < 256; t++) {
    double acc = 0.0;
    for (size_t i = (size_t)t; i < n; i += 256) {
      if (square_dev) {
        const double d = v[i] - mean;
        acc += d * d;
      } else {
        acc += v[i];
      }
    }
    th[t] = acc;
  }
  double w[4];
  for (int wv = 0; wv < 4; wv++) {
    double lane[64], tmp[64];
    memcpy(lane, th + 64 * wv, sizeof lane);
    for (int off = 32; off >= 1; off >>= 1) {
      for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
      memcpy(lane, tmp, sizeof lane);
    }
    w[wv] = lane[0];
  }
  return ((w[0] + w[1]) + w[2]) + w[3];
}

double orc_block_tree_sum(const double *v, size_t n) { return block_tree(v, n, 0, 0.0); }

static double tiled(const double *v, size_t n, int square_dev, double mean) {
  const size_t T = 1024;
  const size_t nt = (n + T - 1) / T;
  double *part = (double *)malloc((nt ? nt : 1) * sizeof(double));
  for (size_t j = 0; j < nt; j++) {
    const size_t len = (n - j * T) < T ? (n - j * T) : T;
    part[j] = block_tree(v + j * T, len, square_dev, mean);
  }
  const double r = block_tree(part, nt, 0, 0.0);
  free(part);
  return r;
}

double orc_tiled_sum(const double *v, size_t n) { return tiled(v, n, 0, 0.0); }
double orc_tiled_sumsq_dev(const double *v, size_t n, double mean) {
  return tiled(v, n, 1, mean);
}

/* The DE engine's one-pass form: every tile of 1024 scores yields (sum_t, M2_t about the
 * tile's own mean), both with the block tree; tiles are merged like shards are (Chan et al.):
 *   total = tree_t(sum_t), mean = total / n,
 *   M2 = tree_t( M2_t + n_t * (mean_t - mean)^2 ),  mean_t = sum_t / n_t.
 * One tile (n <= 1024): identical to orc_tiled_sumsq_dev about the mean. *sum_out = total. */
double orc_tiled_m2_merged(const double *v, size_t n, double *sum_out) {
  const size_t T = 1024;
  const size_t nt = (n + T - 1) / T;
  double *sum = (double *)calloc(nt ? nt : 1, sizeof(double));
  double *term = (double *)malloc((nt ? nt : 1) * sizeof(double));
  for (size_t j = 0; j < nt; j++) {
    const size_t len = (n - j * T) < T ? (n - j * T) : T;
    sum[j] = block_tree(v + j * T, len, 0, 0.0);
    term[j] = block_tree(v + j * T, len, 1, sum[j] / (double)len);
  }
  const double total = block_tree(sum, nt, 0, 0.0);
  const double mean = total / (double)n;
  for (size_t j = 0; j < nt; j++) {
    const size_t len = (n - j * T) < T ? (n - j * T) : T;
    const double dm = sum[j] / (double)len - mean;
    term[j] = term[j] + (double)len * (dm * dm);
  }
  const double m2 = block_tree(term, nt, 0, 0.0);
  free(sum);
  free(term);
  if (sum_out) *sum_out = total;
  return m2;
}

/* std_err with the device tree (same formula as nlsolver.h:2037-2052 but
 * d*d instead of pow(d,2) and tree summation). */
double orc_std_err_tree(const double *x, size_t n) {
  const double mean = orc_tiled_sum(x, n) / (double)n;
  const double ss = orc_tiled_sumsq_dev(x, n, mean);
  return sqrt(ss / (double)(n - 1));
}
