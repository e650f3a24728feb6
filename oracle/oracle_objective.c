/* oracle/oracle_objective.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE. */
#include <math.h>
#include <string.h>

#include "oracle.h"

/* Number of additive terms and the per-element term of each built-in
 * objective. Term i may read x[i] and x[i+1] (chain objectives). */
static size_t n_terms(int obj, size_t D) {
  return obj == ORC_OBJ_ROSENBROCK ? (D ? D - 1 : 0) : D;
}

/* det != 0: the device's arithmetic (deterministic cosine, nlsg_math.h) — the tree / order-1
 * restatements the kernels match bit for bit; det == 0: libm, as the reference — the serial
 * restatements pinned to the goldens. */
static double term(int obj, const double *x, size_t i, int det) {
  switch (obj) {
    case ORC_OBJ_ROSENBROCK: {
      /* example.cpp:43-47: t1*t1 + 100*t2*t2 with t1 = 1-x0, t2 = x1-x0*x0 */
      const double t1 = 1 - x[i];
      const double t2 = (x[i + 1] - x[i] * x[i]);
      return t1 * t1 + 100 * t2 * t2;
    }
    case ORC_OBJ_SPHERE: /* test_functions.h:56 */
      return x[i] * x[i];
    case ORC_OBJ_STYBLINSKI_TANG: { /* test_functions.h:255-257 */
      const double x2 = x[i] * x[i];
      return x2 * x2 - 16 * x2 + 5 * x[i];
    }
    case ORC_OBJ_RASTRIGIN: /* test_functions.h:74-76 */
      return x[i] * x[i] - 10 * (det ? orc_cos_2pi(x[i]) : cos(2 * M_PI * x[i]));
    default:
      return NAN;
  }
}

static double finish(int obj, double sum, size_t D) {
  switch (obj) {
    case ORC_OBJ_STYBLINSKI_TANG:
      return sum / 2.0; /* test_functions.h:258 */
    case ORC_OBJ_RASTRIGIN:
      return 10.0 * (double)D + sum; /* test_functions.h:74 (2*10 + ...) */
    default:
      return sum;
  }
}

double orc_objective_seq(int obj, const double *x, size_t D) {
  double acc = 0.0;
  const size_t n = n_terms(obj, D);
  for (size_t i = 0; i < n; i++) acc += term(obj, x, i, 0);
  return finish(obj, acc, D);
}

double orc_objective_tree(int obj, const double *x, size_t D) {
  double lane[64], tmp[64];
  memset(lane, 0, sizeof lane);
  const size_t n = n_terms(obj, D);
  for (size_t e = 0; e < n; e++) lane[(e % 128) / 2] += term(obj, x, e, 1);
  for (int off = 32; off >= 1; off >>= 1) {
    for (int l = 0; l < 64; l++) tmp[l] = lane[l] + lane[l ^ off];
    memcpy(lane, tmp, sizeof lane);
  }
  return finish(obj, lane[0], D);
}
