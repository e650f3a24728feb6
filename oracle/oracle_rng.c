/* oracle/oracle_rng.c — TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h). */
#include "oracle.h"

#define ORC_GOLDEN 0x9E3779B97F4A7C15ull

/* splitmix64 finaliser; constants from nlsolver.h:1275-1277. */
uint64_t orc_mix64(uint64_t z) {
  z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
  z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
  return z ^ (z >> 31);
}

/* rng::splitmix::yield_init, nlsolver.h:1273-1278: s += golden; mix(s). */
uint64_t orc_splitmix_next(uint64_t *state) {
  *state += ORC_GOLDEN;
  return orc_mix64(*state);
}

/* rng::xorshift ctor, nlsolver.h:1345-1349: x0 = splitmix.yield_init();
 * x1 = x0 >> 32. */
void orc_xorshift_init(orc_xorshift *g) {
  uint64_t s = ORC_SPLITMIX_SEED;
  g->x[0] = orc_splitmix_next(&s);
  g->x[1] = g->x[0] >> 32;
}

/* rng::xorshift::yield, nlsolver.h:1350-1361. (double)UINT64_MAX == 2^64, so
 * the division is an exact scaling of (double)(t+s). */
double orc_xorshift_next(orc_xorshift *g) {
  uint64_t t = g->x[0];
  const uint64_t s = g->x[1];
  g->x[0] = s;
  t ^= t << 23;
  t ^= t >> 18;
  t ^= s ^ (s >> 5);
  g->x[1] = t;
  return (double)(t + s) / (double)18446744073709551615ull;
}

/* Counter RNG: child key / draw number `index` under `parent` = the
 * (index+1)-th output of a splitmix64 stream whose state starts at `parent`. */
uint64_t orc_ctr_key(uint64_t parent, uint64_t index) {
  return orc_mix64(parent + ORC_GOLDEN * (index + 1));
}

double orc_u01(uint64_t bits) { return (double)bits * 0x1p-64; }
