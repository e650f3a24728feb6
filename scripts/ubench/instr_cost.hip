// scripts/ubench/instr_cost.hip — issue cost of single gfx950 vector instructions, in cycles per
// wave and SIMD (measurement aid, not product code): every kernel runs ITER x 16 copies of one
// instruction on four independent register chains, 8 waves per SIMD, so that latency is covered
// and only the issue rate shows.
#include <hip/hip_runtime.h>

#include <cstdio>

constexpr int ITER = 2048;

#define REP4(S) S S S S
#define BODY16(I0, I1, I2, I3) REP4(I0 I1 I2 I3)

#define KERNEL(NAME, I0, I1, I2, I3)                                                      \
  __global__ __launch_bounds__(256) void NAME(double *out) {                               \
    const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;                                \
    double a = 1.0 + tid * 1e-9, b = 1.0 - tid * 1e-9, c = 0.5 + tid * 1e-9, d = 0.25;     \
    double k = 1.0000001;                                                                  \
    unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;                \
    for (int i = 0; i < ITER; i++)                                                         \
      asm volatile(BODY16(I0, I1, I2, I3)                                                  \
                   : "+v"(a), "+v"(b), "+v"(c), "+v"(d), "+v"(ia), "+v"(ib), "+v"(ic),      \
                     "+v"(id)                                                              \
                   : "v"(k)                                                                \
                   : "vcc");                                                               \
    out[tid] = a + b + c + d + ia + ib + ic + id;                                          \
  }

// operands: %0..%3 doubles a..d, %4..%7 uints, %8 double k
KERNEL(k_fma64, "v_fma_f64 %0, %0, %8, %8\n", "v_fma_f64 %1, %1, %8, %8\n",
       "v_fma_f64 %2, %2, %8, %8\n", "v_fma_f64 %3, %3, %8, %8\n")
KERNEL(k_add64, "v_add_f64 %0, %0, %8\n", "v_add_f64 %1, %1, %8\n", "v_add_f64 %2, %2, %8\n",
       "v_add_f64 %3, %3, %8\n")
KERNEL(k_mul64, "v_mul_f64 %0, %0, %8\n", "v_mul_f64 %1, %1, %8\n", "v_mul_f64 %2, %2, %8\n",
       "v_mul_f64 %3, %3, %8\n")
KERNEL(k_mov32, "v_mov_b32 %4, %5\n", "v_mov_b32 %5, %6\n", "v_mov_b32 %6, %7\n",
       "v_mov_b32 %7, %4\n")
KERNEL(k_mov64, "v_mov_b64 %0, %1\n", "v_mov_b64 %1, %2\n", "v_mov_b64 %2, %3\n",
       "v_mov_b64 %3, %0\n")
KERNEL(k_xor32, "v_xor_b32 %4, %4, %5\n", "v_xor_b32 %5, %5, %6\n", "v_xor_b32 %6, %6, %7\n",
       "v_xor_b32 %7, %7, %4\n")
KERNEL(k_add32, "v_add_u32 %4, %4, %5\n", "v_add_u32 %5, %5, %6\n", "v_add_u32 %6, %6, %7\n",
       "v_add_u32 %7, %7, %4\n")
KERNEL(k_cndmask, "v_cndmask_b32 %4, %4, %5, vcc\n", "v_cndmask_b32 %5, %5, %6, vcc\n",
       "v_cndmask_b32 %6, %6, %7, vcc\n", "v_cndmask_b32 %7, %7, %4, vcc\n")
KERNEL(k_mullo, "v_mul_lo_u32 %4, %4, %5\n", "v_mul_lo_u32 %5, %5, %6\n",
       "v_mul_lo_u32 %6, %6, %7\n", "v_mul_lo_u32 %7, %7, %4\n")
KERNEL(k_mulhi, "v_mul_hi_u32 %4, %4, %5\n", "v_mul_hi_u32 %5, %5, %6\n",
       "v_mul_hi_u32 %6, %6, %7\n", "v_mul_hi_u32 %7, %7, %4\n")
KERNEL(k_mad64, "v_mad_u64_u32 %0, vcc, %4, %5, %0\n", "v_mad_u64_u32 %1, vcc, %5, %6, %1\n",
       "v_mad_u64_u32 %2, vcc, %6, %7, %2\n", "v_mad_u64_u32 %3, vcc, %7, %4, %3\n")
KERNEL(k_mul24, "v_mul_u32_u24 %4, %4, %5\n", "v_mul_u32_u24 %5, %5, %6\n",
       "v_mul_u32_u24 %6, %6, %7\n", "v_mul_u32_u24 %7, %7, %4\n")
KERNEL(k_ldexp, "v_ldexp_f64 %0, %0, 1\n", "v_ldexp_f64 %1, %1, 1\n", "v_ldexp_f64 %2, %2, 1\n",
       "v_ldexp_f64 %3, %3, 1\n")
KERNEL(k_cvt_f64_u32, "v_cvt_f64_u32 %0, %4\n", "v_cvt_f64_u32 %1, %5\n",
       "v_cvt_f64_u32 %2, %6\n", "v_cvt_f64_u32 %3, %7\n")
KERNEL(k_cvt_i32_f64, "v_cvt_i32_f64 %4, %0\n", "v_cvt_i32_f64 %5, %1\n",
       "v_cvt_i32_f64 %6, %2\n", "v_cvt_i32_f64 %7, %3\n")
KERNEL(k_rcp64, "v_rcp_f64 %0, %0\n", "v_rcp_f64 %1, %1\n", "v_rcp_f64 %2, %2\n",
       "v_rcp_f64 %3, %3\n")
KERNEL(k_rsq64, "v_rsq_f64 %0, %0\n", "v_rsq_f64 %1, %1\n", "v_rsq_f64 %2, %2\n",
       "v_rsq_f64 %3, %3\n")
KERNEL(k_sqrt64, "v_sqrt_f64 %0, %0\n", "v_sqrt_f64 %1, %1\n", "v_sqrt_f64 %2, %2\n",
       "v_sqrt_f64 %3, %3\n")
KERNEL(k_floor64, "v_floor_f64 %0, %0\n", "v_floor_f64 %1, %1\n", "v_floor_f64 %2, %2\n",
       "v_floor_f64 %3, %3\n")
KERNEL(k_frexp_mant, "v_frexp_mant_f64 %0, %0\n", "v_frexp_mant_f64 %1, %1\n",
       "v_frexp_mant_f64 %2, %2\n", "v_frexp_mant_f64 %3, %3\n")
KERNEL(k_frexp_exp, "v_frexp_exp_i32_f64 %4, %0\n", "v_frexp_exp_i32_f64 %5, %1\n",
       "v_frexp_exp_i32_f64 %6, %2\n", "v_frexp_exp_i32_f64 %7, %3\n")
KERNEL(k_div_scale, "v_div_scale_f64 %0, vcc, %0, %8, %0\n", "v_div_scale_f64 %1, vcc, %1, %8, %1\n",
       "v_div_scale_f64 %2, vcc, %2, %8, %2\n", "v_div_scale_f64 %3, vcc, %3, %8, %3\n")
KERNEL(k_div_fmas, "v_div_fmas_f64 %0, %0, %8, %8\n", "v_div_fmas_f64 %1, %1, %8, %8\n",
       "v_div_fmas_f64 %2, %2, %8, %8\n", "v_div_fmas_f64 %3, %3, %8, %8\n")
KERNEL(k_div_fixup, "v_div_fixup_f64 %0, %0, %8, %8\n", "v_div_fixup_f64 %1, %1, %8, %8\n",
       "v_div_fixup_f64 %2, %2, %8, %8\n", "v_div_fixup_f64 %3, %3, %8, %8\n")
KERNEL(k_lshr64, "v_lshrrev_b64 %0, 3, %0\n", "v_lshrrev_b64 %1, 3, %1\n",
       "v_lshrrev_b64 %2, 3, %2\n", "v_lshrrev_b64 %3, 3, %3\n")
KERNEL(k_lshl_add64, "v_lshl_add_u64 %0, %0, 0, %1\n", "v_lshl_add_u64 %1, %1, 0, %2\n",
       "v_lshl_add_u64 %2, %2, 0, %3\n", "v_lshl_add_u64 %3, %3, 0, %0\n")
KERNEL(k_cmp64, "v_cmp_lt_f64 vcc, %0, %1\n", "v_cmp_lt_f64 vcc, %1, %2\n",
       "v_cmp_lt_f64 vcc, %2, %3\n", "v_cmp_lt_f64 vcc, %3, %0\n")
KERNEL(k_pk_fma32, "v_pk_fma_f32 %0, %0, %8, %8\n", "v_pk_fma_f32 %1, %1, %8, %8\n",
       "v_pk_fma_f32 %2, %2, %8, %8\n", "v_pk_fma_f32 %3, %3, %8, %8\n")
KERNEL(k_mov_dpp, "v_mov_b32_dpp %4, %5 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
       "v_mov_b32_dpp %5, %6 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
       "v_mov_b32_dpp %6, %7 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n",
       "v_mov_b32_dpp %7, %4 quad_perm:[1,0,3,2] row_mask:0xf bank_mask:0xf\n")

// v_cndmask variants: vcc set by a compare first; an SGPR pair as the mask; interleaved with fma
__global__ __launch_bounds__(256) void k_cnd_vccset(double *out) {
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" : : "v"(ia), "v"(ib) : "vcc");
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %1, %1, %2, vcc\n"
                      "v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : );
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_cnd_sgpr(double *out) {
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  uint64_t m;
  asm volatile("v_cmp_lt_u32 %0, %1, %2\n" : "=s"(m) : "v"(ia), "v"(ib));
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cndmask_b32 %0, %0, %1, %4\n v_cndmask_b32 %1, %1, %2, %4\n"
                      "v_cndmask_b32 %2, %2, %3, %4\n v_cndmask_b32 %3, %3, %0, %4\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : "s"(m));
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_cnd_cmp(double *out) {  // compare + select pairs
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                      "v_cmp_lt_u32 vcc, %2, %3\n v_cndmask_b32 %2, %2, %3, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_cnd_indep(double *out) {  // no dependent chain at all
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3, e, f, g, h;
  asm volatile("v_cmp_lt_u32 vcc, %0, %1\n" : : "v"(ia), "v"(ib) : "vcc");
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cndmask_b32 %4, %0, %1, vcc\n v_cndmask_b32 %5, %1, %2, vcc\n"
                      "v_cndmask_b32 %6, %2, %3, vcc\n v_cndmask_b32 %7, %3, %0, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id), "=v"(e), "=v"(f), "=v"(g), "=v"(h));
  out[tid] = ia + ib + ic + id + e + f + g + h;
}

__global__ __launch_bounds__(256) void k_cmp_cnd2(double *out) {  // compare + a 64-bit select
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                      "v_cndmask_b32 %2, %2, %3, vcc\n v_xor_b32 %1, %1, %3\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_cmp_cnd3(double *out) {  // compare + three selects
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n"
                      "v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_pat_ccxc(double *out) {  // cmp, 2 selects, xor, third select, 3 xor (8 instrs)
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_xor_b32 %1, %1, %3\n v_cndmask_b32 %3, %3, %1, vcc\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_pat_cxcxc(double *out) {  // cmp, select, xor, select, xor, select, 2 xor (8 instrs)
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_xor_b32 %1, %1, %3\n v_cndmask_b32 %2, %2, %3, vcc\n v_xor_b32 %1, %1, %3\n v_cndmask_b32 %3, %3, %1, vcc\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_pat_e64(double *out) {  // cmp + three VOP3-encoded selects on vcc (4 instrs)
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32_e64 %0, %0, %1, vcc\n v_cndmask_b32_e64 %2, %2, %3, vcc\n v_cndmask_b32_e64 %1, %1, %3, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_pat_c4(double *out) {  // cmp + four selects + 3 xor (8 instrs)
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void k_pat_fma_between(double *out) {  // cmp, 2 selects, 4 mul_lo, third select (8 instrs)
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_mul_lo_u32 %1, %1, %3\n v_mul_lo_u32 %1, %1, %3\n v_mul_lo_u32 %1, %1, %3\n v_mul_lo_u32 %1, %1, %3\n v_cndmask_b32 %3, %3, %1, vcc\n")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Ccc_Ccc(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Cncc(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Cccc_x(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_xor_b32 %1, %1, %3\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Cccc_xx(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_xor_b32 %1, %1, %3\n v_xor_b32 %1, %1, %3\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Cnccc(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n s_nop 1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Cccc_n(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n s_nop 1\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ __launch_bounds__(256) void g4_Ccccc(double *out) {  // pattern
  const uint64_t tid = blockIdx.x * 256ull + threadIdx.x;
  unsigned ia = tid, ib = tid * 3 + 1, ic = tid * 5 + 2, id = tid * 7 + 3;
  for (int i = 0; i < ITER; i++)
    asm volatile(REP4("v_cmp_lt_u32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc\n v_cndmask_b32 %2, %2, %3, vcc\n v_cndmask_b32 %1, %1, %3, vcc\n v_cndmask_b32 %3, %3, %0, vcc\n ")
                 : "+v"(ia), "+v"(ib), "+v"(ic), "+v"(id) : : "vcc");
  out[tid] = ia + ib + ic + id;
}
__global__ void k_clock(unsigned long long *out) {
  const unsigned long long w0 = wall_clock64(), c0 = clock64();
  double a = threadIdx.x;
  for (int i = 0; i < 200000; i++) asm volatile("v_fma_f64 %0, %0, %0, %0\n" : "+v"(a));
  const unsigned long long w1 = wall_clock64(), c1 = clock64();
  if (threadIdx.x == 0) out[0] = w1 - w0, out[1] = c1 - c0, out[2] = (unsigned long long)a;
}

template <typename K>
void run(K kern, double *out, const char *name, int waves_per_simd) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  const int blocks = 256 * waves_per_simd;
  kern<<<blocks, 256>>>(out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kern<<<blocks, 256>>>(out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::printf("%-16s %d waves/SIMD  %8.3f ms  %6.2f cycles per wave-instruction per SIMD at 2.4 GHz\n",
              name, waves_per_simd, ms, ms * 1e-3 * 2.4e9 / (ITER * 16.0 * waves_per_simd));
}

template <typename K>
void run_group(K kern, double *out, const char *name) {
  hipEvent_t a, b;
  hipEventCreate(&a);
  hipEventCreate(&b);
  kern<<<2048, 256>>>(out);
  hipDeviceSynchronize();
  hipEventRecord(a);
  kern<<<2048, 256>>>(out);
  hipEventRecord(b);
  hipEventSynchronize(b);
  float ms = 0;
  hipEventElapsedTime(&ms, a, b);
  std::printf("%-16s %8.3f ms  %6.2f cycles per pattern group (C = v_cmp, c = VOP2 v_cndmask on vcc, x = v_xor, n = s_nop 1)\n",
              name, ms, ms * 1e-3 * 2.435e9 / (ITER * 4.0 * 8));
}
#define RUNG(K) run_group(K, out, #K);
#define RUN(K) run(K, out, #K, 8); run(K, out, #K, 2);

int main() {
  double *out;
  hipMalloc(&out, 256 * 8 * 256 * sizeof(double));
  RUN(k_fma64) RUN(k_add64) RUN(k_mul64) RUN(k_mov32) RUN(k_mov64) RUN(k_xor32) RUN(k_add32)
  RUN(k_cndmask) RUN(k_mullo) RUN(k_mulhi) RUN(k_mad64) RUN(k_mul24) RUN(k_ldexp)
  RUN(k_cvt_f64_u32) RUN(k_cvt_i32_f64) RUN(k_rcp64) RUN(k_rsq64) RUN(k_sqrt64) RUN(k_floor64)
  RUN(k_frexp_mant) RUN(k_frexp_exp) RUN(k_div_scale) RUN(k_div_fmas) RUN(k_div_fixup)
  RUNG(g4_Ccc_Ccc) RUNG(g4_Cncc) RUNG(g4_Cccc_x) RUNG(g4_Cccc_xx) RUNG(g4_Cnccc) RUNG(g4_Cccc_n) RUNG(g4_Ccccc)
  RUN(k_pat_ccxc) RUN(k_pat_cxcxc) RUN(k_pat_e64) RUN(k_pat_c4) RUN(k_pat_fma_between) RUN(k_cmp_cnd2) RUN(k_cmp_cnd3) RUN(k_cnd_vccset) RUN(k_cnd_sgpr) RUN(k_cnd_cmp) RUN(k_cnd_indep)
  RUN(k_lshr64) RUN(k_lshl_add64) RUN(k_cmp64) RUN(k_pk_fma32) RUN(k_mov_dpp)
  unsigned long long *ck, h[3];
  hipMalloc(&ck, 24);
  k_clock<<<1, 64>>>(ck);
  hipMemcpy(h, ck, 24, hipMemcpyDeviceToHost);
  int wc = 0;
  hipDeviceGetAttribute(&wc, hipDeviceAttributeWallClockRate, 0);
  std::printf("clock64 / wall_clock64 = %.3f; wall clock rate %d kHz => shader clock %.3f GHz (one wave)\n",
              double(h[1]) / double(h[0]), wc, double(h[1]) / double(h[0]) * wc * 1e-6);
  hipFree(out);
  return 0;
}
