// Issue cost of the instructions k_taubin_eigen is made of, on one SIMD: cycles per instruction for a dependent chain
// and for four independent chains, with one and with two waves resident on the SIMD.
//   hipcc --offload-arch=gfx950 -O2 -o /tmp/valu_rates scripts/micro/valu_rates.hip && /tmp/valu_rates
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define REP16(x) x x x x x x x x x x x x x x x x
constexpr int kIters = 2000;  // x 16 x 4 instructions per test

template <int KIND, int CHAINS>
__global__ __launch_bounds__(64) void k(long long* out, double seed)
{
  double a0 = seed + threadIdx.x, a1 = a0 + 1.0, a2 = a0 + 2.0, a3 = a0 + 3.0, b = 1.0000001;
  int i0 = threadIdx.x, i1 = i0 + 1, i2 = i0 + 2, i3 = i0 + 3, idx = ((threadIdx.x + 1) & 63) << 2;
  const long long t0 = wall_clock64();
  asm volatile("s_mov_b32 s10, 0x55555555\n s_mov_b32 s11, 0x55555555" ::: "s10", "s11", "s12");
  for (int it = 0; it < kIters; it++)
  {
#define OP4(INS0, INS1, INS2, INS3) \
  if (CHAINS == 1) { REP16(asm volatile(INS0 : "+v"(a0) : "v"(b)); asm volatile(INS0 : "+v"(a0) : "v"(b)); asm volatile(INS0 : "+v"(a0) : "v"(b)); asm volatile(INS0 : "+v"(a0) : "v"(b));) } \
  else { REP16(asm volatile(INS0 : "+v"(a0) : "v"(b)); asm volatile(INS1 : "+v"(a1) : "v"(b)); asm volatile(INS2 : "+v"(a2) : "v"(b)); asm volatile(INS3 : "+v"(a3) : "v"(b));) }
#define OPI4(INS) \
  if (CHAINS == 1) { REP16(asm volatile(INS : "+v"(i0) : "v"(idx)); asm volatile(INS : "+v"(i0) : "v"(idx)); asm volatile(INS : "+v"(i0) : "v"(idx)); asm volatile(INS : "+v"(i0) : "v"(idx));) } \
  else { REP16(asm volatile(INS : "+v"(i0) : "v"(idx)); asm volatile(INS : "+v"(i1) : "v"(idx)); asm volatile(INS : "+v"(i2) : "v"(idx)); asm volatile(INS : "+v"(i3) : "v"(idx));) }
    if (KIND == 0) { OP4("v_mul_f64 %0, %0, %1", "v_mul_f64 %0, %0, %1", "v_mul_f64 %0, %0, %1", "v_mul_f64 %0, %0, %1") }
    if (KIND == 1) { OP4("v_add_f64 %0, %0, %1", "v_add_f64 %0, %0, %1", "v_add_f64 %0, %0, %1", "v_add_f64 %0, %0, %1") }
    if (KIND == 2) { OP4("v_fma_f64 %0, %0, %1, %1", "v_fma_f64 %0, %0, %1, %1", "v_fma_f64 %0, %0, %1, %1", "v_fma_f64 %0, %0, %1, %1") }
    if (KIND == 3) { OP4("v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf", "v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf", "v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf", "v_mov_b64_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf") }
    if (KIND == 4) { OPI4("v_mov_b32_dpp %0, %0 row_newbcast:3 row_mask:0xf bank_mask:0xf") }
    if (KIND == 5) { OPI4("v_cndmask_b32 %0, %0, %1, vcc") }
    if (KIND == 6) { OPI4("ds_bpermute_b32 %0, %1, %0\n s_waitcnt lgkmcnt(0)") }
    if (KIND == 7) { OP4("v_rsq_f64 %0, %0", "v_rsq_f64 %0, %0", "v_rsq_f64 %0, %0", "v_rsq_f64 %0, %0") }
    if (KIND == 8) { OP4("v_rcp_f64 %0, %0", "v_rcp_f64 %0, %0", "v_rcp_f64 %0, %0", "v_rcp_f64 %0, %0") }
    if (KIND == 9) { OPI4("v_add_u32 %0, %0, %1") }
    if (KIND == 10) { OP4("v_max_f64 %0, %0, %1", "v_max_f64 %0, %0, %1", "v_max_f64 %0, %0, %1", "v_max_f64 %0, %0, %1") }
    if (KIND == 11) { OPI4("v_mul_f32 %0, %0, %1") }
    if (KIND == 12) { OP4("v_cmp_lt_f64 vcc, %0, %1", "v_cmp_lt_f64 vcc, %0, %1", "v_cmp_lt_f64 vcc, %0, %1", "v_cmp_lt_f64 vcc, %0, %1") }
    if (KIND == 14) { OPI4("v_cndmask_b32_e64 %0, %0, %1, s[10:11]") }
    if (KIND == 15) { OPI4("v_bfi_b32 %0, %1, %0, %1") }
    if (KIND == 16) { OPI4("v_and_b32 %0, %0, %1") }
    if (KIND == 17) { OPI4("v_mov_b32 %0, %1") }
    if (KIND == 18) { OPI4("v_cndmask_b32_e64 %0, 0, %1, s[10:11]") }
    if (KIND == 19) { OP4("v_mov_b64 %0, %1", "v_mov_b64 %0, %1", "v_mov_b64 %0, %1", "v_mov_b64 %0, %1") }
    if (KIND == 20) { OPI4("v_readlane_b32 s12, %0, 3\n v_add_u32 %0, %0, %1") }
    if (KIND == 21) { OPI4("v_cmp_lt_i32 vcc, %0, %1\n v_cndmask_b32 %0, %0, %1, vcc") }
    if (KIND == 22) { OPI4("v_cmp_lt_i32 s[10:11], %0, %1\n s_nop 1\n v_cndmask_b32_e64 %0, %0, %1, s[10:11]") }
    if (KIND == 23) { OPI4("v_xor_b32 %0, %0, %1") }
    if (KIND == 13) { OP4("v_ldexp_f64 %0, %0, 1", "v_ldexp_f64 %0, %0, 1", "v_ldexp_f64 %0, %0, 1", "v_ldexp_f64 %0, %0, 1") }
  }
  const long long t1 = wall_clock64();
  if (threadIdx.x == 0)
    out[blockIdx.x] = t1 - t0;
  if (a0 + a1 + a2 + a3 + i0 + i1 + i2 + i3 == 12345.678)
    out[0] = 0;
}

template <int KIND, int CHAINS>
static void run(const char* name, long long* d_out)
{
  // blocks: 1 wave each.  1024 blocks = one wave per SIMD (256 CUs x 4); 2048 = two per SIMD
  for (int waves = 1; waves <= 2; waves++)
  {
    const int blocks = 1024 * waves;
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(64), 0, 0, d_out, 1.0);
    hipLaunchKernelGGL((k<KIND, CHAINS>), dim3(blocks), dim3(64), 0, 0, d_out, 1.0);
    hipDeviceSynchronize();
    std::vector<long long> h(blocks);
    hipMemcpy(h.data(), d_out, sizeof(long long) * blocks, hipMemcpyDeviceToHost);
    double mean = 0;
    for (long long v : h)
      mean += (double) v;
    mean /= blocks;
    const double n = (double) kIters * 64.0;
    // wall_clock64 ticks at 100 MHz; report ns per instruction of one wave, and SIMD-cycles at 2.4 GHz per issued instruction
    std::printf("%-22s chains %d waves/SIMD %d: %.2f ns per instr per wave = %.1f cycles @2.4GHz ; per SIMD issue %.1f cycles\n", name,
      CHAINS, waves, mean * 10.0 / n, mean * 10.0 / n * 2.4, mean * 10.0 / n * 2.4 / waves);
  }
}

int main()
{
  long long* d_out;
  hipMalloc((void**) &d_out, sizeof(long long) * 4096);
#define RUN(K, NAME) run<K, 1>(NAME, d_out); run<K, 4>(NAME, d_out);
  RUN(0, "v_mul_f64") RUN(1, "v_add_f64") RUN(2, "v_fma_f64") RUN(3, "v_mov_b64_dpp bcast") RUN(4, "v_mov_b32_dpp bcast")
  RUN(5, "v_cndmask_b32") RUN(6, "ds_bpermute+wait") RUN(7, "v_rsq_f64") RUN(8, "v_rcp_f64") RUN(9, "v_add_u32") RUN(10, "v_max_f64")
  RUN(11, "v_mul_f32") RUN(12, "v_cmp_lt_f64") RUN(13, "v_ldexp_f64")
  RUN(14, "v_cndmask_e64 sgpr") RUN(18, "v_cndmask_e64 0,v,sgpr") RUN(15, "v_bfi_b32") RUN(16, "v_and_b32") RUN(17, "v_mov_b32") RUN(19, "v_mov_b64")
  RUN(20, "readlane+add (2 instr)") RUN(21, "cmp+cndmask vcc (2)") RUN(22, "cmp+nop+cndmask sgpr (3)") RUN(23, "v_xor_b32")
  return 0;
}
