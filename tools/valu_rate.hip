// valu_rate.hip -- what one gfx950 SIMD issues per clock, by instruction class and waves per SIMD.
//
// The control-step kernels are VALU-issue bound: their roof is "instructions x cycles per
// instruction", so the cycles have to be MEASURED for the instruction mix they use (plain / packed / DPP fp32,
// transcendentals, fp64, cvt) at the occupancies they run at (1, 2, 4, 8 waves per SIMD).
//
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -fno-slp-vectorize tools/valu_rate.hip -o tools/diag/valu_rate
//   tools/diag/valu_rate           -> table: op, waves/SIMD, cycles per wave-instruction per SIMD (s_memtime)
//
// One 256*W-thread workgroup per CU (forced by a > 80 KiB LDS allocation): W waves on each of the 4 SIMDs.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdio>
#include <vector>

typedef float float2_ __attribute__((ext_vector_type(2)));

constexpr int kIters = 512;
constexpr int kUnroll = 16;  // independent chains per wave

enum Op { FMA32, FMA32_DEP, PKFMA32, EXP2, RCP, RSQ, FMA64, ADD64, CVT64, DPP_MOV, DPP_ADD, MIX_PAIR, FMA_SALU, FMA_SALU2, FMA_LDS, FMA_BRANCH, FMA_BRANCH_NT, FMA_EXECZ };

template <int OP>
__global__ void __launch_bounds__(1024) rate_kernel(float* out, unsigned long long* cyc, float seed) {
  extern __shared__ float lds[];
  float a[kUnroll];
  double d[kUnroll];
  float2_ p[kUnroll];
#pragma unroll
  for (int i = 0; i < kUnroll; ++i) {
    a[i] = seed + 0.001f * (float)(threadIdx.x + i);
    d[i] = (double)a[i];
    p[i] = float2_{a[i], a[i] + 1.0f};
  }
  const float m = 0.999f + seed * 1e-6f, c = seed * 1e-3f;
  const double md = (double)m, cd = (double)c;
  const float2_ mp = {m, m}, cp = {c, c};
  float lacc = 0.f;
  lds[threadIdx.x] = seed;
  __syncthreads();
  const unsigned long long t0 = __builtin_amdgcn_s_memtime();
#pragma nounroll
  for (int it = 0; it < kIters; ++it) {
#pragma unroll
    for (int i = 0; i < kUnroll; ++i) {
      if (OP == FMA32) a[i] = fmaf(a[i], m, c);
      if (OP == FMA32_DEP) a[0] = fmaf(a[0], m, c);
      if (OP == PKFMA32) p[i] = __builtin_elementwise_fma(p[i], mp, cp);
      if (OP == EXP2) a[i] = __builtin_amdgcn_exp2f(a[i]);
      if (OP == RCP) a[i] = __builtin_amdgcn_rcpf(a[i]);
      if (OP == RSQ) a[i] = __builtin_amdgcn_rsqf(a[i]);
      if (OP == FMA64) d[i] = fma(d[i], md, cd);
      if (OP == ADD64) d[i] = d[i] + cd;
      if (OP == CVT64) a[i] = fmaf(a[i], m, c), d[i] += (double)a[i];  // the accumulate pattern: fma32 + cvt_f64_f32 + add_f64
      if (OP == DPP_MOV) {
        const int v = __builtin_bit_cast(int, a[i]);
        a[i] = __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false));  // quad_perm [2,3,0,1]
      }
      if (OP == DPP_ADD) {
        const int v = __builtin_bit_cast(int, a[i]);
        a[i] += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false));  // add with a DPP operand
      }
      // does scalar / LDS / branch work issue beside the VALU stream or in its place?  (per 4 FMAs: one s_add_u32, two
      // of them, one ds_read_b32, one taken uniform branch; the table counts the FMAs only)
      if (OP == FMA_SALU || OP == FMA_SALU2 || OP == FMA_LDS || OP == FMA_BRANCH || OP == FMA_BRANCH_NT || OP == FMA_EXECZ) {
        a[i] = fmaf(a[i], m, c);
        if ((i & 3) == 3) {
          if (OP == FMA_SALU || OP == FMA_SALU2) asm volatile("s_add_u32 s90, s90, 1" ::: "s90", "scc");
          if (OP == FMA_SALU2) asm volatile("s_add_u32 s91, s91, 1" ::: "s91", "scc");
          if (OP == FMA_LDS) lacc += lds[(threadIdx.x + i) & 1023];
          if (OP == FMA_BRANCH_NT) asm volatile("s_cmp_eq_u32 s90, 0x12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::: "scc");
          if (OP == FMA_EXECZ) asm volatile("s_cbranch_execz 1f\n\ts_nop 0\n1:" :::);  // the guard of a divergent if: not taken
          if (OP == FMA_BRANCH) asm volatile("s_cmp_lg_u32 s90, 0x12345\n\ts_cbranch_scc1 1f\n\ts_nop 0\n1:" ::: "scc");
        }
      }
      if (OP == MIX_PAIR) {
        // the shape of one obstacle pair: ~8 fma per transcendental
        float x = a[i];
        x = fmaf(x, m, c); x = fmaf(x, m, c); x = fmaf(x, m, c); x = fmaf(x, m, c);
        x = fmaf(x, m, c); x = fmaf(x, m, c); x = fmaf(x, m, c); x = fmaf(x, m, c);
        a[i] = __builtin_amdgcn_rcpf(x);
      }
    }
  }
  const unsigned long long t1 = __builtin_amdgcn_s_memtime();
  float s = 0.f;
#pragma unroll
  for (int i = 0; i < kUnroll; ++i) s += a[i] + (float)d[i] + p[i].x + p[i].y;
  s += lacc;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
  if ((threadIdx.x & 63) == 0) cyc[blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6)] = t1 - t0;
}

template <int OP>
void run(const char* name, double ops_per_inner, float* out, unsigned long long* cyc) {
  hipFuncSetAttribute((const void*)rate_kernel<OP>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
  for (int W : {1, 2, 4}) {
    const int threads = 256 * W, blocks = 256;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    rate_kernel<OP><<<blocks, threads, 96 * 1024>>>(out, cyc, 1.0f);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    rate_kernel<OP><<<blocks, threads, 96 * 1024>>>(out, cyc, 1.0f);
    hipEventRecord(e1);
    hipDeviceSynchronize();
    float ms = 0.f;
    hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(blocks * threads / 64);
    hipMemcpy(h.data(), cyc, h.size() * sizeof(h[0]), hipMemcpyDeviceToHost);
    std::sort(h.begin(), h.end());
    const double med = (double)h[h.size() / 2];  // s_memtime ticks: 100 MHz constant clock on gfx9? reported raw
    const double insts = (double)kIters * kUnroll * ops_per_inner;  // per wave
    // per-SIMD: W waves share a SIMD -> SIMD-cycles per wave-instruction = ticks / (insts * W)
    printf("%-10s W=%d  wave ticks(median) %10.0f  ticks/inst/wave %7.3f  ticks/inst/SIMD %7.3f  wall %8.3f us  ns/inst/SIMD %7.4f\n",
           name, W, med, med / insts, med / (insts * W), ms * 1e3, ms * 1e6 / (insts * W));
  }
}

int main() {
  float* out;
  unsigned long long* cyc;
  hipMalloc(&out, sizeof(float) * 256 * 1024);
  hipMalloc(&cyc, sizeof(unsigned long long) * 256 * 16);
  run<FMA32>("fma32", 1, out, cyc);
  run<FMA32_DEP>("fma32_dep", 1, out, cyc);
  run<PKFMA32>("pk_fma32", 1, out, cyc);
  run<EXP2>("exp2", 1, out, cyc);
  run<RCP>("rcp", 1, out, cyc);
  run<RSQ>("rsq", 1, out, cyc);
  run<FMA64>("fma64", 1, out, cyc);
  run<ADD64>("add64", 1, out, cyc);
  run<CVT64>("fma+cvt+add64", 3, out, cyc);
  run<DPP_MOV>("dpp_mov", 1, out, cyc);
  run<DPP_ADD>("dpp_mov+add", 2, out, cyc);
  run<MIX_PAIR>("8fma+rcp", 9, out, cyc);
  run<FMA_SALU>("4fma|1salu", 1, out, cyc);
  run<FMA_SALU2>("4fma|2salu", 1, out, cyc);
  run<FMA_LDS>("4fma|1lds", 1, out, cyc);
  run<FMA_BRANCH>("4fma|branch", 1, out, cyc);
  run<FMA_BRANCH_NT>("4fma|br-not-taken", 1, out, cyc);
  run<FMA_EXECZ>("4fma|execz-guard", 1, out, cyc);
  return 0;
}
