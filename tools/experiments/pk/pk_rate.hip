// Microbenchmark: issue rate of v_fma_f32 against v_pk_fma_f32 / v_pk_mul_f32 / v_pk_add_f32 on gfx950, 4 waves per SIMD.
// build: hipcc --offload-arch=gfx950 -O3 -o pk_rate pk_rate.hip ; run: ./pk_rate
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));

template <int MODE>
__global__ __launch_bounds__(256) void k(float* out, int iters, float a, float b) {
  float x[16];
  for (int i = 0; i < 16; ++i) x[i] = threadIdx.x * 0.001f + i;
  f2 y[8];
  for (int i = 0; i < 8; ++i) y[i] = f2{x[2 * i], x[2 * i + 1]};
  const f2 a2{a, a * 1.0001f}, b2{b, b * 0.999f};
  for (int it = 0; it < iters; ++it) {
    if (MODE == 0) {
#pragma unroll
      for (int i = 0; i < 16; ++i) x[i] = __builtin_fmaf(x[i], a, b);
    } else if (MODE == 1) {
#pragma unroll
      for (int i = 0; i < 8; ++i) y[i] = __builtin_elementwise_fma(y[i], a2, b2);
    } else if (MODE == 2) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(y[i]) : "v"(a2));
    } else if (MODE == 3) {
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(y[i]) : "v"(b2));
    } else if (MODE == 4) {   // 16 scalar fma through asm (same count of instructions as mode 0, not foldable)
#pragma unroll
      for (int i = 0; i < 16; ++i) asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(x[i]) : "v"(a), "v"(b));
    } else if (MODE == 5) {   // 8 packed fma through asm
#pragma unroll
      for (int i = 0; i < 8; ++i) asm volatile("v_pk_fma_f32 %0, %0, %1, %2" : "+v"(y[i]) : "v"(a2), "v"(b2));
    }
  }
  float s = 0;
  for (int i = 0; i < 16; ++i) s += x[i];
  for (int i = 0; i < 8; ++i) s += y[i].x + y[i].y;
  out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

template <int MODE>
void run(const char* name, int insts_per_iter, int flops_per_inst, int blocks) {
  float* out;
  hipMalloc(&out, (size_t)blocks * 256 * 4);
  const int iters = 20000;
  hipEvent_t e0, e1;
  hipEventCreate(&e0);
  hipEventCreate(&e1);
  k<MODE><<<blocks, 256>>>(out, 100, 1.0001f, 0.001f);
  hipDeviceSynchronize();
  hipEventRecord(e0);
  k<MODE><<<blocks, 256>>>(out, iters, 1.0001f, 0.001f);
  hipEventRecord(e1);
  hipDeviceSynchronize();
  float ms;
  hipEventElapsedTime(&ms, e0, e1);
  const double waves_per_simd = blocks * 4.0 / (256 * 4);
  const double inst_per_simd = (double)iters * insts_per_iter * waves_per_simd;
  printf("%-28s blocks %5d: %8.3f ms  %6.2f ns per wave-instruction per SIMD  %7.1f TFLOP/s\n", name, blocks, ms,
         ms * 1e6 / inst_per_simd, (double)blocks * 256 * iters * insts_per_iter * flops_per_inst * 64 / 64 / (ms * 1e-3) / 1e12);
  hipFree(out);
}

int main() {
  for (int blocks : {256, 1024}) {
    run<0>("v_fma_f32 (compiler)", 16, 2, blocks);
    run<4>("v_fma_f32 (asm)", 16, 2, blocks);
    run<1>("v_pk_fma_f32 (compiler)", 8, 4, blocks);
    run<5>("v_pk_fma_f32 (asm)", 8, 4, blocks);
    run<2>("v_pk_mul_f32 (asm)", 8, 2, blocks);
    run<3>("v_pk_add_f32 (asm)", 8, 2, blocks);
  }
  return 0;
}
