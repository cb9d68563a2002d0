// stream_pairs.hip -- what the memory system gives interface B's access pattern, WITHOUT the control step around it:
// a fleet's explicit-pair arrays p_link / p_obs [R][256][3] fp32 (6 144 B per robot), read once per launch by waves that own 16
// robots each and walk the 8 leaves of 32 pairs in order -- the pattern of rmp2_quad.h's explicit-pair loops -- with a tunable
// amount of arithmetic per pair.  Variants:
//   flat     every thread reads float4s of both arrays, grid stride (the streaming ceiling of this box for these arrays)
//   reg      per leaf each lane issues its 8 + 8 dwordx3 loads (pair sub + 4 i of its robot), then consumes them (round 3's form)
//   dma      half a leaf ahead by global_load_lds_dwordx4 into a 6 KiB buffer, consumed from registers (round 4's form)
//   dma2     a whole leaf ahead: two 12 KiB buffers, consumed straight from LDS
// `work` = dependent fp32 FMAs per pair (the control step spends ~110 VALU instructions per evaluated pair).
//   hipcc --offload-arch=gfx950 -O3 tools/stream_pairs.hip -o tools/diag/stream_pairs && tools/diag/stream_pairs [R]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

struct F3 { float x, y, z; };

__global__ void __launch_bounds__(256) flat(const float4* __restrict__ a, const float4* __restrict__ b, float* out, size_t n) {
  float s = 0.f;
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    const float4 u = a[i], v = b[i];
    s += u.x + u.w + v.y + v.z;
  }
  if (s == 12345.678f) out[0] = s;
}

template <int WORK>
__device__ __forceinline__ float chew(float acc, F3 a, F3 o) {
  float d = (a.x - o.x) * (a.x - o.x) + (a.y - o.y) * (a.y - o.y) + (a.z - o.z) * (a.z - o.z);
#pragma unroll
  for (int k = 0; k < WORK; ++k) d = fmaf(d, 0.999f, 0.001f);
  return acc + d;
}

template <int WORK>
__global__ void __launch_bounds__(64, 2) reg(const float* __restrict__ pl, const float* __restrict__ po, float* out, int R) {
  extern __shared__ float lds[];
  const int lane = threadIdx.x, sub = lane & 3, g = lane >> 2;
  const int robot = min((int)blockIdx.x * 16 + g, R - 1);
  float acc = 0.f;
  for (int leaf = 0; leaf < 8; ++leaf) {
    const size_t base = ((size_t)robot * 256 + leaf * 32) * 3;
    F3 a[8], o[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      a[i] = *reinterpret_cast<const F3*>(pl + base + 3 * (sub + 4 * i));
      o[i] = *reinterpret_cast<const F3*>(po + base + 3 * (sub + 4 * i));
    }
#pragma unroll
    for (int i = 0; i < 8; ++i) acc = chew<WORK>(acc, a[i], o[i]);
  }
  if (acc == 12345.678f) out[0] = acc + lds[0];
}

__device__ __forceinline__ void glds16(const float* g, float* l) {
  __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)g, (__attribute__((address_space(3))) void*)l, 16, 0, 0);
}

template <int WORK>
__global__ void __launch_bounds__(64, 2) dma(const float* __restrict__ pl, const float* __restrict__ po, float* out, int R) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  float* buf = lds;  // 1536 floats; the launch pads the allocation to the control step's 16.5 KB
  const int lane = threadIdx.x, sub = lane & 3, g = lane >> 2;
  const int r0 = blockIdx.x * 16;
  auto issue = [&](int chunk) {  // chunk = 2 leaf + half
#pragma unroll
    for (int i = 0; i < 3; ++i) {
      const int c = 64 * i + lane, rr = (c * 5462) >> 16, piece = c - 12 * rr;
      const size_t off = ((size_t)min(r0 + rr, R - 1) * 256 + chunk * 16) * 3 + 4 * piece;
      glds16(pl + off, buf + 256 * i);
      glds16(po + off, buf + 768 + 256 * i);
    }
  };
  float acc = 0.f;
  issue(0);
  for (int chunk = 0; chunk < 16; ++chunk) {
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    F3 a[4], o[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
      const float* src = buf + g * 48 + 3 * (sub + 4 * i);
      a[i] = *reinterpret_cast<const F3*>(src);
      o[i] = *reinterpret_cast<const F3*>(src + 768);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    if (chunk + 1 < 16) issue(chunk + 1);
#pragma unroll
    for (int i = 0; i < 4; ++i) acc = chew<WORK>(acc, a[i], o[i]);
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <int WORK>
__global__ void __launch_bounds__(64, 2) dma2(const float* __restrict__ pl, const float* __restrict__ po, float* out, int R) {
  extern __shared__ __attribute__((aligned(16))) float lds[];
  const int lane = threadIdx.x, sub = lane & 3, g = lane >> 2;
  const int r0 = blockIdx.x * 16;
  auto issue = [&](int leaf, float* buf) {  // a whole leaf: [16][32][3] of each array, 24 16-byte pieces per robot
#pragma unroll
    for (int i = 0; i < 6; ++i) {
      const int c = 64 * i + lane, rr = (c * 2731) >> 16, piece = c - 24 * rr;  // c / 24 for c < 384
      const size_t off = ((size_t)min(r0 + rr, R - 1) * 256 + leaf * 32) * 3 + 4 * piece;
      glds16(pl + off, buf + 256 * i);
      glds16(po + off, buf + 1536 + 256 * i);
    }
  };
  float acc = 0.f;
  issue(0, lds);
  for (int leaf = 0; leaf < 8; ++leaf) {
    float* buf = lds + (leaf & 1) * 3072;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    if (leaf + 1 < 8) issue(leaf + 1, lds + ((leaf + 1) & 1) * 3072);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
      const float* src = buf + g * 96 + 3 * (sub + 4 * i);
      acc = chew<WORK>(acc, *reinterpret_cast<const F3*>(src), *reinterpret_cast<const F3*>(src + 1536));
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
  }
  if (acc == 12345.678f) out[0] = acc;
}

template <class K>
float time_us(K launch, int reps) {
  hipEvent_t a, b;
  (void)hipEventCreate(&a), (void)hipEventCreate(&b);
  for (int i = 0; i < 5; ++i) launch();
  (void)hipEventRecord(a);
  for (int i = 0; i < reps; ++i) launch();
  (void)hipEventRecord(b);
  (void)hipEventSynchronize(b);
  float ms = 0.f;
  (void)hipEventElapsedTime(&ms, a, b);
  return ms * 1e3f / reps;
}

int main(int argc, char** argv) {
  const int R = argc > 1 ? std::atoi(argv[1]) : 65536;
  const size_t floats = (size_t)R * 256 * 3, bytes = floats * 4;
  float *pl, *po, *out;
  if (hipMalloc(&pl, bytes) != hipSuccess || hipMalloc(&po, bytes) != hipSuccess || hipMalloc(&out, 64) != hipSuccess) return 1;
  (void)hipMemset(pl, 0, bytes), (void)hipMemset(po, 0, bytes);
  const int blocks = (R + 15) / 16;
  const double gb = 2.0 * bytes * 1e-9;
  const size_t lds_step = 10368, lds_dma = 16512, lds_dma2 = 10368 + 24576;
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma2<0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma2);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma2<32>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma2);
  (void)hipFuncSetAttribute(reinterpret_cast<const void*>(dma2<110>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_dma2);
  printf("R = %d robots, %.1f MB per launch (both arrays); LDS per wave: reg %zu, dma %zu, dma2 %zu B\n", R, gb * 1e3, lds_step, lds_dma, lds_dma2);
  float t = time_us([&] { flat<<<8192, 256>>>((const float4*)pl, (const float4*)po, out, floats / 4); }, 50);
  printf("%-28s %8.1f us  %6.2f TB/s\n", "flat float4 stream", t, gb / t * 1e3);
#define RUN(NAME, KERN, LDSB)                                                                      \
  t = time_us([&] { KERN<<<blocks, 64, LDSB>>>(pl, po, out, R); }, 50);                            \
  printf("%-28s %8.1f us  %6.2f TB/s\n", NAME, t, gb / t * 1e3);
  RUN("reg  work 0", reg<0>, lds_step)
  RUN("reg  work 32", reg<32>, lds_step)
  RUN("reg  work 110", reg<110>, lds_step)
  RUN("dma  work 0", dma<0>, lds_dma)
  RUN("dma  work 32", dma<32>, lds_dma)
  RUN("dma  work 110", dma<110>, lds_dma)
  RUN("dma2 work 0", dma2<0>, lds_dma2)
  RUN("dma2 work 32", dma2<32>, lds_dma2)
  RUN("dma2 work 110", dma2<110>, lds_dma2)
  return 0;
}
