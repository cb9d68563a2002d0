// fetch_calib.hip -- calibrates rocprofv3's FETCH_SIZE / WRITE_SIZE for THIS engine's access pattern: contiguous
// 4-byte-per-lane loads and stores (a wave moves 256 B per instruction), on a known byte count.
// MI355X_MICROARCH.md, section HBM: FETCH_SIZE reads exactly half the bytes of 16-B-per-lane streams on gfx950; other
// widths are uncalibrated -- "calibrate on a known byte count in your own access pattern before trusting an absolute".
//   hipcc --offload-arch=gfx950 -O3 tools/fetch_calib.hip -o tools/diag/fetch_calib
//   rocprofv3 --pmc FETCH_SIZE -d out -- tools/diag/fetch_calib ; rocprofv3 --pmc WRITE_SIZE ... (separate passes)
// Kernels: copy4 (4 B per lane) and copy16 (16 B per lane, the guide's reference case), 512 MiB each way
// (beyond the 256 MiB Infinity Cache), 3 launches each.
#include <hip/hip_runtime.h>
#include <cstdio>

__global__ void copy4(const float* __restrict__ a, float* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) b[i] = a[i] + 1.0f;
}
__global__ void copy16(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
    float4 v = a[i];
    v.x += 1.0f;
    b[i] = v;
  }
}
int main() {
  const size_t bytes = 512ull << 20, n = bytes / 4;
  float *a, *b;
  if (hipMalloc(&a, bytes) != hipSuccess || hipMalloc(&b, bytes) != hipSuccess) return 1;
  (void)hipMemset(a, 0, bytes);
  for (int r = 0; r < 3; ++r) copy4<<<4096, 256>>>(a, b, n);
  for (int r = 0; r < 3; ++r) copy16<<<4096, 256>>>((const float4*)a, (float4*)b, n / 4);
  (void)hipDeviceSynchronize();
  printf("bytes per launch, each way: %zu\n", bytes);
  return 0;
}
