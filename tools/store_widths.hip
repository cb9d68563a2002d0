#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x3 __attribute__((ext_vector_type(3)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
__global__ void st3(float* out, size_t n3) {   // n3 = number of 12-byte records
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(f32x3{1.f, 2.f, 3.f}, reinterpret_cast<f32x3*>(out + 3 * i));
}
__global__ void st4(float* out, size_t n4) {
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += (size_t)gridDim.x * blockDim.x)
    __builtin_nontemporal_store(f32x4{1.f, 2.f, 3.f, 4.f}, reinterpret_cast<f32x4*>(out + 4 * i));
}
__global__ void st3p(float* out, size_t n3) {   // plain (temporal) dwordx3
  for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n3; i += (size_t)gridDim.x * blockDim.x)
    *reinterpret_cast<f32x3*>(out + 3 * i) = f32x3{1.f, 2.f, 3.f};
}
template <class K> float t_us(K k) {
  hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
  for (int i = 0; i < 3; ++i) k();
  hipEventRecord(a); for (int i = 0; i < 20; ++i) k(); hipEventRecord(b); hipEventSynchronize(b);
  float ms; hipEventElapsedTime(&ms, a, b); return ms * 50.f;
}
int main() {
  const size_t bytes = 402653184ull;  // the closest-point stage's output at 65 536 robots
  float* o; hipMalloc(&o, bytes);
  float t;
  t = t_us([&] { st3<<<4096, 64>>>(o, bytes / 12); }); printf("dwordx3 nt  : %7.1f us %5.2f TB/s\n", t, bytes / t * 1e-6);
  t = t_us([&] { st4<<<4096, 64>>>(o, bytes / 16); }); printf("dwordx4 nt  : %7.1f us %5.2f TB/s\n", t, bytes / t * 1e-6);
  t = t_us([&] { st3p<<<4096, 64>>>(o, bytes / 12); }); printf("dwordx3     : %7.1f us %5.2f TB/s\n", t, bytes / t * 1e-6);
  t = t_us([&] { hipMemsetAsync(o, 0, bytes, 0); }); printf("memset      : %7.1f us %5.2f TB/s\n", t, bytes / t * 1e-6);
  return 0;
}
