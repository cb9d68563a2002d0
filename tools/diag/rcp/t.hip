#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
__global__ void k(const double* x, double* r0, double* r1) {
  int i = threadIdx.x + blockIdx.x * blockDim.x;
  double v = x[i];
  double r = __builtin_amdgcn_rcp(v);
  r0[i] = r;
  r = fma(fma(-v, r, 1.0), r, r);
  r1[i] = r;
}
int main() {
  const int n = 4096;
  double *x, *r0, *r1;
  hipMallocManaged(&x, n * 8); hipMallocManaged(&r0, n * 8); hipMallocManaged(&r1, n * 8);
  for (int i = 0; i < n; ++i) x[i] = 0.001 + 1000.0 * (double)rand() / RAND_MAX;
  hipLaunchKernelGGL(k, dim3(n / 64), dim3(64), 0, 0, x, r0, r1);
  hipDeviceSynchronize();
  double e0 = 0, e1 = 0;
  for (int i = 0; i < n; ++i) { e0 = fmax(e0, fabs(r0[i] * x[i] - 1.0)); e1 = fmax(e1, fabs(r1[i] * x[i] - 1.0)); }
  printf("rcp_f64 raw max rel err %.3e ; after one Newton step %.3e\n", e0, e1);
  return 0;
}
