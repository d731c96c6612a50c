#include <hip/hip_runtime.h>
#include <cstdio>
#include <cmath>
#include <cstdlib>
__global__ void k(const double *d, double *r0, double *r1, double *r2, int n) {
  int i = blockIdx.x * blockDim.x + threadIdx.x;
  if (i >= n) return;
  const double x = d[i];
  const double a = __builtin_amdgcn_rsq(x);
  r0[i] = a;
  // one step, 3 dependent ops: t = x a; e = fma(-t, a, 1); r = fma(0.5 a, e, a)
  const double t = x * a, e = fma(-t, a, 1.0);
  r1[i] = fma(0.5 * a, e, a);
  // current scheme's hh*2
  double dd = x * a, hh = 0.5 * a, ee = fma(-hh, dd, 0.5);
  dd = fma(dd, ee, dd); hh = fma(hh, ee, hh); ee = fma(-hh, dd, 0.5); dd = fma(dd, ee, dd); hh = fma(hh, ee, hh);
  r2[i] = hh + hh;
}
int main() {
  const int n = 1 << 20;
  double *h = (double *)malloc(8 * n), *g0 = (double *)malloc(8 * n), *g1 = (double *)malloc(8 * n), *g2 = (double *)malloc(8 * n);
  srand(3);
  for (int i = 0; i < n; ++i) h[i] = exp((rand() / (double)RAND_MAX - 0.5) * 40.0) * (1.0 + rand() / (double)RAND_MAX);
  double *d, *a, *b, *c;
  hipMalloc(&d, 8 * n); hipMalloc(&a, 8 * n); hipMalloc(&b, 8 * n); hipMalloc(&c, 8 * n);
  hipMemcpy(d, h, 8 * n, hipMemcpyHostToDevice);
  hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, a, b, c, n);
  hipMemcpy(g0, a, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(g1, b, 8 * n, hipMemcpyDeviceToHost); hipMemcpy(g2, c, 8 * n, hipMemcpyDeviceToHost);
  double m0 = 0, m1 = 0, m2 = 0;
  for (int i = 0; i < n; ++i) {
    const long double ex = 1.0L / sqrtl((long double)h[i]);
    m0 = fmax(m0, (double)fabsl((g0[i] - ex) / ex)); m1 = fmax(m1, (double)fabsl((g1[i] - ex) / ex)); m2 = fmax(m2, (double)fabsl((g2[i] - ex) / ex));
  }
  printf("max rel err: v_rsq_f64 %.3e (2^%.1f)  one step %.3e (2^%.1f)  two coupled steps %.3e (2^%.1f)\n", m0, log2(m0), m1, log2(m1), m2, log2(m2));
  return 0;
}
