#include <hip/hip_runtime.h>
#include <cstdio>
template <int N> __device__ __forceinline__ double row_bcast(double v) {
  const int lo = __builtin_amdgcn_update_dpp(0, __double2loint(v), 0x150 + N, 0xf, 0xf, false);
  const int hi = __builtin_amdgcn_update_dpp(0, __double2hiint(v), 0x150 + N, 0xf, 0xf, false);
  return __hiloint2double(hi, lo);
}
__device__ __forceinline__ double swap16(double v) {   // value of the lane 16 away (rows 0<->1, 2<->3)
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  unsigned lo = __double2loint(v), hi = __double2hiint(v);
  v2u a = __builtin_amdgcn_permlane16_swap(lo, lo, false, false);
  v2u b = __builtin_amdgcn_permlane16_swap(hi, hi, false, false);
  // which element holds the swapped value? print both
  return __hiloint2double((int)b[0], (int)a[0]);
}
__global__ void k(double *out, long long *t, int N) {
  const int lane = threadIdx.x;
  double x = 100.0 + lane;
  out[lane] = row_bcast<3>(x);
  out[64 + lane] = swap16(x);
  typedef unsigned v2u __attribute__((ext_vector_type(2)));
  v2u a = __builtin_amdgcn_permlane16_swap((unsigned)lane, (unsigned)(1000 + lane), false, false);
  out[128 + lane] = a[0];
  out[192 + lane] = a[1];
  double y = 1.0 + 1e-9 * lane;
  long long t0 = wall_clock64();
  for (int i = 0; i < N; ++i) x = row_bcast<5>(x) * y;
  long long t1 = wall_clock64();
  for (int i = 0; i < N; ++i) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(x), 5), hi = __builtin_amdgcn_readlane(__double2hiint(x), 5);
    x = __hiloint2double(hi, lo) * y;
  }
  long long t2 = wall_clock64();
  for (int i = 0; i < N; ++i) x = swap16(x) * y;
  long long t3 = wall_clock64();
  for (int i = 0; i < N; ++i) x = x * y;
  long long t4 = wall_clock64();
  for (int i = 0; i < N; ++i) x = fma(x, y, 1e-9);
  long long t5 = wall_clock64();
  if (lane == 0) { t[0] = t1 - t0; t[1] = t2 - t1; t[2] = t3 - t2; t[3] = t4 - t3; t[4] = t5 - t4; }
  out[256 + lane] = x;
}
int main() {
  double *o; long long *t; hipMalloc(&o, 8 * 512); hipMalloc(&t, 64);
  const int N = 4000;
  for (int r = 0; r < 3; ++r) { hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, o, t, N); hipDeviceSynchronize(); }
  double h[512]; long long ht[8]; hipMemcpy(h, o, 8 * 512, hipMemcpyDeviceToHost); hipMemcpy(ht, t, 64, hipMemcpyDeviceToHost);
  printf("row_bcast<3>:"); for (int i = 0; i < 64; i += 5) printf(" %d:%g", i, h[i]); printf("\n");
  printf("swap16:"); for (int i = 0; i < 64; i += 5) printf(" %d:%g", i, h[64 + i]); printf("\n");
  printf("pl16swap a0:"); for (int i = 0; i < 64; i += 5) printf(" %d:%g", i, h[128 + i]); printf("\n");
  printf("pl16swap a1:"); for (int i = 0; i < 64; i += 5) printf(" %d:%g", i, h[192 + i]); printf("\n");
  printf("ns per step: dpp_bcast*mul %.1f readlane*mul %.1f swap16*mul %.1f mul %.1f fma %.1f\n", ht[0] * 10.0 / N, ht[1] * 10.0 / N, ht[2] * 10.0 / N, ht[3] * 10.0 / N, ht[4] * 10.0 / N);
  return 0;
}
