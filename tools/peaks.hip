// Measured peaks of the box the bench runs on (SURVEY.md 8d "Peaks to divide by"): stream-copy GB/s and an f64 MFMA
// micro-kernel TFLOP/s.  Prints one JSON line; tools/gpu_profile.sh stores it as gpurun_out/peaks_<tag>.json and
// bench.py divides by profiles/r*_peaks.json when present (the vendor figures stay in the line as *_spec).
//   hipcc --offload-arch=gfx950 -O3 tools/peaks.hip -o xrslam_amd/bin/xr-peaks
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)

typedef double double4_t __attribute__((ext_vector_type(4)));

// 16 B per lane, grid-stride: the access pattern of the image kernels (k_pyrdown, k_scharr write 16 B per lane)
__global__ __launch_bounds__(256) void k_copy(const float4* __restrict__ a, float4* __restrict__ b, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  for (; i < n; i += st) b[i] = a[i];
}
__global__ __launch_bounds__(256) void k_read(const float4* __restrict__ a, float* __restrict__ out, size_t n) {
  size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x, st = (size_t)gridDim.x * blockDim.x;
  float s = 0.f;
  for (; i < n; i += st) { float4 v = a[i]; s += v.x + v.y + v.z + v.w; }
  if (s == 123.456f) out[0] = s;
}

// 4 independent accumulator chains per wavefront of v_mfma_f64_16x16x4_f64 (the instruction of kb_schur_aux / km_chol)
__global__ __launch_bounds__(256) void k_mfma64(double* out, int iters) {
  double a = 1.0 + threadIdx.x * 1e-9, b = 1.0 - threadIdx.x * 1e-9;
  double4_t c0 = {0, 0, 0, 0}, c1 = c0, c2 = c0, c3 = c0;
  for (int i = 0; i < iters; ++i) {
    c0 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c0, 0, 0, 0);
    c1 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c1, 0, 0, 0);
    c2 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c2, 0, 0, 0);
    c3 = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, c3, 0, 0, 0);
  }
  double s = c0[0] + c1[1] + c2[2] + c3[3];
  if (s == 0.123) out[0] = s;
}

static float timed(void (*launch)(void*), void* ctx, int reps) {
  hipEvent_t e0, e1;
  CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
  launch(ctx); CK(hipDeviceSynchronize());
  CK(hipEventRecord(e0, 0));
  for (int r = 0; r < reps; ++r) launch(ctx);
  CK(hipEventRecord(e1, 0)); CK(hipEventSynchronize(e1));
  float ms = 0; CK(hipEventElapsedTime(&ms, e0, e1));
  return ms / reps;
}

struct Cp { float4 *a, *b; size_t n; int grid; float* o; };
static void l_copy(void* p) { Cp* c = (Cp*)p; k_copy<<<c->grid, 256>>>(c->a, c->b, c->n); }
static void l_read(void* p) { Cp* c = (Cp*)p; k_read<<<c->grid, 256>>>(c->a, c->o, c->n); }
struct Mf { double* o; int iters, grid; };
static void l_mfma(void* p) { Mf* m = (Mf*)p; k_mfma64<<<m->grid, 256>>>(m->o, m->iters); }

int main() {
  hipDeviceProp_t pr; CK(hipGetDeviceProperties(&pr, 0));
  size_t bytes = (size_t)2 << 30;        // 2 GiB per buffer: far beyond the 256 MiB Infinity Cache
  Cp c; c.n = bytes / 16; c.grid = pr.multiProcessorCount * 8;
  CK(hipMalloc(&c.a, bytes)); CK(hipMalloc(&c.b, bytes)); CK(hipMalloc(&c.o, 64));
  CK(hipMemset(c.a, 1, bytes)); CK(hipMemset(c.b, 2, bytes));
  float ms_c = timed(l_copy, &c, 10), ms_r = timed(l_read, &c, 10);
  Mf m; m.iters = 4096; m.grid = pr.multiProcessorCount * 8; CK(hipMalloc(&m.o, 64));
  float ms_m = timed(l_mfma, &m, 5);
  double flops = (double)m.grid * 4 /*waves*/ * m.iters * 4 /*chains*/ * 2.0 * 16 * 16 * 4;
  printf("{\"device\": \"%s\", \"cus\": %d, \"clock_mhz\": %d, \"stream_copy_gbs\": %.1f, \"stream_read_gbs\": %.1f, "
         "\"mfma_f64_16x16x4_tflops\": %.2f, \"copy_bytes_each_way\": %zu}\n",
         pr.gcnArchName, pr.multiProcessorCount, pr.clockRate / 1000, 2.0 * bytes / (ms_c * 1e-3) / 1e9,
         (double)bytes / (ms_r * 1e-3) / 1e9, flops / (ms_m * 1e-3) / 1e12, bytes);
  return 0;
}
