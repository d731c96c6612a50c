// Issue cost / latency calibration of ONE wavefront on gfx950 for the operations the scheduled Cholesky block is made of
// (tools/gen_diag16.py's machine model): f64 FMA, 64-bit v_readlane broadcast, v_rcp_f64 / v_rsq_f64, scalar-operand FMAs.
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/issue.hip -o xrslam_amd/bin/xr-issue
// Prints one JSON line: shader cycles per operation of each pattern.  Every pattern is unrolled 64x inside a loop; the order of the
// statements is pinned with scheduling barriers.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } } while (0)
#define SB __builtin_amdgcn_sched_barrier(0)

__device__ __forceinline__ double bcast(double v, int src) {
  const int lo = __builtin_amdgcn_readlane(__double2loint(v), src);
  const int hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
  return __hiloint2double(hi, lo);
}

struct Out { long long clk[32]; double sink[32]; };
#define BEGIN() do { SB; c0 = __builtin_readcyclecounter(); SB; } while (0)
#define END(slot, val, per) do { SB; const long long c1 = __builtin_readcyclecounter(); SB; if (threadIdx.x == 0) { o->clk[slot] = (c1 - c0) / (per); o->sink[slot] = (val); } } while (0)

__global__ __launch_bounds__(64) void k_issue(Out *o, int N, double seed) {
  const int lane = threadIdx.x;
  long long c0;
  double y = 1.0 + 1e-9 * lane;
  double a[8];
  for (int i = 0; i < 8; ++i) a[i] = seed + i + lane * 1e-6;
  // 0: 8 independent fma chains (issue of v_fma_f64)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = fma(a[i], y, 1e-9); SB; }
  }
  END(0, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 1: 1 dependent fma chain (latency)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[0] = fma(a[0], y, 1e-9); SB; }
  }
  END(1, a[0], 8 * N);
  // 2: readlane pair immediately consumed by an fma with the scalar operand, 8 independent accumulators
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = fma(bcast(y, i), a[i], 1e-9); SB; }
  }
  END(2, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 3: the same, the eight broadcasts first, then the eight fmas
  BEGIN();
  for (int it = 0; it < N; ++it) {
    double b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { b[i] = bcast(y, i); SB; }
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = fma(b[i], a[i], 1e-9); SB; }
  }
  END(3, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 4: eight broadcasts alone (consumed once at the end of the iteration by cheap scalar adds -> v_readlane issue)
  {
    int acc = 0;
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        acc += __builtin_amdgcn_readlane(__double2loint(y) + it, i) + __builtin_amdgcn_readlane(__double2hiint(y) + it, i);
        SB;
      }
    }
    END(4, (double)acc, 8 * N);
  }
  // 5: fma whose multiplier is a scalar pair that was broadcast long ago (scalar-operand fma issue)
  {
    const double s0 = bcast(y, 3);
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { a[i] = fma(s0, a[i], 1e-9); SB; }
    }
    END(5, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  }
  // 6: dependent chain through a broadcast: x = bcast(x, k) * y  (latency of VALU -> readlane -> VALU)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[0] = bcast(a[0], i) * y; SB; }
  }
  END(6, a[0], 8 * N);
  // 7: v_rcp_f64, 8 independent (issue)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = __builtin_amdgcn_rcp(a[i]); SB; }
  }
  END(7, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 8: v_rcp_f64 dependent chain (latency)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[0] = __builtin_amdgcn_rcp(a[0]); SB; }
  }
  END(8, a[0], 8 * N);
  // 9: v_rsq_f64 dependent chain (latency)
  for (int i = 0; i < 8; ++i) a[i] = 1.5 + i;
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[0] = __builtin_amdgcn_rsq(a[0]); SB; }
  }
  END(9, a[0], 8 * N);
  // 10: v_mul_f64 issue, 8 independent
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = a[i] * y; SB; }
  }
  END(10, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 11: fma chain of depth 2 interleaved 4 wide (what a P/Q pair of the block looks like)
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = fma(a[i], y, 1e-9); SB; }
#pragma unroll
    for (int i = 0; i < 4; ++i) { a[i] = fma(a[i], y, 2e-9); SB; }
  }
  END(11, a[0] + a[1] + a[2] + a[3], 8 * N);
  // 12: 64-bit select (v_cndmask pair), 8 independent
  BEGIN();
  for (int it = 0; it < N; ++it) {
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = (lane == ((it + i) & 63)) ? y : a[i]; SB; }
  }
  END(12, a[0] + a[1] + a[2] + a[3] + a[4] + a[5] + a[6] + a[7], 8 * N);
  // 13: f64 compare + and into a lane mask (the block's positivity test)
  {
    bool ok = true;
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { ok = ok && (a[i] > -1.0 - it); SB; }
    }
    END(13, ok ? 1.0 : 0.0, 8 * N);
  }
  // 14: s_memtime itself
  {
    long long acc = 0;
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc += __builtin_readcyclecounter(); SB; }
    }
    END(14, (double)acc, 8 * N);
  }
  // 15: LDS store -> load of the same wavefront (round trip, dependent)
  {
    __shared__ double lds[64];
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) {
        lds[lane] = a[0];
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        a[0] = lds[lane ^ 1] * y;
        SB;
      }
    }
    END(15, a[0], 8 * N);
  }
  // 16: v_mfma_f64_16x16x4 dependent chain (accumulator latency) and 17: two independent accumulators (issue)
  {
    typedef double d4 __attribute__((ext_vector_type(4)));
    d4 acc = {0, 0, 0, 0}, acc2 = {0, 0, 0, 0};
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 8; ++i) { acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0); SB; }
    }
    END(16, acc[0], 8 * N);
    BEGIN();
    for (int it = 0; it < N; ++it) {
#pragma unroll
      for (int i = 0; i < 4; ++i) {
        acc = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc, 0, 0, 0);
        SB;
        acc2 = __builtin_amdgcn_mfma_f64_16x16x4f64(y, y, acc2, 0, 0, 0);
        SB;
      }
    }
    END(17, acc[0] + acc2[1], 8 * N);
  }
}

int main() {
  Out *o;
  CK(hipMalloc(&o, sizeof(Out)));
  const int N = 500;
  Out r;
  for (int rep = 0; rep < 3; ++rep) {
    hipLaunchKernelGGL(k_issue, dim3(1), dim3(64), 0, 0, o, N, 0.37);
    CK(hipDeviceSynchronize());
    CK(hipMemcpy(&r, o, sizeof(r), hipMemcpyDeviceToHost));
  }
  const char *names[18] = {"fma_f64_issue", "fma_f64_dependent", "bcast_then_fma_issue", "8_bcasts_then_8_fmas", "readlane_pair_issue", "fma_scalar_operand_issue",
                           "bcast_mul_dependent", "rcp_f64_issue", "rcp_f64_dependent", "rsq_f64_dependent", "mul_f64_issue", "fma_depth2_x4", "select64_issue",
                           "cmp_and_issue", "s_memtime", "lds_roundtrip_dependent", "mfma_f64_16x16x4_dependent", "mfma_f64_16x16x4_two_accumulators"};
  printf("{");
  for (int s = 0; s < 18; ++s) printf("%s\"%s\": %lld", s ? ", " : "", names[s], r.clk[s]);
  printf("}\n");
  return 0;
}
