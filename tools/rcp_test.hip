// Accuracy of v_rcp_f64 / v_rsq_f64 and of the cubic correction steps dense_lds.hip.h puts behind them (development aid, GPU box).
//   hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -I xrslam_amd/csrc tools/rcp_test.hip -o xrslam_amd/bin/xr-rcp-test
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include "dense_lds.hip.h"
__global__ void k(const double *d, double *r0, double *r1, double *s0, double *s1, int n) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const double x = d[i];
    r0[i] = __builtin_amdgcn_rcp(x);
    r1[i] = xrhip::rcp_cubic(x);
    s0[i] = __builtin_amdgcn_rsq(x);
    s1[i] = xrhip::rsqrt_cubic(x);
}
int main() {
    const int n = 1 << 20;
    double *h = (double *)malloc(8 * n), *g[4];
    for (auto &p : g) p = (double *)malloc(8 * n);
    srand(3);
    for (int i = 0; i < n; ++i) h[i] = exp((rand() / (double)RAND_MAX - 0.5) * 140.0) * (1.0 + rand() / (double)RAND_MAX);
    double *d, *o[4];
    (void)hipMalloc(&d, 8 * n);
    for (auto &p : o) (void)hipMalloc(&p, 8 * n);
    (void)hipMemcpy(d, h, 8 * n, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, d, o[0], o[1], o[2], o[3], n);
    for (int j = 0; j < 4; ++j) (void)hipMemcpy(g[j], o[j], 8 * n, hipMemcpyDeviceToHost);
    double m[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        const long double er = 1.0L / (long double)h[i], es = 1.0L / sqrtl((long double)h[i]);
        m[0] = fmax(m[0], (double)fabsl((g[0][i] - er) / er));
        m[1] = fmax(m[1], (double)fabsl((g[1][i] - er) / er));
        m[2] = fmax(m[2], (double)fabsl((g[2][i] - es) / es));
        m[3] = fmax(m[3], (double)fabsl((g[3][i] - es) / es));
    }
    printf("{\"max_rel_err\": {\"v_rcp_f64\": %.3e, \"rcp_cubic\": %.3e, \"v_rsq_f64\": %.3e, \"rsqrt_cubic\": %.3e}, \"log2\": [%.1f, %.1f, %.1f, %.1f]}\n", m[0], m[1], m[2],
           m[3], log2(m[0]), log2(m[1]), log2(m[2]), log2(m[3]));
    return 0;
}
