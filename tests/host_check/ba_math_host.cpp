// Test-only: compiles the PRODUCT's device math header (xrslam_amd/csrc/ba_math.hip.h, all functions are
// __host__ __device__) for the host so the factor arithmetic the kernels execute can be unit-tested
// against the oracle without a GPU.  Built by tests/test_ba_math_host.py with hipcc; never shipped.
#include "../../xrslam_amd/csrc/ba_math.hip.h"

#include <cstring>

using namespace xrhip;

extern "C" {

void hc_reprojection(const double *st_t, const double *st_r, double inv_depth, const double *zt, const double *zr,
                     const double *cam7, const double *sic2, double *r2, double *Jt12, double *Jr12, double *Jl2) {
    Ext cam{Q4{cam7[0], cam7[1], cam7[2], cam7[3]}, v3(cam7[4], cam7[5], cam7[6])};
    eval_reprojection(load_state(st_t), load_state(st_r), inv_depth, v3(zt[0], zt[1], zt[2]), v3(zr[0], zr[1], zr[2]),
                      cam, sic2[0], sic2[1], r2, true, Jt12, Jr12, Jl2);
}
void hc_rotation(const double *st_t, const double *st_r, const double *zt, const double *zr, const double *cam7,
                 const double *sic2, double *r2, double *Jq6) {
    Ext cam{Q4{cam7[0], cam7[1], cam7[2], cam7[3]}, v3(cam7[4], cam7[5], cam7[6])};
    eval_rotation(load_state(st_t), load_state(st_r), v3(zt[0], zt[1], zt[2]), v3(zr[0], zr[1], zr[2]), cam, sic2[0],
                  sic2[1], r2, true, Jq6);
}
// whitened residual and Jacobians (15x15 row-major), like the oracle's orc_eval_imu
void hc_imu(const double *st_i, const double *st_j, const double *imu_data, const double *bias_ref6,
            const double *imu7, double *r15, double *Ji225, double *Jj225) {
    Ext imu{Q4{imu7[0], imu7[1], imu7[2], imu7[3]}, v3(imu7[4], imu7[5], imu7[6])};
    FState fi = load_state(st_i), fj = load_state(st_j);
    ImuRec pre = load_imu(imu_data);
    V3 bg0 = v3(bias_ref6[0], bias_ref6[1], bias_ref6[2]), ba0 = v3(bias_ref6[3], bias_ref6[4], bias_ref6[5]);
    double raw[15], Ji[225], Jj[225];
    std::memset(Ji, 0, sizeof(Ji));
    std::memset(Jj, 0, sizeof(Jj));
    imu_raw_residual(fi, fj, pre, bg0, ba0, imu, raw);
    imu_raw_jacobians(fi, fj, pre, bg0, ba0, imu, v3(raw[0], raw[1], raw[2]), Ji, Jj);
    const double *S = imu_data + 56;
    for (int i = 0; i < 15; ++i) {
        double s = 0;
        for (int k = 0; k < 15; ++k) s += S[15 * i + k] * raw[k];
        r15[i] = s;
        for (int j = 0; j < 15; ++j) {
            double a = 0, b = 0;
            for (int k = 0; k < 15; ++k) {
                a += S[15 * i + k] * Ji[15 * k + j];
                b += S[15 * i + k] * Jj[15 * k + j];
            }
            Ji225[15 * i + j] = a;
            Jj225[15 * i + j] = b;
        }
    }
}
void hc_state_plus(const double *s, const double *d15, double *out) { state_plus(s, d15, true, true, out); }
void hc_logmap(const double *q4, double *w3) {
    V3 w = logmap(Q4{q4[0], q4[1], q4[2], q4[3]});
    w3[0] = w.x;
    w3[1] = w.y;
    w3[2] = w.z;
}
void hc_right_jacobian_inv(const double *w3, double *m9) {
    M3 r = inverse3(right_jacobian(v3(w3[0], w3[1], w3[2])));
    std::memcpy(m9, r.m, sizeof(r.m));
}
}
