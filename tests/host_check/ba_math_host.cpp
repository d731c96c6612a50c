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
// the constant-landmark form (reprojection_constants once + eval_reprojection_cached) beside eval_reprojection: r, Jt, Jr twice
void hc_reprojection_cached(const double *st_t, const double *st_r, double inv_depth, const double *zt, const double *zr,
                            const double *cam7, const double *sic2, int ref_free, double *r4, double *Jt24, double *Jr24) {
    Ext cam{Q4{cam7[0], cam7[1], cam7[2], cam7[3]}, v3(cam7[4], cam7[5], cam7[6])};
    const FState t = load_state(st_t), rf = load_state(st_r);
    const V3 z_t = v3(zt[0], zt[1], zt[2]), z_r = v3(zr[0], zr[1], zr[2]);
    const ObsConst c = reprojection_constants(rf, inv_depth, z_t, z_r, cam);
    std::memset(Jt24, 0, sizeof(double) * 24);
    std::memset(Jr24, 0, sizeof(double) * 24);
    eval_reprojection_cached(t, rf, ref_free != 0, c, z_t, cam, sic2[0], sic2[1], r4, true, Jt24, Jr24);
    double Jl[2];
    eval_reprojection(t, rf, inv_depth, z_t, z_r, cam, sic2[0], sic2[1], r4 + 2, true, Jt24 + 12, Jr24 + 12, Jl);
    if (!ref_free) std::memset(Jr24 + 12, 0, sizeof(double) * 12);
}
// the free-target form with per-frame / per-camera tables (frame_table, ext_table, eval_reprojection_tgt) beside eval_reprojection:
// r (2 + 2) and Jt (12 + 12)
void hc_reprojection_tgt(const double *st_t, const double *st_r, double inv_depth, const double *zt, const double *zr,
                         const double *cam7, const double *sic2, double *r4, double *Jt24) {
    Ext cam{Q4{cam7[0], cam7[1], cam7[2], cam7[3]}, v3(cam7[4], cam7[5], cam7[6])};
    const FState t = load_state(st_t), rf = load_state(st_r);
    const V3 z_t = v3(zt[0], zt[1], zt[2]), z_r = v3(zr[0], zr[1], zr[2]);
    const ObsConst c = reprojection_constants(rf, inv_depth, z_t, z_r, cam);
    double ftab[12], ctab[12];
    frame_table(t, ftab);
    ext_table(cam, ctab);
    eval_reprojection_tgt(ftab, ctab, c, z_t, sic2[0], sic2[1], r4, true, Jt24);
    double Jr[12], Jl[2];
    eval_reprojection(t, rf, inv_depth, z_t, z_r, cam, sic2[0], sic2[1], r4 + 2, true, Jt24 + 12, Jr, Jl);
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
// the same raw residual / Jacobians assembled from the pieces kb_chain deals to its four wavefronts (ba_math.hip.h:
// imu_residual_rq / _rest, imu_raw_jacobians_part 2 and 3, imu_jac_pre / _jrinv / _B / _finish0 / _finish1), and the single-lane
// forms beside them: raw15, Ji225, Jj225 twice ([0] = pieces, [1] = imu_raw_residual + imu_raw_jacobians)
void hc_imu_pieces(const double *st_i, const double *st_j, const double *imu_data, const double *bias_ref6, const double *imu7,
                   int need_i, int need_j, double *raw2, double *Ji2, double *Jj2) {
    Ext imu{Q4{imu7[0], imu7[1], imu7[2], imu7[3]}, v3(imu7[4], imu7[5], imu7[6])};
    FState fi = load_state(st_i), fj = load_state(st_j);
    ImuRec pre = load_imu(imu_data);
    V3 bg0 = v3(bias_ref6[0], bias_ref6[1], bias_ref6[2]), ba0 = v3(bias_ref6[3], bias_ref6[4], bias_ref6[5]);
    std::memset(raw2, 0, sizeof(double) * 30);
    std::memset(Ji2, 0, sizeof(double) * 450);
    std::memset(Jj2, 0, sizeof(double) * 450);
    double xch[IMU_XCH];
    // phase A
    const V3 rq = imu_residual_rq(fi, fj, pre, bg0, imu);
    raw2[0] = rq.x; raw2[1] = rq.y; raw2[2] = rq.z;
    imu_raw_jacobians_part(2, fi, fj, pre, bg0, ba0, imu, v3(0, 0, 0), Ji2, Jj2, need_i, need_j);
    imu_raw_jacobians_part(3, fi, fj, pre, bg0, ba0, imu, v3(0, 0, 0), Ji2, Jj2, need_i, need_j);
    imu_residual_rest(fi, fj, pre, bg0, ba0, imu, raw2);
    imu_jac_pre(fi, fj, pre, bg0, imu, xch);
    // phase B
    const M3 Jr_inv = imu_jac_jrinv(rq);
    store33(xch + 45, Jr_inv);
    store33(xch + 36, imu_jac_B(rq));
    // phase C
    imu_jac_finish0(Jr_inv, xch, Ji2, Jj2, need_i, need_j);
    imu_jac_finish1(load33(xch + 45), load33(xch + 36), xch, load33(imu_data + 11), Ji2, need_i);
    // single lane
    imu_raw_residual(fi, fj, pre, bg0, ba0, imu, raw2 + 15);
    imu_raw_jacobians(fi, fj, pre, bg0, ba0, imu, v3(raw2[15], raw2[16], raw2[17]), Ji2 + 225, Jj2 + 225, need_i, need_j);
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
