// tests/host_emul/kf_host.hip -- TEST INFRASTRUCTURE.  The scalar Kalman arithmetic of csrc/kf_math.h (namespace kf: initiate,
// predict, project, update with its 4x4 Cholesky, mean_to_tlbr, gmc -- __host__ __device__) called on the CPU, batch by batch,
// so that tests/test_kf_host.py can hold it against the golden vectors recorded from the reference (tests/golden/g1_kalman.npz,
// g7_gmc.npz) without a GPU.  Built by hipcc with the product's flags (-ffp-contract=off: NumPy never fuses a*b+c).
#include "../../bot-sort-onnx-tensorrt_amd/csrc/kf_math.h"
#include <stdint.h>

extern "C" void kf_host_initiate(int n, const float *z, double *mean, double *cov) {
    for (int t = 0; t < n; ++t) kf::initiate(z + 4 * t, mean + 8 * t, cov + 64 * t);
}
// zero_vel[t] != 0: mean[6], mean[7] = 0 first (STrack.multi_predict for tracks that are not Tracked, D:529-532)
extern "C" void kf_host_predict(int n, double *mean, double *cov, const uint8_t *f32, const uint8_t *zero_vel) {
    for (int t = 0; t < n; ++t) {
        if (zero_vel && zero_vel[t]) { mean[8 * t + 6] = 0.0; mean[8 * t + 7] = 0.0; }
        kf::predict(mean + 8 * t, cov + 64 * t, f32 && f32[t]);
    }
}
extern "C" void kf_host_project(int n, const double *mean, const double *cov, const uint8_t *f32, double *zhat, double *S) {
    for (int t = 0; t < n; ++t) {
        kf::project(mean + 8 * t, cov + 64 * t, f32 && f32[t], S + 16 * t);
        for (int a = 0; a < 4; ++a) zhat[4 * t + a] = mean[8 * t + a];
    }
}
extern "C" void kf_host_update(int n, double *mean, double *cov, const float *z, const uint8_t *f32) {
    for (int t = 0; t < n; ++t) kf::update(mean + 8 * t, cov + 64 * t, z + 4 * t, f32 && f32[t]);
}
extern "C" void kf_host_tlbr(int n, const double *mean, const uint8_t *f32, double *out) {
    for (int t = 0; t < n; ++t) kf::mean_to_tlbr(mean + 8 * t, f32 && f32[t], out + 4 * t);
}
extern "C" void kf_host_gmc(int n, double *mean, double *cov, const double *h6) {
    for (int t = 0; t < n; ++t) kf::gmc(mean + 8 * t, cov + 64 * t, h6);
}
// squared Mahalanobis distance of m measurements to one track (KalmanFilter.gating_distance, metric 'maha', 4 dof): the
// arithmetic of motion_gate_open / k_gating -- project, chol4, forward substitution
extern "C" void kf_host_maha(const double *mean, const double *cov, int m, const double *meas, double *out) {
    double S[16], L[16];
    kf::project(mean, cov, false, S);
    kf::chol4(S, L);
    for (int q = 0; q < m; ++q) {
        double y[4];
        for (int a = 0; a < 4; ++a) y[a] = meas[4 * q + a] - mean[a];
        for (int i = 0; i < 4; ++i) { double t = y[i]; for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * y[k]; y[i] = t / L[i * 4 + i]; }
        double g = y[0] * y[0]; g += y[1] * y[1]; g += y[2] * y[2]; g += y[3] * y[3];
        out[q] = g;
    }
}
