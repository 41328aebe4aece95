// tests/host_emul/kf_host.hip -- TEST INFRASTRUCTURE.  The scalar Kalman arithmetic of csrc/kf_math.h (namespace kf: initiate,
// predict, project, update with its 4x4 Cholesky, mean_to_tlbr, gmc -- __host__ __device__) called on the CPU, batch by batch,
// so that tests/test_product_math_host.py can hold it against the golden vectors recorded from the reference (tests/golden/g1_kalman.npz,
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

// ---- the 8-lane forms (kf8::predict / kf8::update / kf8::tlbr: what tk_predict, the predict workgroups of k_filter and the
// Kalman ranges of tk_mid2 / tk_apply23 execute) on the CPU.  A group's eight lanes run in LOCKSTEP on the device and trade
// values through grp() = __shfl(v, src, 8).  Here grp() is served by REPLAY: to learn what the lanes hold at their c-th grp()
// call, every lane is re-run from the start with the answers of calls 0 .. c-1 and stopped at call c, where it leaves its
// value; when a run passes the last call the lanes' results are final.  (All lanes make the same sequence of grp() calls:
// none sits under a lane-dependent branch.)
#include <array>
#include <vector>
namespace {
struct Stop {};
struct Replay {
    std::vector<std::array<double, 8>> rec;     // rec[c][lane] = the value lane offered at its c-th grp() call
    std::array<double, 8> pending;
    int lane = 0, call = 0;
} R;
}  // namespace
#if !defined(__HIP_DEVICE_COMPILE__)      // (the hook only exists in the host pass of kf_math.h)
double kf8::kf8_host_grp(double v, int src) {
    const int c = R.call++;
    if (c < (int)R.rec.size()) return R.rec[c][src & 7];
    R.pending[R.lane] = v;
    throw Stop();
}
#endif
template <class Body> static void lockstep(Body body) {      // body(lane) runs one lane start to end
    R.rec.clear();
    for (;;) {
        int stopped = 0;
        for (int l = 0; l < 8; ++l) {
            R.lane = l; R.call = 0;
            try { body(l); } catch (const Stop &) { ++stopped; }
        }
        if (stopped == 0) return;                             // every lane ran to its end: results are in place
        R.rec.push_back(R.pending);
    }
}
extern "C" void kf8_host_predict(int n, double *mean, double *cov, const uint8_t *f32, const uint8_t *zero_vel) {
    for (int t = 0; t < n; ++t) {
        double m0[8], P0[64], mo[8], Po[64];
        for (int k = 0; k < 8; ++k) m0[k] = mean[8 * t + k];
        for (int k = 0; k < 64; ++k) P0[k] = cov[64 * t + k];
        if (zero_vel && zero_vel[t]) { m0[6] = 0.0; m0[7] = 0.0; }          // predict_body: `if (i >= 6 && state != Tracked) mi = 0`
        lockstep([&](int i) {
            double mi = m0[i], Pr[8];
            for (int j = 0; j < 8; ++j) Pr[j] = P0[i * 8 + j];
            kf8::predict(mi, Pr, i, f32 && f32[t]);
            mo[i] = mi;
            for (int j = 0; j < 8; ++j) Po[i * 8 + j] = Pr[j];
        });
        for (int k = 0; k < 8; ++k) mean[8 * t + k] = mo[k];
        for (int k = 0; k < 64; ++k) cov[64 * t + k] = Po[k];
    }
}
extern "C" void kf8_host_update(int n, double *mean, double *cov, const float *z, const uint8_t *f32) {
    for (int t = 0; t < n; ++t) {
        double m0[8], P0[64], mo[8], Po[64];
        for (int k = 0; k < 8; ++k) m0[k] = mean[8 * t + k];
        for (int k = 0; k < 64; ++k) P0[k] = cov[64 * t + k];
        lockstep([&](int i) {
            double mi = m0[i], Pr[8];
            for (int j = 0; j < 8; ++j) Pr[j] = P0[i * 8 + j];
            kf8::update(mi, Pr, i, z + 4 * t, f32 && f32[t]);
            mo[i] = mi;
            for (int j = 0; j < 8; ++j) Po[i * 8 + j] = Pr[j];
        });
        for (int k = 0; k < 8; ++k) mean[8 * t + k] = mo[k];
        for (int k = 0; k < 64; ++k) cov[64 * t + k] = Po[k];
    }
}
extern "C" void kf8_host_tlbr(int n, const double *mean, const uint8_t *f32, double *out) {
    for (int t = 0; t < n; ++t) {
        double b[8][4];
        lockstep([&](int i) { kf8::tlbr(mean[8 * t + i], f32 && f32[t], b[i]); });
        for (int c = 0; c < 4; ++c) out[4 * t + c] = b[5][c];               // (every lane gets the box: take any)
    }
}
