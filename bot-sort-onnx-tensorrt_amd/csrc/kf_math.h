// kf_math.h -- per-track Kalman arithmetic, held entirely in registers (8 + 64 doubles).
//
// Restates the 8-state constant-velocity filter of the reference (demo_bottrack_onnx_tflite.py,
// "D:line"): KalmanFilter.initiate D:166-197, predict/multi_predict D:199-302, project D:236-263,
// update D:304-336, gating_distance D:338-380.  The transition matrix F = I + [0 I; 0 0] and the
// observation matrix H = [I 0] are never formed: F P F^T is four block additions and H P H^T is
// the top-left 4x4 block, which is also exactly what the reference's BLAS calls evaluate (every
// other product term is a multiplication by 0 or 1).
//
// Compile with -ffp-contract=off: NumPy does not fuse a*b+c and bit-parity of the predict step
// depends on that.
#pragma once
#include <hip/hip_runtime.h>

namespace kf {
// (the scalar forms below are __host__ __device__: tests/host_emul/kf_host.hip runs them on the CPU against the golden vectors
//  recorded from the reference -- the same arithmetic the standalone kernels and the tracker's initiate / project / gmc use)

constexpr double W_POS = 1.0 / 20;   // D:163
constexpr double W_VEL = 1.0 / 160;  // D:164

// Noise std-devs squared for a box of width w, height h.  `f32` reproduces NumPy-2 scalar
// promotion for float32 state (python_float * np.float32 -> float32), D:183-196 / D:283-292.
__host__ __device__ __forceinline__ double sq_scaled(double weight, double x, bool f32) {
    if (f32) {
        float s = (float)weight * (float)x;
        return (double)(s * s);
    }
    double s = weight * x;
    return s * s;
}

// mean' = F mean ; P' = F P F^T + Q(w,h)   (D:265-302)
__host__ __device__ __forceinline__ void predict(double m[8], double P[64], bool f32) {
    const double qp_w = sq_scaled(W_POS, m[2], f32), qp_h = sq_scaled(W_POS, m[3], f32);
    const double qv_w = sq_scaled(W_VEL, m[2], f32), qv_h = sq_scaled(W_VEL, m[3], f32);
#pragma unroll
    for (int i = 0; i < 4; ++i) m[i] = m[i] + m[i + 4];
    // left = F P : rows 0..3 += rows 4..7
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) P[i * 8 + j] = P[i * 8 + j] + P[(i + 4) * 8 + j];
    // left F^T : cols 0..3 += cols 4..7
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) P[i * 8 + j] = P[i * 8 + j] + P[i * 8 + j + 4];
    P[0] += qp_w;  P[9] += qp_h;  P[18] += qp_w; P[27] += qp_h;
    P[36] += qv_w; P[45] += qv_h; P[54] += qv_w; P[63] += qv_h;
}

// Innovation covariance S = P[:4,:4] + R(w,h)   (D:236-263); zhat = m[:4].
__host__ __device__ __forceinline__ void project(const double m[8], const double P[64], bool f32, double S[16]) {
    const double r_w = sq_scaled(W_POS, m[2], f32), r_h = sq_scaled(W_POS, m[3], f32);
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) S[i * 4 + j] = P[i * 8 + j];
    S[0] += r_w; S[5] += r_h; S[10] += r_w; S[15] += r_h;
}

// Lower Cholesky factor of the 4x4 S (reads the lower triangle, like cho_factor(lower=True)).
__host__ __device__ __forceinline__ void chol4(const double S[16], double L[16]) {
#pragma unroll
    for (int i = 0; i < 16; ++i) L[i] = 0.0;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        double s = S[j * 4 + j];
#pragma unroll
        for (int k = 0; k < j; ++k) s -= L[j * 4 + k] * L[j * 4 + k];
        const double ljj = sqrt(s);
        L[j * 4 + j] = ljj;
#pragma unroll
        for (int i = j + 1; i < 4; ++i) {
            double t = S[i * 4 + j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= L[i * 4 + k] * L[j * 4 + k];
            L[i * 4 + j] = t / ljj;
        }
    }
}

// Solve (L L^T) x = b for one right-hand side of length 4, in place.
__host__ __device__ __forceinline__ void chol4_solve(const double L[16], double b[4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {   // forward: L y = b
        double t = b[i];
#pragma unroll
        for (int k = 0; k < i; ++k) t -= L[i * 4 + k] * b[k];
        b[i] = t / L[i * 4 + i];
    }
#pragma unroll
    for (int i = 3; i >= 0; --i) {  // backward: L^T x = y
        double t = b[i];
#pragma unroll
        for (int k = i + 1; k < 4; ++k) t -= L[k * 4 + i] * b[k];
        b[i] = t / L[i * 4 + i];
    }
}

// Measurement update with z = (cx,cy,w,h) (float32 in the reference, D:304-336):
//   K = P H^T S^-1 via the Cholesky factor; m += K (z - Hm); P -= K S K^T.
__host__ __device__ __forceinline__ void update(double m[8], double P[64], const float z[4], bool f32) {
    double S[16], L[16];
    project(m, P, f32, S);
    chol4(S, L);
    double K[32];  // K[i*4+a]
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double b[4] = {P[i * 8 + 0], P[i * 8 + 1], P[i * 8 + 2], P[i * 8 + 3]};  // row i of P H^T
        chol4_solve(L, b);
#pragma unroll
        for (int a = 0; a < 4; ++a) K[i * 4 + a] = b[a];
    }
    double y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) y[a] = (double)z[a] - m[a];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        double s = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) s += y[a] * K[i * 4 + a];
        m[i] = m[i] + s;
    }
    double KS[32];  // (K S)[i*4+b]
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int b = 0; b < 4; ++b) {
            double s = 0.0;
#pragma unroll
            for (int a = 0; a < 4; ++a) s += K[i * 4 + a] * S[a * 4 + b];
            KS[i * 4 + b] = s;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            double s = 0.0;
#pragma unroll
            for (int b = 0; b < 4; ++b) s += KS[i * 4 + b] * K[j * 4 + b];
            P[i * 8 + j] = P[i * 8 + j] - s;
        }
}

// KalmanFilter.initiate for a float32 measurement (cx,cy,w,h), D:166-197: the std-devs are
// float32 products (2*w_pos = 0.1, 10*w_vel = 0.0625 as python floats, demoted to float32).
__host__ __device__ __forceinline__ void initiate(const float z[4], double m[8], double P[64]) {
#pragma unroll
    for (int i = 0; i < 4; ++i) { m[i] = (double)z[i]; m[i + 4] = 0.0; }
#pragma unroll
    for (int i = 0; i < 64; ++i) P[i] = 0.0;
    const float kp = (float)(2 * W_POS), kv = (float)(10 * W_VEL);
    const float sp_w = kp * z[2], sp_h = kp * z[3], sv_w = kv * z[2], sv_h = kv * z[3];
    P[0] = (double)(sp_w * sp_w);  P[9] = (double)(sp_h * sp_h);
    P[18] = (double)(sp_w * sp_w); P[27] = (double)(sp_h * sp_h);
    P[36] = (double)(sv_w * sv_w); P[45] = (double)(sv_h * sv_h);
    P[54] = (double)(sv_w * sv_w); P[63] = (double)(sv_h * sv_h);
}

// STrack.tlbr of a Kalman mean (D:624-642).  For float32 state every step is float32.
__host__ __device__ __forceinline__ void mean_to_tlbr(const double m[8], bool f32, double out[4]) {
    if (f32) {
        float w = (float)m[2], h = (float)m[3];
        float x = (float)m[0] - w / 2, y = (float)m[1] - h / 2;
        out[0] = (double)x; out[1] = (double)y; out[2] = (double)(w + x); out[3] = (double)(h + y);
    } else {
        double x = m[0] - m[2] / 2, y = m[1] - m[3] / 2;
        out[0] = x; out[1] = y; out[2] = m[2] + x; out[3] = m[3] + y;
    }
}

// STrack.multi_gmc (D:538-554): camera-motion warp  x <- kron(I4, R) x (+ t on the centre),
// P <- R8 P R8^T  with R = H[:2,:2], t = H[:2,2].  R8 is block diagonal: pairs (0,1),(2,3),(4,5),(6,7).
__host__ __device__ __forceinline__ void gmc(double m[8], double P[64], const double H[6]) {
    const double r00 = H[0], r01 = H[1], tx = H[2], r10 = H[3], r11 = H[4], ty = H[5];
#pragma unroll
    for (int p = 0; p < 4; ++p) {
        const double a = m[2 * p], b = m[2 * p + 1];
        m[2 * p] = r00 * a + r01 * b;
        m[2 * p + 1] = r10 * a + r11 * b;
    }
    m[0] += tx; m[1] += ty;
#pragma unroll
    for (int p = 0; p < 4; ++p)            // rows: (R8 P)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const double a = P[(2 * p) * 8 + j], b = P[(2 * p + 1) * 8 + j];
            P[(2 * p) * 8 + j] = r00 * a + r01 * b;
            P[(2 * p + 1) * 8 + j] = r10 * a + r11 * b;
        }
#pragma unroll
    for (int i = 0; i < 8; ++i)            // columns: (.) R8^T
#pragma unroll
        for (int q = 0; q < 4; ++q) {
            const double a = P[i * 8 + 2 * q], b = P[i * 8 + 2 * q + 1];
            P[i * 8 + 2 * q] = a * r00 + b * r01;
            P[i * 8 + 2 * q + 1] = a * r10 + b * r11;
        }
}

}  // namespace kf

// ---- eight lanes per track -------------------------------------------------------------------------
// Lane i of an aligned 8-lane group owns row i of the 8x8 covariance (8 doubles) and element i of
// the mean: a track's 64+8 doubles are one contiguous, fully coalesced HBM row, the per-lane
// register footprint is ~1/8 of the one-thread-per-track form, and the dependent float64 chains
// (Cholesky, solves, K S K^T) are ~8x shorter.  Cross-row terms travel by __shfl inside the group.
// The arithmetic is the same sequence of operations as kf::predict / kf::update above (so the
// results are bit-identical to them).
namespace kf8 {

// (On the CPU -- tests/host_emul/kf_host.hip -- grp() is served by a hook that runs the eight lanes of a group in lockstep by
//  replay, so that these very functions can be held against the reference's golden vectors without a GPU.  The device pass
//  sees exactly the code it always did.)
#if defined(__HIP_DEVICE_COMPILE__) || defined(BTS_HOST_EMU)
__device__ __forceinline__ double grp(double v, int src) { return __shfl(v, src, 8); }
#else
double kf8_host_grp(double v, int src);      // defined by the CPU test harness only
inline double grp(double v, int src) { return kf8_host_grp(v, src); }
#endif

__host__ __device__ __forceinline__ void predict(double &mi, double Pr[8], int i, bool f32) {
    const double w = grp(mi, 2), h = grp(mi, 3);
    const double wh = (i & 1) ? h : w;                         // row i's noise uses w for even, h for odd i
    const double q = kf::sq_scaled(i < 4 ? kf::W_POS : kf::W_VEL, wh, f32);
    const double up = grp(mi, (i + 4) & 7);
    if (i < 4) mi = mi + up;
#pragma unroll
    for (int j = 0; j < 8; ++j) {                              // left = F P : rows 0..3 += rows 4..7
        const double lower = grp(Pr[j], (i + 4) & 7);
        if (i < 4) Pr[j] = Pr[j] + lower;
    }
#pragma unroll
    for (int j = 0; j < 4; ++j) Pr[j] = Pr[j] + Pr[j + 4];     // left F^T : cols 0..3 += cols 4..7
#pragma unroll
    for (int j = 0; j < 8; ++j) if (j == i) Pr[j] += q;
}

__host__ __device__ __forceinline__ void update(double &mi, double Pr[8], int i, const float z[4], bool f32) {
    double S[16], L[16];
#pragma unroll
    for (int a = 0; a < 4; ++a)
#pragma unroll
        for (int b = 0; b < 4; ++b) S[a * 4 + b] = grp(Pr[b], a);
    const double w = grp(mi, 2), h = grp(mi, 3);
    const double r_w = kf::sq_scaled(kf::W_POS, w, f32), r_h = kf::sq_scaled(kf::W_POS, h, f32);
    S[0] += r_w; S[5] += r_h; S[10] += r_w; S[15] += r_h;
    kf::chol4(S, L);
    double K[4] = {Pr[0], Pr[1], Pr[2], Pr[3]};                // row i of P H^T -> row i of the gain
    kf::chol4_solve(L, K);
    double y[4];
#pragma unroll
    for (int a = 0; a < 4; ++a) y[a] = (double)z[a] - grp(mi, a);
    double s = 0.0;
#pragma unroll
    for (int a = 0; a < 4; ++a) s += y[a] * K[a];
    mi = mi + s;
    double KS[4];
#pragma unroll
    for (int b = 0; b < 4; ++b) {
        double t = 0.0;
#pragma unroll
        for (int a = 0; a < 4; ++a) t += K[a] * S[a * 4 + b];
        KS[b] = t;
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        double t = 0.0;
#pragma unroll
        for (int b = 0; b < 4; ++b) t += KS[b] * grp(K[b], j);
        Pr[j] = Pr[j] - t;
    }
}

// STrack.tlbr from the group's mean; every lane gets the box
__host__ __device__ __forceinline__ void tlbr(double mi, bool f32, double out[4]) {
    double m[8];
#pragma unroll
    for (int c = 0; c < 4; ++c) m[c] = grp(mi, c);
    kf::mean_to_tlbr(m, f32, out);
}

}  // namespace kf8
