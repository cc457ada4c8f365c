// slam_oracle_ukf.cpp — CPU ORACLE for the UKF-SLAM predict–update path.  TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Restates ekf_ws/src/localization_pkg/src/ukf.cpp:3-45 (ctor/init), :106-123 (nearestSPD), :125-159 (motion /
// sensing models), :161-372 (update, predictionStage, updateStage, landmarkUpdate, landmarkInsertion) and
// filter.h:207 (W_0 = 0.2f), statement by statement, with the reference's float truncations and its quirks
// (signed process noise Q, z_est(1) never accumulated, sensingModel yaw taken from x_t,
// sigma points not redrawn between landmark updates, plain weighted mean of (cos, sin), updates before insertions).
//
// PARITY STATUS: "parity unpinned" against the reference binary (unbuildable here, see slam_oracle.cpp).  Pinned by
// the UKF known-answer table of SURVEY.md Appendix E (1e-6 absolute) and by an independent numpy transliteration
// that uses LAPACK's eigh for the matrix square root (tests/test_oracle_ukf.py).
//
// Two places where the reference delegates to Eigen routines whose rounding cannot be reproduced:
//  * SelfAdjointEigenSolver + MatrixFunctions .sqrt() (ukf.cpp:116-122,208).  Mathematically sqtP = Qv sqrt(D+) Qv^T
//    is unique; here it is computed with a cyclic Jacobi eigen-iteration in PARALLEL ORDER (n / 2 disjoint rotations per round;
//    jacobi_pair below: the circle method over the indices, or over blocks of two when n is divisible by four) on the
//    exactly-symmetric matrix, the same schedule the GPU kernels run, so GPU == oracle bit for bit.  From the second
//    timestep on the iteration is WARM-STARTED in the previous step's eigenbasis (B = V0^T (A V0), V starts at V0,
//    small-element rule from the first sweep; cold start every 100 steps) — an implementation choice shared with the
//    kernel that halves the sweeps; the eigen-decomposition it converges to is the same mathematical object.
//  * unqualified cos/sin on a float argument (ukf.cpp:39-42,129-133,183-186,358-359): float overload
//    (cfg.ukf_float_trig = 1, default) or double function (0); SURVEY.md Appendix B.
#include <atomic>
#include <chrono>
#include <thread>

#include "oracle_common.h"

namespace {
using namespace orc;

// ---- symmetric eigen-decomposition: cyclic Jacobi, parallel ordering (jacobi_pair) -----------------------------
// A: n x n row-major, exactly symmetric on entry and kept so (only pair-blocks i >= j are computed, then mirrored).
// V: n x n, columns = eigenvectors.  n is even (n = 4 + 2M).  Returns the number of sweeps, or -1 if not converged.
// All rotations of a round read the matrix as it was at the start of the round (disjoint index pairs).
// warm = false: V starts as the identity and the small-element rule applies from the fourth sweep on (classical).
// warm = true : the caller passes A = V0^T A0 V0 (nearly diagonal) and V = V0, the eigenvectors of the previous
//               timestep; the small-element rule applies from the first sweep (the off-diagonal part is a small
//               perturbation from the start), and a few sweeps instead of 8-10 reach convergence.
constexpr int kUkfWarmMaxAge = 100;   // consecutive warm starts before a cold one (bounds the loss of orthogonality in V)
// refmode = true: REFERENCE-ORDER arithmetic, independent of the device's instruction selection - no fused
// multiply-adds (the reference's catkin build has no FMA contraction: -std=c++17 only => SSE2), the textbook rotation parameters
// tau = (a_qq - a_pp) / (2 a_pq), t = sign(tau) / (|tau| + sqrt(tau^2 + 1)), c = 1 / sqrt(t^2 + 1), s = t c.  The default mode
// (false) evaluates the same mathematics the way the kernel does (fused products, tau-free parameters) so that GPU == oracle
// is a bit-exact statement; tests/test_oracle_ukf.py bounds the difference between the two over long trajectories.
// The schedule: which index pairs (p < q) rotate together in round t = 0 .. n - 2 of a sweep (pair k = 0 .. n / 2 - 1; the pairs of a
// round are disjoint, a sweep visits every pair once).  An implementation choice shared with the kernels, like the Jacobi iteration itself.
//   rr_pair      the circle method over the n indices (position 0 fixed, the others move one slot per round).
//   jacobi_pair  n divisible by four: the circle method over the n / 2 BLOCKS of two consecutive indices.  A block round T pairs the
//                blocks into quadruples (a, b | c, d) and takes two rounds: (a, c) (b, d), then (a, d) (b, c); round 0 of the sweep
//                rotates inside the blocks, (a, b) (c, d), indexed by the quadruples of block round 0.  Two consecutive rounds then
//                stay inside the same 4 x 4 blocks of the matrix and the same four columns of V, which the kernel keeps in registers
//                across both (half the passes over LDS).  Pair 2 kb + u belongs to quadruple kb.
//                n = 2 (mod 4): rr_pair.
static inline void rr_pair(int k, int t, int n, int& p, int& q) {
    auto at = [&](int kk) { return kk == 0 ? 0 : 1 + ((kk - 1 + t) % (n - 1)); };
    const int a = at(k), b = at(n - 1 - k);
    p = std::min(a, b); q = std::max(a, b);
}
static inline void jacobi_pair(int k, int t, int n, int& p, int& q) {
    if (n & 2) { rr_pair(k, t, n, p, q); return; }
    const int kb = k >> 1, u = k & 1;
    int X, Y;
    rr_pair(kb, t == 0 ? 0 : (t - 1) >> 1, n / 2, X, Y);
    if (t == 0) { p = 2 * (u ? Y : X); q = p + 1; return; }
    const int s = (t - 1) & 1;
    p = 2 * X + u;
    q = 2 * Y + (s ? 1 - u : u);
}
static inline double mul_add(bool refmode, double a, double b, double c) { return refmode ? a * b + c : std::fma(a, b, c); }
int jacobi_round_robin(double* A, double* V, int n, int max_sweeps, bool warm = false, bool refmode = false) {
    const int m = n / 2;
    std::vector<int> pp(m), qq(m);
    std::vector<double> cs(m), sn(m), tn(m);
    std::vector<int> zr(m);
    const int tiny_from = warm ? 0 : 3;
    if (!warm)
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < n; ++j) V[(size_t)i * n + j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < max_sweeps; ++sweep) {
        // convergence: every off-diagonal element is exactly zero (the small-element rule below makes that reachable)
        double off = 0.0;
        for (int i = 0; i < n; ++i)
            for (int j = 0; j < i; ++j) off = std::max(off, fabs(A[(size_t)i * n + j]));
        if (off == 0.0) return sweep;
        for (int t = 0; t < n - 1; ++t) {
            for (int k = 0; k < m; ++k) {
                jacobi_pair(k, t, n, pp[k], qq[k]);
                const double app = A[(size_t)pp[k] * n + pp[k]], aqq = A[(size_t)qq[k] * n + qq[k]], apq = A[(size_t)qq[k] * n + pp[k]];
                double c = 1.0, s = 0.0, tt = 0.0;
                // small-element rule (classical Jacobi): after three sweeps an off-diagonal element that cannot change
                // either diagonal neighbour in fp64 is set to zero instead of being rotated away
                const double g = 100.0 * fabs(apq);
                const bool tiny = sweep >= tiny_from && (fabs(app) + g == fabs(app)) && (fabs(aqq) + g == fabs(aqq));
                zr[k] = tiny ? 1 : 0;
                if (apq != 0.0 && !tiny) {
                    // t = tan(theta): the smaller root of t^2 + 2 tau t - 1 = 0, tau = (a_qq - a_pp) / (2 a_pq), in the form
                    // d = a_qq - a_pp, h = hypot(d, 2 a_pq), w = |d| + h: t = 2 a_pq / (+-w), c = sqrt(w / (2 h)), s = t c
                    // (three dependent sqrt / div instead of five; the device's critical path per round).  Products that feed
                    // an addition are fused (std::fma) here and in the rotations below, as the kernel evaluates them.
                    const double d = aqq - app, b2 = 2.0 * apq;
                    const double h = refmode ? sqrt(d * d + b2 * b2) : sqrt(std::fma(d, d, b2 * b2));
                    if (refmode && h > 0.0) {
                        const double tau = d / b2;
                        tt = (tau >= 0.0 ? 1.0 : -1.0) / (fabs(tau) + sqrt(tau * tau + 1.0));
                        c = 1.0 / sqrt(tt * tt + 1.0);
                        s = tt * c;
                    } else if (h > 0.0) {   // h == 0: d and a_pq below 1e-154, nothing to rotate (the element is zeroed)
                        const double w = fabs(d) + h;
                        // sign of t = sign of tau = d / (2 a_pq), +1 at d = 0 exactly (equal diagonal entries are common: every
                        // landmark enters P with the same W block; letting the sign follow a_pq there made clusters of equal
                        // eigenvalues cycle at the rounding level instead of settling)
                        const bool pos = (d == 0.0) || ((d > 0.0) == (b2 > 0.0));
                        tt = (pos ? fabs(b2) : -fabs(b2)) / w;
                        c = sqrt(w / (2.0 * h));
                        s = tt * c;
                    }
                }
                cs[k] = c; sn[k] = s; tn[k] = tt;
            }
            // pair-blocks (i, j), i >= j:  B' = R_i^T B R_j  with R = [[c, s], [-s, c]]
            for (int i = 0; i < m; ++i) {
                const int pi = pp[i], qi = qq[i];
                const double ci = cs[i], si = sn[i];
                for (int j = 0; j < i; ++j) {
                    const int pj = pp[j], qj = qq[j];
                    const double cj = cs[j], sj = sn[j];
                    const double b00 = A[(size_t)pi * n + pj], b01 = A[(size_t)pi * n + qj];
                    const double b10 = A[(size_t)qi * n + pj], b11 = A[(size_t)qi * n + qj];
                    const double t00 = mul_add(refmode, ci, b00, -(si * b10)), t01 = mul_add(refmode, ci, b01, -(si * b11));
                    const double t10 = mul_add(refmode, si, b00, ci * b10), t11 = mul_add(refmode, si, b01, ci * b11);
                    const double r00 = mul_add(refmode, t00, cj, -(t01 * sj)), r01 = mul_add(refmode, t00, sj, t01 * cj);
                    const double r10 = mul_add(refmode, t10, cj, -(t11 * sj)), r11 = mul_add(refmode, t10, sj, t11 * cj);
                    A[(size_t)pi * n + pj] = r00; A[(size_t)pj * n + pi] = r00;
                    A[(size_t)pi * n + qj] = r01; A[(size_t)qj * n + pi] = r01;
                    A[(size_t)qi * n + pj] = r10; A[(size_t)pj * n + qi] = r10;
                    A[(size_t)qi * n + qj] = r11; A[(size_t)qj * n + qi] = r11;
                }
            }
            for (int i = 0; i < m; ++i) {  // diagonal blocks
                const int p = pp[i], q = qq[i];
                const double app = A[(size_t)p * n + p], aqq = A[(size_t)q * n + q], apq = A[(size_t)q * n + p];
                A[(size_t)p * n + p] = mul_add(refmode, -tn[i], apq, app);
                A[(size_t)q * n + q] = mul_add(refmode, tn[i], apq, aqq);
                if (apq != 0.0) { A[(size_t)q * n + p] = 0.0; A[(size_t)p * n + q] = 0.0; }   // rotated away, or tiny (zr[i])
            }
            for (int i = 0; i < m; ++i) {  // V <- V J
                const int p = pp[i], q = qq[i];
                const double c = cs[i], s = sn[i];
                for (int k = 0; k < n; ++k) {
                    const double vp = V[(size_t)k * n + p], vq = V[(size_t)k * n + q];
                    V[(size_t)k * n + p] = mul_add(refmode, c, vp, -(s * vq));
                    V[(size_t)k * n + q] = mul_add(refmode, s, vp, c * vq);
                }
            }
        }
    }
    return -1;
}

// State sizes n = 2 (mod 4) in the LDS size classes of the kernels (L_max <= 50): the matrix is PADDED by a decoupled 2 x 2 zero block to
// n + 2 (V by the identity), so that every size walks the schedule over quadruples; the rotations with the two extra indices are the
// identity (a_pq = 0) and the sweep has two more rounds, in which their partners rest.  The kernels keep the two zero rows physically
// (ukf_kernel.hip), the arithmetic on the n x n part is the same.  pad = false (the HBM-streamed class, ukf_big_kernel.hip, whose full
// square matrices have no room for the extra rows): the circle method over the n indices, as before.
int jacobi_parallel(double* A, double* V, int n, int max_sweeps, bool warm, bool refmode, bool pad) {
    if (!(n & 2) || !pad) return jacobi_round_robin(A, V, n, max_sweeps, warm, refmode);
    const int np = n + 2;
    std::vector<double> Ap((size_t)np * np, 0.0), Vp((size_t)np * np, 0.0);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) {
            Ap[(size_t)r * np + c] = A[(size_t)r * n + c];
            if (warm) Vp[(size_t)r * np + c] = V[(size_t)r * n + c];
        }
    if (warm) { Vp[(size_t)n * np + n] = 1.0; Vp[(size_t)(n + 1) * np + n + 1] = 1.0; }
    const int sweeps = jacobi_round_robin(Ap.data(), Vp.data(), np, max_sweeps, warm, refmode);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) { A[(size_t)r * n + c] = Ap[(size_t)r * np + c]; V[(size_t)r * n + c] = Vp[(size_t)r * np + c]; }
    return sweeps;
}

struct Ukf {
    slam_config cfg;
    FilterNoise nz;
    int L_max, math;
    int timestep = 0, M = 0, flags = 0;
    std::vector<double> x_t, x_pred, P_t, P_pred, sqtP, Xp4;  // Xp4: rows 0..3 of X_pred, [4][2n+1]
    std::vector<int> ids;
    int n_sq = 0;          // dimension sqtP currently has
    int last_sweeps = 0;
    std::vector<double> Vprev;   // eigenvectors of the previous nearestSPD matrix (n_sq x n_sq), warm start of the next
    int v_age = -1;              // consecutive warm starts so far; -1 = no usable Vprev
    bool loc = false;      // FilterChoice::UKF_LOC: localisation only, landmarks come from the known map (ukf.cpp:146-154)
    std::vector<float> mapf;   // `filter->map`: [id, x, y] float32 triplets (localization_node.cpp:152-156)

    int n() const { return 4 + 2 * M; }
    static constexpr float W_0 = 0.2f;  // filter.h:207

    Ukf(const slam_config& c, int Lm, int mth) : cfg(c), nz(effective_noise(c)), L_max(Lm), math(mth) { reset_ctor(); }

    template <class MP> float fcos(float a) const {  // unqualified cos(float): float overload or double function
        double s, c; MP::sincos((double)a, &s, &c);
        return (float)c;
    }
    template <class MP> double tcos(float a) const { double s, c; MP::sincos((double)a, &s, &c); return cfg.ukf_float_trig ? (double)(float)c : c; }
    template <class MP> double tsin(float a) const { double s, c; MP::sincos((double)a, &s, &c); return cfg.ukf_float_trig ? (double)(float)s : s; }

    void reset_ctor() {  // UKF::UKF ukf.cpp:3-23
        timestep = 0; M = 0; flags = 0; ids.clear();
        x_t.assign(4, 0.0); x_pred.assign(4, 0.0);
        P_t.assign(16, 0.0);
        P_t[0] = 0.01 * 0.01; P_t[5] = 0.01 * 0.01; P_t[10] = 0.005 * 0.005; P_t[15] = 0.005 * 0.005;
        P_pred = P_t;
        sqtP.clear(); n_sq = 0;
        Vprev.clear(); v_age = -1;
    }
    template <class MP> void init_t(float x0, float y0, float yaw0) {  // ukf.cpp:31-45
        reset_ctor();
        x_t[0] = x0; x_t[1] = y0; x_t[2] = tcos<MP>(yaw0); x_t[3] = tsin<MP>(yaw0);
    }
    void init(float x0, float y0, float yaw0) { math == MATH_DET ? init_t<DetMath>(x0, y0, yaw0) : init_t<LibmMath>(x0, y0, yaw0); }

    template <class MP> float yaw_of(const double* v) const {  // (float) remainder(atan2(v[3], v[2]), 2*pi)
        return (float)remainder(MP::atan2(v[3], v[2]), slam::kTwoPi);
    }

    // weight of sigma point i (ukf.cpp:174-176): (1-W_0)/(2n) evaluated in float, Wts(0) = W_0
    double wt(int i, int nn) const { return i == 0 ? (double)W_0 : (double)((1 - W_0) / (2 * nn)); }

    template <class MP> void prediction_stage(float u_d, float u_th) {
        const int nn = n();
        // ---- nearestSPD (ukf.cpp:106-123) + matrix square root (ukf.cpp:208) ----
        std::vector<double> Y((size_t)nn * nn), V((size_t)nn * nn);
        const float scale_f = (2 * M + 4) / (1 - W_0);   // int / float -> float   ukf.cpp:114
        const double scale = (double)scale_f;
        for (int r = 0; r < nn; ++r)
            for (int c = 0; c < nn; ++c) Y[(size_t)r * nn + c] = (0.5 * (P_t[(size_t)r * nn + c] + P_t[(size_t)c * nn + r])) * scale;
        // Warm start (an implementation choice of this build, not of the reference: Eigen's solver has no such notion;
        // the mathematical result Qv sqrt(D+) Qv^T is the same): rotate into the previous step's eigenbasis, extended by
        // the identity for landmarks inserted since.  T = Y V0, B = V0^T T, sums in ascending k.
        // reference-order mode (cfg.reserved[1] = 1): cold start every step, like Eigen's solver, and unfused arithmetic throughout
        const bool refmode = cfg.reserved[1] == 1;
        const bool warm = !refmode && v_age >= 0 && v_age < kUkfWarmMaxAge && n_sq > 0 && n_sq <= nn;
        if (warm) {
            for (int r = 0; r < nn; ++r)
                for (int c = 0; c < nn; ++c) V[(size_t)r * nn + c] = (r < n_sq && c < n_sq) ? Vprev[(size_t)r * n_sq + c] : (r == c ? 1.0 : 0.0);
            std::vector<double> T((size_t)nn * nn);
            for (int r = 0; r < nn; ++r)
                for (int c = 0; c < nn; ++c) {
                    double acc = 0.0;
                    // each term fused, ascending k: the device forms both products of the warm start with v_mfma_f64_16x16x4_f64
                    for (int k = 0; k < nn; ++k) acc = std::fma(Y[(size_t)r * nn + k], V[(size_t)k * nn + c], acc);
                    T[(size_t)r * nn + c] = acc;
                }
            for (int r = 0; r < nn; ++r)
                for (int c = 0; c <= r; ++c) {
                    double acc = 0.0;
                    for (int k = 0; k < nn; ++k) acc = std::fma(V[(size_t)k * nn + r], T[(size_t)k * nn + c], acc);
                    Y[(size_t)r * nn + c] = acc; Y[(size_t)c * nn + r] = acc;
                }
        }
        const int sweeps = jacobi_parallel(Y.data(), V.data(), nn, 60, warm, refmode, 4 + 2 * L_max <= 104);
        last_sweeps = sweeps;
        if (sweeps >= 0) { Vprev = V; v_age = warm ? v_age + 1 : 0; } else { v_age = -1; }
        if (sweeps < 0) {
            // ukf.cpp:209-211: the exception is swallowed and the stale sqtP is reused.  A stale matrix of the wrong
            // size cannot be used: keep the instance running with a zero spread and flag it.
            flags |= SLAM_INST_SQRT_FAILED;
            if (n_sq != nn) { sqtP.assign((size_t)nn * nn, 0.0); n_sq = nn; }
        } else {
            std::vector<double> sd(nn);
            for (int k = 0; k < nn; ++k) {
                const double d = Y[(size_t)k * nn + k];
                sd[k] = sqrt(d > 0.00000001 ? d : 0.00000001);   // cwiseMax(1e-8) then principal sqrt
            }
            sqtP.assign((size_t)nn * nn, 0.0); n_sq = nn;
            for (int r = 0; r < nn; ++r)
                for (int c = 0; c <= r; ++c) {
                    double acc = 0.0;
                    // (each term fused, ascending k: the kernels form this product on v_mfma_f64_16x16x4_f64 - refmode keeps the plain sum)
                    for (int k = 0; k < nn; ++k) acc = mul_add(refmode, V[(size_t)r * nn + k] * sd[k], V[(size_t)c * nn + k], acc);
                    sqtP[(size_t)r * nn + c] = acc; sqtP[(size_t)c * nn + r] = acc;
                }
        }
        // ---- sigma points through the motion model (ukf.cpp:214-226, 125-135); rows >= 4 are copied ----
        const int ns = 2 * nn + 1;
        Xp4.assign((size_t)4 * ns, 0.0);
        const float dd = u_d + cfg.v_d;
        for (int i = 0; i < ns; ++i) {
            double v[4];
            for (int r = 0; r < 4; ++r) {
                if (i == 0) v[r] = x_t[r];
                else if (i <= nn) v[r] = x_t[r] + sqtP[(size_t)r * nn + (i - 1)];
                else v[r] = x_t[r] - sqtP[(size_t)r * nn + (i - 1 - nn)];
            }
            const float yaw = yaw_of<MP>(v);
            if (cfg.ukf_float_trig) {
                Xp4[(size_t)0 * ns + i] = v[0] + (double)(dd * (float)tcos<MP>(yaw));   // float * float
                Xp4[(size_t)1 * ns + i] = v[1] + (double)(dd * (float)tsin<MP>(yaw));
            } else {
                Xp4[(size_t)0 * ns + i] = v[0] + (double)dd * tcos<MP>(yaw);
                Xp4[(size_t)1 * ns + i] = v[1] + (double)dd * tsin<MP>(yaw);
            }
            const float new_yaw = (float)remainder((double)(yaw + u_th + cfg.v_th), slam::kTwoPi);  // float adds
            Xp4[(size_t)2 * ns + i] = tcos<MP>(new_yaw);
            Xp4[(size_t)3 * ns + i] = tsin<MP>(new_yaw);
        }
        // ---- weighted mean (ukf.cpp:228-232) and covariance (ukf.cpp:235-240), sequential in i ----
        x_pred.assign(nn, 0.0);
        for (int r = 0; r < nn; ++r) {
            double acc = 0.0;
            for (int i = 0; i < ns; ++i) acc = acc + wt(i, nn) * xpred_elem(r, i, nn);
            x_pred[r] = acc;
        }
        P_pred.assign((size_t)nn * nn, 0.0);
        std::vector<double> D((size_t)nn * ns);
        for (int r = 0; r < nn; ++r)
            for (int i = 0; i < ns; ++i) D[(size_t)r * ns + i] = xpred_elem(r, i, nn) - x_pred[r];
        for (int r = 0; r < nn; ++r)
            for (int c = 0; c < nn; ++c) {
                double acc = 0.0;
                // each term fused: acc = fma(w_i d_r, d_c, acc).  The reference leaves the contraction of `P += (w d) d^T` to Eigen and
                // the compiler (ukf.cpp:235-238); the device evaluates it with v_mfma_f64_16x16x4_f64, whose result is this
                // chain in ascending i, bit for bit (tools/ubench_mfma_f64.hip).
                for (int i = 0; i < ns; ++i) acc = mul_add(refmode, wt(i, nn) * D[(size_t)r * ns + i], D[(size_t)c * ns + i], acc);
                P_pred[(size_t)r * nn + c] = acc;
            }
        // + Q (ukf.cpp:182-186, 240): signed diagonal from the yaw of x_t
        const float yaw = yaw_of<MP>(x_t.data());
        P_pred[0] = P_pred[0] + nz.V00 * tcos<MP>(yaw);
        P_pred[(size_t)1 * nn + 1] = P_pred[(size_t)1 * nn + 1] + nz.V00 * tsin<MP>(yaw);
        P_pred[(size_t)2 * nn + 2] = P_pred[(size_t)2 * nn + 2] + nz.V11 * tcos<MP>(yaw);
        P_pred[(size_t)3 * nn + 3] = P_pred[(size_t)3 * nn + 3] + nz.V11 * tsin<MP>(yaw);
    }

    // X_pred(r, i): rows 0..3 from the motion model, rows >= 4 = the sigma point itself
    double xpred_elem(int r, int i, int nn) const {
        const int ns = 2 * nn + 1;
        if (r < 4) return Xp4[(size_t)r * ns + i];
        if (i == 0) return x_t[r];
        if (i <= nn) return x_t[r] + sqtP[(size_t)r * nn + (i - 1)];
        return x_t[r] - sqtP[(size_t)r * nn + (i - 1 - nn)];
    }

    // j: landmark slot (SLAM) or the landmark's id = its row in the known map (LOC, ukf.cpp:300-302)
    template <class MP> void landmark_update(int j, float r_m, float b_m) {  // ukf.cpp:293-349
        const int nn = n(), ns = 2 * nn + 1, li = 2 * j + 4;
        const double mx = loc ? (double)mapf[(size_t)j * 3 + 1] : 0.0, my = loc ? (double)mapf[(size_t)j * 3 + 2] : 0.0;
        const float yaw = yaw_of<MP>(x_t.data());  // sensingModel takes yaw from x_t (ukf.cpp:139)
        std::vector<double> Z0(ns), Z1(ns);
        for (int i = 0; i < ns; ++i) {
            const double dx = (loc ? mx : xpred_elem(li, i, nn)) - xpred_elem(0, i, nn), dy = (loc ? my : xpred_elem(li + 1, i, nn)) - xpred_elem(1, i, nn);
            Z0[i] = ::sqrt(MP::sq(dx) + MP::sq(dy)) + (double)cfg.w_r;
            double yaw_i = (double)yaw;
            if (cfg.ukf_sensing_yaw_from_sigma) {   // quirk D-9 off: the yaw of the sigma point the model is evaluated at
                const double v[4] = {0.0, 0.0, xpred_elem(2, i, nn), xpred_elem(3, i, nn)};
                yaw_i = (double)yaw_of<MP>(v);
            }
            Z1[i] = remainder((MP::atan2(dy, dx) - yaw_i) + (double)cfg.w_b, slam::kTwoPi);
        }
        double z0 = 0.0, z1 = 0.0;
        for (int i = 0; i < ns; ++i) z0 = z0 + wt(i, nn) * Z0[i];   // z_est(1) is never accumulated (ukf.cpp:310-314) ...
        if (cfg.ukf_accumulate_zest1)                               // ... unless quirk D-8 is switched off
            for (int i = 0; i < ns; ++i) z1 = z1 + wt(i, nn) * Z1[i];
        double S[4] = {0, 0, 0, 0};
        for (int i = 0; i < ns; ++i) {
            const double d0 = Z0[i] - z0, d1 = remainder(Z1[i] - z1, slam::kTwoPi);
            const double w0 = wt(i, nn) * d0, w1 = wt(i, nn) * d1;
            S[0] = S[0] + w0 * d0; S[1] = S[1] + w0 * d1; S[2] = S[2] + w1 * d0; S[3] = S[3] + w1 * d1;
        }
        S[0] = S[0] + nz.W00; S[1] = S[1] + 0.0; S[2] = S[2] + 0.0; S[3] = S[3] + nz.W11;
        std::vector<double> C((size_t)nn * 2);
        for (int r = 0; r < nn; ++r) {
            double c0 = 0.0, c1 = 0.0;
            for (int i = 0; i < ns; ++i) {
                const double wd = wt(i, nn) * (xpred_elem(r, i, nn) - x_pred[r]);
                const double d0 = Z0[i] - z0, d1 = remainder(Z1[i] - z1, slam::kTwoPi);
                c0 = c0 + wd * d0; c1 = c1 + wd * d1;
            }
            C[(size_t)2 * r] = c0; C[(size_t)2 * r + 1] = c1;
        }
        double Si[4];
        if (!inv2x2_lu(S, Si)) flags |= SLAM_INST_S_SINGULAR;
        std::vector<double> K((size_t)nn * 2), KS((size_t)nn * 2);
        const double i0 = (double)r_m - z0, i1 = remainder((double)b_m - z1, slam::kTwoPi);
        for (int r = 0; r < nn; ++r) {
            K[(size_t)2 * r] = C[(size_t)2 * r] * Si[0] + C[(size_t)2 * r + 1] * Si[2];
            K[(size_t)2 * r + 1] = C[(size_t)2 * r] * Si[1] + C[(size_t)2 * r + 1] * Si[3];
        }
        for (int r = 0; r < nn; ++r) {
            x_pred[r] = x_pred[r] + (K[(size_t)2 * r] * i0 + K[(size_t)2 * r + 1] * i1);
            KS[(size_t)2 * r] = K[(size_t)2 * r] * S[0] + K[(size_t)2 * r + 1] * S[2];
            KS[(size_t)2 * r + 1] = K[(size_t)2 * r] * S[1] + K[(size_t)2 * r + 1] * S[3];
        }
        for (int r = 0; r < nn; ++r)
            for (int c = 0; c < nn; ++c)
                P_pred[(size_t)r * nn + c] = P_pred[(size_t)r * nn + c] - (KS[(size_t)2 * r] * K[(size_t)2 * c] + KS[(size_t)2 * r + 1] * K[(size_t)2 * c + 1]);
    }

    template <class MP> void landmark_insertion(int id, float r_m, float b_m) {  // ukf.cpp:351-372
        const int nn = n();
        const float yaw = yaw_of<MP>(x_pred.data());
        const float ang = yaw + b_m;
        x_pred.resize(nn + 2);
        if (cfg.ukf_float_trig) {
            x_pred[nn] = x_pred[0] + (double)(r_m * (float)tcos<MP>(ang));
            x_pred[nn + 1] = x_pred[1] + (double)(r_m * (float)tsin<MP>(ang));
        } else {
            x_pred[nn] = x_pred[0] + (double)r_m * tcos<MP>(ang);
            x_pred[nn + 1] = x_pred[1] + (double)r_m * tsin<MP>(ang);
        }
        ids.push_back(id);
        std::vector<double> Pn((size_t)(nn + 2) * (nn + 2), 0.0);
        for (int r = 0; r < nn; ++r)
            for (int c = 0; c < nn; ++c) Pn[(size_t)r * (nn + 2) + c] = P_pred[(size_t)r * nn + c];
        Pn[(size_t)nn * (nn + 2) + nn] = nz.W00;
        Pn[(size_t)(nn + 1) * (nn + 2) + nn + 1] = nz.W11;
        P_pred.swap(Pn);
        M += 1;
    }

    template <class MP> int update_t(float fwd, float ang, const float* meas, int k) {
        timestep += 1;                                    // ukf.cpp:164
        prediction_stage<MP>(fwd, ang);                   // ukf.cpp:189
        std::vector<int> fresh;                           // ukf.cpp:251-287: updates first, insertions last
        for (int l = 0; l < k; ++l) {
            const int id = (int)meas[3 * l];
            int j = -1;
            for (int q = 0; q < M; ++q)
                if (ids[q] == id) { j = q; break; }
            if (loc) {   // ukf.cpp:272-276: every detection is an update against the known map
                if (id >= 0 && (size_t)id * 3 + 2 < mapf.size()) landmark_update<MP>(id, meas[3 * l + 1], meas[3 * l + 2]);
                else flags |= SLAM_INST_INDEX_OOR;   // std::vector out of range: undefined behaviour in the reference
                continue;
            }
            if (j < 0) fresh.push_back(l);
            else landmark_update<MP>(j, meas[3 * l + 1], meas[3 * l + 2]);
        }
        for (int l : fresh) {
            if (M >= L_max) { flags |= SLAM_INST_CAPACITY; continue; }
            landmark_insertion<MP>((int)meas[3 * l], meas[3 * l + 1], meas[3 * l + 2]);
        }
        x_t = x_pred; P_t = P_pred;                       // ukf.cpp:289-290
        bool fin = true;
        for (double v : x_t) fin = fin && std::isfinite(v);
        for (double v : P_t) fin = fin && std::isfinite(v);
        if (!fin) flags |= SLAM_INST_NONFINITE;
        return flags;
    }
    int update(float fwd, float ang, const float* meas, int k) {
        return math == MATH_DET ? update_t<DetMath>(fwd, ang, meas, k) : update_t<LibmMath>(fwd, ang, meas, k);
    }
    double yaw_est() const { return remainder(math == MATH_DET ? DetMath::atan2(x_t[3], x_t[2]) : LibmMath::atan2(x_t[3], x_t[2]), slam::kTwoPi); }
};

}  // namespace

extern "C" {

void* orc_ukf_create(const slam_config* cfg, int L_max, int math) { return new Ukf(*cfg, L_max, math); }
// switch to UKF_LOC with the true map [L][2] (the wire format is float32 [id, x, y])
void orc_ukf_set_loc_map(void* h, const double* map_xy, int L) {
    Ukf* u = (Ukf*)h;
    u->loc = true;
    u->mapf.resize((size_t)3 * L);
    for (int i = 0; i < L; ++i) { u->mapf[3 * i] = (float)i; u->mapf[3 * i + 1] = (float)map_xy[2 * i]; u->mapf[3 * i + 2] = (float)map_xy[2 * i + 1]; }
}
void orc_ukf_destroy(void* h) { delete (Ukf*)h; }
void orc_ukf_init(void* h, float x0, float y0, float yaw0) { ((Ukf*)h)->init(x0, y0, yaw0); }
int orc_ukf_update(void* h, float fwd, float ang, const float* meas, int k) { return ((Ukf*)h)->update(fwd, ang, meas, k); }
void orc_ukf_get(void* h, double* x, double* P, int* M, int* ids, int* timestep, int* sweeps) {
    Ukf* u = (Ukf*)h;
    const int n = u->n();
    if (x) memcpy(x, u->x_t.data(), sizeof(double) * n);
    if (P) memcpy(P, u->P_t.data(), sizeof(double) * n * n);
    if (M) *M = u->M;
    if (ids) memcpy(ids, u->ids.data(), sizeof(int) * u->M);
    if (timestep) *timestep = u->timestep;
    if (sweeps) *sweeps = u->last_sweeps;
}
// matrix square root probe: sqtP of nearestSPD(scale * P) for a given symmetric-ish P (n x n row-major)
// the Jacobi schedule (tests: every pair once per sweep, disjoint rounds, the same pairs as the kernels' jacobi_schedule.h)
void orc_ukf_jacobi_pair(int k, int t, int n, int* p, int* q) { jacobi_pair(k, t, n, *p, *q); }

int orc_ukf_sqrt_probe(const double* P, int n, double scale, double* out) {
    std::vector<double> Y((size_t)n * n), V((size_t)n * n);
    for (int r = 0; r < n; ++r)
        for (int c = 0; c < n; ++c) Y[(size_t)r * n + c] = (0.5 * (P[(size_t)r * n + c] + P[(size_t)c * n + r])) * scale;
    const int sweeps = jacobi_parallel(Y.data(), V.data(), n, 60, false, false, n <= 104);
    if (sweeps < 0) return -1;
    for (int r = 0; r < n; ++r)
        for (int c = 0; c <= r; ++c) {
            double acc = 0.0;
            for (int k = 0; k < n; ++k) {
                const double d = Y[(size_t)k * n + k];
                acc = std::fma(V[(size_t)r * n + k] * sqrt(d > 0.00000001 ? d : 0.00000001), V[(size_t)c * n + k], acc);
            }
            out[(size_t)r * n + c] = acc; out[(size_t)c * n + r] = acc;
        }
    return sweeps;
}

// batch runner: lockstep simulator + UKF, like orc_run_ekf_batch (outputs use n_max = 4 + 2*L_max)
double orc_run_ukf_batch(const slam_config* cfg, int L_max, int math, const double* map_xy, int L, const float* cmds, int T,
                         uint64_t seed, int64_t inst0, int B, int nthreads, double* x_out, double* P_out, int* M_out,
                         int* ids_out, double* avg_err, int* flags, double* truth_out, const double* vision) {
    const int n_max = 4 + 2 * L_max;
    std::atomic<int> next(0);
    auto worker = [&]() {
        std::vector<float> meas((size_t)3 * std::max(L, 1));
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            Ukf ukf(*cfg, L_max, math);
            if (cfg->reserved[0] == 1) {   // reserved[0] = 1: UKF_LOC with the simulator's map
                ukf.loc = true; ukf.mapf.resize((size_t)3 * L);
                for (int i = 0; i < L; ++i) { ukf.mapf[3 * i] = (float)i; ukf.mapf[3 * i + 1] = (float)map_xy[2 * i]; ukf.mapf[3 * i + 2] = (float)map_xy[2 * i + 1]; }
            }
            ukf.init((float)cfg->init_x, (float)cfg->init_y, (float)cfg->init_yaw);
            Sim sim;
            sim.cfg = *cfg; sim.L = L; sim.math = math; sim.map.assign(map_xy, map_xy + 2 * L);
            sim.xv[0] = cfg->init_x; sim.xv[1] = cfg->init_y; sim.xv[2] = cfg->init_yaw;
            double errsum = 0.0;
            for (int t = 0; t < T; ++t) {
                int k = 0;
                if (vision) { sim.cfg.range_max = vision[3 * t]; sim.cfg.fov_min = vision[3 * t + 1]; sim.cfg.fov_max = vision[3 * t + 2]; }
                auto draw = [&](int pair, int which) {
                    double u0, u1;
                    slam::noise_pair(seed, (uint64_t)(inst0 + b), (uint32_t)t, (uint32_t)pair, &u0, &u1);
                    return which ? u1 : u0;
                };
                if (math == MATH_DET) sim.step_t<DetMath>(cmds[2 * t], cmds[2 * t + 1], draw, meas.data(), nullptr, &k);
                else sim.step_t<LibmMath>(cmds[2 * t], cmds[2 * t + 1], draw, meas.data(), nullptr, &k);
                ukf.update(cmds[2 * t], cmds[2 * t + 1], meas.data(), k);
                errsum = errsum + (math == MATH_DET ? step_pos_error<DetMath>(wire_f32(ukf.x_t[0]), wire_f32(ukf.x_t[1]), sim.xv[0], sim.xv[1])
                                                    : step_pos_error<LibmMath>(wire_f32(ukf.x_t[0]), wire_f32(ukf.x_t[1]), sim.xv[0], sim.xv[1]));
            }
            const int n = ukf.n();
            if (x_out) memcpy(x_out + (size_t)b * n_max, ukf.x_t.data(), sizeof(double) * n);
            if (P_out) memcpy(P_out + (size_t)b * n_max * n_max, ukf.P_t.data(), sizeof(double) * n * n);
            if (M_out) M_out[b] = ukf.M;
            if (ids_out) memcpy(ids_out + (size_t)b * L_max, ukf.ids.data(), sizeof(int) * ukf.M);
            if (avg_err) avg_err[b] = T > 0 ? errsum / T : 0.0;
            if (flags) flags[b] = ukf.flags;
            if (truth_out) { truth_out[3 * b] = sim.xv[0]; truth_out[3 * b + 1] = sim.xv[1]; truth_out[3 * b + 2] = sim.xv[2]; }
        }
    };
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return std::chrono::duration<double>(std::chrono::steady_clock::now() - t0).count();
}

}  // extern "C"
