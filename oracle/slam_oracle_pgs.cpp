// slam_oracle_pgs.cpp — CPU ORACLE for pose-graph SLAM.  TEST INFRASTRUCTURE, NOT PRODUCT.
//
// Restates  ekf_ws/src/localization_pkg/src/pose_graph.cpp:68-267  (PoseGraph::init / updateNaiveVehPoseEstimate /
// getLandmarkIndexFromID / onLandmarkMeasurement / update / solvePoseGraph) with the GTSAM implementation selected
// in params.yaml:61.  The arithmetic of the solve lives in GTSAM, a find_package dependency that is NOT under
// /root/reference (localization_pkg/CMakeLists.txt:20, version unpinned; Noetic-era installs are GTSAM 4.0.x/4.1.x)
// and is not installed in this image => PARITY UNPINNED against the reference binary.  What is restated here is
// GTSAM's published algorithm for exactly the objects pose_graph.cpp instantiates (default build flags, i.e. without
// GTSAM_SLOW_BUT_CORRECT_EXPMAP / GTSAM_SLOW_BUT_CORRECT_BETWEENFACTOR):
//   * Pose2 chart at the origin: Retract(v) = Pose2(v0,v1,v2), Local(p) = (x, y, theta); retract(p,v) = p * Retract(v)
//   * PriorFactor<Pose2>:   e = -Local(x^-1 * prior),  H = I                                  (pose_graph.cpp:83-89)
//   * BetweenFactor<Pose2>: h = p1^-1 p2, e = Local(measured^-1 * h), H1 = -Ad(h^-1), H2 = I  (pose_graph.cpp:222)
//   * BearingRangeFactor<Pose2,Point2>: e = (wrap(bearing(p,l) - b), range(p,l) - r) with the exact Jacobians of
//     Pose2::bearing / Pose2::range                                                            (pose_graph.cpp:174)
//   * noiseModel::Diagonal::Sigmas -> whitened residual e_k / sigma_k; objective 0.5 * sum |whitened e|^2
//   * LevenbergMarquardtOptimizer with LevenbergMarquardtParams defaults (pose_graph.cpp:278-279): lambdaInitial 1e-5,
//     lambdaFactor 10, lambdaUpperBound 1e5, lambdaLowerBound 0, diagonalDamping false (damping = lambda * I),
//     useFixedLambdaFactor true, minModelFidelity 1e-3; NonlinearOptimizerParams defaults maxIterations 100,
//     relativeErrorTol 1e-5, absoluteErrorTol 1e-5, errorTol 0; tryLambda / iterate / defaultOptimize /
//     checkConvergence control flow as in LevenbergMarquardtOptimizer.cpp and NonlinearOptimizer.cpp.
//   * the damped normal equations (J^T J + lambda I) delta = -J^T e are solved EXACTLY (GTSAM: multifrontal Cholesky
//     with COLAMD ordering) — any exact elimination order gives the same delta up to rounding.  Here: poses first
//     (their Hessian is block-tridiagonal), dense Schur complement on the landmarks (LIN_SCHUR), or one dense
//     Cholesky of the whole system (LIN_DENSE, small graphs; the two must agree — tests/test_oracle_pgs.py).
// What pins it: (1) LIN_SCHUR vs LIN_DENSE, (2) the minimiser agrees with scipy.optimize.least_squares on the same
// residuals, (3) analytic Jacobians vs finite differences of the residuals along the retraction, (4) the Monte-Carlo
// error statistics land on the reference's published data/*/pose_graph_{init,result}.csv (tests/test_reference_statistics.py).
#include <math.h>
#include <stdint.h>
#include <string.h>

#include <algorithm>
#include <atomic>
#include <chrono>
#include <thread>
#include <vector>

#include "oracle_common.h"

namespace {
using namespace orc;

// LIN_SEG (round 5): the same Schur complement with the poses eliminated in a NESTED-DISSECTION order instead of 0, 1, 2, ...:
// the chain is cut at the separator poses SL, 2 SL, ... into segments whose interiors are eliminated first (independently of each
// other - on the GPU in parallel), then the separators, then the landmarks.  Any order gives the same delta up to rounding; this one
// has depth SL + N / SL instead of N and far less fill (a row of Y only holds the landmarks its own segment sees).  The GPU path
// (pgs_seg_kernel / pgs_sep_kernel / pgs_seg_syrk_kernel / pgs_seg_backsolve_kernel) follows exactly this statement.  lin_mode =
// LIN_SEG | (SL << 8), SL = 0: 32.
enum { LIN_SCHUR = 0, LIN_DENSE = 1, LIN_SEG = 2 };
enum { PGS_FLAG_POSE_CAP = 1, PGS_FLAG_LM_CAP = 2, PGS_FLAG_MEAS_CAP = 4, PGS_FLAG_NOT_CONVERGED = 8, PGS_FLAG_NONFINITE = 16 };

struct LmStats {
    int iterations = 0, trials = 0, flags = 0;
    double err_init = 0, err_final = 0, lambda = 0;
};

struct Pgs {
    slam_config cfg;
    FilterNoise nz;
    int N_max, L_max, KP, math;
    int timestep = 0, M = 0, flags = 0;
    bool solved = false;
    double prior[3] = {0, 0, 0};
    double w_prior[3], w_btw[3], w_meas[2];   // 1/sigma
    double cur[3] = {0, 0, 0};                // cur_veh_pose_estimate == x_t(0..2) (pose_graph.cpp:102-108)
    std::vector<double> pose0, lm0, pose1, lm1;   // initial_estimate / result
    std::vector<int> ids;
    std::vector<float> cmds;                  // [N_max-1][2]
    std::vector<int> cnt, mlm, mfirst;        // per pose: count; per slot: landmark index, first-detection flag
    std::vector<double> mb, mr;               // per slot: bearing, range (float32 wire values widened)

    Pgs(const slam_config& c, int Nmax, int Lmax, int kp, int mth) : cfg(c), N_max(Nmax), L_max(Lmax), KP(kp), math(mth) {
        nz = effective_noise(c);
        // pose_graph.cpp:83 prior sigmas; :52 process sigmas (V00, V00, V11); :54 sensing sigmas (W11, W00) = (bearing, range)
        const double sp[3] = {1.3, 1.3, 1.2};
        for (int k = 0; k < 3; ++k) w_prior[k] = 1.0 / sp[k];
        w_btw[0] = 1.0 / nz.V00; w_btw[1] = 1.0 / nz.V00; w_btw[2] = 1.0 / nz.V11;
        w_meas[0] = 1.0 / nz.W11; w_meas[1] = 1.0 / nz.W00;
        pose0.assign((size_t)3 * N_max, 0.0); pose1 = pose0;
        lm0.assign((size_t)2 * L_max, 0.0); lm1 = lm0;
        ids.assign(L_max, 0);
        cmds.assign((size_t)2 * N_max, 0.f);
        cnt.assign(N_max, 0); mlm.assign((size_t)N_max * KP, 0); mfirst = mlm;
        mb.assign((size_t)N_max * KP, 0.0); mr = mb;
    }
    int N() const { return timestep + 1; }

    void sc(double a, double* s, double* c) const { if (math == MATH_DET) slam::det_sincos(a, s, c); else { *s = ::sin(a); *c = ::cos(a); } }
    double at2(double y, double x) const { return math == MATH_DET ? slam::det_atan2(y, x) : ::atan2(y, x); }

    // PoseGraph::init, pose_graph.cpp:68-95
    void init(float x0, float y0, float yaw0) {
        timestep = 0; M = 0; flags = 0; solved = false;
        cur[0] = x0; cur[1] = y0; cur[2] = yaw0;
        for (int k = 0; k < 3; ++k) { pose0[k] = cur[k]; prior[k] = cur[k]; }
        std::fill(cnt.begin(), cnt.end(), 0);
    }
    // PoseGraph::updateNaiveVehPoseEstimate, pose_graph.cpp:97-119 (update_landmarks_after_adding = false, params.yaml:63)
    void set_secondary(const double sv[3]) { cur[0] = sv[0]; cur[1] = sv[1]; cur[2] = sv[2]; }
    // the graph-building half of PoseGraph::update, pose_graph.cpp:199-256 (the stop/solve logic of :201-214,258-266
    // belongs to the caller)
    int update(float fwd, float ang, const float* meas, int k) {
        if (timestep + 1 >= N_max) { flags |= PGS_FLAG_POSE_CAP; return flags; }
        cmds[2 * timestep] = fwd; cmds[2 * timestep + 1] = ang;       // BetweenFactor(t, t+1, Pose2(fwd, 0, ang)) :222
        timestep += 1;                                                 // :241
        for (int c = 0; c < 3; ++c) pose0[3 * timestep + c] = cur[c];  // initial_estimate.insert(key(t), cur) :248
        int used = 0;
        for (int l = 0; l < k; ++l) {                                  // :252-262
            const int id = (int)meas[3 * l];
            const float r = meas[3 * l + 1], b = meas[3 * l + 2];
            int idx = -1;                                              // getLandmarkIndexFromID :122-147
            for (int j = 0; j < M; ++j) if (ids[j] == id) { idx = j; break; }
            const bool first = idx < 0;
            if (first) {
                if (M >= L_max) { flags |= PGS_FLAG_LM_CAP; continue; }
                idx = M; ids[M] = id; M += 1;
                double s, c;                                           // :162  x_t(0) + range*cos(x_t(2)+bearing)
                sc(cur[2] + (double)b, &s, &c);
                lm0[2 * idx] = cur[0] + (double)r * c;
                lm0[2 * idx + 1] = cur[1] + (double)r * s;
            }
            if (used >= KP) { flags |= PGS_FLAG_MEAS_CAP; continue; }
            const size_t slot = (size_t)timestep * KP + used;          // BearingRangeFactor(key(t), lmkey, Rot2(b), r) :174
            mlm[slot] = idx; mfirst[slot] = first ? 1 : 0; mb[slot] = (double)b; mr[slot] = (double)r;
            used += 1;
        }
        cnt[timestep] = used;
        return flags;
    }

    // ---------------------------------------------------------------------------------------------------------
    // factors: whitened residuals and Jacobians
    // ---------------------------------------------------------------------------------------------------------
    void prior_factor(const double* p, double e[3]) const {
        double s, c;
        sc(p[2], &s, &c);
        const double dx = prior[0] - p[0], dy = prior[1] - p[1];
        e[0] = -(c * dx + s * dy) * w_prior[0];
        e[1] = -(-s * dx + c * dy) * w_prior[1];
        e[2] = -remainder(prior[2] - p[2], slam::kTwoPi) * w_prior[2];
    }
    // e (whitened), J1 (whitened 3x3 wrt pose a); J2 = diag(w_btw)
    void between_factor(const double* pa, const double* pb, float fwd, float ang, double e[3], double J1[9]) const {
        double si, ci, sm, cm;
        sc(pa[2], &si, &ci);
        sc((double)ang, &sm, &cm);
        const double dx = pb[0] - pa[0], dy = pb[1] - pa[1];
        const double hx = ci * dx + si * dy, hy = -si * dx + ci * dy, hth = pb[2] - pa[2];
        const double ux = hx - (double)fwd, uy = hy;
        e[0] = (cm * ux + sm * uy) * w_btw[0];
        e[1] = (-sm * ux + cm * uy) * w_btw[1];
        e[2] = remainder(hth - (double)ang, slam::kTwoPi) * w_btw[2];
        if (J1) {   // -Ad(h^-1):  h^-1 = (-R_h^T t_h, -th_h)
            double sh, ch;
            sc(hth, &sh, &ch);
            const double xi = -(ch * hx + sh * hy), yi = sh * hx - ch * hy;
            J1[0] = -ch * w_btw[0]; J1[1] = -sh * w_btw[0]; J1[2] = -yi * w_btw[0];
            J1[3] = sh * w_btw[1];  J1[4] = -ch * w_btw[1]; J1[5] = xi * w_btw[1];
            J1[6] = 0.0;            J1[7] = 0.0;            J1[8] = -w_btw[2];
        }
    }
    // e (whitened, [bearing, range]), Jp (2x3 whitened), Jl (2x2 whitened)
    void bearing_range_factor(const double* p, const double* l, double b, double r, double e[2], double Jp[6], double Jl[4]) const {
        double s, c, sb, cb;
        sc(p[2], &s, &c);
        sc(b, &sb, &cb);
        const double dx = l[0] - p[0], dy = l[1] - p[1];
        const double qx = c * dx + s * dy, qy = -s * dx + c * dy;   // transformTo
        const double d2 = qx * qx + qy * qy, n = ::sqrt(d2);
        // Rot2::relativeBearing(q) = (qx/n, qy/n); Local(measured, predicted) = theta(measured^-1 * predicted)
        const double cp = qx / n, sp = qy / n;
        e[0] = at2(cb * sp - sb * cp, cb * cp + sb * sp) * w_meas[0];
        e[1] = (n - r) * w_meas[1];
        if (Jp) {
            Jp[0] = (qy / d2) * w_meas[0]; Jp[1] = (-qx / d2) * w_meas[0]; Jp[2] = -w_meas[0];
            Jp[3] = (-qx / n) * w_meas[1]; Jp[4] = (-qy / n) * w_meas[1]; Jp[5] = 0.0;
            // d bearing / d l = [-qy/d2, qx/d2] R^T ;  d range / d l = [dx/n, dy/n]
            Jl[0] = ((-qy / d2) * c + (qx / d2) * (-s)) * w_meas[0];
            Jl[1] = ((-qy / d2) * s + (qx / d2) * c) * w_meas[0];
            Jl[2] = (dx / n) * w_meas[1];
            Jl[3] = (dy / n) * w_meas[1];
        }
    }

    // objective 0.5 * sum |whitened e|^2  (NonlinearFactorGraph::error)
    double cost(const double* pose, const double* lm) const {
        double tot = 0.0, e[3];
        prior_factor(pose, e);
        tot += 0.5 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
        const int n = N();
        for (int i = 0; i < n; ++i) {
            if (i + 1 < n) {
                between_factor(pose + 3 * i, pose + 3 * (i + 1), cmds[2 * i], cmds[2 * i + 1], e, nullptr);
                tot += 0.5 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            }
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                bearing_range_factor(pose + 3 * i, lm + 2 * mlm[k], mb[k], mr[k], e, nullptr, nullptr);
                tot += 0.5 * (e[0] * e[0] + e[1] * e[1]);
            }
        }
        return tot;
    }

    // ---------------------------------------------------------------------------------------------------------
    // normal equations in block form
    // ---------------------------------------------------------------------------------------------------------
    struct Lin {
        std::vector<double> A, C, gp;     // A [N][9] pose diagonal blocks, C [N][9] block H[i+1][i], gp [N][3]
        std::vector<double> E;            // [N*KP][6] block H[pose][landmark] (3x2)
        std::vector<double> D, gl;        // D [M][3] (xx, xy, yy), gl [M][2]
        double err = 0;
    };
    void linearize(const double* pose, const double* lm, Lin& L) const {
        const int n = N();
        L.A.assign((size_t)9 * n, 0.0); L.C.assign((size_t)9 * n, 0.0); L.gp.assign((size_t)3 * n, 0.0);
        L.E.assign((size_t)6 * n * KP, 0.0); L.D.assign((size_t)3 * std::max(M, 1), 0.0); L.gl.assign((size_t)2 * std::max(M, 1), 0.0);
        double tot = 0.0;
        auto add_JtJ = [](double* A, const double* J, int rows) {   // A += J^T J for a rows x 3 J
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) {
                    double v = 0.0;
                    for (int r = 0; r < rows; ++r) v += J[3 * r + a] * J[3 * r + b];
                    A[3 * a + b] += v;
                }
        };
        for (int i = 0; i < n; ++i) {
            double* A = &L.A[9 * i];
            double* g = &L.gp[3 * i];
            double e[3], J1[9];
            if (i == 0) {
                prior_factor(pose, e);
                for (int k = 0; k < 3; ++k) { A[4 * k] += w_prior[k] * w_prior[k]; g[k] += -e[k] * w_prior[k]; }
                tot += 0.5 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            }
            if (i > 0) {   // between (i-1, i): J2 = diag(w)
                between_factor(pose + 3 * (i - 1), pose + 3 * i, cmds[2 * (i - 1)], cmds[2 * (i - 1) + 1], e, J1);
                for (int k = 0; k < 3; ++k) { A[4 * k] += w_btw[k] * w_btw[k]; g[k] += -e[k] * w_btw[k]; }
                double* C = &L.C[9 * (i - 1)];   // H[i][i-1] = J2^T J1
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) C[3 * a + b] = w_btw[a] * J1[3 * a + b];
            }
            if (i + 1 < n) {   // between (i, i+1): J1
                between_factor(pose + 3 * i, pose + 3 * (i + 1), cmds[2 * i], cmds[2 * i + 1], e, J1);
                add_JtJ(A, J1, 3);
                for (int a = 0; a < 3; ++a) g[a] += -(J1[a] * e[0] + J1[3 + a] * e[1] + J1[6 + a] * e[2]);
                tot += 0.5 * (e[0] * e[0] + e[1] * e[1] + e[2] * e[2]);
            }
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                const int j = mlm[k];
                double e2[2], Jp[6], Jl[4];
                bearing_range_factor(pose + 3 * i, lm + 2 * j, mb[k], mr[k], e2, Jp, Jl);
                add_JtJ(A, Jp, 2);
                for (int a = 0; a < 3; ++a) g[a] += -(Jp[a] * e2[0] + Jp[3 + a] * e2[1]);
                double* E = &L.E[6 * k];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 2; ++b) E[2 * a + b] = Jp[a] * Jl[b] + Jp[3 + a] * Jl[2 + b];
                L.D[3 * j] += Jl[0] * Jl[0] + Jl[2] * Jl[2];
                L.D[3 * j + 1] += Jl[0] * Jl[1] + Jl[2] * Jl[3];
                L.D[3 * j + 2] += Jl[1] * Jl[1] + Jl[3] * Jl[3];
                L.gl[2 * j] += -(Jl[0] * e2[0] + Jl[2] * e2[1]);
                L.gl[2 * j + 1] += -(Jl[1] * e2[0] + Jl[3] * e2[1]);
                tot += 0.5 * (e2[0] * e2[0] + e2[1] * e2[1]);
            }
        }
        L.err = tot;
    }

    // dense Cholesky solve in place (lower); false if a pivot is not positive
    static bool chol_solve(std::vector<double>& S, int n, std::vector<double>& rhs) {
        for (int j = 0; j < n; ++j) {
            double d = S[(size_t)j * n + j];
            for (int k = 0; k < j; ++k) d -= S[(size_t)j * n + k] * S[(size_t)j * n + k];
            if (!(d > 0.0)) return false;
            d = ::sqrt(d);
            S[(size_t)j * n + j] = d;
            for (int i = j + 1; i < n; ++i) {
                double v = S[(size_t)i * n + j];
                for (int k = 0; k < j; ++k) v -= S[(size_t)i * n + k] * S[(size_t)j * n + k];
                S[(size_t)i * n + j] = v / d;
            }
        }
        for (int i = 0; i < n; ++i) {
            double v = rhs[i];
            for (int k = 0; k < i; ++k) v -= S[(size_t)i * n + k] * rhs[k];
            rhs[i] = v / S[(size_t)i * n + i];
        }
        for (int i = n - 1; i >= 0; --i) {
            double v = rhs[i];
            for (int k = i + 1; k < n; ++k) v -= S[(size_t)k * n + i] * rhs[k];
            rhs[i] = v / S[(size_t)i * n + i];
        }
        return true;
    }

    // (J^T J + lambda I) delta = g, poses eliminated first.  dp [N][3], dl [M][2].
    bool solve_schur(const Lin& L, double lambda, std::vector<double>& dp, std::vector<double>& dl) const {
        const int n = N(), m2 = 2 * M, W = m2 + 1;   // Y has one extra column: z (the gradient)
        std::vector<double> Linv((size_t)6 * n), G((size_t)9 * n, 0.0), Y((size_t)3 * n * W, 0.0);
        double yprev_dummy = 0; (void)yprev_dummy;
        for (int i = 0; i < n; ++i) {
            double T[9];
            for (int k = 0; k < 9; ++k) T[k] = L.A[9 * i + k];
            T[0] += lambda; T[4] += lambda; T[8] += lambda;
            double* Gi = &G[9 * i];
            if (i > 0) {   // G = C_{i-1} Linv_{i-1}^T ;  T -= G G^T
                const double* C = &L.C[9 * (i - 1)];
                const double* I = &Linv[6 * (i - 1)];   // i00 i10 i11 i20 i21 i22
                for (int r = 0; r < 3; ++r) {
                    Gi[3 * r + 0] = C[3 * r] * I[0];
                    Gi[3 * r + 1] = C[3 * r] * I[1] + C[3 * r + 1] * I[2];
                    Gi[3 * r + 2] = (C[3 * r] * I[3] + C[3 * r + 1] * I[4]) + C[3 * r + 2] * I[5];
                }
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b)
                        T[3 * a + b] -= (Gi[3 * a] * Gi[3 * b] + Gi[3 * a + 1] * Gi[3 * b + 1]) + Gi[3 * a + 2] * Gi[3 * b + 2];
            }
            if (!(T[0] > 0.0)) return false;
            const double l00 = ::sqrt(T[0]), l10 = T[3] / l00, l20 = T[6] / l00;
            const double t11 = T[4] - l10 * l10;
            if (!(t11 > 0.0)) return false;
            const double l11 = ::sqrt(t11), l21 = (T[7] - l20 * l10) / l11;
            const double t22 = (T[8] - l20 * l20) - l21 * l21;
            if (!(t22 > 0.0)) return false;
            const double l22 = ::sqrt(t22);
            double* I = &Linv[6 * i];
            I[0] = 1.0 / l00; I[2] = 1.0 / l11; I[5] = 1.0 / l22;
            I[1] = -(l10 * I[0]) * I[2];
            I[4] = -(l21 * I[2]) * I[5];
            I[3] = -(l20 * I[0] + l21 * I[1]) * I[5];
            // Y_i = Linv (E_row_i - G Y_{i-1})
            double* Yi = &Y[(size_t)3 * i * W];
            const double* Yp = i > 0 ? &Y[(size_t)3 * (i - 1) * W] : nullptr;
            for (int c = 0; c < W; ++c) {
                double u[3] = {0, 0, 0};
                if (c == m2) { u[0] = L.gp[3 * i]; u[1] = L.gp[3 * i + 1]; u[2] = L.gp[3 * i + 2]; }
                if (Yp)
                    for (int r = 0; r < 3; ++r) u[r] -= (Gi[3 * r] * Yp[c] + Gi[3 * r + 1] * Yp[W + c]) + Gi[3 * r + 2] * Yp[2 * W + c];
                Yi[c] = u[0]; Yi[W + c] = u[1]; Yi[2 * W + c] = u[2];
            }
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                const int j = mlm[k];
                for (int r = 0; r < 3; ++r) { Yi[r * W + 2 * j] += L.E[6 * k + 2 * r]; Yi[r * W + 2 * j + 1] += L.E[6 * k + 2 * r + 1]; }
            }
            for (int c = 0; c < W; ++c) {
                const double u0 = Yi[c], u1 = Yi[W + c], u2 = Yi[2 * W + c];
                Yi[c] = I[0] * u0;
                Yi[W + c] = I[1] * u0 + I[2] * u1;
                Yi[2 * W + c] = (I[3] * u0 + I[4] * u1) + I[5] * u2;
            }
        }
        // S = D + lambda I - Y^T Y ; rhs = gl - Y^T z
        std::vector<double> S((size_t)m2 * m2, 0.0), rhs(m2, 0.0);
        for (int j = 0; j < M; ++j) {
            S[(size_t)(2 * j) * m2 + 2 * j] = L.D[3 * j] + lambda;
            S[(size_t)(2 * j + 1) * m2 + 2 * j] = L.D[3 * j + 1];
            S[(size_t)(2 * j) * m2 + 2 * j + 1] = L.D[3 * j + 1];
            S[(size_t)(2 * j + 1) * m2 + 2 * j + 1] = L.D[3 * j + 2] + lambda;
            rhs[2 * j] = L.gl[2 * j]; rhs[2 * j + 1] = L.gl[2 * j + 1];
        }
        for (int k = 0; k < 3 * n; ++k) {
            const double* y = &Y[(size_t)k * W];
            for (int a = 0; a < m2; ++a) {
                const double ya = y[a];
                if (ya == 0.0) continue;
                double* Sa = &S[(size_t)a * m2];
                for (int b = 0; b <= a; ++b) Sa[b] -= ya * y[b];
                rhs[a] -= ya * y[m2];
            }
        }
        dl.assign(std::max(m2, 1), 0.0);
        if (m2 > 0) {
            if (!chol_solve(S, m2, rhs)) return false;
            for (int a = 0; a < m2; ++a) dl[a] = rhs[a];
        }
        // poses: H_pp dp = gp - E dl   (forward / backward over the chain)
        dp.assign((size_t)3 * n, 0.0);
        std::vector<double> zz((size_t)3 * n);
        for (int i = 0; i < n; ++i) {
            double u[3] = {L.gp[3 * i], L.gp[3 * i + 1], L.gp[3 * i + 2]};
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                const int j = mlm[k];
                for (int r = 0; r < 3; ++r) u[r] -= L.E[6 * k + 2 * r] * dl[2 * j] + L.E[6 * k + 2 * r + 1] * dl[2 * j + 1];
            }
            if (i > 0) {
                const double* Gi = &G[9 * i];
                const double* zp = &zz[3 * (i - 1)];
                for (int r = 0; r < 3; ++r) u[r] -= (Gi[3 * r] * zp[0] + Gi[3 * r + 1] * zp[1]) + Gi[3 * r + 2] * zp[2];
            }
            const double* I = &Linv[6 * i];
            zz[3 * i] = I[0] * u[0];
            zz[3 * i + 1] = I[1] * u[0] + I[2] * u[1];
            zz[3 * i + 2] = (I[3] * u[0] + I[4] * u[1]) + I[5] * u[2];
        }
        for (int i = n - 1; i >= 0; --i) {
            double u[3] = {zz[3 * i], zz[3 * i + 1], zz[3 * i + 2]};
            if (i + 1 < n) {
                const double* Gn = &G[9 * (i + 1)];
                const double* dn = &dp[3 * (i + 1)];
                for (int r = 0; r < 3; ++r) u[r] -= (Gn[r] * dn[0] + Gn[3 + r] * dn[1]) + Gn[6 + r] * dn[2];
            }
            const double* I = &Linv[6 * i];   // dp_i = Linv^T u
            dp[3 * i + 2] = I[5] * u[2];
            dp[3 * i + 1] = I[2] * u[1] + I[4] * u[2];
            dp[3 * i] = (I[0] * u[0] + I[1] * u[1]) + I[3] * u[2];
        }
        return true;
    }

    // ---- 3x3 helpers of the segmented elimination (the operation order the GPU kernels mirror) ----
    // I = inverse of the Cholesky factor of T (lower; i00 i10 i11 i20 i21 i22); false if a pivot is not positive
    static bool chol_inv3(const double T[9], double I[6]) {
        if (!(T[0] > 0.0)) return false;
        const double l00 = ::sqrt(T[0]), l10 = T[3] / l00, l20 = T[6] / l00;
        const double t11 = T[4] - l10 * l10;
        if (!(t11 > 0.0)) return false;
        const double l11 = ::sqrt(t11), l21 = (T[7] - l20 * l10) / l11;
        const double t22 = (T[8] - l20 * l20) - l21 * l21;
        if (!(t22 > 0.0)) return false;
        const double l22 = ::sqrt(t22);
        I[0] = 1.0 / l00; I[2] = 1.0 / l11; I[5] = 1.0 / l22;
        I[1] = -(l10 * I[0]) * I[2];
        I[4] = -(l21 * I[2]) * I[5];
        I[3] = -(l20 * I[0] + l21 * I[1]) * I[5];
        return true;
    }
    // G = X Linv^T for a 3x3 X (row r of G from row r of X), Linv lower triangular
    static void mul_linvT(const double* X, const double* I, double* G) {
        for (int r = 0; r < 3; ++r) {
            G[3 * r + 0] = X[3 * r] * I[0];
            G[3 * r + 1] = X[3 * r] * I[1] + X[3 * r + 1] * I[2];
            G[3 * r + 2] = (X[3 * r] * I[3] + X[3 * r + 1] * I[4]) + X[3 * r + 2] * I[5];
        }
    }
    // P = X Z^T (3x3)
    static void mul_abT(const double* X, const double* Z, double* P) {
        for (int a = 0; a < 3; ++a)
            for (int b = 0; b < 3; ++b) P[3 * a + b] = (X[3 * a] * Z[3 * b] + X[3 * a + 1] * Z[3 * b + 1]) + X[3 * a + 2] * Z[3 * b + 2];
    }
    static void linv_apply(const double* I, const double u[3], double y[3]) {   // y = Linv u
        y[0] = I[0] * u[0];
        y[1] = I[1] * u[0] + I[2] * u[1];
        y[2] = (I[3] * u[0] + I[4] * u[1]) + I[5] * u[2];
    }
    static void linvT_apply(const double* I, const double u[3], double y[3]) {   // y = Linv^T u
        y[2] = I[5] * u[2];
        y[1] = I[2] * u[1] + I[4] * u[2];
        y[0] = (I[0] * u[0] + I[1] * u[1]) + I[3] * u[2];
    }
    static void sub_Gv(const double* G, const double v[3], double u[3]) {        // u -= G v
        for (int r = 0; r < 3; ++r) u[r] -= (G[3 * r] * v[0] + G[3 * r + 1] * v[1]) + G[3 * r + 2] * v[2];
    }
    static void sub_GTv(const double* G, const double v[3], double u[3]) {       // u -= G^T v
        for (int r = 0; r < 3; ++r) u[r] -= (G[r] * v[0] + G[3 + r] * v[1]) + G[6 + r] * v[2];
    }

    // (J^T J + lambda I) delta = g with the poses eliminated segment by segment (LIN_SEG, see the top of the file).
    // Separators: poses k SL (k = 1 .. NS, NS = (N - 2) / SL); segment p = 0 .. NS holds the poses strictly between separators p and
    // p + 1 (segment 0 starts at pose 0, the last one ends at pose N - 1).
    bool solve_seg(const Lin& L, double lambda, int SL, std::vector<double>& dp, std::vector<double>& dl) const {
        const int n = N(), m2 = 2 * M, W = m2 + 1;
        const int NS = n >= 2 ? (n - 2) / SL : 0, nseg = NS + 1;
        auto seg_lo = [&](int p) { return p == 0 ? 0 : p * SL + 1; };
        auto seg_hi = [&](int p) { return p < NS ? (p + 1) * SL : n; };
        std::vector<double> Linv((size_t)6 * n), Ginn((size_t)9 * n, 0.0), Gsep((size_t)9 * n, 0.0);
        std::vector<double> accL((size_t)6 * nseg, 0.0), Aright((size_t)6 * nseg, 0.0), Gright((size_t)9 * nseg, 0.0), Hba((size_t)9 * nseg, 0.0);
        // ---- interiors: chain factor inside the segment + the spike towards the left separator ----
        for (int p = 0; p < nseg; ++p) {
            const int lo = seg_lo(p), hi = seg_hi(p), a = p * SL;
            double Bcur[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
            if (p >= 1) {   // H[a][a+1] = C_a^T
                const double* C = &L.C[9 * a];
                for (int r = 0; r < 3; ++r) for (int c = 0; c < 3; ++c) Bcur[3 * r + c] = C[3 * c + r];
            }
            for (int i = lo; i < hi; ++i) {
                double T[9];
                for (int k = 0; k < 9; ++k) T[k] = L.A[9 * i + k];
                T[0] += lambda; T[4] += lambda; T[8] += lambda;
                double* Gi = &Ginn[9 * i];
                if (i > lo) {
                    mul_linvT(&L.C[9 * (i - 1)], &Linv[6 * (i - 1)], Gi);
                    for (int a2 = 0; a2 < 3; ++a2)
                        for (int b2 = 0; b2 < 3; ++b2)
                            T[3 * a2 + b2] -= (Gi[3 * a2] * Gi[3 * b2] + Gi[3 * a2 + 1] * Gi[3 * b2 + 1]) + Gi[3 * a2 + 2] * Gi[3 * b2 + 2];
                    if (p >= 1) {   // fill of eliminating pose i-1: H[a][i] = -(Gsep_{i-1} Ginn_i^T)
                        double P[9];
                        mul_abT(&Gsep[9 * (i - 1)], Gi, P);
                        for (int k = 0; k < 9; ++k) Bcur[k] = -P[k];
                    }
                }
                if (!chol_inv3(T, &Linv[6 * i])) return false;
                if (p >= 1) {
                    double* Gs = &Gsep[9 * i];
                    mul_linvT(Bcur, &Linv[6 * i], Gs);
                    double* aL = &accL[6 * p];   // lower triangle 00 10 11 20 21 22 of sum Gsep Gsep^T
                    aL[0] += (Gs[0] * Gs[0] + Gs[1] * Gs[1]) + Gs[2] * Gs[2];
                    aL[1] += (Gs[3] * Gs[0] + Gs[4] * Gs[1]) + Gs[5] * Gs[2];
                    aL[2] += (Gs[3] * Gs[3] + Gs[4] * Gs[4]) + Gs[5] * Gs[5];
                    aL[3] += (Gs[6] * Gs[0] + Gs[7] * Gs[1]) + Gs[8] * Gs[2];
                    aL[4] += (Gs[6] * Gs[3] + Gs[7] * Gs[4]) + Gs[8] * Gs[5];
                    aL[5] += (Gs[6] * Gs[6] + Gs[7] * Gs[7]) + Gs[8] * Gs[8];
                }
            }
            if (p < NS) {   // the right separator b = hi couples to the last interior pose e = hi - 1 through C_e
                const int e = hi - 1;
                double* Gr = &Gright[9 * p];
                mul_linvT(&L.C[9 * e], &Linv[6 * e], Gr);
                double* aR = &Aright[6 * p];
                aR[0] = (Gr[0] * Gr[0] + Gr[1] * Gr[1]) + Gr[2] * Gr[2];
                aR[1] = (Gr[3] * Gr[0] + Gr[4] * Gr[1]) + Gr[5] * Gr[2];
                aR[2] = (Gr[3] * Gr[3] + Gr[4] * Gr[4]) + Gr[5] * Gr[5];
                aR[3] = (Gr[6] * Gr[0] + Gr[7] * Gr[1]) + Gr[8] * Gr[2];
                aR[4] = (Gr[6] * Gr[3] + Gr[7] * Gr[4]) + Gr[8] * Gr[5];
                aR[5] = (Gr[6] * Gr[6] + Gr[7] * Gr[7]) + Gr[8] * Gr[8];
                if (p >= 1) mul_abT(Gr, &Gsep[9 * e], &Hba[9 * p]);   // H[b][a] = -(this)
            }
        }
        // ---- separators: a chain of NS poses ----
        std::vector<double> LinvS((size_t)6 * std::max(NS, 1)), GS((size_t)9 * std::max(NS, 1), 0.0);
        for (int k = 1; k <= NS; ++k) {
            const int s = k * SL;
            double T[9];
            for (int q = 0; q < 9; ++q) T[q] = L.A[9 * s + q];
            T[0] += lambda; T[4] += lambda; T[8] += lambda;
            const double* aR = &Aright[6 * (k - 1)];
            const double* aL = &accL[6 * k];
            T[0] = (T[0] - aR[0]) - aL[0];
            T[3] = (T[3] - aR[1]) - aL[1]; T[4] = (T[4] - aR[2]) - aL[2];
            T[6] = (T[6] - aR[3]) - aL[3]; T[7] = (T[7] - aR[4]) - aL[4]; T[8] = (T[8] - aR[5]) - aL[5];
            T[1] = T[3]; T[2] = T[6]; T[5] = T[7];
            double* Gk = &GS[9 * (k - 1)];
            if (k >= 2) {
                double Hk[9];
                for (int q = 0; q < 9; ++q) Hk[q] = -Hba[9 * (k - 1) + q];
                mul_linvT(Hk, &LinvS[6 * (k - 2)], Gk);
                for (int a2 = 0; a2 < 3; ++a2)
                    for (int b2 = 0; b2 < 3; ++b2)
                        T[3 * a2 + b2] -= (Gk[3 * a2] * Gk[3 * b2] + Gk[3 * a2 + 1] * Gk[3 * b2 + 1]) + Gk[3 * a2 + 2] * Gk[3 * b2 + 2];
            }
            if (!chol_inv3(T, &LinvS[6 * (k - 1)])) return false;
        }
        // ---- Y = L^-1 [H_pl | g_p] in the same order: interior rows, then separator rows; S = D + lambda I - Y^T Y ----
        // per-pose E blocks by landmark column (dense row of 3 x W per pose would be N x W: walk the factor slots instead)
        auto add_E = [&](int i, std::vector<double>& u /* [3][W] */) {
            for (int s2 = 0; s2 < cnt[i]; ++s2) {
                const size_t k = (size_t)i * KP + s2;
                const int j = mlm[k];
                for (int r = 0; r < 3; ++r) { u[(size_t)r * W + 2 * j] += L.E[6 * k + 2 * r]; u[(size_t)r * W + 2 * j + 1] += L.E[6 * k + 2 * r + 1]; }
            }
        };
        std::vector<double> S((size_t)m2 * m2, 0.0), rhs(m2, 0.0);
        for (int j = 0; j < M; ++j) {
            S[(size_t)(2 * j) * m2 + 2 * j] = L.D[3 * j] + lambda;
            S[(size_t)(2 * j + 1) * m2 + 2 * j] = L.D[3 * j + 1];
            S[(size_t)(2 * j) * m2 + 2 * j + 1] = L.D[3 * j + 1];
            S[(size_t)(2 * j + 1) * m2 + 2 * j + 1] = L.D[3 * j + 2] + lambda;
            rhs[2 * j] = L.gl[2 * j]; rhs[2 * j + 1] = L.gl[2 * j + 1];
        }
        auto syrk_rows = [&](const double* y /* [W] */) {
            for (int a2 = 0; a2 < m2; ++a2) {
                const double ya = y[a2];
                if (ya == 0.0) continue;
                double* Sa = &S[(size_t)a2 * m2];
                for (int b2 = 0; b2 <= a2; ++b2) Sa[b2] -= ya * y[b2];
                rhs[a2] -= ya * y[m2];
            }
        };
        std::vector<double> RcL((size_t)3 * W * nseg, 0.0), RcR((size_t)3 * W * nseg, 0.0), u((size_t)3 * W), yprev((size_t)3 * W), ysep;
        // the separator rows are subtracted from S FIRST (the GPU forms base - Ysep^T Ysep with one tile kernel and then subtracts the
        // segments' products in segment order), so they are computed before the interior rows are accumulated: two passes
        std::vector<double> Yint((size_t)3 * n * W, 0.0);
        for (int p = 0; p < nseg; ++p) {
            const int lo = seg_lo(p), hi = seg_hi(p);
            std::fill(yprev.begin(), yprev.end(), 0.0);
            for (int i = lo; i < hi; ++i) {
                std::fill(u.begin(), u.end(), 0.0);
                u[m2] = L.gp[3 * i]; u[(size_t)W + m2] = L.gp[3 * i + 1]; u[(size_t)2 * W + m2] = L.gp[3 * i + 2];
                const double* Gi = &Ginn[9 * i];
                if (i > lo)
                    for (int c = 0; c < W; ++c) {
                        double v[3] = {yprev[c], yprev[W + c], yprev[2 * (size_t)W + c]}, uu[3] = {u[c], u[W + c], u[2 * (size_t)W + c]};
                        sub_Gv(Gi, v, uu);
                        u[c] = uu[0]; u[W + c] = uu[1]; u[2 * (size_t)W + c] = uu[2];
                    }
                add_E(i, u);
                double* Yi = &Yint[(size_t)3 * i * W];
                for (int c = 0; c < W; ++c) {
                    const double uu[3] = {u[c], u[W + c], u[2 * (size_t)W + c]};
                    double y[3];
                    linv_apply(&Linv[6 * i], uu, y);
                    Yi[c] = y[0]; Yi[W + c] = y[1]; Yi[2 * (size_t)W + c] = y[2];
                    yprev[c] = y[0]; yprev[W + c] = y[1]; yprev[2 * (size_t)W + c] = y[2];
                    if (p >= 1) {   // R_a -= Gsep_i Y_i, accumulated with a plus sign
                        const double* Gs = &Gsep[9 * i];
                        double* rc = &RcL[(size_t)3 * W * p];
                        for (int r = 0; r < 3; ++r) rc[(size_t)r * W + c] += (Gs[3 * r] * y[0] + Gs[3 * r + 1] * y[1]) + Gs[3 * r + 2] * y[2];
                    }
                }
            }
            if (p < NS) {
                const double* Gr = &Gright[9 * p];
                double* rc = &RcR[(size_t)3 * W * p];
                for (int c = 0; c < W; ++c)
                    for (int r = 0; r < 3; ++r) rc[(size_t)r * W + c] = (Gr[3 * r] * yprev[c] + Gr[3 * r + 1] * yprev[W + c]) + Gr[3 * r + 2] * yprev[2 * (size_t)W + c];
            }
        }
        std::vector<double> Ysep((size_t)3 * std::max(NS, 1) * W, 0.0);
        for (int k = 1; k <= NS; ++k) {
            const int s = k * SL;
            std::fill(u.begin(), u.end(), 0.0);
            u[m2] = L.gp[3 * s]; u[(size_t)W + m2] = L.gp[3 * s + 1]; u[(size_t)2 * W + m2] = L.gp[3 * s + 2];
            const double* rr = &RcR[(size_t)3 * W * (k - 1)];
            const double* rl = &RcL[(size_t)3 * W * k];
            for (size_t q = 0; q < (size_t)3 * W; ++q) u[q] = (u[q] - rr[q]) - rl[q];
            if (k >= 2) {
                const double* Gk = &GS[9 * (k - 1)];
                const double* Yp = &Ysep[(size_t)3 * (k - 2) * W];
                for (int c = 0; c < W; ++c) {
                    double v[3] = {Yp[c], Yp[W + c], Yp[2 * (size_t)W + c]}, uu[3] = {u[c], u[W + c], u[2 * (size_t)W + c]};
                    sub_Gv(Gk, v, uu);
                    u[c] = uu[0]; u[W + c] = uu[1]; u[2 * (size_t)W + c] = uu[2];
                }
            }
            add_E(s, u);
            double* Yk = &Ysep[(size_t)3 * (k - 1) * W];
            for (int c = 0; c < W; ++c) {
                const double uu[3] = {u[c], u[W + c], u[2 * (size_t)W + c]};
                double y[3];
                linv_apply(&LinvS[6 * (k - 1)], uu, y);
                Yk[c] = y[0]; Yk[W + c] = y[1]; Yk[2 * (size_t)W + c] = y[2];
            }
        }
        for (int k = 0; k < 3 * NS; ++k) syrk_rows(&Ysep[(size_t)k * W]);
        for (int p = 0; p < nseg; ++p)
            for (int k = 3 * seg_lo(p); k < 3 * seg_hi(p); ++k) syrk_rows(&Yint[(size_t)k * W]);
        dl.assign(std::max(m2, 1), 0.0);
        if (m2 > 0) {
            if (!chol_solve(S, m2, rhs)) return false;
            for (int a2 = 0; a2 < m2; ++a2) dl[a2] = rhs[a2];
        }
        // ---- poses: forward over interiors, then separators; backward over separators, then interiors ----
        dp.assign((size_t)3 * n, 0.0);
        std::vector<double> zz((size_t)3 * n), racc((size_t)3 * nseg, 0.0), rright((size_t)3 * nseg, 0.0), zs((size_t)3 * std::max(NS, 1)), ds((size_t)3 * (NS + 2), 0.0);
        auto rhs_pose = [&](int i, double uu[3]) {
            uu[0] = L.gp[3 * i]; uu[1] = L.gp[3 * i + 1]; uu[2] = L.gp[3 * i + 2];
            for (int s2 = 0; s2 < cnt[i]; ++s2) {
                const size_t k = (size_t)i * KP + s2;
                const int j = mlm[k];
                for (int r = 0; r < 3; ++r) uu[r] -= L.E[6 * k + 2 * r] * dl[2 * j] + L.E[6 * k + 2 * r + 1] * dl[2 * j + 1];
            }
        };
        for (int p = 0; p < nseg; ++p) {
            const int lo = seg_lo(p), hi = seg_hi(p);
            for (int i = lo; i < hi; ++i) {
                double uu[3];
                rhs_pose(i, uu);
                if (i > lo) sub_Gv(&Ginn[9 * i], &zz[3 * (i - 1)], uu);
                linv_apply(&Linv[6 * i], uu, &zz[3 * i]);
                if (p >= 1) {
                    const double* Gs = &Gsep[9 * i];
                    for (int r = 0; r < 3; ++r) racc[3 * p + r] += (Gs[3 * r] * zz[3 * i] + Gs[3 * r + 1] * zz[3 * i + 1]) + Gs[3 * r + 2] * zz[3 * i + 2];
                }
            }
            if (p < NS) {
                const double* Gr = &Gright[9 * p];
                const double* ze = &zz[3 * (hi - 1)];
                for (int r = 0; r < 3; ++r) rright[3 * p + r] = (Gr[3 * r] * ze[0] + Gr[3 * r + 1] * ze[1]) + Gr[3 * r + 2] * ze[2];
            }
        }
        for (int k = 1; k <= NS; ++k) {
            double uu[3];
            rhs_pose(k * SL, uu);
            for (int r = 0; r < 3; ++r) uu[r] = (uu[r] - rright[3 * (k - 1) + r]) - racc[3 * k + r];
            if (k >= 2) sub_Gv(&GS[9 * (k - 1)], &zs[3 * (k - 2)], uu);
            linv_apply(&LinvS[6 * (k - 1)], uu, &zs[3 * (k - 1)]);
        }
        for (int k = NS; k >= 1; --k) {   // ds[k] = step of separator k (ds[NS + 1] = 0)
            double v[3] = {zs[3 * (k - 1)], zs[3 * (k - 1) + 1], zs[3 * (k - 1) + 2]};
            if (k < NS) sub_GTv(&GS[9 * k], &ds[3 * (k + 1)], v);
            linvT_apply(&LinvS[6 * (k - 1)], v, &ds[3 * k]);
            for (int r = 0; r < 3; ++r) dp[3 * (k * SL) + r] = ds[3 * k + r];
        }
        for (int p = 0; p < nseg; ++p) {
            const int lo = seg_lo(p), hi = seg_hi(p);
            for (int i = hi - 1; i >= lo; --i) {
                double v[3] = {zz[3 * i], zz[3 * i + 1], zz[3 * i + 2]};
                if (i + 1 < hi) sub_GTv(&Ginn[9 * (i + 1)], &dp[3 * (i + 1)], v);
                else if (p < NS) sub_GTv(&Gright[9 * p], &ds[3 * (p + 1)], v);
                if (p >= 1) sub_GTv(&Gsep[9 * i], &ds[3 * p], v);
                linvT_apply(&Linv[6 * i], v, &dp[3 * i]);
            }
        }
        return true;
    }

    // the same system as ONE dense matrix (validation of the elimination above; small graphs only)
    bool solve_dense(const Lin& L, double lambda, std::vector<double>& dp, std::vector<double>& dl) const {
        const int n = N(), np = 3 * n, nt = np + 2 * M;
        std::vector<double> H((size_t)nt * nt, 0.0), g(nt, 0.0);
        for (int i = 0; i < n; ++i) {
            for (int a = 0; a < 3; ++a)
                for (int b = 0; b < 3; ++b) H[(size_t)(3 * i + a) * nt + 3 * i + b] = L.A[9 * i + 3 * a + b];
            if (i + 1 < n)
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) {
                        H[(size_t)(3 * (i + 1) + a) * nt + 3 * i + b] = L.C[9 * i + 3 * a + b];
                        H[(size_t)(3 * i + b) * nt + 3 * (i + 1) + a] = L.C[9 * i + 3 * a + b];
                    }
            for (int a = 0; a < 3; ++a) g[3 * i + a] = L.gp[3 * i + a];
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                const int j = mlm[k];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 2; ++b) {
                        H[(size_t)(np + 2 * j + b) * nt + 3 * i + a] += L.E[6 * k + 2 * a + b];
                        H[(size_t)(3 * i + a) * nt + np + 2 * j + b] += L.E[6 * k + 2 * a + b];
                    }
            }
        }
        for (int j = 0; j < M; ++j) {
            H[(size_t)(np + 2 * j) * nt + np + 2 * j] = L.D[3 * j];
            H[(size_t)(np + 2 * j + 1) * nt + np + 2 * j] = L.D[3 * j + 1];
            H[(size_t)(np + 2 * j) * nt + np + 2 * j + 1] = L.D[3 * j + 1];
            H[(size_t)(np + 2 * j + 1) * nt + np + 2 * j + 1] = L.D[3 * j + 2];
            g[np + 2 * j] = L.gl[2 * j]; g[np + 2 * j + 1] = L.gl[2 * j + 1];
        }
        for (int k = 0; k < nt; ++k) H[(size_t)k * nt + k] += lambda;
        if (!chol_solve(H, nt, g)) return false;
        dp.assign(g.begin(), g.begin() + np);
        dl.assign(g.begin() + np, g.end());
        if (dl.empty()) dl.push_back(0.0);
        return true;
    }

    // 0.5 |J delta + e|^2 over all factors (GaussianFactorGraph::error(delta) of the UNDAMPED linearisation)
    double linear_error(const double* pose, const double* lm, const double* dp, const double* dl) const {
        double tot = 0.0, e[3], J1[9];
        prior_factor(pose, e);
        for (int k = 0; k < 3; ++k) { const double v = e[k] + w_prior[k] * dp[k]; tot += 0.5 * v * v; }
        const int n = N();
        for (int i = 0; i < n; ++i) {
            if (i + 1 < n) {
                between_factor(pose + 3 * i, pose + 3 * (i + 1), cmds[2 * i], cmds[2 * i + 1], e, J1);
                for (int r = 0; r < 3; ++r) {
                    const double v = (e[r] + ((J1[3 * r] * dp[3 * i] + J1[3 * r + 1] * dp[3 * i + 1]) + J1[3 * r + 2] * dp[3 * i + 2])) + w_btw[r] * dp[3 * (i + 1) + r];
                    tot += 0.5 * v * v;
                }
            }
            for (int s = 0; s < cnt[i]; ++s) {
                const size_t k = (size_t)i * KP + s;
                const int j = mlm[k];
                double e2[2], Jp[6], Jl[4];
                bearing_range_factor(pose + 3 * i, lm + 2 * j, mb[k], mr[k], e2, Jp, Jl);
                for (int r = 0; r < 2; ++r) {
                    const double v = (e2[r] + ((Jp[3 * r] * dp[3 * i] + Jp[3 * r + 1] * dp[3 * i + 1]) + Jp[3 * r + 2] * dp[3 * i + 2])) + (Jl[2 * r] * dl[2 * j] + Jl[2 * r + 1] * dl[2 * j + 1]);
                    tot += 0.5 * v * v;
                }
            }
        }
        return tot;
    }

    void retract(const double* pose, const double* lm, const double* dp, const double* dl, double* pose_n, double* lm_n) const {
        const int n = N();
        for (int i = 0; i < n; ++i) {
            double s, c;
            sc(pose[3 * i + 2], &s, &c);
            pose_n[3 * i] = pose[3 * i] + (c * dp[3 * i] - s * dp[3 * i + 1]);
            pose_n[3 * i + 1] = pose[3 * i + 1] + (s * dp[3 * i] + c * dp[3 * i + 1]);
            pose_n[3 * i + 2] = remainder(pose[3 * i + 2] + dp[3 * i + 2], slam::kTwoPi);
        }
        for (int a = 0; a < 2 * M; ++a) lm_n[a] = lm[a] + dl[a];
    }

    // PoseGraph::solvePoseGraph (pose_graph.cpp:269-300): LevenbergMarquardtOptimizer(graph, initial_estimate).optimize()
    LmStats solve(int lin_mode) {
        LmStats st;
        const int n = N();
        std::vector<double> pose(pose0.begin(), pose0.begin() + 3 * n), lm(lm0.begin(), lm0.begin() + 2 * std::max(M, 1));
        std::vector<double> pose_n(pose.size()), lm_n(lm.size()), dp, dl;
        double lambda = 1e-5;
        const double lambdaFactor = 10.0, lambdaUpper = 1e5, minFidelity = 1e-3, relTol = 1e-5, absTol = 1e-5;
        const int maxIter = 100;
        Lin L;
        double error = cost(pose.data(), lm.data());
        st.err_init = error;
        for (;;) {   // defaultOptimize: do { currentError = error(); iterate(); } while (...)
            const double currentError = error;
            linearize(pose.data(), lm.data(), L);
            for (;;) {   // iterate(): while (!tryLambda()) {}
                st.trials += 1;
                bool success = false, stop = false;
                const int lmode = lin_mode & 0xff, seg_len = (lin_mode >> 8) > 0 ? (lin_mode >> 8) : 32;
                const bool ok = lmode == LIN_DENSE ? solve_dense(L, lambda, dp, dl)
                                : (lmode == LIN_SEG ? solve_seg(L, lambda, seg_len, dp, dl) : solve_schur(L, lambda, dp, dl));
                double newError = 0.0, fidelity = 0.0;
                if (ok) {
                    const double oldLin = L.err;                       // linear.error(zero)
                    const double newLin = linear_error(pose.data(), lm.data(), dp.data(), dl.data());
                    const double linChange = oldLin - newLin;
                    if (linChange >= 0.0) {
                        retract(pose.data(), lm.data(), dp.data(), dl.data(), pose_n.data(), lm_n.data());
                        newError = cost(pose_n.data(), lm_n.data());
                        const double costChange = error - newError;
                        if (linChange > 2.220446049250313e-16 * oldLin) {
                            fidelity = costChange / linChange;
                            success = fidelity > minFidelity;
                        }
                        if (fabs(costChange) < relTol * error) stop = true;
                    }
                }
                if (success) {   // decreaseLambda (useFixedLambdaFactor)
                    lambda = lambda / lambdaFactor;
                    pose.swap(pose_n); lm.swap(lm_n);
                    error = newError;
                    st.iterations += 1;
                    break;
                } else if (!stop) {
                    lambda = lambda * lambdaFactor;
                    if (lambda >= lambdaUpper) break;
                } else {
                    break;
                }
            }
            if (!std::isfinite(error)) { st.flags |= PGS_FLAG_NONFINITE; break; }
            if (st.iterations >= maxIter) { st.flags |= PGS_FLAG_NOT_CONVERGED; break; }
            const double absDec = currentError - error, relDec = absDec / currentError;   // checkConvergence
            if (error <= 0.0 || relDec <= relTol || absDec <= absTol) break;
            if (!std::isfinite(currentError)) break;
        }
        st.err_final = error; st.lambda = lambda;
        std::copy(pose.begin(), pose.end(), pose1.begin());
        std::copy(lm.begin(), lm.begin() + 2 * M, lm1.begin());
        solved = true;
        flags |= st.flags;
        return st;
    }
    // `this->initial_estimate = this->result` (pose_graph.cpp:263)
    void adopt() { pose0 = pose1; lm0 = lm1; }
};

// compute_average_error as the pose-graph plot calls it (plotting_node.py:432-434,203-213): pose i of the message
// (i < timestep, float32 on the wire) against true_poses[i], i.e. the true pose AFTER step i+1.
double pgs_avg_error(const double* pose, const double* truth_xy, int timestep) {
    double sum = 0.0;
    for (int i = 0; i < timestep; ++i) {
        const double ex = (double)(float)pose[3 * i] - truth_xy[2 * i], ey = (double)(float)pose[3 * i + 1] - truth_xy[2 * i + 1];
        sum = sum + ::sqrt(ex * ex + ey * ey);
    }
    return timestep > 0 ? sum / timestep : 0.0;
}

}  // namespace

extern "C" {

void* orc_pgs_create(const slam_config* cfg, int N_max, int L_max, int KP, int math) { return new Pgs(*cfg, N_max, L_max, KP, math); }
void orc_pgs_destroy(void* h) { delete (Pgs*)h; }
void orc_pgs_init(void* h, float x0, float y0, float yaw0) { ((Pgs*)h)->init(x0, y0, yaw0); }
void orc_pgs_set_secondary(void* h, const double* sv) { ((Pgs*)h)->set_secondary(sv); }
int orc_pgs_update(void* h, float fwd, float ang, const float* meas, int k) { return ((Pgs*)h)->update(fwd, ang, meas, k); }
// out: iterations, trials, flags ; dout: err_init, err_final, lambda
void orc_pgs_solve(void* h, int lin_mode, int* out, double* dout) {
    LmStats st = ((Pgs*)h)->solve(lin_mode);
    if (out) { out[0] = st.iterations; out[1] = st.trials; out[2] = st.flags; }
    if (dout) { dout[0] = st.err_init; dout[1] = st.err_final; dout[2] = st.lambda; }
}
void orc_pgs_adopt(void* h) { ((Pgs*)h)->adopt(); }
// which: 0 = initial_estimate, 1 = result
void orc_pgs_get(void* h, int which, double* poses, double* lms, int* timestep, int* M, int* ids) {
    Pgs* p = (Pgs*)h;
    const std::vector<double>& ps = which ? p->pose1 : p->pose0;
    const std::vector<double>& ls = which ? p->lm1 : p->lm0;
    if (poses) memcpy(poses, ps.data(), sizeof(double) * 3 * p->N());
    if (lms) memcpy(lms, ls.data(), sizeof(double) * 2 * p->M);
    if (timestep) *timestep = p->timestep;
    if (M) *M = p->M;
    if (ids) memcpy(ids, p->ids.data(), sizeof(int) * p->M);
}
// msg_measurement_connections (pose_graph.cpp:176-177): pairs (timestep, lm_index) with lm_index = -1 for the first
// detection of a landmark (getLandmarkIndexFromID returns -1 then).  Returns the number of pairs.
int orc_pgs_connections(void* h, int* conn, int cap) {
    Pgs* p = (Pgs*)h;
    int nc = 0;
    for (int i = 0; i <= p->timestep; ++i)
        for (int s = 0; s < p->cnt[i]; ++s) {
            const size_t k = (size_t)i * p->KP + s;
            if (nc < cap) { conn[2 * nc] = i; conn[2 * nc + 1] = p->mfirst[k] ? -1 : p->mlm[k]; }
            nc += 1;
        }
    return nc;
}
double orc_pgs_cost(void* h, int which) {
    Pgs* p = (Pgs*)h;
    return which ? p->cost(p->pose1.data(), p->lm1.data()) : p->cost(p->pose0.data(), p->lm0.data());
}
// residual vector (whitened) of every factor at given values: order prior(3), then per pose i: between(i,i+1)(3),
// measurements (2 each).  Returns the length.  Used by the scipy cross-check and the finite-difference Jacobian test.
int orc_pgs_residuals(void* h, const double* poses, const double* lms, double* out, int cap) {
    Pgs* p = (Pgs*)h;
    std::vector<double> r;
    double e[3];
    p->prior_factor(poses, e);
    r.insert(r.end(), e, e + 3);
    const int n = p->N();
    for (int i = 0; i < n; ++i) {
        if (i + 1 < n) { p->between_factor(poses + 3 * i, poses + 3 * (i + 1), p->cmds[2 * i], p->cmds[2 * i + 1], e, nullptr); r.insert(r.end(), e, e + 3); }
        for (int s = 0; s < p->cnt[i]; ++s) {
            const size_t k = (size_t)i * p->KP + s;
            p->bearing_range_factor(poses + 3 * i, lms + 2 * p->mlm[k], p->mb[k], p->mr[k], e, nullptr, nullptr);
            r.insert(r.end(), e, e + 2);
        }
    }
    for (size_t i = 0; i < r.size() && (int)i < cap; ++i) out[i] = r[i];
    return (int)r.size();
}
// gradient J^T e of the objective at given values in tangent coordinates (poses [N][3], landmarks [M][2]), using the
// factor Jacobians above; its norm at the result measures stationarity.
void orc_pgs_gradient(void* h, const double* poses, const double* lms, double* gp, double* gl) {
    Pgs* p = (Pgs*)h;
    Pgs::Lin L;
    p->linearize(poses, lms, L);
    for (int i = 0; i < 3 * p->N(); ++i) gp[i] = -L.gp[i];
    for (int i = 0; i < 2 * p->M; ++i) gl[i] = -L.gl[i];
}
// retract helper for the finite-difference test
void orc_pgs_retract(void* h, const double* poses, const double* lms, const double* dp, const double* dl, double* poses_n, double* lms_n) {
    ((Pgs*)h)->retract(poses, lms, dp, dl, poses_n, lms_n);
}

// ---- batch runner: simulator (get_cmd) + NaiveFilter secondary (filter.h:342-348) + graph building + ONE solve at the
// end (solve_graph_every_iteration = false) for instances inst0..inst0+B-1; T commands => T+1 poses.
// Outputs (any may be NULL): pose_init/pose_res [B][T+1][3], lm_res [B][L_max][2], M_out [B], ids_out [B][L_max],
// istats [B][3] (iterations, trials, flags), dstats [B][3] (err_init, err_final, lambda), avg_err [B][2] (initial,
// result; plotting_node.py alignment), truth_xy [B][T][2], meas_out [B][T][KP][3] + cnt_out [B][T] (the streams, so a
// test can feed the identical input to the GPU path).  Returns seconds spent in solve() only.
// every_iteration != 0: solve_graph_every_iteration (params.yaml:64; pose_graph.cpp:258-264) - after every update the graph is solved
// and `initial_estimate = result`; istats then holds the LM iterations / lambda trials SUMMED over the T ticks, tick_counts (optional,
// [B][T][2]) those of every tick, pose_init the last adopted estimate, and the returned seconds cover all T solves.
double orc_run_pgs_batch_ex(const slam_config* cfg, int L_max, int KP, int math, int lin_mode, const double* map_xy, int L,
                            const float* cmds, int T, uint64_t seed, int64_t inst0, int B, int nthreads,
                            double* pose_init, double* pose_res, double* lm_res, int* M_out, int* ids_out, int* istats,
                            double* dstats, double* avg_err, double* truth_xy, float* meas_out, int* cnt_out,
                            int every_iteration, int* tick_counts) {
    std::atomic<int> next(0);
    std::atomic<long long> solve_ns(0);
    const int N = T + 1;
    auto worker = [&]() {
        std::vector<float> meas((size_t)3 * std::max(L, 1));
        std::vector<double> truth((size_t)2 * T);
        for (;;) {
            const int b = next.fetch_add(1);
            if (b >= B) break;
            Pgs g(*cfg, N, L_max, KP, math);
            g.init((float)cfg->init_x, (float)cfg->init_y, (float)cfg->init_yaw);
            Sim sim;
            sim.cfg = *cfg; sim.L = L; sim.math = math; sim.map.assign(map_xy, map_xy + 2 * L);
            sim.xv[0] = cfg->init_x; sim.xv[1] = cfg->init_y; sim.xv[2] = cfg->init_yaw;
            double nv[3] = {(double)(float)cfg->init_x, (double)(float)cfg->init_y, (double)(float)cfg->init_yaw};
            int it_sum = 0, tr_sum = 0;
            LmStats last{};
            for (int t = 0; t < T; ++t) {
                int k = 0;
                auto draw = [&](int pair, int which) { double u0, u1; slam::noise_pair(seed, (uint64_t)(inst0 + b), (uint32_t)t, (uint32_t)pair, &u0, &u1); return which ? u1 : u0; };
                if (math == MATH_DET) sim.step_t<DetMath>(cmds[2 * t], cmds[2 * t + 1], draw, meas.data(), nullptr, &k);
                else sim.step_t<LibmMath>(cmds[2 * t], cmds[2 * t + 1], draw, meas.data(), nullptr, &k);
                truth[2 * t] = sim.xv[0]; truth[2 * t + 1] = sim.xv[1];
                double s, c;   // NaiveFilter::update, filter.h:342-348
                g.sc(nv[2], &s, &c);
                nv[0] = nv[0] + (double)cmds[2 * t] * c;
                nv[1] = nv[1] + (double)cmds[2 * t] * s;
                nv[2] = remainder(nv[2] + (double)cmds[2 * t + 1], slam::kTwoPi);
                g.set_secondary(nv);
                g.update(cmds[2 * t], cmds[2 * t + 1], meas.data(), k);
                if (meas_out) {
                    float* mo = meas_out + ((size_t)b * T + t) * KP * 3;
                    for (int i = 0; i < 3 * std::min(k, KP); ++i) mo[i] = meas[i];
                }
                if (cnt_out) cnt_out[(size_t)b * T + t] = k;
                if (every_iteration) {   // pose_graph.cpp:258-264
                    const auto t0 = std::chrono::steady_clock::now();
                    const LmStats si = g.solve(lin_mode);
                    g.adopt();
                    solve_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
                    it_sum += si.iterations; tr_sum += si.trials;
                    if (tick_counts) { tick_counts[((size_t)b * T + t) * 2] = si.iterations; tick_counts[((size_t)b * T + t) * 2 + 1] = si.trials; }
                    last = si;
                }
            }
            LmStats st = last;
            if (!every_iteration) {
                const auto t0 = std::chrono::steady_clock::now();
                st = g.solve(lin_mode);
                solve_ns += std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t0).count();
            } else { st.iterations = it_sum; st.trials = tr_sum; }
            if (pose_init) memcpy(pose_init + (size_t)b * N * 3, g.pose0.data(), sizeof(double) * 3 * N);
            if (pose_res) memcpy(pose_res + (size_t)b * N * 3, g.pose1.data(), sizeof(double) * 3 * N);
            if (lm_res) memcpy(lm_res + (size_t)b * L_max * 2, g.lm1.data(), sizeof(double) * 2 * g.M);
            if (M_out) M_out[b] = g.M;
            if (ids_out) memcpy(ids_out + (size_t)b * L_max, g.ids.data(), sizeof(int) * g.M);
            if (istats) { istats[3 * b] = st.iterations; istats[3 * b + 1] = st.trials; istats[3 * b + 2] = g.flags; }
            if (dstats) { dstats[3 * b] = st.err_init; dstats[3 * b + 1] = st.err_final; dstats[3 * b + 2] = st.lambda; }
            if (avg_err) {
                avg_err[2 * b] = pgs_avg_error(g.pose0.data(), truth.data(), g.timestep);
                avg_err[2 * b + 1] = pgs_avg_error(g.pose1.data(), truth.data(), g.timestep);
            }
            if (truth_xy) memcpy(truth_xy + (size_t)b * T * 2, truth.data(), sizeof(double) * 2 * T);
        }
    };
    std::vector<std::thread> th;
    for (int i = 1; i < nthreads; ++i) th.emplace_back(worker);
    worker();
    for (auto& t : th) t.join();
    return (double)solve_ns.load() * 1e-9;
}

double orc_run_pgs_batch(const slam_config* cfg, int L_max, int KP, int math, int lin_mode, const double* map_xy, int L,
                         const float* cmds, int T, uint64_t seed, int64_t inst0, int B, int nthreads,
                         double* pose_init, double* pose_res, double* lm_res, int* M_out, int* ids_out, int* istats,
                         double* dstats, double* avg_err, double* truth_xy, float* meas_out, int* cnt_out) {
    return orc_run_pgs_batch_ex(cfg, L_max, KP, math, lin_mode, map_xy, L, cmds, T, seed, inst0, B, nthreads, pose_init, pose_res, lm_res, M_out,
                                ids_out, istats, dstats, avg_err, truth_xy, meas_out, cnt_out, 0, nullptr);
}

}  // extern "C"
