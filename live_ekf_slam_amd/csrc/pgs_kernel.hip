// pgs_kernel.hip — batched pose-graph SLAM for gfx950 (MI355X): graph building + Levenberg–Marquardt solve of B
// independent graphs (Monte-Carlo instances over one map / command sequence), one workgroup per instance.
//
// Reference path: PoseGraph::{init, updateNaiveVehPoseEstimate, update, onLandmarkMeasurement, solvePoseGraph}
// (ekf_ws/src/localization_pkg/src/pose_graph.cpp:68-300) with `implementation: gtsam` (params.yaml:61).  The solve
// is gtsam::LevenbergMarquardtOptimizer(graph, initial_estimate).optimize() with default parameters over
//   PriorFactor<Pose2> (pose 0), BetweenFactor<Pose2> (t, t+1), BearingRangeFactor<Pose2, Point2> (t, landmark).
// The factor definitions, the LM control flow and its constants are the ones documented in oracle/slam_oracle_pgs.cpp.
//
// MI355X design.  One LM trial (= one tryLambda of every active instance) is six launches:
//   linearize   per pose: 3x3 Hessian blocks A_i, C_i (block-tridiagonal H_pp), 3x2 pose-landmark blocks E_k,
//               gradient; per landmark: 2x2 block D_j, gradient (factors of one landmark are chained in a list)
//   chain       poses are eliminated FIRST: block-tridiagonal Cholesky of H_pp + lambda I fused with the forward
//               recurrence  Y_i = L_i^-1 (E_i - G_i Y_{i-1})  — one thread per landmark COLUMN, sequential over the
//               poses; Y (3N x (2M+1), last column = transformed gradient) goes to HBM once
//   syrk        Schur complement  S = D + lambda I - Y^T Y  on the landmarks with v_mfma_f64_16x16x4_f64 (the one
//               GEMM-shaped piece: 2 * 3N * (2M)^2 / 2 FLOP), 64x64 tiles, k range trimmed by first-detection pose
//   chol        dense blocked Cholesky of S (2M x 2M) + forward/backward substitution -> landmark step
//   backsolve   pose step from the chain factor (forward/backward over the block-bidiagonal factor)
//   evaluate    linearised and true cost of the candidate, retraction, GTSAM's accept / lambda / convergence logic
// All per-instance decisions live on the device; the host only polls the number of active instances.
#include "pgs_kernel.h"
#include "lds_attr.h"

#include <mutex>

#include "slam_math.h"
#include "slam_rng.h"
#include "sim_device.h"

namespace slam {
namespace {

#include "pgs_factors.h"
#include "pgs_graph.h"
#include "pgs_linearize.h"
#include "pgs_chain.h"
#include "pgs_syrk.h"
#include "pgs_chol.h"
#include "pgs_backsolve.h"
#include "pgs_lm_control.h"
#include "pgs_seg_impl.h"

}  // namespace

hipError_t pgs_launch_init(const PgsParams& p, float x0, float y0, float yaw0, hipStream_t s) {
    hipLaunchKernelGGL(pgs_init_kernel, dim3((p.B + 255) / 256), dim3(256), 0, s, p, (double)x0, (double)y0, (double)yaw0);
    return hipGetLastError();
}

hipError_t pgs_launch_append(const PgsParams& p, const float* d_meas, const int32_t* d_count, int k_stride, const double* d_sec, hipStream_t s) {
    hipLaunchKernelGGL(pgs_append_kernel, dim3((p.B + 63) / 64), dim3(64), 0, s, p, d_meas, d_count, k_stride, d_sec);
    return hipGetLastError();
}

hipError_t pgs_launch_run_sim(const PgsParams& p, int T, uint32_t step0, hipStream_t s) {
    hipLaunchKernelGGL(pgs_run_sim_kernel, dim3(p.B), dim3(64), 0, s, p, T, step0);
    return hipGetLastError();
}

hipError_t pgs_launch_seg_plan(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_seg_plan_kernel, dim3(p.B), dim3(256), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_lm_begin(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_lm_begin_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_trial_kernel(const PgsParams& p, int which, hipStream_t s) {
    const int nslot = pgs_nslot(p);   // slots covered: the instances of the group and their active lambda lanes, or the compacted list
    if (which == 6) {   // the decide kernel alone (split_decide): one workgroup per graph of the group, whatever the list holds
        hipLaunchKernelGGL(pgs_decide_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
        return hipGetLastError();
    }
    if (nslot <= 0) return hipSuccess;
    switch (which) {
    case 0:
        if (p.nfact_max > 0) hipLaunchKernelGGL(pgs_lin_factor_kernel, dim3(nslot * ((p.nfact_max + LF_TPB - 1) / LF_TPB)), dim3(LF_TPB), 0, s, p);
        hipLaunchKernelGGL(pgs_linearize_kernel, dim3(nslot), dim3(TPB), 0, s, p);
        break;
    case 1: {
        if (p.seg_on) {   // segmented elimination: the interiors of all segments side by side, then the separator chain
            const int nseg = seg_ns(p.N, p.seg_len) + 1;
            hipLaunchKernelGGL(pgs_seg_chain_kernel, dim3(nslot), dim3(64 * ((nseg + 63) / 64)), 0, s, p);   // the segments' 3x3 chains: a lane each
            hipLaunchKernelGGL(pgs_seg_kernel, dim3(nslot * nseg), dim3(SG_TPB), 0, s, p);                     // their column recurrences: a workgroup each
            if (nseg > 1) hipLaunchKernelGGL(pgs_sep_kernel, dim3(nslot), dim3(64 + p.LD), 0, s, p);
            break;
        }
        if (p.fused) {   // chain + SYRK in one launch, Y stays in LDS
            // 22 KB static + the Y buffers (89 KB at LD = 448) of the 160 KB: per device, checked (lds_attr.h)
            const void* fk = p.fused == 4 ? (const void*)pgs_chain_syrk_kernel<3, 4> : p.fused == 3 ? (const void*)pgs_chain_syrk_kernel<4, 3> : (const void*)pgs_chain_syrk_kernel<6, 2>;
            if (const hipError_t e = slam_allow_full_lds(fk); e != hipSuccess) return e;
            const size_t lds = sizeof(double) * 2 * FC_ROWS * (size_t)(p.LD + 16);
            // p.fused = workgroups per instance: the more, the fewer tiles (and MFMA time) per workgroup next to the recursion
            if (p.fused == 4) hipLaunchKernelGGL((pgs_chain_syrk_kernel<3, 4>), dim3(4 * nslot), dim3(FC_TPB), lds, s, p);
            else if (p.fused == 3) hipLaunchKernelGGL((pgs_chain_syrk_kernel<4, 3>), dim3(3 * nslot), dim3(FC_TPB), lds, s, p);
            else hipLaunchKernelGGL((pgs_chain_syrk_kernel<6, 2>), dim3(2 * nslot), dim3(FC_TPB), lds, s, p);
            break;
        }
        hipLaunchKernelGGL(pgs_chain_kernel, dim3(nslot), dim3(64 + p.LD), 0, s, p);
        break;
    }
    case 2: {
        if (p.seg_on) {   // the segments' Gram matrices side by side, then [D + lambda I; g_l^T] - Ysep^T Ysep - sum_p T_p by the tile kernel (its epilogue)
            PgsParams q = p;
            q.syrk_row0 = p.yr_sep; q.syrk_rows = 3 * seg_ns(p.N, p.seg_len); q.syrk_first = p.sep_first;
            const int nt = (p.LD + 63) / 64;
            hipLaunchKernelGGL(pgs_seg_gram_kernel, dim3(nslot * (seg_ns(p.N, p.seg_len) + 1)), dim3(GR_TPB), 0, s, p);
            hipLaunchKernelGGL(pgs_syrk_kernel<32>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, q);
            break;
        }
        if (p.fused) break;   // done by the chain launch
        if (p.syrk_wave_tile == 1) {   // instance-resident accumulators
            if (const hipError_t e = slam_allow_full_lds((const void*)pgs_syrk_inst_kernel); e != hipSuccess) return e;
            const size_t lds = sizeof(double) * 2 * SI_ROWS * (size_t)(p.LD + 16);
            hipLaunchKernelGGL(pgs_syrk_inst_kernel, dim3(8 * SI_NB * ((nslot + 7) / 8)), dim3(SI_TPB), lds, s, p);
            break;
        }
        if (p.syrk_wave_tile == 64) {
            const int nt = (p.LD + 127) / 128;
            hipLaunchKernelGGL(pgs_syrk_kernel<64>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, p);
        } else {
            const int nt = (p.LD + 63) / 64;
            hipLaunchKernelGGL(pgs_syrk_kernel<32>, dim3(8 * (nt * (nt + 1) / 2) * ((nslot + 7) / 8)), dim3(256), 0, s, p);
        }
        break;
    }
    case 3: {
        const size_t lds = sizeof(double) * (size_t)(p.LD + 16) * 17;   // panel rows x (NB + 1)
        // panels of L_max > 235 need more than the default 64 KiB of dynamic LDS (gfx950: 160 KiB); solve groups launch from several
        // host threads and a single-process host may hold several devices: per (kernel, device), checked (lds_attr.h)
        const void* ck = p.chol_threads == 256 ? (const void*)pgs_chol_kernel<256> : (p.chol_ll == 1 && p.LD <= 448) ? (const void*)pgs_chol_ll_kernel<1024>
                       : (p.chol_ll && p.LD <= 448) ? (const void*)pgs_chol_ll_kernel<768> : (const void*)pgs_chol_kernel<1024>;
        if (const hipError_t e = slam_allow_full_lds(ck); e != hipSuccess) return e;
        if (p.chol_threads == 256) { hipLaunchKernelGGL(pgs_chol_kernel<256>, dim3(nslot), dim3(256), lds, s, p); break; }
        if (p.chol_ll && p.LD <= 448) {   // left-looking: two consecutive panels [<= LD + 1][17] each, the staged block rows over the older one (its staging registers are sized for LD <= 448)
            const size_t lds_ll = sizeof(double) * 2 * (size_t)(p.LD + 1) * 17;
            if (p.chol_ll == 1) hipLaunchKernelGGL(pgs_chol_ll_kernel<1024>, dim3(nslot), dim3(1024), lds_ll, s, p);
            else hipLaunchKernelGGL(pgs_chol_ll_kernel<768>, dim3(nslot), dim3(768), lds_ll, s, p);
            break;
        }
        hipLaunchKernelGGL(pgs_chol_kernel<1024>, dim3(nslot), dim3(1024), lds, s, p);
        break;
    }
    case 4:
        if (p.seg_on && p.N <= kSegBackLdsPoses && !p.seg_back_global) {   // the chains out of LDS (12 doubles per pose)
            if (const hipError_t e = slam_allow_full_lds((const void*)pgs_seg_backsolve_lds_kernel); e != hipSuccess) return e;
            hipLaunchKernelGGL(pgs_seg_backsolve_lds_kernel, dim3(nslot), dim3(SBL_TPB), sizeof(double) * 12 * (size_t)p.N, s, p);
        } else if (p.seg_on) hipLaunchKernelGGL(pgs_seg_backsolve_kernel, dim3(nslot), dim3(SB_TPB), 0, s, p);
        else hipLaunchKernelGGL(pgs_backsolve_kernel, dim3(nslot), dim3(BTPB), 0, s, p);
        break;
    default:   // the candidates of every slot, then GTSAM's accept / lambda / convergence logic per instance
        if (p.nfact_max > 0) hipLaunchKernelGGL(pgs_eval_factor_kernel, dim3(nslot * ((p.nfact_max + LF_TPB - 1) / LF_TPB)), dim3(LF_TPB), 0, s, p);
        hipLaunchKernelGGL(pgs_evaluate_kernel, dim3(nslot), dim3(TPB), 0, s, p);
        if (!p.split_decide) hipLaunchKernelGGL(pgs_decide_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
        break;
    }
    return hipGetLastError();
}

hipError_t pgs_launch_tick(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_tick_kernel, dim3(p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(pgs_seg_plan_kernel, dim3(p.B), dim3(256), 0, s, p);
    hipLaunchKernelGGL(pgs_lm_begin_kernel, dim3(p.B), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_clone(const PgsParams& p, const PgsCloneTable& t, int lanes, hipStream_t s) {
    if (p.n_list <= 0 || lanes <= 1) return hipSuccess;
    hipLaunchKernelGGL(pgs_clone_kernel, dim3(p.n_list, lanes - 1), dim3(256), 0, s, p, t);
    return hipGetLastError();
}

hipError_t pgs_launch_lm_end(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_lm_end_kernel, dim3(p.b_cnt), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_adopt(const PgsParams& p, hipStream_t s) {
    hipLaunchKernelGGL(pgs_adopt_kernel, dim3(p.B), dim3(TPB), 0, s, p);
    return hipGetLastError();
}

hipError_t pgs_launch_avg_error(const PgsParams& p, int which, double* out, hipStream_t s) {
    hipLaunchKernelGGL(pgs_avg_error_kernel, dim3(p.B), dim3(TPB), 0, s, p, which, out);
    return hipGetLastError();
}

}  // namespace slam
