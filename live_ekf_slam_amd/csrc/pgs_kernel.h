// pgs_kernel.h — parameter block and launchers of the batched pose-graph SLAM solver (gfx950).
// Reference: ekf_ws/src/localization_pkg/src/pose_graph.cpp (PoseGraph with the GTSAM implementation): graph building
// (:68-256) and solvePoseGraph (:269-300) = gtsam::LevenbergMarquardtOptimizer with default parameters.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

namespace slam {

enum { PGS_FLAG_POSE_CAP = 1, PGS_FLAG_LM_CAP = 2, PGS_FLAG_MEAS_CAP = 4, PGS_FLAG_NOT_CONVERGED = 8, PGS_FLAG_NONFINITE = 16,
       PGS_FLAG_SEG_LIMIT = 32 /* asynchronous ticks: a segment of the graph sees more landmarks than the segmented elimination holds */ };
static constexpr int kPgsFirstBit = 1 << 30;   // mlm: this factor is the first detection of its landmark

// One workgroup per instance in every kernel; instance b owns slab b of every array (strides in elements).
struct PgsParams {
    int32_t B, N_max, L_max, KP, LD;   // LD = leading dimension of Y / S = roundup(2*L_max + 1, 64)
    int32_t N;                         // poses in the graph now (timestep + 1), the same for every instance
    int32_t b_off, b_cnt;              // LM kernels: the launch covers instances [b_off, b_off + b_cnt) (one solve group)
    int32_t chol_threads;              // 1024 or 256: workgroup size of the dense Cholesky of the next trial
    int32_t chol_ll;                   // 1: the 1024-thread Cholesky runs left-looking (pgs_chol_ll_kernel), 0: right-looking
    int32_t syrk_notrim;               // experiment: do not trim the k range (tiles of an instance then march in step)
    int32_t syrk_wave_tile;            // 64 or 32: SYRK variant of the next trial (chosen by the host from the active count)
    int32_t fused;                     // 0: chain and SYRK are two launches; 2 | 3 | 4: one (pgs_chain_syrk_kernel), that many workgroups per instance
    // ---- the graph (pose_graph.cpp: graph + initial_estimate + result) ----
    double* pose0; double* lm0;        // initial_estimate: [B][N_max][3], [B][L_max][2]
    double* pose1; double* lm1;        // result
    int32_t* ids; int32_t* M; int32_t* flags;       // lm_IDs [B][L_max], M [B], status [B]
    int32_t* cnt;                      // [B][N_max]      bearing-range factors attached to pose i
    int32_t* mlm;                      // [B][N_max*KP]   landmark index of factor (i, s) | kPgsFirstBit
    int32_t* mnext;                    // [B][N_max*KP]   next (newer) factor slot of the same landmark, -1 = none
    int32_t* lm_head;                  // [B][L_max]      oldest factor slot of landmark j (-1: none stored)
    int32_t* lm_last;                  // [B][L_max]      newest factor slot of landmark j
    int32_t* lm_first;                 // [B][L_max]      pose index of the first detection of landmark j
    double* mb; double* mr;            // [B][N_max*KP]   measured bearing / range (float32 wire values widened)
    const float* cmds;                 // [N_max][2]      BetweenFactor measurements Pose2(fwd, 0, ang), shared
    double* cur;                       // [B][3]          cur_veh_pose_estimate (secondary filter's pose)
    // ---- simulator (pgs_run_sim) ----
    double* truth;                     // [B][3]
    double* truth_hist;                // [B][N_max][2]   true (x, y) after step t at row t-1
    const double* map; int32_t L;
    double sV00, sV11, sW00, sW11, d_max, th_max, range_max, fov_min, fov_max;
    uint64_t seed; int64_t inst0;
    // ---- LM work space ----
    double* pw; double* lw;            // current values
    double* pn; double* ln;            // candidate values
    double* A; double* C; double* gp;  // [B][N_max*9], [B][N_max*9], [B][N_max*3]
    double* E;                         // [B][N_max*KP*6]
    double* Wl;                        // [B][N_max*KP*5]  per factor: Jl^T Jl (xx, xy, yy), -Jl^T e (x, y)
    // factors regrouped by (landmark, time) for the chain kernel; built by lm_begin
    int32_t* evt_start;                // [B][L_max+1]    first event of landmark j
    int32_t* evt_pose;                 // [B][N_max*KP]   pose index of event e
    int32_t* slot_pos;                 // [B][N_max*KP]   factor slot -> event position
    double* Elm;                       // [B][N_max*KP*6] E blocks in event order
    int32_t* evt_slot;                 // [B][N_max*KP]   factor slot of event e (the inverse of slot_pos)
    double* PF;                        // [S][N_max*KP*12] per factor slot: its share of the pose block (Jp^T Jp: 9, -Jp^T e: 3) between
                                       //                  pgs_lin_factor_kernel and pgs_linearize_kernel; later in the trial its cost terms
                                       //                  (pgs_eval_factor_kernel -> pgs_evaluate_kernel: three per factor slot, compact at the head of the slot's block)
    int32_t* lin_ok;                   // [S] the slot's linearisation (A, C, g_p, E, D, g_l) belongs to its current values: a trial after a
                                       //     FAILED one re-uses it, like GTSAM's inner lambda loop (pgs_decide_kernel clears it on an accept)
    int32_t* fact_cnt;                 // [B] factors of the instance (pgs_seg_plan_kernel), nfact_max: the largest (grid of the per-factor kernels)
    int32_t nfact_max;
    double* D; double* gl;             // [B][L_max*3], [B][L_max*2]
    double* Linv; double* G;           // [B][N_max*6], [B][N_max*9]
    double* Y;                         // [B][Yrows][LD]
    int64_t y_stride;                  // elements per instance in Y
    double* S;                         // [B][LD*LD]
    double* dl; double* dp;            // [B][L_max*2], [B][N_max*3]
    double* lambda; double* error; double* cur_error; double* err_init;
    int32_t* iters; int32_t* trials; int32_t* state; int32_t* solve_ok;   // state: 0 = active, 1 = done, 2 = waiting for a running slot (streaming);
                                       // asynchronous ticks: 3 = solve converged, 5 = first tick (nothing to adopt), 6 = tick appended, 4 = next solve prepared
    int32_t* n_active;                 // [4] per solve group: active instances after the trial, lanes the next trial needs, active SLOTS,
                                       //     (streaming) workgroups of the decide kernel that finished; [4]: copy of the wait cursor for the host
    // The slots that run in the next trial, compacted by pgs_decide_kernel (order = arrival order of its atomics; the mapping of
    // workgroups to instances does not touch any result).  A launch over the list has exactly one workgroup (or FC_NB / SI_NB) per
    // running slot with consecutive ids - the dispatcher deals ids round-robin over XCDs and shader engines, so a sparse set of
    // live ids in a full-size grid left engines idle while others queued (158 live workgroups of 512 took two rounds).
    int32_t* alist;                    // [b_cnt * lanes_max] of this solve group
    int32_t use_list, n_list;          // the launch's block index -> slot through alist[0 .. n_list) (else: lane * B + instance)
    // ---- streaming (round 6): graphs of a solve group wait for one of `slots_cap` RUNNING SLOTS ----
    // A group holds b_cnt graphs but only slots_cap of them are in flight; the rest wait in state 2.  The last workgroup of
    // pgs_decide_kernel to finish admits waiting graphs (in index order, cursor *wait_next) into the list of the next trial until it holds
    // slots_cap slots again, so every trial runs a full list while graphs wait.  The host then no longer decides anything per trial:
    // launches cover slots_cap slots, the LIST LENGTH is read on the device (n_list_dev = the counter block the previous trial's decide
    // kernel wrote; block ids beyond it map to `dead_slot`, whose state is 1 = every kernel returns at entry), and trials are enqueued ahead.
    int32_t slots_cap;                 // 0: lockstep (every instance of the group runs from the first trial on)
    int32_t dead_slot;                 // index of a slot that is never active
    int32_t* wait_next;                // [1] of this group: next waiting instance (b_off + b_cnt = none left)
    const int32_t* n_list_dev;         // NULL: the list length is n_list (host-known)
    // ---- speculative lambda lanes ----
    // Every instance b owns `lanes_max` slots of every per-instance array: slot b (the instance itself) and the clones
    // j * B + b, j = 1 .. lanes_max - 1.  After a failed tryLambda GTSAM multiplies lambda by 10 and tries again on the same
    // linearisation.  An instance runs the next `nl[b]` lambdas of that sequence at once, one per slot (the clones hold a copy
    // of its graph and values and are ordinary instances to kernels 0..4) - speculatively: slot j only matters if slots 0..j-1
    // fail - and pgs_decide_kernel replays GTSAM's sequential logic over the slots in lambda order: the outcome, the
    // iteration and the trial counts are those of the sequential loop.
    int32_t lanes_max;                 // slots per instance (1 = no speculation)
    int32_t lanes;                     // lanes covered by this launch of a trial kernel (grid = b_cnt * lanes)
    int32_t lanes_next;                // slots an instance may use in the NEXT trial (the host's choice: 1 while the GPU is busy with
                                       // many instances, lanes_max once few are left and the per-trial latency is what counts)
    int32_t* nl;                       // [B]   lanes instance b runs in the current trial
    double* nlin; double* nerr;        // [slots] linearised / true cost of the slot's candidate (pgs_evaluate_kernel)
    int32_t* nok;                      // [slots] the slot's linear solve succeeded
    double* inst_flop;                 // [B] algorithmic FLOP of ONE Schur-complement SYRK of instance b (pgs_lm_begin_kernel)
    double* work;                      // [2] algorithmic SYRK FLOP of the trials GTSAM's loop consumed so far in this solve, by path:
                                       //     [0] pgs_syrk_*_kernel launches, [1] pgs_chain_syrk_kernel (pgs_decide_kernel adds)
    // ---- segmented elimination of the pose chain (round 5, pgs_seg_impl.h) ----
    // seg_len = SL > 0: the poses k SL (k = 1 .. NS, NS = (N - 2) / SL) are SEPARATORS; segment p = 0 .. NS holds the poses strictly
    // between separators p and p + 1.  The interiors of all segments are eliminated first, independently (one workgroup each), then the
    // NS separators as a short chain, then the landmarks: depth SL + NS instead of N, and a row of Y only has the columns of the
    // landmarks its own segment sees.  seg_on: this solve runs that order (the host's choice, from the plan below).
    int32_t seg_len, seg_on, nseg_max; // nseg_max: segments the arrays are sized for
    int32_t seg_back_global;           // 1: the pose step's chains read their factor from global memory (test switch; graphs beyond 1365 poses always do)
    int32_t* seg_ncol;                 // [S][nseg_max]          landmarks seen from the segment's interior poses (its columns: 2 ncol + 1)
    int32_t* seg_lm;                   // [S][nseg_max * L_max]  local landmark -> landmark, ascending
    int32_t* seg_inv;                  // [S][nseg_max * L_max]  landmark -> local landmark of the segment, -1: not seen there
    int32_t* seg_evt;                  // [S][nseg_max * L_max]  local landmark -> its first event (evt_* order) at or after the segment's first pose
    int32_t* seg_blk;                  // [S][nseg_max * nb1]    local landmarks of the segment below landmark 16 * block (nb1 = seg_nb1(L_max) entries):
                                       //                        the local range of a 32-row block of S_ext without a search
    int32_t* sep_evt;                  // [S][nseg_max * L_max]  separator k, landmark j -> the landmark's first event AT the separator's pose, -1: none
    double* segT;                      // [S][nseg_max * seg_tld^2]  Gram matrix Y_p^T Y_p of the segment's columns (lower 16x16 tiles); allocated by the first solve that runs this order
    int32_t seg_tld;                   // its leading dimension: roundup(2 min(L_max, kPgsSegMaxLm) + 1, 16) <= 128
    int32_t* sep_first;                // [S][L_max]             first separator (0-based) whose row of Y can be non-zero in the landmark's columns
    int32_t* seg_umax;                 // [B]                    largest seg_ncol of the instance (the host picks the path from it)
    double* Gs;                        // [S][N_max * 9]         spike blocks: coupling of interior pose i to its segment's LEFT separator
    double* segout;                    // [S][nseg_max * 32]     per segment: sum Gs Gs^T (6) | Gright Gright^T (6) | Gright (9) | Gright Gs_e^T (9)
    double* sepfac;                    // [S][nseg_max * 16]     per separator: Linv (6) | G (9) of the separator chain
    int64_t yr_rc, yr_sep;             // rows of Y (leading dimension LD) where the segments' contributions to their separators' right-hand
                                       // sides ([nseg][6]: left 3, right 3, local columns) and the separators' rows of Y ([NS][3], global columns) live
    // the tile SYRK over an arbitrary block of Y rows: row offset, rows (-1: 3 N), per-landmark first non-zero row / 3 (NULL: lm_first)
    int64_t syrk_row0; int32_t syrk_rows; const int32_t* syrk_first;
    // ---- asynchronous ticks (round 6): solve_graph_every_iteration without a batch-wide barrier per tick ----
    // Graph b is at its OWN tick: Nv[b] poses.  When its solve converges pgs_decide_kernel parks it (state 3); pgs_tick_kernel - on a
    // second stream, beside the next trial of the others - stores the result, adopts it, runs the graph's next simulator tick and appends
    // it; pgs_seg_plan_kernel and pgs_lm_begin_kernel (same stream, only for graphs in state 6) prepare the next solve (state 4), and the
    // next decide kernel lists the graph again.  A graph is finished (state 1) at timestep T_end.  Kernels take a graph's pose count from
    // pgs_N(p, slot); p.N is then only the launch's upper bound (grid sizes, block-index decomposition, dynamic LDS).
    int32_t* Nv;                       // NULL: every graph has N poses (lockstep).  Else [slots + 1]
    int32_t async_ticks, T_end;        // T_end: the timestep at which a graph is finished
    int32_t split_decide;              // trial kernel 5 ends before pgs_decide_kernel (launched on its own as kernel 6)
    int32_t max_trials;                // async: lambda trials after which a graph's solve is cut off (NOT_CONVERGED); lockstep: the host's cap
    int32_t* mono;                     // [2] never reset during a run: most factors of a graph, most poses of a graph (the host sizes grids from them)
    int32_t* tick_acc;                 // optional [B][2]: pgs_adopt_kernel adds the solve's LM iterations / lambda trials (solve_graph_every_iteration: sums over the ticks)
    double* tick_flop;                 // optional [B][2]: ... and the algorithmic FLOP of its trials: Schur-complement SYRK (inst_flop) | dense Cholesky + substitutions (n^3/3 + 2 n^2, n = 2 M)
    unsigned long long* prof;          // optional [B][8] phase timers of the chol kernel (100 MHz wall clock), debug only
    // ---- factor constants ----
    double prior[3];
    double w_prior[3], w_btw[3], w_meas[2];   // 1 / sigma
};

// geometry of the segmented elimination: separators at the poses k SL, k = 1 .. seg_ns; segment ps holds the poses [seg_lo, seg_hi)
__host__ __device__ inline int seg_ns(int N, int SL) { return N >= 2 ? (N - 2) / SL : 0; }
__host__ __device__ inline int seg_lo(int ps, int SL) { return ps == 0 ? 0 : ps * SL + 1; }
__host__ __device__ inline int seg_hi(int ps, int SL, int NS, int N) { return ps < NS ? (ps + 1) * SL : N; }
__host__ __device__ inline int seg_nb1(int L_max) { return 4 * ((L_max + 63) / 64) + 2; }   // entries of a segment's seg_blk row (the block of
                                                                                               // the right-hand-side row, landmark index M <= L_max, has an end too)

hipError_t pgs_launch_init(const PgsParams& p, float x0, float y0, float yaw0, hipStream_t s);
// append one timestep: BetweenFactor is implied by cmds[t]; meas [B][k_stride][3], count [B] (device); sec_pose [B][3]
// (device) or NULL to keep `cur`.  p.N = number of poses BEFORE the call.
hipError_t pgs_launch_append(const PgsParams& p, const float* d_meas, const int32_t* d_count, int k_stride, const double* d_sec, hipStream_t s);
// T timesteps of simulator + NaiveFilter secondary + append, on the device (cmds already in p.cmds)
hipError_t pgs_launch_run_sim(const PgsParams& p, int T, uint32_t step0, hipStream_t s);
hipError_t pgs_launch_lm_begin(const PgsParams& p, hipStream_t s);
// the segments' column sets of every instance (from the graph alone; before pgs_launch_lm_begin of a solve)
hipError_t pgs_launch_seg_plan(const PgsParams& p, hipStream_t s);
static constexpr int kPgsSegMaxLen = 32;      // poses a segment holds at most (seg_len <= this)
static constexpr int kPgsSegMaxLm = 63;       // landmarks a segment's column set may hold for the segmented path (2 * 63 + 1 = 127 columns)
static constexpr int kPgsSegMaxSep = 128;     // separators the separator kernel stages in LDS
// one tryLambda for every active instance = kernels 0..5 in order: linearize, chain, syrk, chol, backsolve, evaluate (+ decide; with
// p.split_decide the decide kernel is launch 6 of its own)
static constexpr int kPgsTrialKernels = 6;
// asynchronous ticks: store + adopt the converged graphs' results, their next simulator tick + append, plan and begin of the next solve
hipError_t pgs_launch_tick(const PgsParams& p, hipStream_t s);
hipError_t pgs_launch_trial_kernel(const PgsParams& p, int which, hipStream_t s);
// The lambda lanes' copies of the graph / plan arrays of the instances in the CURRENT list (p.alist[0 .. p.n_list): instances, lanes off so far):
// slot b's share of every listed array to the slots j B + b, j = 1 .. lanes - 1.  Per-slot sizes in 4-byte words.
struct PgsCloneTable { void* ptr[28]; uint32_t words[28]; int32_t n; };
hipError_t pgs_launch_clone(const PgsParams& p, const PgsCloneTable& t, int lanes, hipStream_t s);
hipError_t pgs_launch_lm_end(const PgsParams& p, hipStream_t s);     // result = current values, flags
hipError_t pgs_launch_adopt(const PgsParams& p, hipStream_t s);      // initial_estimate = result
// avg position error (plotting_node.py:203-213 alignment) of initial (which = 0) / result (1) vs truth_hist: out [B]
hipError_t pgs_launch_avg_error(const PgsParams& p, int which, double* out, hipStream_t s);

}  // namespace slam
