// pgs_lm_control.h — evaluation of a candidate, GTSAM's accept / lambda / convergence logic (pgs_decide_kernel, incl. the streaming refill and the asynchronous ticks), end / adopt / tick / average error.
// Part of pgs_kernel.hip (round 6: split by phase, pure moves); included there inside namespace slam { namespace {.  DESIGN.md 4.4.
#pragma once

// p * Pose2(v): the retraction of one pose (the same expressions wherever a candidate pose is formed)
__device__ __forceinline__ void retract_pose(const double* ps, const double* d, double out[3]) {
    double s, c;
    det_sincos(ps[2], &s, &c);
    out[0] = ps[0] + (c * d[0] - s * d[1]);
    out[1] = ps[1] + (s * d[0] + c * d[1]);
    out[2] = remainder(ps[2] + d[2], kTwoPi);
}

// Evaluation, part 1: one thread per FACTOR (see pgs_lin_factor_kernel) - its two terms of the linearised cost 0.5 |J delta + e|^2 at the
// current values and its term of the true cost at the candidate (the factor forms the candidate pose / landmark itself, with the
// expressions part 2 stores them with).  PF[3 slot ..] = {0.5 v_0^2, 0.5 v_1^2, 0.5 |e(candidate)|^2}; part 2 adds them where the one-kernel
// version added them (bit-identical sums).
__global__ __launch_bounds__(LF_TPB) void pgs_eval_factor_kernel(const PgsParams p) {
    const int nfb = (p.nfact_max + LF_TPB - 1) / LF_TPB;
    const int bl = blockIdx.x / nfb, fb = blockIdx.x - bl * nfb;
    const int b = pgs_slot(p, bl);
    if (p.state[b] || !p.solve_ok[b]) return;
    const int e = fb * LF_TPB + threadIdx.x;
    const int M = p.M[b], KP = p.KP;
    if (e >= p.evt_start[(size_t)b * (p.L_max + 1) + M]) return;
    const Inst g = inst_view(p, b);
    const int i = p.evt_pose[(size_t)b * p.N_max * KP + e];
    const size_t k = (size_t)p.evt_slot[(size_t)b * p.N_max * KP + e];
    const double* pose = p.pw + (size_t)b * p.N_max * 3 + 3 * i;
    const double* dp = p.dp + (size_t)b * p.N_max * 3 + 3 * i;
    const int j = g.mlm[k] & (kPgsFirstBit - 1);
    const double* lm = p.lw + (size_t)b * p.L_max * 2 + 2 * j;
    const double* dl = p.dl + (size_t)b * p.L_max * 2 + 2 * j;
    const double bb = g.mb[k], rr = g.mr[k];
    double e2[2], Jp[6], Jl[4];
    bearing_range_factor<true>(p, pose, lm, bb, rr, e2, Jp, Jl);
    // The three terms go COMPACT into the head of the slot's PF block, 24 bytes per factor slot (round 6; the linearisation's 96-byte records there are
    // dead once pgs_linearize_kernel has run): pgs_evaluate_kernel then reads a pose's terms as one contiguous run instead of three doubles out of every
    // 96-byte record - 27 -> 11 GB fetched per solve of 2048 graphs, that kernel 16.8 -> 10.5 ms (profiles/r06_pgs/summary.txt).  The writes here stay
    // scattered (threads run in event order, slots are pose-major): 24 bytes per dirtied line, 23 GB per solve.
    double* PF = p.PF + (size_t)b * p.N_max * KP * 12 + 3 * k;
#pragma unroll
    for (int r = 0; r < 2; ++r) {
        const double v = (e2[r] + ((Jp[3 * r] * dp[0] + Jp[3 * r + 1] * dp[1]) + Jp[3 * r + 2] * dp[2])) + (Jl[2 * r] * dl[0] + Jl[2 * r + 1] * dl[1]);
        PF[r] = 0.5 * v * v;
    }
    double pn[3], ln[2], en[2];
    retract_pose(pose, dp, pn);
    ln[0] = lm[0] + dl[0]; ln[1] = lm[1] + dl[1];
    bearing_range_factor<false>(p, pn, ln, bb, rr, en, nullptr, nullptr);
    PF[2] = 0.5 * (en[0] * en[0] + en[1] * en[1]);
}

// linearised cost of the step, retraction, true cost of the candidate (part 2: the prior / between factors and the sums); GTSAM's
// tryLambda / iterate / defaultOptimize decisions follow in pgs_decide_kernel (LevenbergMarquardtOptimizer.cpp, NonlinearOptimizer.cpp).
__global__ __launch_bounds__(TPB) void pgs_evaluate_kernel(const PgsParams p) {
    __shared__ double s_buf[TPB];
    const int b = pgs_slot(p, blockIdx.x), tid = threadIdx.x;
    if (p.state[b]) return;
    const int N = pgs_N(p, b), KP = p.KP, M = p.M[b];
    const Inst g = inst_view(p, b);
    double* pose = p.pw + (size_t)b * p.N_max * 3;
    double* lm = p.lw + (size_t)b * p.L_max * 2;
    double* pose_n = p.pn + (size_t)b * p.N_max * 3;
    double* lm_n = p.ln + (size_t)b * p.L_max * 2;
    const double* dp = p.dp + (size_t)b * p.N_max * 3;
    const double* dl = p.dl + (size_t)b * p.L_max * 2;
    const double* PFb = p.PF + (size_t)b * p.N_max * KP * 12;
    const bool ok = p.solve_ok[b] != 0;
    double newLin = 0.0, newError = 0.0;
    if (ok) {
        double acc = 0.0;
        for (int i = tid; i < N; i += TPB) {   // 0.5 |J delta + e|^2 of the UNDAMPED linearisation
            double e[3], J1[9];
            if (i == 0) {
                prior_factor(p, pose, e);
#pragma unroll
                for (int k = 0; k < 3; ++k) { const double v = e[k] + p.w_prior[k] * dp[k]; acc = acc + 0.5 * v * v; }
            }
            if (i + 1 < N) {
                between_factor<true>(p, pose + 3 * i, pose + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, J1);
#pragma unroll
                for (int r = 0; r < 3; ++r) {
                    const double v = (e[r] + ((J1[3 * r] * dp[3 * i] + J1[3 * r + 1] * dp[3 * i + 1]) + J1[3 * r + 2] * dp[3 * i + 2])) + p.w_btw[r] * dp[3 * (i + 1) + r];
                    acc = acc + 0.5 * v * v;
                }
            }
            const int kc = g.cnt[i];
            const double* PF = PFb + (size_t)i * KP * 3;   // (compact: three terms per factor slot, pgs_eval_factor_kernel)
            constexpr int UB = 8;   // the factors' terms (pgs_eval_factor_kernel) are fetched eight factors at a time, added in slot order
            int s = 0;
#pragma unroll 1
            for (; s + UB <= kc; s += UB) {
                double w[UB][2];
#pragma unroll
                for (int u = 0; u < UB; ++u) { w[u][0] = PF[3 * (size_t)(s + u)]; w[u][1] = PF[3 * (size_t)(s + u) + 1]; }
#pragma unroll
                for (int u = 0; u < UB; ++u) { acc = acc + w[u][0]; acc = acc + w[u][1]; }
            }
            for (; s < kc; ++s) { acc = acc + PF[3 * (size_t)s]; acc = acc + PF[3 * (size_t)s + 1]; }
            double pn[3];
            retract_pose(pose + 3 * i, dp + 3 * i, pn);
            pose_n[3 * i] = pn[0]; pose_n[3 * i + 1] = pn[1]; pose_n[3 * i + 2] = pn[2];
        }
        for (int a = tid; a < 2 * M; a += TPB) lm_n[a] = lm[a] + dl[a];
        newLin = block_sum<TPB>(acc, s_buf);
        __syncthreads();   // candidate values are visible to the block
        double acc2 = 0.0;   // the true cost of the candidate: block_cost with the factors' terms taken from PF
        for (int i = tid; i < N; i += TPB) {
            double e[3], pc = 0.0;   // pose_cost's own accumulator: the pose's terms are summed first, then added to the thread's
            if (i == 0) {
                prior_factor(p, pose_n, e);
                pc = pc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
            }
            if (i + 1 < N) {
                between_factor<false>(p, pose_n + 3 * i, pose_n + 3 * (i + 1), p.cmds[2 * i], p.cmds[2 * i + 1], e, nullptr);
                pc = pc + 0.5 * ((e[0] * e[0] + e[1] * e[1]) + e[2] * e[2]);
            }
            const int kc = g.cnt[i];
            const double* PF = PFb + (size_t)i * KP * 3 + 2;
            constexpr int UB = 8;
            int s = 0;
#pragma unroll 1
            for (; s + UB <= kc; s += UB) {
                double w[UB];
#pragma unroll
                for (int u = 0; u < UB; ++u) w[u] = PF[3 * (size_t)(s + u)];
#pragma unroll
                for (int u = 0; u < UB; ++u) pc = pc + w[u];
            }
            for (; s < kc; ++s) pc = pc + PF[3 * (size_t)s];
            acc2 = acc2 + pc;
        }
        newError = block_sum<TPB>(acc2, s_buf);
    }
    if (tid == 0) {   // the decision is pgs_decide_kernel's: it needs the slots of an instance in lambda order
        p.nok[b] = ok ? 1 : 0; p.nlin[b] = newLin; p.nerr[b] = newError;
        p.solve_ok[b] = 1;
    }
}

__global__ __launch_bounds__(TPB) void pgs_decide_kernel(const PgsParams p) {
    __shared__ int s_win, s_next;
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const bool running = p.state[b] == 0;
    if (p.async_ticks) {
        // asynchronous ticks: a graph whose next solve is prepared (state 4: pgs_lm_begin_kernel on the tick stream, complete before this
        // launch) joins the next trial's list; the counters the host sizes the coming grids from ride along
        if (tid == 0) {
            if (blockIdx.x == 0) { p.n_active[5] = p.mono[0]; p.n_active[6] = p.mono[1]; }
            const int stt = p.state[b];
            if (stt == 4) {
                p.state[b] = 0;
                atomicAdd(p.n_active, 1); atomicMax(p.n_active + 1, 1);
                p.alist[atomicAdd(p.n_active + 2, 1)] = b;
            } else if (stt == 3 || stt == 5 || stt == 6) atomicAdd(p.n_active, 1);   // between two solves: still counts as unfinished
        }
        if (!running) return;
    }
    if (!running && p.slots_cap <= 0) return;
    const int N = pgs_N(p, b), M = p.M[b], B = p.B;
    if (running) {
    if (tid == 0) {
        const double lambdaFactor = 10.0, lambdaUpper = 1e5, minFidelity = 1e-3, relTol = 1e-5, absTol = 1e-5;
        const int maxIter = 100;
        double lambda = p.lambda[b], error = p.error[b];
        int iters = p.iters[b], trials = p.trials[b];
        const int nl = p.nl[b] > 0 ? p.nl[b] : 1;
        int win = -1, done = 0, fl = 0;
        bool end_inner = false;
        for (int j = 0; j < nl && !end_inner; ++j) {
            const int sl = j * B + b;
            const bool ok = p.nok[sl] != 0;
            const double newLin = p.nlin[sl], newError = p.nerr[sl];
            bool success = false, stop = false;
            if (ok) {
                const double oldLin = error;
                const double linChange = oldLin - newLin;
                if (linChange >= 0.0) {
                    const double costChange = error - newError;
                    if (linChange > 2.220446049250313e-16 * oldLin) success = (costChange / linChange) > minFidelity;
                    if (fabs(costChange) < relTol * error) stop = true;
                }
            }
            trials += 1;
            if (success) {
                lambda = lambda / lambdaFactor; error = newError; iters += 1; end_inner = true; win = j;
            } else if (!stop) {
                lambda = lambda * lambdaFactor;
                if (lambda >= lambdaUpper) end_inner = true;
            } else {
                end_inner = true;
            }
        }
        if (end_inner) {   // defaultOptimize's loop condition
            const double currentError = p.cur_error[b];
            const double absDec = currentError - error, relDec = absDec / currentError;
            if (!(fabs(error) <= 1.79769313486231570e308)) { done = 1; fl = PGS_FLAG_NONFINITE; }
            else if (iters >= maxIter) { done = 1; fl = PGS_FLAG_NOT_CONVERGED; }
            else if (error <= 0.0 || relDec <= relTol || absDec <= absTol) done = 1;
            else p.cur_error[b] = error;
        }
        // the trial cap is per GRAPH: where the host's count of launches is not a graph's count of trials (asynchronous ticks, streaming) the decide
        // step applies it (lockstep: the host stops launching + pgs_lm_end_kernel)
        if ((p.async_ticks || p.slots_cap > 0) && !done && trials >= p.max_trials) { done = 1; fl = PGS_FLAG_NOT_CONVERGED; }
        atomicAdd(p.work + (p.seg_on ? 2 : (p.fused ? 1 : 0)), (double)(trials - p.trials[b]) * p.inst_flop[b]);   // reporting only
        p.lambda[b] = lambda; p.error[b] = error; p.iters[b] = iters; p.trials[b] = trials;
        // the next trial runs the next `lanes_next` lambdas of the sequence GTSAM would walk if every one of them failed:
        // lambda, 10 lambda, ... (lambda_j < lambdaUpper for j >= 1: reaching the bound ends the inner loop before that trial)
        int nnext = 1;
        if (!done) {
            double lj = lambda;
            const int want = p.lanes_next < p.lanes_max ? p.lanes_next : p.lanes_max;
            while (nnext < want) {
                lj = lj * lambdaFactor;
                if (lj >= lambdaUpper) break;
                p.lambda[nnext * B + b] = lj;
                nnext += 1;
            }
        }
        for (int j = 1; j < p.lanes_max; ++j) p.state[j * B + b] = (!done && j < nnext) ? 0 : 1;
        p.nl[b] = nnext;
        if (done) {
            p.flags[b] |= fl;
            if (p.async_ticks) { p.state[b] = 3; atomicAdd(p.n_active, 1); }   // parked until pgs_tick_kernel has advanced it (or finished it)
            else p.state[b] = 1;
        }
        else {
            atomicAdd(p.n_active, 1); atomicMax(p.n_active + 1, nnext);
            const int at = atomicAdd(p.n_active + 2, nnext);
            for (int j = 0; j < nnext; ++j) p.alist[at + j] = j * B + b;
        }
        s_win = win; s_next = done ? 0 : nnext;
    }
    __syncthreads();
    double* pose = p.pw + (size_t)b * p.N_max * 3;
    double* lm = p.lw + (size_t)b * p.L_max * 2;
    if (s_win >= 0) {   // accept the winning slot's candidate
        if (tid < p.lanes_max) p.lin_ok[(size_t)tid * B + b] = 0;   // the values change: every slot of the instance linearises anew
        const int sl = s_win * B + b;
        const double* pose_n = p.pn + (size_t)sl * p.N_max * 3;
        const double* lm_n = p.ln + (size_t)sl * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += TPB) pose[i] = pose_n[i];
        for (int a = tid; a < 2 * M; a += TPB) lm[a] = lm_n[a];
    }
    // clones that run in the next trial linearise at the instance's current values (after the accept above, if any: every
    // thread re-reads the elements it wrote itself)
    for (int j = 1; j < s_next; ++j) {
        double* cp = p.pw + (size_t)(j * B + b) * p.N_max * 3;
        double* cl = p.lw + (size_t)(j * B + b) * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += TPB) cp[i] = pose[i];
        for (int a = tid; a < 2 * M; a += TPB) cl[a] = lm[a];
    }
    }   // running
    if (p.slots_cap > 0) {
        // Streaming: the LAST workgroup of the launch to arrive here (every workgroup counts, also those of finished and waiting
        // graphs) refills the list: waiting graphs take the running slots this trial freed, in index order.  Which graph runs when
        // touches no result - a graph's LM sequence depends on nothing but the graph.
        __syncthreads();
        if (tid == 0) {
            __threadfence();
            const int arrived = atomicAdd(p.n_active + 3, 1);
            if (arrived == (int)gridDim.x - 1) {
                __threadfence();
                int nslots = atomicAdd(p.n_active + 2, 0), nact = atomicAdd(p.n_active, 0);
                int w = *p.wait_next;
                const int wend = p.b_off + p.b_cnt;
                while (nslots < p.slots_cap && w < wend) {
                    p.state[w] = 0;
                    p.alist[nslots] = w;
                    nslots += 1; nact += 1; w += 1;
                }
                *p.wait_next = w;
                p.n_active[0] = nact; p.n_active[2] = nslots; p.n_active[4] = w;
                if (nact > 0) atomicMax(p.n_active + 1, 1);
            }
        }
    }
}

// The lambda lanes' graph copies, on demand (round 6): until a solve's first trial with lanes the clones hold nothing; then the instances still
// listed get their arrays copied, one workgroup per (listed instance, lane).  Copying every instance's arrays for every lane when the solve began
// was ~100 device copies and 1 GB per solve - 4 % of a run of the every-iteration mode, where most ticks' solves need lanes for a handful of graphs.
__global__ __launch_bounds__(256) void pgs_clone_kernel(const PgsParams p, const PgsCloneTable t) {
    const int b = p.alist[blockIdx.x], j = blockIdx.y + 1, tid = threadIdx.x;
    if (b >= p.B) return;   // (a lane's slot in the list: lanes were on before - its instance is listed too)
    for (int a = 0; a < t.n; ++a) {
        const uint32_t w = t.words[a];
        const uint32_t* src = (const uint32_t*)t.ptr[a] + (size_t)b * w;
        uint32_t* dst = (uint32_t*)t.ptr[a] + ((size_t)j * p.B + b) * w;
        for (uint32_t i = tid; i < w; i += 256) dst[i] = src[i];
    }
}

// result <- current values (also for instances cut off by the trial cap)
__global__ __launch_bounds__(TPB) void pgs_lm_end_kernel(const PgsParams p) {
    const int b = blockIdx.x + p.b_off, tid = threadIdx.x;
    const int N = pgs_N(p, b), M = p.M[b];
    const double* pw = p.pw + (size_t)b * p.N_max * 3;
    const double* lw = p.lw + (size_t)b * p.L_max * 2;
    double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
    double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
    for (int i = tid; i < 3 * N; i += TPB) p1[i] = pw[i];
    for (int i = tid; i < 2 * M; i += TPB) l1[i] = lw[i];
    if (tid == 0 && p.state[b] != 1) { p.state[b] = 1; p.flags[b] |= PGS_FLAG_NOT_CONVERGED; }   // still running or still waiting at the trial cap
}

__global__ __launch_bounds__(TPB) void pgs_adopt_kernel(const PgsParams p) {
    const int b = blockIdx.x, tid = threadIdx.x;
    const int N = pgs_N(p, b), M = p.M[b];
    double* p0 = p.pose0 + (size_t)b * p.N_max * 3;
    double* l0 = p.lm0 + (size_t)b * p.L_max * 2;
    const double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
    const double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
    for (int i = tid; i < 3 * N; i += TPB) p0[i] = p1[i];
    for (int i = tid; i < 2 * M; i += TPB) l0[i] = l1[i];
    if (tid == 0 && p.tick_acc) { p.tick_acc[2 * b] += p.iters[b]; p.tick_acc[2 * b + 1] += p.trials[b]; }
    if (tid == 0 && p.tick_flop) {
        const double n = 2.0 * M, tr = (double)p.trials[b];
        p.tick_flop[2 * b] += tr * p.inst_flop[b];
        p.tick_flop[2 * b + 1] += tr * (n * n * n / 3.0 + 2.0 * n * n);
    }
}

// Asynchronous ticks: the step between two solves of ONE graph (pose_graph.cpp:258-264, then the next timer tick's :216-256).  result <-
// current values (pgs_lm_end_kernel), initial_estimate <- result (pgs_adopt_kernel), the sums over the ticks; then - unless the graph has
// reached T_end - the graph's next simulator tick, NaiveFilter::update and the append (pgs_run_sim_kernel's body for one timestep, with the
// graph's own timestep as the noise stream's step index).  State 3 (solve converged) / 5 (first tick: nothing to adopt) -> 6, or 1 = finished.
__global__ __launch_bounds__(256) void pgs_tick_kernel(const PgsParams p) {
    constexpr int KCAP = 256;   // (every detection reaches append_step: pgs_run_sim_kernel)
    __shared__ float s_meas[3 * KCAP];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int st = p.state[b];
    if (st != 3 && st != 5) return;
    const int N = p.Nv[b], M = p.M[b];
    if (st == 3) {
        const double* pw = p.pw + (size_t)b * p.N_max * 3;
        const double* lw = p.lw + (size_t)b * p.L_max * 2;
        double* p0 = p.pose0 + (size_t)b * p.N_max * 3;
        double* l0 = p.lm0 + (size_t)b * p.L_max * 2;
        double* p1 = p.pose1 + (size_t)b * p.N_max * 3;
        double* l1 = p.lm1 + (size_t)b * p.L_max * 2;
        for (int i = tid; i < 3 * N; i += 256) { const double v = pw[i]; p1[i] = v; p0[i] = v; }
        for (int i = tid; i < 2 * M; i += 256) { const double v = lw[i]; l1[i] = v; l0[i] = v; }
        if (tid == 0 && p.tick_acc) { p.tick_acc[2 * b] += p.iters[b]; p.tick_acc[2 * b + 1] += p.trials[b]; }
        if (tid == 0 && p.tick_flop) {
            const double n = 2.0 * M, tr = (double)p.trials[b];
            p.tick_flop[2 * b] += tr * p.inst_flop[b];
            p.tick_flop[2 * b + 1] += tr * (n * n * n / 3.0 + 2.0 * n * n);
        }
    }
    const int i = N - 1, t1 = N;   // the graph's timestep, the pose the tick adds
    if (i >= p.T_end || t1 >= p.N_max) {
        if (tid == 0) { p.state[b] = 1; if (i < p.T_end) p.flags[b] |= PGS_FLAG_POSE_CAP; }
        return;
    }
    if (tid >= 64) return;
    const int lane = tid;
    double tx = p.truth[3 * b], ty = p.truth[3 * b + 1], tth = p.truth[3 * b + 2];
    double lmx = 0.0, lmy = 0.0;
    if (lane < p.L) { lmx = p.map[2 * lane]; lmy = p.map[2 * lane + 1]; }
    const float fwd = p.cmds[2 * i], ang = p.cmds[2 * i + 1];
    int k = sim_wave<KCAP>(p, b, lane, fwd, ang, (uint32_t)i, tx, ty, tth, lmx, lmy, s_meas);
    if (k > KCAP) { k = KCAP; if (lane == 0) p.flags[b] |= PGS_FLAG_MEAS_CAP; }
    __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
    if (lane == 0) {
        double s, c;
        const double th = p.cur[3 * b + 2];
        det_sincos(th, &s, &c);
        p.cur[3 * b] = p.cur[3 * b] + (double)fwd * c;
        p.cur[3 * b + 1] = p.cur[3 * b + 1] + (double)fwd * s;
        p.cur[3 * b + 2] = remainder(th + (double)ang, kTwoPi);
        double* th_hist = p.truth_hist + ((size_t)b * p.N_max + (t1 - 1)) * 2;
        th_hist[0] = tx; th_hist[1] = ty;
        append_step(p, b, t1, s_meas, k);
        p.Nv[b] = N + 1;
        atomicMax(p.mono + 1, N + 1);
        p.state[b] = 6;
    }
}

// compute_average_error as the pose-graph plot calls it (plotting_node.py:203-213,432-434): pose i of the message
// (i < timestep, float32 on the wire) against true_poses[i] = the true pose after step i+1.
__global__ __launch_bounds__(TPB) void pgs_avg_error_kernel(const PgsParams p, int which, double* out) {
    __shared__ double s_buf[TPB];
    const int b = blockIdx.x, tid = threadIdx.x;
    const int ts = pgs_N(p, b) - 1;
    const double* pose = (which ? p.pose1 : p.pose0) + (size_t)b * p.N_max * 3;
    const double* th = p.truth_hist + (size_t)b * p.N_max * 2;
    double acc = 0.0;
    for (int i = tid; i < ts; i += TPB) {
        const double ex = (double)(float)pose[3 * i] - th[2 * i], ey = (double)(float)pose[3 * i + 1] - th[2 * i + 1];
        acc = acc + sqrt(ex * ex + ey * ey);
    }
    const double tot = block_sum<TPB>(acc, s_buf);
    if (tid == 0) out[b] = ts > 0 ? tot / ts : 0.0;
}
