"""Host-side model of the intra-workgroup protocol of the EKF step kernel's decoupled loop (ekf_kernel_impl.h: CONTROL /
STREAMERS), checked EXHAUSTIVELY over every interleaving of its LDS operations (one variant of the real
kernel deadlocked in round 2, commit 9eaa85c, and long random runs were the only coverage).

Agents and the shared words they use (names as in the kernel):
  * CONTROL (wavefront 0): per timestep waits for the generator, optionally gathers a row of P from HBM (hold = s_ring[6],
    waits for pass-in-flight = s_ring[7] to clear, reads applied = s_ring[1]), then publishes its updates into the ring of KG
    slots (waits while published - applied == KG; published = s_ring[0]); sets exit = s_ring[3] after the last step.
  * the pass LEADER (wavefront 1): polls applied / published / exit / hold, claims a pass (pass-in-flight = 1, re-checks hold and
    backs off), publishes the pass descriptor (s_pass[0..3]); fp32 storage: a pass must end where a timestep ends (s_wend).
  * every STREAMER (the leader included): waits for a new pass id, streams its share of P, bumps s_pass[3]; the leader then
    advances applied and clears pass-in-flight.  The LAST streamer also runs the measurement generator ahead of the control
    wavefront while it waits (s_sim[0] = generated, s_sim[1] = timestep the filter is at, ring of SD timesteps).

Each transition is ONE shared-memory operation (a sequentially consistent model: the wavefronts of a workgroup share one LDS,
whose operations from one wavefront stay in order, and the kernel puts seq_cst fences around the hold / in-flight handshake).
Checked: (i) no reachable state from which the all-terminated state is unreachable (deadlock / livelock trap); (ii) a gather
never overlaps a pass and sees applied == the updates HBM really holds; (iii) a ring slot is never rewritten while a pass may
read it; (iv) at termination everything published has been applied.  The checker is validated on the configuration that hung
the GPU (KG smaller than the pass threshold without the clamp): it must FIND that deadlock.
"""
from collections import deque

import pytest

# program counters
C_TOP, C_WAITGEN, C_HOLD, C_WAITPASS, C_READAPP, C_GATHER, C_UNHOLD, C_SLOT, C_PUBLISH, C_EXIT, C_DONE = range(11)
(L_APP, L_PUB, L_EXIT, L_DECIDE, L_RECHECK, L_PREFIX, L_CLAIM, L_BACKOFF, L_POST, S_WAIT, S_STREAM, S_END, S_LEADWAIT, S_DONE) = range(14)


def explore(KG, pass_min, NS, steps, f32, SD=2, k_choices=None, clamp=True, max_states=3_000_000, mutant=None):
    """Breadth-first search of the whole state space.  Returns (number of states, list of violations)."""
    PM = min(pass_min, KG) if clamp else pass_min
    ks = sorted(set(k_choices if k_choices is not None else (0, 1, KG)))
    gen_agent = NS - 1            # the last streamer generates (the leader itself when it is the only streamer)

    # state = (shared, control, streamers) with
    #   shared   = (pub, app, exit, hold, infl, p0, p1, p2, p3, wend, gen, cur, hb)
    #   control  = (pc, si, ul, a, plan)            plan = (k, gather) chosen for the current step
    #   streamer = (pc, seen, lo, cnt, la, lp, lex, stop)   (la.. only used by the leader = streamer 0)
    init = ((0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0, 0), (C_TOP, 0, 0, 0, (0, 0)), tuple((L_APP if i == 0 else S_WAIT, 0, 0, 0, 0, 0, 0, 0) for i in range(NS)))
    violations = []

    def succ(state):
        sh, ct, st = state
        pub, app, ex, hold, infl, p0, p1, p2, p3, wend, gen, cur, hb = sh
        out = []

        def mk(**kw):
            d = dict(pub=pub, app=app, ex=ex, hold=hold, infl=infl, p0=p0, p1=p1, p2=p2, p3=p3, wend=wend, gen=gen, cur=cur, hb=hb)
            d.update(kw)
            return (d["pub"], d["app"], d["ex"], d["hold"], d["infl"], d["p0"], d["p1"], d["p2"], d["p3"], d["wend"], d["gen"], d["cur"], d["hb"])

        # ---------------------------------------------------------------- CONTROL
        pc, si, ul, a, plan = ct
        if pc == C_TOP:
            if si == steps:
                out.append((sh, (C_EXIT, si, 0, a, plan), st))
            else:
                for k in ks:
                    for g in (0, 1):
                        out.append((mk(cur=si), (C_WAITGEN, si, k, a, (k, g)), st))
        elif pc == C_WAITGEN:           # the measurements of timestep si come from the generator wavefront
            if gen > si:
                out.append((sh, (C_HOLD if plan[1] else C_SLOT, si, ul, a, plan), st))
            else:
                out.append(state)
        elif pc == C_HOLD:
            out.append((mk(hold=1), (C_WAITPASS, si, ul, a, plan), st))
        elif pc == C_WAITPASS:
            out.append((sh, (C_READAPP, si, ul, a, plan), st) if (infl == 0 or mutant == "no_wait_for_pass_in_flight") else state)
        elif pc == C_READAPP:
            if app != hb:
                violations.append(("gather: applied counter differs from what HBM holds", state))
            out.append((sh, (C_GATHER, si, ul, app, plan), st))
        elif pc == C_GATHER:            # reading the row / column from HBM, applying the pending updates a .. pub-1 from the ring
            out.append((sh, (C_UNHOLD, si, ul, a, plan), st))
        elif pc == C_UNHOLD:
            out.append((mk(hold=0), (C_SLOT, si, ul, a, plan), st))
        elif pc == C_SLOT:
            if ul == 0:
                out.append((sh, (C_TOP, si + 1, 0, a, plan), st))
            elif pub - app < KG:
                out.append((sh, (C_PUBLISH, si, ul, a, plan), st))
            else:
                out.append(state)
        elif pc == C_PUBLISH:
            if pub - app >= KG:
                violations.append(("ring slot rewritten while pending", state))
            slot = pub % KG
            w = (wend | (1 << slot)) if ul == 1 else (wend & ~(1 << slot))
            out.append((mk(pub=pub + 1, wend=w), (C_SLOT, si, ul - 1, a, plan), st))
        elif pc == C_EXIT:
            out.append((mk(ex=1), (C_DONE, si, 0, a, plan), st))

        # ---------------------------------------------------------------- STREAMERS (0 = pass leader)
        for i in range(NS):
            spc, seen, lo, cnt, la, lp, lex, stop = st[i]

            def put(new, shared=sh):
                out.append((shared, ct, st[:i] + (new,) + st[i + 1:]))

            def maybe_generate():
                """the generator runs in this wavefront's waiting loop: one timestep ahead if the ring of SD slots has room"""
                if i == gen_agent and gen < steps and gen < cur + SD and not ex:
                    put(st[i], mk(gen=gen + 1))
                    return True
                return False

            if spc == L_APP:
                put((L_PUB, seen, lo, cnt, app, lp, lex, stop))
            elif spc == L_PUB:
                put((L_EXIT, seen, lo, cnt, la, pub - la, lex, stop))
            elif spc == L_EXIT:
                put((L_DECIDE, seen, lo, cnt, la, lp, ex, stop))
            elif spc == L_DECIDE:
                if lp > 0 and (lp >= PM or lex) and not hold:
                    put((L_PREFIX, seen, lo, cnt, la, lp, lex, 0))
                elif lex and lp == 0:
                    put((L_RECHECK, seen, lo, cnt, la, lp, lex, stop))
                else:
                    if not (NS == 1 and maybe_generate()):
                        pass
                    put((L_APP, seen, lo, cnt, la, lp, lex, stop))   # s_sleep, poll again
            elif spc == L_RECHECK:
                if pub - la == 0:
                    put((L_POST, seen, lo, cnt, la, lp, lex, 1))
                else:
                    put((L_APP, seen, lo, cnt, la, lp, lex, stop))
            elif spc == L_PREFIX:
                c = min(lp, KG)
                if f32:
                    while c > 0 and not (wend >> ((la + c - 1) % KG)) & 1:
                        c -= 1
                if c == 0:
                    put((L_APP, seen, lo, cnt, la, lp, lex, stop))
                else:
                    put((L_CLAIM, seen, lo, c, la, lp, lex, stop))
            elif spc == L_CLAIM:
                put((L_BACKOFF, seen, lo, cnt, la, lp, lex, stop), mk(infl=1))
            elif spc == L_BACKOFF:
                if hold and mutant != "no_backoff":
                    put((L_APP, seen, lo, cnt, la, lp, lex, stop), mk(infl=0))
                else:
                    put((L_POST, seen, lo, cnt, la, lp, lex, stop))
            elif spc == L_POST:
                put((S_WAIT, seen, lo, cnt, la, lp, lex, stop), mk(p1=la, p2=(-1 if stop else cnt), p3=0, p0=seen + 1))
            elif spc == S_WAIT:
                if p0 > seen:
                    put((S_DONE if p2 < 0 else S_STREAM, seen + 1, p1, p2, la, lp, lex, stop))
                else:
                    if not (NS > 1 and maybe_generate()):
                        out.append(state)
            elif spc == S_STREAM:       # streams its share of P (reads ring slots lo .. lo+cnt-1)
                put((S_END, seen, lo, cnt, la, lp, lex, stop))
            elif spc == S_END:
                nhb = lo + cnt if p3 + 1 == NS else hb
                put((S_LEADWAIT if i == 0 else S_WAIT, seen, lo, cnt, la, lp, lex, stop), mk(p3=p3 + 1, hb=nhb))
            elif spc == S_LEADWAIT:
                if p3 == NS:
                    put((L_APP, seen, lo, cnt, la, lp, lex, stop), mk(app=lo + cnt, infl=0))
                else:
                    out.append(state)
        return out

    def check(state):
        sh, ct, st = state
        gathering = ct[0] in (C_GATHER, C_UNHOLD)       # between reading `applied` and releasing the hold
        streaming = any(s[0] in (S_STREAM, S_END) for s in st)
        if gathering and streaming:
            violations.append(("a pass streams P while the control wavefront gathers from it", state))
        if sh[1] > sh[0]:
            violations.append(("applied > published", state))

    def terminal(state):
        sh, ct, st = state
        return ct[0] == C_DONE and all(s[0] == S_DONE for s in st)

    seen_states = {init: 0}
    order = [init]
    edges = []            # (from, to) indices
    q = deque([init])
    while q:
        s = q.popleft()
        si = seen_states[s]
        check(s)
        if terminal(s):
            sh = s[0]
            if not (sh[0] == sh[1] == sh[12]):
                violations.append(("terminated with published != applied", s))
            continue
        for t in succ(s):
            ti = seen_states.get(t)
            if ti is None:
                ti = len(order)
                seen_states[t] = ti
                order.append(t)
                q.append(t)
                if ti >= max_states:
                    raise RuntimeError("state space larger than expected")
            edges.append((si, ti))
    # liveness: every reachable state must be able to reach a terminal state
    n = len(order)
    rev = [[] for _ in range(n)]
    for a, b in edges:
        if a != b:
            rev[b].append(a)
    ok = [False] * n
    dq = deque(i for i, s in enumerate(order) if terminal(s))
    for i in dq:
        ok[i] = True
    while dq:
        b = dq.popleft()
        for a in rev[b]:
            if not ok[a]:
                ok[a] = True
                dq.append(a)
    stuck = [order[i] for i in range(n) if not ok[i]]
    if stuck:
        violations.append((f"{len(stuck)} reachable states cannot reach termination (deadlock)", stuck[0]))
    return n, violations


@pytest.mark.parametrize("KG", [2, 3, 4, 5, 6])
@pytest.mark.parametrize("f32", [0, 1])
def test_decoupled_loop_protocol_has_no_deadlock_and_keeps_its_invariants(KG, f32):
    # the kernel's thresholds (ekf_kernel_impl.h, kPassMinCfg): fp64 passes start at SLAM_PASS_MIN = 4 pending updates, from six ring slots
    # on at KG - 1; fp32 storage at 3, from five slots on at KG - 2 (round 5); clamped to the ring size
    pass_min = (KG - 2 if KG > 4 else 3) if f32 else (KG - 1 if KG > 5 else 4)
    total = 0
    for NS, steps in ((1, 3), (2, 3), (3, 2)):          # W = 2, 3, 4 wavefronts per filter
        n, bad = explore(KG, pass_min, NS, steps, f32)
        assert not bad, (KG, f32, NS, bad[0][0], bad[0][1])
        total += n
    assert total > 1000


@pytest.mark.parametrize("pass_min", [1, 2])
def test_protocol_with_eager_passes(pass_min):
    """passes from the first pending update (the measured-and-rejected `passes from 2 pending updates` variants)"""
    n, bad = explore(4, pass_min, 2, 3, 0)
    assert not bad, bad[0]
    n, bad = explore(4, pass_min, 2, 3, 1)
    assert not bad, bad[0]


def test_the_checker_finds_the_deadlock_that_hung_the_gpu():
    """KG = 3 with a pass threshold of 4 and no clamp: the control wavefront waits for a ring slot, the leader for a fourth
    update that cannot be published (the KG = 3 sweep variant of round 2).  The model must report it."""
    n, bad = explore(3, 4, 2, 3, 0, clamp=False)
    assert any("deadlock" in b[0] for b in bad)


@pytest.mark.parametrize("mutant", ["no_backoff", "no_wait_for_pass_in_flight"])
def test_the_checker_rejects_broken_handshakes(mutant):
    """Mutants of the hold / pass-in-flight handshake (the leader does not re-check `hold` after claiming a pass; the control
    wavefront does not wait for a pass in flight): the model must see a pass streaming P during a gather."""
    n, bad = explore(4, 4, 2, 3, 0, mutant=mutant)
    assert any("gathers" in b[0] or "differs" in b[0] for b in bad), [b[0] for b in bad][:3]


def test_the_checker_finds_the_fp32_deadlock_of_steps_larger_than_the_ring():
    """Round 3 let the decoupled loop take timesteps of up to 2 KP detections (several groups per step).  fp32 storage cuts
    its passes where a timestep ends, so a step with more updates than the ring has slots leaves the control wavefront waiting
    for a slot and the leader for a step end: a deadlock.  The model with its old step sizes (k <= KG) could not see it; a
    random soak on the GPU did (tools/gpu_soak_ekf.py, SLAM_INST_WATCHDOG on the default fp32 kernels).  With k = KG + 1
    in the menu the checker must report it for fp32 storage - and only for fp32 storage (fp64 passes may end anywhere)."""
    n, bad = explore(4, 3, 2, 3, 1, k_choices=(0, 1, 4, 5))
    assert any("deadlock" in b[0] for b in bad)
    n, bad = explore(4, 4, 2, 3, 0, k_choices=(0, 1, 4, 5, 8))
    assert not bad, bad[0]
    # the fix (fastable: k <= min(2 KP, KG) for fp32 storage) = the step sizes the kernel now admits
    for KG in (2, 3, 4):
        n, bad = explore(KG, 3, 2, 3, 1, k_choices=tuple(range(KG + 1)))
        assert not bad, (KG, bad[0])
