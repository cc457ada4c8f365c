"""Host-side model of the EKF pre-step's rule for NEW landmark ids (live_ekf_slam_amd/csrc/ekf_kernel_impl.h, known-id association):
the kernel decides insertions, the freeze point and the capacity flag of a whole message with ballots; the reference decides them
detection by detection (ekf.cpp:99-108,141-173, restated in oracle/slam_oracle.cpp).  Both are re-stated here in plain Python and
compared on random messages, including the cases the round-3 soak found the old rule wrong on (a repeat of an id that found no
room; a capacity skip after the freeze point).  No GPU needed."""
import numpy as np


def sequential(ids_known, msg, L_max):
    """The reference's loop: returns (didx per detection or None when frozen, insertions, frozen, capacity)."""
    ids = list(ids_known); M0 = len(ids); cap = False; out = []
    for d in msg:
        i = ids.index(d) if d in ids else -1
        if i >= 0:
            if i >= M0:
                return None, 0, True, cap          # found among the ids pushed this step: x_t indexed out of range
            out.append(i)
        elif len(ids) >= L_max:
            cap = True; out.append(-1)
        else:
            ids.append(d); out.append(len(ids) - 1)
    return out, len(ids) - M0, False, cap


def parallel(ids_known, msg, L_max):
    """The kernel's formulation (one lane per detection, ballots as Python sets)."""
    M = len(ids_known); k = len(msg); room = L_max - M
    found = [ids_known.index(d) if d in ids_known else -1 for d in msg]
    isnew = [f < 0 for f in found]
    firstl = [min(q for q in range(l + 1) if msg[q] == msg[l]) for l in range(k)]
    isfirst = [isnew[l] and firstl[l] == l for l in range(k)]
    rankf = [sum(1 for q in range(firstl[l]) if isfirst[q]) for l in range(k)]
    insf = [rankf[l] < room for l in range(k)]
    fz = [l for l in range(k) if isnew[l] and not isfirst[l] and insf[l]]
    cm = [l for l in range(k) if isnew[l] and not insf[l]]
    lstar = fz[0] if fz else k
    cap = any(l < lstar for l in cm)
    if fz:
        return None, 0, True, cap
    didx = [found[l] if not isnew[l] else (M + rankf[l] if isfirst[l] and insf[l] else -1) for l in range(k)]
    return didx, min(sum(isfirst), max(room, 0)), False, cap


def test_parallel_new_id_rule_equals_the_reference_loop():
    rng = np.random.default_rng(7)
    for _ in range(20000):
        L_max = int(rng.integers(1, 9))
        M = int(rng.integers(0, L_max + 1))
        known = [int(v) for v in rng.choice(40, size=M, replace=False)]
        k = int(rng.integers(0, 12))
        pool = known + [int(v) for v in rng.integers(40, 46, 4)]
        msg = [int(rng.choice(pool)) if pool else 40 for _ in range(k)]
        assert parallel(known, msg, L_max) == sequential(known, msg, L_max), (known, msg, L_max)


def test_negative_ids_are_ordinary_ids():
    """ekf.cpp:99-108 compares ints: -1 and -2 are ids like any other (the register path of the pre-step once padded the unused lanes of
    lm_IDs with -2 and matched without a lane guard: a detection with id -2 'found' landmark M)."""
    rng = np.random.default_rng(11)
    for _ in range(5000):
        L_max = int(rng.integers(1, 9))
        M = int(rng.integers(0, L_max + 1))
        known = [int(v) for v in rng.choice(np.arange(-4, 12), size=M, replace=False)]
        k = int(rng.integers(0, 10))
        msg = [int(v) for v in rng.integers(-4, 12, k)]
        assert parallel(known, msg, L_max) == sequential(known, msg, L_max), (known, msg, L_max)
    assert sequential([5], [-2], 4) == ([1], 1, False, False) == parallel([5], [-2], 4)
    assert sequential([5, -2], [-2, -1], 4) == ([1, 2], 1, False, False) == parallel([5, -2], [-2, -1], 4)


def test_the_cases_the_soak_found():
    # a repeat of an id that found no room is skipped again: capacity, no freeze (the old rule froze)
    assert sequential([1, 2], [7, 7], 2) == ([-1, -1], 0, False, True) == parallel([1, 2], [7, 7], 2)
    # a repeat of an id this message inserted: freeze; the capacity skip AFTER it never happens (the old rule flagged it)
    assert sequential([1], [7, 7, 8], 2) == (None, 0, True, False) == parallel([1], [7, 7, 8], 2)
    # ... but a skip BEFORE the freeze point is flagged
    assert sequential([1], [7, 8, 7], 2) == (None, 0, True, True) == parallel([1], [7, 8, 7], 2)
    # a mapped landmark seen twice is two updates
    assert sequential([1, 2], [2, 2], 4) == ([1, 1], 0, False, False) == parallel([1, 2], [2, 2], 4)
