#!/usr/bin/env python3
"""LDS bank-conflict model of the rotation phase of ukf_sqrt_kernel<44, 256> (the pass-table path), per array.

VERDICT r04 item 4 asks for the conflicts of the rotation phase (27.8 % of the kernel's LDS cycles, profiles/r04_ukf/pmc_summary_quad.txt)
to be attributed per array.  The addresses of every LDS access of a pass are a pure function of the schedule (jacobi_schedule.h) and of the
pass table (launch_ukf_quad_table), so the attribution can be computed instead of measured by ablation: this script rebuilds the table's
owner assignment, lists the byte addresses each lane touches, and prices every wave-instruction with the banking rules of
/opt/skills/guides/MI355X_MICROARCH.md §LDS (lane groups per instruction, bank = (a / 4) mod 64 for ds_read_b64 / b128, mod 32 for the
stores; an extra distinct address on a busy bank inside a lane group = one extra LDS cycle).
usage: ukf_lds_bank_model.py [n]      (n = padded state size, default 44 = 20 landmarks)"""
import sys
from collections import defaultdict

n = int(sys.argv[1]) if len(sys.argv) > 1 else 44
TPB, m, mq = 256, n // 2, n // 4
MMAX = 22

def rr_pair(k, t, nn):
    nm1 = nn - 1
    x = k - 1 + t
    if x >= nm1: x -= nm1
    a = 0 if k == 0 else 1 + x
    y = nm1 - k - 1 + t
    if y >= nm1: y -= nm1
    b = 1 + y
    return (a, b) if a < b else (b, a)

def tri(r, c):
    return 8 * (r * (r + 1) // 2 + c if r >= c else c * (c + 1) // 2 + r)

B128_GROUPS = [[0, 1, 2, 3, 12, 13, 14, 15, 20, 21, 22, 23, 24, 25, 26, 27], [4, 5, 6, 7, 8, 9, 10, 11, 16, 17, 18, 19, 28, 29, 30, 31]]
B128_GROUPS = B128_GROUPS + [[l + 32 for l in g] for g in B128_GROUPS]

def cycles(kind, addrs):
    """addrs: {lane: byte address} of the active lanes of ONE wave-instruction.  Returns (cycles without conflicts, extra conflict cycles)."""
    if kind == "rb64": groups, width, mod = [range(0, 32), range(32, 64)], 8, 64
    elif kind == "rb128": groups, width, mod = B128_GROUPS, 16, 64
    elif kind == "wb64": groups, width, mod = [range(16 * g, 16 * g + 16) for g in range(4)], 8, 32
    elif kind == "wb128": groups, width, mod = [range(8 * g, 8 * g + 8) for g in range(8)], 16, 32
    base = extra = 0
    for g in groups:
        banks = defaultdict(set)
        for l in g:
            if l in addrs:
                for d in range(0, width, 4):
                    banks[((addrs[l] + d) // 4) % mod].add((addrs[l] + d) // 4)
        if banks:
            base += 1
            extra += max(len(s) for s in banks.values()) - 1
    return base, extra

tot = defaultdict(lambda: [0, 0])
def price(name, kind, per_thread):
    """per_thread: {tid: address}; split into wavefronts"""
    for w in range(TPB // 64):
        a = {t - 64 * w: v for t, v in per_thread.items() if 64 * w <= t < 64 * w + 64}
        if a:
            b, e = cycles(kind, a)
            tot[name][0] += b; tot[name][1] += e

SVT_ROW = 8 * n   # bytes per row of V^T
owner0 = None
for T in range(m - 1):
    X, Y, quad_of = {}, {}, {}
    for q in range(mq):
        X[q], Y[q] = rr_pair(q, T, m); quad_of[X[q]] = q; quad_of[Y[q]] = q
    owner = {}
    Tn = T + 1 if T + 1 < m - 1 else 0
    ncrit = 0
    for q in range(mq):
        xn, yn = rr_pair(q, Tn, m)
        I, J = quad_of[xn], quad_of[yn]
        if I < J: I, J = J, I
        if I != J and (I, J) not in owner: owner[(I, J)] = 4 * ncrit; ncrit += 1
    nother = 0
    for I in range(1, mq):
        for J in range(I):
            if (I, J) not in owner: owner[(I, J)] = 64 + 4 * nother; nother += 1
    # ---- A: the 4 x 4 blocks between quadruples (four ds_read_b64 + four ds_write_b64 per lane and pass) ----
    rd = [dict() for _ in range(4)]; wr = [dict() for _ in range(4)]
    for (I, J), o in owner.items():
        row = [2 * X[I], 2 * X[I] + 1, 2 * Y[I], 2 * Y[I] + 1]; col = [2 * X[J], 2 * X[J] + 1, 2 * Y[J], 2 * Y[J] + 1]
        for i in range(2):
            for j in range(2):
                t = o + 2 * i + j
                first = [tri(row[i], col[j]), tri(row[i], col[j + 2]), tri(row[i + 2], col[j]), tri(row[i + 2], col[j + 2])]
                second = [tri(row[i], col[j]), tri(row[i], col[3 - j]), tri(row[3 - i], col[j]), tri(row[3 - i], col[3 - j])]
                for e in range(4): rd[e][t] = first[e]; wr[e][t] = second[e]
    for e in range(4):
        price("A blocks: ds_read_b64", "rb64", rd[e]); price("A blocks: ds_write_b64", "wb64", wr[e])
    # ---- rotation parameters cs[] (double2 = ds_read_b128): four per block lane, four per V item ----
    CS0 = 0   # addresses relative to s_csn: only their spread over banks matters
    for s in (1, 2):
        for which in (0, 1):
            a = {}
            for (I, J), o in owner.items():
                for i in range(2):
                    for j in range(2):
                        a[o + 2 * i + j] = CS0 + 16 * (s * MMAX + (2 * I + i if which == 0 else 2 * J + j))
            price("cs[] of the block lanes: ds_read_b128", "rb128", a)
    # ---- V items: threads 64 .., two items each; rows a, b, c, d of V^T, one 16-byte pair of columns ----
    for u in range(2):
        rdv = [dict() for _ in range(4)]
        csv = [dict() for _ in range(4)]
        for tid in range(64, TPB):
            it = tid - 64 + (TPB - 64) * u
            Q = it // m
            if Q >= mq: continue
            kp = it - Q * m
            ra = 16 * kp + 2 * SVT_ROW * X[Q]; rc = 16 * kp + 2 * SVT_ROW * Y[Q]
            for e, a in enumerate((ra, ra + SVT_ROW, rc, rc + SVT_ROW)): rdv[e][tid] = a
            for e, a in enumerate((MMAX + 2 * Q, MMAX + 2 * Q + 1, 2 * MMAX + 2 * Q, 2 * MMAX + 2 * Q + 1)): csv[e][tid] = 16 * a
        for e in range(4):
            price("V^T items: ds_read_b128", "rb128", rdv[e]); price("V^T items: ds_write_b128", "wb128", rdv[e])
            price("cs[] of the V items: ds_read_b128", "rb128", csv[e])
    # ---- parameter lanes (wavefront 0, lanes < 2 mq): per round 3 + 4 loads and 3 + 4 stores on the quadruple's diagonal block ----
    if T + 1 < m - 1:
        for rnd in (1, 2):
            ld = [dict() for _ in range(7)]
            for q in range(mq):
                x, y = rr_pair(q, T + 1, m)
                a, c = 2 * x, 2 * y
                b, d = a + 1, c + 1
                for u in range(2):
                    t = 2 * q + u
                    if rnd == 1: pp, qq, pq, xs = (b, b) if u else (a, a), (d, d) if u else (c, c), (d, b) if u else (c, a), [(b, a), (c, b), (d, a), (d, c)]
                    else: pp, qq, pq, xs = (b, b) if u else (a, a), (c, c) if u else (d, d), (c, b) if u else (d, a), [(b, a), (d, b), (c, a), (d, c)]
                    for e, (r, cc) in enumerate((pp, qq, pq)): ld[e][t] = tri(r, cc)
                    if u == 1:
                        for e, (r, cc) in enumerate(xs): ld[3 + e][t] = tri(r, cc)
            for e in range(7):
                price("parameter lanes: ds_read_b64", "rb64", ld[e]); price("parameter lanes: ds_write_b64", "wb64", ld[e])

print(f"ukf_sqrt_kernel<44, 256>, pass-table path, padded state size n = {n}: LDS cycles of one sweep's {m - 1} passes (model)")
print(f"{'access':46s} {'conflict-free':>14s} {'extra (conflicts)':>18s} {'share of the extra':>19s}")
B = sum(v[0] for v in tot.values()); E = sum(v[1] for v in tot.values())
for k, (b, e) in sorted(tot.items(), key=lambda kv: -kv[1][1]):
    print(f"{k:46s} {b:14d} {e:18d} {100.0 * e / max(E, 1):18.1f}%")
print(f"{'total':46s} {B:14d} {E:18d}   conflict share of the LDS cycles = {E / (B + E):.3f}")
