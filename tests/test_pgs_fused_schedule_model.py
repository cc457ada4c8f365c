"""Host-side model of the buffer schedule of pgs_chain_syrk_kernel (live_ekf_slam_amd/csrc/pgs_kernel.hip): the producer, the
staging wavefront, the column phase and the tile phase of one workgroup meet at ONE barrier per iteration and pass data
through double-buffered LDS arrays indexed by iteration parity.  The model replays the kernel's index expressions and
checks, for every buffer, that (a) nothing is read in the barrier interval in which it is written, (b) every read sees the
chunk it is meant to see, (c) nothing is overwritten before its last reader has passed a barrier.  It also checks the tile
deal: every tile of the lower triangle is held by exactly one wavefront of the NB workgroups and no wavefront holds more
than NS.  No GPU needed."""
import itertools

import pytest


def accesses(it, nch):
    """(role, buffer, parity, 'r' | 'w', chunk) of iteration `it` (0 .. nch + 1), as written in the kernel."""
    a = []
    if it < nch:                       # producer: recursion of chunk `it` from s_in[it & 1] into s_ring[it & 1], inputs of chunk it + 1
        a.append(("producer", "s_in", it & 1, "r", it))
        a.append(("producer", "s_ring", it & 1, "w", it))
        if it + 1 < nch:
            a.append(("producer", "s_in", (it + 1) & 1, "w", it + 1))
        a.append(("stager", "s_E", it & 1, "w", it))          # wavefront 4: E blocks + index of chunk `it`
    if 1 <= it <= nch:                 # columns of chunk it - 1
        c = it - 1
        a.append(("columns", "s_ring", c & 1, "r", c))
        a.append(("columns", "s_E", c & 1, "r", c))
        a.append(("columns", "s_yb", c & 1, "w", c))
    if it >= 2:                        # tiles + z row over chunk it - 2, complete in s_yb[it & 1]
        a.append(("tiles", "s_yb", it & 1, "r", it - 2))
    return a


@pytest.mark.parametrize("nch", [1, 2, 3, 7, 250])
def test_double_buffers_of_the_fused_kernel_never_race(nch):
    content = {}                                   # (buffer, parity) -> chunk it holds (as of the last barrier)
    consumed = {("s_yb", c): False for c in range(nch)}
    ring_read = {c: False for c in range(nch)}
    for it in range(nch + 2):
        acc = accesses(it, nch)
        written = {(b, par): ch for role, b, par, rw, ch in acc if rw == "w"}
        for role, b, par, rw, ch in acc:
            if rw != "r":
                continue
            if b == "s_in" and role == "producer":
                # the only same-wavefront pair: the producer reads s_in[it & 1] and writes s_in[(it + 1) & 1] - other parity
                assert (b, par) not in written
                expected = content.get((b, par), 0 if ch == 0 else None)   # chunk 0 is stored before the loop
                assert expected == ch
                continue
            assert (b, par) not in written, (it, role, b, par)             # (a) no read of a buffer written in this interval
            assert content.get((b, par)) == ch, (it, role, b, par, content.get((b, par)), ch)   # (b) the right chunk
            if b == "s_yb":
                consumed[("s_yb", ch)] = True
            if b == "s_ring":
                ring_read[ch] = True
        for (b, par), ch in written.items():                               # (c) the value being replaced has been read
            old = content.get((b, par))
            if old is not None and b == "s_yb":
                assert consumed[("s_yb", old)], (it, old)
            if old is not None and b == "s_ring":
                assert ring_read[old], (it, old)
        content.update(written)                                            # the barrier publishes the interval's writes
    assert all(consumed.values()) and all(ring_read.values())


@pytest.mark.parametrize("NS,NB", [(6, 2), (4, 3), (3, 4)])
def test_tile_deal_covers_the_lower_triangle_once(NS, NB):
    NW = 6 * NB
    for M in (1, 8, 16, 17, 100, 170, 176):
        m2 = 2 * M
        nt = (m2 + 31) >> 5
        ntile = nt * (nt + 1) // 2
        assert ntile <= 72 and NW * NS >= 72          # the host's admission rule (pgs_solve: nt * (nt + 1) / 2 <= 72)
        held = {}
        for hb, w in itertools.product(range(NB), (1, 2, 3, 5, 6, 7)):
            mw = (w - 1 if w < 4 else w - 2) * NB + hb
            for s in range(NS):
                t = mw + NW * s
                if t < ntile:
                    assert t not in held
                    held[t] = (hb, w, s)
        assert sorted(held) == list(range(ntile))
        # tile index -> (ti, tj) as decoded in the kernel covers rows < 2M exactly
        rows = set()
        for t in held:
            ti = int(((8 * t + 1) ** 0.5 - 1) / 2)
            while ti * (ti + 1) // 2 > t:
                ti -= 1
            while (ti + 1) * (ti + 2) // 2 <= t:
                ti += 1
            tj = t - ti * (ti + 1) // 2
            assert 0 <= tj <= ti < nt
            rows.add(ti)
        assert rows == set(range(nt))
