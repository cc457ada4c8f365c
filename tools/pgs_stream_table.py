#!/usr/bin/env python3
"""Pose graph (BASELINE configs[4]): throughput and slot occupancy of the streaming solve (pgs_set_slots) against the lockstep solve.

  python tools/pgs_stream_table.py [--graphs 1024,2048] [--slots 0,256,512] [--groups 0,2,4] [--poses 1000 --landmarks 200] [--depth 3]

Per (graphs, slots, groups): solves/s over `--solves` timed solves (HIP events on the handle's stream), trials launched, slot-trials run,
mean occupancy = slot-trials / (trials x slots) per group, and a check that iteration / trial counts equal the lockstep solve's."""
import argparse
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--graphs", default="1024")
    ap.add_argument("--slots", default="0,256,512")
    ap.add_argument("--groups", default="0")
    ap.add_argument("--poses", type=int, default=1000)
    ap.add_argument("--landmarks", type=int, default=200)
    ap.add_argument("--k-per-pose", type=int, default=32)
    ap.add_argument("--solves", type=int, default=2)
    ap.add_argument("--timeline", action="store_true", help="print the running slots of every trial of the last solve")
    args = ap.parse_args()
    import torch
    import live_ekf_slam_amd as S
    from live_ekf_slam_amd.scenario import make_scenario
    L, N = args.landmarks, args.poses
    lm, cmds = make_scenario(1234, L, N - 1)
    dev = torch.device("cuda", 0)
    print(f"# {N} poses x {L} landmarks; solves/s = graphs x {args.solves} / HIP-event time; occupancy = slot-trials / (trials x slots of the group)")
    print("# graphs slots groups | solves/s  ms/solve-batch | trials  slot-trials  occupancy | counts == lockstep")
    for B in [int(x) for x in args.graphs.split(",")]:
        pg = S.BatchedPoseGraph(B, num_iterations=N, L_max=L, k_per_pose=args.k_per_pose, device=0).readParams()
        stream = torch.cuda.Stream(device=dev)
        pg.set_stream(stream.cuda_stream)
        pg.set_map(lm); pg.set_seed(2025); pg.init(0.0, 0.0, 0.0)
        ref = None
        with torch.cuda.stream(stream):
            pg.run_sim(cmds)
            for G in [int(x) for x in args.groups.split(",")]:
                for slots in [int(x) for x in args.slots.split(",")]:
                    if slots >= B:
                        continue
                    pg.set_groups(G); pg.set_slots(slots)
                    pg.solvePoseGraph(); torch.cuda.synchronize(dev)
                    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                    e0.record(stream)
                    for _ in range(args.solves):
                        pg.solvePoseGraph()
                    e1.record(stream)
                    torch.cuda.synchronize(dev)
                    ms = e0.elapsed_time(e1) / args.solves
                    st = pg.stats()
                    tl = pg.last_solve_timeline()
                    if ref is None:
                        ref = (st["iterations"].copy(), st["trials"].copy(), [pg.get_graph(b, 1)["poses"].copy() for b in (0, B - 1)])
                    same = "counts %s, poses of 2 instances %s" % (
                        "equal" if (np.array_equal(ref[0], st["iterations"]) and np.array_equal(ref[1], st["trials"])) else
                        "DIFFER in %d instances" % int(((ref[0] != st["iterations"]) | (ref[1] != st["trials"])).sum()),
                        "bit-identical" if all(np.array_equal(ref[2][i], pg.get_graph(b, 1)["poses"]) for i, b in enumerate((0, B - 1))) else
                        "differ by %.1e m" % max(float(np.abs(ref[2][i] - pg.get_graph(b, 1)["poses"]).max()) for i, b in enumerate((0, B - 1))))
                    ng = len(tl)
                    cap = -(-slots // ng) if slots else -(-B // ng)
                    ntr = max(len(a) for a in tl)
                    slot_trials = int(sum(int(a.sum()) for a in tl))
                    occ = float(np.mean([a.sum() / (len(a) * cap) for a in tl]))
                    print(f"{B:6d} {slots:5d} {ng:4d}   | {B / ms * 1e3:8.0f}  {ms:8.2f}       | {ntr:5d}  {slot_trials:9d}  {occ:8.3f}    | {same}", flush=True)
                    if args.timeline:
                        for g, a in enumerate(tl):
                            print(f"#   group {g}: " + " ".join(str(int(v)) for v in a))
        print(f"#   LM iterations mean {st['iterations'].mean():.2f}, trials mean {st['trials'].mean():.2f}, max {st['trials'].max()}, flagged {(st['flags'] != 0).sum()}")
        pg.close()


if __name__ == "__main__":
    main()
