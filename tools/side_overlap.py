#!/usr/bin/env python3
"""Which side-stream kernels slow the main stream down, from ONE rocprofv3 kernel trace of the bench.

The main queue issues the same kernel sequence every step.  For position i of the sequence the shortest duration over
the traced steps is the kernel's undisturbed time; what an instance takes beyond that is its excess, and the excess is
shared among the side-queue kernels that ran during the instance, in proportion to their overlap with it.  Summed by
side-kernel name this says where the "the side stream costs the main stream 2 ms" comes from.

  python tools/side_overlap.py /tmp/kt/kt_kernel_trace.csv [--steps 6]
"""
import bisect
import collections
import csv
import sys


def short(n):
    n = n.split("(")[0]
    if "rocprim" in n:
        for k in ("onesweep", "histogram", "radix_sort", "scan", "partition", "reduce", "transform", "merge"):
            if k in n:
                return "rocprim " + k
        return "rocprim other"
    return n.replace("void ", "")[:48]


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 6
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    busy = collections.Counter()
    for s, e, q, n in rows:
        busy[q] += e - s
    mainq = busy.most_common(1)[0][0]
    main_rows = [r for r in rows if r[2] == mainq]
    side_rows = [r for r in rows if r[2] != mainq]
    stems = [i for i, r in enumerate(main_rows) if "k_spconv_stem" in r[3]]
    assert len(stems) >= 2 * (nsteps + 1), f"only {len(stems)} stem launches on the main queue"
    bounds = [stems[-2 * (nsteps + 1) + 2 * k] for k in range(nsteps + 1)]
    steps = [main_rows[bounds[k]:bounds[k + 1]] for k in range(nsteps)]
    L = min(len(s) for s in steps)
    same = all(len(s) == L for s in steps) and all(short(steps[0][i][3]) == short(s[i][3]) for s in steps for i in range(L))
    print(f"main queue {mainq}: {nsteps} steps of {L} kernels; sequences identical: {same}")
    if not same:
        # keep the positions whose names agree in every step (the optimizer / loss tail may differ by a launch)
        pass
    side_starts = [r[0] for r in side_rows]
    max_side = max((r[1] - r[0] for r in side_rows), default=0)
    by_side = collections.defaultdict(lambda: [0.0, 0.0, 0])     # name -> [overlap ns, attributed excess ns, instances]
    by_main = collections.defaultdict(lambda: [0.0, 0.0, 0.0])   # name -> [duration, baseline, excess with no side kernel]
    tot_excess = tot_base = tot_dur = unattributed = 0.0
    for i in range(L):
        if any(short(s[i][3]) != short(steps[0][i][3]) for s in steps):
            continue
        base = min(s[i][1] - s[i][0] for s in steps)
        for s in steps:
            a, b, _, n = s[i]
            dur = b - a
            ex = dur - base
            tot_dur += dur
            tot_base += base
            tot_excess += ex
            m = by_main[short(n)]
            m[0] += dur
            m[1] += base
            lo = bisect.bisect_left(side_starts, a - max_side)
            hi = bisect.bisect_right(side_starts, b)
            ov = []
            for r in side_rows[lo:hi]:
                o = min(b, r[1]) - max(a, r[0])
                if o > 0:
                    ov.append((o, short(r[3])))
            tot_ov = sum(o for o, _ in ov)
            if tot_ov == 0:
                unattributed += ex
                m[2] += ex
                continue
            for o, sn in ov:
                e = by_side[sn]
                e[0] += o
                e[1] += ex * o / tot_ov
                e[2] += 1
    k = 1e6 * nsteps
    print(f"per step: main-queue kernel time {tot_dur / k:.2f} ms, undisturbed (min over steps per position) {tot_base / k:.2f} ms, "
          f"excess {tot_excess / k:.2f} ms, of which with no side kernel running {unattributed / k:.2f} ms")
    print(f"\n{'side kernel':50s} {'overlap ms/step':>16s} {'excess ms/step':>15s} {'excess/overlap':>15s} {'launches/step':>14s}")
    for sn, (o, e, c) in sorted(by_side.items(), key=lambda kv: -kv[1][1])[:30]:
        print(f"{sn:50s} {o / k:16.3f} {e / k:15.3f} {e / max(o, 1):15.2f} {c / nsteps:14.1f}")
    print(f"\n{'main kernel':50s} {'ms/step':>10s} {'undisturbed':>12s} {'excess':>8s}")
    for mn, (d, b, u) in sorted(by_main.items(), key=lambda kv: -(kv[1][0] - kv[1][1]))[:20]:
        print(f"{mn:50s} {d / k:10.3f} {b / k:12.3f} {(d - b) / k:8.3f}")


if __name__ == "__main__":
    main()
