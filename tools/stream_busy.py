#!/usr/bin/env python3
"""Per-queue GPU busy time of the bench from a rocprofv3 kernel trace (kt_kernel_trace.csv):
which stream is the critical path, how much of it is idle between kernels, and what runs on it.

  python tools/stream_busy.py /tmp/kt/kt_kernel_trace.csv [--tail-frac 0.5]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    tail = float(sys.argv[sys.argv.index("--tail-frac") + 1]) if "--tail-frac" in sys.argv else 0.5
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    # one whole TTA step: from the start of the 4-channel stem conv of one teacher forward to the start of the next one
    # (k_spconv_scalar runs twice per step: teacher, student)
    stem = [r[0] for r in rows if "k_spconv_stem" in r[3] or "k_spconv_scalar" in r[3]]
    nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 2
    if len(stem) >= 2 * (nsteps + 1):
        lo, hi = stem[-2 * (nsteps + 1)], stem[-2]
    else:
        t0, t1 = rows[0][0], max(r[1] for r in rows)
        lo, hi = t1 - int((t1 - t0) * tail), t1
    rows = [r for r in rows if lo <= r[0] < hi]
    print(f"{nsteps} step(s)")
    span = (max(r[1] for r in rows) - rows[0][0]) / 1e6
    print(f"window {span:.2f} ms, {len(rows)} dispatches")
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    # union of busy intervals over all queues
    ev = sorted((r[0], r[1]) for r in rows)
    busy, cur_s, cur_e = 0, ev[0][0], ev[0][1]
    for s, e in ev[1:]:
        if s > cur_e:
            busy += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    busy += cur_e - cur_s
    print(f"GPU busy (any queue) {busy / 1e6:.2f} ms = {100 * busy / 1e6 / span:.1f} % of the window")
    for q, rs in sorted(byq.items(), key=lambda kv: -sum(r[1] - r[0] for r in kv[1])):
        dur = sum(r[1] - r[0] for r in rs) / 1e6
        gaps = [rs[i + 1][0] - rs[i][1] for i in range(len(rs) - 1)]
        small = [g for g in gaps if 0 < g < 50_000]
        print(f"queue {q}: {len(rs)} dispatches, kernel time {dur:.2f} ms ({100 * dur / span:.1f} %), "
              f"gaps<50us: {len(small)} totalling {sum(small) / 1e6:.2f} ms (median {sorted(small)[len(small) // 2] / 1e3 if small else 0:.1f} us), "
              f"gaps>=50us: {sum(g for g in gaps if g >= 50_000) / 1e6:.2f} ms")
        agg = collections.defaultdict(lambda: [0, 0])
        for r in rs:
            k = r[3].replace("(anonymous namespace)::", "").replace("at::native::", "")
            k = k.split("(")[0][:110]
            agg[k][0] += 1
            agg[k][1] += r[1] - r[0]
        for k, (n, d) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:60]:
            print(f"    {d / 1e6:8.3f} ms {n:6d}  {k}")


if __name__ == "__main__":
    main()
