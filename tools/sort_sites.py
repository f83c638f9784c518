#!/usr/bin/env python3
"""Which call site the library kernels without a name of ours (rocPRIM sorts / scans / selects, runtime fills and copies)
belong to: each one is attributed to the last kernel of ours before it on the same queue.

  python tools/sort_sites.py /tmp/kt/kt_kernel_trace.csv [--steps 3]
"""
import collections
import csv
import sys


def main():
    path = sys.argv[1]
    nsteps = int(sys.argv[sys.argv.index("--steps") + 1]) if "--steps" in sys.argv else 3
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    stem = [r[0] for r in rows if "k_spconv_stem" in r[3]]
    lo, hi = stem[-2 * (nsteps + 1)], stem[-2]
    rows = [r for r in rows if lo <= r[0] < hi]
    byq = collections.defaultdict(list)
    for r in rows:
        byq[r[2]].append(r)
    for q, rs in byq.items():
        agg = collections.defaultdict(lambda: collections.defaultdict(lambda: [0, 0]))
        site = "?"
        for s, e, _, name in rs:
            short = name.split("(")[0].replace("void ", "")
            anon = "rocprim" in name or "__amd_rocclr" in name
            if not anon and (short.startswith("k_") or short.startswith("k_")):
                site = short[:40]
                continue
            if anon:
                kind = "fill" if "fillBuffer" in name else "copy" if "copyBuffer" in name else "sort" if "radix" in name else "rocprim-other"
                agg[site][kind][0] += 1
                agg[site][kind][1] += e - s
        tot = sum(v[1] for d in agg.values() for v in d.values())
        print(f"queue {q}: library kernels {tot / 1e6 / nsteps:.2f} ms per step")
        for site, d in sorted(agg.items(), key=lambda kv: -sum(v[1] for v in kv[1].values())):
            parts = ", ".join(f"{k} {v[0] / nsteps:.1f}x {v[1] / 1e3 / nsteps:.0f} us" for k, v in sorted(d.items(), key=lambda kv: -kv[1][1]))
            print(f"   after {site:42s} {sum(v[1] for v in d.values()) / 1e3 / nsteps:8.0f} us/step   {parts}")


if __name__ == "__main__":
    main()
