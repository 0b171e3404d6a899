"""GPU idle time inside one proof, from a rocprofv3 kernel trace: python tools/timeline.py <kernel_trace.csv> [last-kernel-substring]

Splits the trace into proofs at every occurrence of the last kernel of a proof (default: the opening gather, k_gather_batch),
takes the LAST complete proof, and reports its span, the time at least one kernel was running (union over all streams), the idle
remainder, and the largest gaps with the kernels on either side -- where the host's round trips (channel draws, small copies) show."""
import csv
import re
import sys


def main():
    path = sys.argv[1]
    last = sys.argv[2] if len(sys.argv) > 2 else "k_gather_batch"
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            m = re.search(r"\bk_\w+(<[^>]*>)?", r["Kernel_Name"])
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), m.group(0) if m else r["Kernel_Name"][:60]))
    rows.sort()
    ends = [i for i, r in enumerate(rows) if last in r[2]]
    if len(ends) < 3:
        raise SystemExit("fewer than three proofs in the trace")
    lo, hi = ends[-2] + 1, ends[-1] + 1      # the last complete proof: everything after the previous proof's opening gather
    proof = rows[lo:hi]
    t0, t1 = proof[0][0], max(r[1] for r in proof)
    busy, cur_end, gaps = 0, t0, []
    prev = None
    for s, e, name in proof:
        if s > cur_end:
            gaps.append((s - cur_end, prev, name, cur_end - t0))
            busy_start = s
        if e > cur_end:
            busy += e - max(s, cur_end)
            cur_end, prev = e, name
    span = t1 - t0
    print("proof span %.3f ms, kernels running %.3f ms, idle %.3f ms in %d gaps (%d kernels)" % (span / 1e6, busy / 1e6, (span - busy) / 1e6, len(gaps), len(proof)))
    print("gaps by size class: >=20us %.3f ms | 5-20us %.3f ms | <5us %.3f ms" % (
        sum(g[0] for g in gaps if g[0] >= 20000) / 1e6, sum(g[0] for g in gaps if 5000 <= g[0] < 20000) / 1e6, sum(g[0] for g in gaps if g[0] < 5000) / 1e6))
    for g in sorted(gaps, reverse=True)[:25]:
        print("  %7.1f us at %8.3f ms  after %-40s before %s" % (g[0] / 1e3, g[3] / 1e6, (g[1] or "")[:40], g[2][:60]))


if __name__ == "__main__":
    main()
