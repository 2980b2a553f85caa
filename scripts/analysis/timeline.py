"""Timeline of a rocprofv3 --kernel-trace CSV: per step (delimited by a marker kernel, by default the
optimiser's adam_step_kernel) the wall time, the time at least one kernel is running, the time two or more
overlap, the idle gaps, and the busy time per queue.  usage: timeline.py <kernel_trace.csv> [marker]"""
import csv
import sys
from collections import defaultdict


def load(path):
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r["Queue_Id"]))
    rows.sort()
    return rows


def union(intervals):
    tot, cur_s, cur_e = 0, None, None
    for s, e in sorted(intervals):
        if cur_e is None or s > cur_e:
            if cur_e is not None:
                tot += cur_e - cur_s
            cur_s, cur_e = s, e
        else:
            cur_e = max(cur_e, e)
    if cur_e is not None:
        tot += cur_e - cur_s
    return tot


def overlap2(intervals):
    """time during which at least two intervals are open"""
    ev = []
    for s, e in intervals:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, last, tot = 0, None, 0
    for t, d in ev:
        if depth >= 2:
            tot += t - last
        depth += d
        last = t
    return tot


def main():
    rows = load(sys.argv[1])
    marker = sys.argv[2] if len(sys.argv) > 2 else "adam_step_kernel"
    ends = [e for s, e, k, q in rows if marker in k]
    print("%d dispatches, %d step markers" % (len(rows), len(ends)))
    for i in range(1, len(ends)):
        t0, t1 = ends[i - 1], ends[i]
        step = [(s, e, k, q) for s, e, k, q in rows if s >= t0 and e <= t1]
        iv = [(s, e) for s, e, _, _ in step]
        busy = union(iv)
        per_q = defaultdict(list)
        for s, e, k, q in step:
            per_q[q].append((s, e))
        gaps = sorted(((b[0] - a, a) for a, b in zip([t0] + sorted(e for _, e in iv), sorted(iv) + [(t1, t1)])), reverse=True)
        print("step %d: wall %.2f ms, >=1 kernel running %.2f ms, >=2 running %.2f ms, sum of kernels %.2f ms, %d launches" % (
            i, (t1 - t0) / 1e6, busy / 1e6, overlap2(iv) / 1e6, sum(e - s for s, e in iv) / 1e6, len(step)))
        for q, v in sorted(per_q.items()):
            print("   queue %s: %4d launches, busy %.2f ms" % (q, len(v), union(v) / 1e6))
        # idle time = wall - busy; list kernels that follow the largest idle stretches
        idle = []
        cur_end = t0
        for s, e, k, q in sorted(step):
            if s > cur_end:
                idle.append((s - cur_end, k))
            cur_end = max(cur_end, e)
        idle.sort(reverse=True)
        print("   idle %.2f ms in %d gaps; largest: %s" % (sum(d for d, _ in idle) / 1e6, len(idle),
              ", ".join("%.0f us before %s" % (d / 1e3, k.split("(")[0][-40:]) for d, k in idle[:6])))


if __name__ == "__main__":
    main()
