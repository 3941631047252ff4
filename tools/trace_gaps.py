"""GPU busy / idle analysis of a rocprofv3 --kernel-trace CSV (developer tool).

    rocprofv3 --kernel-trace --output-format csv -d /tmp/tr -o t -- python bench.py --no-cpu-baseline
    python tools/trace_gaps.py /tmp/tr/<host>/t_kernel_trace.csv [n_last_ms]

Prints, for the last `n_last_ms` of the trace (default: the last 300 ms, i.e. the timed steps): wall time, the union of
kernel execution intervals (GPU busy), the sum of kernel durations (=> average concurrency), the idle gaps longer than
20 us with the kernels either side of the largest ones, and the busy time split by number of co-running kernels."""
import csv
import sys


def main():
    path = sys.argv[1]
    last_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 300.0
    rows = []
    with open(path) as f:
        for r in csv.DictReader(f):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"][:70]))
    rows.sort()
    t_end = max(e for _, e, _ in rows)
    t0 = t_end - int(last_ms * 1e6)
    rows = [r for r in rows if r[0] >= t0]
    wall = (t_end - rows[0][0]) / 1e6
    ssum = sum(e - s for s, e, _ in rows) / 1e6
    # sweep
    ev = []
    for s, e, _ in rows:
        ev.append((s, 1))
        ev.append((e, -1))
    ev.sort()
    depth, prev, by_depth = 0, ev[0][0], {}
    for t, d in ev:
        by_depth[depth] = by_depth.get(depth, 0) + (t - prev)
        prev = t
        depth += d
    busy = sum(v for k, v in by_depth.items() if k > 0) / 1e6
    print("window %.1f ms: %d kernels, GPU busy (union) %.2f ms = %.1f %%, sum of kernel durations %.2f ms (avg concurrency %.2f while busy)"
          % (wall, len(rows), busy, 100 * busy / wall, ssum, ssum / busy))
    print("time by number of co-running kernels (ms):", {k: round(v / 1e6, 2) for k, v in sorted(by_depth.items())})
    # which kernels run ALONE (depth 1), and for how long: an under-filled kernel running alone is where a step's wall time hides
    ev2 = []
    for i, (s_, e_, _) in enumerate(rows):
        ev2.append((s_, 1, i))
        ev2.append((e_, -1, i))
    ev2.sort()
    live, prev_t, solo = set(), ev2[0][0], {}
    for t, d, i in ev2:
        if len(live) == 1:
            k = rows[next(iter(live))][2].split("(")[0][:64]
            solo[k] = solo.get(k, 0) + (t - prev_t)
        prev_t = t
        if d > 0:
            live.add(i)
        else:
            live.discard(i)
    print("kernels running alone (ms over the window):")
    for k, v in sorted(solo.items(), key=lambda kv: -kv[1])[:14]:
        print("  %8.2f  %s" % (v / 1e6, k))
    # gaps
    gaps = []
    cur_end, cur_name = rows[0][1], rows[0][2]
    for s, e, n in rows[1:]:
        if s > cur_end:
            gaps.append((s - cur_end, cur_end, cur_name, n))
        if e > cur_end:
            cur_end, cur_name = e, n
    big = [g for g in gaps if g[0] > 20000]
    print("idle gaps: %d total %.2f ms; %d gaps > 20 us total %.2f ms" % (len(gaps), sum(g[0] for g in gaps) / 1e6, len(big), sum(g[0] for g in big) / 1e6))
    for g in sorted(big, reverse=True)[:12]:
        print("  %.3f ms at t=%.2f ms  after %s  before %s" % (g[0] / 1e6, (g[1] - t0) / 1e6, g[2][:48], g[3][:48]))


if __name__ == "__main__":
    main()
