"""Developer probe: what ran when.  Reads a rocprofv3 --kernel-trace CSV (kernel_trace.csv) and prints, for the last
`window` ms of the run: how many kernels were running at once (time share), time per kernel name (sum of durations, mean,
count), the idle time, and a coarse text timeline per queue.
Usage: python tools/timeline.py <kernel_trace.csv> [window_ms=200] [columns=160] [end_before_last_ms=0 | -1 = around the median kernel start]"""
import collections
import csv
import sys


def short(name):
    name = name.replace("void ", "").replace("bz2gpu::", "")
    return name.split("(")[0]


def main():
    path = sys.argv[1]
    window_ms = float(sys.argv[2]) if len(sys.argv) > 2 else 200.0
    columns = int(sys.argv[3]) if len(sys.argv) > 3 else 160
    back_ms = float(sys.argv[4]) if len(sys.argv) > 4 else 0.0
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), short(r["Kernel_Name"]), r.get("Queue_Id", "0"),
                     int(r.get("Grid_Size", 0) or 0)))
    rows.sort()
    if back_ms < 0:      # the window around the median kernel start: the middle of the run
        starts = sorted(r[0] for r in rows)
        t_end = starts[len(starts) // 2] + int(window_ms * 1e6 / 2)
    else:
        t_end = max(r[1] for r in rows) - int(back_ms * 1e6)
    t0 = t_end - int(window_ms * 1e6)
    rows = [(s, min(e, t_end), n, q, g) for s, e, n, q, g in rows if e > t0 and s < t_end]
    # concurrency histogram
    events = []
    for s, e, *_ in rows:
        events.append((max(s, t0), 1))
        events.append((e, -1))
    events.sort()
    share = collections.Counter()
    level, last = 0, t0
    for t, d in events:
        share[level] += t - last
        last = t
        level += d
    total = t_end - t0
    print(f"window {total / 1e6:.1f} ms, {len(rows)} kernels")
    print("kernels running at once (share of the window): " +
          ", ".join(f"{k}: {100 * v / total:.1f} %" for k, v in sorted(share.items())))
    by = collections.defaultdict(list)
    for s, e, n, q, g in rows:
        by[n].append((e - max(s, t0)) / 1e6)
    print(f"{'kernel':28s} {'count':>6s} {'sum ms':>9s} {'mean ms':>9s} {'max ms':>9s}")
    for n, d in sorted(by.items(), key=lambda kv: -sum(kv[1])):
        print(f"{n:28s} {len(d):6d} {sum(d):9.2f} {sum(d) / len(d):9.3f} {max(d):9.3f}")
    # timeline per queue
    queues = sorted({r[3] for r in rows})
    letters = {}
    for n in sorted(by, key=lambda n: -sum(by[n])):
        base = n.replace("k_", "")
        for ch in base + base.upper() + "0123456789":
            if ch.isalnum() and ch not in letters.values():
                letters[n] = ch
                break
    print("legend: " + ", ".join(f"{v}={k}" for k, v in letters.items()))
    cell = total / columns
    for q in queues:
        line = [" "] * columns
        for s, e, n, qq, g in rows:
            if qq != q:
                continue
            a = int((max(s, t0) - t0) / cell)
            b = max(a + 1, int((e - t0) / cell))
            for i in range(a, min(b, columns)):
                line[i] = letters[n]
        print(f"q{q:>3s} |{''.join(line)}|")


if __name__ == "__main__":
    main()
