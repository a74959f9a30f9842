"""GPU occupancy statistics of a rocprofv3 kernel trace (all streams): how much of the wall-clock window has >= 1, >= 2,
>= 3 kernels in flight, the idle gaps, and the kernels that most often run alone.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/trace -- python3 bench.py --steps 10 --no-cpu-baseline
    python tools/timeline_stats.py gpurun_out/trace [from_fraction=0.5] [to_fraction=1.0]
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    skip = float(sys.argv[2]) if len(sys.argv) > 2 else 0.5
    upto = float(sys.argv[3]) if len(sys.argv) > 3 else 1.0
    rows = []
    for f in glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True):
        with open(f, newline='') as fh:
            for r in csv.DictReader(fh):
                rows.append((int(r['Start_Timestamp']), int(r['End_Timestamp']), r.get('Kernel_Name') or r.get('Name')))
    rows.sort()
    t0, t1 = rows[0][0], max(r[1] for r in rows)
    lo, hi = t0 + (t1 - t0) * skip, t0 + (t1 - t0) * upto      # a steady-state slice of the run
    rows = [r for r in rows if r[0] >= lo and r[1] <= hi]
    ev = []
    for s, e, n in rows:
        ev.append((s, 1, n)); ev.append((e, -1, n))
    ev.sort()
    level_time = defaultdict(int)
    alone = defaultdict(int)
    gaps = []
    gap_ctx = []                       # (length, kernel that ended before the gap, kernel that starts after it)
    last_end_name = ''
    cur, prev_t, running = 0, ev[0][0], {}
    for t, d, n in ev:
        dt = t - prev_t
        level_time[cur] += dt
        if cur == 1:
            alone[next(iter(running))] += dt
        if cur == 0 and dt > 0:
            gaps.append(dt)
            gap_ctx.append((dt, last_end_name, n))
        if d == -1:
            last_end_name = n
        prev_t = t
        if d == 1:
            running[n] = running.get(n, 0) + 1
        else:
            running[n] -= 1
            if running[n] == 0:
                del running[n]
        cur += d
    wall = ev[-1][0] - ev[0][0]
    busy = sum(e - s for s, e, _ in rows)
    print(f'window {wall / 1e6:.3f} ms, {len(rows)} kernels, summed kernel time {busy / 1e6:.3f} ms ({busy / wall:.2f} average in flight)')
    for k in sorted(level_time):
        print(f'  {k} kernels in flight: {level_time[k] / 1e6:8.3f} ms  {100.0 * level_time[k] / wall:5.1f} %')
    gaps.sort(reverse=True)
    print(f'  idle gaps: {len(gaps)}, total {sum(gaps) / 1e6:.3f} ms, largest {[round(g / 1e3, 1) for g in gaps[:8]]} us')
    # idle time without the profiler's own buffer flushes (gaps of milliseconds), and where the idle time sits
    real = [g for g in gap_ctx if g[0] < 2_000_000]
    print(f'  idle gaps below 2 ms: {len(real)}, total {sum(g[0] for g in real) / 1e6:.3f} ms of {(wall - sum(g[0] for g in gap_ctx if g[0] >= 2_000_000)) / 1e6:.3f} ms')
    by_pair = defaultdict(lambda: [0, 0])
    for dt, a, b in real:
        k = (a.split('(')[0][-48:], b.split('(')[0][-48:])
        by_pair[k][0] += dt; by_pair[k][1] += 1
    print('  idle time by (kernel before, kernel after), top 14:')
    for k, (t, c) in sorted(by_pair.items(), key=lambda kv: -kv[1][0])[:14]:
        print(f'    {t / 1e3:9.1f} us in {c:4d} gaps   {k[0]}  ->  {k[1]}')
    print('  kernels running alone (top 12 by time):')
    for n, t in sorted(alone.items(), key=lambda kv: -kv[1])[:12]:
        print(f'    {t / 1e6:8.3f} ms  {n[:110]}')


if __name__ == '__main__':
    main()
