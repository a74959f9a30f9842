"""Per-(kernel, grid) duration summary of a rocprofv3 kernel trace.

    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/prof -- python3 bench.py ...
    python tools/prof_by_grid.py gpurun_out/prof > profiles/rNN_<tag>_by_grid.csv

The --stats table aggregates by kernel NAME, which mixes the different layers a template instance serves (three grids of
wino_kernel<2,8,2> at 80 us average hide the 102 us level-0 launch the roofline is quoted on).  This groups the raw
*_kernel_trace.csv rows by (kernel, grid, workgroup) instead: calls, avg / min / max / total microseconds, sorted by total.
"""
import csv
import glob
import os
import sys
from collections import defaultdict


def main():
    root = sys.argv[1]
    skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0      # drop the first N dispatches of every group (warm-up)
    files = glob.glob(os.path.join(root, '**', '*kernel_trace.csv'), recursive=True)
    if not files:
        sys.exit(f'no *kernel_trace.csv under {root}')
    groups = defaultdict(list)
    for f in files:
        with open(f, newline='') as fh:
            for row in csv.DictReader(fh):
                name = row.get('Kernel_Name') or row.get('Name')
                grid = tuple(int(row.get(f'Grid_Size_{a}', row.get(f'Grid_Size{a}', 0)) or 0) for a in 'XYZ')
                wg = tuple(int(row.get(f'Workgroup_Size_{a}', row.get(f'Workgroup_Size{a}', 0)) or 0) for a in 'XYZ')
                dur = (int(row['End_Timestamp']) - int(row['Start_Timestamp'])) / 1000.0
                groups[(name, grid, wg)].append(dur)
    rows = []
    for (name, grid, wg), durs in groups.items():
        d = durs[skip:] if len(durs) > skip else durs
        rows.append((sum(d), name, grid, wg, len(d), sum(d) / len(d), min(d), max(d)))
    rows.sort(reverse=True)
    total = sum(r[0] for r in rows)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'grid', 'workgroup', 'calls', 'avg_us', 'min_us', 'max_us', 'total_us', 'percent'])
    for tot, name, grid, wg, n, avg, lo, hi in rows:
        w.writerow([name, 'x'.join(map(str, grid)), 'x'.join(map(str, wg)), n, f'{avg:.3f}', f'{lo:.3f}', f'{hi:.3f}',
                    f'{tot:.1f}', f'{100 * tot / total:.2f}'])


if __name__ == '__main__':
    main()
