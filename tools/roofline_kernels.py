"""profiles/roofline_kernels.json from a per-grid kernel summary (tools/prof_by_grid.py output of `bench.py --no-overlap`):
the kernel bench.py's `roofline` names (the fused 3x3 coupling conv at level 0: wino_kernel<2, 8, 2>, grid 131072 x 1) and the
kernel that is dominant by time, each with its rocprofv3 average duration, and the commit the trace was taken at.
    python tools/roofline_kernels.py profiles/r04_cfg1_single_stream_by_grid.csv <commit> > profiles/roofline_kernels.json"""
import csv
import json
import sys


def main():
    path, commit = sys.argv[1], (sys.argv[2] if len(sys.argv) > 2 else 'unknown')
    rows = list(csv.DictReader(open(path)))
    named = next(r for r in rows if r['kernel'].startswith('void sininn::wino_kernel<2, 8, 2>') and r['grid'].startswith('131072x1'))
    by_time = rows[0]                                    # the file is sorted by total time

    def rec(r):
        return {'kernel': r['kernel'], 'grid': r['grid'], 'workgroup': r['workgroup'], 'calls': int(r['calls']), 'avg_us': float(r['avg_us']),
                'min_us': float(r['min_us']), 'max_us': float(r['max_us']), 'percent': float(r['percent'])}
    json.dump({'commit': commit, 'source': path, 'named': rec(named), 'by_time': rec(by_time)}, sys.stdout, indent=1)
    sys.stdout.write('\n')


if __name__ == '__main__':
    main()
