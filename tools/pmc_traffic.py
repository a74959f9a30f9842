"""HBM traffic of one kernel from two rocprofv3 PMC passes (FETCH_SIZE and WRITE_SIZE collected separately, as
MI355X_MICROARCH.md prescribes: they do not fit one pass).  FETCH_SIZE / WRITE_SIZE are reported in KiB-like units of
1024 bytes... (rocprofv3 reports them in KB); FETCH_SIZE is doubled per the gfx950 rule for wide coalesced reads.

    python tools/pmc_traffic.py <fetch_dir> <write_dir> <kernel substring> <grid size> > profiles/traffic_dominant.json
"""
import csv
import glob
import json
import os
import sys


def mean_counter(root, name, kernel, grid):
    """grid 'auto': the grid size with the largest total counter value among the kernel's dispatches (the dominant launch shape);
    returns (mean, dispatches, grid)."""
    by_grid = {}
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == name and kernel in r['Kernel_Name']:
                by_grid.setdefault(str(r['Grid_Size']), []).append(float(r['Counter_Value']))
    if str(grid) == 'auto' and by_grid:
        grid = max(by_grid, key=lambda g: sum(by_grid[g]))
    vals = by_grid.get(str(grid), [])
    if not vals:
        sys.exit(f'no {name} rows for {kernel} grid {grid} under {root} (grids seen: {sorted(by_grid)})')
    return sum(vals) / len(vals), len(vals), grid


def main():
    fetch_dir, write_dir, kernel, grid = sys.argv[1:5]
    fetch_kb, nf, grid = mean_counter(fetch_dir, 'FETCH_SIZE', kernel, grid)
    write_kb, nw, _ = mean_counter(write_dir, 'WRITE_SIZE', kernel, grid)
    out = {'kernel': f'{kernel}, grid {grid}', 'dispatches_averaged': [nf, nw],
           'fetch_size_kb': fetch_kb, 'write_size_kb': write_kb,
           'hbm_bytes_per_launch': 2.0 * fetch_kb * 1024 + write_kb * 1024,
           'source': 'rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE, separate passes over `bench.py --no-overlap` (tools/refresh_profiles.sh); FETCH_SIZE doubled per MI355X_MICROARCH.md (HBM section, gfx950 rule)',
           'commit': os.popen('git rev-parse --short HEAD 2>/dev/null').read().strip() or os.environ.get('SININN_COMMIT', 'unknown')}
    print(json.dumps(out, indent=1))


if __name__ == '__main__':
    main()
