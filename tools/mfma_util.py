"""Matrix-pipe utilisation per (kernel, grid) from a rocprofv3 --pmc pass (north_star: "rocprof ... MFMA utilisation ... against gfx950 peak").

    rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES GRBM_GUI_ACTIVE --output-format csv -d OUT -- python3 bench.py --no-overlap ...
    python tools/mfma_util.py OUT > profiles/rNN_cfg1_mfma_util.csv

SQ_VALU_MFMA_BUSY_CYCLES sums the pipe cycles of every MFMA the dispatch issued (32 per v_mfma_f32_16x16x4_f32 / 32x32x16_bf16, 64
per 32x32x2_f32; checked against the instruction count of conv_sub1_bwd_kernel: 3.146 M MFMAs x 32 = 1.007e8, counter 1.007e8).
GRBM_GUI_ACTIVE counts the cycles the dispatch kept the GPU busy, summed over the 8 XCDs.  With 1024 SIMDs (256 CUs x 4):
    mfma_util = MFMA_BUSY_CYCLES / (GRBM_GUI_ACTIVE / 8 * 1024)
i.e. the fraction of the chip's matrix-pipe cycles that carried an MFMA while the kernel ran -- the executed-FLOP roofline fraction
measured by the hardware instead of derived from a FLOP count and a duration."""
import collections
import csv
import glob
import os
import sys


def main():
    root = sys.argv[1]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            acc[(r['Kernel_Name'], r['Grid_Size'], r['Workgroup_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    rows = []
    for (k, grid, wg), d in acc.items():
        m = {c: sum(v) / len(v) for c, v in d.items()}
        busy, gui = m.get('SQ_VALU_MFMA_BUSY_CYCLES', 0.0), m.get('GRBM_GUI_ACTIVE', 0.0)
        if busy <= 0 or gui <= 0:
            continue
        n = len(d['SQ_VALU_MFMA_BUSY_CYCLES'])
        rows.append((busy * n, k, grid, wg, n, busy, gui / 8.0, busy / (gui / 8.0 * 1024.0), m.get('SQ_BUSY_CU_CYCLES', 0.0), m.get('SQ_WAVE_CYCLES', 0.0)))
    rows.sort(reverse=True)
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'grid', 'workgroup', 'dispatches', 'mfma_busy_cycles', 'gpu_active_cycles_per_xcd', 'mfma_util', 'sq_busy_cu_cycles', 'sq_wave_quad_cycles'])
    for _, k, grid, wg, n, busy, gui, util, cu, wc in rows:
        w.writerow([k, grid, wg, n, f'{busy:.4g}', f'{gui:.4g}', f'{util:.3f}', f'{cu:.4g}', f'{wc:.4g}'])


if __name__ == '__main__':
    main()
