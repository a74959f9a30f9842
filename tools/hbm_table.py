"""GB/s of the HBM-bound kernels of the training step from a per-(kernel, grid) rocprofv3 summary (tools/prof_by_grid.py
output of a SINGLE-STREAM run: `bench.py --no-overlap`), using the algorithmic bytes of BASELINE configs[1]
(B = 16, 256 x 256, -c 4, lr_window 10; DESIGN 3 gives the formulas).

    python tools/hbm_table.py profiles/r02_cfg1_single_stream_by_grid.csv > profiles/r02_cfg1_hbm_kernels.csv
"""
import csv
import sys

B, H, W = 16, 256, 256
E = B * 3 * H * W                       # elements of a flow tensor
M0, M1 = B * 64 * 64, B * 32 * 32
P = 3692416                             # parameters (-c 4)
PEAK = 8000.0

# kernel substring -> list of (grid or None, algorithmic bytes per launch, description)
SPEC = [
    ('adam_kernel', None, 7 * P * 4, 'fused Adam: p, g, m, v read; p, m, v written'),
    ('squeeze_rows_kernel', None, 2 * E * 4, 'squeeze / unsqueeze / permute: one read + one write of the flow tensor'),
    ('squeeze_kernel', None, 2 * E * 4, 'generic strided squeeze (fallback)'),
    ('sample_windows_dense_kernel', None, E * 1 + E * 4 + B * 32 * 32 * 84 * 5, 'u8 clip -> f32 batch (HR + LR window)'),
    ('sample_windows_kernel', None, E * 1 + E * 4 + B * 32 * 32 * 84 * 5, 'u8 clip -> f32 batch (generic strides)'),
    ('sqdiff_sum_kernel', None, None, 'loss sums (size depends on the call: HR 2E*4, LR 2*B*84*32*32*4)'),
    ('sqdiff_bwd_kernel', None, None, 'loss gradients'),
    ('coupling_bwd_kernel', None, None, 'coupling backward tail: 6 x M x Co floats'),
    ('wgrad_reduce', None, None, 'ordered slab reduce'),
    ('pack_batch_kernel', None, None, 'weight packs of the whole model (per optimiser step)'),
]


def main():
    rows = list(csv.DictReader(open(sys.argv[1])))
    w = csv.writer(sys.stdout)
    w.writerow(['kernel', 'grid', 'calls', 'avg_us', 'alg_bytes_per_launch', 'GB_per_s', 'frac_of_8TBs', 'what'])
    for r in rows:
        name = r['kernel']
        for key, grid, nbytes, what in SPEC:
            if key in name and (grid is None or grid == r['grid']):
                if 'coupling_bwd_kernel' in key:
                    g = int(r['grid'].split('x')[0])             # threads = M * Co / 4 (capped at 8192 blocks)
                    # distinguish levels by the launch grid: level 0 M0*24/4 threads, level 1 M1*96/4
                    nbytes = 6 * M0 * 24 * 4 if g == M0 * 24 // 4 else (6 * M1 * 96 * 4 if g == M1 * 96 // 4 else None)
                avg = float(r['avg_us'])
                gbs = nbytes / avg / 1e3 if nbytes else None
                w.writerow([name.split('(')[0][-60:], r['grid'], r['calls'], r['avg_us'], nbytes or '',
                            f'{gbs:.0f}' if gbs else '', f'{gbs / PEAK:.3f}' if gbs else '', what])
                break


if __name__ == '__main__':
    main()
