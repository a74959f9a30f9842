"""Per-kernel SQ counter ratios from a rocprofv3 --pmc run (counter values divided by SQ_WAVE_CYCLES where that makes sense).

    rocprofv3 --pmc SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_VALU_MFMA_BUSY_CYCLES \
              SQ_BUSY_CU_CYCLES SQ_ACTIVE_INST_LDS --output-format csv -d gpurun_out/pmc -- python3 tools/bench_pair.py --reps 5
    python tools/pmc_summary.py gpurun_out/pmc conv_pair conv32_kernel<1 conv_mfma_kernel<1
"""
import collections
import csv
import glob
import os
import sys


def main():
    root, pats = sys.argv[1], sys.argv[2:]
    acc = collections.defaultdict(lambda: collections.defaultdict(list))
    for f in glob.glob(os.path.join(root, '**', '*counter_collection.csv'), recursive=True):
        for r in csv.DictReader(open(f)):
            k = r['Kernel_Name']
            if not pats or any(p in k for p in pats):
                acc[(k[:70], r['Grid_Size'])][r['Counter_Name']].append(float(r['Counter_Value']))
    for k, d in sorted(acc.items()):
        m = {c: sum(v) / len(v) for c, v in d.items()}
        wc = m.get('SQ_WAVE_CYCLES', 0) or 1.0
        print(k[0], 'grid', k[1])
        print('   ', ', '.join(f'{c[3:]} {v / wc:.3f}' for c, v in sorted(m.items()) if c != 'SQ_WAVE_CYCLES'), f'| wave quad-cycles {wc:.3g}')


if __name__ == '__main__':
    main()
