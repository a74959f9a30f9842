"""Print the headline and the per-class roofline of a bench.py JSON line (file holding the line, or stdin)."""
import json
import sys

line = open(sys.argv[1]).read() if len(sys.argv) > 1 else sys.stdin.read()
line = [l for l in line.splitlines() if l.startswith('{')][-1]
d = json.loads(line)
r = d.get('roofline') or {}
print(f"{d['metric']}: {d['value'] and round(d['value'], 1)} {d['unit']}  {d['ms_per_step']:.3f} ms/step  dtype {d['dtype']}  n_gpus {d['n_gpus']}")
if r:
    print(f"dominant kernel: {r['avg_ms'] * 1e3:.1f} us/launch  {r['achieved']:.1f} / {r['peak']:.0f} TF/s = {r['frac']:.3f}"
          + (f"  executed-MFMA {r['executed_mfma_frac']:.3f}" if 'executed_mfma_frac' in r else ''))
    if r.get('step'):
        print(f"whole step: {r['step']['alg_tflops']:.1f} TF/s algorithmic = {r['step']['frac_of_peak']:.3f} of peak; "
              f"single-stream step {r.get('single_stream_ms_per_step', 0):.3f} ms")
    tot = 0.0
    for c in r.get('classes') or []:
        tot += c['ms_per_step']
        print(f"  {c['ms_per_step']:8.3f} ms  x{c['launches_per_step']:<3d} {c.get('alg_tflops', 0):8.1f} TF/s  frac {c.get('frac', 0):.3f}"
              + (f" exec {c['executed_mfma_frac']:.3f}" if 'executed_mfma_frac' in c else '           ') + f"  {c['class']}")
    print(f"  {tot:8.3f} ms  executor kernels per step (single stream)")
if d.get('cpu_baseline'):
    print('cpu_baseline:', d['cpu_baseline']['value'], d['cpu_baseline']['unit'], 'on', d['cpu_baseline']['cores'], 'cores')
