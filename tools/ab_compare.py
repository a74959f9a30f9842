"""Compare two variants in the output of several `tools/bench_kernels.py` runs captured in one log:
    for L in a.so b.so a.so b.so; do echo "$L"; SININN_LIB=$PWD/sin-inn_amd/$L python tools/bench_kernels.py ...; done > ab.log
    python tools/ab_compare.py ab.log
Lines that are not kernel timings are taken as the label of the following block; the best (minimum) time of every
kernel per label is compared.  Always A/B on ONE box: boxes differ by up to 10 %."""
import re, sys
blocks={}; cur=None
for l in open(sys.argv[1]):
    l=l.rstrip()
    m=re.match(r'(.*?)\s+([\d.]+) us',l)
    if m and cur: blocks[cur][-1][m.group(1).strip()]=float(m.group(2))
    elif l and not l.startswith('/opt') and not l.startswith('sum'): cur=l; blocks.setdefault(cur,[]); blocks[cur].append({})
keys=list(blocks)
a=blocks[keys[0]]; b=blocks[keys[1]]
ta=tb=0
for k in a[0]:
    pa=min(x[k] for x in a); pb=min(x[k] for x in b); ta+=pa; tb+=pb
    print(f'{k:45s} {keys[0][:14]} {pa:7.1f}  {keys[1][:14]} {pb:7.1f}  {100*(pb/pa-1):+5.1f}%')
print(f'total {ta:.1f} -> {tb:.1f}  {100*(tb/ta-1):+.1f}%')
