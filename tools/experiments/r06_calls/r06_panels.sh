#!/bin/bash
# round 6, call 3: tile-order panels of P column tiles (RON_PANEL_COLS) on every four-wave launch: per-launch time (one in flight) and
# fabric traffic (PMC) for P = default / 0 (column tiles fastest) / 1 (row tiles fastest) / 2 / 4 / 8, and the default step with two in flight
set -u
export TMPDIR=/tmp
export GPU_MAX_HW_QUEUES=8
O=gpurun_out/r06_panels
mkdir -p $O
for P in default 0 1 2 3 4 8; do
  if [ "$P" = default ]; then unset RON_PANEL_COLS; else export RON_PANEL_COLS=$P; fi
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 --in-flight 1 --layers $O/layers_P$P.txt > $O/bench_if1_P$P.json 2>> $O/err.txt
  python3 bench.py --no-cpu-baseline --no-parity-mode --sustained-seconds 0 > $O/bench_if2_P$P.json 2>> $O/err.txt
  bash tools/pmc_bench.sh $O/pmc_P$P --in-flight 1 > $O/pmc_P$P.log 2>&1
  cp $O/pmc_P$P/traffic_layers_full_bf16_bs32.txt $O/traffic_layers_P$P.txt
  rm -rf $O/pmc_P$P/pmc_fetch $O/pmc_P$P/pmc_write
done
unset RON_PANEL_COLS
python3 - <<'PY'
import json, glob, re
O='gpurun_out/r06_panels'
Ps=['default','0','1','2','3','4','8']
def layers(f):
    out={}
    for l in open(f):
        p=l.split()
        if len(p)>=5 and not l.startswith('#') and p[0]!='launch': out[p[0]]=float(p[2])
    return out
def traffic(f):
    out={}
    for l in open(f):
        p=l.split()
        if len(p)>=6 and not l.startswith('#') and p[0] not in ('launch','TOTAL'):
            try: out[p[0]]=float(p[-1])
            except ValueError: pass
        if p and p[0]=='TOTAL': out['TOTAL']=float(p[-1])
    return out
L={P:layers('%s/layers_P%s.txt'%(O,P)) for P in Ps}
T={P:traffic('%s/traffic_layers_P%s.txt'%(O,P)) for P in Ps}
print('%-34s'%'launch (us | MB)'+''.join('%16s'%('P='+P) for P in Ps))
for k in L['default']:
    print('%-34s'%k[:34]+''.join('%8.1f %7.0f'%(L[P].get(k,0),T[P].get(k,0)) for P in Ps))
print('%-34s'%'TOTAL'+''.join('%8.1f %7.0f'%(sum(L[P].values()),T[P].get('TOTAL',0)) for P in Ps))
for P in Ps:
    for m in ('if1','if2'):
        try:
            d=json.loads([l for l in open('%s/bench_%s_P%s.json'%(O,m,P)) if l.startswith('{')][-1]); print('P',P,m,round(d['value'],1))
        except Exception as e: print(P,m,'ERR',e)
PY
