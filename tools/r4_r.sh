set -e
O=gpurun_out/r4r
mkdir -p $O
python bench.py --cpu-baseline none --no-harness --steps 10 --warmup 3 > $O/bench.json 2> $O/bench.err || true
tail -3 $O/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r4r/bench.json').read().strip().splitlines()[-1])
print(round(d['value'],2),'fps')
r=d['roofline']
print({k:r[k] for k in ('bound','kernel','achieved','peak','frac','ms_per_frame','traffic') if k in r})
for k,v in r['families'].items(): print(k, {a:(round(b,3) if isinstance(b,float) else b) for a,b in v.items() if a in ('ms_per_frame','launches_per_frame','achieved','frac','unit')})
print('corr', round(r['correlation_kernel']['launch_ms'],3), round(r['correlation_kernel']['frac'],3))
print('bf16', d.get('bf16_letter',{}).get('value'), 'f32grade', d.get('f32_grade',{}).get('value'), 'train', d.get('train',{}).get('value'))
PY
