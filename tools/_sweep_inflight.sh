for f in 1 2 3; do
  python bench.py --no-cpu-baseline --no-harness --no-extras --inflight $f 2>/dev/null | tail -1 > /tmp/b_$f.json
  python -c "import json; d=json.load(open('/tmp/b_$f.json')); print('inflight $f', round(d['value'],2), 'fps  corr', round(d['roofline']['launch_ms'],2), 'ms frac', round(d['roofline']['frac'],3))"
done
