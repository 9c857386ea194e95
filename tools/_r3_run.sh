python -m pytest tests/test_gpu_train.py tests/test_gpu_grad.py -x -q -m gpu -s 2>&1 | grep "bf16x3 vs\|passed\|failed\|Error" | head
python bench.py --train --batch 20 --steps 3 --warmup 1 --precision f32 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('f32   ', round(d['value'],1), 'crops/s', d['ms'])"
python bench.py --train --batch 20 --steps 3 --warmup 1 --precision bf16x3 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16x3', round(d['value'],1), 'crops/s', d['ms'])"
