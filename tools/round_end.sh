# Round-end evidence in one GPU call: `gpurun -- bash tools/round_end.sh <commit>`: the default bench line, tools/prof_all.sh (kernel traces, PMC passes,
# training trace, timeline) and the parity ablation; outputs under gpurun_out/ (copy gpurun_out/r04/* and the bench line into profiles/).
set -e
REV=${1:-final}
mkdir -p gpurun_out/r04
python bench.py > gpurun_out/r4_bench_final.json 2> gpurun_out/r4_bench_final.err
python -c "import json;d=json.loads(open('gpurun_out/r4_bench_final.json').read().strip().splitlines()[-1]);print(round(d['value'],2),'fps', round(d['roofline']['correlation_kernel']['launch_ms'],2),'ms corr; bf16', round(d['bf16_letter']['value'],2),'; harness', round(d['harness']['value'],2),'; train', round(d['train']['value'],1),'; cpu', d['cpu_baseline']['sample'][:120])"
bash tools/prof_all.sh $REV > gpurun_out/r4_prof_all.log 2>&1
tail -3 gpurun_out/r4_prof_all.log
python tools/ablate_parity.py --out gpurun_out/r04/ablate_parity.json > gpurun_out/r04/r04_parity_ablation.txt 2>&1
tail -3 gpurun_out/r04/r04_parity_ablation.txt | cut -c1-160
