set -e
B="python bench.py --steps 20 --warmup 5 --no-harness --no-cpu-baseline"
$B > gpurun_out/r3_b_default.json 2>gpurun_out/r3_b_default.err
$B --knobs '{"stage": {"glue": {"precision": "bf16x3", "commute_any": true}}}' > gpurun_out/r3_b_glue.json 2>>gpurun_out/r3_b_default.err
$B --knobs '{"stage": {"glue": {"precision": "bf16x3", "commute_any": true}, "dec2": {"precision": "bf16x3"}}}' > gpurun_out/r3_b_glue_dec2.json 2>>gpurun_out/r3_b_default.err
for f in default glue glue_dec2; do python -c "import json;d=json.load(open('gpurun_out/r3_b_$f.json'));print('$f', round(d['value'],2),'fps', round(d['ms_per_step'],2),'ms  corr', round(d['roofline']['launch_ms'],2))"; done
python tools/ablate_parity.py --cases g10_fwd_40x60_mixed,g10_fwd_200x200_noref,g10_fwd_200x200,g10_fwd_100x100,g15_fwd_720p_noref,g16_fwd_480x640_mixed,g14_fwd_720p,g17_fwd_720p_edges --configs "bench mode,glue only,glue + dec2" --out gpurun_out/r3_ablate3.json > gpurun_out/r3_ablate3.log 2>&1; cut -c1-170 gpurun_out/r3_ablate3.log
