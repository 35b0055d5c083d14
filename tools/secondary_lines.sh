set -x
R=${R:-r04}
B="python bench.py --no-cpu-baseline --no-bert512"
run() { name=$1; shift; timeout -k 10 300 $B "$@" 2>gpurun_out/sec_$name.err | python -c "
import json,sys
d=json.loads(sys.stdin.read()); d['args']='$*'
print(json.dumps(d))" >> gpurun_out/${R}_secondary_lines.jsonl; tail -c 400 gpurun_out/sec_$name.err | tail -1; }
rm -f gpurun_out/${R}_secondary_lines.jsonl
run fp16 --precision fp16
run bf16 --precision bf16
run x3 --precision x3
run fp32 --precision fp32 --steps 6 --warmup 2
run fwd --mode fwd
run inline --no-prefetch
run h2d --include-h2d
run dp --force-dp
run stress --stress --steps 6 --warmup 2
run stress_bf16 --stress --precision bf16 --steps 6 --warmup 2
run frozen_dropout --frozen-dropout --no-parity
run frozen_dropout_x3 --frozen-dropout --precision x3 --no-parity --steps 6 --warmup 2
run unlock_x3 --unlock-bert --train-gemm x3 --steps 4 --warmup 2
run unlock_16gemm --unlock-bert --train-gemm 16gemm --steps 4 --warmup 2
run unlock16 --unlock-bert --steps 8 --warmup 3
run unlock16_dp --unlock-bert --force-dp --steps 8 --warmup 3
run unlock16_stress --stress --unlock-bert --steps 3 --warmup 1
timeout -k 10 300 python bench.py --mode bert512 --precision fp16 --steps 30 --warmup 10 > gpurun_out/${R}_bert512_line.json 2>/dev/null
timeout -k 10 400 python bench.py > gpurun_out/${R}_bench_line.json 2>gpurun_out/${R}_bench_line.err
cat gpurun_out/${R}_bench_line.json
R=$R python - <<'PY'
import json, os
R = os.environ['R']
for l in open('gpurun_out/%s_secondary_lines.jsonl' % R):
    d=json.loads(l); print("%-45s %9.1f samples/s %8.2f ms  parity %s" % (d['args'], d['value'], d['ms_per_step'], (d.get('parity') or {}).get('max_abs_err_vs_reference')))
d=json.load(open('gpurun_out/%s_bert512_line.json' % R)); print('bert512', d['value'], d['ms_per_step'], d['roofline']['one_pass'])
PY
