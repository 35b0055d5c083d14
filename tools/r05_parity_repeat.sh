run() { env $1 python3 bench.py --no-prefetch --steps 10 --warmup 3 --no-cpu-baseline --no-bert512 2>/dev/null | python3 -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['parity']['max_abs_err_vs_reference'], d['parity']['mean_abs_err'])"; }
for i in 1 2 3 4; do run "X=1"; done
for i in 1 2 3; do run "RUART_LN_FOLD=0"; done
for i in 1 2 3; do run "RUART_STREAMS=0"; done
