#!/bin/bash
# round 5, call 6: the torch-graphed trunk beside the encoder - which ingredient makes it slow (priority, CU mask, the encoder itself)?
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r05; mkdir -p $O
show() { python3 -c "
import json,sys
d=json.loads(open('$1').read().strip().splitlines()[-1])
t=d.get('timeline_ms') or {}
print('%-44s %6.2f ms  median %6.2f  enq %5.2f | fwd_end %s bwd_end %s enc_end %s gemm %s/%s' % ('$2', d['ms_per_step'], d['step_ms']['median'], d['step_ms'].get('host_enqueue_median',0), t.get('trunk_forward_end'), t.get('trunk_backward_end'), t.get('encoder_last_gemm_end'), t.get('gemm_us_beside_trunk'), t.get('gemm_us_after_trunk')))"; }
B="python3 bench.py --no-cpu-baseline --no-bert512 --no-parity"
run() { name=$1; shift; env "$@" $B $EXTRA > $O/m_$name.json 2> $O/m_$name.err && show $O/m_$name.json "$name" || tail -3 $O/m_$name.err; }
EXTRA="--graph-trunk 1"
run g1_default RUART_X=0
run g1_prio0 RUART_TRUNK_PRIORITY=0
run g1_priohigh RUART_TRUNK_PRIORITY=-1
run g1_nomask RUART_PREFETCH_CUS=0
run g1_nomask_prio0 RUART_PREFETCH_CUS=0 RUART_TRUNK_PRIORITY=0
run g1_1stream_prio0 RUART_STREAMS=0 RUART_TRUNK_PRIORITY=0
EXTRA="--graph-trunk 1 --no-prefetch"
run g1_inline RUART_X=0
EXTRA="--no-prefetch"
run g0_inline RUART_X=0
EXTRA=""
run g0_default RUART_X=0
run g0_1stream RUART_STREAMS=0
