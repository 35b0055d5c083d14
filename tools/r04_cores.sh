#!/bin/bash
# round 4: does a GEMM that leaves registers free for co-resident trunk waves shorten the pipelined step? + op inventory with shapes
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r04; mkdir -p $O
B="python3 bench.py --no-cpu-baseline --no-parity --no-bert512"
for i in 1 2; do
  $B > $O/cores_base_$i.json 2>/dev/null && RUART_HIP_LIB=build/libruart_hip_v224.so $B > $O/cores_v224_$i.json 2>/dev/null
done
python3 - <<'PY'
import json,glob
for f in sorted(glob.glob('gpurun_out/r04/cores_*.json')):
    d=json.loads(open(f).read().strip().splitlines()[-1]); r=d['roofline']
    print(f.split('/')[-1], d['ms_per_step'], 'timed avg us', r['avg_launch_us'], 'alone', r['alone']['avg_launch_us'])
PY
python3 tools/op_table.py --shapes > $O/op_table_shapes.log 2>&1; tail -95 $O/op_table_shapes.log | cut -c1-170
rocprofv3 --kernel-trace --output-format csv -d $O/kt -o p -- python3 bench.py --no-cpu-baseline --no-roofline --no-parity --no-bert512 --no-prefetch --steps 2 --warmup 1 > $O/kt.log 2>&1
python3 - <<'PY'
import csv,glob
seen={}
for f in glob.glob('gpurun_out/r04/kt/**/*kernel_trace.csv',recursive=True):
    for r in csv.DictReader(open(f)):
        k=r['Kernel_Name'][:80]
        if k not in seen:
            seen[k]=(r.get('VGPR_Count'),r.get('Accum_VGPR_Count'),r.get('SGPR_Count'),r.get('LDS_Block_Size'),r.get('Workgroup_Size_X') or r.get('Workgroup_Size'),r.get('Grid_Size_X') or r.get('Grid_Size'))
with open('gpurun_out/r04/kernel_resources.csv','w') as o:
    o.write('kernel,vgpr,agpr,sgpr,lds,wg,grid\n')
    for k,v in seen.items(): o.write('"%s",%s\n'%(k,','.join(str(x) for x in v)))
print(len(seen),'kernels')
PY
rm -rf $O/kt
