#!/bin/bash
# round 6: the load-order probe (tools/r06_load_order_probe.hip, built by hipcc into build/) and the full-size schedule test
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/r06; mkdir -p $O
timeout -k 10 120 build/r06_load_order_probe > $O/load_order_probe.log 2>&1; echo "probe rc $?"; cat $O/load_order_probe.log
timeout -k 10 500 python3 -m pytest tests/test_gpu_streams.py -x -q -s > $O/test_streams.log 2>&1; echo "streams tests rc $?"; grep -v amdgpu.ids $O/test_streams.log | tail -8
