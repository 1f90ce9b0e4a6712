#!/bin/bash
# GPU box: every fuzzer in guard mode on the round's last tree; the summary lines go to profiles/rNN_fuzz_campaign_guard_mode.txt
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r4
export DOPF_GUARD=1
( timeout -k 10 500 python scripts/fuzz_lean.py 400 101 | tail -n 1 ) > gpurun_out/r4/camp.txt 2>&1
( timeout -k 10 300 python scripts/fuzz_parity.py 1500 102 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
( timeout -k 10 300 python scripts/fuzz_free.py 300 103 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
( GPU_MAX_HW_QUEUES=8 timeout -k 10 300 python scripts/fuzz_sharded.py 300 104 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
( timeout -k 10 300 python scripts/fuzz_quiet.py 150 105 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
( timeout -k 10 400 python scripts/fuzz_net_wide.py 60 106 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
( ORACLE_MODE=0 timeout -k 10 300 python scripts/fuzz_parity.py 200 107 | tail -n 1 ) >> gpurun_out/r4/camp.txt 2>&1
cat gpurun_out/r4/camp.txt
