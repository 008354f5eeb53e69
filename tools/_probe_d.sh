#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "importance_dot or importance_owner or importance_golden or importance_attn or transposed" 2>&1 | tail -3
run() { echo "## $*"; env "$@" tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8 2>&1 | grep -E "k_lq_dot|k_lq_mfma"; }
run FF_K5_PROBE=0
run FF_K5_PROBE=1
run FF_K5_PROBE=2
run FF_K5_PROBE=4
run FF_K5_PROBE=7
run FF_K5_KPW=64
run FF_K5_KPW=128
run FF_K5_KPW=192
run FF_K5_KPW=512
run FF_K5_U=4
run FF_K5_U=4 FF_K5_KPW=128
run FF_K5_U=1 FF_K5_KPW=64
run FF_K5_U=1
echo "## small shapes, default"
tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 13474 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 4066 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
FF_K5_KPW=64 tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 13474 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
FF_K5_KPW=64 tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 4066 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
