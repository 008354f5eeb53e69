#!/bin/bash
timeout 900 python -m pytest tests -m gpu -x -q -k "importance_dot or importance_owner or importance_golden or importance_attn or transposed" 2>&1 | tail -15
run() { echo "## $*"; env "$@" tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8 2>&1 | grep -E "k_lq_dot|k_lq_mfma|k_lq_finish"; }
run FF_K5_U=2 FF_K5_D=1
run FF_K5_U=2 FF_K5_D=2
run FF_K5_U=1 FF_K5_D=1
run FF_K5_U=1 FF_K5_D=2
run FF_K5_U=1 FF_K5_D=2 FF_K5_PROBE=8
run FF_K5_U=1 FF_K5_D=1 FF_K5_KPW=160
run FF_K5_U=1 FF_K5_D=2 FF_K5_KPW=160
run FF_K5_U=1 FF_K5_D=2 FF_K5_KPW=96
run FF_K5_U=1 FF_K5_D=1 FF_K5_PROBE=7
echo "## 7B shape"
for v in "FF_K5_U=2 FF_K5_D=1" "FF_K5_U=1 FF_K5_D=1" "FF_K5_U=1 FF_K5_D=2"; do echo "# $v"; env $v tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 13474 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_dot|k_lq_finish"; done
