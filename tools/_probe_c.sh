#!/bin/bash
timeout 1200 python -m pytest tests -m gpu -x -q -k "importance or attention or sdpa or c5 or c3 or golden or integration or prune or head_mean or plan_kernel or full_size" 2>&1 | tail -5
for dot in 0 1; do
  echo "## FF_K5_DOT=$dot"
  FF_K5_DOT=$dot tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8 2>&1 | grep -E "k_lq_|weights"
  FF_K5_DOT=$dot tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 13474 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
  FF_K5_DOT=$dot tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 4066 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
done
echo "## K4 slots at 128 x 576 x 4096 (rule = 7) vs 37 vs 11"
for s in 0 37 11 5; do
  FF_MERGE_SLOTS=$s python tools/kbench.py --frames 128 --patches 576 --dim 4096 2>&1 | grep -E "merge_compact"
done
echo "## K4 slots at 96 x 576 x 4096"
for s in 0 29 11; do
  FF_MERGE_SLOTS=$s python tools/kbench.py --frames 96 --patches 576 --dim 4096 2>&1 | grep -E "merge_compact"
done
echo "## soak importance 90 s"
timeout 300 python tests/soak_gpu.py importance 90 2>&1 | tail -2
