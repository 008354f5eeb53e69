#!/bin/bash
# round-3 K5 A/B: shipped one-round score kernel (FF_K5_WGS=0) vs the streaming one at several workgroup counts
timeout 900 python -m pytest tests -m gpu -x -q -k "importance or attention or sdpa or c5 or c3 or golden or integration or prune or head_mean" 2>&1 | tail -3
FF_K5_WGS=512 timeout 900 python -m pytest tests -m gpu -x -q -k "importance or attention or sdpa or c5 or c3 or golden or integration or prune or head_mean" 2>&1 | tail -3
for wgs in 0 256 512 768 1024; do
  echo "## FF_K5_WGS=$wgs"
  FF_K5_WGS=$wgs tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8 2>&1 | grep -E "k_lq_|weights"
  FF_K5_WGS=$wgs tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 9011 --dim 3584 --H 28 --Hkv 4 --num 4 2>&1 | grep -E "k_lq_|weights"
  FF_K5_WGS=$wgs tools/prof_cmd.sh gpurun_out/k5p python tools/kbench_prune.py --S 13474 --dim 3584 --H 28 --Hkv 4 --num 1 2>&1 | grep -E "k_lq_|weights"
done
