#!/bin/bash
# session probe: K4 slots-per-workgroup sweep below the shipped range at C2 (in-step timeline), bare reads at K's sizes,
# Qwen2-VL e2e prefill, importance soak, full GPU suite
echo "## membench 72 MB / 9 MB"
./tools/membench/membench 72 2>&1 | head -16
./tools/membench/membench 9 2>&1 | head -6
for s in 0 7 11 13 17 19 23 29; do
  echo "## FF_MERGE_SLOTS=$s (C2 step)"
  FF_MERGE_SLOTS=$s tools/prof_step.sh gpurun_out/slots_$s 2>&1 | grep -E "calls;|k_pair|k_plan|k_merge"
done
echo "## 7B shape / C5 shape stage timing"
for s in 0 7 11; do
  FF_MERGE_SLOTS=$s python tools/kbench.py --frames 64 --patches 210 --dim 3584 2>&1 | grep -E "merge_compact|similarity"
  FF_MERGE_SLOTS=$s python tools/kbench.py --frames 64 --patches 576 --dim 8192 2>&1 | grep -E "merge_compact|similarity"
done
echo "## e2e qwen2vl"
python bench.py --e2e qwen2vl --steps 20 --warmup 5 --no-cpu-baseline --no-extra > gpurun_out/e2e_qwen2vl.json 2> gpurun_out/e2e_qwen2vl.err; tail -3 gpurun_out/e2e_qwen2vl.err; python -c "
import json; d=json.load(open('gpurun_out/e2e_qwen2vl.json')); e=d['extra']['e2e_prefill_qwen2vl']; print(e['workload'], e['dense_prefill_ms'])
for r in e['regimes']: print(r)"
echo "## soak importance"
timeout 400 python tests/soak_gpu.py importance 120 2>&1 | tail -4
echo "## full gpu suite"
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -4
