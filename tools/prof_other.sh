#!/bin/bash
# profiles/<tag>_other_calls.txt: rocprofv3 kernel stats (tools/prof_cmd.sh) of every call outside the headline step.
#   tools/prof_other.sh <tag> [out_dir = gpurun_out/<tag>_profiles]
tag=$1; out=${2:-gpurun_out/${tag}_profiles}
mkdir -p "$out"; f="$out/${tag}_other_calls.txt"
sec() { # title, command...
  title=$1; shift
  echo "## $title" >> "$f"
  tools/prof_cmd.sh "$out/tmp" "$@" > /dev/null 2>&1
  cat "$out/tmp/kernel_stats.txt" >> "$f"
  grep -v "amdgpu.ids" "$out/tmp/cmd.out" | tail -8 >> "$f"
  echo >> "$f"
  rm -rf "$out/tmp"
}
{
echo "# rocprofv3 --kernel-trace --stats (tools/prof_cmd.sh) of the calls outside the headline step, MI355X, round ${tag#r}."
echo "# Per-kernel averages over the command's launches, then the command's own output (hipEvent / host timings)."
echo "# Produced by tools/prof_other.sh $tag."
echo
} > "$f"
sec "threshold-regime prefill cascade at C2 (tools/kbench_cascade.py): merge 36898->19005, identity call, prune (head mean of [1,28,1,S] weights)" python tools/kbench_cascade.py
sec "C5 prune side (tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8): K5 importance kernels + prune call" python tools/kbench_prune.py --S 35072 --dim 8192 --H 64 --Hkv 8
sec "C3 prune side (--S 9011 --dim 3584 --H 28 --Hkv 4 --num 4)" python tools/kbench_prune.py --S 9011 --dim 3584 --H 28 --Hkv 4 --num 4
sec "K0 by-patch order (tools/kbench_order.py): closed form vs counting sort" python tools/kbench_order.py
sec "fixed-sparsity merging baseline, 28 layers x sparsity 0.1 at 64x576x4096 (tools/kbench_baseline.py)" python tools/kbench_baseline.py
sec "patch_type layout builders (tools/kbench_layout.py)" python tools/kbench_layout.py
sec "merge call per stage at the C5 / 7B / 128-frame shapes (tools/kbench.py)" bash -c "python tools/kbench.py --frames 64 --patches 576 --dim 8192 | tail -6; python tools/kbench.py --frames 64 --patches 210 --dim 3584 | tail -6; python tools/kbench.py --frames 128 --patches 576 --dim 4096 | tail -6"
cat "$f"
