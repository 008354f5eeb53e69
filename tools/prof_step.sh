#!/bin/bash
# rocprofv3 kernel trace of `python bench.py` -> per-call timeline + per-kernel stats (development tool).
#   tools/prof_step.sh <out_dir> [extra bench.py args]
out=$1; shift
mkdir -p "$out"
root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o trace -- \
    python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-pmc "$@" > "$out/prof_bench.json" 2> "$out/prof.err"
trace=$(find "$out/prof" -name '*kernel_trace.csv' | head -1)
stats=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
python tools/timeline.py "$trace" > "$out/timeline.txt"
grep -E "Name|ff::" "$stats" | sed 's/void //' > "$out/kernel_stats.csv"
cat "$out/timeline.txt"
cut -d, -f1-4 "$out/kernel_stats.csv" | head -20
rm -rf "$out/prof"
