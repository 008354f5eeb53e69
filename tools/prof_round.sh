#!/bin/bash
# The committed evidence of a round: bench line, kernel trace (stats + per-call timeline) and the two PMC passes
# (FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only - MI355X_MICROARCH.md "rocprofv3 PMC slots").
#   tools/prof_round.sh <round tag, e.g. r02> [out_dir = gpurun_out/<tag>_profiles]
tag=$1; out=${2:-gpurun_out/${tag}_profiles}
mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
python bench.py --steps 100 --warmup 20 > "$out/${tag}_bench.json" 2> "$out/bench.err"
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof/trace" -o trace -- $B > /dev/null 2> "$out/trace.err"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/prof/pmc_fetch" -o pmc -- $B > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/prof/pmc_write" -o pmc -- $B > /dev/null 2> "$out/write.err"
python tools/summarise_prof.py "$out/prof" "$out" "$tag"
python tools/timeline.py "$(find "$out/prof/trace" -name '*kernel_trace.csv' | head -1)" > "$out/${tag}_timeline.txt"
cat "$out/${tag}_timeline.txt"; cat "$out/${tag}_pmc_fetch_summary.csv" "$out/${tag}_pmc_write_summary.csv"
rm -rf "$out/prof"
