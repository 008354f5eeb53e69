#!/bin/bash
# The committed evidence of a round: the two PMC passes (FETCH_SIZE and WRITE_SIZE in separate runs, kernel trace only -
# MI355X_MICROARCH.md "rocprofv3 PMC slots"), the kernel trace (stats + per-call timeline) and the bench line - in that
# order, so that the bench line's `roofline.traffic` comes from PMC passes of the very build it runs on.
#   FF_COMMIT=$(git rev-parse --short HEAD) tools/prof_round.sh <round tag, e.g. r03> [out_dir = gpurun_out/<tag>_profiles]
tag=$1; out=${2:-gpurun_out/${tag}_profiles}
mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
B="python bench.py --steps 40 --warmup 5 --no-cpu-baseline --no-extra --no-pmc"
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d "$out/prof/pmc_fetch" -o pmc -- $B > /dev/null 2> "$out/fetch.err"
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d "$out/prof/pmc_write" -o pmc -- $B > /dev/null 2> "$out/write.err"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof/trace" -o trace -- $B > /dev/null 2> "$out/trace.err"
python tools/summarise_prof.py "$out/prof" "$out" "$tag"
python -c "import json, sys; sys.path.insert(0, '.'); from framefusion_amd import _lib; print(json.dumps({'source_hash': _lib.source_hash(), 'commit': '${FF_COMMIT:-unknown}', 'command': '$B'}))" > "$out/${tag}_pmc_meta.json"
python tools/timeline.py "$(find "$out/prof/trace" -name '*kernel_trace.csv' | head -1)" > "$out/${tag}_timeline.txt"
# what bench.py reads its traffic figure from (profiles/ of THIS copy of the repo; committed from gpurun_out afterwards)
cp "$out/${tag}_pmc_fetch_summary.csv" "$out/${tag}_pmc_write_summary.csv" "$out/${tag}_pmc_meta.json" profiles/
python bench.py --steps 100 --warmup 20 > "$out/${tag}_bench.json" 2> "$out/bench.err"
cat "$out/${tag}_timeline.txt"; cat "$out/${tag}_pmc_fetch_summary.csv" "$out/${tag}_pmc_write_summary.csv"
rm -rf "$out/prof"
