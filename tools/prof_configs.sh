#!/bin/bash
# profiles/<tag>_timeline_{7b,c3,c5}.txt: per-kernel timelines (durations + gaps) of the prefill cascades of BASELINE.json's other
# configurations, back to back - what extra.configs[*].us_back_to_back of the bench line is made of.
#   tools/prof_configs.sh <tag, e.g. r04> [out_dir = gpurun_out/<tag>_profiles]
tag=$1; out=${2:-gpurun_out/${tag}_profiles}
mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
hash=$(python -c "import sys; sys.path.insert(0, '.'); from framefusion_amd import _lib; print(_lib.source_hash())")
for cfg in 7b c3 c5 c5topk c2thr; do
  rm -rf "$out/prof_$cfg"
  rocprofv3 --kernel-trace --output-format csv -d "$out/prof_$cfg" -o trace -- python tools/trace_config.py --config $cfg --reps 20 > "$out/trace_$cfg.json" 2> "$out/trace_$cfg.err"
  trace=$(find "$out/prof_$cfg" -name '*kernel_trace.csv' | head -1)
  n=$(python -c "import json,sys; print(json.loads([l for l in open('$out/trace_$cfg.json') if l.startswith('{')][-1])['back_to_back_cascades'])")
  {
    echo "# tools/timeline_cascade.py over rocprofv3 --kernel-trace -- python tools/trace_config.py --config $cfg --reps 20; MI355X; sources $hash (commit ${FF_COMMIT:-unknown})"
    python -c "import json; r=json.loads([l for l in open('$out/trace_$cfg.json') if l.startswith('{')][-1]); print('# calls:', ' '.join(r['calls']), '| host clock: isolated', round(r['us'],1), 'us, back to back', round(r['us_back_to_back'],1), 'us per cascade; algorithmic bytes', r['algorithmic_bytes'])"
    python tools/timeline_cascade.py "$trace" "$n"
  } > "$out/${tag}_timeline_$cfg.txt"
  cat "$out/${tag}_timeline_$cfg.txt"
  cp "$out/${tag}_timeline_$cfg.txt" profiles/      # (what bench.py reads extra.configs[*].kernel_us from; committed from gpurun_out afterwards)
  rm -rf "$out/prof_$cfg"
done
