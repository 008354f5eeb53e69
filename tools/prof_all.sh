#!/bin/bash
# Everything a round commits under profiles/, in one gpurun call on the final sources:
#   FF_COMMIT=$(git rev-parse --short HEAD) tools/prof_all.sh r05
# order: per-configuration timelines (bench.py reads extra.configs[*].kernel_us from them), PMC passes + kernel trace + the bench line
# (prof_round.sh), the driver's short form, the two-samples-per-GPU line, everything outside the headline step.
tag=$1; out=gpurun_out/${tag}_profiles
mkdir -p "$out"
tools/prof_configs.sh "$tag" "$out" > "$out/prof_configs.log" 2>&1
tools/prof_round.sh "$tag" "$out" > "$out/prof_round.log" 2>&1
python bench.py --steps 20 --warmup 5 > "$out/${tag}_bench_short.json" 2> "$out/bench_short.err"
python bench.py --samples 2 --steps 100 --warmup 20 --no-cpu-baseline --no-extra --no-pmc > "$out/${tag}_bench_two_samples.json" 2> "$out/bench_s2.err"
python bench.py --steps 20 --warmup 5 --force-dist --no-cpu-baseline --no-extra --no-pmc > "$out/${tag}_rccl_one_rank.json" 2> "$out/bench_rccl.err"
tools/prof_other.sh "$tag" "$out" > "$out/prof_other.log" 2>&1
cp profiles/${tag}_timeline_*.txt "$out/" 2>/dev/null
ls "$out"
