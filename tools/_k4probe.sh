#!/bin/bash
# Kernel timing probes (development): rebuild ff_merge.o / ff_similarity.o with extra -D switches on the GPU box, in-step
# timeline of bench.py each.
#   tools/_k4probe.sh <out_dir> "<EXTRA flags>" ["<EXTRA flags>" ...]
out=$1; shift
mkdir -p "$out"
for m in "$@"; do
  rm -f framefusion_amd/csrc/ff_merge.o framefusion_amd/csrc/ff_similarity.o
  make -C framefusion_amd/csrc EXTRA="$m" > "$out/build.log" 2>&1 || { echo "build failed for $m"; tail -5 "$out/build.log"; continue; }
  echo "## EXTRA=$m"
  tools/prof_step.sh "$out/run" 2>&1 | grep -E "calls;|  [012] k_" | head -4
done
rm -f framefusion_amd/csrc/ff_merge.o framefusion_amd/csrc/ff_similarity.o
make -C framefusion_amd/csrc > /dev/null 2>&1
