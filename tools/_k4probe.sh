#!/bin/bash
# Kernel timing probes (development): rebuild ff_merge.o / ff_similarity.o with extra -D switches on the GPU box, in-step
# timeline of bench.py each.  The switches themselves (FF_K4_PROBE, FF_K1_PROBE, FF_K4_SLOTS, FF_K4_LOOK ...) are added to the
# sources for one run and removed again: profiles/r03_k4_probes.txt and r03_k1_probes.txt say which ones were used.
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
