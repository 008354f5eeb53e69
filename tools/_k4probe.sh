#!/bin/bash
# K4 timing probes (development): rebuild ff_merge.o with -DFF_K4_PROBE=<mask> on the GPU box, in-step timeline each.
#   tools/_k4probe.sh <out_dir> <mask> [<mask> ...]
out=$1; shift
mkdir -p "$out"
for m in "$@"; do
  rm -f framefusion_amd/csrc/ff_merge.o framefusion_amd/csrc/ff_similarity.o
  make -C framefusion_amd/csrc EXTRA="$m" > "$out/build_$(echo "$m" | tr -c 'A-Za-z0-9=\n' '_').log" 2>&1 || { echo "build failed for $m"; tail -5 "$out"/build_*.log; continue; }
  echo "## EXTRA=$m"
  tools/prof_step.sh "$out/run" 2>&1 | grep -E "K1|plan|K4|span|similarity|merge_compact|mean|median" | head -12
done
rm -f framefusion_amd/csrc/ff_merge.o framefusion_amd/csrc/ff_similarity.o
make -C framefusion_amd/csrc > /dev/null 2>&1
