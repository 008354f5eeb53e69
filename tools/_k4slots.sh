#!/bin/bash
# K4 slot-count sweep on an arbitrary command (development): -DFF_K4_SLOTS=<n> builds on the GPU box.
#   tools/_k4slots.sh "<command>" <grep pattern> <n> [<n> ...]   (n = 0: the shipped rule)
cmd=$1; pat=$2; shift 2
for n in "$@"; do
  rm -f framefusion_amd/csrc/ff_merge.o
  if [ "$n" = 0 ]; then make -C framefusion_amd/csrc > /dev/null 2>&1; else make -C framefusion_amd/csrc EXTRA="-DFF_K4_SLOTS=$n" > /dev/null 2>&1; fi
  echo "## slots=$n"
  tools/prof_cmd.sh gpurun_out/_slots bash -c "$cmd" 2>&1 | grep -E "$pat"
done
rm -f framefusion_amd/csrc/ff_merge.o; make -C framefusion_amd/csrc > /dev/null 2>&1
