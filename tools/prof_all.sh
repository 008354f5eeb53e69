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
python bench.py > "$out/${tag}_bench_default.json" 2> "$out/bench_default.err"
python tools/kernel_fractions.py "$tag" > "$out/${tag}_kernel_fractions.txt" 2> "$out/fractions.err"
# the one-launch merge call against the three launches, whole cascades, host included (same process, alternating)
{
  echo "# tools/flow_ab.py: FrameFusion.one_launch on / off inside bench.cascade (default instance, exactly sized outputs); us of one"
  echo "# isolated cascade (synchronise before and after) / per cascade back to back; MI355X, sources $(python -c "import sys; sys.path.insert(0, '.'); from framefusion_amd import _lib; print(_lib.source_hash())")"
  python tools/flow_ab.py --configs 7b 7b32 7b128 c3 2>&1 | grep -v amdgpu.ids
  echo
  echo "# tools/flow_stamps.py: where the host's time goes inside one default-instance call of the 7B layout, back to back"
  python tools/flow_stamps.py 2>&1 | grep -v amdgpu.ids
  echo
  echo "# the same through the three launches"
  python tools/flow_stamps.py --three 2>&1 | grep -v amdgpu.ids
  echo
  echo "# tools/flow_residual.py: call A + call B (the decoder's residual add fused in) of one prefill, back to back"
  python tools/flow_residual.py --config 7b 2>&1 | grep -v amdgpu.ids
  python tools/flow_residual.py --config c3 2>&1 | grep -v amdgpu.ids
} > "$out/${tag}_flow.txt"
# device time of the one-launch kernel without the tracer: per-workgroup stamps (a library built with -DFF_RES_WGSTAMPS, then the
# product build again)
{
  echo "# tools/res_stamps.py (top-k branch: first mail slot; --p_change 0.6: threshold branch, second slot) and tools/flow_stamps.py --wg"
  echo "# on a library built with EXTRA=-DFF_RES_WGSTAMPS (first workgroup start -> last workgroup end, device clock)"
  python tools/res_stamps.py 2>&1 | grep -v amdgpu.ids | tail -5
  python tools/res_stamps.py --p_change 0.6 2>&1 | grep -v amdgpu.ids | tail -3
  make -B -C framefusion_amd/csrc EXTRA=-DFF_RES_WGSTAMPS > /dev/null 2>&1
  python tools/flow_stamps.py --wg 2>&1 | grep -v amdgpu.ids | head -14
  make -B -C framefusion_amd/csrc > /dev/null 2>&1
} > "$out/${tag}_resident_stamps.txt"
python bench.py --e2e all --steps 20 --warmup 5 --no-cpu-baseline --no-extra --no-pmc > "$out/${tag}_e2e_prefill.json" 2> "$out/e2e.err"
cp profiles/${tag}_timeline_*.txt "$out/" 2>/dev/null
ls "$out"
