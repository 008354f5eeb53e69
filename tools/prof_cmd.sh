#!/bin/bash
# rocprofv3 --kernel-trace --stats of an arbitrary command -> per-kernel table (ff:: kernels only) in <out>/kernel_stats.csv
#   tools/prof_cmd.sh <out_dir> <command...>
out=$1; shift
mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --stats --output-format csv -d "$out/prof" -o trace -- "$@" > "$out/cmd.out" 2> "$out/prof.err"
stats=$(find "$out/prof" -name '*kernel_stats.csv' | head -1)
python - "$stats" <<'PY' | tee "$out/kernel_stats.txt"
import csv, sys
rows = [r for r in csv.reader(open(sys.argv[1]))]
print(f"{'kernel':60s} {'calls':>6s} {'avg us':>9s} {'min us':>9s} {'max us':>9s}")
for r in rows[1:]:
    if "ff::" in r[0]:
        name = r[0].replace("void ", "").split("(")[0]
        print(f"{name[:60]:60s} {r[1]:>6s} {float(r[3])/1e3:9.1f} {float(r[5])/1e3:9.1f} {float(r[6])/1e3:9.1f}")
PY
tail -3 "$out/cmd.out"
rm -rf "$out/prof"
