#!/bin/bash
# rocprofv3 kernel trace of the threshold-regime cascade (tools/kbench_cascade.py): kernel list of the last prefill
out=$1; shift
mkdir -p "$out"; root=$(pwd)
cd /tmp && export TMPDIR=/tmp && cd "$root"
rocprofv3 --kernel-trace --output-format csv -d "$out/prof" -o trace -- python tools/kbench_cascade.py --reps 6 "$@" > "$out/cascade.txt" 2> "$out/prof.err"
trace=$(find "$out/prof" -name '*kernel_trace.csv' | head -1)
python - "$trace" <<'PY' > "$out/cascade_kernels.txt"
import csv, sys
rows = sorted(csv.DictReader(open(sys.argv[1])), key=lambda r: int(r["Start_Timestamp"]))
rows = rows[-40:]
t0 = int(rows[0]["Start_Timestamp"])
prev = None
for r in rows:
    s, e = int(r["Start_Timestamp"]), int(r["End_Timestamp"])
    name = r["Kernel_Name"].replace("void ", "").split("(")[0][:60]
    print(f"{(s - t0) / 1e3:9.1f} us  dur {(e - s) / 1e3:7.1f}  gap {((s - prev) / 1e3) if prev else 0:7.1f}  {name}")
    prev = e
PY
cat "$out/cascade.txt"; tail -30 "$out/cascade_kernels.txt"; rm -rf "$out/prof"
