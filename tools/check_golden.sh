#!/bin/bash
# Regenerate every tests/golden/*.npz from the reference (/root/reference, build container only) into a temp dir and compare
# array by array with the committed fixtures: the pin is reproducible when every array is bit-identical.
#   tools/check_golden.sh [--keep]     (exit 0: identical; 1: differences listed)
set -e
root=$(cd "$(dirname "$0")/.." && pwd)
tmp=$(mktemp -d /tmp/ff_golden.XXXXXX)
export FF_GOLDEN_DIR=$tmp
cd "$root"
for gen in make_golden.py make_golden_mask.py make_golden_c1.py make_golden_baseline.py make_golden_layout.py make_golden_full.py; do
  echo "== oracle/$gen"
  python oracle/$gen > "$tmp/$gen.log" 2>&1 || { tail -20 "$tmp/$gen.log"; exit 2; }
done
set +e
python - "$tmp" "$root/tests/golden" <<'PY'
import sys, os, numpy as np
new, old = sys.argv[1], sys.argv[2]
total = diff = 0
for f in sorted(os.listdir(old)):
    if not f.endswith(".npz"):
        continue
    a, b = np.load(os.path.join(old, f)), np.load(os.path.join(new, f))
    if set(a.files) != set(b.files):
        print(f"{f}: array names differ: {sorted(set(a.files) ^ set(b.files))[:8]}")
        diff += 1
    for k in sorted(set(a.files) & set(b.files)):
        total += 1
        x, y = a[k], b[k]
        if x.shape != y.shape or x.dtype != y.dtype or not np.array_equal(x, y, equal_nan=(x.dtype.kind == "f")):
            n = int((x != y).sum()) if x.shape == y.shape else -1
            print(f"{f}: {k}: differs ({n} of {x.size} elements)")
            diff += 1
print(f"{total} arrays compared, {diff} differ")
sys.exit(1 if diff else 0)
PY
rc=$?
[ "$1" == "--keep" ] && echo "kept $tmp" || rm -rf "$tmp"
exit $rc
