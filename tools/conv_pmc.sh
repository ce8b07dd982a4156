#!/bin/bash
# Regenerates profiles/<round>_conv_pmc.json: HBM traffic per launch of the dominant conv symbol (conv_regb_kernel<8,16,3,3, 2x2 waves>:
# convc2 + convf2 in one launch, and convm -- averaged over the symbol's dispatches like bench.py's `roofline.avg_launch_ms`), from TWO
# separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) over the default bench, corrected as MI355X_MICROARCH.md's HBM section
# prescribes and as tools/lookup_pmc.sh does (FETCH_SIZE doubled on gfx950 for 16-B-per-lane coalesced reads).  On the GPU box:
#   tools/conv_pmc.sh r03   ->  gpurun_out/r03_conv_pmc.json  (copy to profiles/ and commit)
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --steps 2 --warmup 1 --no-alt-precisions --no-alt-corr --no-cpu-baseline --no-ladder"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/cpmc_$c
    (cd $root && rocprofv3 --pmc $c --output-format csv -d /tmp/cpmc_$c -o p -- $cmd) > /tmp/cpmc_$c.log 2>&1
    f=$(find /tmp/cpmc_$c -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp "$f" /tmp/cpmc_$c.csv || { echo "pass $c failed"; tail -5 /tmp/cpmc_$c.log; exit 1; }
done
python - "$tag" "$root" <<'PY'
import csv, json, sys
from collections import defaultdict
tag, root = sys.argv[1], sys.argv[2]
def means(path, counter):
    acc = defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}
fetch = means("/tmp/cpmc_FETCH_SIZE.csv", "FETCH_SIZE")
write = means("/tmp/cpmc_WRITE_SIZE.csv", "WRITE_SIZE")
sym = next(k for k in fetch if "conv_regb_kernel<8, 16, 3, 3, 2, 3," in k)
cal = next((k for k in write if "inorm_apply_kernel" in k), None)
P = 135 * 240
f_kb, n = fetch[sym]
w_kb, _ = write[sym]
# algorithmic bytes, fp32 activations: convc2 + convf2 (read c1 256 ch + fl1 128 ch, write cf 256 ch) and convm (read cf 256, write 128)
algo = 0.5 * ((256 + 128 + 256) + (256 + 128)) * 4 * P
out = {"kernel": sym[:120], "resolution": [1080, 1920], "n_pix": P, "fetch_size_kb_raw": f_kb, "fetch_correction": 2.0,
       "write_size_kb": w_kb, "dispatches_averaged": n, "traffic_bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024)),
       "algorithmic_activation_bytes_per_launch": int(algo),
       "write_size_calibration_kb": {"inorm_apply_kernel": write[cal][0]} if cal else None,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over bench.py --steps 2 --warmup 1 (tools/conv_pmc.sh); "
               "mean over the symbol's dispatches (convc2 + convf2 in one launch, and convm).  FETCH_SIZE doubled per MI355X_MICROARCH.md; "
               "the weight stream (0.9-1.2 MB per layer, re-read by every workgroup) is served by L2 / MALL and is not in the algorithmic figure."}
json.dump(out, open(f"{root}/gpurun_out/{tag}_conv_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
