#!/bin/bash
# Regenerates profiles/<round>_lookup_pmc.json: HBM traffic per launch of the volume lookup (corr_lookup_kernel<4>), from
# TWO separate rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE: they do not fit one pass) over the volume-mode bench, exactly
# as MI355X_MICROARCH.md's HBM section prescribes (FETCH_SIZE doubled on gfx950 for 16-B-per-lane coalesced reads;
# WRITE_SIZE calibrated in the same run on a kernel of known output size).  Run on the GPU box:
#   tools/lookup_pmc.sh r02      ->  gpurun_out/r02_lookup_pmc.json  (copy to profiles/ and commit)
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
mkdir -p $root/gpurun_out
cd /tmp && export TMPDIR=/tmp
cmd="python $root/bench.py --corr volume --steps 2 --warmup 1 --no-alt-precisions --no-alt-corr --no-cpu-baseline --no-ladder"
for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf /tmp/lpmc_$c
    (cd $root && rocprofv3 --pmc $c --output-format csv -d /tmp/lpmc_$c -o p -- $cmd) > /tmp/lpmc_$c.log 2>&1
    f=$(find /tmp/lpmc_$c -name '*counter_collection.csv' | head -1)
    [ -n "$f" ] && cp "$f" /tmp/lpmc_$c.csv || { echo "pass $c failed"; tail -5 /tmp/lpmc_$c.log; exit 1; }
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
fetch = means("/tmp/lpmc_FETCH_SIZE.csv", "FETCH_SIZE")
write = means("/tmp/lpmc_WRITE_SIZE.csv", "WRITE_SIZE")
lk = next(k for k in fetch if "corr_lookup_kernel" in k)
cal = next((k for k in write if "inorm_apply_kernel" in k), None)
P = 135 * 240
f_kb, n = fetch[lk]
w_kb, _ = write[lk]
out = {"kernel": "corr_lookup_kernel<4>", "resolution": [1080, 1920], "n_pix": P, "fetch_size_kb_raw": f_kb,
       "fetch_correction": 2.0, "write_size_kb": w_kb, "dispatches_averaged": n,
       "traffic_bytes_per_launch": int(round((2.0 * f_kb + w_kb) * 1024)),
       "algorithmic_bytes_per_launch": 2896 * P,
       "write_size_calibration_kb": {"inorm_apply_kernel": write[cal][0]} if cal else None,
       "note": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over bench.py --corr volume --steps 2 "
               "--warmup 1 (tools/lookup_pmc.sh).  FETCH_SIZE doubled per MI355X_MICROARCH.md (gfx950 tallies the 128-B "
               "requests of 16-B-per-lane coalesced reads at 64 B; the lookup's volume reads are aligned float4 loads); "
               "WRITE_SIZE used as reported (calibration kernel: inorm_apply_kernel, whose output size is known)."}
json.dump(out, open(f"{root}/gpurun_out/{tag}_lookup_pmc.json", "w"), indent=1)
print(json.dumps(out))
PY
