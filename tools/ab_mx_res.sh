# f16mx8 (default layer choice) against bf16x3 at the other resolutions of the profile pass, one call
q="--warmup 5 --no-alt-corr --no-alt-precisions --no-cpu-baseline --no-ladder"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2))"; }
for rep in 1 2; do
for res in "480 640 100" "720 1280 100" "2160 3840 12"; do set -- $res
  python bench.py --height $1 --width $2 --steps $3 $q 2>/dev/null | pr "$1x$2 bf16x3"
  python bench.py --height $1 --width $2 --steps $3 --precision f16mx8 $q 2>/dev/null | pr "$1x$2 f16mx8"
done; done
