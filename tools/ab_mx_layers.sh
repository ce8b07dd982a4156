q="--steps 60 --warmup 5 --no-alt-corr --no-alt-precisions --no-cpu-baseline --no-ladder"
pr() { python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('$1', round(d['value'],2), d.get('flow_epe_vs_cpu_oracle'))"; }
for rep in 1 2; do
for z in all auto; do WOFT_MX_LAYERS=$z python bench.py --precision f16mx8 $q 2>/dev/null | pr "f16mx8/$z"; done
python bench.py $q 2>/dev/null | pr bf16x3
done

