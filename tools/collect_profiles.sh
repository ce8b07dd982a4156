#!/bin/bash
# One pass over everything profiles/<round>_* is made of (run on the GPU box through gpurun; results -> gpurun_out/<tag>_*):
#   tools/collect_profiles.sh r02
tag=${1:-r04}
root=${GRAFT_REPO_ROOT:-/root/repo}
out=$root/gpurun_out
mkdir -p $out
cd $root
python bench.py > $out/${tag}_bench_default.json 2> $out/${tag}_bench_default.err
python bench.py --gpus 2 --steps 10 --warmup 3 > $out/${tag}_bench_2ranks_one_device.json 2>/dev/null
python bench.py --height 720 --width 1280 --no-cpu-baseline > $out/${tag}_bench_720p.json 2>/dev/null
python bench.py --height 720 --width 1280 --tracker-config WOFT_IRLS --no-cpu-baseline --no-alt-precisions > $out/${tag}_bench_720p_irls.json 2>/dev/null
python bench.py --iters 32 --precision bf16 --no-cpu-baseline --no-alt-precisions > $out/${tag}_bench_1080p_it32_bf16.json 2>/dev/null
python bench.py --iters 32 --no-cpu-baseline --no-alt-precisions > $out/${tag}_bench_1080p_it32_bf16x3.json 2>/dev/null
python bench.py --height 2160 --width 3840 --steps 10 --no-cpu-baseline --no-alt-precisions > $out/${tag}_bench_4k.json 2>/dev/null
python bench.py --height 480 --width 640 --no-cpu-baseline --no-ladder --no-alt-precisions --no-alt-corr > $out/${tag}_bench_480p.json 2>/dev/null
python tools/bench_hfit.py > $out/${tag}_hfit_fullframe.txt 2>/dev/null
{ python tools/bench_lookup.py; python tools/bench_lookup.py --storage bf16;
  for abl in 1 2 3; do LOOKUP_ABL=$abl python tools/bench_lookup.py; done; } 2>/dev/null | grep lookup > $out/${tag}_lookup_isolated.txt
{ OTF_ABL=16 python tools/bench_lookup_otf.py; OTF_ABL=64 python tools/bench_lookup_otf.py; for abl in 1 2 3 8 15; do OTF_ABL=$abl python tools/bench_lookup_otf.py; done; SMOOTH=0 python tools/bench_lookup_otf.py; SMOOTH=0 FLOW=8 python tools/bench_lookup_otf.py; } 2>/dev/null | grep -v amdgpu > $out/${tag}_lookup_otf_timeline.txt
python tools/layer_times.py 2>/dev/null | grep -v amdgpu > $out/${tag}_layer_times_bf16x3.txt
python tools/layer_times.py --precision fp32 2>/dev/null | grep -v amdgpu > $out/${tag}_layer_times_fp32.txt
python tools/regb_check.py 2>/dev/null | grep -v amdgpu > $out/${tag}_conv_kernels_ab.txt
tools/prof_stats.sh ${tag}_bench_default > /dev/null 2>&1
mv $out/${tag}_bench_default_kernel_stats.csv $out/${tag}_bench_kernel_stats_bf16x3.csv
tools/prof_stats.sh ${tag}_vol --corr volume > /dev/null 2>&1
mv $out/${tag}_vol_kernel_stats.csv $out/${tag}_bench_kernel_stats_bf16x3_volume.csv
tools/prof_stats.sh ${tag}_full --full-weight-head > /dev/null 2>&1
mv $out/${tag}_full_kernel_stats.csv $out/${tag}_bench_kernel_stats_bf16x3_full_weight_head.csv
tools/lookup_pmc.sh $tag > /dev/null 2>&1
cd /tmp && export TMPDIR=/tmp
rm -rf /tmp/pmc_busy
(cd $root && rocprofv3 --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE --output-format csv -d /tmp/pmc_busy -o p -- \
    python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-ladder --no-alt-precisions --no-alt-corr) > /tmp/pmc_busy.log 2>&1
f=$(find /tmp/pmc_busy -name '*counter_collection.csv' | head -1)
[ -n "$f" ] && python $root/tools/mfma_busy.py "$f" bf16x3 > $out/${tag}_pmc_mfma_util_conv.csv
ls -la $out | grep ${tag}_
