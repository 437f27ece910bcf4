#!/bin/bash
# On the GPU box: for every experiment library exp/NAME (arguments), phase cycles of both kernel forms + bench time.
PKG=paper_gorbani_2025_humanoids_multi-rate-mpc-ironcub_amd
mkdir -p gpurun_out/exp
for name in "$@"; do
  cp exp/$name/libvsmpc.so $PKG/libvsmpc.so || exit 1
  for form in throughput latency; do
    VSMPC_FORM=$form timeout -k 10 120 python tools/gpu_phases.py > gpurun_out/exp/${name}_${form}.txt 2>&1 || exit 1
    VSMPC_FORM=$form timeout -k 10 120 python bench.py --no-cpu-baseline --no-extra --no-latency > gpurun_out/exp/${name}_${form}.json 2>/dev/null || exit 1
    echo "$name $form: $(python -c "import json;d=json.load(open('gpurun_out/exp/${name}_${form}.json'));print(d['roofline']['kernel_us_per_launch'])") us; $(grep -E 'total cycles' gpurun_out/exp/${name}_${form}.txt | head -1 | cut -c1-70)"
    grep -E "P1 condense|P3 chol|P1 MFMA|P1 barrier" gpurun_out/exp/${name}_${form}.txt | head -4
  done
done
