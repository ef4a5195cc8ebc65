#!/bin/bash
# round 3, GPU call 6: Wan VAE — shared frame buffers (memory), scaled hi/lo split; distributed workers with the shared communication stream
set +e
OUT=gpurun_out/r03_call6
mkdir -p $OUT
export TMPDIR=/tmp HSA_ENABLE_IPC_MODE_LEGACY=0
: > $OUT/summary.txt
timeout 900 python -m pytest tests/test_gpu_vae.py tests/test_gpu_dist.py tests/test_gpu_hunyuan_vae.py -q --timeout 600 > $OUT/pytest_vae_dist.log 2>&1; echo "pytest vae+dist rc=$?" | tee -a $OUT/summary.txt
tail -12 $OUT/pytest_vae_dist.log | cut -c1-300 >> $OUT/summary.txt
for cf in 2 4; do
  timeout 300 python tools/vae_bench.py --split --chunk-frames $cf 2>&1 | tail -1 | tee -a $OUT/summary.txt
done
timeout 300 python tools/vae_bench.py --chunk-frames 2 2>&1 | tail -1 | tee -a $OUT/summary.txt
timeout 300 python tools/vae_bench.py --conv16 --chunk-frames 2 2>&1 | tail -1 | tee -a $OUT/summary.txt
cat $OUT/summary.txt
