#!/bin/bash
# round 5: concurrent batch groups of the multi-block Lanczos (block_batch_groups) on BASELINE config 4, same session A/B
cd "$GRAFT_REPO_ROOT" || exit 1
mkdir -p gpurun_out
for g in 1 2 4 1 2 4; do
  timeout 300 python bench.py --workload mimo --steps 60 --warmup 10 --no-cpu --block-batch-groups $g 2> gpurun_out/mimo_ab_g$g.err | python -c "
import sys, json
d = json.loads(sys.stdin.read().strip().splitlines()[-1])
print('groups $g: %.1f it/s  %.3f ms/step  matvecs/step %.0f restarts/step %.1f' % (d['value'], d['ms_per_step'], d['lanczos_matvecs_per_step'], d['lanczos_restarts_per_step']))
" >> gpurun_out/mimo_ab.log
done
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_sharded_gpu.py -x -q -m gpu -s -k "config4 or mimo or multi_block or two_shards" > gpurun_out/mimo_tests.log 2>&1; echo "rc $?" >> gpurun_out/mimo_tests.log
cat gpurun_out/mimo_ab.log; tail -12 gpurun_out/mimo_tests.log
