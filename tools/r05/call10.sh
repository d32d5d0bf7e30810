#!/bin/bash
# round 5, GPU call 10: k-tile depth / ring length of the fp32 wgrad+update kernel at 256 frames (dev build, BP_WGRAD_VARIANT)
O=gpurun_out/r05c10; mkdir -p $O
DEV=$PWD/dnn-for-speech-enhancement_amd/libbp_hip_dev.so
for v in 1 2; do
  BP_HIP_LIB=$DEV BP_WGRAD_VARIANT=$v timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "golden or full_size or seeded" > $O/pytest_v$v.log 2>&1; echo "variant $v parity rc=$?"; tail -1 $O/pytest_v$v.log
done
ab() {
  BP_HIP_LIB=$DEV BP_WGRAD_VARIANT=$1 timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('variant $1 %.4f ms/step  wgrad in-step %.2f us' % (j['ms_per_step'], 1e3*j['roofline']['kernels_in_step_ms']['wgrad_update_grouped']))"
}
for i in 1 2 3; do ab 0; ab 1; ab 2; done
