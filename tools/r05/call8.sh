#!/bin/bash
# round 5, GPU call 8: split-K output layer finished in ONE launch (bp_out_splitk): parity + same-box A/B against the two-launch form
O=gpurun_out/r05c8; mkdir -p $O
timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_shim.py tests/test_bpforward.py tests/test_bptrain.py -m gpu -x -q > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
ab() {
  local label=$1 lib=$2
  BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib timeout 300 python bench.py --gpus 1 --steps 2000 --warmup 100 --no-cpu-baseline --no-extras --sustained-s 0 2>/dev/null | python -c "import sys,json; j=json.loads([l for l in sys.stdin if l.startswith('{')][-1]); print('$label %.4f ms/step  fwd_out in-step %.2f us  dgrad_out %.2f' % (j['ms_per_step'], 1e3*j['roofline']['kernels_in_step_ms']['fwd_out'], 1e3*j['roofline']['kernels_in_step_ms']['dgrad_out']))"
}
for i in 1 2 3; do ab two_launches libbp_hip_old.so; ab one_launch libbp_hip.so; done
echo "== window path (staging rides in the fused launch)"
for lib in libbp_hip_old.so libbp_hip.so libbp_hip_old.so libbp_hip.so; do echo -n "$lib: "; BP_HIP_LIB=$PWD/dnn-for-speech-enhancement_amd/$lib python tools/bench_windows.py 2>/dev/null | tail -1 | cut -c1-200; done
