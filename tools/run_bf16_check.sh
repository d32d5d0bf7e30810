#!/bin/bash
# bf16 parity tests + A/B of the LDS-DMA staged 4096-wide GEMMs (BP_BF16_GEMM_NO_DMA = the register-staged loop) + the probe
O=gpurun_out/bf2; mkdir -p $O
timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_autograd.py tests/test_dp_native.py tests/test_bpforward.py tests/test_bptrain.py -m gpu -x -q -k "bf16 or config5 or compute_dtype" > $O/pytest.log 2>&1; echo "pytest rc=$?"; tail -3 $O/pytest.log
for v in dma nodma dma nodma; do
  if [ $v = nodma ]; then export BP_BF16_GEMM_NO_DMA=1; else unset BP_BF16_GEMM_NO_DMA; fi
  echo $v; python tools/bench_bf16.py c5bf16 2>$O/err.txt | tail -1
done
[ -x tools/bin/bf16_gemm_probe ] && timeout 250 tools/bin/bf16_gemm_probe > gpurun_out/bf16_gemm_probe.txt 2>&1
