#!/bin/bash
# round 5, GPU call 9: the driver's N>1 launch form on the one device (2 ranks sharing it), bptrain end to end, window path
O=gpurun_out/r05c9; mkdir -p $O
( time python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29517 bench.py --gpus 2 --steps 20 --warmup 5 > $O/bench2_torchrun.json 2> $O/bench2_torchrun.err ) 2>&1 | grep real
tail -c 700 $O/bench2_torchrun.json; echo; grep -v "amdgpu.ids" $O/bench2_torchrun.err | tail -3
timeout 600 python tools/bench_bptrain.py 4000 420 2>$O/bptrain.err | tail -2
python tools/bench_windows.py 2>/dev/null | tail -2
