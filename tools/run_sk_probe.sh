#!/bin/bash
O=gpurun_out/sk4; mkdir -p $O
for v in tr trp4 trp1 trp16; do
timeout 120 tools/bin/wgrad_sk_probe_$v.bin c2 > $O/${v}_c2.txt 2>&1; echo $v; grep -E "TRACE span|TRACE within|TIMING|RESULT" $O/${v}_c2.txt
done
