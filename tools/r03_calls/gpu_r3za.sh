#!/bin/bash
export TMPDIR=/tmp
O=gpurun_out/r3za
mkdir -p $O
timeout 300 ./tools/exp/ws_probe > $O/ws_probe.txt 2>&1; grep -E "^mfma only|^VALU helper only|^both, VALU helper  |^stores only|^both \(4 store|^bf16" $O/ws_probe.txt | cut -c1-120
