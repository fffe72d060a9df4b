#!/usr/bin/env bash
# debug: time the K2 conv with each ablation variant of the library (built into _variants/)
cp boosting_nerv_amd/libbnerv_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp _variants/lib_$v.so boosting_nerv_amd/libbnerv_hip.so
  echo "== $v"; python tools/kbench.py 30 2>/dev/null | grep -E "K2\)|K3\)|K1\)" | head -3
done
cp /tmp/lib_orig.so boosting_nerv_amd/libbnerv_hip.so
