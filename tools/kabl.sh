#!/usr/bin/env bash
# debug: run kbench with several builds of the library (from _variants/); usage: tools/kabl.sh <grep-pattern> v1 v2 ...
pat="$1"; shift
cp boosting_nerv_amd/libbnerv_hip.so /tmp/lib_orig.so
for v in "$@"; do
  cp _variants/lib_$v.so boosting_nerv_amd/libbnerv_hip.so
  echo "== $v"; python tools/kbench.py 30 2>/dev/null | grep -E "$pat"
done
cp /tmp/lib_orig.so boosting_nerv_amd/libbnerv_hip.so
