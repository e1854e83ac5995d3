#!/bin/bash
# usage: tools/gpu_variants.sh <spec> name1 name2 ...   (on the GPU box: time each variants/<name>.so as librt_hip.so)
spec=$1; shift
cp raytracer_amd/csrc/librt_hip.so /tmp/librt_hip.orig.so
for v in "$@"; do
  cp variants/$v.so raytracer_amd/csrc/librt_hip.so
  timeout -k 10 120 python tools/gpu_time.py $v $spec 2>/dev/null || { echo "$v failed"; break; }
done
cp /tmp/librt_hip.orig.so raytracer_amd/csrc/librt_hip.so
