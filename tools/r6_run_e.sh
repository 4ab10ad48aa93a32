#!/bin/bash
for rep in 1 2; do
for cfg in offline2 offline48_2; do
  a=$(timeout 300 python bench.py --side-config $cfg 2>/dev/null | grep "^SIDE" | cut -c1-160)
  b=$(DPDFNET_HIP_LIB=$PWD/build_ab/lib_c8_16.so timeout 300 python bench.py --side-config $cfg 2>/dev/null | grep "^SIDE" | cut -c1-160)
  echo "$cfg default: $a"; echo "$cfg cluster8 at 16 tiles: $b"
done; done
