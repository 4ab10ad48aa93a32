#!/bin/bash
set -u
O=gpurun_out/hazard; mkdir -p $O
run() { name=$1; shift; echo "== $name: $*"; ( time timeout 1200 "$@" ) > $O/$name.txt 2>&1; cut -c1-330 $O/$name.txt | tail -${TAILN:-30}; }
TAILN=40 run 81_coissue_forms  tools/pk_fma_coissue_probe 3
# the failing combination of round 5 (plain merged tap loads) with and without packed fp32 math in the library: limb vs fp32 kernels, 256 clips x 8 chunks
DPDFNET_HIP_LIB=$PWD/build_ab/lib_pk_plain.so   run 82_limb_check_pk_plain    python tools/limb_check4.py 20
DPDFNET_HIP_LIB=$PWD/build_ab/lib_nopk_plain.so run 83_limb_check_nopk_plain  python tools/limb_check4.py 60
for cfg in "16000 4" "48000 2" "16000 2" "48000 8"; do
  TAILN=6 run 84_ab_nopk_$(echo $cfg | tr ' ' _) python tools/lib_ab.py build_ab/lib_nopk.so $cfg
done
