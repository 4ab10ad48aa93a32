set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hops_now; rm -rf $OUT; mkdir -p $OUT; cd $R
for cfg in "48000 8 64 hop_48k_nb8_64streams" "16000 2 1 hop_16k_nb2_1stream"; do
    set -- $cfg
    D=$OUT/$4
    rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/hop_trace.py run $1 $2 $3 > $D.log 2>&1
    grep "wall us/hop" $D.log
    f=$(find $D -name '*kernel_trace.csv' | head -1)
    python tools/hop_timeline.py $f > $OUT/$4_timeline.txt
    rm -rf $D
done
