set -u
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$(pwd)}
OUT=$R/gpurun_out/hop_serial; rm -rf $OUT; mkdir -p $OUT; cd $R
D=$OUT/t
rocprofv3 --kernel-trace --stats --output-format csv -d $D -o t -- python tools/hop_trace.py run 48000 8 64 overlap=25 > $D.log 2>&1
grep "wall us/hop" $D.log
f=$(find $D -name '*kernel_trace.csv' | head -1)
python tools/hop_timeline.py $f > $OUT/timeline.txt
rm -rf $D
tail -42 $OUT/timeline.txt
