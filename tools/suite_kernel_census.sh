cd /tmp && export TMPDIR=/tmp
cd $GRAFT_REPO_ROOT
timeout 1500 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/sp -o suite -- python -m pytest tests -q -m gpu -x 2>&1 | tail -3
mkdir -p gpurun_out/suite_prof
find /tmp/sp -name "*kernel_stats*" | while read f; do n=$(echo $f | tr '/' '_'); cut -d, -f1,2 "$f" > gpurun_out/suite_prof/$n; done
ls gpurun_out/suite_prof | wc -l
