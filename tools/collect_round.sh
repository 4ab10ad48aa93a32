# One lease, the round's final build: GPU suite -> bench line -> rocprofv3 traces + PMC -> side-configuration traces -> latency / hop / soak tools.
# usage (on the GPU box): bash tools/collect_round.sh r6        then, back home: bash tools/bank_profiles.sh r6
set -x
T=${1:-r6}
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/${T}_gpu_tests.log 2>&1; tail -3 gpurun_out/${T}_gpu_tests.log
python bench.py > gpurun_out/${T}_bench_default.json 2> gpurun_out/${T}_bench_default.err
bash tools/profile_round.sh $T > gpurun_out/${T}_profile_round.log 2>&1; tail -12 gpurun_out/${T}_profile_round.log
bash tools/profile_extra.sh $T > gpurun_out/${T}_profile_extra.log 2>&1; tail -12 gpurun_out/${T}_profile_extra.log
python tools/latency_bench.py > gpurun_out/${T}_latency.txt 2>&1
python tools/hop_ab2.py - > gpurun_out/${T}_hop_ab.txt 2>&1
(python tools/stress.py 120 7; python tools/stream_soak.py 120 11; python tools/host_pipe_soak.py 120 13) > gpurun_out/${T}_soak.txt 2>&1
tail -3 gpurun_out/${T}_soak.txt
timeout 120 tools/pk_fma_coissue_probe 3 > gpurun_out/${T}_pk_fma_coissue_probe.txt 2>&1; head -8 gpurun_out/${T}_pk_fma_coissue_probe.txt | grep -v "^    thread"
bash tools/suite_kernel_census.sh > gpurun_out/${T}_suite_kernel_census.log 2>&1; tail -2 gpurun_out/${T}_suite_kernel_census.log
