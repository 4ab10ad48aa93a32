set -x
cd $GRAFT_REPO_ROOT
python -m pytest tests -q -m gpu > gpurun_out/r5_gpu_tests.log 2>&1; tail -3 gpurun_out/r5_gpu_tests.log
python bench.py > gpurun_out/r5_bench_default.json 2> gpurun_out/r5_bench_default.err
bash tools/profile_round.sh r5 > gpurun_out/r5_profile_round.log 2>&1; tail -12 gpurun_out/r5_profile_round.log
bash tools/profile_extra.sh r5 > gpurun_out/r5_profile_extra.log 2>&1; tail -12 gpurun_out/r5_profile_extra.log
python tools/latency_bench.py > gpurun_out/r5_latency.txt 2>&1
python tools/hop_ab2.py - > gpurun_out/r5_hop_ab.txt 2>&1
(python tools/stress.py 120 7; python tools/stream_soak.py 120 11; python tools/host_pipe_soak.py 120 13) > gpurun_out/r5_soak.txt 2>&1
tail -3 gpurun_out/r5_soak.txt
