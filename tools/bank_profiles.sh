#!/bin/bash
# Copy the evidence collected on the GPU box (gpurun_out/prof_<tag>*, tools/profile_round.sh + tools/profile_extra.sh + a default
# bench.py run) into profiles/ under the round's names.  usage: tools/bank_profiles.sh r3
set -e
T=${1:-r3}; P=gpurun_out/prof_$T; X=gpurun_out/prof_${T}_extra
cp $P/trace_pipelined/bench_kernel_stats.csv profiles/${T}_pipelined_kernel_stats.csv
cp $P/trace_serial/bench_kernel_stats.csv profiles/${T}_serial_kernel_stats.csv
cp $P/bench_line_under_trace_pipelined.json profiles/${T}_bench_line_under_trace_pipelined.json
cp $P/bench_line_under_trace_serial.json profiles/${T}_bench_line_under_trace_serial.json
for m in pipelined serial; do
    [ -f $P/trace_fp32mfma_$m/bench_kernel_stats.csv ] && cp $P/trace_fp32mfma_$m/bench_kernel_stats.csv profiles/${T}_fp32mfma_${m}_kernel_stats.csv
    [ -f $P/bench_line_under_trace_fp32mfma_$m.json ] && cp $P/bench_line_under_trace_fp32mfma_$m.json profiles/${T}_bench_line_under_trace_fp32mfma_$m.json
done
cp $P/pmc_summary.json profiles/${T}_pmc_summary.json; cp $P/pmc_summary.json profiles/pmc_summary.json
cp $P/build_manifest.json profiles/build_manifest.json
cp gpurun_out/${T}_bench_default.json profiles/${T}_bench_line_default_run.json
cp $X/hop_48k_nb8_64streams_kernel_stats.csv profiles/${T}_stream_hop_kernel_stats.csv
cp $X/hop_48k_nb8_64streams_timeline.txt profiles/${T}_stream_hop_timeline.txt
cp $X/hop_16k_nb2_1stream_kernel_stats.csv profiles/${T}_stream_hop_16k_1stream_kernel_stats.csv
cp $X/hop_16k_nb2_1stream_timeline.txt profiles/${T}_stream_hop_16k_1stream_timeline.txt
for c in 16k_nb2 16k_nb8 48k_nb2 48k_nb8; do cp $X/offline_${c}_256x10s_kernel_stats.csv profiles/${T}_offline_${c}_256x10s_kernel_stats.csv; done
cp gpurun_out/${T}_latency.txt profiles/${T}_latency_bench.txt; cp gpurun_out/${T}_hop_ab.txt profiles/${T}_hop_same_process.txt
grep "wall us/hop\|ms/step" $X/*.log
[ -f gpurun_out/${T}_pk_fma_coissue_probe.txt ] && grep -v "^    thread" gpurun_out/${T}_pk_fma_coissue_probe.txt > profiles/${T}_pk_fma_coissue_probe_final_build.txt
tail -4 gpurun_out/${T}_gpu_tests.log > profiles/${T}_gpu_tests_tail.txt
cp gpurun_out/${T}_soak.txt profiles/${T}_soak.txt 2>/dev/null || true
[ -f gpurun_out/suite_kernel_census.csv ] && cp gpurun_out/suite_kernel_census.csv profiles/${T}_suite_kernel_census.csv
