#!/bin/bash
# Files what tools/round_end.sh <name> left under gpurun_out/ into profiles/ (tracked) under a prefix, and the hash-keyed tables bench.py replays
# (traffic*.json, valu_mix.json, clock.json) in place.   usage: tools/file_round_artifacts.sh <name> <prefix>      e.g. r05b r05b
N=${1:?name}; P=${2:?prefix}
R=$(git -C "$(dirname "$0")" rev-parse --show-toplevel); G=$R/gpurun_out; O=$R/profiles; cd $R
cp $G/$N/traffic.json $O/traffic.json; cp $G/${N}hd/traffic_hd1080.json $O/traffic_hd1080.json; cp $G/${N}hd/valu_mix.json $O/valu_mix.json 2>/dev/null || cp $G/$N/valu_mix.json $O/valu_mix.json
cp $G/${N}clk/clock.json $O/clock.json
c() { [ -s "$1" ] && cp "$1" "$2"; }
c $G/$N/bench_final.json $O/${P}_bench.json; c $G/$N/bench_final.stdout $O/${P}_bench_stdout_tail.txt && tail -n 1 $G/$N/bench_final.stdout > $O/${P}_bench_line.json && rm -f $O/${P}_bench_stdout_tail.txt
c $G/$N/bench.json $O/${P}_bench_second_collection.json
for k in extract_only noise one_lane region_timing match100k_popcount match100k_int8 two_ranks_one_gpu_gloo; do c $G/$N/bench_$k.json $O/${P}_bench_$k.json; done
c $G/$N/bench_torchrun_two_ranks.json $O/${P}_bench_torchrun_two_ranks_one_gpu_gloo.json
c $G/${N}hd/bench_hd.json $O/${P}_bench_hd1080.json
c $G/$N/stats_kernel_stats.csv $O/${P}_kernel_stats.csv; c $G/$N/stats_overlap_kernel_stats.csv $O/${P}_kernel_stats_overlap.csv; c $G/$N/stats_match_kernel_stats.csv $O/${P}_kernel_stats_match100k.csv
c $G/$N/pmc_fetch_write.txt $O/${P}_pmc_fetch_write.txt; c $G/$N/pmc_sq_counters.txt $O/${P}_pmc_sq_counters.txt
c $G/${N}clk/pmc_clock.txt $O/${P}_pmc_clock.txt; c $G/${N}ta/pmc_ta.txt $O/${P}_pmc_ta.txt; c $G/${N}mfma/pmc_mfma.txt $O/${P}_pmc_mfma.txt
for k in corun_probe kf_search frontend_w15 frontend_w100 frontend_w15_warp orbmatcher_dropin fuzz_batch_400 fuzz_parity_4000 fuzz_frontend_3000 fuzz_match_20000 fuzz_orbmatcher_10000 fuzz_orbmatcher_real_access_10000; do c $G/$N/$k.json $O/${P}_$k.json; done
c $G/$N/cpp_example_lanes.txt $O/${P}_cpp_example_lanes.txt; c $G/$N/single_frame.txt $O/${P}_single_frame_latency.txt; c $G/$N/pytest_gpu.txt $O/${P}_pytest_gpu.txt
# keep the kernel-stats tables short: the library's kernels only
for f in $O/${P}_kernel_stats.csv $O/${P}_kernel_stats_overlap.csv $O/${P}_kernel_stats_match100k.csv; do [ -s $f ] && { head -1 $f; grep 'orbx::\|rocclr' $f; } > $f.tmp && mv $f.tmp $f; done
ls $O | grep "^${P}_" | wc -l
