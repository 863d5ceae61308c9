#!/bin/bash
# usage (on the GPU box): bash tools/refresh_profiles.sh <outdir>   -- every artefact profiles/ holds for a round, from one build
out=$1; mkdir -p $out
python bench.py > $out/bench_massive.json 2>/dev/null
python bench.py --workload square --batch 65536 > $out/bench_square.json 2>/dev/null
python bench.py --workload mixed --batch 1000000 --steps 20 > $out/bench_mixed_1M.json 2>/dev/null
python bench.py --workload massive50000 --batch 64 --steps 20 > $out/bench_ladder200k.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_m -- python3 bench.py --cpu-seconds 0 --extras 0 > /dev/null 2>&1
find $out/stats_m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/massive_b4096_kernel_stats.csv; rm -rf $out/stats_m
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s -- python3 bench.py --workload square --batch 65536 --cpu-seconds 0 --extras 0 > /dev/null 2>&1
find $out/stats_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/square_b65536_kernel_stats.csv; rm -rf $out/stats_s
bash tools/pmc.sh $out/pmc > /dev/null 2>&1
cp $out/pmc/summary.txt $out/massive_b4096_pmc_summary.txt; rm -rf $out/pmc
(echo "# python tools/sketch_scaling.py  (one connected sketch of mixed kinds, tests/gen.py:connected_sketch; default = batch-throughput launch shape)"; python tools/sketch_scaling.py 8 25 75 150 400 1000 2500 2>&1 | grep npts; echo "# TEAM=4294967295 (EZPZ_TEAM_AUTO_LATENCY: the launch shape ezpz_solve uses for one solve)"; TEAM=4294967295 python tools/sketch_scaling.py 25 75 150 400 2>&1 | grep npts) > $out/sketch_scaling.txt
head -3 $out/massive_b4096_kernel_stats.csv; cat $out/massive_b4096_pmc_summary.txt
