#!/bin/bash
# usage (on the GPU box): bash tools/refresh_profiles_r06.sh <outdir>   -- round 6's artefacts of profiles/ from one build (the
# rest of tools/refresh_profiles.sh -- one-call latencies, sketch scaling, fronts -- did not change this round and keeps its r05 files).
out=$1; mkdir -p $out
PY=$(python -c 'import os,sys;print(os.path.realpath(sys.executable))')
python bench.py --steps 20 --warmup 5 > $out/bench_massive.json 2>/dev/null   # the driver's invocation: headline + legs over every BASELINE config
EZPZ_JIT_AHEAD=0 python bench.py --steps 20 --warmup 5 --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_massive_loop_kernel.json 2>/dev/null   # round 5's kernel in the same build
python bench.py --workload massive600 --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_massive600.json 2>/dev/null
python bench.py --workload massive200 --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_massive200.json 2>/dev/null
python bench.py --workload massive500o --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_massive500_overconstrained.json 2>/dev/null
python bench.py --workload massive50000 --batch 64 --steps 20 --legs 0 --extras 0 > $out/bench_ladder200k.json 2>/dev/null
python bench.py --workload massive50000 --batch 256 --steps 20 --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_ladder200k_b256.json 2>/dev/null
EZPZ_JIT_AHEAD=0 python bench.py --workload massive50000 --batch 64 --steps 20 --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_ladder200k_loop_kernel.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_m -- $PY bench.py --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 > /dev/null 2>&1
find $out/stats_m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/massive_b65536_kernel_stats.csv; rm -rf $out/stats_m
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x -- $PY bench.py --workload massive50000 --batch 64 --steps 20 --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 > /dev/null 2>&1
find $out/stats_x -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/ladder200k_kernel_stats.csv; rm -rf $out/stats_x
# same-box A/B: round 5's kernels (EZPZ_JIT_AHEAD=0: every verdict waited for) against the default, alternating
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print("   %.4g solves/s, %.4f ms per launch, bitwise equal to the oracle: %s" % (d["value"], d["ms_per_step"], d.get("oracle_check",{}).get("bitwise_equal")))'
B="--legs 0 --pmc 0 --cpu-seconds 0 --extras 0"
(echo "# EZPZ_JIT_AHEAD=<0|1> python bench.py <workload> $B   (0 = the loop kernels of round 5: every verdict of the LM control waited for; 1 = the default)"
for i in 1 2; do for v in 0 1; do for w in "--steps 20 --warmup 5" "--workload massive200 --batch 65536" "--workload massive600 --batch 65536" "--workload massive50000 --batch 64 --steps 20" "--workload massive50000 --batch 256 --steps 20"; do
  echo "EZPZ_JIT_AHEAD=$v bench.py $w"; EZPZ_JIT_AHEAD=$v python bench.py $w $B 2>/dev/null | python -c "$P"; done; done; done) > $out/fast_ab.txt
(echo "# EZPZ_TICKETS=<0|1> python bench.py --steps 20 --warmup 5 $B   (0 = fixed shares of the batch per workgroup, 1 = the default: workgroups draw their systems)"
for i in 1 2; do for v in 0 1; do echo "EZPZ_TICKETS=$v"; EZPZ_TICKETS=$v python bench.py --steps 20 --warmup 5 $B 2>/dev/null | python -c "$P"; done; done) > $out/tickets_ab.txt
(python tools/resident_cost.py 2>&1 | grep -v amdgpu.ids) > $out/resident_cost.txt
(echo "# python tools/ladder_stamps.py 50000 280  (the kernel that does not wait for verdicts, compiled with its time stamps: 0 start, 1 both steps taken and stores issued, 2 the barrier, 3 partials published, 4 its turn at the totals of the system before; us)"; python tools/ladder_stamps.py 50000 280 2>&1 | grep -v "amdgpu.ids\|^  system") > $out/ladder_stamps.txt
hipcc --offload-arch=gfx950 -O3 -std=c++17 -o tools/row_copy_bench.bin tools/row_copy_bench.hip 2>/dev/null && (echo "# tools/row_copy_bench.bin  (what memory allows a kernel that streams 16 KB rows in and out at the solve kernels' occupancy: their access pattern against full lines)"; ./tools/row_copy_bench.bin 2>&1) > $out/row_copy_bench.txt
rm -f gpurun_out/parity_bar.txt gpurun_out/freedom_probes_fuzz.txt
python -m pytest tests -m gpu -q 2>&1 | tail -6 > $out/pytest_gpu.txt
cp gpurun_out/parity_bar.txt $out/parity_bar_raw.txt 2>/dev/null; cp gpurun_out/freedom_probes_fuzz.txt $out/freedom_probes_fuzz.txt 2>/dev/null
python tools/parity_bar_summary.py $out/parity_bar_raw.txt "python -m pytest tests -m gpu (round 6, the whole GPU suite)" > $out/parity_bar.txt 2>/dev/null && rm -f $out/parity_bar_raw.txt
python tools/reference_benches.py > $out/reference_benches.txt 2>/dev/null
# the ladder's two compilations by systems per launch (EZPZ_JIT_GRID_STAGE=0: the first one whatever the launch)
(echo "# EZPZ_JIT_GRID_STAGE=<0|1> python bench.py --workload massive50000 --batch <systems per launch> --steps 20 $B   (0 = the compilation at four wavefronts per SIMD whatever the launch: 10 systems in flight, values stored slot by slot; 1 = the default: beyond 60 systems per launch the compilation at three, 7 in flight, values staged in LDS and stored back to back)"
for i in 1 2; do for w in 32 48 60 64 80 128 256 1024; do for v in 0 1; do echo "EZPZ_JIT_GRID_STAGE=$v --batch $w"; EZPZ_JIT_GRID_STAGE=$v python bench.py --workload massive50000 --batch $w --steps 20 $B 2>/dev/null | python -c "$P"; done; done; done) > $out/ladder_variants.txt
(echo "# python tools/qr_routes.py  (FreedomAnalysis by the pivoted QR of 800 / 1400 / 2000 variables, ms per call by systems per call: chain of launches | EZPZ_FREEDOM_CHAIN=2 | default)"; python tools/qr_routes.py 2>&1 | grep -v amdgpu.ids) > $out/qr_routes_now.txt
# the stress tools: random block systems whose workgroups draw, LDS patterns, a co-running kernel
(echo "# python tools/stress_random_blocks.py 24; ... all 30; ... grid 10"; python tools/stress_random_blocks.py 24 2>&1 | grep -v amdgpu.ids; python tools/stress_random_blocks.py all 30 2>&1 | grep -v amdgpu.ids; python tools/stress_random_blocks.py grid 10 2>&1 | grep -v amdgpu.ids) > $out/stress_blocks_now.txt
hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/liblds_poison.so tools/lds_poison.hip 2>/dev/null && (echo "# python tools/stress_lds_poison.py"; python tools/stress_lds_poison.py 2>&1 | grep -v amdgpu.ids) > $out/stress_lds_poison.txt
hipcc --offload-arch=gfx950 -O2 -shared -fPIC -o tools/libnoise.so tools/noise.hip 2>/dev/null && (echo "# EZPZ_DEBUG=hip python tools/stress_noise.py"; EZPZ_DEBUG=hip timeout 1200 python tools/stress_noise.py 2>&1 | grep -v "amdgpu.ids\|cooperative launch attribute" | uniq -c) > $out/stress_noise.txt
(echo "# python tools/stress_threads.py 12 120"; python tools/stress_threads.py 12 120 2>&1 | grep -v amdgpu.ids) > $out/stress_threads.txt
head -3 $out/massive_b65536_kernel_stats.csv; tail -3 $out/pytest_gpu.txt
