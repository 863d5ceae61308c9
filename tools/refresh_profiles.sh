#!/bin/bash
# usage (on the GPU box): bash tools/refresh_profiles.sh <outdir>   -- every artefact profiles/ holds for a round, from one build.
# Bench lines carry their own PMC counters (bench.py runs rocprofv3 --pmc child passes); the kernel-trace stats of the
# same command are filed beside them so that the kernel's average duration can be checked against roofline.kernel_ms.
out=$1; mkdir -p $out
# (the interpreter itself after `--`: a python3 shim that exec()s the real one would be an exec hop under the profiler)
PY=$(python -c 'import os,sys;print(os.path.realpath(sys.executable))')
python bench.py --steps 20 --warmup 5 > $out/bench_massive.json 2>/dev/null   # the driver's invocation: headline + legs over every BASELINE config
EZPZ_JIT=0 python bench.py --specialize 0 --legs 0 > $out/bench_massive_interpreter.json 2>/dev/null   # (EZPZ_JIT=0: with the on-disk cache a kernel compiled earlier would be picked up)
EZPZ_JIT=0 EZPZ_COMP=0 python bench.py --specialize 0 --legs 0 > $out/bench_massive_listwalk.json 2>/dev/null
EZPZ_JIT_FASTDIV=0 python bench.py --legs 0 --cpu-seconds 0 --extras 0 > $out/bench_massive_plain_divisions.json 2>/dev/null
python bench.py --batch 4096 --pmc 0 --cpu-seconds 0 --extras 0 --legs 0 > $out/bench_massive_b4096.json 2>/dev/null
python bench.py --batch 16384 --pmc 0 --cpu-seconds 0 --extras 0 --legs 0 > $out/bench_massive_b16384.json 2>/dev/null   # (rounds 2-3 quoted this batch)
python bench.py --workload massive600 --legs 0 > $out/bench_massive600.json 2>/dev/null   # (round 5: with the PMC roofs)
python bench.py --workload massive200 > $out/bench_massive200.json 2>/dev/null
python bench.py --workload massive500o --legs 0 > $out/bench_massive500_overconstrained.json 2>/dev/null
python bench.py --workload square --batch 65536 > $out/bench_square.json 2>/dev/null
python bench.py --workload mixed --batch 1000000 --steps 20 > $out/bench_mixed_1M.json 2>/dev/null
python bench.py --workload massive50000 --batch 64 --steps 20 > $out/bench_ladder200k.json 2>/dev/null
python bench.py --workload sketch150 --batch 262144 --steps 10 --warmup 2 > $out/bench_sketch_300vars_b262144.json 2>/dev/null
python bench.py --workload sketch150 --batch 32768 --steps 10 --warmup 2 > $out/bench_sketch_300vars_b32768.json 2>/dev/null   # (below the lanes' batch: the teams' record walk)
# round 5: ONE solve of a large connected sketch (2000 / 5000 variables) on the frontal shape (0xFFFFFFFF = the automatic latency shape),
# with the PMC roofs of the CUs it runs on, and the same solve on round 4's record walk (0xFFFFFFF9)
python bench.py --workload sketch1000 --batch 1 --team 0xFFFFFFFF --max-iterations 60 --legs 0 --steps 50 --warmup 5 --cpu-seconds 6 > $out/bench_sketch2000_one_solve.json 2>/dev/null
python bench.py --workload sketch2500 --batch 1 --team 0xFFFFFFFF --max-iterations 60 --legs 0 --steps 20 --warmup 3 --cpu-seconds 6 > $out/bench_sketch5000_one_solve.json 2>/dev/null
python bench.py --workload sketch1000 --batch 1 --team 0xFFFFFFF9 --max-iterations 60 --legs 0 --steps 20 --warmup 3 --cpu-seconds 0 > $out/bench_sketch2000_one_solve_records.json 2>/dev/null
python bench.py --workload sketch2500 --batch 1 --team 0xFFFFFFF9 --max-iterations 60 --legs 0 --steps 5 --warmup 2 --cpu-seconds 0 > $out/bench_sketch5000_one_solve_records.json 2>/dev/null
cd /tmp && export TMPDIR=/tmp; cd $GRAFT_REPO_ROOT
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_m -- $PY bench.py --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 > /dev/null 2>&1
find $out/stats_m -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/massive_b65536_kernel_stats.csv; rm -rf $out/stats_m
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_s -- $PY bench.py --workload square --batch 65536 --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 > /dev/null 2>&1
find $out/stats_s -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/square_b65536_kernel_stats.csv; rm -rf $out/stats_s
for w in "mixed 1048576 mixed_1M" "massive50000 64 ladder200k" "sketch150 262144 sketch_300vars_b262144" "sketch150 32768 sketch_300vars_b32768"; do set -- $w
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x -- $PY bench.py --workload $1 --batch $2 --steps 10 --warmup 2 --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 > /dev/null 2>&1
find $out/stats_x -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/$3_kernel_stats.csv; rm -rf $out/stats_x; done
for w in "sketch1000 sketch2000_one_solve" "sketch2500 sketch5000_one_solve"; do set -- $w
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats_x -- $PY bench.py --workload $1 --batch 1 --team 0xFFFFFFFF --max-iterations 60 --steps 20 --warmup 3 --cpu-seconds 0 --extras 0 --pmc 0 --legs 0 --check 0 > /dev/null 2>&1
find $out/stats_x -name "*kernel_stats.csv" | head -1 | xargs -I{} cp {} $out/$2_kernel_stats.csv; rm -rf $out/stats_x; done
# (twice: the first process of a box compiles the small systems' kernels in the background while it times them; the second finds them in the on-disk cache)
python tools/reference_benches.py > $out/reference_benches_first_process.txt 2>/dev/null
python tools/reference_benches.py > $out/reference_benches.txt 2>/dev/null
# one solve() call, stage by stage, and the kernels' durations from a kernel trace of the same systems
python tools/solve_call_breakdown.py > $out/solve_call_breakdown.txt 2>/dev/null
rocprofv3 --kernel-trace -d $out/scb -o scb --output-format csv -- $PY tools/solve_call_breakdown.py --launch-only > /dev/null 2>&1
find $out/scb -name "*kernel_trace.csv" | head -1 | xargs -I{} python tools/solve_call_breakdown.py --from-trace {} >> $out/solve_call_breakdown.txt; rm -rf $out/scb
(echo "# python tools/sketch_scaling.py  (one connected sketch of mixed kinds, tests/gen.py:connected_sketch; default = the automatic batch shape: the record walk on 64 / 128 / 512 lanes per system while the state fits the LDS -- team_mode 4 -- and lanes across the batch from 65 536 systems per call)"; python tools/sketch_scaling.py 8 16 25 32 50 75 100 150 250 400 1000 2500 2>&1 | grep npts
echo "# TEAM=4294967294 (EZPZ_TEAM_AUTO_LISTS: the list-walk shapes batches ran on before the record walk -- one wavefront, a lean 128-lane workgroup, dense phases on top)"; EZPZ_LANES=0 TEAM=4294967294 python tools/sketch_scaling.py 32 50 75 100 150 250 400 1000 2500 2>&1 | grep npts
echo "# EZPZ_LANES=0 BATCH=32768 (the automatic batch shape on the per-system teams alone)"; EZPZ_LANES=0 BATCH=32768 python tools/sketch_scaling.py 32 50 75 100 150 250 400 2>&1 | grep npts
echo "# TEAM=4294967295 (EZPZ_TEAM_AUTO_LATENCY: the launch shape ezpz_solve uses for one solve: from 48 variables the frontal shape, team_mode 5 -- a tree of dense fronts on one or several workgroups; BATCH=256: its batch column is not what the shape is for)"; BATCH=256 TEAM=4294967295 python tools/sketch_scaling.py 10 16 25 32 75 150 250 400 1000 2500 5000 2>&1 | grep npts
echo "# TEAM=4294967289 (EZPZ_TEAM_LATENCY_RECORDS: round 4's shape for one solve: the record walk on 256-512 lanes)"; TEAM=4294967289 python tools/sketch_scaling.py 10 16 25 32 75 150 250 400 1000 2500 2>&1 | grep npts
echo "# TEAM=4294967292 (EZPZ_TEAM_LATENCY_PHASES: one solve's shape before the record walk: level lists and dense phases)"; TEAM=4294967292 python tools/sketch_scaling.py 16 25 32 75 150 250 400 2>&1 | grep npts
echo "# BATCH=262144 (a device-filling batch: one lane per system, batch_kernel.hip.hpp)"; BATCH=262144 python tools/sketch_scaling.py 25 75 150 2>&1 | grep npts
echo "# BATCH=65536 / 32768 (larger sketches on the lanes: 500, 800 and 2000 variables)"; BATCH=65536 python tools/sketch_scaling.py 250 400 2>&1 | grep npts; BATCH=32768 python tools/sketch_scaling.py 1000 2>&1 | grep npts
echo "# EZPZ_LANES=0 BATCH=262144 (the per-system teams on the same batch)"; EZPZ_LANES=0 BATCH=262144 python tools/sketch_scaling.py 25 75 150 2>&1 | grep npts) > $out/sketch_scaling.txt
[ -f ezpz_amd/libezpz_amd_stamps.so ] && (echo "# EZPZ_AMD_LIB=ezpz_amd/libezpz_amd_stamps.so python tools/rec_rounds.py 150  (a -DEZPZ_STAMPS -DEZPZ_REC_TIMES build: cycles of every round of the record walk on wavefront 0, one solve of 300 variables on 512 lanes, second LM iteration; every stamp costs ~150 cycles itself)"; EZPZ_AMD_LIB=$PWD/ezpz_amd/libezpz_amd_stamps.so python tools/rec_rounds.py 150 2>&1 | grep -E "^#|round") > $out/rec_rounds.txt
[ -f ezpz_amd/libezpz_amd_stamps.so ] && (echo "# EZPZ_AMD_LIB=ezpz_amd/libezpz_amd_stamps.so python tools/front_stamps.py <points> 0 2  (cycle stamps of workgroup 0, thread 0 over the third LM iteration of one solve on the frontal shape: 300 variables on one workgroup, 2000 on 14; every stamp costs ~150 cycles itself)"; for p in 150 1000; do python tools/front_stamps.py $p 0 2 2>&1 | grep -v "^rounds\|^level "; done) > $out/front_stamps.txt
(echo "# python tools/lanes_rounds.py 150 262144  (the jittered sketch150 batch with max_iterations capped: cost of each round of LM iterations on the lanes-across-the-batch kernel)"; python tools/lanes_rounds.py 150 262144 2>&1 | grep cap) > $out/lanes_rounds.txt
(echo "# EZPZ_LANES_STRAGGLERS=0 python tools/lanes_rounds.py 150 262144  (no hand-over of stragglers to the teams)"; EZPZ_LANES_STRAGGLERS=0 python tools/lanes_rounds.py 150 262144 2>&1 | grep cap) >> $out/lanes_rounds.txt
(echo "# python tools/pcie_bw.py  (host link of the GPU box)"; python tools/pcie_bw.py 2>&1) > $out/pcie_bw.txt
python tools/ab_microbench.py $out > /dev/null 2>&1
(echo "# ezpz_amd/ezpz-amd --filepath tests/golden/test_cases/<case>/problem.md  (the reference CLI's protocol, main.rs:86-100; steady state = the 100-run loop alone)"; for c in tiny square arc_radius two_rectangles massive_parallel_system; do echo "## $c"; ./ezpz_amd/ezpz-amd --filepath tests/golden/test_cases/$c/problem.md | grep -E "Problem size|Iterations|Solved in|i.e.|Steady"; done) > $out/cli_latency.txt
(echo "# python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_proptests.py -m gpu  (parity fuzz of the connected-sketch shapes and the reference's property tests on the HIP path: sensitivity-aware bar, no exclusions)"; python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_proptests.py -q -m gpu 2>&1 | tail -3) > $out/fuzz_tests.txt
# round 4: the floors the one-call path and the host-to-host pipeline were designed against, and what they reach
hipcc --offload-arch=gfx950 -O2 -o tools/launch_floor.bin tools/launch_floor.hip 2>/dev/null && (echo "# tools/launch_floor.bin  (host -> device -> host round trips by completion method)"; ./tools/launch_floor.bin 2>&1) > $out/launch_floor.txt
hipcc --offload-arch=gfx950 -O2 -o tools/pcie_duplex.bin tools/pcie_duplex.hip 2>/dev/null && (echo "# tools/pcie_duplex.bin  (both directions of the host link at once: queues and piece sizes)"; ./tools/pcie_duplex.bin 2>&1) > $out/pcie_duplex.txt
(echo "# python tools/h2h_rate.py [lines batch]  (ezpz_system_solve_batch between host buffers: pageable, then registered = the pipelined path)"; python tools/h2h_rate.py 2>&1 | tail -1; python tools/h2h_rate.py 600 16384 2>&1 | tail -1; python tools/h2h_rate.py 200 65536 2>&1 | tail -1) > $out/h2h_rate.txt
(echo "# python tools/freedom_wide.py 150 400 850 1000  (FreedomAnalysis of one large connected component: null-space probes on the frontal factorisation -- round 5, no QR --, then EZPZ_FREEDOM_PROBES=0: the pivoted QR with the matrix resident in registers, then on top EZPZ_FREEDOM_CHAIN=2: one cooperative launch streaming the trailing matrix, then EZPZ_FREEDOM_CHAIN=1: round 3's chain of a launch pair per Householder step)"; python tools/freedom_wide.py 150 400 850 1000 2500 2>&1 | grep variables; EZPZ_FREEDOM_PROBES=0 python tools/freedom_wide.py 150 400 850 1000 2>&1 | grep variables; EZPZ_FREEDOM_PROBES=0 EZPZ_FREEDOM_CHAIN=2 python tools/freedom_wide.py 150 400 850 1000 2>&1 | grep variables; EZPZ_FREEDOM_PROBES=0 EZPZ_FREEDOM_CHAIN=1 python tools/freedom_wide.py 150 400 850 1000 2>&1 | grep variables) > $out/freedom_wide.txt
# round 5: who solves which system -- the headline with drawn systems against fixed shares (same build), small calls of batch systems, and
# what a resident one-call kernel costs a batch on another stream (with the residency switched off as the control)
(echo "# EZPZ_TICKETS=<0|1> python bench.py --steps 20 --warmup 5 --legs 0 --cpu-seconds 0 --extras 0 --pmc 0  (0 = fixed shares of the batch per workgroup, 1 = the default: workgroups draw their systems from eight counters)"; for t in 0 1 0 1; do EZPZ_TICKETS=$t python bench.py --steps 20 --warmup 5 --legs 0 --cpu-seconds 0 --extras 0 --pmc 0 2>/dev/null | $PY -c "import json,sys; l=json.loads(sys.stdin.read().strip().splitlines()[-1]); print('EZPZ_TICKETS=$t:', round(l['value']/1e6,2), 'M solves/s,', round(l['ms_per_step'],4), 'ms per launch of', l['config']['systems_per_launch_per_gpu'], 'systems; oracle check', l['oracle_check']['bitwise_equal'])"; done) > $out/tickets_ab.txt
(echo "# python tools/small_calls.py; BATCHES=1,8,24,42,54,55,64 python tools/small_calls.py 400 1000 2500"; python tools/small_calls.py 2>&1 | grep -v amdgpu.ids; BATCHES=1,8,24,42,54,55,64 python tools/small_calls.py 400 1000 2500 2>&1 | grep -v amdgpu.ids) > $out/small_calls.txt
(echo "# python tools/graph_families.py  (the graph families and seeds of tests/test_gpu_fuzz.py, one solve: the automatic latency shape against the record walk)"; python tools/graph_families.py 2>&1 | grep -v amdgpu.ids) > $out/graph_families.txt
(python tools/resident_cost.py 2>&1 | grep -v amdgpu.ids) > $out/resident_cost.txt
(EZPZ_RESIDENT_US=0 python tools/resident_cost.py 2>&1 | grep -v amdgpu.ids) > $out/resident_cost_no_residency.txt
(EZPZ_TICKETS=0 python tools/resident_cost.py 2>&1 | grep -v amdgpu.ids) > $out/resident_cost_fixed_shares.txt
head -3 $out/massive_b65536_kernel_stats.csv
