# A/B runs of the class-specialised kernels' switches on the GPU box (bench.py lines, no PMC): usage bash tools/ab_fastdiv.sh
# (parity of every variant is checked by the bench line itself: results_ok + bitwise_equal against the oracle)
B="--extras 0 --cpu-seconds 0 --pmc 0 --steps 200 --warmup 30"
run() { # name, env..., args
  name=$1; shift
  env "$@" > gpurun_out/ab_$name.json 2> gpurun_out/ab_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/ab_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"]/1e6,2), "M/s kernel_ms", round(d["roofline"]["kernel_ms"],4), "ok", d["results_ok"], d.get("oracle_check",{}).get("bitwise_equal"))
except Exception as e:
    print("$name", "FAILED", e)
PY
}
run massive_fd3 EZPZ_JIT_FASTDIV=3 python bench.py $B
run massive_fd0 EZPZ_JIT_FASTDIV=0 python bench.py $B
run massive_fd3_mw2 EZPZ_JIT_FASTDIV=3 EZPZ_JIT_MINWAVES=2 python bench.py $B
run massive_fd3_w2 EZPZ_JIT_FASTDIV=3 EZPZ_JIT_WAVES=2 python bench.py $B
run massive_fd3_w8 EZPZ_JIT_FASTDIV=3 EZPZ_JIT_WAVES=8 python bench.py $B
run massive600_fd3 EZPZ_JIT_FASTDIV=3 python bench.py $B --workload massive600
run massive200_fd3 EZPZ_JIT_FASTDIV=3 python bench.py $B --workload massive200
run square_fd3 python bench.py $B --workload square --batch 65536
run mixed_fd3 python bench.py $B --workload mixed --batch 1048576 --steps 50
run massiveo_fd3 EZPZ_JIT_FASTDIV=3 python bench.py $B --workload massive500o
run massiveo_fd0 EZPZ_JIT_FASTDIV=0 python bench.py $B --workload massive500o
run ladder_fd3 EZPZ_JIT_FASTDIV=3 python bench.py $B --workload massive50000 --batch 64 --steps 50
run ladder_fd0 EZPZ_JIT_FASTDIV=0 python bench.py $B --workload massive50000 --batch 64 --steps 50
