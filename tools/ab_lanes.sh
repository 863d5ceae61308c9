# A/B runs of the lanes-across-the-batch kernel's launch parameters (bench.py lines, no PMC): bash tools/ab_lanes.sh
B="--extras 0 --cpu-seconds 0 --pmc 0 --legs 0 --workload sketch150 --warmup 1"
run() { name=$1; shift
  env "$@" > gpurun_out/abl_$name.json 2> gpurun_out/abl_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/abl_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"]/1e6,3), "M/s kernel_ms", round(d["roofline"]["kernel_ms"],3), "ok", d["results_ok"], d.get("oracle_check"))
except Exception as e:
    print("$name", "FAILED", e)
PY
}
run s0 EZPZ_LANES_STRAGGLERS=0 python bench.py $B --batch 262144 --steps 5
run s2 EZPZ_LANES_STRAGGLERS=2 python bench.py $B --batch 262144 --steps 5
run s4 EZPZ_LANES_STRAGGLERS=4 python bench.py $B --batch 262144 --steps 5
run s8 EZPZ_LANES_STRAGGLERS=8 python bench.py $B --batch 262144 --steps 5
run s12 EZPZ_LANES_STRAGGLERS=12 python bench.py $B --batch 262144 --steps 5
run s32 EZPZ_LANES_STRAGGLERS=32 python bench.py $B --batch 262144 --steps 5
run s40 EZPZ_LANES_STRAGGLERS=40 python bench.py $B --batch 262144 --steps 5
run s16 EZPZ_LANES_STRAGGLERS=16 python bench.py $B --batch 262144 --steps 5
run s24 EZPZ_LANES_STRAGGLERS=24 python bench.py $B --batch 262144 --steps 5
run s4_b524k_r8 EZPZ_LANES_STRAGGLERS=4 EZPZ_LANES_REFILL=8 python bench.py $B --batch 524288 --steps 3
run s4_b65k EZPZ_LANES_STRAGGLERS=4 python bench.py $B --batch 65536 --steps 5
run s0_b65k EZPZ_LANES_STRAGGLERS=0 python bench.py $B --batch 65536 --steps 5
run w2048_r8 EZPZ_LANES_WAVES=2048 EZPZ_LANES_REFILL=8 python bench.py $B --batch 262144 --steps 5
run r22_b524k EZPZ_LANES_REFILL=22 python bench.py $B --batch 524288 --steps 3
