# A/B sweep of the class-specialised kernel's launch shape (wavefronts per system x occupancy hint) against the automatic choice, on
# the GPU box: bash tools/sweep_jit_shapes.sh   (round 3: the automatic choice is the best or within 2 % at 800, 2400 vars; 4000: T=4 +8 %)
B="--extras 0 --cpu-seconds 0 --pmc 0 --legs 0 --check 0 --steps 200 --warmup 30"
run() { name=$1; shift
  env "$@" > gpurun_out/sw_$name.json 2> gpurun_out/sw_$name.err
  python - <<PY
import json
try:
    d=json.loads(open("gpurun_out/sw_$name.json").read().strip().splitlines()[-1])
    print("$name", round(d["value"]/1e6,2), "M/s kernel_ms", round(d["roofline"]["kernel_ms"],4), "ok", d["results_ok"])
except Exception as e:
    print("$name", "FAILED", e)
PY
}
for wl in massive600 massive200 massive1000; do
run ${wl}_default python bench.py $B --workload $wl
for w in 2 4 8; do for mw in 2 3 4; do
run ${wl}_w${w}_mw${mw} EZPZ_JIT_WAVES=$w EZPZ_JIT_MINWAVES=$mw python bench.py $B --workload $wl
done; done; done
