#!/bin/bash
# Same-box A/B of the run-time compiled kernels against an earlier commit of this repository: box-to-box differences (+-4 %)
# are larger than most kernel edits, so the two libraries run on ONE box, alternating.
#   here (no GPU):   bash tools/ab_previous.sh build <commit>     -- checks the commit out into tools/_prev and builds it
#   on the GPU box:  bash tools/ab_previous.sh run                -- headline, ladder, square: previous / current, twice
set -e
if [ "$1" = build ]; then
    rm -rf tools/_prev && mkdir -p tools/_prev && git archive "$2" | tar -x -C tools/_prev
    (cd tools/_prev && python -c "import ezpz_amd.build as b; b.build(verbose=False)")
    exit 0
fi
P='import sys,json; d=json.loads(sys.stdin.read().strip().split(chr(10))[-1]); print(d["value"], d["ms_per_step"])'
B="--legs 0 --pmc 0 --cpu-seconds 0 --extras 0"
for i in 1 2; do
    for w in "--steps 20 --warmup 5" "--workload massive50000 --batch 64 --steps 20" "--workload square --batch 65536"; do
        echo "previous: bench.py $w"; (cd tools/_prev && EZPZ_JIT_CACHE=0 python bench.py $w $B 2>/dev/null | python -c "$P")
        echo "current:  bench.py $w"; EZPZ_JIT_CACHE=0 python bench.py $w $B 2>/dev/null | python -c "$P"
    done
done
