python -m pytest tests -q -m gpu 2>&1 | tail -30
for b in 32768 65536; do echo "BATCH=$b lanes"; BATCH=$b python tools/sketch_scaling.py 50 75 150 250 2>&1 | grep -E "npts" | cut -c1-220; echo "BATCH=$b teams"; EZPZ_LANES=0 BATCH=$b python tools/sketch_scaling.py 50 75 150 250 2>&1 | grep -E "npts" | cut -c1-220; done
