for w in massive50 massive125 massive250; do for t in 0 64; do
echo -n "$w team=$t: "; python bench.py --workload $w --batch 16384 --team $t --cpu-seconds 0 --extras 0 --check 0 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().split('\n')[-1]); print(round(d['value']/1e6,3), d['config'].get('team_size'), d['config'].get('team_mode'))"
done; done
