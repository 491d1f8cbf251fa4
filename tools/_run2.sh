cmd="python bench.py --steps 40 --warmup 3 --no-cpu-baseline --no-solver --no-loop --no-variants --details ''"
run() { $cmd "$@" 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('$*', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))"; }
run --streams 1; run --streams 2; run --streams 3; run --streams 1 --batch 512; run --streams 2 --batch 128
