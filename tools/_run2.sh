timeout 900 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py tests/test_raster_corner_cases_gpu.py -x -q -m gpu 2>&1 | tail -2
cmd="python bench.py --steps 30 --warmup 3 --no-cpu-baseline --no-solver --no-loop --no-variants --details ''"
for i in 1 2 3; do $cmd 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); print('default', round(d['value']), round(d['ms_per_step'],3), round(d['roofline']['avg_launch_ms'],4))"; done
