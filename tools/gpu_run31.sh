#!/bin/bash
cd $GRAFT_REPO_ROOT
for v in 4096 100000; do
echo "lane_from $v"; KH_FIND_VALID_LANE_FROM=$v timeout 300 python tools/prof_legs.py loop 2>/dev/null | python -c "import sys,json; b=json.loads(sys.stdin.read()); print({k:b[k] for k in ('loop_batch_ms','loop_gpu_ms')})"
done
KH_FIND_VALID_LANE_FROM=100000 timeout 600 python -m pytest tests/test_baseline_shapes_gpu.py::test_config2_loop_batch_256_pairs -m gpu -x -q 2>&1 | tail -2
