#!/bin/bash
cd $GRAFT_REPO_ROOT
timeout 1200 python -m pytest tests/test_matcher_gpu.py tests/test_baseline_shapes_gpu.py::test_config2_loop_batch_256_pairs tests/test_baseline_shapes_gpu.py::test_golden_vectors_on_the_hip_path -m gpu -x -q 2>&1 | grep -v "Registering\|amdgpu" | tail -4
python tools/seq_latency.py 20 resident
timeout 300 python tools/replay.py --scans 3000 2>/dev/null | cut -c1-200
timeout 300 python tools/prof_legs.py loop 2>/dev/null | cut -c1-330
