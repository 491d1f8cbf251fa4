timeout 1200 python -m pytest tests/test_group_gpu.py tests/test_comm_gpu.py tests/test_spa_sharded_gpu.py tests/test_spa_gpu.py -x -q -m gpu 2>&1 | tail -15
