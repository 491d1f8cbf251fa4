timeout 900 python -m pytest tests/test_spa_gpu.py tests/test_posegraph_gpu.py -x -q -m gpu 2>&1 | tail -2
timeout 600 python -m pytest tests/test_baseline_shapes_gpu.py -x -q -m gpu -k "spa or solve or graph" 2>&1 | tail -2
python -c "
import bench
o = bench.solver_leg(cpu=False)
print({k: round(o[k], 2) for k in ('solve_ms', 'solve_ms_cached_analysis', 'solve_symbolic_ms')}, o['solve_rooflines'][0]['gpu_ms'], o['solve_rooflines'][0]['frac'])
" 2>&1 | tail -1
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats -d $GRAFT_REPO_ROOT/gpurun_out/r4spa -o t --output-format csv -- python $GRAFT_REPO_ROOT/tools/quick_spa.py > /dev/null 2>&1
cd $GRAFT_REPO_ROOT; head -8 gpurun_out/r4spa/t_kernel_stats.csv | cut -c1-110
