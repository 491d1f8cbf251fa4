KH_SPA_DEBUG=1 python -c "
import bench
o = bench.solver_leg(cpu=False)
" 2>&1 | grep "kh_spa\] host\|analysis\|upload\|ms" | grep -v level | head -20
