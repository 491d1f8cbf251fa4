#!/bin/bash
# every -m gpu test, the way the round-end driver runs them, then smoke()
out=$GRAFT_REPO_ROOT/gpurun_out/${1:-full}
mkdir -p $out
cd $GRAFT_REPO_ROOT
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest.log 2>&1
echo "pytest rc=$?" >> $out/pytest.log
grep -v "^Registering\|^Unregistering\|amdgpu.ids\|^\[W9\|Gloo\|^RCCL\|^HIP ver\|^ROCm\|^Hostname\|^Librccl" $out/pytest.log | tail -12
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | grep -v "amdgpu.ids\|Registering" | tail -3
