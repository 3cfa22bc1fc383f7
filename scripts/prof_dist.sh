export TMPDIR=/tmp
python scripts/dbg_dist.py 20000 120000 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Librccl\|amdgpu.ids\|socket.cpp"
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/prof_dist -o d -- python $GRAFT_REPO_ROOT/scripts/dbg_dist.py 20000 120000 > /dev/null 2>&1
cut -d, -f1-4 $GRAFT_REPO_ROOT/gpurun_out/prof_dist/d_kernel_stats.csv | cut -c1-160 | head -16
