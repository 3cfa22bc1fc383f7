#!/bin/bash
# Probe build of the library with cycle marks in find_peaks_prom (-DD4W_FP_TIMING: thread 0 of every workgroup adds the cycles
# between phase marks to a device table) -> das4whales_amd/lib/probe/libd4w_fpt.so.  Run HERE (hipcc cross-compiles), then
#   gpurun -- 'D4W_LIB=$PWD/das4whales_amd/lib/probe/libd4w_fpt.so python scripts/probe/fp_timing.py'
set -e
cd "$(dirname "$0")/../.."
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p das4whales_amd/lib/probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include -DD4W_FP_TIMING \
    -c das4whales_amd/csrc/spectral.hip -o das4whales_amd/lib/probe/spectral_fpt.o
objs=$(ls das4whales_amd/lib/obj/*.o | grep -v spectral.hip.o)
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs das4whales_amd/lib/probe/spectral_fpt.o -o das4whales_amd/lib/probe/libd4w_fpt.so
rm das4whales_amd/lib/probe/spectral_fpt.o
echo built das4whales_amd/lib/probe/libd4w_fpt.so
