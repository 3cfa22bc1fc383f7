#!/bin/bash
# Probe build with TWO units compiled with extra flags: the overlap-save kernels with their 16-byte LDS stores (most sensitive)
# and stft_mm.hip with one of its probe switches -> das4whales_amd/lib/probe/libd4w_<name>.so
#   bash scripts/probe/build_pair.sh nomfma -DD4W_SM_NO_MFMA
set -e
cd "$(dirname "$0")/../.."
name=$1; shift
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p das4whales_amd/lib/probe
F="--offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include"
/opt/rocm/bin/hipcc $F -DD4W_XF_ST128 -c das4whales_amd/csrc/xcorr_fft.hip -o das4whales_amd/lib/probe/xf.$name.o &
/opt/rocm/bin/hipcc $F "$@" -c das4whales_amd/csrc/stft_mm.hip -o das4whales_amd/lib/probe/sm.$name.o &
wait
objs=$(ls das4whales_amd/lib/obj/*.o | grep -v "/xcorr_fft.hip.o\|/stft_mm.hip.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs das4whales_amd/lib/probe/xf.$name.o das4whales_amd/lib/probe/sm.$name.o -o das4whales_amd/lib/probe/libd4w_$name.so
rm das4whales_amd/lib/probe/xf.$name.o das4whales_amd/lib/probe/sm.$name.o
echo built das4whales_amd/lib/probe/libd4w_$name.so
