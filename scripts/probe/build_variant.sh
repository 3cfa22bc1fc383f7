#!/bin/bash
# A second build of the library with ONE translation unit compiled with extra flags -> das4whales_amd/lib/probe/libd4w_<name>.so
# (loaded through D4W_LIB):   bash scripts/probe/build_variant.sh strict xcorr_fft.hip -DD4W_XF_STRICT
set -e
cd "$(dirname "$0")/../.."
name=$1; unit=$2; shift 2
python -c "import __graft_entry__ as g; g.build()" > /dev/null
mkdir -p das4whales_amd/lib/probe
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -fno-slp-vectorize -I include "$@" \
    -c das4whales_amd/csrc/$unit -o das4whales_amd/lib/probe/$unit.$name.o
objs=$(ls das4whales_amd/lib/obj/*.o | grep -v "/$unit.o")
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC $objs das4whales_amd/lib/probe/$unit.$name.o -o das4whales_amd/lib/probe/libd4w_$name.so
rm das4whales_amd/lib/probe/$unit.$name.o
echo built das4whales_amd/lib/probe/libd4w_$name.so
