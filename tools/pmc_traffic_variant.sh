#!/bin/bash
# TCC traffic of the scan kernel for a build variant: VARIANT="-DMK_ABLATE=3" bash tools/pmc_traffic_variant.sh
cd $GRAFT_REPO_ROOT/metakssd_amd/csrc
cp ../lib/libmetakssd_hip.so /tmp/lib_orig.so
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Ihost -Wno-unused-value $VARIANT -c mk_engine.hip -o /tmp/mk_engine_v.o 2>/dev/null
/opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmetakssd_hip.so /tmp/mk_engine_v.o build/mk_setop.o build/mk_shuf_params.o build/mk_frontend.o build/mk_sketchdir.o
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/traffic_*
bash tools/pmc_traffic.sh | tail -1
cp /tmp/lib_orig.so metakssd_amd/lib/libmetakssd_hip.so
