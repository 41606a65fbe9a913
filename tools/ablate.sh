#!/bin/bash
# experimental: build variants of the engine with extra -D flags and time the scan kernel with each
# usage: VARIANTS="-DMK_ABLATE=3|-DMK_FILTER_BITS=4" bash tools/ablate.sh
cd $GRAFT_REPO_ROOT/metakssd_amd/csrc
cp ../lib/libmetakssd_hip.so /tmp/lib_orig.so
gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_shuf_params.c -o /tmp/a1.o; gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_frontend.c -o /tmp/a2.o; gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_sketchdir.c -o /tmp/a3.o
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Ihost -c mk_setop.hip -o /tmp/a4.o 2>/dev/null
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Ihost -c mk_mco.hip -o /tmp/a5.o 2>/dev/null
gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_distprint.c -o /tmp/a6.o
IFS='|' read -ra VS <<< "${VARIANTS:-none|-DMK_ABLATE=3}"
for v in "${VS[@]}"; do
  if [ "$v" != "none" ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Ihost -Wno-unused-value $v -c mk_engine.hip -o /tmp/mk_engine_ab.o 2>/dev/null
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmetakssd_hip.so /tmp/mk_engine_ab.o /tmp/a1.o /tmp/a2.o /tmp/a3.o /tmp/a4.o /tmp/a5.o /tmp/a6.o -lm
  else
    cp /tmp/lib_orig.so ../lib/libmetakssd_hip.so
  fi
  cd $GRAFT_REPO_ROOT
  python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('variant [$v] scan_ms', round(d['roofline']['avg_launch_ms'],3), 'resolve', round(d['phases_ms_per_step']['resolve'],3), 'finish', round(d['phases_ms_per_step']['finish'],3), 'Gb/s', round(d['value'],1), 'distinct', d['config']['distinct_keys'])"
  cd metakssd_amd/csrc
done
cp /tmp/lib_orig.so ../lib/libmetakssd_hip.so
