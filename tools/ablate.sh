#!/bin/bash
# experimental: build ablated variants of the engine into /tmp and time the scan kernel with each
cd $GRAFT_REPO_ROOT/metakssd_amd/csrc
cp ../lib/libmetakssd_hip.so /tmp/lib_orig.so
for ab in ${ABS:-0 5 6}; do
  if [ $ab != 0 ]; then
    /opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -I../../include -I. -Ihost -Wno-unused-value -DMK_ABLATE=$ab -c mk_engine.hip -o /tmp/mk_engine_ab.o 2>/dev/null
    gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_shuf_params.c -o /tmp/a1.o; gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_frontend.c -o /tmp/a2.o; gcc -std=gnu11 -O2 -fPIC -I../../include -Ihost -c host/mk_sketchdir.c -o /tmp/a3.o
    /opt/rocm/bin/hipcc -shared -fPIC --offload-arch=gfx950 -o ../lib/libmetakssd_hip.so /tmp/mk_engine_ab.o /tmp/a1.o /tmp/a2.o /tmp/a3.o
  fi
  cd $GRAFT_REPO_ROOT
  MK_SCAN_THREADS=768 python bench.py --steps 4 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('ablate=$ab scan_ms', round(d['roofline']['avg_launch_ms'],3), 'distinct', d['config']['distinct_keys'])"
  cd metakssd_amd/csrc
done
cp /tmp/lib_orig.so ../lib/libmetakssd_hip.so
