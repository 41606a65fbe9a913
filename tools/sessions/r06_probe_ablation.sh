#!/bin/bash
# round 6: what FEWER filter probes per 8-base window would buy the scan kernel at most -- a copy of the sources with the tuned loop's four pair
# probes cut to three and to two (the dropped pairs are simply not tested: results WRONG, timing only), built beside the product, bench.py's
# serial-finish passes on each in turn on one box.  The shipped sources are not touched.
cd "$(dirname "$0")/../.." || exit 1
O=gpurun_out/r06f; mkdir -p $O
rm -rf /tmp/abl && mkdir -p /tmp/abl && cp -r metakssd_amd include /tmp/abl/
for NP in 3 2; do
  python3 - $NP <<'PY'
import sys
np_ = int(sys.argv[1])
p = '/tmp/abl/metakssd_amd/csrc/mk_kernels.hip.h'
s = open('metakssd_amd/csrc/mk_kernels.hip.h').read()
old_loop = "              for (uint32_t t = 0; t < 4; t++) {\n                const uint32_t wsrc = bj[2u * t + 1u];"
assert s.count(old_loop) == 1
s = s.replace(old_loop, old_loop.replace("t < 4", "t < %du" % np_))
old_test = "              const uint32_t tt0 = mm[0] & ~dd[0], tt1 = mm[1] & ~dd[1], tt2 = mm[2] & ~dd[2], tt3 = mm[3] & ~dd[3];\n              fired = min(min(tt0, tt1), min(tt2, tt3)) == 0u;"
assert s.count(old_test) == 1
new_test = ("              const uint32_t tt0 = mm[0] & ~dd[0], tt1 = mm[1] & ~dd[1], tt2 = mm[2] & ~dd[2];\n              fired = min(min(tt0, tt1), tt2) == 0u;" if np_ == 3 else
            "              const uint32_t tt0 = mm[0] & ~dd[0], tt1 = mm[1] & ~dd[1];\n              fired = min(tt0, tt1) == 0u;")
s = s.replace(old_test, new_test)
open(p, 'w').write(s)
PY
  make -s -C /tmp/abl/metakssd_amd/csrc ../lib/libmetakssd_hip.so > $O/abl_build_$NP.log 2>&1
  cp /tmp/abl/metakssd_amd/lib/libmetakssd_hip.so /tmp/abl/lib_np$NP.so
done
for rep in 1 2; do
  for v in product np3 np2; do
    lib=metakssd_amd/lib/libmetakssd_hip.so; [ $v = np3 ] && lib=/tmp/abl/lib_np3.so; [ $v = np2 ] && lib=/tmp/abl/lib_np2.so
    MK_LIBRARY=$lib python3 bench.py --steps 40 --warmup 5 --no-host-legs --no-cpu-baseline --no-split-leg --serial-finish 2> /dev/null | python3 -c "
import json,sys
j=json.loads(sys.stdin.readline()); p=j['phases_ms_per_step']
print(json.dumps({'variant':'$v','rep':$rep,'ms_per_step':round(j['ms_per_step'],4),'scan_ms':round(j['roofline']['avg_launch_ms'],4),'resolve_ms':round(p['resolve'],4),'distinct_keys':j['config']['distinct_keys']}))" >> $O/probe_ablation.jsonl
  done
done
cat $O/probe_ablation.jsonl
