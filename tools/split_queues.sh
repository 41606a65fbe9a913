#!/bin/bash
# bench.py's timed region with one engine on one queue against two engines in turn on split queues (MK_OPT_SPLIT_CUS), DESIGN.md 4.2.
cd $GRAFT_REPO_ROOT
out=gpurun_out/split_queues.txt; : > $out
for r in ${SPLITS:-0 32 48 32q 24q 0}; do   # NNq: a scan queue per engine instead of one shared
  extra=""; case $r in *q) extra="--split-two-scan-queues"; r=${r%q};; esac
  line=$(python3 bench.py --steps ${STEPS:-300} --no-cpu-baseline --no-host-legs --no-traffic --no-queue-trial --split-cus $r $extra 2>gpurun_out/split_queues.err | tail -1)
  echo "split-cus $r $extra: $(echo "$line" | python3 -c 'import json,sys; j=json.loads(sys.stdin.read()); p=j["phases_ms_per_step"]; print("value %.0f ms_per_step %.3f scan %.3f resolve %.3f finish %.3f side %.3f clear %.3f frac %.3f distinct %s" % (j["value"], j["ms_per_step"], p["scan"], p["resolve"], p["finish"], p["finish_side_stream"], p["clear"], j["roofline"]["frac"], j["config"]["distinct_keys"]))' 2>&1)" | tee -a $out
  tail -2 gpurun_out/split_queues.err | grep -v "^$" >> $out
done
