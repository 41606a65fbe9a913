#!/bin/bash
# t_e2e with the framers allowed to run ahead by N row buffers (--ahead)
cd $GRAFT_REPO_ROOT
for a in 0 48 96 0 48 96; do
  MK_E2E_FLAGS="--ahead $a" timeout 600 python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-config5 2>/dev/null | tail -1 > /tmp/b.json
  python3 - $a <<'PY'
import json, sys
d = json.loads(open("/tmp/b.json").read()); t = d["t_e2e"]
print("ahead", sys.argv[1], "gbases_s", t["gbases_s"], "excl_init", t["gbases_s_excl_init"], "runs", [r["written_s"] for r in t["all_runs"]], {k: t["timeline_s"][k] for k in ("engine_ready", "first_push", "last_push", "stream_setup_s", "stream_wait_frame_s")})
PY
done
