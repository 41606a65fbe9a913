cd $GRAFT_REPO_ROOT
export MK_LIBRARY=metakssd_amd/lib_tuning/base/libmetakssd_hip.so
for r in 32 24 16 40; do
  MK_TUNE_SPLIT_NOMASK=1 python3 bench.py --steps 300 --no-cpu-baseline --no-host-legs --no-traffic --no-one-queue --split-cus $r 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); p=j['phases_ms_per_step']; print('nomask R=$r value %.0f ms %.3f scan %.3f resolve %.3f finish %.3f clear %.3f distinct %s | %s' % (j['value'], j['ms_per_step'], p['scan'], p['resolve'], p['finish'], p['clear'], j['config']['distinct_keys'], j['config']['queues'][:40]))"
done
python3 bench.py --steps 300 --no-cpu-baseline --no-host-legs --no-traffic --split-cus 32 2>/dev/null | tail -1 | python3 -c "
import json,sys; j=json.loads(sys.stdin.read()); print('masked R=32 value %.0f ms %.3f one_queue %.3f' % (j['value'], j['ms_per_step'], j['one_queue']['ms_per_step']))"
