# usage: CFGS="768:80:1 1024:80:0" bash tools/sweep_scan.sh     (threads:max_column_block:onepass)
for cfg in ${CFGS:-1024:80:0 768:80:1 768:80:0}; do IFS=: read t cb op <<< "$cfg";
  MK_SCAN_THREADS=$t MK_SCAN_CB=$cb MK_SCAN_ONEPASS=${op:-0} MK_DEBUG=1 python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | grep -E "^\{|scan cfg" | sort -u | python -c "
import sys,json
for l in sys.stdin:
    if l.startswith('{'):
        d=json.loads(l); print('  ->', round(d['value'],1), 'Gb/s scan_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), d['phases_ms_per_step'])
    else: print(l.strip())
"
done
