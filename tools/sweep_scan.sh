for t in ${THREADS:-768 1024 512}; do for cb in ${CBS:-128 80}; do
  MK_SCAN_THREADS=$t MK_SCAN_CB=$cb python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>&1 | tail -1 | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('threads=$t cb=$cb', round(d['value'],1), 'scan_ms', round(d['roofline']['avg_launch_ms'],3), 'frac', round(d['roofline']['frac'],4), 'finish', round(d['phases_ms_per_step']['finish'],3))"
done; done
