cd $GRAFT_REPO_ROOT
python - <<PY
import sys; sys.path.insert(0,'.')
from metakssd_amd import capi
capi.lib.mk_synth_fastq_write(b"/dev/shm/in.fq", 1, 0, 10000000, 150)
capi.Shuf.generate(11,6,3,11).write("/dev/shm/L3K11.shuf")
PY
MK_DEBUG=1 ./metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/out1 --quiet /dev/shm/in.fq
MK_DEBUG=1 ./metakssd_amd/bin/metakssd dist -L /dev/shm/L3K11.shuf -A -o /dev/shm/out2 --quiet /dev/shm/in.fq
rm -rf /dev/shm/in.fq /dev/shm/L3K11.shuf /dev/shm/out1 /dev/shm/out2
