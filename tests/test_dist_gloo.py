"""N>1 path on CPU: world_size 2 and 3 over gloo (the GPU run uses the same shard.py code over RCCL)"""
import os
import subprocess
import sys

import pytest

import util_inputs as ui

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.parametrize("world", [2, 3, 4])
def test_sharded_sketch_merge_over_gloo(world, tmp_path):
    """both exchanges of SURVEY.md 8e: the gather of whole lists to rank 0, and the all-to-all by key % world with the gather of
    reduced slices (tests/dist_worker.py)"""
    result = str(tmp_path / "result.txt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MK_DIST_RESULT=result)
    port = ui.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(result).read().startswith("OK")


@pytest.mark.parametrize("world", [2, 3])
def test_file_sharded_sketches_gather_in_file_order_over_gloo(world, tmp_path):
    """config 5: whole files are the unit, no reduction: rank 0 receives every file's per-component id
    arrays in file order"""
    result = str(tmp_path / "result.txt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MK_DIST_RESULT=result)
    port = ui.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_files_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(result).read() == "OK"


@pytest.mark.parametrize("world", [2, 3])
def test_query_sharded_search_rows_gather_over_gloo(world, tmp_path):
    """`dist -r` search: query sketches are the unit, every rank counts its block against the whole database,
    rank 0 receives the rows in sketch order -- equal to the unsharded matrix"""
    result = str(tmp_path / "result.txt")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", MK_DIST_RESULT=result)
    port = ui.free_port()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.join(ROOT, "tests", "dist_search_worker.py")]
    r = subprocess.run(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
    assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
    assert open(result).read() == "OK"


def test_shard_ranges_cover_and_order():
    sys.path.insert(0, ROOT)
    from metakssd_amd.shard import shard_range
    for total in (0, 1, 7, 64, 1000003):
        for world in (1, 2, 3, 8):
            r = [shard_range(total, k, world) for k in range(world)]
            assert r[0][0] == 0 and r[-1][1] == total
            assert all(a[1] == b[0] for a, b in zip(r, r[1:]))
            assert max(hi - lo for lo, hi in r) - min(hi - lo for lo, hi in r) <= 1
