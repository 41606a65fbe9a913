"""Golden vectors produced by the REAL reference (tests/golden/make_golden.py, oracle/_ref/metakssd -p 1).

 * CPU: the oracle restatement reproduces them byte for byte  -> the oracle is pinned.
 * GPU: the product CLI (`metakssd dist`, HIP engine)  reproduces them byte for byte.
cofiles.stat is compared field-wise: the reference leaves 3 padding bytes uninitialised."""
import filecmp
import hashlib
import json
import os
import struct
import subprocess

import pytest

import golden_cases as gc

ROOT = gc.ROOT
MANIFEST = json.load(open(os.path.join(gc.GOLDEN, "manifest.json")))
ORACLE_CLI = os.path.join(ROOT, "oracle", "kssd_oracle_cli")
PRODUCT_CLI = os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")
HEAVY = {"pool2000_L2K11", "fasta_L2K11", "fasta_uniq_L2K11"}  # 4.3 GB oracle table each


def parse_stat(path):
    b = open(path, "rb").read()
    shuf_id, koc = struct.unpack_from("<IB", b, 0)
    kmerlen, dim_rd_len, comp_num, infile_num, all_ctx = struct.unpack_from("<iiiiQ", b, 8)
    cts = list(struct.unpack_from("<%dI" % infile_num, b, 32))
    names = [b[32 + 4 * infile_num + 256 * i: 32 + 4 * infile_num + 256 * (i + 1)].split(b"\0", 1)[0].decode()
             for i in range(infile_num)]
    assert len(b) == 32 + 260 * infile_num
    return dict(shuf_id=shuf_id, koc=koc, kmerlen=kmerlen, dim_rd_len=dim_rd_len, comp_num=comp_num,
                infile_num=infile_num, all_ctx_ct=all_ctx, ctx_ct=cts), names


@pytest.fixture(scope="module")
def shuf_files(tmp_path_factory):
    d = tmp_path_factory.mktemp("shuf")
    cache = {}

    def get(name):
        if name not in cache:
            p = str(d / (name + ".shuf"))
            gc.make_shuf(name, p)
            cache[name] = p
        return cache[name]
    return get


def check_against_golden(case, outdir, input_path):
    entry = MANIFEST["cases"][case]
    exp = os.path.join(gc.GOLDEN, "expected", case)
    want_files = sorted(os.listdir(exp))
    got_files = sorted(f for f in os.listdir(outdir) if f.startswith("combco"))
    assert got_files == want_files
    for f in want_files:
        assert filecmp.cmp(os.path.join(exp, f), os.path.join(outdir, f), shallow=False), "%s: %s differs" % (case, f)
    stat, names = parse_stat(os.path.join(outdir, "cofiles.stat"))
    assert stat == entry["stat"]
    assert names == [input_path]


@pytest.mark.parametrize("name", sorted(MANIFEST["shufs"]))
def test_shuf_generator_is_stable(name, shuf_files):
    """the seeded .shuf generator must keep producing the bytes the golden vectors were made with"""
    got = hashlib.sha256(open(shuf_files(name), "rb").read()).hexdigest()
    assert got == MANIFEST["shufs"][name]["sha256"]


@pytest.mark.parametrize("case", sorted(MANIFEST["cases"]))
def test_oracle_reproduces_reference_golden(case, shuf_files, tmp_path):
    entry = MANIFEST["cases"][case]
    inp = gc.build_input(case, str(tmp_path))
    out = str(tmp_path / "out")
    r = subprocess.run([ORACLE_CLI, "-L", shuf_files(entry["shuf"])] + entry["flags"] + ["-o", out, inp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if entry["aborted"]:
        assert r.returncode != 0  # the reference aborts with "the context space is too crowd"
        return
    assert r.returncode == 0, r.stderr.decode()
    check_against_golden(case, out, inp)


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST["cases"]))
def test_product_cli_reproduces_reference_golden(case, shuf_files, tmp_path):
    """`metakssd dist -L x.shuf [-A] [-u] -o out input` through the HIP engine == the reference's sketch directory"""
    entry = MANIFEST["cases"][case]
    inp = gc.build_input(case, str(tmp_path))
    out = str(tmp_path / "out")
    r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files(entry["shuf"])] + entry["flags"] + ["-p", "8", "-o", out, inp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if entry["aborted"]:
        assert r.returncode != 0 and b"too crowd" in r.stderr
        return
    assert r.returncode == 0, r.stderr.decode()
    check_against_golden(case, out, inp)
