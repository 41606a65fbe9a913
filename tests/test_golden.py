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
HEAVY = {"pool2000_L2K11", "fasta_L2K11", "fasta_uniq_L2K11", "lowcov_n3_L2K11"}  # 4.3 GB oracle table each


def golden_listing(rel):
    """file names the manifest pins directly under tests/golden/<rel> -- the expectation is the manifest, never os.listdir of
    the working tree: a fixture that is not in the checkout must fail the case, not shrink it"""
    pre = rel.rstrip("/") + "/"
    names = sorted(k[len(pre):] for k in MANIFEST["tree"] if k.startswith(pre) and "/" not in k[len(pre):])
    assert names, "manifest pins nothing under " + rel
    return names


def test_golden_tree_is_complete_and_unchanged():
    """every fixture the manifest pins exists with the pinned bytes, and nothing unpinned sits beside them
    (the count half of the -A vectors, combco.N.a, pins iseq2comem.c:516-562)"""
    tree = MANIFEST["tree"]
    for rel, want in sorted(tree.items()):
        p = os.path.join(gc.GOLDEN, rel)
        assert os.path.isfile(p), "golden fixture missing from the checkout: " + rel
        assert hashlib.sha256(open(p, "rb").read()).hexdigest() == want, "golden fixture changed: " + rel
    have = set()
    for sub in ("expected", "inputs"):
        for dp, _, fs in os.walk(os.path.join(gc.GOLDEN, sub)):
            have.update(os.path.relpath(os.path.join(dp, f), gc.GOLDEN) for f in fs)
    assert have == set(tree), "unpinned files: %s" % sorted(have - set(tree))
    for sec in ("cases", "set_cases"):  # the per-case hashes written when the reference made the vectors agree with the tree
        for case, e in MANIFEST[sec].items():
            if isinstance(e.get("files"), dict):
                for f, h in e["files"].items():
                    assert tree["expected/%s/%s" % (case, f)] == h
    acount = [k for k in tree if k.endswith(".a")]
    assert len(acount) >= 29  # every -A case carries its 16-bit counts


def test_golden_fixtures_are_tracked_by_git():
    """.gitignore once hid the 29 combco.N.a files (pattern *.a): the tree must be what a clone gets"""
    if not os.path.isdir(os.path.join(ROOT, ".git")):
        pytest.skip("not a git checkout (gpurun snapshot)")
    r = subprocess.run(["git", "-C", ROOT, "ls-files", "tests/golden"], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if r.returncode != 0:
        pytest.skip("git unavailable")
    tracked = {ln[len("tests/golden/"):] for ln in r.stdout.decode().splitlines()}
    missing = sorted(set(MANIFEST["tree"]) - tracked)
    assert not missing, "pinned but untracked (check .gitignore): %s" % missing[:5]


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
    want_files = golden_listing("expected/" + case)
    assert want_files == sorted(entry["files"])
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


REF_HIP = os.path.join(ROOT, "oracle", "_ref_hip", "metakssd")


@pytest.mark.gpu
@pytest.mark.parametrize("shuf,flags", [("L3K10", []), ("L2K11", ["-u"]), ("L1K7", [])])
def test_product_cli_batches_with_files_that_cannot_travel_in_one(shuf, flags, shuf_files, tmp_path):
    """a directory whose FASTA files go to the device in batches, with files in between that cannot (a .gz genome, a FASTQ file):
    those are sketched alone in their place in the input order; the directory is the one `--no-batch` (file by file) writes, also
    with batches of two files and of 1 MiB; with files that GREW or SHRANK between the planning stat() and the readers' pread()
    (MK_TEST_PLAN_SKEW: a grown file is sketched alone from all of its text, a shrunk one is what it is now -- both read to EOF like
    the reference's zcat -fc | fread, iseq2comem.c:226-233); and with 512-slot tables per file (MK_BATCH_TAB_BITS=9), which send most
    files to the sketched-alone path of mk_sketch_batch_end while the next batch is queued"""
    import gzip
    import numpy as np
    import util_inputs as ui
    rs = np.random.RandomState(93)
    d = tmp_path / "dir"
    d.mkdir()
    for i in range(14):
        n = [90000, 700, 30000, 250000][i % 4]
        fa = ui.fasta_bytes([ui.rand_seq(rs, n // 2 + 40), ui.rand_seq(rs, n - n // 2)])
        if i in (6, 7):  # an N in every row the readers pack: wide rows then all have extension rows, more than the file's place in the
            sq = bytearray(ui.rand_seq(rs, 200000))  # batch buffer allows for -- the rows go to memory of their own (breader_run)
            sq[::97] = b"N" * len(sq[::97])
            fa = ui.fasta_bytes([bytes(sq)])
        if i == 4:
            with gzip.open(str(d / ("g%02d.fna.gz" % i)), "wb") as f:
                f.write(fa)
        elif i == 9:
            open(str(d / ("g%02d.fq" % i)), "wb").write(ui.fastq_bytes(ui.pool_reads(rs, 20000, 300)))
        else:
            open(str(d / ("g%02d.fna" % i)), "wb").write(fa)
    base = [PRODUCT_CLI, "dist", "-L", shuf_files(shuf)] + flags
    outs = {}
    variants = (("batches", [], {}), ("file_by_file", ["--no-batch"], {}), ("pairs", ["--batch-files", "2"], {}), ("one_mib", ["--batch-mib", "1", "-p", "3"], {}),
                ("text", ["--batch-text"], {}), ("narrow", ["--batch-narrow"], {}),
                ("grew", ["--batch-files", "3"], {"MK_TEST_PLAN_SKEW": "2:-301"}), ("grew_text", ["--batch-text"], {"MK_TEST_PLAN_SKEW": "3:-17"}),
                ("shrank", [], {"MK_TEST_PLAN_SKEW": "2:4099"}), ("shrank_text", ["--batch-text", "--batch-files", "4"], {"MK_TEST_PLAN_SKEW": "3:1000"}),
                ("tiny_tables", ["--batch-files", "3"], {"MK_BATCH_TAB_BITS": "9"}), ("tiny_tables_text", ["--batch-text", "--batch-files", "4"], {"MK_BATCH_TAB_BITS": "9"}))
    for tag, extra, env in variants:
        out = str(tmp_path / tag)
        r = subprocess.run(base + extra + ["-o", out, str(d)], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=dict(os.environ, **env))
        assert r.returncode == 0, (tag, r.stderr.decode())
        outs[tag] = out
    ref = outs["file_by_file"]
    for tag in [v[0] for v in variants if v[0] != "file_by_file"]:
        assert sorted(os.listdir(outs[tag])) == sorted(os.listdir(ref)), tag
        for f in sorted(os.listdir(ref)):
            if f == "cofiles.stat":
                assert parse_stat(os.path.join(outs[tag], f)) == parse_stat(os.path.join(ref, f)), tag
            else:
                assert filecmp.cmp(os.path.join(outs[tag], f), os.path.join(ref, f), shallow=False), (tag, f)
    st, names = parse_stat(os.path.join(ref, "cofiles.stat"))
    assert st["infile_num"] == 14 and sum(st["ctx_ct"]) > 50


@pytest.mark.gpu
@pytest.mark.parametrize("shuf,flags", [("L3K11", ["-A"]), ("L2K11", ["-A"]), ("L3K10", ["-n", "2", "-Q", "50"]), ("L1K7", ["-A"])])
def test_product_cli_packed_rows_equal_text_rows(shuf, flags, shuf_files, tmp_path):
    """the command line frames FASTQ reads of up to 152 bases as 64-byte packed rows where the geometry has a tuned kernel (the default)
    and as text rows with --ascii-rows (and for L1K7, which has none): the sketch directories are the same byte for byte -- reads
    with N, lower case, ragged lengths, a few reads beyond 152 bases (their buffers fall back to text rows), CRLF"""
    import numpy as np
    import util_inputs as ui
    rs = np.random.RandomState(91)
    g = ui.rand_seq(rs, 80000)
    seqs = []
    for i in range(30000):
        n = int(rs.randint(0, 153)) if i % 400 else int(rs.randint(153, 400))
        a = int(rs.randint(0, len(g) - 400))
        q = bytearray(g[a:a + n])
        if n and i % 9 == 0:
            q[int(rs.randint(0, n))] = ord("N")
        if i % 13 == 0:
            q = bytearray(bytes(q).lower())
        seqs.append(bytes(q))
    quals = [bytes(rs.randint(44, 64, len(x)).astype(np.uint8)) for x in seqs]
    outs = {}
    for crlf in (False, True):
        path = str(tmp_path / ("reads%d.fq" % crlf))
        open(path, "wb").write(ui.fastq_bytes(seqs, crlf=crlf, quals=quals))
        for tag, extra in (("packed", []), ("text", ["--ascii-rows"]), ("packed_small_chunks", ["--chunk-mib", "1", "-p", "5"])):
            out = str(tmp_path / ("%s%d" % (tag, crlf)))
            r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files(shuf)] + flags + extra + ["-o", out, path], stdout=subprocess.PIPE,
                               stderr=subprocess.PIPE)
            assert r.returncode == 0, r.stderr.decode()
            outs[(tag, crlf)] = out
        for tag in ("text", "packed_small_chunks"):
            a, b = outs[("packed", crlf)], outs[(tag, crlf)]
            names = sorted(f for f in os.listdir(a) if f.startswith("combco"))
            assert names and names == sorted(f for f in os.listdir(b) if f.startswith("combco"))
            for f in names:
                assert filecmp.cmp(os.path.join(a, f), os.path.join(b, f), shallow=False), (tag, crlf, f)
        assert "-A" not in flags or os.path.getsize(os.path.join(outs[("packed", crlf)], "combco.0")) >= 40


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST["cases"]))
def test_reference_program_with_dropin_tu_reproduces_golden(case, shuf_files, tmp_path):
    """oracle/_ref_hip/metakssd = the reference's own translation units (its main, option parser, dist_dispatch, run_stageI
    with the per-component concatenation and cofiles.stat, command_dist.c:341-500) linked with integration/iseq2comem_hip.c
    in place of iseq2comem.c (oracle/Makefile, target ref_hip): the library is a drop-in under the reference's call sites
    (command_dist.c:380-398), and the sketch directories are the ones the unmodified reference wrote"""
    if not os.path.exists(REF_HIP):
        pytest.skip("oracle/_ref_hip/metakssd not built (needs the reference sources: make -C oracle ref_hip)")
    entry = MANIFEST["cases"][case]
    inp = gc.build_input(case, str(tmp_path))
    out = str(tmp_path / "out")
    r = subprocess.run([REF_HIP, "dist", "-L", shuf_files(entry["shuf"])] + entry["flags"] + ["-p", "4", "-o", out, inp],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    if entry["aborted"]:
        assert b"too crowd" in r.stderr  # err(errno, ..): the exit status is whatever errno was, as in the reference
        return
    assert r.returncode == 0, r.stderr.decode()
    check_against_golden(case, out, inp)


def _write_multi_inputs(d):
    """three FASTA genomes + (separately) three FASTQ files, deterministic"""
    import numpy as np
    import util_inputs as ui
    rs = np.random.RandomState(77)
    fa, fq = [], []
    os.makedirs(os.path.join(d, "genomes"))
    os.makedirs(os.path.join(d, "reads"))
    for i, n in enumerate((9000, 14000, 5000)):
        g = ui.rand_seq(rs, n)
        p = os.path.join(d, "genomes", "g%d.%s" % (i, ("fa", "fna", "fasta")[i]))
        open(p, "wb").write(ui.fasta_bytes([g[: n // 2], g[n // 2:]], width=(60, 70, 80)[i]))
        fa.append(p)
    for i, n in enumerate((300, 1, 120)):
        p = os.path.join(d, "reads", "s%d.fq" % i)
        open(p, "wb").write(ui.fastq_bytes(ui.pool_reads(rs, 5000, n)))
        fq.append(p)
    return sorted(fa), sorted(fq)


@pytest.mark.gpu
@pytest.mark.parametrize("kind,shuf,flags", [("fasta", "L1K7", []), ("fasta", "L1K7", ["-u"]), ("fasta", "L2K11", []),
                                               ("fastq", "L1K7", ["-A"]), ("fastq", "L0K6", ["-A"])])
def test_product_cli_multi_file_directory_equals_oracle(kind, shuf, flags, shuf_files, tmp_path):
    """stage-I bookkeeping over several inputs (command_dist.c:408-500): per-file blocks in input order,
    combco.index.N cumulative counts, cofiles.stat counts and names.  The product CLI expands a directory in sorted
    order; the oracle CLI (pinned to the reference) gets the same files in that order."""
    fa, fq = _write_multi_inputs(str(tmp_path))
    files = fa if kind == "fasta" else fq
    dir_arg = os.path.dirname(files[0])
    out_p, out_o = str(tmp_path / "out_product"), str(tmp_path / "out_oracle")
    r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files(shuf)] + flags + ["-o", out_p, dir_arg],
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    r = subprocess.run([ORACLE_CLI, "-L", shuf_files(shuf)] + flags + ["-o", out_o] + files,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    names_p = sorted(f for f in os.listdir(out_p) if f.startswith("combco"))
    names_o = sorted(f for f in os.listdir(out_o) if f.startswith("combco"))
    assert names_p == names_o and names_p
    for f in names_p:
        assert filecmp.cmp(os.path.join(out_p, f), os.path.join(out_o, f), shallow=False), f
    sp, np_ = parse_stat(os.path.join(out_p, "cofiles.stat"))
    so, no_ = parse_stat(os.path.join(out_o, "cofiles.stat"))
    assert sp == so and np_ == no_ == files
    assert sp["infile_num"] == 3 and sum(sp["ctx_ct"]) == sp["all_ctx_ct"]
    idx = struct.unpack("<4Q", open(os.path.join(out_p, "combco.index.0"), "rb").read())
    assert idx[0] == 0 and list(idx) == sorted(idx)


# ---- `set -u` / `set -q`: pan.N / uniq_pan.N made by the reference's own `set` --------------------


def parse_header(path):
    b = open(path, "rb").read()
    assert len(b) == 32  # sketch_union() writes the co_dstat_t header only (command_set.c:274)
    shuf_id, koc = struct.unpack_from("<IB", b, 0)
    kmerlen, dim_rd_len, comp_num, infile_num, all_ctx = struct.unpack_from("<iiiiQ", b, 8)
    return dict(shuf_id=shuf_id, koc=koc, kmerlen=kmerlen, dim_rd_len=dim_rd_len, comp_num=comp_num,
                infile_num=infile_num, all_ctx_ct=all_ctx)


def check_set_against_golden(case, outdir, inputs):
    entry = MANIFEST["set_cases"][case]
    exp = os.path.join(gc.GOLDEN, "expected", case)
    want = golden_listing("expected/" + case)
    got = sorted(f for f in os.listdir(outdir) if f != "cofiles.stat")
    assert got == want  # in particular: no combco.N.a after -i / -s even when the header says koc (command_set.c:321-425)
    for f in want:
        assert filecmp.cmp(os.path.join(exp, f), os.path.join(outdir, f), shallow=False), "%s: %s differs" % (case, f)
    if entry["op"] == "-g":  # grouped directory: new header, per-taxon counts, "<taxid>_<name>" names (command_set.c:929-966)
        stat, names = parse_stat(os.path.join(outdir, "cofiles.stat"))
        assert stat == entry["stat"] and names == entry["names"]
        assert stat["koc"] == 0 and sum(stat["ctx_ct"]) == stat["all_ctx_ct"]
    elif "stat" in entry:  # -i / -s: the whole stat file of the input directory with recounted per-file sizes
        stat, names = parse_stat(os.path.join(outdir, "cofiles.stat"))
        assert stat == entry["stat"] and names == inputs
        assert sum(stat["ctx_ct"]) != stat["all_ctx_ct"] or entry["ids"] == stat["all_ctx_ct"]  # header total is NOT recounted
    else:
        assert parse_header(os.path.join(outdir, "cofiles.stat")) == entry["header"]


def run_set_case(case, shuf_files, tmp_path, dist_cmd, set_cmd):
    entry = MANIFEST["set_cases"][case]
    inputs = gc.build_set_inputs(case, str(tmp_path))
    sk, out = str(tmp_path / "sk"), str(tmp_path / "pan")
    r = subprocess.run(dist_cmd + ["-L", shuf_files(entry["shuf"])] + entry["flags"] + ["-o", sk] + inputs,
                       stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    op_args = [entry["op"]]
    if entry["op"] == "-g":
        taxf = str(tmp_path / "tax.tsv")
        open(taxf, "w").write("".join(t + "\n" for t in entry["tax"]))
        op_args = ["-g", taxf]
    if "pan" in entry:
        pin = gc.build_set_inputs(case, str(tmp_path), pan=True)
        psk, pdir = str(tmp_path / "psk"), str(tmp_path / "pdir")
        r = subprocess.run(dist_cmd + ["-L", shuf_files(entry["shuf"])] + entry["pan"]["flags"] + ["-o", psk] + pin,
                           stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        r = subprocess.run(set_cmd + [entry["pan"]["op"], "-o", pdir, psk], input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        op_args = [entry["op"], pdir]
    r = subprocess.run(set_cmd + op_args + ["-o", out, sk], input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    check_set_against_golden(case, out, inputs)


@pytest.mark.parametrize("case", sorted(MANIFEST["set_cases"]))
def test_oracle_set_reproduces_reference_golden(case, shuf_files, tmp_path):
    run_set_case(case, shuf_files, tmp_path, [ORACLE_CLI], [ORACLE_CLI, "set"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST["set_cases"]))
def test_product_cli_set_reproduces_reference_golden(case, shuf_files, tmp_path):
    """`metakssd dist ...` then `metakssd set -u|-q` through the HIP engine and the device dictionaries"""
    run_set_case(case, shuf_files, tmp_path, [PRODUCT_CLI, "dist", "-p", "4"], [PRODUCT_CLI, "set"])


@pytest.mark.gpu
def test_product_cli_set_single_sketch_rename(shuf_files, tmp_path):
    """one sketch in the directory and the answer Y: combco.N is renamed in place (command_set.c:254-267)"""
    inputs = gc.build_set_inputs("set_u_single_N_L1K7", str(tmp_path))
    sk = str(tmp_path / "sk")
    r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files("L1K7"), "-o", sk] + inputs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    before = open(os.path.join(sk, "combco.0"), "rb").read()
    r = subprocess.run([PRODUCT_CLI, "set", "-q", "-o", str(tmp_path / "unused"), sk], input=b"y\n", stdout=subprocess.PIPE,
                       stderr=subprocess.PIPE)
    assert r.returncode == 0 and b"only 1 sketch" in r.stdout
    assert not os.path.exists(os.path.join(sk, "combco.0")) and not os.path.exists(str(tmp_path / "unused"))
    assert open(os.path.join(sk, "uniq_pan.0"), "rb").read() == before


@pytest.mark.gpu
def test_product_cli_set_print_names_and_id_mismatch(shuf_files, tmp_path):
    """-P prints "<count>\\t<name>" per sketch (command_set.c:610-631); -i with a pan of another .shuf id is refused (:341)"""
    inputs = gc.build_set_inputs("set_u_strains_L1K7", str(tmp_path))
    sk, sk2 = str(tmp_path / "sk"), str(tmp_path / "sk2")
    for d, sh in ((sk, "L1K7"), (sk2, "L0K6")):
        r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files(sh), "-o", d] + inputs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
    stat, names = parse_stat(os.path.join(sk, "cofiles.stat"))
    r = subprocess.run([PRODUCT_CLI, "set", "-P", sk], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0
    assert r.stdout.decode().splitlines() == ["%d\t%s" % (c, n) for c, n in zip(stat["ctx_ct"], names)]
    pan2 = str(tmp_path / "pan2")
    r = subprocess.run([PRODUCT_CLI, "set", "-u", "-o", pan2, sk2], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0
    r = subprocess.run([PRODUCT_CLI, "set", "-i", pan2, "-o", str(tmp_path / "x"), sk], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode != 0 and b"sketcing id not match" in r.stderr


@pytest.mark.gpu
def test_product_cli_set_g_taxon_without_kmer_in_a_component(shuf_files, tmp_path):
    """three genomes of a few kilobases over L2K11's 16 components: some (taxon, component) pairs hold no k-mer at all.  The reference
    evaluates LOG2(0) there (command_set.c:878) and writes an EMPTY block for that component -- pinned against the compiled reference in
    oracle/check_vs_ref.py (case set_g_empty_component); the product used to stop with an error (found by tools/fuzz_pipeline.py,
    round 6).  Product == oracle on the same sketch directory, and the README's next steps (-q, -i) take the grouped directory."""
    import numpy as np
    import util_inputs as ui
    g = ui.rand_seq(np.random.RandomState(66), 30000)
    refs = []
    for i, (a, b) in enumerate([(0, 3000), (5000, 9000), (20000, 22500)]):
        p = str(tmp_path / ("tiny%d.fa" % i))
        open(p, "wb").write(ui.fasta_bytes([g[a:b]]))
        refs.append(p)
    taxf = str(tmp_path / "tax.tsv")
    open(taxf, "w").write("1\tone\n2\n3\tthree\n")
    sk = str(tmp_path / "sk")
    r = subprocess.run([PRODUCT_CLI, "dist", "-L", shuf_files("L2K11"), "-o", sk] + refs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    sizes = np.stack([np.diff(np.fromfile(os.path.join(sk, "combco.index.%d" % c), np.uint64).astype(np.int64)) for c in range(16)])
    assert (sizes == 0).any() and sizes.sum() > 0  # the case is what it says
    outs = {}
    for who, cli in (("product", [PRODUCT_CLI, "set"]), ("oracle", [ORACLE_CLI, "set"])):
        d = str(tmp_path / ("grp_" + who))
        r = subprocess.run(cli + ["-g", taxf, "-o", d, sk], input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (who, r.stderr.decode())
        outs[who] = d
    assert sorted(os.listdir(outs["product"])) == sorted(os.listdir(outs["oracle"]))
    for f in sorted(os.listdir(outs["oracle"])):
        a, b = os.path.join(outs["product"], f), os.path.join(outs["oracle"], f)
        if f == "cofiles.stat":
            assert parse_stat(a) == parse_stat(b)
        else:
            assert filecmp.cmp(a, b, shallow=False), f
    uq, db = str(tmp_path / "uq"), str(tmp_path / "db")
    for cmd in ([PRODUCT_CLI, "set", "-q", "-o", uq, outs["product"]], [PRODUCT_CLI, "set", "-i", uq, "-o", db, outs["product"]]):
        r = subprocess.run(cmd, input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (cmd, r.stderr.decode())


# ---- composite -q: marker database by set -g / -q / -i, then the abundance report ------------------
def run_composite_case(case, shuf_files, tmp_path, dist_cmd, set_cmd, comp_cmd):
    entry = MANIFEST["composite_cases"][case]
    refs, qry = gc.build_composite_inputs(case, str(tmp_path))
    sk, grp, uq, db, qsk = (str(tmp_path / n) for n in ("sk", "grp", "uq", "db", "qsk"))
    taxf = str(tmp_path / "tax.tsv")
    open(taxf, "w").write("".join(t + "\n" for t in entry["tax"]))
    for cmd in (dist_cmd + ["-L", shuf_files(entry["shuf"]), "-o", sk] + refs,
                set_cmd + ["-g", taxf, "-o", grp, sk], set_cmd + ["-q", "-o", uq, grp], set_cmd + ["-i", uq, "-o", db, grp],
                dist_cmd + ["-L", shuf_files(entry["shuf"]), "-A", "-o", qsk] + qry):
        r = subprocess.run(cmd, input=b"N\n", stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, (cmd, r.stderr.decode())
    exp = os.path.join(gc.GOLDEN, "expected", case)
    r = subprocess.run(comp_cmd + ["-r", db, "-q", qsk], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    got = []
    for ln in r.stdout.decode().splitlines():
        f = ln.split("\t")
        f[0] = os.path.basename(f[0])
        got.append("\t".join(f))
    assert got == open(os.path.join(exp, "composite.tsv")).read().splitlines()
    assert len(got) == entry["lines"] and len(got) > 0
    abv = str(tmp_path / "abv")
    r = subprocess.run(comp_cmd + ["-r", db, "-q", qsk, "-b", "-o", abv], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    want_abv = sorted(f for f in golden_listing("expected/" + case) if f.endswith(".abv"))
    assert sorted(os.listdir(abv)) == want_abv
    for f in want_abv:
        assert filecmp.cmp(os.path.join(exp, f), os.path.join(abv, f), shallow=False), f


@pytest.mark.parametrize("case", sorted(MANIFEST["composite_cases"]))
def test_oracle_composite_reproduces_reference_golden(case, shuf_files, tmp_path):
    run_composite_case(case, shuf_files, tmp_path, [ORACLE_CLI], [ORACLE_CLI, "set"], [ORACLE_CLI, "composite"])


@pytest.mark.gpu
@pytest.mark.parametrize("case", sorted(MANIFEST["composite_cases"]))
def test_product_cli_composite_reproduces_reference_golden(case, shuf_files, tmp_path):
    """the README's MarkerDB recipe and the profiling step, every stage on the device build"""
    run_composite_case(case, shuf_files, tmp_path, [PRODUCT_CLI, "dist", "-p", "4"], [PRODUCT_CLI, "set"], [PRODUCT_CLI, "composite"])


# ---- stage II + `dist -r` search --------------------------------------------------------------
SEARCH_DBS = MANIFEST.get("search_dbs", {})
SEARCH_CASES = MANIFEST.get("search_cases", {})


def _run(cmd, cwd):
    r = subprocess.run(cmd, cwd=cwd, stdin=subprocess.DEVNULL, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, (cmd, r.stderr.decode(errors="replace")[-400:])
    return r


def make_search_db_sketch(db, shuf_files, tmp_path, dist_cmd):
    """the database's sketch directory `<db>.sk` inside tmp_path, names relative like in make_golden.py"""
    entry = SEARCH_DBS[db]
    refs = gc.build_search_inputs(db, entry["refs"], str(tmp_path))
    _run(dist_cmd + ["-L", shuf_files(entry["shuf"]), "-o", db + ".sk"] + refs, str(tmp_path))
    return db + ".sk"


def make_search_query_sketch(case, shuf_files, tmp_path, dist_cmd):
    entry = SEARCH_CASES[case]
    qry = gc.build_search_inputs(case, entry["query"], str(tmp_path))
    _run(dist_cmd + ["-L", shuf_files(SEARCH_DBS[entry["db"]]["shuf"])] + entry["qflags"] + ["-o", case + ".qsk"] + qry, str(tmp_path))
    return case + ".qsk"


def dense_index_xxh64(row_ids, row_ends):
    """hash of the 2^32-entry mco.index.N that the row table stands for, built in 2^26-row pieces"""
    import numpy as np
    import xxhash
    h = xxhash.xxh64()
    ends = np.concatenate([[0], np.asarray(row_ends, np.uint64)]).astype(np.uint64)
    rid = np.asarray(row_ids, np.uint64)
    step = 1 << 26
    buf = np.empty(step, np.uint64)                     # one reused 512 MiB piece: no per-piece allocation, no tobytes() copy
    for r0 in range(0, 1 << 32, step):
        lo, hi = np.searchsorted(rid, [r0, r0 + step], side="left")
        if lo == hi:
            buf.fill(ends[lo])
        else:
            buf[:] = ends[np.searchsorted(rid, np.arange(r0, r0 + step, dtype=np.uint64), side="right")]
        h.update(buf)
    return h.hexdigest()


@pytest.mark.parametrize("db", sorted(SEARCH_DBS))
def test_oracle_stage2_reproduces_reference_golden(db, shuf_files, tmp_path):
    sk = make_search_db_sketch(db, shuf_files, tmp_path, [ORACLE_CLI])
    _run([ORACLE_CLI, "stage2", "--no-index", "-o", db + ".mco", sk], str(tmp_path))
    exp = os.path.join(gc.GOLDEN, "expected", db)
    for f in ("mco.0", "mcofiles.stat"):
        assert filecmp.cmp(os.path.join(exp, f), str(tmp_path / (db + ".mco") / f), shallow=False), f
    if db == "db_strains_L3K10":   # the 32 GiB index once on the CPU: from the oracle's row table
        import numpy as np
        import oracle_binding as ob
        ids = np.fromfile(str(tmp_path / sk / "combco.0"), np.uint32)
        index = np.fromfile(str(tmp_path / sk / "combco.index.0"), np.uint64)
        _, ri, re_ = ob.mco_build(ids, index)
        assert dense_index_xxh64(ri, re_) == SEARCH_DBS[db]["index_xxh64"]


@pytest.mark.parametrize("case", sorted(SEARCH_CASES))
def test_oracle_search_reproduces_reference_golden(case, shuf_files, tmp_path):
    entry = SEARCH_CASES[case]
    sk = make_search_db_sketch(entry["db"], shuf_files, tmp_path, [ORACLE_CLI])
    qsk = make_search_query_sketch(case, shuf_files, tmp_path, [ORACLE_CLI])
    _run([ORACLE_CLI, "search", "--refco", "-r", sk, "-o", "out", "--keepskf"] + entry["flags"] + [qsk], str(tmp_path))
    exp = os.path.join(gc.GOLDEN, "expected", case)
    for f in ("distance.out", "sharedk_ct.dat"):
        assert filecmp.cmp(os.path.join(exp, f), str(tmp_path / "out" / f), shallow=False), f
    assert len(open(os.path.join(exp, "distance.out")).read().splitlines()) == entry["lines"] > 1


def _flags_to_kwargs(flags):
    kw, it = {}, iter(flags)
    for f in it:
        v = next(it)
        kw[{"-M": "metric", "-O": "outfields", "-N": "num_neigb", "-D": "dthreshold", "--correction": "correction"}[f]] = \
            float(v) if f == "-D" else int(v)
    return kw


@pytest.mark.gpu
@pytest.mark.parametrize("db", sorted(SEARCH_DBS))
def test_product_abi_stage2_search_reproduces_reference_golden(db, shuf_files, tmp_path):
    """sketch directories by the product CLI, inverted index / dense index / counts through the C ABI, text by mk_dist_print"""
    import numpy as np
    import xxhash
    from metakssd_amd import capi
    sk = make_search_db_sketch(db, shuf_files, tmp_path, [PRODUCT_CLI, "dist", "--quiet", "-p", "4"])
    exp = os.path.join(gc.GOLDEN, "expected", db)
    ids = np.fromfile(str(tmp_path / sk / "combco.0"), np.uint32)
    index = np.fromfile(str(tmp_path / sk / "combco.index.0"), np.uint64)
    rstat, rnames = parse_stat(str(tmp_path / sk / "cofiles.stat"))
    m = capi.Mco(0)
    gids, ri, re_ = m.build(ids, index)
    assert np.array_equal(gids, np.fromfile(os.path.join(exp, "mco.0"), np.uint32))
    h = xxhash.xxh64()
    for r0 in range(0, 1 << 32, 1 << 27):
        h.update(m.index_rows(r0, 1 << 27).tobytes())
    assert h.hexdigest() == SEARCH_DBS[db]["index_xxh64"]
    for case in sorted(c for c, e in SEARCH_CASES.items() if e["db"] == db):
        entry = SEARCH_CASES[case]
        qsk = make_search_query_sketch(case, shuf_files, tmp_path, [PRODUCT_CLI, "dist", "--quiet", "-p", "4"])
        qids = np.fromfile(str(tmp_path / qsk / "combco.0"), np.uint32)
        qindex = np.fromfile(str(tmp_path / qsk / "combco.index.0"), np.uint64)
        qstat, qnames = parse_stat(str(tmp_path / qsk / "cofiles.stat"))
        ct = m.count(rstat["infile_num"], qindex, qstat["ctx_ct"], [{"qry_ids": qids}])
        cexp = os.path.join(gc.GOLDEN, "expected", case)
        assert np.array_equal(ct.ravel(), np.fromfile(os.path.join(cexp, "sharedk_ct.dat"), np.uint32)), case
        out = str(tmp_path / (case + ".distance.out"))
        assert capi.dist_print(out, rstat["ctx_ct"], qstat["ctx_ct"], rnames, qnames, ct, qstat["kmerlen"], qstat["dim_rd_len"],
                               **_flags_to_kwargs(entry["flags"])) == 0
        assert filecmp.cmp(os.path.join(cexp, "distance.out"), out, shallow=False), case
    m.close()


@pytest.mark.gpu
def test_product_cli_stage2_search_end_to_end(shuf_files, tmp_path):
    """`metakssd dist -o <mco> <sketches>` (the 32 GiB mco.index.0 on disk) and `metakssd dist -r <mco> ... <query>` against the
    reference's files; skipped where the scratch disk cannot hold the index"""
    import shutil
    import xxhash
    db = "db_strains_L1K7"
    if shutil.disk_usage(str(tmp_path)).free < 48 << 30:
        pytest.skip("needs 32 GiB of scratch space for mco.index.0 (three times in a row)")
    dist = [PRODUCT_CLI, "dist", "--quiet", "-p", "4"]
    sk = make_search_db_sketch(db, shuf_files, tmp_path, dist)
    try:
        _run([PRODUCT_CLI, "dist", "--quiet", "-o", db + ".mco", sk], str(tmp_path))
        exp = os.path.join(gc.GOLDEN, "expected", db)
        for f in ("mco.0", "mcofiles.stat"):
            assert filecmp.cmp(os.path.join(exp, f), str(tmp_path / (db + ".mco") / f), shallow=False), f
        h = xxhash.xxh64()
        with open(str(tmp_path / (db + ".mco") / "mco.index.0"), "rb") as f:
            for b in iter(lambda: f.read(1 << 26), b""):
                h.update(b)
        assert h.hexdigest() == SEARCH_DBS[db]["index_xxh64"]
        for case in sorted(c for c, e in SEARCH_CASES.items() if e["db"] == db):
            entry = SEARCH_CASES[case]
            qsk = make_search_query_sketch(case, shuf_files, tmp_path, dist)
            _run([PRODUCT_CLI, "dist", "--quiet", "-r", db + ".mco", "-o", case + ".out", "--keepskf"] + entry["flags"] + [qsk], str(tmp_path))
            for f in ("distance.out", "sharedk_ct.dat"):
                assert filecmp.cmp(os.path.join(gc.GOLDEN, "expected", case, f), str(tmp_path / (case + ".out") / f), shallow=False), (case, f)
        # the same search with the query sketches sharded over two ranks (one device, gloo): rows gathered on rank 0
        import sys
        case = "search_ctm_n2_L1K7"
        tool = os.path.join(ROOT, "tools", "search_multi.py")
        for world in (1, 2):
            launch = [sys.executable] if world == 1 else [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2",
                                                          "--master-addr", "127.0.0.1", "--master-port", str(__import__("util_inputs").free_port())]
            extra = [] if world == 1 else ["--backend", "gloo", "--same-device"]
            r = subprocess.run(launch + [tool, "-r", db + ".mco", "-o", "multi%d" % world] + SEARCH_CASES[case]["flags"] + [case + ".qsk"] + extra,
                               cwd=str(tmp_path), env=dict(os.environ, MASTER_ADDR="127.0.0.1"), stdin=subprocess.DEVNULL,
                               stdout=subprocess.PIPE, stderr=subprocess.STDOUT, timeout=600)
            assert r.returncode == 0, r.stdout.decode(errors="replace")[-2000:]
            for f in ("distance.out", "sharedk_ct.dat"):
                assert filecmp.cmp(os.path.join(gc.GOLDEN, "expected", case, f), str(tmp_path / ("multi%d" % world) / f), shallow=False), (world, f)
        # without --keepskf the counts file is removed (command_dist.c:1633); a second run into the same directory works
        case = "search_default_L1K7"
        _run([PRODUCT_CLI, "dist", "--quiet", "-r", db + ".mco", "-o", "again", case + ".qsk"], str(tmp_path))
        assert sorted(os.listdir(str(tmp_path / "again"))) == ["distance.out"]
        # -f: print again from a kept counts file, other options; the file given with -f is removed like any other (:1633)
        case = "search_default_L1K7"
        shutil.copy(str(tmp_path / (case + ".out") / "sharedk_ct.dat"), str(tmp_path / "kept.dat"))
        _run([PRODUCT_CLI, "dist", "--quiet", "-r", db + ".mco", "-o", "fromskf", "-f", "kept.dat", "-M", "1", "-O", "1", "-N", "2",
              "search_ctm_n2_L1K7.qsk"], str(tmp_path))
        assert filecmp.cmp(os.path.join(gc.GOLDEN, "expected", "search_ctm_n2_L1K7", "distance.out"), str(tmp_path / "fromskf" / "distance.out"),
                           shallow=False)
        assert not os.path.exists(str(tmp_path / "kept.dat"))
        os.remove(str(tmp_path / (db + ".mco") / "mco.index.0"))
        # `-r <sketch directory>`: stage II goes next to the sketches (command_dist.c:119-122), then the search runs against it
        shutil.copytree(str(tmp_path / sk), str(tmp_path / "inplace.sk"))
        _run([PRODUCT_CLI, "dist", "--quiet", "-r", "inplace.sk", "-o", "inplace.out", "--keepskf", case + ".qsk"], str(tmp_path))
        assert filecmp.cmp(os.path.join(exp, "mco.0"), str(tmp_path / "inplace.sk" / "mco.0"), shallow=False)
        for f in ("distance.out", "sharedk_ct.dat"):
            assert filecmp.cmp(os.path.join(gc.GOLDEN, "expected", case, f), str(tmp_path / "inplace.out" / f), shallow=False), f
        os.remove(str(tmp_path / "inplace.sk" / "mco.index.0"))
        # `-L .. -r <genome files>`: stage I (without abundances) and stage II into the output directory (:68-114)
        os.makedirs(str(tmp_path / "genomes"))
        for i, f in enumerate(gc.build_search_inputs(db, SEARCH_DBS[db]["refs"], str(tmp_path))):
            shutil.copy(str(tmp_path / f), str(tmp_path / "genomes" / ("g%d.fa" % i)))
        _run([PRODUCT_CLI, "dist", "--quiet", "-L", shuf_files(SEARCH_DBS[db]["shuf"]), "-A", "-r", "genomes", "-o", "fromraw"], str(tmp_path))
        assert sorted(os.listdir(str(tmp_path / "fromraw"))) == ["cofiles.stat", "combco.0", "combco.index.0", "mco.0", "mco.index.0", "mcofiles.stat"]
        assert filecmp.cmp(os.path.join(exp, "mco.0"), str(tmp_path / "fromraw" / "mco.0"), shallow=False)
        os.remove(str(tmp_path / "fromraw" / "mco.index.0"))
    finally:
        for d in (db + ".mco", "inplace.sk", "fromraw"):
            try:
                os.remove(str(tmp_path / d / "mco.index.0"))
            except OSError:
                pass


@pytest.mark.gpu
@pytest.mark.parametrize("shuf,flags", [("L1K7", ["-A"]), ("L0K6", ["-A"]), ("L1K7", ["-n", "2"])])
def test_cli_several_engines_equal_one(shuf, flags, shuf_files, tmp_path):
    """`metakssd dist --devices a,b,..` (libmetakssd_multi.so): the FASTQ stream's row buffers are dealt round-robin to one
    engine per listed GPU, mk_multi_finish gathers the partial sketches on the first one (RCCL between distinct GPUs; on
    this one-GPU box the list names GPU 0 several times and the lists move with device copies), one import launch, finish.
    The sketch directory is byte-identical to the single-engine run -- SURVEY.md 8e's merge algebra through the C product.
    Both merges: the gather to engine 0 (default below four engines) and SURVEY.md 8e's key slices (all-to-all by key % n, every
    engine folds its slice, the reduced slices are engine 0's key list: default from four engines on; MK_MULTI_MERGE forces one)."""
    import numpy as np
    import util_inputs as ui
    rs = np.random.RandomState(5)
    seqs = ui.pool_reads(rs, 3000 if shuf == "L0K6" else 30000, 9000) + ui.ragged_reads(rs, 300)
    fq = str(tmp_path / "in.fq")
    open(fq, "wb").write(ui.fastq_bytes(seqs, quals=ui.random_quals(rs, seqs)))
    assert os.path.getsize(fq) > 2 << 20  # several 1 MiB chunks
    outs = []
    for devs, merge in ((None, None), ("0,0", None), ("0,0,0", None), ("0,0", "slices"), ("0,0,0", "slices"), ("0,0,0,0,0", None), ("0,0,0,0", "gather")):
        out = str(tmp_path / ("out_" + (devs or "single").replace(",", "_") + (merge or "")))
        cmd = [PRODUCT_CLI, "dist", "-L", shuf_files(shuf)] + flags + ["-p", "4", "--chunk-mib", "1", "--timing", "-o", out]
        if devs:
            cmd += ["--devices", devs]
        env = dict(os.environ)
        env.pop("MK_MULTI_MERGE", None)
        if merge:
            env["MK_MULTI_MERGE"] = merge
        r = subprocess.run(cmd + [fq], stdout=subprocess.PIPE, stderr=subprocess.PIPE, env=env)
        assert r.returncode == 0, r.stderr.decode()
        if devs:
            tm = [json.loads(ln)["timing"] for ln in r.stdout.decode().splitlines() if ln.startswith('{"timing"')][0]
            assert tm["gpus"] == len(devs.split(",")) and tm["transport"] == "device copies"
            want_merge = merge or ("slices" if len(devs.split(",")) >= 4 else "gather")
            assert ("key slices" in r.stderr.decode()) == (want_merge == "slices"), r.stderr.decode()
        outs.append(out)
    names = sorted(f for f in os.listdir(outs[0]) if f.startswith("combco"))
    assert names and os.path.getsize(os.path.join(outs[0], "combco.0")) > 4000
    for other in outs[1:]:
        assert sorted(f for f in os.listdir(other) if f.startswith("combco")) == names
        for f in names:
            assert filecmp.cmp(os.path.join(outs[0], f), os.path.join(other, f), shallow=False), (other, f)
        assert parse_stat(os.path.join(outs[0], "cofiles.stat")) == parse_stat(os.path.join(other, "cofiles.stat"))


@pytest.mark.gpu
def test_cli_devices_fail_loudly(shuf_files, tmp_path):
    """a device list the node cannot serve must end the run with a message, not fall back to something else: a GPU that does
    not exist (one FASTQ: the merged-sketch path; several files: the file-sharded path), and more engines than the count
    field of a table slot admits imports (mk_multi.hip: 16)"""
    import util_inputs as ui
    import numpy as np
    rs = np.random.RandomState(6)
    fq = str(tmp_path / "in.fq")
    open(fq, "wb").write(ui.fastq_bytes(ui.pool_reads(rs, 3000, 500)))
    fa = []
    for i in range(3):
        fa.append(str(tmp_path / ("g%d.fna" % i)))
        open(fa[-1], "wb").write(ui.fasta_bytes([ui.rand_seq(rs, 3000)]))
    base = [PRODUCT_CLI, "dist", "-L", shuf_files("L1K7")]
    for extra, inputs in ((["-A", "--devices", "0,63"], [fq]), (["--devices", "0,63"], fa), (["-A", "--devices", ",".join(["0"] * 17)], [fq])):
        r = subprocess.run(base + extra + ["-o", str(tmp_path / "out")] + inputs, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode != 0, (extra, r.stdout.decode(), r.stderr.decode())
        assert b"metakssd:" in r.stderr or b"failed" in r.stderr, r.stderr.decode()


@pytest.mark.gpu
@pytest.mark.parametrize("shuf,flags", [("L1K7", []), ("L3K10", ["-u"]), ("L2K11", []), ("L1K7", ["-A"])])
def test_cli_many_small_inputs_equal_one_by_one(shuf, flags, shuf_files, tmp_path):
    """many inputs in one run (worker threads prepare the files ahead, one engine sketches them in input order, its tables
    cleared -- sparsely at L2K11 -- in between): the directory (ids, index, counts, cofiles.stat) must be what sketching the
    files one by one and concatenating gives -- file order and empty sketches included."""
    import numpy as np
    import util_inputs as ui
    rs = np.random.RandomState(11)
    fastq = "-A" in flags
    paths = []
    for i in range(21):
        n = [60000, 500, 0, 12000][i % 4] if i != 7 else 150000
        if fastq:
            seqs = ui.pool_reads(rs, 20000, max(1, n // 150))
            data = ui.fastq_bytes(seqs)
            path = str(tmp_path / ("s%02d.fq" % i))
        else:
            data = ui.fasta_bytes([ui.rand_seq(rs, n // 2 + 30), ui.rand_seq(rs, n - n // 2 + 25)])
            path = str(tmp_path / ("g%02d.fna" % i))
        open(path, "wb").write(data)
        paths.append(path)
    base = [PRODUCT_CLI, "dist", "-L", shuf_files(shuf)] + flags
    whole = str(tmp_path / "whole")
    r = subprocess.run(base + ["-p", "6", "-o", whole] + paths, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
    assert r.returncode == 0, r.stderr.decode()
    names = sorted(f for f in os.listdir(whole) if f.startswith("combco") and ".index." not in f)
    assert names
    cat = {f: b"" for f in names}
    counts, per_comp = [], []
    for i, path in enumerate(paths):
        one = str(tmp_path / ("one%02d" % i))
        r = subprocess.run(base + ["-p", "1", "-o", one, path], stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        for f in names:
            cat[f] += open(os.path.join(one, f), "rb").read()
        per_comp.append(os.path.getsize(os.path.join(one, "combco.0")) // 4)
        counts.append(sum(os.path.getsize(os.path.join(one, f)) // 4 for f in names if not f.endswith(".a")))
    for f in names:
        assert open(os.path.join(whole, f), "rb").read() == cat[f], f
    idx = np.fromfile(os.path.join(whole, "combco.index.0"), dtype=np.uint64)
    assert list(np.diff(idx)) == per_comp and sum(counts) > 100
    st, stnames = parse_stat(os.path.join(whole, "cofiles.stat"))
    assert st["infile_num"] == len(paths) and stnames == paths and st["ctx_ct"] == counts and (fastq or counts[2] == 0)
    # the same inputs dealt to three engines file by file (--devices with several files: whole files are the unit, results
    # written in file order, command_dist.c:363-372), with one, two and four engines taking the files in turn on the one GPU
    # (the default above is two), and with the FASTA text windowed on the host
    variants = [("sharded", ["--devices", "0,0,0", "-p", "6"]), ("engines1", ["--engines", "1", "-p", "6"]), ("engines2", ["--engines", "2", "-p", "3"]),
                ("engines4", ["--engines", "4", "-p", "3"])]
    if not fastq:
        variants.append(("hostfasta", ["--host-fasta", "-p", "6"]))
    for tag, extra in variants:
        other = str(tmp_path / tag)
        r = subprocess.run(base + extra + ["-o", other] + paths, stdout=subprocess.PIPE, stderr=subprocess.PIPE)
        assert r.returncode == 0, r.stderr.decode()
        assert sorted(os.listdir(other)) == sorted(os.listdir(whole)), tag
        for f in sorted(os.listdir(whole)):
            if f == "cofiles.stat":
                assert parse_stat(os.path.join(other, f)) == parse_stat(os.path.join(whole, f)), tag
            else:
                assert filecmp.cmp(os.path.join(whole, f), os.path.join(other, f), shallow=False), (tag, f)


# ---- sixteen components end to end (stage I, combine, stage II, search) with COMPONENT_SZ = 6 --------------------------------
CSZ6 = MANIFEST.get("csz6")


@pytest.mark.gpu
def test_product_cli_sixteen_component_database_end_to_end(shuf_files, tmp_path):
    """k - drlevel above COMPONENT_SZ splits every sketch into 16 component files (iseq2comem.c:64-65, 543-549) and stage II
    and the search loop over them (co2mco.c:25, command_dist.c:990-1056).  With the real COMPONENT_SZ = 8 a 16-component
    database needs 16 x 32 GiB of index; the reference built with -DCOMPONENT_SZ=6 (oracle/_ref/metakssd_csz6) needs 16 x
    128 MiB, the product takes the same constant as --component-sz 6.
      stage I + combine_queries: byte-identical to that reference build's directories (tests/golden/expected/csz6_db);
      stage II + search: that reference build aborts in combco2mco() on the second component (double free, recorded in the
      manifest), so the product's files are checked component by component against the oracle's restatement of the loops."""
    import numpy as np
    import oracle_binding as ob
    assert CSZ6 and CSZ6["reference_stage2_aborts"]
    shuf = shuf_files(CSZ6["shuf"])
    dist = [PRODUCT_CLI, "dist", "--quiet", "-p", "4", "--component-sz", "6"]
    exp = os.path.join(gc.GOLDEN, "expected", "csz6_db")

    def sketch(tag, specs, want_files):
        files = gc.build_search_inputs("csz6_" + tag, specs, str(tmp_path))
        assert files == want_files
        dirs = []
        for i, f in enumerate(files):  # one directory per input and then combine_queries: the way the golden directory was made
            _run(dist + ["-L", shuf, "-o", "%s_%d.sk" % (tag, i), f], str(tmp_path))
            dirs.append("%s_%d.sk" % (tag, i))
        _run([PRODUCT_CLI, "dist", "-o", tag + ".sk"] + dirs, str(tmp_path))
        got = sorted(f for f in os.listdir(str(tmp_path / (tag + ".sk"))) if f.startswith("combco"))
        want = golden_listing("expected/csz6_db/" + tag + "_sk")
        assert got == want and len(got) == 32
        for f in want:
            assert filecmp.cmp(os.path.join(exp, tag + "_sk", f), str(tmp_path / (tag + ".sk") / f), shallow=False), (tag, f)
        stat, names = parse_stat(str(tmp_path / (tag + ".sk") / "cofiles.stat"))
        assert stat == CSZ6[tag + "_sk_stat"] and names == files
        # all inputs in one stage-I run: the same directory again (our file order is the argument order)
        _run(dist + ["-L", shuf, "-o", tag + "_oneshot.sk"] + files, str(tmp_path))
        for f in want:
            assert filecmp.cmp(os.path.join(exp, tag + "_sk", f), str(tmp_path / (tag + "_oneshot.sk") / f), shallow=False), (tag, f)
        return stat, names

    rstat, rnames = sketch("ref", gc.CSZ6["refs"], CSZ6["ref_files"])
    qstat, qnames = sketch("qry", gc.CSZ6["query"], CSZ6["qry_files"])
    _run([PRODUCT_CLI, "dist", "--quiet", "--component-sz", "6", "-o", "db.mco", "ref.sk"], str(tmp_path))
    rows = 1 << 24
    want_ct = np.zeros((qstat["infile_num"], rstat["infile_num"]), np.uint32)
    for c in range(16):
        ids = np.fromfile(str(tmp_path / "ref.sk" / ("combco.%d" % c)), np.uint32)
        index = np.fromfile(str(tmp_path / "ref.sk" / ("combco.index.%d" % c)), np.uint64)
        og, ori, ore = ob.mco_build(ids, index)
        assert np.array_equal(np.fromfile(str(tmp_path / "db.mco" / ("mco.%d" % c)), np.uint32), og), c
        dense = np.concatenate([[0], ore]).astype(np.uint64)[np.searchsorted(ori.astype(np.uint64), np.arange(rows, dtype=np.uint64), side="right")]
        got_index = np.fromfile(str(tmp_path / "db.mco" / ("mco.index.%d" % c)), np.uint64)
        assert got_index.size == rows and np.array_equal(got_index, dense), c
        qids = np.fromfile(str(tmp_path / "qry.sk" / ("combco.%d" % c)), np.uint32)
        qindex = np.fromfile(str(tmp_path / "qry.sk" / ("combco.index.%d" % c)), np.uint64)
        want_ct += ob.mco_count(og, ori, ore, qids, qindex, qstat["ctx_ct"], rstat["infile_num"])
    assert want_ct.sum() > 1000
    for i, flags in enumerate(gc.CSZ6["search_flags"]):
        out = "search%d.out" % i
        _run([PRODUCT_CLI, "dist", "--quiet", "--component-sz", "6", "-r", "db.mco", "-o", out, "--keepskf"] + flags + ["qry.sk"], str(tmp_path))
        assert np.array_equal(np.fromfile(str(tmp_path / out / "sharedk_ct.dat"), np.uint32), want_ct.ravel()), flags
        want_txt = str(tmp_path / ("want%d.txt" % i))
        assert ob.dist_print(want_txt, rstat["ctx_ct"], qstat["ctx_ct"], rnames, qnames, want_ct, qstat["kmerlen"], qstat["dim_rd_len"],
                             **_flags_to_kwargs(flags)) == 0
        assert filecmp.cmp(want_txt, str(tmp_path / out / "distance.out"), shallow=False), flags
