"""the golden cases: which .shuf, which flags, which input.  Shared by tests/golden/make_golden.py (runs the real
reference, in the build container only) and by the tests (oracle CLI on CPU, product CLI on the GPU)."""
import gzip
import os

import numpy as np

import util_inputs as ui

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

# name -> (k, subk, drlevel, seed); same as tests/conftest.py
SHUF_SPECS = {
    "L3K11": (11, 6, 3, 11), "L3K10": (10, 6, 3, 10), "L2K11": (11, 5, 2, 211),
    "L3K9": (9, 6, 3, 9), "L0K6": (6, 3, 0, 6), "L1K7": (7, 4, 1, 7),
    "L1K8": (8, 4, 1, 81),   # k - drlevel = 7: sixteen components under a -DCOMPONENT_SZ=6 build (the csz6 database)
}

# ---- a database with SIXTEEN components that fits a test: the reference built with -DCOMPONENT_SZ=6 (global_basic.h:35-37),
# whose stage II index has 16^6 rows (128 MiB) per component instead of 16^8 (32 GiB).  Stage I per genome, combine_queries,
# stage II and the search all by that build (oracle/_ref/metakssd_csz6); the product runs with --component-sz 6.
CSZ6 = {"shuf": "L1K8", "component_sz": 6, "refs": ["fa:sA", "fa:sB", "fa:sC", "fa:genome"], "query": ["fa:sB", "fq:mix", "fa:genome"],
        "search_flags": [["-N", "2"], ["-M", "1", "-O", "1"]]}


def make_shuf(name, path):
    """write the .shuf `name` with the product's seeded generator (L0K6z: L0K6 with substring 0 -> 0)"""
    from metakssd_amd import capi
    base = name[:-1] if name.endswith("z") else name
    k, subk, drl, seed = SHUF_SPECS[base]
    s = capi.Shuf.generate(k, subk, drl, seed)
    if name.endswith("z"):
        t = s.table
        j = int(np.nonzero(t == 0)[0][0])
        t[j], t[0] = t[0], 0
    s.write(path)


def _committed(name):
    return os.path.join(GOLDEN, "inputs", name)


def _seqs(kind):
    rs = np.random.RandomState({"dense": 101, "pool": 102, "ragged": 103, "homo": 0, "lowcov": 105, "qual": 106,
                                "long": 107, "mix": 109, "mixs": 109}[kind])
    if kind in ("lowcov", "qual"):   # ~3x coverage of a 60 kb pool, both strands: occurrence counts spread over 1..10
        return ui.pool_reads(rs, 60000, 1200, p_n=0.0 if kind == "qual" else 0.05)
    if kind == "mixs":               # a small version for the accept-everything table (131 071 slots)
        return _seqs("mix")[:400]
    if kind == "mix":                # metagenome-like reads: 70 % from strain sA, 30 % from strain sC (composite cases)
        ga, gc_ = b"".join(_strain("sA")), b"".join(_strain("sC"))
        out = []
        for _ in range(4000):
            g = ga if rs.rand() < 0.7 else gc_
            a = rs.randint(0, len(g) - 150)
            r = g[a:a + 150]
            out.append(ui.revcomp(r) if rs.rand() < 0.5 else r)
        return out
    if kind == "long":               # fastq2co()'s fgets width is 20000: reads beyond the 4096-byte row limit
        return [ui.rand_seq(rs, L) for L in (4094, 4095, 4096, 5000, 8191, 12000, 19997, 150, 0, 7000)]
    if kind == "dense":
        return [ui.rand_seq(rs, 150) for _ in range(150)]
    if kind == "pool":
        return ui.pool_reads(rs, 8000, 2000)
    if kind == "ragged":
        return ui.ragged_reads(rs, 500)
    return [b"A" * 150, b"C" * 150, b"G" * 150, b"T" * 150, b"AC" * 75, b"ACGT" * 37] * 20


def _genome():
    rs = np.random.RandomState(104)
    g = ui.rand_seq(rs, 30000)
    return [g[:20000], g[5000:12000], b"A" * 100 + g[500:900] + b"T" * 50, g[20000:20021], b"", g[20021:]]


def _strain(name):
    """three overlapping 'strains' cut from one seeded 90 kb sequence: shared, pairwise-shared and private k-mers"""
    g = ui.rand_seq(np.random.RandomState(108), 90000)
    a, b = {"sA": (0, 50000), "sB": (20000, 70000), "sC": (40000, 90000)}[name]
    return [g[a:(a + b) // 2], g[(a + b) // 2:b]]


# `set -u` / `set -q`: the sketch directory is made by `dist` from `inputs`, then the set operation
SET_CASES = {
    "set_u_strains_L1K7": {"shuf": "L1K7", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-u"},
    "set_q_strains_L1K7": {"shuf": "L1K7", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-q"},
    "set_u_strains_L0K6": {"shuf": "L0K6", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-u"},
    "set_q_strains_L0K6": {"shuf": "L0K6", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-q"},
    "set_u_strains_L2K11": {"shuf": "L2K11", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-u"},  # 16 components
    "set_q_strains_L2K11": {"shuf": "L2K11", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-q"},
    "set_u_reads_A_L1K7": {"shuf": "L1K7", "flags": ["-A"], "inputs": ["fq:pool", "fq:lowcov", "fq:ragged"], "op": "-u"},
    "set_q_reads_L0K6": {"shuf": "L0K6", "flags": ["-n", "2"], "inputs": ["fq:lowcov", "fq:qual"], "op": "-q"},
    "set_u_single_N_L1K7": {"shuf": "L1K7", "flags": [], "inputs": ["fa:sA"], "op": "-u"},  # one sketch: prompt answered N
    # `set -i <pan>` / `set -s <pan>` (sketch_operate): the pan directory is made by dist + set from "pan"
    "set_i_markers_L1K7": {"shuf": "L1K7", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-i",
                           "pan": {"flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-q"}},  # README.md:98-103
    "set_s_private_L1K7": {"shuf": "L1K7", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-s",
                           "pan": {"flags": [], "inputs": ["fa:sA", "fa:sB"], "op": "-u"}},
    "set_i_dense_L0K6": {"shuf": "L0K6", "flags": [], "inputs": ["fa:sC", "fa:sA", "fa:sB"], "op": "-i",
                         "pan": {"flags": [], "inputs": ["fa:sB", "fa:sC"], "op": "-q"}},
    "set_s_reads_A_L1K7": {"shuf": "L1K7", "flags": ["-A"], "inputs": ["fq:pool", "fq:lowcov"], "op": "-s",
                           "pan": {"flags": ["-A"], "inputs": ["fq:ragged", "fq:pool"], "op": "-u"}},  # koc stays 1, no .a written
    "set_i_strains_L2K11": {"shuf": "L2K11", "flags": [], "inputs": ["fa:sA", "fa:sB", "fa:sC"], "op": "-i",
                            "pan": {"flags": [], "inputs": ["fa:sB", "fa:sC"], "op": "-u"}},  # 16 components
    # `set -g <tax.tsv>` (grouping_genomes): line i classifies sketch i, so the sketch directory is made in the GIVEN
    # order (the reference's dist permutes its inputs with a time seed): make_golden builds it with the pinned oracle
    # CLI and runs the reference's `set -g` on it.  taxid 0 = left out; one taxon without a name.
    "set_g_taxa_L1K7": {"shuf": "L1K7", "flags": [], "op": "-g",
                        "inputs": ["fa:sA", "fa:sB", "fa:genome", "fa:sC", "fa:sB", "fa:sA"],
                        "tax": ["562\tEscherichia coli", "28901\tSalmonella enterica", "0\tunclassified", "562\tEscherichia coli",
                                "1280", "28901\tSalmonella enterica"]},
    "set_g_taxa_L0K6": {"shuf": "L0K6", "flags": [], "op": "-g",
                        "inputs": ["fa:sC", "fa:sA", "fa:sB", "fa:genome"],
                        "tax": ["9\tnine", "7", "9\tnine", "131567\tcellular organisms"]},
    "set_g_reads_A_L1K7": {"shuf": "L1K7", "flags": ["-A"], "op": "-g", "inputs": ["fq:pool", "fq:lowcov", "fq:ragged"],
                           "tax": ["3\tthree", "3\tthree", "4\tfour"]},  # koc becomes 0 in the grouped directory
    "set_g_taxa_L2K11": {"shuf": "L2K11", "flags": [], "op": "-g", "inputs": ["fa:sA", "fa:sB", "fa:sC"],
                         "tax": ["11\televen", "12\ttwelve", "11\televen"]},  # 16 components
}


# `composite -r <markers> -q <qry>`.  The marker database is built the way the reference's README does
# (:80-104): dist (sketch directory in the given order, laid out by the pinned oracle CLI) -> set -g tax -> set -q -> set -i;
# the query is the -A sketch of "query".  All set steps and composite itself are run by the reference in make_golden.
COMPOSITE_CASES = {
    "composite_mix_L1K7": {"shuf": "L1K7", "refs": ["fa:sA", "fa:sB", "fa:sC", "fa:genome"],
                           "tax": ["1\tstrain A", "2\tstrain B", "3\tstrain C", "4"], "query": ["fq:mix"]},
    "composite_two_queries_L1K7": {"shuf": "L1K7", "refs": ["fa:sA", "fa:sB", "fa:sC"],
                                   "tax": ["10\tten", "10\tten", "30\tthirty"], "query": ["fq:mix", "fq:lowcov", "fq:mixs"]},
    "composite_small_L0K6": {"shuf": "L0K6", "refs": ["fa:sA", "fa:sC"], "tax": ["5\tfive", "6"], "query": ["fq:mixs"]},
    "composite_mix_L2K11": {"shuf": "L2K11", "refs": ["fa:sA", "fa:sB", "fa:sC"],
                            "tax": ["1\tA", "2\tB", "3\tC"], "query": ["fq:mix"]},  # 16 components
}


# stage II + `dist -r` search.  Only single-component tables: the reference writes a 32 GiB mco.index.N per
# component.  A database = `dist -o <mco> <sketch dir>` of the reference on a sketch directory laid out in the given order;
# a case = the reference's `dist -r <mco> -o out <flags> --keepskf <query sketch dir>`.  Names inside the stat files are the
# paths given on the command line: everything runs inside the work directory with relative names.
SEARCH_DBS = {
    "db_strains_L1K7": {"shuf": "L1K7", "refs": ["fa:sA", "fa:sB", "fa:sC", "fa:genome"]},
    "db_small_L0K6": {"shuf": "L0K6", "refs": ["fa:sA", "fa:sC"]},
    "db_strains_L3K10": {"shuf": "L3K10", "refs": ["fa:sA", "fa:sB", "fa:sC", "fa:genome"]},
}
SEARCH_CASES = {
    "search_default_L1K7": {"db": "db_strains_L1K7", "qflags": [], "query": ["fa:sB", "fq:mix", "fa:genome"], "flags": []},
    "search_ctm_n2_L1K7": {"db": "db_strains_L1K7", "qflags": [], "query": ["fa:sB", "fq:mix", "fa:genome"], "flags": ["-M", "1", "-O", "1", "-N", "2"]},
    "search_corr_d_L1K7": {"db": "db_strains_L1K7", "qflags": [], "query": ["fa:sB", "fq:mix", "fa:genome"],
                           "flags": ["--correction", "1", "-D", "0.08", "-O", "0"]},
    "search_koc_qry_L1K7": {"db": "db_strains_L1K7", "qflags": ["-A"], "query": ["fq:mix", "fq:lowcov"], "flags": ["-N", "1"]},
    "search_small_L0K6": {"db": "db_small_L0K6", "qflags": [], "query": ["fq:mixs", "fa:sC"], "flags": []},
    "search_ctm_L3K10": {"db": "db_strains_L3K10", "qflags": [], "query": ["fa:sB", "fa:genome", "fq:pool"], "flags": ["-M", "1"]},
}


def build_search_inputs(name, specs, workdir, write_committed=False):
    """input files for a search database / query, returned as names relative to workdir"""
    return [os.path.basename(build_input("%s_%d" % (name, i), workdir, write_committed=write_committed, spec=sp)) for i, sp in enumerate(specs)]


CASES = {
    # BASELINE.json config 1/2: 100 k synthetic 150 bp reads, L3K11 -A (input regenerated from the formula)
    "syn100k_L3K11": {"shuf": "L3K11", "flags": ["-A"], "input": "synth:seed=1,first=0,n=100000,len=150"},
    "syn20k_L3K9": {"shuf": "L3K9", "flags": ["-A"], "input": "synth:seed=2,first=0,n=20000,len=150"},
    # accept-everything table (131 071 slots): collision order decides the bytes
    "dense150_L0K6": {"shuf": "L0K6", "flags": ["-A"], "input": "fq:dense"},
    "pool2000_L0K6": {"shuf": "L0K6", "flags": ["-A"], "input": "fq:pool"},
    "pool2000_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:pool"},
    "pool2000_L2K11": {"shuf": "L2K11", "flags": ["-A"], "input": "fq:pool"},  # 16 components
    "ragged500_L0K6": {"shuf": "L0K6", "flags": ["-A"], "input": "fq:ragged"},
    "ragged500_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:ragged"},
    "ragged500_crlf_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:ragged:crlf"},
    "ragged500_trunc_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:ragged:trunc"},
    "ragged500_nonl_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:ragged:nonl"},
    "ragged500_gz_L1K7": {"shuf": "L1K7", "flags": ["-A"], "input": "fq:ragged:gz"},
    "key0_L0K6z": {"shuf": "L0K6z", "flags": ["-A"], "input": "fq:homo"},
    "saturate_L0K6": {"shuf": "L0K6", "flags": ["-A"], "input": "repeat:seed=9,len=150,times=70000"},
    "crowded_L0K6": {"shuf": "L0K6", "flags": ["-A"], "input": "synth:seed=5,first=0,n=5000,len=150"},  # reference aborts
    # FASTQ without -A: fastq2co + write_fqco2file (-n minimum occurrence, -Q minimum quality byte)
    "syn100k_set_L3K11": {"shuf": "L3K11", "flags": [], "input": "synth:seed=1,first=0,n=100000,len=150"},
    "lowcov_n1_L0K6": {"shuf": "L0K6", "flags": [], "input": "fq:lowcov"},
    "lowcov_n2_L0K6": {"shuf": "L0K6", "flags": ["-n", "2"], "input": "fq:lowcov"},
    "lowcov_n4_L1K7": {"shuf": "L1K7", "flags": ["-n", "4"], "input": "fq:lowcov"},
    "lowcov_n7_L0K6": {"shuf": "L0K6", "flags": ["-n", "7"], "input": "fq:lowcov"},
    "lowcov_n9_L1K7": {"shuf": "L1K7", "flags": ["-n", "9"], "input": "fq:lowcov"},  # clamps to 7
    "lowcov_n3_L2K11": {"shuf": "L2K11", "flags": ["-n", "3"], "input": "fq:lowcov"},  # 16 components
    "ragged500_set_L1K7": {"shuf": "L1K7", "flags": [], "input": "fq:ragged"},
    "ragged500_crlf_set_L1K7": {"shuf": "L1K7", "flags": [], "input": "fq:ragged:crlf"},
    "ragged500_trunc_set_L1K7": {"shuf": "L1K7", "flags": [], "input": "fq:ragged:trunc"},
    "ragged500_nonl_set_L1K7": {"shuf": "L1K7", "flags": [], "input": "fq:ragged:nonl"},  # last record not walked
    "key0_n2_L0K6z": {"shuf": "L0K6z", "flags": ["-n", "2"], "input": "fq:homo"},
    "qual_Q0_L1K7": {"shuf": "L1K7", "flags": ["-Q", "0"], "input": "fq:qual"},
    "qual_Q54_L1K7": {"shuf": "L1K7", "flags": ["-Q", "54"], "input": "fq:qual"},
    "qual_Q54_n2_L0K6": {"shuf": "L0K6", "flags": ["-Q", "54", "-n", "2"], "input": "fq:qual"},
    "qual_Q73_L0K6": {"shuf": "L0K6", "flags": ["-Q", "73"], "input": "fq:qual"},
    "qual_Q74_L1K7": {"shuf": "L1K7", "flags": ["-Q", "74"], "input": "fq:qual"},  # nothing passes
    "long_set_L1K7": {"shuf": "L1K7", "flags": [], "input": "fq:long"},
    "long_n2_L0K6": {"shuf": "L0K6", "flags": ["-n", "2"], "input": "fq:long"},
    # 97 k distinct keys in 131 071 slots, above hashlimit 78 642: fastq2co never advances its key counter, no abort
    "dense700_set_L0K6": {"shuf": "L0K6", "flags": [], "input": "synth:seed=6,first=0,n=700,len=150"},
    # FASTA (config 5 family)
    "fasta_L0K6": {"shuf": "L0K6", "flags": [], "input": "fa:genome"},
    "fasta_uniq_L0K6": {"shuf": "L0K6", "flags": ["-u"], "input": "fa:genome"},
    "fasta_L0K6z": {"shuf": "L0K6z", "flags": [], "input": "fa:genome"},
    "fasta_L1K7": {"shuf": "L1K7", "flags": [], "input": "fa:genome"},
    "fasta_uniq_L1K7": {"shuf": "L1K7", "flags": ["-u"], "input": "fa:genome"},
    "fasta_L3K10": {"shuf": "L3K10", "flags": [], "input": "fa:genome"},
    "fasta_L2K11": {"shuf": "L2K11", "flags": [], "input": "fa:genome"},
    "fasta_uniq_L2K11": {"shuf": "L2K11", "flags": ["-u"], "input": "fa:genome"},
}


def build_input(case, workdir, write_committed=False, spec=None):
    """materialise the input file of `case` in workdir; returns its path.  Committed inputs live gzip'ed under
    tests/golden/inputs (written only by make_golden.py).  `spec` overrides the case's own input (set cases name
    their files <case>)."""
    from metakssd_amd import capi
    spec = spec or CASES[case]["input"]
    kind, _, rest = spec.partition(":")
    if kind == "synth":
        kv = dict(x.split("=") for x in rest.split(","))
        path = os.path.join(workdir, case + ".fq")
        rc = capi.lib.mk_synth_fastq_write(path.encode(), int(kv["seed"]), int(kv["first"]), int(kv["n"]), int(kv["len"]))
        assert rc == 0
        return path
    if kind == "repeat":
        kv = dict(x.split("=") for x in rest.split(","))
        rs = np.random.RandomState(int(kv["seed"]))
        one = ui.rand_seq(rs, int(kv["len"]))
        path = os.path.join(workdir, case + ".fq")
        rec = b"@r\n" + one + b"\n+\n" + b"I" * len(one) + b"\n"
        with open(path, "wb") as f:
            f.write(rec * int(kv["times"]))
        return path
    if kind == "fq":
        parts = rest.split(":")
        base, variant = parts[0], (parts[1] if len(parts) > 1 else "")
        stored = _committed("fq_%s%s.fq.gz" % (base, "_" + variant if variant in ("crlf", "trunc", "nonl") else ""))
        if write_committed:
            seqs = _seqs(base)
            quals = ui.random_quals(np.random.RandomState(206), seqs) if base == "qual" else None
            data = ui.fastq_bytes(seqs, crlf=variant == "crlf", final_newline=variant != "nonl",
                                  drop_last_qual=variant == "trunc", quals=quals)
            with gzip.GzipFile(stored, "wb", mtime=0) as f:
                f.write(data)
        data = gzip.open(stored, "rb").read()
        if variant == "gz":
            path = os.path.join(workdir, case + ".fq.gz")
            with gzip.GzipFile(path, "wb", mtime=0) as f:
                f.write(data)
            return path
        path = os.path.join(workdir, case + ".fq")
        open(path, "wb").write(data)
        return path
    if kind == "fa":
        stored = _committed("fa_%s.fa.gz" % rest)
        if write_committed:
            with gzip.GzipFile(stored, "wb", mtime=0) as f:
                f.write(ui.fasta_bytes(_genome() if rest == "genome" else _strain(rest)))
        path = os.path.join(workdir, case + ".fa")
        open(path, "wb").write(gzip.open(stored, "rb").read())
        return path
    raise ValueError(spec)


def build_set_inputs(case, workdir, write_committed=False, pan=False):
    """input files of a SET_CASES entry (or of its "pan" part), in order"""
    out = []
    specs = SET_CASES[case]["pan"]["inputs"] if pan else SET_CASES[case]["inputs"]
    for i, spec in enumerate(specs):
        out.append(build_input("%s_%s%d" % (case, "pan" if pan else "in", i), workdir, write_committed=write_committed, spec=spec))
    return out


def build_composite_inputs(case, workdir, write_committed=False):
    c = COMPOSITE_CASES[case]
    refs = [build_input("%s_ref%d" % (case, i), workdir, write_committed=write_committed, spec=sp) for i, sp in enumerate(c["refs"])]
    qry = [build_input("%s_qry%d" % (case, i), workdir, write_committed=write_committed, spec=sp) for i, sp in enumerate(c["query"])]
    return refs, qry
