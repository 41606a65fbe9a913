"""deterministic test inputs (numpy RandomState streams are stable across numpy versions)"""
import numpy as np

ACGT = np.frombuffer(b"ACGT", dtype=np.uint8)
_COMP = np.zeros(256, np.uint8)
for a, b in zip(b"ACGTacgtN", b"TGCAtgcaN"):
    _COMP[a] = b


def rand_seq(rs, n):
    return ACGT[rs.randint(0, 4, size=n)].tobytes()


def revcomp(s):
    return _COMP[np.frombuffer(s, np.uint8)][::-1].tobytes()


def rows_from_seqs(seqs, stride):
    rows = np.zeros(len(seqs) * stride, dtype=np.uint8)
    for i, s in enumerate(seqs):
        assert len(s) + 1 <= stride
        rows[i * stride: i * stride + len(s)] = np.frombuffer(s, np.uint8)
        rows[i * stride + len(s)] = 10
    return rows


def fastq_bytes(seqs, crlf=False, final_newline=True, drop_last_qual=False, quals=None):
    nl = b"\r\n" if crlf else b"\n"
    out = []
    for i, s in enumerate(seqs):
        rec = [b"@r%d" % i, s, b"+", quals[i] if quals is not None else b"I" * len(s)]
        if drop_last_qual and i == len(seqs) - 1:
            rec = rec[:3]
        out.append(nl.join(rec) + nl)
    txt = b"".join(out)
    if not final_newline and txt:
        txt = txt[: -len(nl)]
    return txt


def fasta_bytes(contigs, width=70):
    out = []
    for i, c in enumerate(contigs):
        out.append(b">contig_%d some description\n" % i)
        for j in range(0, len(c), width):
            out.append(c[j:j + width] + b"\n")
    return b"".join(out)


def pool_reads(rs, pool_len, nreads, read_len=150, p_rc=0.5, p_n=0.05, p_lower=0.1):
    pool = rand_seq(rs, pool_len)
    seqs = []
    for _ in range(nreads):
        a = rs.randint(0, pool_len - read_len)
        s = pool[a:a + read_len]
        if rs.rand() < p_rc:
            s = revcomp(s)
        if rs.rand() < p_n:
            j = rs.randint(0, read_len)
            s = s[:j] + b"N" + s[j + 1:]
        if rs.rand() < p_lower:
            s = s.lower()
        seqs.append(s)
    return seqs


def ragged_reads(rs, nreads, lens=(0, 1, 21, 22, 23, 50, 100, 150, 151, 250, 300)):
    seqs = []
    for _ in range(nreads):
        L = int(lens[rs.randint(0, len(lens))])
        s = rand_seq(rs, L)
        if rs.rand() < 0.3:
            s = s.lower()
        if L > 30 and rs.rand() < 0.3:
            j = rs.randint(0, L - 5)
            s = s[:j] + b"NNN" + s[j + 3:]
        seqs.append(s)
    return seqs


def random_quals(rs, seqs, p_low=0.08, low=b"#+5?", high=b"I"):
    """per-base quality bytes: mostly `high`, a fraction p_low drawn from `low`"""
    lo = np.frombuffer(low, np.uint8)
    out = []
    for s in seqs:
        q = np.full(len(s), high[0], np.uint8)
        m = rs.rand(len(s)) < p_low
        q[m] = lo[rs.randint(0, len(lo), size=int(m.sum()))]
        out.append(q.tobytes())
    return out


def free_port():
    """a TCP port nobody listens on right now (for torch.distributed rendezvous on 127.0.0.1): asked from the kernel, not derived from the pid --
    several suites at a time met each other's pid-derived ports (round 6)"""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p
