"""ctypes binding of oracle/libkssd_oracle.so -- TEST INFRASTRUCTURE ONLY (the parity checker)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "libkssd_oracle.so")


class KoParams(C.Structure):
    _fields_ = [("shuf_id", C.c_int), ("k", C.c_int), ("subk", C.c_int), ("drlevel", C.c_int),
                ("half_outctx_len", C.c_int), ("TL", C.c_int), ("crvsaddmove", C.c_int),
                ("component_num", C.c_int), ("comp_code_bits", C.c_int), ("dim_start", C.c_int), ("dim_end", C.c_int),
                ("hashsize", C.c_uint), ("hashlimit", C.c_uint),
                ("tupmask", C.c_ulonglong), ("domask", C.c_ulonglong), ("undomask", C.c_ulonglong)]


def load():
    if not os.path.exists(LIB):
        subprocess.check_call(["make", "-s", "-C", ORACLE_DIR, "libkssd_oracle.so"])
    lib = C.CDLL(LIB)
    vp = C.c_void_p
    lib.ko_params_derive.argtypes = [C.c_int] * 4 + [C.POINTER(KoParams)]
    lib.ko_koc_from_fastq_bytes.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, vp, C.POINTER(C.c_ulonglong)]
    lib.ko_koc_from_rows.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, C.c_size_t, vp, C.c_int, C.POINTER(C.c_uint)]
    lib.ko_koc_from_rows_omp.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, C.c_size_t, vp, C.c_int, C.c_int]
    lib.ko_dump_koc.argtypes = [C.POINTER(KoParams), vp, vp, vp, vp]
    lib.ko_dump_koc.restype = C.c_uint
    lib.ko_co_from_fastq_bytes.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, C.c_int, C.c_int, vp,
                                           C.POINTER(C.c_ulonglong)]
    lib.ko_dump_fqco.argtypes = [C.POINTER(KoParams), vp, vp, vp]
    lib.ko_dump_fqco.restype = C.c_uint
    lib.ko_co_from_fasta_bytes.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, vp, C.c_int]
    lib.ko_dump_co.argtypes = [C.POINTER(KoParams), vp, vp, vp]
    lib.ko_dump_co.restype = C.c_uint
    lib.ko_partial_from_rows.argtypes = [C.POINTER(KoParams), vp, vp, C.c_size_t, C.c_size_t, C.c_ulonglong, vp, vp, vp,
                                         C.c_size_t, C.POINTER(C.c_size_t)]
    lib.ko_layout_from_partials.argtypes = [C.POINTER(KoParams), C.c_int, vp, vp, vp, vp, vp]
    return lib


class Oracle:
    """the reference algorithm at -p 1 for one .shuf (table given as numpy int32)"""

    def __init__(self, shuf_id, k, subk, drlevel, table):
        self.lib = load()
        self.P = KoParams()
        rc = self.lib.ko_params_derive(shuf_id, k, subk, drlevel, C.byref(self.P))
        if rc:
            raise ValueError("ko_params_derive rc=%d" % rc)
        self.table = np.ascontiguousarray(table, dtype=np.int32)
        self.co = np.zeros(self.P.hashsize, dtype=np.uint64)
        self.keycount = C.c_uint(0)

    def _dump(self, koc):
        Cn = self.P.component_num
        nout = (C.c_size_t * Cn)()
        if koc:
            self.lib.ko_dump_koc(C.byref(self.P), self.co.ctypes.data, None, None, nout)
        else:
            self.lib.ko_dump_co(C.byref(self.P), self.co.ctypes.data, None, nout)
        ids = [np.zeros(nout[c], np.uint32) for c in range(Cn)]
        cnts = [np.zeros(nout[c], np.uint16) for c in range(Cn)]
        pi = (C.c_void_p * Cn)(*[a.ctypes.data for a in ids])
        pc = (C.c_void_p * Cn)(*[a.ctypes.data for a in cnts])
        if koc:
            self.lib.ko_dump_koc(C.byref(self.P), self.co.ctypes.data, pi, pc, nout)
            return [(ids[c], cnts[c]) for c in range(Cn)]
        self.lib.ko_dump_co(C.byref(self.P), self.co.ctypes.data, pi, nout)
        return [(ids[c], None) for c in range(Cn)]

    def koc_from_fastq(self, data):
        b = np.frombuffer(data, dtype=np.uint8)
        n = C.c_ulonglong(0)
        rc = self.lib.ko_koc_from_fastq_bytes(C.byref(self.P), self.table.ctypes.data, b.ctypes.data if len(b) else None,
                                              len(b), self.co.ctypes.data, C.byref(n))
        self.last_nreads = n.value
        if rc:
            return rc, None
        return 0, self._dump(True)

    def koc_from_rows(self, rows, stride, clear=True, dump=True):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n = rows.size // stride
        rc = self.lib.ko_koc_from_rows(C.byref(self.P), self.table.ctypes.data, rows.ctypes.data, stride, n,
                                       self.co.ctypes.data, 1 if clear else 0, C.byref(self.keycount))
        if rc:
            return rc, None
        return 0, (self._dump(True) if dump else None)

    def koc_from_rows_omp(self, rows, stride, nthreads):
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n = rows.size // stride
        return self.lib.ko_koc_from_rows_omp(C.byref(self.P), self.table.ctypes.data, rows.ctypes.data, stride, n,
                                             self.co.ctypes.data, 1, nthreads)

    def co_from_fastq(self, data, Q=0, M=1):
        """fastq2co(Q, M) + write_fqco2file: [(ids, None)] per component"""
        b = np.frombuffer(data, dtype=np.uint8)
        nl = C.c_ulonglong(0)
        rc = self.lib.ko_co_from_fastq_bytes(C.byref(self.P), self.table.ctypes.data, b.ctypes.data if len(b) else None,
                                             len(b), Q, M, self.co.ctypes.data, C.byref(nl))
        self.last_nlines = nl.value
        if rc:
            return rc, None
        Cn = self.P.component_num
        nout = (C.c_size_t * Cn)()
        self.lib.ko_dump_fqco(C.byref(self.P), self.co.ctypes.data, None, nout)
        ids = [np.zeros(nout[c], np.uint32) for c in range(Cn)]
        pi = (C.c_void_p * Cn)(*[a.ctypes.data for a in ids])
        self.lib.ko_dump_fqco(C.byref(self.P), self.co.ctypes.data, pi, nout)
        return 0, [(ids[c], None) for c in range(Cn)]

    def co_from_fasta(self, data, uniq=False):
        b = np.frombuffer(data, dtype=np.uint8)
        rc = self.lib.ko_co_from_fasta_bytes(C.byref(self.P), self.table.ctypes.data, b.ctypes.data if len(b) else None,
                                             len(b), self.co.ctypes.data, 1 if uniq else 0)
        if rc:
            return rc, None
        return 0, self._dump(False)


    def partial_from_rows(self, rows, stride, first_read_ordinal):
        """-> (keys u64, counts u32, ords u64) of the distinct keys of this read range"""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        n = rows.size // stride
        cap = int(self.P.hashlimit) + 1
        keys, cnts, ords = np.zeros(cap, np.uint64), np.zeros(cap, np.uint32), np.zeros(cap, np.uint64)
        m = C.c_size_t(0)
        rc = self.lib.ko_partial_from_rows(C.byref(self.P), self.table.ctypes.data, rows.ctypes.data, stride, n,
                                           first_read_ordinal, keys.ctypes.data, cnts.ctypes.data, ords.ctypes.data, cap,
                                           C.byref(m))
        assert rc == 0
        return keys[:m.value].copy(), cnts[:m.value].copy(), ords[:m.value].copy()

    def layout_from_partials(self, parts):
        """parts: list of (keys, counts, ords) -> sketch after the merge, in reference slot order"""
        k = [np.ascontiguousarray(p[0], np.uint64) for p in parts]
        c = [np.ascontiguousarray(p[1], np.uint32) for p in parts]
        o = [np.ascontiguousarray(p[2], np.uint64) for p in parts]
        n = (C.c_size_t * len(parts))(*[len(x) for x in k])
        pk = (C.c_void_p * len(parts))(*[x.ctypes.data for x in k])
        pc = (C.c_void_p * len(parts))(*[x.ctypes.data for x in c])
        po = (C.c_void_p * len(parts))(*[x.ctypes.data for x in o])
        rc = self.lib.ko_layout_from_partials(C.byref(self.P), len(parts), pk, pc, po, n, self.co.ctypes.data)
        assert rc == 0
        return self._dump(True)


def mco_build(ids, index):
    """ko_mco_build (stage II, co2mco.c:12-87) -> (gids, row_ids, row_ends)"""
    lib = load()
    vp = C.c_void_p
    lib.ko_mco_build.argtypes = [vp, vp, C.c_int, vp, C.POINTER(vp), C.POINTER(vp), C.POINTER(C.c_size_t)]
    lib.ko_mco_build.restype = C.c_int
    ids = np.ascontiguousarray(ids, dtype=np.uint32)
    index = np.ascontiguousarray(index, dtype=np.uint64)
    gids = np.zeros(max(ids.size, 1), np.uint32)
    ri, re_, nr = vp(), vp(), C.c_size_t(0)
    rc = lib.ko_mco_build(ids.ctypes.data, index.ctypes.data, index.size - 1, gids.ctypes.data, C.byref(ri), C.byref(re_), C.byref(nr))
    assert rc == 0
    n = nr.value
    row_ids = np.ctypeslib.as_array(C.cast(ri, C.POINTER(C.c_uint32)), shape=(max(n, 1),))[:n].copy()
    row_ends = np.ctypeslib.as_array(C.cast(re_, C.POINTER(C.c_uint64)), shape=(max(n, 1),))[:n].copy()
    libc = C.CDLL(None)
    libc.free.argtypes = [vp]
    libc.free(ri)
    libc.free(re_)
    return gids[:ids.size], row_ids, row_ends


def mco_count(gids, row_ids, row_ends, qry_ids, qry_index, qry_ctx_ct, ref_num):
    """ko_mco_count (command_dist.c:1033-1049) with the sparse row table -> qry_num x ref_num uint32"""
    lib = load()
    vp = C.c_void_p
    lib.ko_mco_count.argtypes = [vp, vp, vp, vp, C.c_size_t, vp, vp, C.c_int, vp, C.c_int, vp]
    lib.ko_mco_count.restype = None
    gids = np.ascontiguousarray(gids, dtype=np.uint32)
    row_ids = np.ascontiguousarray(row_ids, dtype=np.uint32)
    row_ends = np.ascontiguousarray(row_ends, dtype=np.uint64)
    qry_ids = np.ascontiguousarray(qry_ids, dtype=np.uint32)
    qry_index = np.ascontiguousarray(qry_index, dtype=np.uint64)
    ctx = np.ascontiguousarray(qry_ctx_ct, dtype=np.uint32)
    ct = np.zeros((ctx.size, ref_num), np.uint32)
    lib.ko_mco_count(gids.ctypes.data, None, row_ids.ctypes.data, row_ends.ctypes.data, row_ids.size, qry_ids.ctypes.data,
                     qry_index.ctypes.data, ctx.size, ctx.ctypes.data, ref_num, ct.ctypes.data)
    return ct


class KoDistOpts(C.Structure):
    _fields_ = [("metric", C.c_int), ("outfields", C.c_int), ("correction", C.c_int), ("dthreshold", C.c_double),
                ("num_neigb", C.c_int), ("keep_shared", C.c_int)]


def dist_print(path, ref_ctx_ct, qry_ctx_ct, refnames, qrynames, ct, kmerlen, dim_rd_len, metric=0, outfields=2, correction=0,
               num_neigb=0, dthreshold=1.0):
    """ko_dist_print (command_dist.c:1531-1690) into `path`; returns the oracle's status"""
    lib = load()
    vp = C.c_void_p
    lib.ko_dist_print.argtypes = [vp, C.POINTER(KoDistOpts), C.c_int, C.c_int, C.c_int, C.c_int, vp, vp, C.c_char_p, C.c_char_p, vp]
    lib.ko_dist_print.restype = C.c_int
    libc = C.CDLL(None)
    libc.fopen.restype = vp
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [vp]

    def pack(names):
        b = bytearray(256 * len(names))
        for i, nm in enumerate(names):
            e = nm.encode()[:255]
            b[256 * i:256 * i + len(e)] = e
        return bytes(b)
    rc_, qc_ = np.ascontiguousarray(ref_ctx_ct, dtype=np.uint32), np.ascontiguousarray(qry_ctx_ct, dtype=np.uint32)
    ct = np.ascontiguousarray(ct, dtype=np.uint32)
    o = KoDistOpts(metric, outfields, correction, dthreshold, num_neigb, 0)
    fp = libc.fopen(path.encode(), b"w")
    rc = lib.ko_dist_print(fp, C.byref(o), kmerlen, dim_rd_len, rc_.size, qc_.size, rc_.ctypes.data, qc_.ctypes.data, pack(refnames),
                           pack(qrynames), ct.ctypes.data)
    libc.fclose(fp)
    return rc
