"""ctypes binding of libmetakssd_hip.so (include/metakssd_hip.h) -- the binding a Python caller of the
reference's sketching path would use.  No compute happens here and there is no fallback: if the
shared library is missing, import fails; if no HIP device is usable, Engine() raises.

Names follow the reference's domain: shuf, params, sketch, reads/rows, components.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("MK_LIBRARY") or os.path.join(_HERE, "lib", "libmetakssd_hip.so")  # MK_LIBRARY: an experiment build (make tuning)

MK_OK = 0
MK_ERR_ARG, MK_ERR_NO_DEVICE, MK_ERR_HIP, MK_ERR_CROWDED = -1, -2, -3, -4
MK_ERR_STATE, MK_ERR_IO, MK_ERR_FORMAT, MK_ERR_NOMEM = -5, -6, -7, -8
MK_MODE_KOC, MK_MODE_SET, MK_MODE_UNIQ_SET, MK_MODE_OCC_SET = 0, 1, 2, 3
MK_ROWS_PACKED, MK_PACKED_PITCH, MK_PACKED_MAX_BASES = 0x80000000, 64, 152
MK_ROWS_WIDE, MK_WIDE_MAX_BASES = 0x40000000, 240
MK_OPT_SPARSE, MK_OPT_CAND_CAP, MK_OPT_RESULT_CAP, MK_OPT_DIRECT_HOST, MK_OPT_FRONT_BITS, MK_OPT_KEYLIST_CAP, MK_OPT_BATCH_TAB_BITS = 1, 2, 3, 4, 5, 6, 7
MK_BEGIN_NOTHING_FOLLOWS = 0x100  # mk_sketch_begin(mode | this): the last sketch of a run on split queues -- its tail on the whole device
MK_OPT_SPLIT_CUS = 10  # scan kernel on a queue of its own, the rest on this many compute units (two engines in turn: bench.py)


class MkError(RuntimeError):
    def __init__(self, code, msg=""):
        super().__init__("metakssd_hip error %d: %s" % (code, msg))
        self.code = code


class CrowdedError(MkError):
    """more than hashlimit distinct keys: the reference aborts with 'the context space is too crowd' (iseq2comem.c:708-709)"""


class ShufC(C.Structure):
    _fields_ = [("id", C.c_int32), ("k", C.c_int32), ("subk", C.c_int32), ("drlevel", C.c_int32),
                ("table", C.POINTER(C.c_int32)), ("len", C.c_uint64)]


class ParamsC(C.Structure):
    _fields_ = [("shuf_id", C.c_int32), ("k", C.c_int32), ("subk", C.c_int32), ("drlevel", C.c_int32),
                ("half_outctx_len", C.c_int32), ("TL", C.c_int32), ("crvsaddmove", C.c_int32),
                ("component_num", C.c_int32), ("comp_code_bits", C.c_int32),
                ("dim_start", C.c_int32), ("dim_end", C.c_int32),
                ("hashsize", C.c_uint32), ("hashlimit", C.c_uint32),
                ("tupmask", C.c_uint64), ("domask", C.c_uint64), ("undomask", C.c_uint64),
                ("shuf_table", C.POINTER(C.c_int32)), ("shuf_len", C.c_uint64), ("component_sz", C.c_int32), ("reserved", C.c_int32)]


class ComponentC(C.Structure):
    _fields_ = [("ids", C.POINTER(C.c_uint32)), ("counts", C.POINTER(C.c_uint16)), ("n", C.c_uint64)]


class ResultC(C.Structure):
    _fields_ = [("component_num", C.c_int32), ("total", C.c_uint64), ("components", C.POINTER(ComponentC))]


class ProfileC(C.Structure):
    _fields_ = [("scan_ms", C.c_double), ("scan_launches", C.c_uint64), ("resolve_ms", C.c_double), ("clear_ms", C.c_double),
                ("finish_ms", C.c_double), ("bases_scanned", C.c_uint64), ("rows_scanned", C.c_uint64), ("finish_side_ms", C.c_double)]


class FastaStateC(C.Structure):
    _fields_ = [("TL", C.c_uint32), ("in_header", C.c_uint32), ("fill", C.c_uint32), ("fresh", C.c_uint32),
                ("pending", C.c_uint8 * 4096)]


class FastqOptsC(C.Structure):
    _fields_ = [("occ", C.c_int32), ("qmin", C.c_int32), ("TL", C.c_int32), ("nthreads", C.c_int32), ("inflight", C.c_int32),
                ("chunk_bytes", C.c_uint64), ("drop_pages", C.c_int32), ("ahead", C.c_int32), ("packed", C.c_int32), ("fd", C.c_int32),
                ("early_chunks", C.c_int32), ("reserved2", C.c_int32), ("pool_bytes", C.c_uint64)]


class FastqStatsC(C.Structure):
    _fields_ = [("rows", C.c_uint64), ("records", C.c_uint64), ("chunks", C.c_uint64), ("chunks_discarded", C.c_uint64),
                ("serial_rows", C.c_uint64), ("threads", C.c_uint32), ("t_setup_s", C.c_double), ("t_wait_frame_s", C.c_double),
                ("t_push_s", C.c_double), ("t_total_s", C.c_double), ("t_push_call_s", C.c_double), ("t_wait_call_s", C.c_double),
                ("t_push_call_max_s", C.c_double), ("t_first_push_call_s", C.c_double)]


_PUSH_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_uint32, C.c_uint64, C.c_uint64, C.POINTER(C.c_uint64))
_WAIT_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_uint64)
_ALLOC_FN = C.CFUNCTYPE(C.c_void_p, C.c_void_p, C.c_size_t)
_RELEASE_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)
_READY_FN = C.CFUNCTYPE(None, C.c_void_p, C.c_void_p, C.c_size_t)


class RowsSinkC(C.Structure):
    _fields_ = [("ctx", C.c_void_p), ("push", _PUSH_FN), ("wait", _WAIT_FN), ("alloc", _ALLOC_FN), ("release", _RELEASE_FN),
                ("ready", C.c_void_p)]


class BatchFileC(C.Structure):
    _fields_ = [("text", C.c_void_p), ("n", C.c_uint64)]


class DistOptsC(C.Structure):
    _fields_ = [("metric", C.c_int32), ("outfields", C.c_int32), ("correction", C.c_int32), ("num_neigb", C.c_int32),
                ("dthreshold", C.c_double)]


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                          "(make -C metakssd_amd/csrc)" % LIB_PATH)
    lib = C.CDLL(LIB_PATH)
    vp, u8p, u64, u32, i32 = C.c_void_p, C.POINTER(C.c_uint8), C.c_uint64, C.c_uint32, C.c_int32
    sig = {
        "mk_shuf_read": [C.c_char_p, C.POINTER(ShufC)],
        "mk_shuf_generate": [i32, i32, i32, u64, C.POINTER(ShufC)],
        "mk_shuf_write": [C.POINTER(ShufC), C.c_char_p],
        "mk_params_init": [C.POINTER(ShufC), C.POINTER(ParamsC)],
        "mk_params_init_csz": [C.POINTER(ShufC), i32, C.POINTER(ParamsC)],
        "mk_device_count": [C.POINTER(C.c_int)],
        "mk_engine_create": [C.POINTER(ParamsC), C.c_int, C.POINTER(vp)],
        "mk_engine_destroy": [vp],
        "mk_engine_set_option": [vp, C.c_int, C.c_int64],
        "mk_engine_set_stream": [vp, vp],
        "mk_engine_use_own_stream": [vp],
        "mk_engine_share_scan_queue": [vp, vp],
        "mk_sketch_begin": [vp, C.c_int],
        "mk_sketch_begin_occ": [vp, C.c_int],
        "mk_sketch_push_reads": [vp, vp, u32, u64, u64],
        "mk_sketch_push_reads_device": [vp, vp, u32, u64, u64],
        "mk_sketch_push_reads_async": [vp, vp, u32, u64, u64, C.POINTER(u64)],
        "mk_sketch_push_wait": [vp, u64],
        "mk_sketch_push_stream": [vp, vp, u64, C.c_int],
        "mk_partial_count_begin": [vp],
        "mk_partial_export_async": [vp, vp, vp, vp, u64, C.POINTER(u64)],
        "mk_pack_rows_host": [vp, u32, u64, vp],
        "mk_params_packed_ok": [C.POINTER(ParamsC)],
        "mk_sketch_batch_begin": [vp, C.c_int, vp, u32],
        "mk_sketch_batch_begin_rows": [vp, C.c_int, u32, vp, u32],
        "mk_fasta_pack_rows": [vp, C.c_size_t, C.c_int32, u32, vp, u64, C.POINTER(u64)],
        "mk_sketch_batch_end": [vp, vp],
        "mk_sketch_finish": [vp, C.POINTER(ResultC)],
        "mk_sketch_finish_begin": [vp],
        "mk_sketch_finish_end": [vp, C.POINTER(ResultC)],
        "mk_result_release": [vp, C.POINTER(ResultC)],
        "mk_engine_sync": [vp],
        "mk_host_alloc": [C.POINTER(vp), C.c_size_t],
        "mk_host_free": [vp],
        "mk_host_arena_alloc": [C.POINTER(vp), C.c_size_t],
        "mk_host_arena_free": [vp, C.c_size_t],
        "mk_host_register": [vp, C.c_size_t],
        "mk_host_unregister": [vp],
        "mk_partial_count": [vp, C.POINTER(u64)],
        "mk_partial_export": [vp, vp, vp, vp, u64, C.POINTER(u64)],
        "mk_partial_import": [vp, vp, vp, vp, u64],
        "mk_partial_export_split": [vp, u32, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)],
        "mk_partial_export_split_async": [vp, u32, vp, vp, vp, u64, C.POINTER(u64), C.POINTER(u64)],
        "mk_partial_restart": [vp],
        "mk_partial_list_reserve": [vp, u64, C.POINTER(vp), C.POINTER(vp), C.POINTER(vp)],
        "mk_partial_list_commit": [vp, u64],
        "mk_partial_list_adopt": [vp, vp, vp, vp, u64, u64],
        "mk_profile_enable": [vp, C.c_int],
        "mk_profile_reset": [vp],
        "mk_profile_get": [vp, C.POINTER(ProfileC)],
        "mk_synth_rows_host": [u64, u64, u64, u32, u32, vp],
        "mk_synth_rows_device": [C.c_int, vp, u64, u64, u64, u32, u32, vp],
        "mk_synth_fastq_write": [C.c_char_p, u64, u64, u64, u32],
        "mk_synth_fastq_write_mt": [C.c_char_p, u64, u64, u64, u32, C.c_int],
        "mk_fastq_stream": [vp, C.c_size_t, C.POINTER(FastqOptsC), C.POINTER(RowsSinkC), u64, C.POINTER(FastqStatsC)],
        "mk_sketch_push_fastq": [vp, vp, C.c_size_t, C.POINTER(FastqOptsC), u64, C.POINTER(FastqStatsC)],
        "mk_fastq_frame": [vp, C.c_size_t, C.c_int, vp, u32, u64, C.POINTER(u64), C.POINTER(C.c_size_t)],
        "mk_fastq_frame_q": [vp, C.c_size_t, C.c_int, i32, i32, u64, vp, u32, u64, C.POINTER(u64), C.POINTER(u64),
                             C.POINTER(C.c_size_t)],
        "mk_fastq_frame_mt": [vp, C.c_size_t, C.c_int, C.c_int, i32, i32, u64, vp, u32, u64, C.c_int, C.POINTER(u64), C.POINTER(u64),
                              C.POINTER(C.c_size_t)],
        "mk_fasta_window_init": [C.POINTER(FastaStateC), i32],
        "mk_fasta_window": [C.POINTER(FastaStateC), vp, C.c_size_t, C.c_int, vp, u32, u64, C.POINTER(u64),
                            C.POINTER(C.c_size_t)],
        "mk_setop_create": [C.c_int, C.POINTER(vp)],
        "mk_setop_destroy": [vp],
        "mk_setop_begin": [vp, C.c_int],
        "mk_setop_add": [vp, vp, u64],
        "mk_setop_add_device": [vp, vp, u64],
        "mk_setop_finish": [vp, C.POINTER(vp), C.POINTER(u64)],
        "mk_setop_result_device": [vp, C.POINTER(vp), C.POINTER(u64)],
        "mk_setop_join": [vp, vp, vp, u64, vp, u64, vp, u32, C.POINTER(vp), C.POINTER(u64), vp],
        "mk_setop_last_join_ms": [vp, C.POINTER(C.c_double)],
        "mk_mco_create": [C.c_int, C.POINTER(vp)],
        "mk_mco_destroy": [vp],
        "mk_mco_build": [vp, vp, vp, u32, C.POINTER(vp), C.POINTER(u64), C.POINTER(vp), C.POINTER(vp), C.POINTER(u64)],
        "mk_mco_index_rows": [vp, u64, u64, vp],
        "mk_mco_sort_pairs": [vp, vp, vp, u64],
        "mk_mco_set_option": [vp, C.c_int, C.c_int64],
        "mk_mco_count_begin": [vp, u32, u32],
        "mk_mco_count_add": [vp, vp, u64, vp, vp, vp, vp, vp],
        "mk_mco_count_finish": [vp, vp],
        "mk_mco_last_kernel_ms": [vp, C.POINTER(C.c_double), C.POINTER(C.c_double)],
        "mk_dist_print": [vp, C.POINTER(DistOptsC), i32, i32, u32, u32, vp, vp, vp, vp, vp],
        "mk_sketchdir_open": [C.c_char_p, C.POINTER(ParamsC), C.c_int, C.c_int, C.POINTER(vp)],
        "mk_sketchdir_add": [vp, C.c_char_p, C.POINTER(ResultC)],
        "mk_sketchdir_close": [vp],
    }
    for name, args in sig.items():
        fn = getattr(lib, name)
        fn.argtypes = args
        fn.restype = C.c_int
    lib.mk_shuf_free.argtypes = [C.POINTER(ShufC)]
    lib.mk_shuf_free.restype = None
    lib.mk_setop_last_error.argtypes = [vp]
    lib.mk_setop_last_error.restype = C.c_char_p
    lib.mk_mco_last_error.argtypes = [vp]
    lib.mk_mco_last_error.restype = C.c_char_p
    lib.mk_setop_stream.argtypes = [vp]
    lib.mk_setop_stream.restype = vp
    lib.mk_last_error.argtypes = [vp]
    lib.mk_last_error.restype = C.c_char_p
    lib.mk_fasta_pack_bound.argtypes = [C.c_size_t, C.c_int32, u32]
    lib.mk_fasta_pack_bound.restype = u64
    return lib


lib = _load()


def _check(rc, eng=None):
    if rc == MK_OK:
        return
    msg = lib.mk_last_error(eng).decode(errors="replace") if (eng or rc in (MK_ERR_NO_DEVICE, MK_ERR_HIP, MK_ERR_NOMEM)) else ""
    raise (CrowdedError if rc == MK_ERR_CROWDED else MkError)(rc, msg)


class Shuf:
    """a .shuf dimension-shuffle table (command_shuffle.h:4-16)"""

    def __init__(self, c):
        self.c = c

    @classmethod
    def read(cls, path):
        c = ShufC()
        _check(lib.mk_shuf_read(os.fsencode(path), C.byref(c)))
        return cls(c)

    @classmethod
    def generate(cls, k, subk, drlevel, seed):
        c = ShufC()
        _check(lib.mk_shuf_generate(k, subk, drlevel, seed, C.byref(c)))
        return cls(c)

    def write(self, path):
        _check(lib.mk_shuf_write(C.byref(self.c), os.fsencode(path)))

    @property
    def table(self):
        return np.ctypeslib.as_array(self.c.table, shape=(self.c.len,))

    def params(self, component_sz=8):
        p = ParamsC()
        _check(lib.mk_params_init_csz(C.byref(self.c), component_sz, C.byref(p)))
        return p

    def __del__(self):
        try:
            lib.mk_shuf_free(C.byref(self.c))
        except Exception:
            pass


def device_count():
    n = C.c_int(0)
    rc = lib.mk_device_count(C.byref(n))
    return n.value if rc == MK_OK else 0


def synth_rows_host(seed, first_read, nreads, length, stride):
    rows = np.empty(nreads * stride, dtype=np.uint8)
    _check(lib.mk_synth_rows_host(seed, first_read, nreads, length, stride, rows.ctypes.data))
    return rows


def pack_rows_host(rows, stride):
    """ASCII rows -> 64-byte packed rows (mk_pack_rows_host); 16-byte aligned result"""
    rows = np.ascontiguousarray(rows, dtype=np.uint8)
    n = rows.size // stride
    raw = np.zeros(n * MK_PACKED_PITCH + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    out = raw[off:off + n * MK_PACKED_PITCH]
    _check(lib.mk_pack_rows_host(rows.ctypes.data if n else None, stride, n, out.ctypes.data if n else None))
    return out


def fasta_pack_rows(text, TL, fmt=None):
    """a whole FASTA text -> its packed rows (mk_fasta_pack_rows; fmt MK_ROWS_PACKED or MK_ROWS_WIDE); -> (rows u8 [nrows * 64], 16-byte aligned, rc)"""
    fmt = MK_ROWS_PACKED if fmt is None else fmt
    b = np.frombuffer(bytes(text), dtype=np.uint8)
    bound = int(lib.mk_fasta_pack_bound(len(b), TL, fmt))
    raw = np.zeros(bound * MK_PACKED_PITCH + 64, dtype=np.uint8)
    off = (-raw.ctypes.data) % 64
    n = C.c_uint64(0)
    rc = lib.mk_fasta_pack_rows(b.ctypes.data if len(b) else None, len(b), TL, fmt, raw.ctypes.data + off, bound, C.byref(n))
    return raw[off:off + n.value * MK_PACKED_PITCH], rc


def fastq_frame(buf, stride, final=True, max_rows=None):
    """FASTQ bytes -> fixed-stride rows (numpy u8 [nrows*stride]); returns (rows, nrows, consumed, rc)"""
    b = np.frombuffer(buf, dtype=np.uint8)
    max_rows = max_rows if max_rows is not None else len(b) // 4 + 1
    rows = np.zeros(max(1, max_rows) * stride, dtype=np.uint8)
    n, used = C.c_uint64(0), C.c_size_t(0)
    rc = lib.mk_fastq_frame(b.ctypes.data if len(b) else None, len(b), 1 if final else 0, rows.ctypes.data, stride,
                            max_rows, C.byref(n), C.byref(used))
    return rows[: n.value * stride], n.value, used.value, rc


def fastq_frame_q(buf, stride, TL, qmin=0, final=True, records_before=0, max_rows=None):
    """FASTQ bytes -> rows the way fastq2co() reads them (quality mask, record rule, long-read windows);
    returns (rows, nrows, nrecords, consumed, rc)"""
    b = np.frombuffer(buf, dtype=np.uint8)
    max_rows = max_rows if max_rows is not None else len(b) // 4 + len(b) // max(1, stride - TL) + 2
    rows = np.zeros(max(1, max_rows) * stride, dtype=np.uint8)
    n, nrec, used = C.c_uint64(0), C.c_uint64(0), C.c_size_t(0)
    rc = lib.mk_fastq_frame_q(b.ctypes.data if len(b) else None, len(b), 1 if final else 0, qmin, TL, records_before,
                              rows.ctypes.data, stride, max_rows, C.byref(n), C.byref(nrec), C.byref(used))
    return rows[: n.value * stride], n.value, nrec.value, used.value, rc


def fastq_frame_mt(buf, stride, nthreads, occ=False, TL=22, qmin=0, final=True, records_before=0, max_rows=None):
    """the threaded framer; returns (rows, nrows, nrecords, consumed, rc)"""
    b = np.frombuffer(buf, dtype=np.uint8)
    max_rows = max_rows if max_rows is not None else len(b) // 4 + len(b) // max(1, stride - TL) + 2
    rows = np.zeros(max(1, max_rows) * stride, dtype=np.uint8)
    n, nrec, used = C.c_uint64(0), C.c_uint64(0), C.c_size_t(0)
    rc = lib.mk_fastq_frame_mt(b.ctypes.data if len(b) else None, len(b), 1 if final else 0, 1 if occ else 0, qmin, TL, records_before,
                               rows.ctypes.data, stride, max_rows, nthreads, C.byref(n), C.byref(nrec), C.byref(used))
    return rows[: n.value * stride], n.value, nrec.value, used.value, rc


def fastq_stream(buf, nthreads=4, chunk_bytes=0, occ=False, TL=22, qmin=0, first_ordinal=0, inflight=2, drop_pages=False, packed=False, pool_bytes=0,
                 ready_log=None, via_fd=False, early_chunks=0):
    """the whole-file FASTQ stream (mk_fastq_stream) into host memory: returns (list of (rows u8 array, stride, nrows,
    first ordinal) in push order, stats, rc).  Buffers come from malloc here; the engine-bound form is Engine.push_fastq."""
    b = np.frombuffer(buf, dtype=np.uint8)
    pushes, keep = [], {}
    libc = C.CDLL(None)
    libc.malloc.restype = C.c_void_p
    libc.malloc.argtypes = [C.c_size_t]
    libc.free.argtypes = [C.c_void_p]

    def push(ctx, rows, stride, nrows, ord0, token):
        a = np.ctypeslib.as_array(C.cast(rows, C.POINTER(C.c_uint8)), shape=(nrows * (stride & ~MK_ROWS_PACKED),)).copy()
        pushes.append((a, stride, nrows, ord0))  # (stride & MK_ROWS_PACKED: 64-byte packed rows)
        token[0] = len(pushes)
        return 0

    def wait(ctx, token):
        return 0

    blocks = []

    def alloc(ctx, n):
        p = libc.malloc(n)
        blocks.append((p, n))
        return p

    def ready(ctx, rows, n):  # called on the framers' threads: a buffer has been written and waits for its push
        if ready_log is not None:
            ready_log.append((rows - blocks[-1][0], n))
    ready_fn = _READY_FN(ready)
    sink = RowsSinkC(None, _PUSH_FN(push), _WAIT_FN(wait), _ALLOC_FN(alloc), _RELEASE_FN(lambda ctx, p, n: libc.free(p)),
                     C.cast(ready_fn, C.c_void_p) if ready_log is not None else None)
    keep["sink"] = (sink, ready_fn)
    fd = -1
    if via_fd:  # the same bytes through a file descriptor: the framers pread pieces instead of reading a mapping (mk_fastq_opts.fd)
        import tempfile
        tf = tempfile.TemporaryFile()
        tf.write(bytes(buf))
        tf.flush()
        fd = tf.fileno()
        keep["file"] = tf
    o = FastqOptsC(1 if occ else 0, qmin, TL, nthreads, inflight, chunk_bytes, 1 if drop_pages else 0, 0, 1 if packed else 0, fd if via_fd else 0, early_chunks, 0, pool_bytes)
    if ready_log is not None:
        ready_log.append(blocks)
    st = FastqStatsC()
    rc = lib.mk_fastq_stream(None if via_fd or not len(b) else b.ctypes.data, len(b), C.byref(o), C.byref(sink), first_ordinal, C.byref(st))
    return pushes, st, rc


def fasta_windows(buf, TL, stride, chunk=None):
    """FASTA bytes -> overlapped fixed-stride rows; `chunk` feeds the input in pieces of that many bytes"""
    b = np.frombuffer(buf, dtype=np.uint8)
    st = FastaStateC()
    _check(lib.mk_fasta_window_init(C.byref(st), TL))
    out = []
    max_rows = 4096
    rows = np.zeros(max_rows * stride, dtype=np.uint8)
    pos, step = 0, (chunk or max(1, len(b)))
    while True:
        end = min(len(b), pos + step)
        final = end >= len(b)
        off = pos
        while True:
            n, used = C.c_uint64(0), C.c_size_t(0)
            ptr = b[off:end].ctypes.data if end > off else None
            _check(lib.mk_fasta_window(C.byref(st), ptr, end - off, 1 if final else 0, rows.ctypes.data, stride, max_rows,
                                       C.byref(n), C.byref(used)))
            if n.value:
                out.append(rows[: n.value * stride].copy())
            off += used.value
            if off >= end and n.value < max_rows:
                break
        pos = end
        if final:
            break
    return np.concatenate(out) if out else np.zeros(0, dtype=np.uint8)


class Engine:
    """one GPU's sketch engine (mk_engine): begin -> push_reads* -> finish"""

    def __init__(self, shuf, device=0, sparse=None, cand_cap=None, component_sz=8, front_bits=None):
        self.shuf = shuf  # keeps the host table alive
        self.params = shuf.params(component_sz)
        self.h = C.c_void_p()
        rc = lib.mk_engine_create(C.byref(self.params), device, C.byref(self.h))
        if rc != MK_OK:
            raise MkError(rc, lib.mk_last_error(None).decode(errors="replace"))
        self.device = device
        if sparse is not None:
            self.set_option(MK_OPT_SPARSE, int(sparse))
        if cand_cap is not None:
            self.set_option(MK_OPT_CAND_CAP, int(cand_cap))
        if front_bits is not None:
            self.set_option(MK_OPT_FRONT_BITS, int(front_bits))

    def set_option(self, option, value):
        """mk_engine_set_option: MK_OPT_SPARSE (-1/0/1), MK_OPT_CAND_CAP (records per scan wave); between sketches only"""
        _check(lib.mk_engine_set_option(self.h, option, value), self.h)

    def close(self):
        if self.h:
            lib.mk_engine_destroy(self.h)
            self.h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def set_stream(self, stream_handle):
        """run the engine's work on this hipStream_t handle (0/None = HIP's default stream, e.g. torch's current stream)"""
        _check(lib.mk_engine_set_stream(self.h, C.c_void_p(stream_handle or None)), self.h)

    def use_own_stream(self):
        _check(lib.mk_engine_use_own_stream(self.h), self.h)

    def share_scan_queue(self, owner):
        """mk_engine_share_scan_queue: this engine's scans go to OWNER's scan queue (both with the same MK_OPT_SPLIT_CUS setting)"""
        _check(lib.mk_engine_share_scan_queue(self.h, owner.h), self.h)

    def begin(self, mode=MK_MODE_KOC):
        _check(lib.mk_sketch_begin(self.h, mode), self.h)

    def begin_occ(self, min_occurrence=1):
        """FASTQ without -A (fastq2co): ids of keys seen at least min_occurrence times"""
        _check(lib.mk_sketch_begin_occ(self.h, min_occurrence), self.h)

    def push_reads(self, rows, stride, first_read_ordinal=0):
        """rows: host numpy u8 array of nreads*stride bytes (stride | MK_ROWS_PACKED: 64-byte packed rows)"""
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        pitch = stride & ~MK_ROWS_PACKED
        assert rows.size % pitch == 0
        n = rows.size // pitch
        if n:
            _check(lib.mk_sketch_push_reads(self.h, rows.ctypes.data, stride, n, first_read_ordinal), self.h)
        return n

    def push_fastq(self, buf, nthreads=8, chunk_bytes=0, occ=False, TL=22, qmin=0, first_ordinal=0, inflight=2, packed=False):
        """whole FASTQ text (bytes / numpy u8 / mmap) -> framed by host threads and pushed (mk_sketch_push_fastq); returns stats"""
        b = np.frombuffer(buf, dtype=np.uint8)
        o = FastqOptsC(1 if occ else 0, qmin, TL, nthreads, inflight, chunk_bytes, 0, 0, 1 if packed else 0)
        st = FastqStatsC()
        _check(lib.mk_sketch_push_fastq(self.h, b.ctypes.data if len(b) else None, len(b), C.byref(o), first_ordinal, C.byref(st)), self.h)
        return st

    def push_stream(self, text, final=True, piece=None):
        """FASTA text as it is in the file -> mk_sketch_push_stream; piece: push in pieces of that many bytes (the last one final)"""
        b = np.frombuffer(bytes(text), dtype=np.uint8)
        if not piece:
            _check(lib.mk_sketch_push_stream(self.h, b.ctypes.data if len(b) else None, len(b), 1 if final else 0), self.h)
            return
        at = 0
        while True:
            n = min(piece, len(b) - at)
            last = at + n >= len(b)
            part = np.ascontiguousarray(b[at:at + n])
            _check(lib.mk_sketch_push_stream(self.h, part.ctypes.data if n else None, n, 1 if (last and final) else 0), self.h)
            at += n
            if last:
                break

    def push_reads_async(self, rows, stride, first_read_ordinal=0):
        """queue the copies and the scan; `rows` (host numpy u8) must stay untouched until push_wait(ticket)"""
        assert rows.dtype == np.uint8 and rows.flags["C_CONTIGUOUS"] and rows.size % stride == 0
        t = C.c_uint64(0)
        _check(lib.mk_sketch_push_reads_async(self.h, rows.ctypes.data, stride, rows.size // stride, first_read_ordinal, C.byref(t)), self.h)
        return t.value

    def push_wait(self, ticket):
        _check(lib.mk_sketch_push_wait(self.h, ticket), self.h)

    def push_reads_device(self, dev_ptr, stride, nreads, first_read_ordinal=0):
        _check(lib.mk_sketch_push_reads_device(self.h, C.c_void_p(dev_ptr), stride, nreads, first_read_ordinal), self.h)

    def finish(self):
        """-> list over components of (ids u32 array, counts u16 array or None), copies of the engine-owned result"""
        r = ResultC()
        _check(lib.mk_sketch_finish(self.h, C.byref(r)), self.h)
        out = []
        for c in range(r.component_num):
            comp = r.components[c]
            n = comp.n
            ids = np.ctypeslib.as_array(comp.ids, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            cnt = None
            if comp.counts:
                cnt = np.ctypeslib.as_array(comp.counts, shape=(n,)).copy() if n else np.zeros(0, np.uint16)
            out.append((ids, cnt))
        self.last_total = r.total
        lib.mk_result_release(self.h, C.byref(r))
        return out

    def batch_begin(self, texts, mode=MK_MODE_SET, one_buffer=False):
        """mk_sketch_batch_begin over a list of byte strings (FASTA texts).  one_buffer: lay the texts into ONE
        buffer at 1 KiB-aligned offsets (the single-copy layout) instead of separate arrays.  The buffers are kept until batch_end()."""
        arrs = []
        if one_buffer:
            offs, at = [], 0
            for t in texts:
                offs.append(at)
                at += (len(t) + 1023) // 1024 * 1024
            buf = np.zeros(max(at, 1024), dtype=np.uint8)  # (pageable here; the command line uses a pinned buffer)
            for t, o in zip(texts, offs):
                buf[o:o + len(t)] = np.frombuffer(bytes(t), dtype=np.uint8)
            arrs = [buf[o:o + len(t)] for t, o in zip(texts, offs)]
            keep = [buf]
        else:
            arrs = [np.frombuffer(bytes(t), dtype=np.uint8).copy() for t in texts]
            keep = arrs
        files = (BatchFileC * len(texts))()
        for i, a in enumerate(arrs):
            files[i].text = (keep[0].ctypes.data + offs[i]) if one_buffer else (a.ctypes.data if a.size else None)
            files[i].n = a.size
        _check(lib.mk_sketch_batch_begin(self.h, mode, C.cast(files, C.c_void_p), len(texts)), self.h)
        self._batches = getattr(self, "_batches", [])
        self._batches.append((keep, len(texts)))

    def batch_begin_rows(self, row_arrays, mode=MK_MODE_SET, pinned=False, fmt=MK_ROWS_PACKED, gap=MK_PACKED_PITCH):
        """mk_sketch_batch_begin_rows over the files' packed rows (fasta_pack_rows).  pinned: the rows are laid into ONE registered buffer,
        file behind file with `gap` bytes of 0xA5 between them (never looked at) -- the layout the scan kernel reads in place; otherwise separate pageable arrays."""
        n = len(row_arrays)
        files = (BatchFileC * n)()
        if pinned:
            # (hipHostMalloc, not hipHostRegister on a numpy array: registering and unregistering pieces of the malloc heap left the
            # runtime with stale pinned ranges -- a pageable copy 44 fuzz cases later faulted on the device, tools/fuzz_parity.py seed 2010)
            total = sum(a.size + gap for a in row_arrays) + 4096
            p = C.c_void_p()
            _check(lib.mk_host_alloc(C.byref(p), total))
            buf = np.ctypeslib.as_array((C.c_uint8 * total).from_address(p.value))
            buf[:] = 0xA5
            at = 0
            for i, a in enumerate(row_arrays):
                buf[at:at + a.size] = a
                files[i].text = p.value + at
                files[i].n = a.size
                at += a.size + gap
            del buf
            keep = ("pinned", None, p.value)
        else:
            arrs = []
            for i, a in enumerate(row_arrays):
                raw = np.zeros(a.size + 64, dtype=np.uint8)
                off = (-raw.ctypes.data) % 64
                raw[off:off + a.size] = a
                arrs.append(raw)
                files[i].text = (raw.ctypes.data + off) if a.size else None
                files[i].n = a.size
            keep = arrs
        _check(lib.mk_sketch_batch_begin_rows(self.h, mode, fmt, C.cast(files, C.c_void_p), n), self.h)
        self._batches = getattr(self, "_batches", [])
        self._batches.append((keep, n))

    def batch_end(self):
        """-> per file (status, alone, [ids per component]) of the oldest batch in flight"""
        if not getattr(self, "_batches", None):  # nothing in flight: let the library say so
            _check(lib.mk_sketch_batch_end(self.h, C.cast((C.c_uint8 * 64)(), C.c_void_p)), self.h)
            raise MkError(MK_ERR_STATE, "mk_sketch_batch_end returned MK_OK without a batch in flight")
        keep, n = self._batches.pop(0)

        class BatchResultC(C.Structure):
            _fields_ = [("status", C.c_int32), ("alone", C.c_int32), ("r", ResultC)]
        out = (BatchResultC * n)()
        _check(lib.mk_sketch_batch_end(self.h, C.cast(out, C.c_void_p)), self.h)
        res = []
        for i in range(n):
            comps = []
            if out[i].status == MK_OK:
                for c in range(out[i].r.component_num):
                    comp = out[i].r.components[c]
                    comps.append(np.ctypeslib.as_array(comp.ids, shape=(comp.n,)).copy() if comp.n else np.zeros(0, np.uint32))
            res.append((out[i].status, out[i].alone, comps))
        if isinstance(keep, tuple) and keep[0] == "pinned":
            lib.mk_host_free(keep[2])
        del keep
        return res

    def finish_raw(self):
        r = ResultC()
        _check(lib.mk_sketch_finish(self.h, C.byref(r)), self.h)
        return r

    def finish_begin(self):
        """first half of a finish: when it returns the next sketch may be begun and pushed; finish_end() hands out the result"""
        _check(lib.mk_sketch_finish_begin(self.h), self.h)

    def finish_end_raw(self):
        r = ResultC()
        _check(lib.mk_sketch_finish_end(self.h, C.byref(r)), self.h)
        return r

    def finish_end(self):
        r = self.finish_end_raw()
        out = []
        for c in range(r.component_num):
            comp = r.components[c]
            n = comp.n
            ids = np.ctypeslib.as_array(comp.ids, shape=(n,)).copy() if n else np.zeros(0, np.uint32)
            cnt = None
            if comp.counts:
                cnt = np.ctypeslib.as_array(comp.counts, shape=(n,)).copy() if n else np.zeros(0, np.uint16)
            out.append((ids, cnt))
        self.last_total = r.total
        return out

    def sync(self):
        _check(lib.mk_engine_sync(self.h), self.h)

    def partial_count(self):
        n = C.c_uint64(0)
        _check(lib.mk_partial_count(self.h, C.byref(n)), self.h)
        return n.value

    def partial_export(self, keys_ptr, counts_ptr, ords_ptr, capacity):
        n = C.c_uint64(0)
        _check(lib.mk_partial_export(self.h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), C.c_void_p(ords_ptr), capacity,
                                     C.byref(n)), self.h)
        return n.value

    def partial_import(self, keys_ptr, counts_ptr, ords_ptr, n):
        _check(lib.mk_partial_import(self.h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), C.c_void_p(ords_ptr), n), self.h)

    # ---- the merge by key slices (include/metakssd_hip.h) ----
    def partial_export_split(self, nparts, keys_ptr, counts_ptr, ords_ptr, capacity):
        """the distinct list cut into nparts parts by key % nparts -> (n, [part sizes])"""
        n = C.c_uint64(0)
        parts = (C.c_uint64 * 16)()
        _check(lib.mk_partial_export_split(self.h, nparts, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), C.c_void_p(ords_ptr), capacity,
                                           parts, C.byref(n)), self.h)
        return n.value, [int(parts[g]) for g in range(nparts)]

    def partial_restart(self):
        _check(lib.mk_partial_restart(self.h), self.h)

    def partial_list_reserve(self, n):
        """device pointers (keys, counts, ords) of the engine's own key list, room for n entries"""
        k, c, o = C.c_void_p(), C.c_void_p(), C.c_void_p()
        _check(lib.mk_partial_list_reserve(self.h, n, C.byref(k), C.byref(c), C.byref(o)), self.h)
        return k.value, c.value, o.value

    def partial_list_adopt(self, keys_ptr, counts_ptr, ords_ptr, n, offset):
        _check(lib.mk_partial_list_adopt(self.h, C.c_void_p(keys_ptr), C.c_void_p(counts_ptr), C.c_void_p(ords_ptr), n, offset), self.h)

    def partial_list_commit(self, n):
        _check(lib.mk_partial_list_commit(self.h, n), self.h)

    def profile_enable(self, on=True):
        _check(lib.mk_profile_enable(self.h, 1 if on else 0), self.h)

    def profile_reset(self):
        _check(lib.mk_profile_reset(self.h), self.h)

    def profile(self):
        p = ProfileC()
        _check(lib.mk_profile_get(self.h, C.byref(p)), self.h)
        return {f: getattr(p, f) for f, _ in ProfileC._fields_}


def synth_rows_device(device, stream, seed, first_read, nreads, length, stride, dev_ptr):
    _check(lib.mk_synth_rows_device(device, C.c_void_p(stream), seed, first_read, nreads, length, stride, C.c_void_p(dev_ptr)))


# ---- libmetakssd_multi.so (include/metakssd_multi.h): several engines in ONE process, the product's own exchange (RCCL) ----
MK_MULTI_ALLOW_DEVICE_COPIES, MK_MULTI_FORCE_DEVICE_COPIES = 1, 2
MK_MULTI_MERGE_AUTO, MK_MULTI_MERGE_GATHER, MK_MULTI_MERGE_SLICES = 0, 1, 2
MULTI_LIB_PATH = os.path.join(os.path.dirname(LIB_PATH), "libmetakssd_multi.so")
_mlib = None


class MultiTimesC(C.Structure):
    _fields_ = [(f, C.c_double) for f in ("export_ms", "exchange_ms", "import_ms", "gather_ms", "finish_ms", "total_ms")]


def _load_multi():
    global _mlib
    if _mlib is not None:
        return _mlib
    if not os.path.exists(MULTI_LIB_PATH):
        raise ImportError("%s not built (needs rccl/rccl.h and librccl.so: make -C metakssd_amd/csrc)" % MULTI_LIB_PATH)
    m = C.CDLL(MULTI_LIB_PATH)  # (pulls in librccl.so: only callers of Multi pay for that)
    vp = C.c_void_p
    m.mk_multi_create_ex.argtypes = [C.POINTER(ParamsC), C.POINTER(C.c_int), C.c_int, C.c_uint, C.POINTER(vp)]
    m.mk_multi_destroy.argtypes = [vp]
    m.mk_multi_last_error.argtypes = [vp]
    m.mk_multi_last_error.restype = C.c_char_p
    m.mk_multi_count.argtypes = [vp]
    m.mk_multi_engine.argtypes = [vp, C.c_int]
    m.mk_multi_engine.restype = vp
    m.mk_multi_transport.argtypes = [vp]
    m.mk_multi_transport.restype = C.c_char_p
    m.mk_multi_begin.argtypes = [vp, C.c_int]
    m.mk_multi_begin_occ.argtypes = [vp, C.c_int]
    m.mk_multi_finish.argtypes = [vp, C.POINTER(ResultC), C.POINTER(C.c_double), C.POINTER(C.c_double)]
    m.mk_multi_set_merge.argtypes = [vp, C.c_int]
    m.mk_multi_last_merge.argtypes = [vp]
    m.mk_multi_last_merge.restype = C.c_char_p
    m.mk_multi_last_times.argtypes = [vp, C.POINTER(MultiTimesC)]
    _mlib = m
    return m


class Multi:
    """one sketch over several engines of ONE process (mk_multi_*): what `metakssd dist --devices a,b,..` runs on"""

    def __init__(self, shuf, devices, flags=0, component_sz=8):
        self.m = _load_multi()
        self.params = shuf.params(component_sz)
        arr = (C.c_int * len(devices))(*devices)
        h = C.c_void_p()
        rc = self.m.mk_multi_create_ex(C.byref(self.params), arr, len(devices), flags, C.byref(h))
        if rc != MK_OK:
            raise MkError(rc, self.m.mk_multi_last_error(None).decode(errors="replace"))
        self.h = h
        self.n = len(devices)
        self.devices = list(devices)

    def _check(self, rc):
        if rc != MK_OK:
            raise (CrowdedError if rc == MK_ERR_CROWDED else MkError)(rc, self.m.mk_multi_last_error(self.h).decode(errors="replace"))

    def engine(self, i):
        return self.m.mk_multi_engine(self.h, i)

    def transport(self):
        return self.m.mk_multi_transport(self.h).decode()

    def set_merge(self, how):
        self._check(self.m.mk_multi_set_merge(self.h, how))

    def begin(self, mode=MK_MODE_KOC):
        self._check(self.m.mk_multi_begin(self.h, mode))

    def begin_occ(self, min_occurrence):
        self._check(self.m.mk_multi_begin_occ(self.h, min_occurrence))

    def push_reads_device(self, i, dev_ptr, stride, nreads, first_read_ordinal):
        e = self.engine(i)
        _check(lib.mk_sketch_push_reads_device(e, C.c_void_p(dev_ptr), stride, nreads, first_read_ordinal), e)

    def push_reads(self, i, rows, stride, first_read_ordinal):
        e = self.engine(i)
        rows = np.ascontiguousarray(rows, dtype=np.uint8)
        _check(lib.mk_sketch_push_reads(e, rows.ctypes.data_as(C.c_void_p), stride, rows.size // stride, first_read_ordinal), e)

    def finish_raw(self):
        """-> (ResultC of engine 0, gather_ms, tail_ms)"""
        r = ResultC()
        g, t = C.c_double(0), C.c_double(0)
        self._check(self.m.mk_multi_finish(self.h, C.byref(r), C.byref(g), C.byref(t)))
        return r, g.value, t.value

    def finish(self):
        r, g, t = self.finish_raw()
        out = []
        for c in range(r.component_num):
            comp = r.components[c]
            ids = np.ctypeslib.as_array(comp.ids, shape=(comp.n,)).copy() if comp.n else np.zeros(0, np.uint32)
            cnt = (np.ctypeslib.as_array(comp.counts, shape=(comp.n,)).copy() if comp.n else np.zeros(0, np.uint16)) if comp.counts else None
            out.append((ids, cnt))
        return out

    def last_merge(self):
        return self.m.mk_multi_last_merge(self.h).decode()

    def last_times(self):
        t = MultiTimesC()
        self._check(self.m.mk_multi_last_times(self.h, C.byref(t)))
        return {f: getattr(t, f) for f, _ in MultiTimesC._fields_}

    def close(self):
        if getattr(self, "h", None):
            self.m.mk_multi_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:  # noqa: BLE001
            pass


MK_SET_UNION, MK_SET_UNIQ_UNION = 0, 1


class SetOp:
    """`metakssd set -u / -q` dictionaries on the device (mk_setop_*)"""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib.mk_setop_create(device, C.byref(self.h))
        if rc:
            raise MkError(rc, (lib.mk_setop_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc:
            raise MkError(rc, (lib.mk_setop_last_error(self.h) or b"").decode())

    def union(self, id_lists, uniq=False):
        """id_lists: iterable of uint32 numpy arrays -> ascending uint32 array"""
        self._check(lib.mk_setop_begin(self.h, MK_SET_UNIQ_UNION if uniq else MK_SET_UNION))
        for a in id_lists:
            a = np.ascontiguousarray(a, dtype=np.uint32)
            if a.size:
                self._check(lib.mk_setop_add(self.h, a.ctypes.data, a.size))
        out, n = C.c_void_p(), C.c_uint64(0)
        self._check(lib.mk_setop_finish(self.h, C.byref(out), C.byref(n)))
        if n.value == 0:
            return np.zeros(0, np.uint32)
        return np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(n.value,)).copy()

    def begin(self, uniq=False):
        self._check(lib.mk_setop_begin(self.h, MK_SET_UNIQ_UNION if uniq else MK_SET_UNION))

    def add_device(self, dev_ptr, n):
        self._check(lib.mk_setop_add_device(self.h, C.c_void_p(dev_ptr), n))

    def finish_count(self):
        out, n = C.c_void_p(), C.c_uint64(0)
        self._check(lib.mk_setop_finish(self.h, C.byref(out), C.byref(n)))
        return n.value

    def stream(self):
        return lib.mk_setop_stream(self.h)

    def join(self, qry_ids, qry_counts, ref_ids, bounds):
        """composite -q's join (mk_setop_join): the query's count for every reference position whose id the query holds, in reference
        order, and where each reference sketch's segment of them ends -> (counts uint32, bounds_out uint64)"""
        q = np.ascontiguousarray(qry_ids, dtype=np.uint32)
        qa = np.ascontiguousarray(qry_counts, dtype=np.uint16)
        r = np.ascontiguousarray(ref_ids, dtype=np.uint32)
        b = np.ascontiguousarray(bounds, dtype=np.uint64)
        bout = np.zeros(b.size, np.uint64)
        out, n = C.c_void_p(), C.c_uint64(0)
        self._check(lib.mk_setop_join(self.h, q.ctypes.data if q.size else None, qa.ctypes.data if qa.size else None, q.size,
                                      r.ctypes.data if r.size else None, r.size, b.ctypes.data if b.size else None, b.size,
                                      C.byref(out), C.byref(n), bout.ctypes.data if b.size else None))
        cts = np.ctypeslib.as_array(C.cast(out, C.POINTER(C.c_uint32)), shape=(n.value,)).copy() if n.value else np.zeros(0, np.uint32)
        return cts, bout

    def last_join_ms(self):
        ms = C.c_double(0.0)
        self._check(lib.mk_setop_last_join_ms(self.h, C.byref(ms)))
        return ms.value

    def close(self):
        if self.h:
            lib.mk_setop_destroy(self.h)
            self.h = C.c_void_p()


class Mco:
    """stage II inverted index and shared-k-mer counting on the device (mk_mco_*)"""

    def __init__(self, device=0):
        self.h = C.c_void_p()
        rc = lib.mk_mco_create(device, C.byref(self.h))
        if rc:
            raise MkError(rc, (lib.mk_mco_last_error(None) or b"").decode())

    def _check(self, rc):
        if rc:
            raise MkError(rc, (lib.mk_mco_last_error(self.h) or b"").decode())

    def build(self, ids, index, copy=True):
        """ids: uint32 (one component's combco.N), index: uint64[cofnum + 1] -> (gids, row_ids, row_ends).
        copy=False: views of the library's pinned result buffers, as the C caller gets them (valid until the next build / close)"""
        ids = np.ascontiguousarray(ids, dtype=np.uint32)
        index = np.ascontiguousarray(index, dtype=np.uint64)
        g, ri, re_ = C.c_void_p(), C.c_void_p(), C.c_void_p()
        n, nr = C.c_uint64(0), C.c_uint64(0)
        self._check(lib.mk_mco_build(self.h, ids.ctypes.data if ids.size else None, index.ctypes.data, index.size - 1, C.byref(g),
                                     C.byref(n), C.byref(ri), C.byref(re_), C.byref(nr)))

        def arr(p, ct, k):
            if not k:
                return np.zeros(0, ct)
            a = np.ctypeslib.as_array(C.cast(p, C.POINTER(ct)), shape=(k,))
            return a.copy() if copy else a
        return arr(g, C.c_uint32, n.value), arr(ri, C.c_uint32, nr.value), arr(re_, C.c_uint64, nr.value)

    def sort_pairs(self, keys, vals):
        """stable sort of (u32 key, u32 value) pairs by key on the device (the sort inside build()); returns new arrays"""
        k = np.ascontiguousarray(keys, dtype=np.uint32).copy()
        v = np.ascontiguousarray(vals, dtype=np.uint32).copy()
        assert k.size == v.size
        self._check(lib.mk_mco_sort_pairs(self.h, k.ctypes.data if k.size else None, v.ctypes.data if v.size else None, k.size))
        return k, v

    def index_rows(self, row0, nrows, pinned=False):
        """rows [row0, row0 + nrows) of the dense index.  pinned=True: into a pinned buffer this object keeps (what the command line
        does, mk_host_alloc), returned as a view that the next pinned call overwrites -- the slab then crosses at PCIe speed"""
        if pinned and nrows:
            if getattr(self, "_slab_cap", 0) < nrows:
                if getattr(self, "_slab", None):
                    lib.mk_host_free(self._slab)
                self._slab, self._slab_cap = C.c_void_p(), 0
                rc = lib.mk_host_alloc(C.byref(self._slab), nrows * 8)
                if rc:
                    raise MkError(rc, "mk_host_alloc")
                self._slab_cap = nrows
            self._check(lib.mk_mco_index_rows(self.h, row0, nrows, self._slab))
            return np.ctypeslib.as_array(C.cast(self._slab, C.POINTER(C.c_uint64)), shape=(nrows,))
        out = np.empty(nrows, np.uint64)
        self._check(lib.mk_mco_index_rows(self.h, row0, nrows, out.ctypes.data if nrows else None))
        return out

    def count(self, ref_num, qry_index, qry_ctx_ct, components):
        """components: iterable of dicts {gids (or None: last build), qry_ids, ext_start, ext_end (or None: device row table)};
        qry_index per component inside the dict.  Returns the qry_num x ref_num uint32 matrix."""
        qry_num = len(qry_ctx_ct)
        self._check(lib.mk_mco_count_begin(self.h, ref_num, qry_num))
        ctx = np.ascontiguousarray(qry_ctx_ct, dtype=np.uint32)
        for comp in components:
            qi = np.ascontiguousarray(comp.get("qry_index", qry_index), dtype=np.uint64)
            gids = comp.get("gids")
            gids = None if gids is None else np.ascontiguousarray(gids, dtype=np.uint32)
            qids = comp.get("qry_ids")
            qids = None if qids is None else np.ascontiguousarray(qids, dtype=np.uint32)
            es, ee = comp.get("ext_start"), comp.get("ext_end")
            es = None if es is None else np.ascontiguousarray(es, dtype=np.uint64)
            ee = None if ee is None else np.ascontiguousarray(ee, dtype=np.uint64)
            self._check(lib.mk_mco_count_add(self.h, gids.ctypes.data if gids is not None and gids.size else (None if gids is None else 0),
                                             0 if gids is None else gids.size,
                                             qids.ctypes.data if qids is not None and qids.size else None,
                                             es.ctypes.data if es is not None else None, ee.ctypes.data if ee is not None else None,
                                             qi.ctypes.data, ctx.ctypes.data if ctx.size else None))
        ct = np.zeros((qry_num, ref_num), np.uint32)
        if ct.size:
            self._check(lib.mk_mco_count_finish(self.h, ct.ctypes.data))
        return ct

    def last_kernel_ms(self):
        """(radix sort of the last build, kernels of the last count_add) in ms, from the handle's HIP events"""
        a, b = C.c_double(0.0), C.c_double(0.0)
        self._check(lib.mk_mco_last_kernel_ms(self.h, C.byref(a), C.byref(b)))
        return a.value, b.value

    def close(self):
        if self.h:
            if getattr(self, "_slab", None):
                lib.mk_host_free(self._slab)
                self._slab, self._slab_cap = None, 0
            lib.mk_mco_destroy(self.h)
            self.h = C.c_void_p()


def dist_print(path, ref_ctx_ct, qry_ctx_ct, refnames, qrynames, ct, kmerlen, dim_rd_len, metric=0, outfields=2, correction=0,
               num_neigb=0, dthreshold=1.0):
    """write distance.out through mk_dist_print (host code: needs no GPU); names: lists of str"""
    libc = C.CDLL(None)
    libc.fopen.restype = C.c_void_p
    libc.fopen.argtypes = [C.c_char_p, C.c_char_p]
    libc.fclose.argtypes = [C.c_void_p]

    def pack(names):
        b = bytearray(256 * len(names))
        for i, nm in enumerate(names):
            e = nm.encode()[:255]
            b[256 * i:256 * i + len(e)] = e
        return bytes(b)
    rn, qn = pack(refnames), pack(qrynames)
    rc_, qc_ = np.ascontiguousarray(ref_ctx_ct, dtype=np.uint32), np.ascontiguousarray(qry_ctx_ct, dtype=np.uint32)
    ct = np.ascontiguousarray(ct, dtype=np.uint32)
    o = DistOptsC(metric, outfields, correction, num_neigb, dthreshold)
    fp = libc.fopen(path.encode(), b"w")
    if not fp:
        raise OSError("cannot open %s" % path)
    rc = lib.mk_dist_print(fp, C.byref(o), kmerlen, dim_rd_len, rc_.size, qc_.size, rc_.ctypes.data if rc_.size else None,
                           qc_.ctypes.data if qc_.size else None, rn, qn, ct.ctypes.data if ct.size else None)
    libc.fclose(fp)
    return rc
