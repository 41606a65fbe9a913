"""the C-ABI library loads without a GPU, exports every symbol include/metakssd_hip.h declares, and fails loudly
(no CPU fallback) when no HIP device is usable"""
import ctypes
import os
import re

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
HEADER = os.path.join(ROOT, "include", "metakssd_hip.h")
LIB = os.path.join(ROOT, "metakssd_amd", "lib", "libmetakssd_hip.so")


MULTI_HEADER = os.path.join(ROOT, "include", "metakssd_multi.h")
MULTI_LIB = os.path.join(ROOT, "metakssd_amd", "lib", "libmetakssd_multi.so")


def declared_functions(header=HEADER):
    src = open(header).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    return sorted(set(re.findall(r"\b(mk_[a-z0-9_]+)\s*\(", src)))


def test_header_declares_the_expected_surface():
    names = declared_functions()
    for must in ("mk_engine_create", "mk_sketch_begin", "mk_sketch_push_reads", "mk_sketch_push_reads_device",
                 "mk_sketch_finish", "mk_partial_export", "mk_partial_import", "mk_shuf_read", "mk_params_init",
                 "mk_fastq_frame", "mk_fasta_window", "mk_sketchdir_open", "mk_last_error"):
        assert must in names


def test_library_exports_every_declared_symbol():
    lib = ctypes.CDLL(LIB)
    missing = [n for n in declared_functions() if not hasattr(lib, n)]
    assert not missing, "declared in include/metakssd_hip.h but not exported: %s" % missing


def test_multi_gpu_library_exports_every_declared_symbol():
    """include/metakssd_multi.h -> libmetakssd_multi.so (engines on several GPUs + the RCCL exchange); loading it pulls in
    librccl.so, which is why it is a library of its own"""
    lib = ctypes.CDLL(MULTI_LIB)
    names = declared_functions(MULTI_HEADER)
    assert "mk_multi_create" in names and "mk_multi_finish" in names
    missing = [n for n in names if n.startswith("mk_multi_") and not hasattr(lib, n)]
    assert not missing, missing
    deps = os.popen("ldd %s" % os.path.join(ROOT, "metakssd_amd", "bin", "metakssd")).read()
    assert "rccl" not in deps and "metakssd_multi" not in deps, "the single-GPU command line must not load RCCL at start-up"


def test_no_cpu_fallback_without_device():
    from metakssd_amd import capi
    if capi.device_count() > 0:
        pytest.skip("a HIP device is present")
    shuf = capi.Shuf.generate(7, 4, 1, 7)
    with pytest.raises(capi.MkError) as ei:
        capi.Engine(shuf, 0)
    assert ei.value.code == capi.MK_ERR_NO_DEVICE
    assert "no CPU path" in str(ei.value)


def test_product_does_not_link_or_import_the_oracle():
    """the oracle is test infrastructure: nothing under metakssd_amd/ or include/ may mention it"""
    bad = []
    for base in ("metakssd_amd", "include"):
        for dp, dn, fs in os.walk(os.path.join(ROOT, base)):
            if "build" in dp.split(os.sep):
                continue
            for f in fs:
                if f.endswith((".py", ".c", ".h", ".hip", ".cpp", "Makefile")):
                    txt = open(os.path.join(dp, f), errors="replace").read()
                    if re.search(r"kssd_oracle|oracle_binding|oracle/", txt):
                        bad.append(os.path.join(dp, f))
    assert not bad, bad
