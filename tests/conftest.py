import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
for p in (ROOT, os.path.join(ROOT, "tests")):
    if p not in sys.path:
        sys.path.insert(0, p)


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session", autouse=True)
def _built():
    """the C-ABI library and the oracle must exist; build them if this checkout is fresh (CPU-only is fine)"""
    import subprocess
    lib = os.path.join(ROOT, "metakssd_amd", "lib", "libmetakssd_hip.so")
    if not os.path.exists(lib):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "metakssd_amd", "csrc")])
    ora = os.path.join(ROOT, "oracle", "libkssd_oracle.so")
    if not os.path.exists(ora) or not os.path.exists(os.path.join(ROOT, "oracle", "kssd_oracle_cli")):
        subprocess.check_call(["make", "-s", "-C", os.path.join(ROOT, "oracle")])


# name -> (k, subk, drlevel, seed); the same seeds as oracle/check_vs_ref.py and tests/golden/make_golden.py
SHUF_SPECS = {
    "L3K11": (11, 6, 3, 11),   # BASELINE config: 22-mers, 1/4096 accepted, table 33 554 393
    "L3K10": (10, 6, 3, 10),   # config 5: table 2 097 143
    "L2K11": (11, 5, 2, 211),  # config 5: 16 components, table 536 870 909
    "L3K9": (9, 6, 3, 9),      # table 131 071
    "L0K6": (6, 3, 0, 6),      # every k-mer accepted, table 131 071: collisions from tiny inputs
    "L1K7": (7, 4, 1, 7),      # 1/16 accepted, table 131 071
}

_shuf_cache = {}


@pytest.fixture(scope="session")
def shufs():
    from metakssd_amd import capi

    def get(name):
        if name not in _shuf_cache:
            if name == "L0K6z":  # L0K6 with inner substring 0 mapped to 0: poly-A/T reads give key 0
                import numpy as np
                base = get("L0K6")
                s = capi.Shuf.generate(6, 3, 0, 6)
                t = s.table
                j = int(np.nonzero(base.table == 0)[0][0])
                t[j], t[0] = t[0], 0
                _shuf_cache[name] = s
            else:
                k, subk, drl, seed = SHUF_SPECS[name]
                _shuf_cache[name] = capi.Shuf.generate(k, subk, drl, seed)
        return _shuf_cache[name]

    return get


@pytest.fixture(scope="session")
def oracle_for():
    from oracle_binding import Oracle

    def make(shuf):
        return Oracle(shuf.c.id, shuf.c.k, shuf.c.subk, shuf.c.drlevel, shuf.table)

    return make


# ---- failing tests keep their evidence ------------------------------------------------------------------------------------------
# Every subprocess.run() of a test is logged (command, return code, ends of stdout / stderr); when the test fails, the log, pytest's
# report and the test's tmp_path (files up to 4 MiB each, 48 MiB in all; no .shuf tables) are copied to gpurun_out/fail_<test>/ --
# the directory gpurun brings home from the GPU box.  A pipeline test of seven product processes then says WHICH process differed.
_FAIL_ROOT = os.path.join(ROOT, "gpurun_out")


@pytest.fixture(autouse=True)
def _stage_log(request, monkeypatch):
    import subprocess
    log = []
    real_run = subprocess.run

    def run(*args, **kw):
        r = real_run(*args, **kw)
        try:
            def tail(b):
                if b is None:
                    return ""
                if isinstance(b, bytes):
                    b = b.decode(errors="replace")
                return b[-2000:]
            log.append({"cmd": r.args if isinstance(r.args, (list, tuple)) else [r.args], "rc": r.returncode,
                        "stdout_tail": tail(r.stdout), "stderr_tail": tail(r.stderr)})
        except Exception:  # the log must never fail a test
            pass
        return r

    monkeypatch.setattr(subprocess, "run", run)
    request.node._mk_stage_log = log
    yield


@pytest.hookimpl(hookwrapper=True)
def pytest_runtest_makereport(item, call):
    outcome = yield
    rep = outcome.get_result()
    if rep.when != "call" or not rep.failed:
        return
    try:
        import json
        import re
        import shutil
        name = re.sub(r"[^A-Za-z0-9_.-]+", "_", item.nodeid.split("::", 1)[-1])[:120]
        dst = os.path.join(_FAIL_ROOT, "fail_" + name)
        os.makedirs(dst, exist_ok=True)
        with open(os.path.join(dst, "report.txt"), "w") as f:
            f.write(item.nodeid + "\n\n" + str(rep.longrepr) + "\n")
        with open(os.path.join(dst, "stages.json"), "w") as f:
            json.dump([{**s, "cmd": [str(c) for c in s["cmd"]]} for s in getattr(item, "_mk_stage_log", [])], f, indent=1)
        tmp = item.funcargs.get("tmp_path") if hasattr(item, "funcargs") else None
        if tmp is not None and os.path.isdir(str(tmp)):
            budget = 48 << 20
            for d, _, files in os.walk(str(tmp)):
                for fn in sorted(files):
                    src = os.path.join(d, fn)
                    sz = os.path.getsize(src)
                    if fn.endswith(".shuf") or sz > (4 << 20) or sz > budget:
                        continue
                    out = os.path.join(dst, "tmp", os.path.relpath(src, str(tmp)))
                    os.makedirs(os.path.dirname(out), exist_ok=True)
                    shutil.copyfile(src, out)
                    budget -= sz
    except Exception as exc:  # never mask the test's own failure
        sys.stderr.write("conftest: could not keep the failure's evidence: %r\n" % (exc,))


def pytest_runtest_logreport(report):
    """MK_TEST_DURATIONS=<file>: one line per finished test (seconds, outcome, id), written as the run goes -- a suite that is stopped from
    outside (gpurun's limit) still says where its time went"""
    path = os.environ.get("MK_TEST_DURATIONS")
    if path and report.when == "call":
        try:
            with open(path, "a") as f:
                f.write("%9.3f %s %s\n" % (report.duration, report.outcome, report.nodeid))
        except OSError:
            pass
