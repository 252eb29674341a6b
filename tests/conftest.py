import os
import sys

import pytest

# the library's experiment / test switches (MK_PREFILTER_PATH, MK_TEST_ENTRY_BASE, MK_INDEX_BUILD, MK_CLI_BATCH_NT ...) are read only under
# MK_DEBUG=1 (metaeuk_amd/csrc/mk_host.cpp: knob); the tests use them to force paths, so the whole session -- and every process it starts -- has it
os.environ["MK_DEBUG"] = "1"

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture(scope="session")
def small_workload():
    from metaeuk_amd import synth
    return synth.make_workload(12, 200, seed=7)


@pytest.fixture(scope="session")
def gpu_api():
    from metaeuk_amd import api
    api.init(0)
    return api


@pytest.fixture(autouse=True)
def _drop_large_scratch_files(request, tmp_path_factory):
    """pytest keeps every test's tmp_path until the session ends; the k = 7 index DBs (10 GB of list offsets each) and the scale tests' sequence DBs
    would pile up to more than the 79 GB of a GPU box's scratch disk and starve the tests behind them (round 5: the 60 M-protein split test was
    skipped for want of 33 GB).  After every test: files of more than 256 MB under the session's temp directory are removed."""
    yield
    try:
        base = str(tmp_path_factory.getbasetemp())
    except Exception:
        return
    for root, _, files in os.walk(base):
        for f in files:
            p = os.path.join(root, f)
            try:
                if os.path.isfile(p) and not os.path.islink(p) and os.path.getsize(p) > (256 << 20):
                    os.remove(p)
            except OSError:
                pass
