"""The GPU fuzz tool as a test: a short seeded run of all five sections (plain calls over every kernel family / mode / dtype, the fused
products of the 1024 kernel, the batch APIs, the GUI flow on one signal, streaming) must agree with the oracle.  The long runs are
recorded in profiles/r02_fuzz_gpu.txt; this keeps a slice of them in every `pytest -m gpu`."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_short_fuzz_run_agrees_with_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "fuzz_gpu.py"), "160", "20261004"], capture_output=True, text=True,
                       timeout=600, cwd=ROOT)
    tail = "\n".join(r.stdout.splitlines()[-12:]) + "\n" + "\n".join(r.stderr.splitlines()[-8:])
    assert r.returncode == 0, tail
    for what in ("cases agree", "fused-product cases agree", "batch-API cases agree", "GUI-flow cases agree", "streaming cases agree"):
        assert what in r.stdout, tail
    assert "FAIL" not in r.stdout and "EXC" not in r.stdout, tail
