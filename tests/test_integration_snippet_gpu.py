"""-m gpu: the C snippet of INTEGRATION.md (table formats of ABI v9) runs and returns 0 = every call answered what the text says."""
import subprocess

import pytest

from test_abi_load import integration_snippet

pytestmark = pytest.mark.gpu


def test_integration_snippet_runs(tmp_path):
    r = subprocess.run([integration_snippet("formats", tmp_path)], capture_output=True, text=True)
    assert r.returncode == 0, (r.returncode, r.stdout, r.stderr)
