"""INTEGRATION.md section B prints "the whole binding for get_codes" as a ctypes stub.  This test takes the stub FROM THE DOCUMENT (the second
python block), runs it in a fresh interpreter and compares what it returns with the oracle -- the text a maintainer copies is the text that
is tested.  The stub of tools/check_integration_stub.py must be that same text."""
import os
import re
import subprocess
import sys

import pytest

from conftest import ROOT, has_gpu

pytestmark = [pytest.mark.gpu, pytest.mark.skipif(not has_gpu(), reason="needs a HIP device")]


def _stub():
    text = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    blocks = re.findall(r"```python\n(.*?)```", text, re.S)
    stub = [b for b in blocks if "slgc_codes" in b and "def get_codes" in b]
    assert len(stub) == 1, "INTEGRATION.md must hold exactly one get_codes stub"
    return stub[0]


def test_the_stub_printed_in_the_document_runs_and_matches_the_oracle(tmp_path):
    check = '''
import os, sys
sys.path.insert(0, os.path.join(%r, "oracle"))
import oracle_c as oc
st = np.random.default_rng(0).integers(0, 256, (26, 20, 33), dtype=np.uint8)
for stack in (st, st.astype(np.float64)):
    h, v = get_codes(stack)
    rh, rv = oc.get_codes(st)
    assert h.dtype == np.int8 and np.array_equal(h, rh) and np.array_equal(v, rv)
try:
    get_codes(np.zeros((10, 4, 4), np.uint8))
    raise SystemExit("expected an error for N < 14")
except RuntimeError as e:
    assert "14" in str(e) or "frames" in str(e), e
print("stub ok")
''' % ROOT
    script = tmp_path / "stub.py"
    script.write_text(_stub() + check)
    r = subprocess.run([sys.executable, str(script)], cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0 and "stub ok" in r.stdout, r.stdout + r.stderr


def test_the_tool_carries_the_same_stub():
    tool = open(os.path.join(ROOT, "tools", "check_integration_stub.py")).read()
    for line in _stub().strip().splitlines():
        assert line in tool, line
