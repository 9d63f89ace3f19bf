"""The driver's entry points on a GPU box: what `__graft_entry__.smoke()` asserts has to hold in the suite as well (it once
fell behind the library: the one-launch loop had learnt to run on several blocks and smoke() still expected it not to)."""
import os
import subprocess
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.gpu
def test_smoke_as_the_driver_runs_it():
    """In a process of its own (smoke() creates its own context and loads the checker), from the repo root."""
    p = subprocess.run([sys.executable, "-c", "import __graft_entry__ as g; g.smoke()"], cwd=ROOT, capture_output=True, text=True,
                       timeout=600)
    assert p.returncode == 0, (p.stdout[-2000:], p.stderr[-4000:])
    assert "smoke OK" in p.stdout
