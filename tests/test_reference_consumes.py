"""Build-container only (skipped where /root/reference is absent, e.g. on the GPU box): the REAL reference
(src.dataset.activations.init_sae_from_checkpoint, src.scripts.train_sae.load_checkpoint) consumes checkpoints written
by the HIP engine on an MI355X (tests/golden/engine_ckpt/, produced by tools/make_engine_checkpoints.py)."""
import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.mark.skipif(not os.path.isdir("/root/reference/src"), reason="needs the reference checkout (build container)")
def test_reference_loads_engine_written_checkpoints():
    import subprocess
    out = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "reference_consumes_checkpoint.py")],
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-3000:] + out.stderr[-3000:]
    assert "RESULT: PASS" in out.stdout
