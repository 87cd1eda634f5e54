"""Random call sequences on the FrameDecoder surface (init / decode_blocks with every strategy / collect / read / the counters), on mutated
corpus and dictionary frames, zgpu against the oracle call by call: tools/dev/soak_api.py with a fixed seed (the soak runs of a round use
more inputs: profiles/r06/notes.md)."""
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_frame_decoder_call_sequences_match_the_oracle():
    r = subprocess.run([sys.executable, os.path.join(ROOT, "tools", "dev", "soak_api.py"), "600", "11"], capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-2000:]
    assert "disagreements 0" in r.stdout
