"""GPU: a short run of the randomised differential test (tools/fuzz_parity.py): random k / solid / commonKMers / minAnchors / maxMSA,
depths 1-150, window lengths 60-900, error rates 0-30 % with PacBio-, ONT-, substitution-, insertion- and deletion-only mixes."""
import os
import sys

import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools"))

pytestmark = pytest.mark.gpu


def test_random_configurations_match_the_oracle(monkeypatch, capsys):
    import fuzz_parity

    monkeypatch.setattr(sys, "argv", ["fuzz_parity.py", "20", "12345"])
    assert fuzz_parity.main() == 0
    out = capsys.readouterr().out
    assert "0 differences" in out
