"""CPU: every environment switch the library, the executables or bench.py read is named in INTEGRATION.md §6 (a switch that steers a
measurement has to be findable by whoever reads the number)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_switch_is_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names = set()
    for pat in ("consent_amd/csrc/*.cpp", "consent_amd/csrc/*.h", "consent_amd/cli/*.cpp"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            names |= set(re.findall(r'getenv\("(CW_[A-Z0-9_]+)"\)', open(f).read()))
            names |= set(re.findall(r'knob(?:_u)?\("(CW_[A-Z0-9_]+)"', open(f).read()))
    assert len(names) > 30
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
