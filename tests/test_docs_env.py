"""CPU: every environment switch the library, the executables or bench.py read is named in INTEGRATION.md §6 (a switch that steers a
measurement has to be findable by whoever reads the number)."""
import glob
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_every_environment_switch_is_documented():
    doc = open(os.path.join(ROOT, "INTEGRATION.md")).read()
    names, product = set(), set()
    for pat in ("consent_amd/csrc/*.cpp", "consent_amd/csrc/*.h", "consent_amd/cli/*.cpp"):
        for f in glob.glob(os.path.join(ROOT, pat)):
            src = open(f).read()
            runtime = set(re.findall(r'[^_]getenv\("(CW_[A-Z0-9_]+)"\)', src))
            aids = set(re.findall(r'CW_AID_ENV\("(CW_[A-Z0-9_]+)"\)', src)) | set(re.findall(r'knob(?:_u)?\("(CW_[A-Z0-9_]+)"', src))
            names |= runtime | aids
            product |= runtime
    assert len(names) > 30
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing
    # the product library itself reads a dozen switches at most: everything else is a constant there (csrc/cw_env.h)
    assert len(product) <= 12, sorted(product)
