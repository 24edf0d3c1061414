"""Every environment switch the library reads is documented: each DEP_* name passed to getenv() in csrc/ or read from os.environ in the host
modules / bench.py appears in INTEGRATION.md's switch table (a switch that changes which kernel runs must not be findable only in the source)."""
import os
import re

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _names():
    names = set()
    csrc = os.path.join(ROOT, 'icassp2022-depression_amd', 'csrc')
    for f in os.listdir(csrc):
        if f.endswith(('.hip', '.h')):
            names |= set(re.findall(r'getenv\("(DEP_[A-Z0-9_]+)"\)', open(os.path.join(csrc, f)).read()))
    host = [os.path.join(ROOT, 'icassp2022-depression_amd', f) for f in os.listdir(os.path.join(ROOT, 'icassp2022-depression_amd')) if f.endswith('.py')]
    for f in host + [os.path.join(ROOT, 'bench.py')]:
        names |= set(re.findall(r"environ(?:\.get\(|\[)\s*['\"](DEP_[A-Z0-9_]+)", open(f).read()))
    return names


def test_every_environment_switch_is_listed_in_integration_md():
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    names = _names()
    assert len(names) >= 25         # the scan itself works
    missing = sorted(n for n in names if n not in doc)
    assert not missing, missing


def test_no_documented_switch_is_dead():
    """... and the other way round: INTEGRATION.md's switch table names no DEP_* variable that nothing reads any more."""
    doc = open(os.path.join(ROOT, 'INTEGRATION.md')).read()
    table = doc[doc.index('## 5. Environment switches'):]
    table = table[:table.index('\n## ', 5)] if '\n## ' in table[5:] else table
    listed = set(re.findall(r'`(DEP_[A-Z0-9_]+)', table))
    dead = sorted(listed - _names())
    assert not dead, dead
