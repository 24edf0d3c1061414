"""Device-bit anchors (NOT reference-derived fixtures: a self-regression record).

Runs tests/pk_probe.py (one seeded forward + backward of the GRU-256 x2 / BiLSTM-128 x2 stack through the C-ABI) for the cases below and
writes the sha256 of every output array to tests/golden/device_bits.json.  Recorded ONCE on an MI355X with the round-5 library (commit
aaf8e26: `DEP_LIB_PATH=<that build> python tests/golden/make_device_bits.py`), whose default forms GPUTEST_r05 asserted bit-identical to
the forms they replaced (flag hand-off, LDS-plane BiLSTM forward, burst-stream BiLSTM backward, reduce-scatter GRU backward).  Round 6
deleted those superseded forms; tests/test_presplit_gpu.py::test_gradients_are_bit_identical_to_the_recorded_device_bits keeps the same
guarantee against this record instead, and every kernel rewrite of round 6 (GEMM staging, sentinel backward hand-off) has to reproduce it.
The arithmetic is deterministic on gfx950 (fixed split-K order, fixed member-order sums, counter-based dropout masks)."""
import hashlib
import json
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
TESTS = os.path.dirname(HERE)

# (name, B, T, F, flags, env)
CASES = [
    ('gru_416x20x64', 416, 20, 64, [], {}),
    ('gru_416x22x256_dx', 416, 22, 256, ['dx'], {}),
    ('gru_160x6x256', 160, 6, 256, [], {}),
    ('gru_160x6x256_dx', 160, 6, 256, ['dx'], {}),
    ('gru_512x300x256', 512, 300, 256, [], {}),
    ('gru_416x20x64_nody', 416, 20, 64, ['nody'], {}),
    ('gru_160x6x256_nody', 160, 6, 256, ['nody'], {}),
    ('gru_40x2x256_nody', 40, 2, 256, ['nody'], {}),
    ('gru_512x300x256_nody', 512, 300, 256, ['nody'], {}),
    ('gru_512x300x256_nody_perlayer', 512, 300, 256, ['nody'], {'DEP_FUSED2_BWD': '0'}),
    ('lstm_416x5x64', 416, 5, 64, ['lstm'], {}),
    ('lstm_416x22x64_dx', 416, 22, 64, ['lstm', 'dx'], {}),
    ('lstm_416x22x64', 416, 22, 64, ['lstm'], {}),
    ('lstm_416x6x64_dx', 416, 6, 64, ['lstm', 'dx'], {}),
    ('lstm_416x22x1024', 416, 22, 1024, ['lstm'], {}),
    ('lstm_512x300x1024', 512, 300, 1024, ['lstm'], {}),
]


def _sha(path, name):
    z = np.load(path)
    res = {}
    for k in sorted(z.files):
        a = np.ascontiguousarray(z[k])
        assert np.isfinite(a).all(), (name, k)
        res[k] = hashlib.sha256(a.tobytes()).hexdigest()[:24]
    return res


def all_digests(cases=None):
    """{case name: {array: sha256}} of `cases` (default: all) -- the cases that share an environment run in ONE pk_probe.py process."""
    cases = CASES if cases is None else cases
    by_env = {}
    for c in cases:
        by_env.setdefault(tuple(sorted(c[5].items())), []).append(c)
    rec = {}
    with tempfile.TemporaryDirectory() as d:
        for env, group in by_env.items():
            plan = [[os.path.join(d, c[0] + '.npz'), c[1], c[2], c[3], c[4]] for c in group]
            pf = os.path.join(d, 'plan_%d.json' % len(rec))
            json.dump(plan, open(pf, 'w'))
            e = dict(os.environ); e.update(dict(env))
            r = subprocess.run([sys.executable, os.path.join(TESTS, 'pk_probe.py'), '--batch', pf], env=e, capture_output=True, text=True, timeout=1200)
            assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
            for c, pl in zip(group, plan):
                rec[c[0]] = _sha(pl[0], c[0])
    return rec


def digests(case):
    return all_digests([case])[case[0]]


if __name__ == '__main__':
    rec = all_digests()
    dst = sys.argv[1] if len(sys.argv) > 1 else os.path.join(HERE, 'device_bits.json')
    json.dump({'library': os.environ.get('DEP_LIB_PATH', 'in-tree'), 'cases': rec}, open(dst, 'w'), indent=1, sort_keys=True)
    print('wrote', dst, len(rec), 'cases')
