#!/usr/bin/env python3
"""Turn the --pmc passes of tools/prof_round.sh into profiles/pmc_traffic.json (what bench.py reports as roofline.traffic).

    python tools/update_pmc_traffic.py gpurun_out/<tag> [train steps per profiled run, default 4 = --steps 3 --warmup 1]

HBM bytes per launch = (2 * FETCH_SIZE + WRITE_SIZE) KiB * 1024: FETCH_SIZE is doubled as MI355X_MICROARCH.md prescribes for
gfx950 (it reports half of wide coalesced reads; calibrated in round 1 on the LayerNorm kernel: 76.8 MiB reported for a
153.6 MiB read).  The record is stamped with the digest of the kernel sources (libdep_rnn.so.stamp) and the commit; bench.py
refuses it when the kernels changed since."""
import json
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CATS = (('gru2_fwd_fused', 'gru_fwd_sweep'), ('gru2_fwd_df', 'gru_fwd_sweep'), ('gru_fwd_cluster', 'gru_fwd_sweep'), ('gru2_bwd_fused', 'gru_bwd_sweep'),
        ('gru_bwd_cluster', 'gru_bwd_sweep'),
        ('lstm_fwd_cluster', 'lstm_fwd_sweep'), ('lstm_bwd_cluster', 'lstm_bwd_sweep'))


def main(src, steps_total=4):
    out = {'_note': __doc__.split('\n\n')[2].replace('\n', ' '), 'workloads': {}, 'step_bytes': {},
           '_step_bytes_note': 'sum over ALL kernels of (2*FETCH_SIZE + WRITE_SIZE) * launches / train steps of the profiled run '
                               '(bench.py --profile-run: warm-up + timed steps only)'}
    out['kernel_digest'] = open(os.path.join(ROOT, 'icassp2022-depression_amd', 'libdep_rnn.so.stamp')).read().strip()
    out['commit'] = subprocess.run(['git', 'rev-parse', '--short', 'HEAD'], cwd=ROOT, capture_output=True, text=True).stdout.strip()
    for wl in ('audio_gru', 'text_bilstm', 'fusion'):
        f = os.path.join(src, f'pmc_{wl}', 'pmc_per_launch_kb.json')
        if not os.path.exists(f):
            continue
        rec = json.load(open(f))
        cat = {}
        for name, v in rec.items():
            for key, c in CATS:
                if key in name and 'FETCH_SIZE' in v and 'WRITE_SIZE' in v:
                    byts = int((2 * v['FETCH_SIZE'] + v['WRITE_SIZE']) * 1024)
                    if byts > cat.get(c, 0):                 # several template variants of a kernel: keep the dominant one
                        cat[c] = byts
        out['workloads'][wl] = cat
        out['step_bytes'][wl] = int(sum((2 * v.get('FETCH_SIZE', 0) + v.get('WRITE_SIZE', 0)) * 1024 * v.get('launches', 0)
                                        for v in rec.values()) / steps_total)
    json.dump(out, open(os.path.join(ROOT, 'profiles', 'pmc_traffic.json'), 'w'), indent=1)
    print(json.dumps({'per_launch': out['workloads'], 'per_step': out['step_bytes']}, indent=1))


if __name__ == '__main__':
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 4)
