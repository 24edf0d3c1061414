"""Round 4: the GRU backward sweep writes its gate gradients as the pre-split PK image (gemm_bf16x3.hip FMT_PK) and the three
contractions that read them stage that image without converting.  The image holds exactly the (hi, lo) bf16 planes the on-the-fly
split forms, so EVERY gradient must come out BIT-IDENTICAL to the fp32-array path (DEP_DGI_PK=0)."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _run(tmp_path, tag, pk, B, T, F, dx, lstm=False, env=None, nody=False):
    out = str(tmp_path / f'{tag}_{pk}.npz')
    e = dict(os.environ, DEP_DGI_PK=str(pk))
    e.update(env or {})
    r = subprocess.run([sys.executable, os.path.join(HERE, 'pk_probe.py'), out, str(B), str(T), str(F)] + (['dx'] if dx else []) + (['lstm'] if lstm else []) + (['nody'] if nody else []),
                       env=e, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
    return np.load(out)


# B = 416: 26 tiles -> every burst phase 0..3 (the PK flush lags one step for the odd ones); T even, with 0 / 2 steps after the last
# dirty step; F = 64 / 256: both layers' contractions above the three-term kernel's size threshold; dx: the NN form of layer 0 too
# nody: the training step's call form (dpooled only, no dX): gru2_bwd_fused<.., HASDY = false, ..> -- three input slots, prefetch distance 2
@pytest.mark.parametrize('B,T,F,dx,nody', [(416, 20, 64, False, False), (416, 22, 256, True, False), (160, 6, 256, False, True), (40, 2, 256, False, True), (512, 300, 256, False, True)])
def test_pk_gate_gradients_leave_every_gradient_bit_identical(tmp_path, B, T, F, dx, nody):
    a = _run(tmp_path, 'a', 0, B, T, F, dx, nody=nody)
    b = _run(tmp_path, 'b', 1, B, T, F, dx, nody=nody)
    for k in a.files:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))


# Round 5: dW_ih and dW_hh of a GRU layer whose input is as wide as its state (layer 1 always; layer 0 when F = H) are ONE launch over the
# shared PK image (gemm_bf16x3_tn_pair).  Same tiles, K chunks and split-K order per contraction: every gradient bit-identical to the two
# launches (DEP_DW_PAIR=0).  F = 64: only layer 1 pairs; F = 256: both; the benchmark's full shape once.
@pytest.mark.parametrize('B,T,F,dx,nody', [(416, 20, 64, False, False), (416, 22, 256, True, False), (512, 300, 256, False, True)])
def test_paired_weight_gradient_launch_leaves_every_gradient_bit_identical(tmp_path, B, T, F, dx, nody):
    a = _run(tmp_path, 'a', 1, B, T, F, dx, env={'DEP_DW_PAIR': '0'}, nody=nody)
    b = _run(tmp_path, 'b', 1, B, T, F, dx, env={'DEP_DW_PAIR': '1'}, nody=nody)
    for k in a.files:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))


# the BiLSTM-128 x2 stack (both directions in one launch; the reverse direction's step pairs are rows (ka, ka + 1)): B = 416 -> 26 tiles x 2
# directions, burst phases 0..3 again; cfg3's full shape once
@pytest.mark.parametrize('B,T,F,dx', [(416, 20, 64, True), (416, 22, 1024, False), (512, 300, 1024, False)])
def test_pk_gate_gradients_of_the_bilstm_stack_are_bit_identical_too(tmp_path, B, T, F, dx):
    a = _run(tmp_path, 'a', 0, B, T, F, dx, lstm=True)
    b = _run(tmp_path, 'b', 1, B, T, F, dx, lstm=True)
    for k in a.files:
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))


# Round 6: the weight-gradient contractions (gemm_bf16x3_tn_dma) and the input projections (gemm_bf16x3_nt_dma) are fed by LDS-DMA where the shape
# fits (256 x 256 / 128 x 256 tiles, K chunks of 16 rows).  Same K chunks, same order of k, same three products per 16 k in the same order: every output
# and every gradient bit-identical to the register-staged kernels (DEP_GEMM_TN_DMA=0 DEP_GEMM_NT_DMA=0) -- and the DMA kernels must really have run.
# (the projection kernel wants B T % 128 == 0 and an unsplit contraction, the weight-gradient kernel a split-K one with chunks of 16 rows)
@pytest.mark.parametrize('B,T,F,dx,nody,lstm,tn', [(512, 300, 256, False, True, False, True), (384, 24, 256, True, False, False, True), (64, 32, 256, False, True, False, True),
                                                   (512, 300, 1024, False, False, True, True), (128, 32, 1024, True, False, True, True)])
def test_dma_fed_contractions_leave_every_output_and_gradient_bit_identical(tmp_path, B, T, F, dx, nody, lstm, tn):
    a = _run(tmp_path, 'a', 1, B, T, F, dx, lstm=lstm, env={'DEP_GEMM_TN_DMA': '0', 'DEP_GEMM_NT_DMA': '0', 'PROBE_INSTANCES': '1'}, nody=nody)
    b = _run(tmp_path, 'b', 1, B, T, F, dx, lstm=lstm, env={'PROBE_INSTANCES': '1'}, nody=nody)
    ia, ib = ' '.join(a['instances']), ' '.join(b['instances'])
    assert 'nt_dma' not in ia and 'tn_dma' not in ia
    assert 'gemm_bf16x3_nt_dma' in ib and ('gemm_bf16x3_tn_dma' in ib) == tn, ib
    for k in a.files:
        if k == 'instances':
            continue
        assert np.isfinite(a[k]).all(), k
        assert np.array_equal(a[k], b[k]), (k, float(np.abs(a[k] - b[k]).max()))


# Round 6: the superseded forms round 5 kept for these A/Bs (flag hand-off of the fused forward, LDS-plane BiLSTM forward, burst-stream BiLSTM
# backward, reduce-scatter GRU backward) are deleted.  The same guarantee -- nothing of the arithmetic moved -- is held against
# tests/golden/device_bits.json: sha256 of every gradient / output of these seeded runs, recorded on an MI355X with the round-5 library whose
# defaults GPUTEST_r05 had asserted bit-identical to those forms (tests/golden/make_device_bits.py).  Every later kernel rewrite (GEMM operand
# staging, backward hand-off) has to reproduce the record.
sys.path.insert(0, os.path.join(HERE, 'golden'))
import make_device_bits as _bits  # noqa: E402


_BITS_NOW = {}


@pytest.mark.parametrize('case', _bits.CASES, ids=[c[0] for c in _bits.CASES])
def test_gradients_are_bit_identical_to_the_recorded_device_bits(case):
    rec = json.load(open(os.path.join(HERE, 'golden', 'device_bits.json')))['cases'][case[0]]
    if not _BITS_NOW:                                 # all sixteen cases in two pk_probe.py processes (one per environment), once per session
        _BITS_NOW.update(_bits.all_digests())
    got = _BITS_NOW[case[0]]
    assert sorted(got) == sorted(rec)
    bad = [k for k in got if got[k] != rec[k]]
    assert not bad, bad


@pytest.mark.parametrize('form', [['nody']])
def test_16bit_saved_gates_move_no_gradient_by_more_than_1e4_of_its_scale(tmp_path, form):
    """Round 4: the GRU stack saves r, z (unorm16) and n (snorm16) as 16-bit fixed point (|error| <= 7.6e-6 / 1.5e-5) instead of fp32.
    Forward outputs cannot change (the gates are only stored for the backward); every gradient of the cfg2-shaped stack must stay
    within 1e-4 of its tensor's scale of the fp32-gates run (measured: a few 1e-6), and the contractions' operands stay bit-exact
    functions of the gate gradients (PK on in both runs)."""
    outs = {}
    for sv in ('0', '1'):
        out = str(tmp_path / f'sv{sv}.npz')
        e = dict(os.environ, DEP_SV16=sv)
        r = subprocess.run([sys.executable, os.path.join(HERE, 'pk_probe.py'), out, '512', '300', '256'] + form, env=e, capture_output=True,
                           text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[sv] = np.load(out)
    worst = 0.0
    for k in outs['0'].files:
        a, b = outs['0'][k].astype(np.float64), outs['1'][k].astype(np.float64)
        rel = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
        worst = max(worst, rel)
        assert rel < 1e-4, (k, rel)
    assert worst > 0.0            # the switch really changed the stored gates


@pytest.mark.parametrize('form', [['nody']])
def test_bf16_storage_mode_has_its_own_tolerance(tmp_path, form):
    """dep_set_gemm_mode(3) / DEP_GEMM_MODE=bf16s (BASELINE configs[1]'s "bf16", a labelled throughput mode, NEVER the parity path): the
    hidden sequences, hn and the gate gradients live in HBM as bf16 (the saved gates as 16-bit fixed point), state / accumulation /
    the recurrence stay fp32.  Against mode 2 (single bf16 products, fp32 storage) the FORWARD must be bit-identical -- only what is
    stored for the backward changes -- and every gradient within 1e-2 of its tensor's scale (bf16 has 8 significant bits; measured ~2e-3);
    and the mode must really be in effect (some gradient differs)."""
    outs = {}
    for mode in ('bf16', 'bf16s'):
        out = str(tmp_path / f'{mode}.npz')
        e = dict(os.environ, DEP_GEMM_MODE=mode)
        r = subprocess.run([sys.executable, os.path.join(HERE, 'pk_probe.py'), out, '512', '300', '256'] + form, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[mode] = np.load(out)
    assert np.array_equal(outs['bf16']['pooled'], outs['bf16s']['pooled'])
    worst = 0.0
    for k in outs['bf16'].files:
        a, b = outs['bf16'][k].astype(np.float64), outs['bf16s'][k].astype(np.float64)
        assert np.isfinite(b).all(), k
        rel = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
        worst = max(worst, rel)
        assert rel < 1e-2, (k, rel)
    assert worst > 1e-5


@pytest.mark.parametrize('form', [['lstm', 'dx'], ['nody']])
def test_bf16_product_mode_reads_the_pk_image_and_keeps_its_tolerance(tmp_path, form):
    """dep_set_gemm_mode(2) / DEP_GEMM_MODE=bf16 (single bf16 products in the time-parallel contractions, fp32 storage; the labelled throughput
    line of extra.other_workloads.*.bf16_products and extra.bf16_products, NEVER the parity path).  Round 6: the contractions read the gate
    gradients through the hi rows of the sweep's PK image instead of converting fp32 rows (cfg3's weight gradients 2.66 -> 2.32 ms per step in
    this mode -- still slower than the three-term path's 1.72, which has the paired launches; the mode is a label, not a tuned path).  Against the three-term default: the forward output bit-identical for the GRU stack's pooled output only where no GEMM feeds it --
    so only finiteness is asked of it here -- and every gradient within 1e-2 of its tensor's scale (bf16: 8 significant bits; measured ~2e-3),
    with the mode really in effect.  DEP_DGI_PK=0 (fp32 gate-gradient rows, converted while staging) must give the SAME bits in this mode:
    the hi rows of the image are exactly the values that conversion forms."""
    outs = {}
    for tag, env in (('x3', {}), ('bf16', {'DEP_GEMM_MODE': 'bf16'}), ('bf16_rows', {'DEP_GEMM_MODE': 'bf16', 'DEP_DGI_PK': '0'})):
        out = str(tmp_path / f'{tag}.npz')
        e = dict(os.environ); e.update(env)
        shape = ['416', '22', '1024'] if 'lstm' in form else ['512', '300', '256']      # (the BiLSTM case at a short T: its contractions are above the split threshold there too)
        r = subprocess.run([sys.executable, os.path.join(HERE, 'pk_probe.py'), out] + shape + form, env=e, capture_output=True, text=True, timeout=600)
        assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-3000:]
        outs[tag] = np.load(out)
    worst = 0.0
    for k in outs['x3'].files:
        a, b = outs['x3'][k].astype(np.float64), outs['bf16'][k].astype(np.float64)
        assert np.isfinite(b).all(), k
        rel = np.abs(a - b).max() / max(np.abs(a).max(), 1e-30)
        worst = max(worst, rel)
        # (gradients: 1e-2; the forward's final state h_n / pooled output has passed T = 300 recurrent steps on single-product projections: 5e-2)
        assert rel < (5e-2 if k in ('h_n', 'pooled') else 1e-2), (k, rel)
        assert np.array_equal(outs['bf16'][k], outs['bf16_rows'][k]), k
    assert worst > 1e-5
