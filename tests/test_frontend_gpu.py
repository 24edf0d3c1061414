"""Feature front-end on the GPU: wav2vlad's two halves through the C-ABI against the oracle
(oracle/ref_frontend.py; parity unpinned -- the reference has no fixture for this path and its dependencies are absent)."""
import os
import wave

import numpy as np
import pytest

from oracle import ref_frontend as RF

torch = pytest.importorskip('torch')
pytestmark = pytest.mark.gpu

if torch.cuda.is_available():
    from icassp2022_depression_amd import audio_features_whole as m


def _speechlike(rng, n, sr):
    """Broadband test signal at int16 scale (like np.frombuffer(..., np.short)): noise shaped by a few resonances + tones."""
    t = np.arange(n) / sr
    y = rng.standard_normal(n) * 800.0
    for f0, a in ((180.0, 3000.0), (950.0, 2000.0), (2600.0, 1500.0), (5100.0, 600.0)):
        y += a * np.sin(2 * np.pi * f0 * t + rng.uniform(0, 6.28)) * (0.6 + 0.4 * np.sin(2 * np.pi * 3.0 * t))
    return np.round(y).clip(-32768, 32767)


@pytest.mark.parametrize('sr,seconds', [(16000, 2.0), (22050, 1.3), (8000, 5.0)])
def test_log_mel_matches_oracle(sr, seconds):
    rng = np.random.default_rng(int(sr + seconds * 10))
    y = _speechlike(rng, int(sr * seconds) + 37, sr)            # length not a multiple of the hop
    lm = m.log_melspectrogram(y, sr).cpu().numpy().astype(np.float64)
    ref = RF.log_melspectrogram(y, sr)
    assert lm.shape == ref.shape
    # fp32 DFT by GEMM: relative error of a mel power ~1e-5 of the frame's total power; the signal is broadband, so every
    # band holds a fair share of it and the log differs by < 1e-3
    assert np.abs(lm - ref).max() < 1e-3, np.abs(lm - ref).max()


def test_log_floor_applies():
    sr = 16000
    y = np.zeros(sr); y[:8] = [1e-3, -1e-3, 2e-3, 0, 0, 1e-3, 0, -2e-3]       # almost silent: mel power far below the floor
    lm = m.log_melspectrogram(y, sr).cpu().numpy()
    assert np.allclose(lm[5:], np.log(1e-6), atol=1e-6)


def test_netvlad_matches_oracle():
    rng = np.random.default_rng(3)
    N, F, K, D = 63, 80, 16, 256
    x = (rng.standard_normal((N, F)) * 3.0 + 5.0).astype(np.float32)
    layer = m.NetVLAD(F, N, K, D, seed=11)
    W = {k: v.cpu().numpy().astype(np.float64) for k, v in layer.weights.items()}
    out = layer(torch.from_numpy(x).cuda()).cpu().numpy()
    ref = RF.netvlad(x.astype(np.float64), W)
    assert out.shape == (1, D)
    assert np.abs(out - ref).max() < 1e-5 * max(1.0, np.abs(ref).max())
    # same seed -> same layer, another seed -> another (the reference draws fresh weights per call)
    assert torch.equal(m.NetVLAD(F, N, K, D, seed=11).weights['hidden1_weights'], layer.weights['hidden1_weights'])
    assert not torch.equal(m.NetVLAD(F, N, K, D, seed=12).weights['hidden1_weights'], layer.weights['hidden1_weights'])


def test_wav2vlad_and_extract_features_on_wav_files(tmp_path):
    """Reference lines 74-131 end to end on a synthetic corpus: three responses per volunteer -> (3, 1, 256) features, the
    loader of the training scripts (squeeze axis 2) accepts the saved file, and wav2vlad equals the oracle on the same
    weights."""
    rng = np.random.default_rng(5)
    sr = 16000
    for part, ids in (('Data', (1, 2)), ('ValidationData', (1,))):
        for i in ids:
            d = tmp_path / part / str(i); os.makedirs(d)
            for name in ('positive_out.wav', 'neutral_out.wav', 'negative_out.wav'):
                y = _speechlike(rng, sr + int(rng.integers(0, 4000)), sr).astype(np.int16)
                w = wave.open(str(d / name), 'wb'); w.setnchannels(1); w.setsampwidth(2); w.setframerate(sr); w.writeframes(y.tobytes()); w.close()
            (d / 'new_label.txt').write_text('%d\n' % (40 + i))
    feats, targs = m.main(str(tmp_path), n=3)
    assert len(feats) == 3 and targs == [41.0, 42.0, 41.0]
    arr = np.load(tmp_path / 'Features/AudioWhole/whole_samples_reg_256.npz')['arr_0']
    assert arr.shape == (3, 3, 1, 256) and np.squeeze(arr, axis=2).shape == (3, 3, 256) and np.isfinite(arr).all()
    y = _speechlike(rng, sr * 2, sr)
    layer = m.NetVLAD(80, 1 + len(y) // 512, 16, 256, seed=2)
    W = {k: v.cpu().numpy().astype(np.float64) for k, v in layer.weights.items()}
    got = m.wav2vlad(y, sr, weights=layer.weights)
    ref = RF.wav2vlad(y, sr, W)
    assert np.abs(got - ref).max() < 2e-4 * max(1.0, np.abs(ref).max())
