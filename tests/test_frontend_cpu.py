"""Feature front-end on CPU: the oracle's restatement of librosa's log-mel spectrogram and LOUPE's NetVLAD
checked through properties the published algorithms guarantee (there is no reference fixture for this path: parity
unpinned), and the product's host-side tables against the oracle's independent construction."""
import numpy as np

from oracle import ref_frontend as RF


def test_mel_filterbank_properties():
    sr, n_fft, n_mels = 16000, 2048, 80
    W = RF.mel_filterbank(sr, n_fft, n_mels)
    assert W.shape == (n_mels, 1 + n_fft // 2) and (W >= 0).all()
    freqs = np.linspace(0, sr / 2, 1 + n_fft // 2)
    peaks = freqs[W.argmax(1)]
    assert (np.diff(peaks) > 0).all()                                     # centres increase
    edges = RF.mel_to_hz(np.linspace(RF.hz_to_mel(0.0), RF.hz_to_mel(sr / 2.0), n_mels + 2))
    # Slaney area normalisation: every triangle integrates to ~1 over frequency
    area = (W * (freqs[1] - freqs[0])).sum(1)
    assert np.allclose(area[10:], 1.0, atol=0.08), area[:12]
    # the scale is linear below 1 kHz (200/3 Hz per mel) and logarithmic above; the two maps are inverse
    assert np.allclose(RF.mel_to_hz(RF.hz_to_mel(np.array([10.0, 500.0, 999.0, 1000.0, 4000.0, 7999.0]))), [10, 500, 999, 1000, 4000, 7999])
    assert abs(RF.hz_to_mel(1000.0) - 15.0) < 1e-12 and np.all(np.diff(edges) > 0)


def test_log_mel_of_a_pure_tone_peaks_at_the_tone():
    sr = 16000
    t = np.arange(sr) / sr
    f0 = 1234.0
    y = 1000.0 * np.sin(2 * np.pi * f0 * t)
    lm = RF.log_melspectrogram(y, sr)
    assert lm.shape == (1 + len(y) // 512, 80)
    W = RF.mel_filterbank(sr)
    freqs = np.linspace(0, sr / 2, 1025)
    expect = int(np.argmax(W[:, np.argmin(np.abs(freqs - f0))]))
    assert abs(int(lm[10].argmax()) - expect) <= 1
    # Parseval on one frame: sum of the one-sided power spectrum ~ n_fft * sum(frame^2) / 2 (+ DC / Nyquist terms)
    fr = RF.frames_centered(y)[10] * RF.hann_periodic(2048)
    spec = np.fft.rfft(fr)
    p = spec.real ** 2 + spec.imag ** 2
    assert np.isclose(2 * p.sum() - p[0] - p[-1], 2048 * (fr ** 2).sum(), rtol=1e-9)


def test_netvlad_invariants():
    rng = np.random.default_rng(0)
    N, F, K, D = 50, 80, 16, 256
    x = rng.standard_normal((N, F))
    W = {'cluster_weights': rng.standard_normal((F, K)) / np.sqrt(F), 'cluster_biases': rng.standard_normal(K) / np.sqrt(F),
         'cluster_weights2': rng.standard_normal((F, K)) / np.sqrt(F), 'hidden1_weights': np.eye(K * F)[:, :D]}
    out = RF.netvlad(x, W)
    assert out.shape == (1, D)
    # with an identity projection the output is a slice of the globally normalised VLAD: norm <= 1, and permuting the frames
    # (VLAD is an orderless pooling) changes nothing
    assert np.linalg.norm(out) <= 1.0 + 1e-12
    assert np.allclose(RF.netvlad(x[rng.permutation(N)], W), out, atol=1e-12)


def test_product_mel_table_equals_oracle():
    from icassp2022_depression_amd import audio_features_whole as m
    for sr in (8000, 16000, 22050, 44100):
        assert np.abs(m.mel_filters(sr, 2048, 80) - RF.mel_filterbank(sr, 2048, 80)).max() < 1e-12


def test_power_spectrogram_equals_scipy_stft():
    """The STFT half of the restatement (reflect-padded centred frames, periodic Hann, |rfft|^2) against an INDEPENDENT
    implementation that is in the image: scipy.signal.stft.  This pins the framing / window / power
    conventions to scipy's, not to the reference -- librosa and LOUPE are importable neither here nor in /root/reference, so
    row f4 stays "parity unpinned"; what scipy cannot vouch for is the Slaney mel table and NetVLAD."""
    import pytest
    scipy_signal = pytest.importorskip("scipy.signal")
    from oracle import ref_frontend as RF
    rng = np.random.default_rng(4)
    sr, n_fft, hop = 16000, 2048, 512
    y = rng.standard_normal(sr // 2 + 333)
    fr = RF.frames_centered(y, n_fft, hop) * RF.hann_periodic(n_fft)[None, :]
    power = np.abs(np.fft.rfft(fr, axis=1)) ** 2                         # what log_melspectrogram forms before the mel table
    # scipy: same centring when the signal is reflect-padded by hand (boundary=None, padded=False), 'hann' = periodic (fftbins=True)
    ypad = np.pad(y, n_fft // 2, mode='reflect')
    _, _, Z = scipy_signal.stft(ypad, fs=sr, window='hann', nperseg=n_fft, noverlap=n_fft - hop, nfft=n_fft, boundary=None,
                                padded=False, return_onesided=True, scaling='spectrum')
    win = scipy_signal.get_window('hann', n_fft, fftbins=True)
    assert np.allclose(win, RF.hann_periodic(n_fft), atol=1e-15)
    Zs = Z * win.sum()                                                    # scaling='spectrum' divides by sum(window)
    assert Zs.shape == (n_fft // 2 + 1, power.shape[0])
    got = np.abs(Zs.T) ** 2
    assert np.abs(got - power).max() <= 1e-9 * power.max()
