"""Oracle of the audio feature front-end  --  TEST INFRASTRUCTURE ONLY, **parity unpinned**.

Reference call site: Classification/audio_features_whole.py:57-72 (`wav2vlad`):

    melspec = librosa.feature.melspectrogram(signal, n_mels=80, sr=sr).astype(np.float32).T
    melspec = np.log(np.maximum(1e-6, melspec))
    feat = lpk.NetVLAD(feature_size=80, max_samples=frames, cluster_size=16, output_dim=256)(melspec)

Both halves live in third-party packages that are ABSENT from /root/reference and from this image (no network):
  * librosa (version not pinned by the reference; the 0.7-0.9 defaults are restated: n_fft 2048, hop 512, periodic Hann
    window, center=True with reflect padding, power 2, Slaney-style mel filters: `htk=False`, area normalisation);
  * loupe_keras.NetVLAD (antoine77340/LOUPE, Keras port; not pinned): soft-assignment VLAD with a cluster bias, the
    `cluster_weights2` centres, intra-cluster then global L2 normalisation and the `hidden1_weights` projection.
The reference holds no test, fixture or saved weight for this path (the NetVLAD layer is freshly random-initialised inside
every wav2vlad call), so nothing exists to pin this restatement to: it follows the packages' published algorithms, and the
HIP path is compared with THIS file only.  fp64 numpy.
"""
import numpy as np


# ----------------------------------------------------------------------------- librosa.feature.melspectrogram
def hann_periodic(n):
    """scipy.signal.get_window('hann', n, fftbins=True), librosa's default window."""
    return 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(n) / n)


def frames_centered(y, n_fft=2048, hop=512):
    """librosa.stft framing with center=True, pad_mode='reflect'."""
    y = np.asarray(y, np.float64)
    ypad = np.pad(y, n_fft // 2, mode='reflect')
    n_frames = 1 + (len(ypad) - n_fft) // hop
    idx = np.arange(n_fft)[None, :] + hop * np.arange(n_frames)[:, None]
    return ypad[idx]


def hz_to_mel(f):
    """Slaney's auditory-toolbox scale (librosa htk=False): linear below 1 kHz, logarithmic above."""
    f = np.asarray(f, np.float64)
    f_sp = 200.0 / 3
    mels = f / f_sp
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(f >= min_log_hz, min_log_mel + np.log(np.maximum(f, 1e-30) / min_log_hz) / logstep, mels)


def mel_to_hz(m):
    m = np.asarray(m, np.float64)
    f_sp = 200.0 / 3
    min_log_hz = 1000.0
    min_log_mel = min_log_hz / f_sp
    logstep = np.log(6.4) / 27.0
    return np.where(m >= min_log_mel, min_log_hz * np.exp(logstep * (m - min_log_mel)), f_sp * m)


def mel_filterbank(sr, n_fft=2048, n_mels=80, fmin=0.0, fmax=None):
    """librosa.filters.mel(sr, n_fft, n_mels, htk=False, norm='slaney') -> (n_mels, 1 + n_fft // 2)."""
    fmax = sr / 2.0 if fmax is None else fmax
    fftfreqs = np.linspace(0, sr / 2.0, 1 + n_fft // 2)
    mel_f = mel_to_hz(np.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n_mels + 2))
    fdiff = np.diff(mel_f)
    ramps = mel_f[:, None] - fftfreqs[None, :]
    W = np.zeros((n_mels, 1 + n_fft // 2))
    for i in range(n_mels):
        lower = -ramps[i] / fdiff[i]
        upper = ramps[i + 2] / fdiff[i + 1]
        W[i] = np.maximum(0, np.minimum(lower, upper))
    enorm = 2.0 / (mel_f[2:n_mels + 2] - mel_f[:n_mels])
    return W * enorm[:, None]


def log_melspectrogram(y, sr, n_fft=2048, hop=512, n_mels=80, floor=1e-6):
    """np.log(np.maximum(1e-6, librosa.feature.melspectrogram(y, sr, n_mels=80).T))  -> (frames, n_mels)."""
    fr = frames_centered(y, n_fft, hop) * hann_periodic(n_fft)[None, :]
    spec = np.fft.rfft(fr, axis=1)
    power = spec.real ** 2 + spec.imag ** 2
    mel = power @ mel_filterbank(sr, n_fft, n_mels).T
    return np.log(np.maximum(floor, mel))


# ----------------------------------------------------------------------------- loupe_keras.NetVLAD
def netvlad(x, W):
    """x: (N, F) frames.  W: dict with cluster_weights (F,K), cluster_biases (K), cluster_weights2 (F,K), hidden1_weights (K*F, D).
    Returns (1, D)."""
    x = np.asarray(x, np.float64)
    act = x @ W['cluster_weights'] + W['cluster_biases']
    act = np.exp(act - act.max(1, keepdims=True)); act /= act.sum(1, keepdims=True)       # softmax over clusters
    a_sum = act.sum(0, keepdims=True)                                                       # (1, K)
    a = a_sum * W['cluster_weights2']                                                       # (F, K)
    vlad = (act.T @ x).T - a                                                                # (F, K)
    vlad = vlad / np.sqrt(np.maximum((vlad ** 2).sum(0, keepdims=True), 1e-12))             # tf.nn.l2_normalize over the feature axis
    flat = vlad.reshape(1, -1)                                                              # row-major (F, K): index f*K + k
    flat = flat / np.sqrt(np.maximum((flat ** 2).sum(), 1e-12))
    return flat @ W['hidden1_weights']


def wav2vlad(wave, sr, W):
    return netvlad(log_melspectrogram(wave, sr), W)
