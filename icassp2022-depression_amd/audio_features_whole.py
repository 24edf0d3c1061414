"""Audio feature front-end on the GPU: `wav2vlad` of Classification/audio_features_whole.py:57-72 --
log-mel spectrogram (librosa.feature.melspectrogram defaults + log) and NetVLAD pooling (loupe_keras.NetVLAD) -- plus the
per-volunteer driver `extract_features` (reference lines 74-114).  Arithmetic runs in libdep_rnn.so (dep_frame_window,
dep_gemm_f32, dep_power_spectrum, dep_log_floor, dep_row_softmax, dep_colsum, dep_vlad_normalize); host code builds the
constant tables once (DFT basis, mel filterbank) and owns the NetVLAD weights.

librosa and loupe_keras are third-party packages the reference imports but does not ship; neither is in this image.  Their
published algorithms are restated (oracle/ref_frontend.py names what was assumed); the reference keeps no fixture for this
path, and its NetVLAD layer is freshly random-initialised on every call, so outputs are reproducible here only through
`NetVLAD(seed=...)` -- parity for this row is UNPINNED by construction.
"""
import math
import os
import wave

import numpy as np
import torch

from . import _lib as L

cluster_size = 16
prefix = os.path.abspath(os.path.join(os.getcwd(), '.'))
min_len = 100
max_len = -1

_tables = {}


def _hz_to_mel(f):
    f = np.asarray(f, np.float64)
    lin = 3.0 * f / 200.0
    return np.where(f >= 1000.0, 15.0 + 27.0 * np.log(np.maximum(f, 1e-30) / 1000.0) / math.log(6.4), lin)


def _mel_to_hz(m):
    m = np.asarray(m, np.float64)
    return np.where(m >= 15.0, 1000.0 * np.exp(math.log(6.4) / 27.0 * (m - 15.0)), 200.0 * m / 3.0)


def mel_filters(sr, n_fft=2048, n_mels=80):
    """Triangular Slaney-normalised mel filters (n_mels, 1 + n_fft/2), `librosa.filters.mel(sr, n_fft, n_mels)`."""
    freqs = np.linspace(0.0, sr / 2.0, 1 + n_fft // 2)
    edges = _mel_to_hz(np.linspace(_hz_to_mel(0.0), _hz_to_mel(sr / 2.0), n_mels + 2))
    lo, ce, hi = edges[:-2, None], edges[1:-1, None], edges[2:, None]
    tri = np.maximum(0.0, np.minimum((freqs[None, :] - lo) / (ce - lo), (hi - freqs[None, :]) / (hi - ce)))
    return tri * (2.0 / (hi - lo))


def _get_tables(sr, n_fft, n_mels, device):
    key = (int(sr), n_fft, n_mels, str(device))
    t = _tables.get(key)
    if t is None:
        bins = 1 + n_fft // 2
        ang = 2.0 * np.pi * np.outer(np.arange(n_fft), np.arange(bins)) / n_fft
        basis = np.concatenate([np.cos(ang), -np.sin(ang)], 1).astype(np.float32)           # (n_fft, 2*bins)
        melw = mel_filters(sr, n_fft, n_mels).astype(np.float32)                            # (n_mels, bins)
        t = (torch.from_numpy(basis).to(device), torch.from_numpy(melw).to(device))
        _tables[key] = t
    return t


def log_melspectrogram(wave_data, sr, n_fft=2048, hop=512, n_mels=80, floor=1e-6):
    """np.log(np.maximum(1e-6, librosa.feature.melspectrogram(signal, n_mels=80, sr=sr).T)) -> device tensor (frames, n_mels)."""
    from . import nn
    dev = nn._device()
    y = torch.as_tensor(np.ascontiguousarray(wave_data, dtype=np.float32)).to(dev)
    n = y.numel()
    if n <= n_fft // 2:
        raise L.DepError(f'signal of {n} samples is shorter than the reflect padding ({n_fft // 2})')
    basis, melw = _get_tables(sr, n_fft, n_mels, dev)
    bins = 1 + n_fft // 2
    nfr = 1 + n // hop
    lib = L.load()
    frames = torch.empty(nfr, n_fft, dtype=torch.float32, device=dev)
    L.check(lib.dep_frame_window(y.data_ptr(), n, n_fft, hop, nfr, frames.data_ptr(), L.stream()), 'dep_frame_window')
    reim = torch.empty(nfr, 2 * bins, dtype=torch.float32, device=dev)
    L.gemm(0, 0, nfr, 2 * bins, n_fft, frames, n_fft, basis, 2 * bins, reim, 2 * bins)
    power = torch.empty(nfr, bins, dtype=torch.float32, device=dev)
    L.check(lib.dep_power_spectrum(reim.data_ptr(), nfr, bins, 2 * bins, power.data_ptr(), L.stream()), 'dep_power_spectrum')
    mel = torch.empty(nfr, n_mels, dtype=torch.float32, device=dev)
    L.gemm(0, 1, nfr, n_mels, bins, power, bins, melw, bins, mel, n_mels)
    L.check(lib.dep_log_floor(mel.data_ptr(), mel.data_ptr(), mel.numel(), floor, L.stream()), 'dep_log_floor')
    return mel


class NetVLAD:
    """loupe_keras.NetVLAD(feature_size, max_samples, cluster_size, output_dim): random-normal initial weights with the
    layer's standard deviations (1/sqrt(feature_size) for the cluster tensors, 1/sqrt(cluster_size) for the projection)."""

    def __init__(self, feature_size, max_samples, cluster_size, output_dim, seed=None, weights=None):
        from . import nn
        self.feature_size, self.max_samples, self.cluster_size, self.output_dim = feature_size, max_samples, cluster_size, output_dim
        dev = nn._device()
        if weights is None:
            g = nn.make_generator(seed)
            sf, sc = 1.0 / math.sqrt(feature_size), 1.0 / math.sqrt(cluster_size)
            weights = {'cluster_weights': torch.randn(feature_size, cluster_size, generator=g) * sf,
                       'cluster_biases': torch.randn(cluster_size, generator=g) * sf,
                       'cluster_weights2': torch.randn(feature_size, cluster_size, generator=g) * sf,
                       'hidden1_weights': torch.randn(cluster_size * feature_size, output_dim, generator=g) * sc}
        self.weights = {k: torch.as_tensor(np.asarray(v, dtype=np.float32) if not torch.is_tensor(v) else v).float().contiguous().to(dev)
                        for k, v in weights.items()}

    def __call__(self, x):
        W = self.weights
        F, K, D = self.feature_size, self.cluster_size, self.output_dim
        N = x.shape[0]
        dev = x.device
        lib = L.load()
        act = torch.empty(N, K, dtype=torch.float32, device=dev)
        L.gemm(0, 0, N, K, F, x, F, W['cluster_weights'], K, act, K, bias=W['cluster_biases'])
        L.check(lib.dep_row_softmax(act.data_ptr(), act.data_ptr(), N, K, L.stream()), 'dep_row_softmax')
        a_sum = torch.empty(K, dtype=torch.float32, device=dev)
        L.colsum(act, a_sum)
        vkf = torch.empty(K, F, dtype=torch.float32, device=dev)
        ws = L.gemm_ws(1, 0, K, F, N, dev)
        L.gemm(1, 0, K, F, N, act, K, x, F, vkf, F, ws=ws)                               # assignment^T x frames
        flat = torch.empty(1, F * K, dtype=torch.float32, device=dev)
        L.check(lib.dep_vlad_normalize(vkf.data_ptr(), a_sum.data_ptr(), W['cluster_weights2'].data_ptr(), flat.data_ptr(),
                                       F, K, L.stream()), 'dep_vlad_normalize')
        out = torch.empty(1, D, dtype=torch.float32, device=dev)
        L.gemm(0, 0, 1, D, F * K, flat, F * K, W['hidden1_weights'], D, out, D)
        return out


def wav2vlad(wave_data, sr, seed=None, weights=None):
    """Reference lines 57-72: (1, cluster_size * 16) NetVLAD feature of one response; numpy array like the reference's `r`."""
    melspec = log_melspectrogram(wave_data, sr, n_mels=80)
    layer = NetVLAD(feature_size=melspec.shape[1], max_samples=melspec.shape[0], cluster_size=cluster_size,
                    output_dim=cluster_size * 16, seed=seed, weights=weights)
    return layer(melspec).cpu().numpy()


def _read_wav(path):
    f = wave.open(path)
    sr, nframes = f.getframerate(), f.getnframes()
    data = np.frombuffer(f.readframes(nframes), dtype=np.short).astype(np.float64)
    f.close()
    return data, sr, nframes / sr


def extract_features(number, audio_features, targets, path):
    """Reference lines 74-114: the three responses (positive / neutral / negative) of volunteer `number` -> three wav2vlad
    features appended as one [3 x (1, 256)] entry, the SDS score appended to `targets`."""
    global max_len, min_len
    base = os.path.join(prefix, '{1}/{0}'.format(number, path))
    if not os.path.exists(os.path.join(base, 'positive_out.wav')):
        return
    waves = []
    for name in ('positive_out.wav', 'neutral_out.wav', 'negative_out.wav'):
        data, sr, length = _read_wav(os.path.join(base, name))
        max_len = max(max_len, length); min_len = min(min_len, length)
        if data.shape[0] < 1:
            data = np.array([1e-4] * sr * 5)
        waves.append((data, sr))
    with open(os.path.join(base, 'new_label.txt')) as fli:
        target = float(fli.readline())
    audio_features.append([wav2vlad(d, sr) for d, sr in waves])
    targets.append(target)


def main(root=None, n=114):
    """Reference lines 117-131: both corpus halves -> Features/AudioWhole/whole_{samples,labels}_reg_256.npz."""
    global prefix
    if root is not None:
        prefix = os.path.abspath(root)
    audio_features, audio_targets = [], []
    for part in ('Data', 'ValidationData'):
        for index in range(n):
            extract_features(index + 1, audio_features, audio_targets, part)
    print("Saving npz file locally...")
    os.makedirs(os.path.join(prefix, 'Features/AudioWhole'), exist_ok=True)
    np.savez(os.path.join(prefix, 'Features/AudioWhole/whole_samples_reg_%d.npz' % (cluster_size * 16)), audio_features)
    np.savez(os.path.join(prefix, 'Features/AudioWhole/whole_labels_reg_%d.npz' % (cluster_size * 16)), audio_targets)
    print(max_len, min_len)
    return audio_features, audio_targets


if __name__ == '__main__':
    main()
