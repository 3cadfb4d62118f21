"""Weights-directory naming contract + INIT_LEN helper of the reference
(code/utilities/utilities.py:872-919; the directory name is written by code/train.py:103)."""
import re


def parse_hidden_size(model_name):
    """'...-HS[64]-...' -> 64   (code/utilities/utilities.py:872-884)."""
    return int(re.search(r'-HS\[(.\d*)\]-', model_name).group(1))


def parse_model(model_name):
    """Text before the first '-'   (code/utilities/utilities.py:887-899)."""
    return model_name[:model_name.find('-')]


def parse_loss(model_name):
    """'...]-L[DCPreESR]-DS[...' -> 'DCPreESR'   (code/utilities/utilities.py:902-914)."""
    return re.search(r'\]-L\[(.*)\]-DS\[', model_name).group(1)


def nextpow2(number):
    """Next power of two >= number   (code/utilities/utilities.py:917-919)."""
    return 2**(number - 1).bit_length()


def mel_filterbank_sparse(sr=44100, n_fft=2048, n_mels=160, fmin=0.0, fmax=None):
    """The `mel_basis` of the reference's TimeFreqConverter (code/utilities/utilities.py:639-646:
    `librosa.filters.mel(sr, n_fft, n_mels, fmin, fmax)`, Slaney mel scale, unit-area triangles) in the row-compressed
    form the device kernel walks: -> (first_bin int32 [n_mels], row_start int32 [n_mels + 1], weights float32 [nnz]);
    filter m = weights[row_start[m] : row_start[m + 1]] applied to bins first_bin[m] ....  librosa itself is neither a
    dependency of this engine nor available to pin against: the published algorithm, parity unpinned."""
    import numpy as np
    fmax = sr / 2.0 if fmax is None else fmax
    to_mel = lambda f: 15.0 + np.log(f / 1000.0) / (np.log(6.4) / 27.0) if f >= 1000.0 else 3.0 * f / 200.0      # noqa: E731
    to_hz = lambda m: 1000.0 * np.exp((np.log(6.4) / 27.0) * (m - 15.0)) if m >= 15.0 else 200.0 * m / 3.0       # noqa: E731
    edges = np.array([to_hz(m) for m in np.linspace(to_mel(fmin), to_mel(fmax), n_mels + 2)])
    freqs = np.arange(1 + n_fft // 2) * (sr / float(n_fft))
    first, start, weights = [], [0], []
    for m in range(n_mels):
        lo, mid, hi = edges[m], edges[m + 1], edges[m + 2]
        w = np.maximum(0.0, np.minimum((freqs - lo) / (mid - lo), (hi - freqs) / (hi - mid))) * (2.0 / (hi - lo))
        nz = np.nonzero(w)[0]
        b0, b1 = (int(nz[0]), int(nz[-1]) + 1) if len(nz) else (0, 0)
        first.append(b0)
        weights.append(w[b0:b1])
        start.append(start[-1] + (b1 - b0))
    return (np.asarray(first, np.int32), np.asarray(start, np.int32),
            np.concatenate(weights).astype(np.float32) if start[-1] else np.zeros(0, np.float32))
