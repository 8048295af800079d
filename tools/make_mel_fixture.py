"""tests/golden/mel_fbank.npz: the model's mel filterbank (n_freqs=2049, f_min=30, f_max=8000, n_mels=229, fs=44100;
pretrained/2.0.conf:6-34) derived INDEPENDENTLY of transkun_amd/frontend.py, in float64 and band by band, from the definition
torchaudio documents for torchaudio.functional.melscale_fbanks with its defaults (mel_scale="htk", norm=None):

    mel(f) = 2595 log10(1 + f / 700);  n_mels + 2 points equally spaced in mel between f_min and f_max, mapped back to Hz;
    band m is the triangle rising from point m to its peak 1 at point m+1 and falling to 0 at point m+2, sampled at the
    n_freqs bin frequencies linspace(0, fs // 2, n_freqs).

torchaudio itself is not installed in the build container, so this pins the product's filterbank to the published formula,
not to torchaudio's binary output.  Stored sparsely (rows, cols, float64 values) plus every band's three corner frequencies.
Runs anywhere (numpy only):  python tools/make_mel_fixture.py"""
import os

import numpy as np

n_freqs, f_min, f_max, n_mels, fs = 2049, 30.0, 8000.0, 229, 44100
bins = np.arange(n_freqs, dtype=np.float64) * ((fs // 2) / (n_freqs - 1))
mel_lo, mel_hi = 2595.0 * np.log10(1.0 + f_min / 700.0), 2595.0 * np.log10(1.0 + f_max / 700.0)
corners = np.empty((n_mels, 3))
rows, cols, vals = [], [], []
for m in range(n_mels):
    pts_mel = [mel_lo + (mel_hi - mel_lo) * (m + i) / (n_mels + 1) for i in range(3)]
    lo, mid, hi = (700.0 * (10.0 ** (x / 2595.0) - 1.0) for x in pts_mel)
    corners[m] = (lo, mid, hi)
    for r, f in enumerate(bins):
        if lo < f < hi:
            v = (f - lo) / (mid - lo) if f <= mid else (hi - f) / (hi - mid)
            if v > 0.0:
                rows.append(r); cols.append(m); vals.append(v)
out = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "mel_fbank.npz")
np.savez_compressed(out, rows=np.asarray(rows, np.int32), cols=np.asarray(cols, np.int32), vals=np.asarray(vals, np.float64),
                    corners=corners, meta=np.asarray([n_freqs, f_min, f_max, n_mels, fs], np.float64))
print(out, len(vals), "non-zero entries")
