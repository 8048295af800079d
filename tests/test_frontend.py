"""Audio front-end restatement (transkun_amd/frontend.py) against the reference's own makeFrame / GaussianWindows / Spectrum
(tests/golden/frontend.npz, tools/make_golden.py:case_frontend).  The mel filterbank is parity-unpinned (torchaudio is not
installed in the build container): structural checks only.  CPU."""
import numpy as np
import torch

from conftest import load_golden, rel_err


def test_framing_windows_and_spectrum_match_reference():
    from transkun_amd import synth
    from transkun_amd.frontend import GaussianWindows, Spectrum, makeFrame
    g = load_golden("frontend")
    x = synth.hash_normal(2 * 20000, 950, "cpu").view(2, 20000)
    for hop, win in ((1024, 4096), (160, 400)):
        fr = makeFrame(x, hop, win)
        assert list(fr.shape) == list(g[f"frames_{hop}_shape"])
        assert np.array_equal(fr.double().sum(-1).numpy(), g[f"frames_{hop}_rowsum"])          # framing is a copy: exact
        assert np.array_equal(fr[0, :2, :8].numpy(), g[f"frames_{hop}_first"]) and np.array_equal(fr[1, -2:, -8:].numpy(), g[f"frames_{hop}_last"])
    gw = GaussianWindows(5, 4096)
    with torch.no_grad():
        gw.sigma.copy_(torch.from_numpy(g["gw_sigma"])); gw.center.copy_(torch.from_numpy(g["gw_center"]))
    Y = gw.get().detach()
    assert rel_err(Y.double().sum(0).numpy(), g["gw_colsum"]) < 1e-6
    assert np.allclose(Y[[0, 1000, 2048, 4095]].numpy(), g["gw_rows"], rtol=1e-5, atol=1e-7)
    sp = Spectrum(4096, nExtraWins=5)
    with torch.no_grad():
        sp.winGen.sigma.copy_(gw.sigma); sp.winGen.center.copy_(gw.center)
        S = sp(makeFrame(x, 1024, 4096))
    assert list(S.shape) == list(g["spec_shape"])
    P = S.abs().pow(2)
    assert rel_err(P.double().sum(dim=(0, 1, 2)).numpy(), g["spec_power_sum"]) < 1e-5
    assert np.allclose(P[0, 3, [0, 1, 17, 500, 2048], :].numpy(), g["spec_power_bins"], rtol=1e-4, atol=1e-7)
    assert np.allclose(torch.view_as_real(S[1, 5, [2, 300], :]).numpy(), g["spec_re_im"], rtol=1e-4, atol=1e-5)


def test_mel_filterbank_structure_and_log_mel_range():
    """melscale_fbanks (parity unpinned): non-negative triangles with peak <= 1, one peak per band at increasing
    frequencies inside [f_min, f_max], zero outside; MelSpectrum(log=True) maps into [0, ~1]."""
    from transkun_amd import synth
    from transkun_amd.frontend import MelSpectrum, makeFrame, melscale_fbanks, normalize_gain
    fb = melscale_fbanks(2049, 30, 8000, 229, 44100)
    assert fb.shape == (2049, 229) and float(fb.min()) >= 0.0 and float(fb.max()) <= 1.0 + 1e-6
    freqs = torch.linspace(0, 22050, 2049)
    assert float(fb[freqs < 30].abs().max()) == 0.0 and float(fb[freqs > 8000].abs().max()) == 0.0
    peaks = fb.argmax(0)
    assert bool((peaks[1:] >= peaks[:-1]).all()) and int(peaks[0]) >= 2 and int(peaks[-1]) <= int(8000 / 22050 * 2048) + 1
    m = MelSpectrum(4096, 30, 8000, 229, 44100, nExtraWins=5, log=True, toMono=True)
    x = synth.hash_normal(2 * 30000, 960, "cpu").view(1, 2, 30000)                 # [nBatch, nAudioChannel, nSample]
    frames = normalize_gain(makeFrame(x, 1024, 4096))
    feat = m(frames)
    assert feat.shape == (1, 1, frames.shape[-2], 229, 6)
    assert float(feat.min()) >= 0.0 and torch.isfinite(feat).all()
