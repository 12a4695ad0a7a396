"""Stand-ins for the reference's neural modules, for measurements and integration tests on the GPU box where the
reference checkout (and its third-party dependencies) cannot travel.  NOT part of the product and not a re-implementation
of the reference's networks: random-weight torch modules with the same INTERFACE and roughly the same size, so that
(a) ``bench.py`` can time "module mode" (controls produced by a network on the GPU, SURVEY.md 8-d control mode (i)) and
(b) the cfg-5 seam (DDSP synthesiser -> log-mel -> denoiser -> NSF source, main_diff.py:356-359,378) can run end to end
around the HIP kernels.  The drop-in modules of ``ddsp_svc_amd.vocoder`` take them through ``unit2ctrl_factory``.
"""
import math

import torch
import torch.nn as nn


class StandInUnit2Control(nn.Module):
    """Interface of ddsp/unit2control.py:26-109: ``forward(units, f0, phase, volume, spk_id, spk_mix_dict, aug_shift)
    -> (dict of [B,F,n_i] views of one [B,F,sum n_i] tensor, hidden [B,F,256])``.  Shape: the reference's conv stack
    (Conv1d k3 - GroupNorm - LeakyReLU - Conv1d k3), the three scalar embeddings, a 3-layer / 8-head / 256-wide encoder in
    place of PCmer, LayerNorm, dense output -- 3.4 M parameters for n_unit = 768 and 768 output channels (the reference's
    module: 3.8 M)."""

    def __init__(self, input_channel, n_spk, output_splits, **kwargs):
        super().__init__()
        self.output_splits = dict(output_splits)
        self.kwargs = kwargs
        self.f0_embed = nn.Linear(1, 256)
        self.phase_embed = nn.Linear(1, 256)
        self.volume_embed = nn.Linear(1, 256)
        self.n_spk = n_spk
        if n_spk is not None and n_spk > 1:
            self.spk_embed = nn.Embedding(n_spk, 256)
        self.stack = nn.Sequential(nn.Conv1d(input_channel, 256, 3, 1, 1), nn.GroupNorm(4, 256), nn.LeakyReLU(),
                                   nn.Conv1d(256, 256, 3, 1, 1))
        layer = nn.TransformerEncoderLayer(256, 8, dim_feedforward=1024, dropout=0.0, batch_first=True, norm_first=True)
        self.decoder = nn.TransformerEncoder(layer, 3, enable_nested_tensor=False)
        self.norm = nn.LayerNorm(256)
        self.dense_out = nn.Linear(256, sum(self.output_splits.values()))

    def forward(self, units, f0, phase, volume, spk_id=None, spk_mix_dict=None, aug_shift=None):
        x = self.stack(units.transpose(1, 2)).transpose(1, 2)
        x = x + self.f0_embed((1 + f0 / 700).log()) + self.phase_embed(phase / math.pi) + self.volume_embed(volume)
        if self.n_spk is not None and self.n_spk > 1 and spk_id is not None:
            x = x + self.spk_embed(spk_id - 1)
        x = self.norm(self.decoder(x))
        e = self.dense_out(x)
        return dict(zip(self.output_splits, torch.split(e, list(self.output_splits.values()), dim=-1))), x


class StandInDenoiser(nn.Module):
    """Takes the place of the diffusion / reflow sampler between the DDSP mel and the vocoder in the cascade
    (diffusion/vocoder.py:248-262): log-mel [B,F,128] + hidden [B,F,256] in, refined log-mel [B,F,128] out.  A few
    residual conv blocks; the sampler's cost is not what the seam test measures."""

    def __init__(self, n_mels=128, n_hidden=256, width=256, layers=4):
        super().__init__()
        self.inp = nn.Conv1d(n_mels + n_hidden, width, 1)
        self.blocks = nn.ModuleList(nn.Sequential(nn.Conv1d(width, width, 3, 1, 2 ** (i % 3), dilation=2 ** (i % 3)), nn.GELU(),
                                                  nn.Conv1d(width, width, 1)) for i in range(layers))
        self.out = nn.Conv1d(width, n_mels, 1)

    def forward(self, mel, hidden):
        x = self.inp(torch.cat([mel, hidden], -1).transpose(1, 2))
        for b in self.blocks:
            x = x + b(x)
        return mel + self.out(x).transpose(1, 2)


class StandInGeneratorBody(nn.Module):
    """The part of NSF-HiFiGAN's ``Generator`` (nsf_hifigan/models.py:253-330) AROUND its harmonic source: mel [B,128,F]
    is upsampled by (8, 8, 2, 2, 2) = 512 with transposed convolutions and the source excitation [B,T] is injected at
    every scale through strided convolutions, as the reference does -- with a fraction of its channels."""

    RATES = (8, 8, 2, 2, 2)

    def __init__(self, n_mels=128, channels=64):
        super().__init__()
        self.pre = nn.Conv1d(n_mels, channels, 7, 1, 3)
        self.ups, self.src = nn.ModuleList(), nn.ModuleList()
        c, remaining = channels, 512
        for r in self.RATES:
            remaining //= r
            self.ups.append(nn.ConvTranspose1d(c, c // 2, 2 * r, r, r // 2))
            self.src.append(nn.Conv1d(1, c // 2, 2 * remaining, remaining, remaining // 2) if remaining > 1
                            else nn.Conv1d(1, c // 2, 1))
            c //= 2
        self.post = nn.Conv1d(c, 1, 7, 1, 3)

    def forward(self, mel, excitation):
        x = self.pre(mel)
        e = excitation.unsqueeze(1)
        for up, s in zip(self.ups, self.src):
            x = up(torch.nn.functional.leaky_relu(x, 0.1))
            x = x + s(e)[..., :x.shape[-1]]
        return torch.tanh(self.post(torch.nn.functional.leaky_relu(x))).squeeze(1)


class CascadeSeam(nn.Module):
    """BASELINE cfg 5's seam (main_diff.py:356-359,378; diffusion/vocoder.py:234-262; nsf_hifigan/models.py:287) built
    from the drop-in pieces around stand-in networks:

        drop-in CombSubSuperFast (stand-in Unit2Control)  ->  STFT.get_mel (k_mel)  ->  stand-in denoiser
            ->  SourceModuleHnNSF (k_sinegen) + stand-in generator body  ->  waveform [B, T]

    ``forward`` runs the modules; ``op_by_op`` runs the same chain through the functional entry points one operation at
    a time with the same random draws (same torch seed, same draw order), for the equality check of the seam test."""

    def __init__(self, sr=44100, hop=512, n_unit=768, win=2048):
        super().__init__()
        from ddsp_svc_amd import mel, nsf_source, vocoder
        self.sr, self.hop = sr, hop
        self.ddsp = vocoder.CombSubSuperFast(sr, hop, win, n_unit=n_unit, n_spk=1, unit2ctrl_factory=StandInUnit2Control)
        self.stft = mel.STFT(sr, 128, 2048, 2048, hop, 40, 16000)
        self.denoiser = StandInDenoiser()
        self.source = nsf_source.SourceModuleHnNSF(sr, harmonic_num=8)
        self.body = StandInGeneratorBody()

    @torch.no_grad()
    def forward(self, units, f0, volume):
        ddsp_wav, hidden, _ = self.ddsp(units, f0, volume, infer=True)              # main_diff.py:356-358
        ddsp_mel = self.stft.get_mel(ddsp_wav).transpose(1, 2)                      # diffusion/vocoder.py:248 -> [B,F,128]
        mel = self.denoiser(ddsp_mel, hidden)
        exc = self.source(f0[..., 0], self.hop)[..., 0]                             # models.py:287
        return self.body(mel.transpose(1, 2), exc), ddsp_wav, ddsp_mel

    @torch.no_grad()
    def op_by_op(self, units, f0, volume):
        from ddsp_svc_amd import mel as M, nsf_source, synth
        st = synth.fast_source(f0, self.sr, self.hop)
        ctrls, hidden = self.ddsp.unit2ctrl(units, f0, st.phase_frames, volume, spk_id=None, spk_mix_dict=None, aug_shift=None)
        B, F = f0.shape[0], f0.shape[1]
        gauss = torch.randn(B, F * self.hop, dtype=torch.float32, device=f0.device)
        wav = synth.combsubsuperfast_synth(f0, st, ctrls["harmonic_magnitude"], ctrls["harmonic_phase"],
                                           ctrls["noise_magnitude"], ctrls["noise_phase"], gauss, self.ddsp.window,
                                           self.sr, self.hop)
        basis, band, window = self.stft._tables(wav.device)
        ddsp_mel = M.mel_spectrogram(wav, window, basis, band, self.hop, self.stft.clip_val).transpose(1, 2)
        mel = self.denoiser(ddsp_mel, hidden)
        ri = torch.rand(1, 1, 9, device=f0.device)
        ri[..., 0] = 0
        nz = torch.randn(B, F * self.hop, 9, dtype=torch.float32, device=f0.device)
        exc = nsf_source.sine_source(f0[..., 0], self.hop, self.sr, self.source.l_linear.weight, self.source.l_linear.bias,
                                     ri, nz)
        return self.body(mel.transpose(1, 2), exc), wav, ddsp_mel
