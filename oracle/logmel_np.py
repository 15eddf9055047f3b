"""numpy restatement of LogMelSpectrogramExtractorImpl (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/lyra/log_mel_spectrogram_extractor_impl.cc:53-126 and
the audio_dsp Spectrogram / MelFilterbank semantics it calls (third-party,
un-vendored: com_google_audio_dsp @14a45c5, WORKSPACE:68-78), restated from
SURVEY.md A.4.  Pinned by the reference's own golden vectors
(log_mel_spectrogram_extractor_impl_test.cc:37-59) in tests/test_oracle_logmel.py.
"""
import numpy as np


def next_pow2(n):
    p = 1
    while p < n:
        p *= 2
    return p


class MelFilterbank:
    def __init__(self, num_fft_bins, sample_rate, num_mel, lower_hz, upper_hz):
        mel = lambda f: 1127.0 * np.log1p(f / 700.0)
        mel_low = mel(lower_hz)
        mel_hi = mel(upper_hz)
        spacing = (mel_hi - mel_low) / (num_mel + 1)
        self.center = np.array([mel_low + spacing * (i + 1) for i in range(num_mel + 1)])
        hz_per_bin = 0.5 * sample_rate / (num_fft_bins - 1)
        self.start = int(1.5 + lower_hz / hz_per_bin)
        self.end = int(upper_hz / hz_per_bin)
        self.num_mel = num_mel
        self.band = np.full(num_fft_bins, -2, np.int64)
        self.weight = np.zeros(num_fft_bins, np.float64)
        channel = 0
        for i in range(num_fft_bins):
            melf = mel(i * hz_per_bin)
            if i < self.start or i > self.end:
                continue
            while channel < num_mel and self.center[channel] < melf:
                channel += 1
            self.band[i] = channel - 1
        for i in range(num_fft_bins):
            ch = self.band[i]
            if i < self.start or i > self.end:
                continue
            melf = mel(i * hz_per_bin)
            if ch >= 0:
                self.weight[i] = (self.center[ch + 1] - melf) / (self.center[ch + 1] - self.center[ch])
            else:
                self.weight[i] = (self.center[0] - melf) / (self.center[0] - mel_low)

    def compute(self, sq):
        out = np.zeros(self.num_mel, np.float64)
        for i in range(self.start, self.end + 1):
            v = np.sqrt(sq[i])
            w = v * self.weight[i]
            ch = self.band[i]
            if ch >= 0:
                out[ch] += w
            ch += 1
            if ch < self.num_mel:
                out[ch] += v - w
        return out


class LogMelExtractor:
    """One stream.  extract(hop int16[hop]) -> float32[num_mel]."""

    def __init__(self, sample_rate=16000, hop=320, window=640, num_mel=160):
        self.hop, self.window = hop, window
        self.fft = next_pow2(window)
        self.bins = self.fft // 2 + 1
        self.hann = 0.5 - 0.5 * np.cos(2.0 * np.pi * np.arange(window) / window)
        self.mel = MelFilterbank(self.bins, sample_rate, num_mel, 0.0, 0.495 * sample_rate)
        # primed with one all-zero window (impl.cc:69-77): queue keeps window-hop zeros
        self.queue = np.zeros(window - hop, np.float64)

    def extract(self, audio):
        audio = np.asarray(audio)
        assert audio.size == self.hop
        frame = np.concatenate([self.queue, audio.astype(np.float64)])
        self.queue = frame[self.hop:]
        buf = np.zeros(self.fft)
        buf[:self.window] = frame * self.hann
        spec = np.fft.rfft(buf)
        sq = spec.real ** 2 + spec.imag ** 2
        m = self.mel.compute(sq).astype(np.float32)
        return (np.log(np.maximum(m, np.float32(500.0))) / np.float32(10.0)).astype(np.float32)
