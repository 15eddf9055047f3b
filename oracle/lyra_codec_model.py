"""Per-stream restatement of the reference's public codec classes on top of the oracle components
(TEST INFRASTRUCTURE ONLY): LyraEncoder::Encode (lyra/lyra_encoder.cc:113-156) and LyraDecoder
(lyra/lyra_decoder.cc:172-373) -- resampling, DTX, packet FIFO, packet-loss concealment, comfort noise, cross-fades,
noise-estimator updates -- statement for statement.  tests/ compare the batched C++ twins
(lyra_amd/host/lyra_batch_codec.cc, device-backed) with this model stream by stream.
"""
import math

import numpy as np

from . import lyra_oracle as lo

HOP = 320
CONCEAL = 1280      # GetConcealmentDurationSamples (lyra_decoder.cc:41-51)
FADE = 640          # GetFadeDurationSamples (lyra_decoder.cc:53-62)
TO_CNG, FROM_CNG = 1, -1


class OracleKit:
    """The components the two classes below are assembled from: the oracle's (default), or stand-ins with the same
    interface (tests/host_stub/fake_kit.py, for the CPU test of the C++ twins' host logic)."""

    def __init__(self, oracle):
        self.o = oracle
        self.Resampler = lo.Resampler

    def Stream(self):
        return lo.Stream(self.o)

    def NoiseEstimator(self, side, sample_rate_hz=16000):
        return lo.NoiseEstimator(self.o, sample_rate_hz=sample_rate_hz)

    def ComfortNoiseGenerator(self, seed):
        return lo.ComfortNoiseGenerator(self.o, seed=seed)

    def rvq_encode(self, feat, num_stages):
        return self.o.rvq_encode(feat, num_stages)

    def pack(self, idx, num_stages):
        return self.o.pack(idx, num_stages)

    def unpack(self, packets, num_stages):
        return self.o.unpack(packets, num_stages)

    def rvq_decode(self, idx):
        return self.o.rvq_decode(idx)


class RefLyraEncoder:
    def __init__(self, oracle, sample_rate_hz, num_bits, enable_dtx, kit=None):
        self.o = kit if kit is not None else OracleKit(oracle)
        self.rate, self.bits, self.dtx = sample_rate_hz, num_bits, enable_dtx
        self.resampler = self.o.Resampler(sample_rate_hz, 16000) if sample_rate_hz != 16000 else None
        self.stream = self.o.Stream()
        # NoiseEstimator::Create(sample_rate_hz, ...): the encoder hands over its EXTERNAL rate (lyra_encoder.cc:82-85)
        self.noise = self.o.NoiseEstimator(0, sample_rate_hz) if enable_dtx else None

    def Encode(self, audio):
        audio = np.asarray(audio, np.int16)
        if self.resampler is not None:
            audio = self.resampler.Resample(audio)
        assert audio.size == HOP
        if self.dtx:
            is_noise, _ = self.noise.ReceiveSamples(audio)
            if is_noise:
                return np.zeros(0, np.uint8)          # Packet<0>: the empty packet
        feat = self.stream.encode(audio)
        idx = self.o.rvq_encode(feat, self.bits // 4)
        return self.o.pack(idx, self.bits // 4)[0]


class _Fifo:
    """GenerativeModel base class (generative_model_interface.h:45-134)."""

    def __init__(self, conditioning):
        self.q, self.next, self.hop, self.conditioning = [], 0, None, conditioning

    def available(self):
        return len(self.q) * HOP - self.next

    def add(self, features):
        self.q.append(features)

    def generate(self, n):
        if n == 0:
            return np.zeros(0, np.int16)
        assert self.available() > 0
        if self.next == 0:
            self.hop = self.conditioning(self.q[0])
        assert n <= HOP - self.next
        out = self.hop[self.next:self.next + n].copy()
        self.next += n
        if self.next == HOP:
            self.next = 0
            self.q.pop(0)
        return out


class RefLyraDecoder:
    def __init__(self, oracle, sample_rate_hz, cng_seed, kit=None):
        self.o = kit if kit is not None else OracleKit(oracle)
        self.rate = sample_rate_hz
        self.stream = self.o.Stream()
        self.noise = self.o.NoiseEstimator(1)
        self.cng_gen = self.o.ComfortNoiseGenerator(cng_seed)
        self.model = _Fifo(lambda f: self.stream.decode(f))
        self.cng = _Fifo(lambda f: self.cng_gen.generate(f))
        self.resampler = self.o.Resampler(16000, sample_rate_hz) if sample_rate_hz != 16000 else None
        self.leftover = np.zeros(0, np.int16)
        self.concealment, self.fade, self.fade_dir = 0, 0, FROM_CNG
        self.noise_buf = np.zeros(0, np.int16)

    def SetEncodedPacket(self, packet):
        packet = np.asarray(packet, np.uint8)
        bits = {8: 64, 15: 120, 23: 184}[packet.size]
        if self.concealment == CONCEAL:
            self.concealment = -self.cng.available()
        elif self.concealment > 0:
            self.concealment = -self.model.available()
        idx = self.o.unpack(packet[None], bits // 4)
        self.model.add(self.o.rvq_decode(idx)[0])

    def is_comfort_noise(self):
        return self.fade == FADE

    def DecodeSamples(self, n):
        if self.resampler is None:
            return self._internal(n)
        used = min(self.leftover.size, n)
        internal = 0
        if n > self.leftover.size:
            ratio = np.float32(self.rate) / np.float32(16000)
            internal = int(math.ceil(np.float32(n - self.leftover.size) / ratio))
        out = np.empty(n, np.int16)
        out[:used] = self.leftover[:used]
        self.leftover = self.leftover[used:]
        samples = self._internal(internal)
        if internal == 0:
            return out
        ext = self.resampler.Resample(samples)
        out[used:] = ext[:n - used]
        self.leftover = np.concatenate([self.leftover, ext[n - used:]])
        return out

    def _n_to_generate(self, requested, done):
        if self.concealment < 0:
            rem = abs(self.concealment)
        elif self.concealment < CONCEAL:
            rem = self.model.available() % HOP
        else:
            rem = self.cng.available()
        if rem == 0:
            rem = HOP
        return min(requested - done, rem)

    def _internal(self, total):
        result = []
        done = 0
        while done < total:
            n = self._n_to_generate(total, done)
            received = self.model.available() > 0 and self.concealment == 0
            if received:
                self.fade_dir = FROM_CNG
            elif self.concealment == CONCEAL:
                self.fade_dir = TO_CNG
            else:
                self.concealment += n
            cng_n = gen_n = n
            next_fade = self.fade + self.fade_dir * n
            if self.fade_dir == TO_CNG and self.fade == FADE:
                next_fade, gen_n = FADE, 0
            elif self.fade_dir == FROM_CNG and self.fade == 0:
                next_fade, cng_n = 0, 0
            if gen_n > 0 and self.model.available() == 0:
                self.model.add(np.zeros(64, np.float32))          # ZeroFeatureEstimator::Estimate
            audio = self.model.generate(gen_n)
            if cng_n > 0 and self.cng.available() == 0:
                self.cng.add(self.noise.noise_estimate())
            noise = self.cng.generate(cng_n)
            if noise.size == 0:
                result.append(audio)
            elif audio.size == 0:
                result.append(noise)
            else:
                fade = self.fade
                mixed = np.empty(n, np.int16)
                for i in range(n):
                    w = np.float32((np.float32(1.0) + math.cos(fade * math.pi / FADE)) / np.float32(2.0))
                    v = np.float32(np.float32(audio[i]) * w) + np.float32(np.float32(noise[i]) * np.float32(np.float32(1.0) - w))
                    mixed[i] = np.int16(np.trunc(np.float32(v)))
                    fade += self.fade_dir
                result.append(mixed)
            self.fade = next_fade
            if received:
                self.noise_buf = np.concatenate([self.noise_buf, audio])
                if self.noise_buf.size == HOP:
                    self.noise.ReceiveSamples(self.noise_buf)
                    self.noise_buf = np.zeros(0, np.int16)
            done += n
        return np.concatenate(result) if result else np.zeros(0, np.int16)
