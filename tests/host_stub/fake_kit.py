"""TEST INFRASTRUCTURE: the arithmetic of tests/host_stub/fake_lyra_hip_codec.cc as the component kit of
oracle/lyra_codec_model.py, so that the model and the C++ batch codec (built against the fake C ABI) can be compared on
the CPU: what is under test is the host logic around the components, not the components."""
import numpy as np

HOP = 320


def is_noise_hop(pcm):
    return bool(np.all(np.abs(pcm.astype(np.int64)) < 64))


class _State:
    def __init__(self):
        self.enc_hops = self.dec_hops = self.cng_hops = 0
        self.noise_calls = [0, 0]


class FakeKit:
    """One per stream and side pair (the model builds an encoder and a decoder per stream; each gets its own kit)."""

    def __init__(self):
        self.s = _State()
        kit = self

        class Resampler:
            def __init__(self, in_rate, out_rate):
                self.i, self.o = in_rate, out_rate

            def Resample(self, audio):
                audio = np.asarray(audio, np.int16)
                n_out = audio.size * self.o // self.i
                return audio[(np.arange(n_out, dtype=np.int64) * self.i) // self.o].copy()

        class Stream:
            def encode(self, audio):                     # -> "features" = the packet source
                return np.asarray(audio, np.int16).copy()

            def decode(self, feat):
                s = kit.s
                i = np.arange(HOP, dtype=np.int64)
                if feat.dtype == np.float32:             # ZeroFeatureEstimator: packet-loss concealment
                    out = -500 + i + 7 * s.dec_hops + int(feat[0])
                else:
                    out = feat.astype(np.int64)[i % feat.size] * 64 + i + 7 * s.dec_hops
                s.dec_hops += 1
                return out.astype(np.int16)              # (values stay inside int16 by construction)

        class NoiseEstimator:
            def __init__(self, side):
                self.side = side

            def ReceiveSamples(self, pcm):
                kit.s.noise_calls[self.side] += 1
                return is_noise_hop(np.asarray(pcm, np.int16)), None

            def noise_estimate(self):
                return None                              # the comfort-noise fake reads the call count itself

        class ComfortNoiseGenerator:
            def generate(self, _features):
                s = kit.s
                i = np.arange(HOP, dtype=np.int64)
                out = 2000 + (i & 31) + 3 * s.noise_calls[1] + 11 * s.cng_hops
                s.cng_hops += 1
                return out.astype(np.int16)

        self.Resampler, self._Stream, self._Noise, self._Cng = Resampler, Stream, NoiseEstimator, ComfortNoiseGenerator

    def Stream(self):
        return self._Stream()

    def NoiseEstimator(self, side, sample_rate_hz=16000):
        return self._Noise(side)

    def ComfortNoiseGenerator(self, seed):
        return self._Cng()

    # quantizer stand-ins: the "indices" are the packet itself
    def rvq_encode(self, feat, num_stages):
        nbytes = (num_stages * 4 + 7) // 8
        j = np.arange(nbytes, dtype=np.int64)
        pk = (feat.astype(np.int64)[(j * 13) % HOP] + 31 * j + self.s.enc_hops) & 255
        self.s.enc_hops += 1
        return pk.astype(np.uint8)

    def pack(self, idx, num_stages):
        return idx[None]

    def unpack(self, packets, num_stages):
        return packets

    def rvq_decode(self, idx):
        return idx
