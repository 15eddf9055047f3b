"""ctypes binding of oracle/_ref/liblyra_ref.so (TEST INFRASTRUCTURE ONLY): the REFERENCE'S OWN LyraEncoder, LyraDecoder,
NoiseEstimator, BufferedResampler, Packet<> and GenerativeModel FIFO, compiled from /root/reference where it lies with
oracle-backed network / DSP components injected (oracle/ref_glue.cc, oracle/ref_shims/README.md).

The classes below have the interface of oracle/lyra_codec_model.py so the same scripted sessions can be checked against
either; where this library is present it is the checker of record for rows f1 / f2 / f3 and lyra_codec_model.py is a
convenience (it is what still runs where the library has not been built).
"""
import ctypes as C
import os

import numpy as np

from . import lyra_oracle as lo

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "liblyra_ref.so")
_lib = None


def available():
    return os.path.exists(LIB)


def load(oracle):
    """-> the library, computing with `oracle` (an oracle.lyra_oracle.Oracle).  One oracle model per process."""
    global _lib
    if _lib is None:
        C.CDLL(os.path.join(HERE, "liblyra_oracle.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(LIB)
        vp, i32p = C.c_void_p, C.POINTER(C.c_int32)
        L.ref_set_model.argtypes = [vp]
        L.ref_set_cng_seed.argtypes = [C.c_uint64]
        L.ref_decoder_new.restype = vp
        L.ref_decoder_new.argtypes = [C.c_int]
        L.ref_decoder_free.argtypes = [vp]
        L.ref_decoder_set_packet.argtypes = [vp, vp, C.c_int]
        L.ref_decoder_decode.argtypes = [vp, C.c_int, vp]
        L.ref_decoder_is_comfort_noise.argtypes = [vp]
        L.ref_encoder_new.restype = vp
        L.ref_encoder_new.argtypes = [C.c_int, C.c_int, C.c_int]
        L.ref_encoder_free.argtypes = [vp]
        L.ref_encoder_encode.argtypes = [vp, vp, C.c_int, vp, C.c_int]
        L.ref_encoder_set_bitrate.argtypes = [vp, C.c_int]
        L.ref_noise_new.restype = vp
        L.ref_noise_free.argtypes = [vp]
        L.ref_noise_receive.argtypes = [vp, vp, C.c_int]
        L.ref_noise_get.argtypes = [vp, vp, vp]
        L.ref_packet_pack.argtypes = [C.c_char_p, vp, C.c_int]
        L.ref_packet_unpack.argtypes = [vp, C.c_int, C.c_char_p, C.c_int]
        L.ref_version.restype = C.c_char_p
        L.ref_encode_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_char_p]
        L.ref_decode_file.argtypes = [C.c_char_p, C.c_char_p, C.c_int, C.c_int, C.c_int, C.c_float, C.c_float, vp, vp, C.c_int,
                                      C.c_char_p]
        L.ref_unit_to_int16.argtypes = [vp, C.c_long, vp]
        L.ref_int16_to_unit.argtypes = [vp, C.c_long, vp]
        L.ref_log_spectral_distance.restype = C.c_float
        L.ref_log_spectral_distance.argtypes = [vp, vp, C.c_int]
        _lib = L
    _lib.ref_set_model(oracle.h)
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


BITRATE = {64: 3200, 120: 6000, 184: 9200}


class LyraEncoder:
    """chromemedia::codec::LyraEncoder (lyra/lyra_encoder.cc), private constructor reached through LyraEncoderPeer."""

    def __init__(self, oracle, sample_rate_hz, num_bits, enable_dtx):
        self.L = load(oracle)
        self.h = self.L.ref_encoder_new(sample_rate_hz, BITRATE[num_bits], int(enable_dtx))
        assert self.h, "LyraEncoder could not be created"

    def __del__(self):
        try:
            self.L.ref_encoder_free(self.h)
        except Exception:
            pass

    def Encode(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        out = np.empty(32, np.uint8)
        n = self.L.ref_encoder_encode(self.h, _p(audio), audio.size, _p(out), out.size)
        return None if n < 0 else out[:n].copy()


class LyraDecoder:
    """chromemedia::codec::LyraDecoder (lyra/lyra_decoder.cc) through LyraDecoderPeer; its comfort noise is the oracle's
    generator with `cng_seed`."""

    def __init__(self, oracle, sample_rate_hz, cng_seed):
        self.L = load(oracle)
        self.L.ref_set_cng_seed(cng_seed)
        self.h = self.L.ref_decoder_new(sample_rate_hz)
        assert self.h, "LyraDecoder could not be created"

    def __del__(self):
        try:
            self.L.ref_decoder_free(self.h)
        except Exception:
            pass

    def SetEncodedPacket(self, packet):
        packet = np.ascontiguousarray(packet, np.uint8)
        return bool(self.L.ref_decoder_set_packet(self.h, _p(packet), packet.size))

    def DecodeSamples(self, n):
        out = np.empty(max(n, 1), np.int16)
        got = self.L.ref_decoder_decode(self.h, n, _p(out))
        return None if got < 0 else out[:got].copy()

    def is_comfort_noise(self):
        return bool(self.L.ref_decoder_is_comfort_noise(self.h))


class NoiseEstimator:
    """chromemedia::codec::NoiseEstimator (lyra/noise_estimator.cc:96-245) as NoiseEstimator::Create builds it for the
    codec; its log-mel front end is the oracle's."""

    def __init__(self, oracle, sample_rate_hz=16000):
        self.L = load(oracle)
        self.L.ref_noise_new_rate.restype = C.c_void_p
        self.L.ref_noise_new_rate.argtypes = [C.c_int]
        self.h = self.L.ref_noise_new_rate(int(sample_rate_hz))
        assert self.h

    def __del__(self):
        try:
            self.L.ref_noise_free(self.h)
        except Exception:
            pass

    def ReceiveSamples(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        r = self.L.ref_noise_receive(self.h, _p(pcm), pcm.size)
        assert r >= 0
        return bool(r)

    def noise_estimate(self):
        e = np.empty(160, np.float32)
        self.L.ref_noise_get(self.h, _p(e), None)
        return e

    def noise_bound(self):
        b = np.empty(160, np.float32)
        self.L.ref_noise_get(self.h, None, _p(b))
        return b


def bits_string(indices, num_stages):
    """stage indices -> the '0'/'1' string ResidualVectorQuantizer::Quantize returns (first stage leading)."""
    return "".join(format(int(i), "04b") for i in indices[:num_stages])


def pack(oracle, bits):
    """Packet<>::PackQuantized (packet.h:91-122) -> uint8 array."""
    L = load(oracle)
    out = np.empty(32, np.uint8)
    n = L.ref_packet_pack(bits.encode(), _p(out), out.size)
    assert n >= 0
    return out[:n].copy()


def unpack(oracle, packet):
    """Packet<>::UnpackPacket (packet.h:124-146) -> '0'/'1' string, or None."""
    L = load(oracle)
    packet = np.ascontiguousarray(packet, np.uint8)
    buf = C.create_string_buffer(256)
    n = L.ref_packet_unpack(_p(packet), packet.size, buf, 256)
    return None if n < 0 else buf.value.decode()


def unit_to_int16(oracle, x):
    """UnitToInt16Scalar over an array (dsp_utils.h:76-88)."""
    L = load(oracle)
    x = np.ascontiguousarray(x, np.float32)
    out = np.empty(x.shape, np.int16)
    L.ref_unit_to_int16(_p(x), x.size, _p(out))
    return out


def int16_to_unit(oracle, x):
    """Int16ToUnitScalar<float> over an array (dsp_utils.h:104-108)."""
    L = load(oracle)
    x = np.ascontiguousarray(x, np.int16)
    out = np.empty(x.shape, np.float32)
    L.ref_int16_to_unit(_p(x), x.size, _p(out))
    return out


def log_spectral_distance(oracle, a, b):
    """LogSpectralDistance (dsp_utils.cc:27-41)."""
    L = load(oracle)
    a, b = np.ascontiguousarray(a, np.float32), np.ascontiguousarray(b, np.float32)
    return float(L.ref_log_spectral_distance(_p(a), _p(b), a.size))


def make_model_dir(path):
    """A directory LyraEncoder::Create / LyraDecoder::Create accept as model_path: AreParamsSupported
    (lyra_config.h:117-168) probes for the three asset names and reads lyra_config.binarypb (identifier = 3); the
    factories of oracle/ref_glue.cc never open the assets."""
    os.makedirs(path, exist_ok=True)
    for name in ("quantizer.tflite", "lyragan.tflite", "soundstream_encoder.tflite"):
        open(os.path.join(path, name), "wb").close()
    with open(os.path.join(path, "lyra_config.binarypb"), "wb") as f:
        f.write(bytes([0x08, 0x03]))
    return str(path)


def encode_file(oracle, wav_path, out_path, bitrate, model_dir, enable_dtx=False):
    """chromemedia::codec::EncodeFile (cli_example/encoder_main_lib.cc:99-140)."""
    L = load(oracle)
    return bool(L.ref_encode_file(str(wav_path).encode(), str(out_path).encode(), bitrate, 0, int(enable_dtx),
                                  str(model_dir).encode()))


def decode_file(oracle, encoded_path, out_path, sample_rate_hz, bitrate, model_dir, cng_seed=0, randomize_requests=False,
                loss_starts=(), loss_durations=()):
    """chromemedia::codec::DecodeFile (cli_example/decoder_main_lib.cc:142-222); loss pattern in seconds
    (FixedPacketLossModel) or none."""
    L = load(oracle)
    L.ref_set_cng_seed(cng_seed)
    a = np.ascontiguousarray(loss_starts, np.float32)
    b = np.ascontiguousarray(loss_durations, np.float32)
    return bool(L.ref_decode_file(str(encoded_path).encode(), str(out_path).encode(), sample_rate_hz, bitrate,
                                  int(randomize_requests), 0.0, 1.0, _p(a), _p(b), a.size, str(model_dir).encode()))
