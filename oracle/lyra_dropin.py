"""ctypes binding of oracle/_ref/liblyra_ref_hip.so (TEST INFRASTRUCTURE ONLY): the LITERAL drop-in of INTEGRATION.md 2a --
the reference's own lyra_encoder.cc / lyra_decoder.cc / lyra_benchmark_lib.cc / cli_example/*_main_lib.cc compiled from
/root/reference where they lie, with the three hot factories of lyra_components.cc:42-55 supplied by the product's
lyra_amd/host/lyra_hip_components.cc over liblyra_hip.so (oracle/ref_dropin.cc, oracle/Makefile).  The classes mirror
oracle/lyra_ref.py (the oracle-backed build of the same sources), so one session can be run through both and compared.
"""
import ctypes as C
import os
import shutil

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
LIB = os.path.join(HERE, "_ref", "liblyra_ref_hip.so")
PACK = os.path.join(HERE, "..", "lyra_amd", "assets", "lyra_v1.lyrapack")
_lib = None


def available():
    return os.path.exists(LIB)


def load(oracle, lib_path=None):
    """-> the library; `oracle` backs the audio_dsp shadows (log-mel of the NoiseEstimator, resampler, comfort noise)."""
    global _lib
    if _lib is None:
        # One HIP runtime per process (see lyra_amd/codec.py _load): the drop-in pulls in liblyra_hip.so and with it the
        # system libamdhip64; a torch imported afterwards would be handed that copy instead of its own and find no GPU.
        import importlib.util
        import sys
        if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None and \
                os.environ.get("LYRA_HIP_NO_TORCH_PRELOAD") != "1":
            try:
                import torch  # noqa: F401
            except Exception:
                pass
        C.CDLL(os.path.join(HERE, "liblyra_oracle.so"), mode=C.RTLD_GLOBAL)
        L = C.CDLL(lib_path or LIB)
        vp, ci = C.c_void_p, C.c_int
        L.dropin_set_oracle_model.argtypes = [vp]
        L.dropin_set_cng_seed.argtypes = [C.c_uint64]
        L.dropin_set_max_streams.argtypes = [ci]
        L.dropin_encoder_new.restype = vp
        L.dropin_encoder_new.argtypes = [ci, ci, ci, ci, C.c_char_p]
        L.dropin_encoder_free.argtypes = [vp]
        L.dropin_encoder_encode.argtypes = [vp, vp, ci, vp, ci]
        L.dropin_encoder_set_bitrate.argtypes = [vp, ci]
        L.dropin_decoder_new.restype = vp
        L.dropin_decoder_new.argtypes = [ci, ci, C.c_char_p]
        L.dropin_decoder_free.argtypes = [vp]
        L.dropin_decoder_set_packet.argtypes = [vp, vp, ci]
        L.dropin_decoder_decode.argtypes = [vp, ci, vp]
        L.dropin_decoder_is_comfort_noise.argtypes = [vp]
        L.dropin_lyra_benchmark.argtypes = [ci, C.c_char_p, ci, ci, ci]
        L.dropin_encode_file.argtypes = [C.c_char_p, C.c_char_p, ci, ci, ci, C.c_char_p]
        L.dropin_decode_file.argtypes = [C.c_char_p, C.c_char_p, ci, ci, C.c_char_p]
        L.dropin_call_stats.argtypes = [C.POINTER(C.c_long), C.POINTER(C.c_long)]
        _lib = L
    _lib.dropin_set_oracle_model(oracle.h)
    return _lib


def make_model_dir(path):
    """A model_path both sides accept: the asset names + lyra_config.binarypb that AreParamsSupported probes for
    (lyra_config.h:117-168; the reference's .tflite files are not in this repository, so the names are empty files) and
    the packed v1.3.2 coefficients lyra_hip_create reads first (lyra_v1.lyrapack)."""
    os.makedirs(path, exist_ok=True)
    for name in ("quantizer.tflite", "lyragan.tflite", "soundstream_encoder.tflite"):
        open(os.path.join(path, name), "wb").close()
    with open(os.path.join(path, "lyra_config.binarypb"), "wb") as f:
        f.write(bytes([0x08, 0x03]))
    shutil.copyfile(PACK, os.path.join(path, "lyra_v1.lyrapack"))
    return str(path)


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


class LyraEncoder:
    """chromemedia::codec::LyraEncoder::Create (lyra_encoder.cc:43-96) -- the public factory, HIP plugins inside."""

    def __init__(self, oracle, sample_rate_hz, bitrate, enable_dtx, model_dir):
        self.L = load(oracle)
        self.h = self.L.dropin_encoder_new(sample_rate_hz, 1, bitrate, int(enable_dtx), str(model_dir).encode())
        assert self.h, "LyraEncoder::Create returned nullptr"

    def close(self):
        if self.h:
            self.L.dropin_encoder_free(self.h)
            self.h = None

    __del__ = close

    def Encode(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        out = np.empty(32, np.uint8)
        n = self.L.dropin_encoder_encode(self.h, _p(audio), audio.size, _p(out), out.size)
        return None if n < 0 else out[:n].copy()

    def set_bitrate(self, bitrate):
        return bool(self.L.dropin_encoder_set_bitrate(self.h, bitrate))


class LyraDecoder:
    """chromemedia::codec::LyraDecoder::Create (lyra_decoder.cc:97-155)."""

    def __init__(self, oracle, sample_rate_hz, model_dir, cng_seed=0):
        self.L = load(oracle)
        self.L.dropin_set_cng_seed(cng_seed)
        self.h = self.L.dropin_decoder_new(sample_rate_hz, 1, str(model_dir).encode())
        assert self.h, "LyraDecoder::Create returned nullptr"

    def close(self):
        if self.h:
            self.L.dropin_decoder_free(self.h)
            self.h = None

    __del__ = close

    def SetEncodedPacket(self, packet):
        packet = np.ascontiguousarray(packet, np.uint8)
        return bool(self.L.dropin_decoder_set_packet(self.h, _p(packet), packet.size))

    def DecodeSamples(self, n):
        out = np.empty(max(n, 1), np.int16)
        got = self.L.dropin_decoder_decode(self.h, n, _p(out))
        return None if got < 0 else out[:got].copy()

    def is_comfort_noise(self):
        return bool(self.L.dropin_decoder_is_comfort_noise(self.h))


def lyra_benchmark(oracle, num_cond_vectors, model_dir, feature_extraction=True, quantizer=True, generative_model=True):
    """chromemedia::codec::lyra_benchmark (lyra_benchmark_lib.cc:199-293); the table goes to stderr and /tmp/benchmarks."""
    L = load(oracle)
    return L.dropin_lyra_benchmark(num_cond_vectors, str(model_dir).encode(), int(feature_extraction), int(quantizer),
                                   int(generative_model))


def encode_file(oracle, wav_path, out_path, bitrate, model_dir, enable_dtx=False):
    L = load(oracle)
    return bool(L.dropin_encode_file(str(wav_path).encode(), str(out_path).encode(), bitrate, 0, int(enable_dtx),
                                     str(model_dir).encode()))


def decode_file(oracle, encoded_path, out_path, sample_rate_hz, bitrate, model_dir):
    L = load(oracle)
    return bool(L.dropin_decode_file(str(encoded_path).encode(), str(out_path).encode(), sample_rate_hz, bitrate,
                                     str(model_dir).encode()))


def call_stats(oracle):
    L = load(oracle)
    a, b = C.c_long(0), C.c_long(0)
    L.dropin_call_stats(C.byref(a), C.byref(b))
    return a.value, b.value
