"""ctypes binding of oracle/liblyra_oracle.so (TEST INFRASTRUCTURE ONLY).

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg import
this module.  See oracle/lyra_oracle.c for what it restates and how it is
pinned.
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
SO = os.path.join(HERE, "liblyra_oracle.so")
DEFAULT_PACK = os.path.join(HERE, "..", "lyra_amd", "assets", "lyra_v1.lyrapack")

MODES = {"exact": 0, "gemmlowp_double": 1, "xnnpack": 2, "builtin_mixed": 3}


def build(force=False):
    src = os.path.join(HERE, "lyra_oracle.c")
    if force or not os.path.exists(SO) or os.path.getmtime(SO) < os.path.getmtime(src):
        subprocess.check_call(["make", "-C", HERE, "-B", "liblyra_oracle.so"], stdout=subprocess.DEVNULL)
    return SO


_lib = None
_lib_xnn = None
SO_XNN = os.path.join(HERE, "_xnn", "liblyra_oracle_xnn.so")


def xnn_engine_available():
    """The second build of lyra_oracle.c (-DLO_XNNPACK): same plumbing, every arithmetic operator an XNNPACK operator of
    torch's libtorch_cpu.so -- bench.py's cpu_baseline_xnnpack (oracle/Makefile target _xnn/liblyra_oracle_xnn.so)."""
    return os.path.exists(SO_XNN)


def lib(engine="scalar"):
    global _lib, _lib_xnn
    if engine == "xnnpack":
        if _lib_xnn is None:
            import torch  # noqa: F401  (maps libtorch_cpu.so, which carries the XNNPACK, and its dependencies first)
            _lib_xnn = _bind(C.CDLL(SO_XNN))
            _lib_xnn.lo_engine.restype = C.c_char_p
            assert _lib_xnn.lo_engine() == b"xnnpack-operators"
        return _lib_xnn
    if _lib is None:
        if not os.path.exists(SO):
            build()
        _lib = _bind(C.CDLL(SO))
    return _lib


def _bind(L):
    if True:
        L.lo_load.restype = C.c_void_p
        L.lo_load.argtypes = [C.c_char_p, C.c_int]
        L.lo_free.argtypes = [C.c_void_p]
        L.lo_stream_new.restype = C.c_void_p
        L.lo_stream_reset.argtypes = [C.c_void_p]
        L.lo_stream_free.argtypes = [C.c_void_p]
        L.lo_stream_set_trace.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
        L.lo_stream_trace_count.argtypes = [C.c_void_p]
        L.lo_stream_trace_offset.argtypes = [C.c_void_p, C.c_int]
        L.lo_encode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_rvq_encode.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.lo_rvq_decode.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_noise_new.restype = C.c_void_p
        L.lo_noise_new.argtypes = [C.c_int, C.c_float, C.c_float]
        L.lo_unit_to_int16.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.lo_int16_to_unit.argtypes = [C.c_void_p, C.c_long, C.c_void_p]
        L.lo_noise_new_rate.restype = C.c_void_p
        L.lo_noise_new_rate.argtypes = [C.c_int]
        L.lo_noise_free.argtypes = [C.c_void_p]
        L.lo_noise_compute_is_noise.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_noise_update.argtypes = [C.c_void_p, C.c_void_p]
        L.lo_noise_receive.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_noise_get.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_resampler_new.restype = C.c_void_p
        L.lo_resampler_new.argtypes = [C.c_int, C.c_int]
        L.lo_resampler_free.argtypes = [C.c_void_p]
        L.lo_resampler_reset.argtypes = [C.c_void_p]
        L.lo_resample.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.lo_resampler_design.argtypes = [C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_cng_new.restype = C.c_void_p
        L.lo_cng_new.argtypes = [C.c_uint64]
        L.lo_cng_free.argtypes = [C.c_void_p]
        L.lo_cng_generate.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_rvq_encode_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_long, C.c_int, C.c_void_p, C.c_int]
        L.lo_pack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lo_unpack.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        L.lo_decode_frame.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_logmel.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        L.lo_run_batch.restype = C.c_double
        L.lo_run_batch.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p,
                                   C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        L.lo_set_canonical_last.argtypes = [C.c_int]
    return L


def _p(a):
    return a.ctypes.data_as(C.c_void_p) if a is not None else None


class Oracle:
    """The model (weights + requantisation mode)."""

    def __init__(self, pack=DEFAULT_PACK, mode="exact", engine="scalar"):
        """engine "scalar": the C restatement; "xnnpack": the same plumbing over XNNPACK operators (mode "xnnpack" only)."""
        self.L = lib(engine)
        self.engine = engine
        self.h = self.L.lo_load(os.path.abspath(pack).encode(), MODES[mode])
        if not self.h:
            raise RuntimeError(f"cannot load {pack}")
        self.mode = mode

    def __del__(self):
        try:
            self.L.lo_free(self.h)
        except Exception:
            pass

    def rvq_encode(self, feat, num_stages):
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1, 64)
        idx = np.empty((feat.shape[0], 46), np.int32)
        for i in range(feat.shape[0]):
            self.L.lo_rvq_encode(self.h, _p(feat[i]), num_stages, _p(idx[i]))
        return idx

    def rvq_encode_batch(self, feat, num_stages, threads=8):
        """Many vectors at once (threaded in C): float32 [n][64] -> int32 [n][46]."""
        feat = np.ascontiguousarray(feat, np.float32).reshape(-1, 64)
        idx = np.empty((feat.shape[0], 46), np.int32)
        self.L.lo_rvq_encode_batch(self.h, _p(feat), feat.shape[0], num_stages, _p(idx), threads)
        return idx

    def rvq_decode(self, idx):
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 46)
        out = np.empty((idx.shape[0], 64), np.float32)
        for i in range(idx.shape[0]):
            self.L.lo_rvq_decode(self.h, _p(idx[i]), _p(out[i]))
        return out

    def pack(self, idx, num_stages):
        idx = np.ascontiguousarray(idx, np.int32).reshape(-1, 46)
        out = np.zeros((idx.shape[0], num_stages // 2), np.uint8)
        for i in range(idx.shape[0]):
            self.L.lo_pack(_p(idx[i]), num_stages, _p(out[i]))
        return out

    def unpack(self, packets, num_stages):
        packets = np.ascontiguousarray(packets, np.uint8).reshape(-1, num_stages // 2)
        out = np.empty((packets.shape[0], 46), np.int32)
        for i in range(packets.shape[0]):
            self.L.lo_unpack(_p(packets[i]), num_stages, _p(out[i]))
        return out


class Stream:
    """Per-stream codec state (encoder + decoder + log-mel history)."""

    def __init__(self, oracle, trace_cap=0):
        self.o = oracle
        self.L = oracle.L
        self.h = self.L.lo_stream_new()
        self.trace = None
        if trace_cap:
            self.trace = np.zeros(trace_cap, np.float32)
            self.L.lo_stream_set_trace(self.h, _p(self.trace), trace_cap)

    def __del__(self):
        try:
            self.L.lo_stream_free(self.h)
        except Exception:
            pass

    def reset(self):
        self.L.lo_stream_reset(self.h)

    def _rearm(self):
        if self.trace is not None:
            self.L.lo_stream_set_trace(self.h, _p(self.trace), self.trace.size)

    def taps(self):
        n = self.L.lo_stream_trace_count(self.h)
        offs = [self.L.lo_stream_trace_offset(self.h, i) for i in range(n + 1)]
        return [self.trace[offs[i]:offs[i + 1]].copy() for i in range(n)]

    def encode(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size == 320
        feat = np.empty(64, np.float32)
        self._rearm()
        self.L.lo_encode_frame(self.o.h, self.h, _p(pcm), _p(feat))
        return feat

    def decode(self, feat, want_float=False):
        feat = np.ascontiguousarray(feat, np.float32)
        assert feat.size == 64
        pcm = np.empty(320, np.int16)
        pf = np.empty(320, np.float32)
        self._rearm()
        self.L.lo_decode_frame(self.o.h, self.h, _p(feat), _p(pcm), _p(pf))
        return (pcm, pf) if want_float else pcm

    def logmel(self, pcm):
        pcm = np.ascontiguousarray(pcm, np.int16)
        mel = np.empty(160, np.float32)
        self.L.lo_logmel(self.o.h, self.h, _p(pcm), _p(mel))
        return mel


class NoiseEstimator:
    """lyra/noise_estimator.{h,cc}: one instance per encoder (DTX) or decoder stream."""

    def __init__(self, oracle, num_hops_per_update=0, max_smoothing=0.0, bound_decay=0.0, sample_rate_hz=None):
        """sample_rate_hz: what NoiseEstimator::Create is given -- 16000 by the decoder, the EXTERNAL rate by a DTX
        encoder (lyra_encoder.cc:82-85); the time constants follow from it."""
        self.o = oracle
        self.L = oracle.L
        if sample_rate_hz is not None:
            self.h = self.L.lo_noise_new_rate(sample_rate_hz)
        else:
            self.h = self.L.lo_noise_new(num_hops_per_update, max_smoothing, bound_decay)

    def __del__(self):
        try:
            self.L.lo_noise_free(self.h)
        except Exception:
            pass

    def ReceiveSamples(self, pcm):
        """One full hop (320 samples) -> is_noise; also returns the hop's log-mel."""
        pcm = np.ascontiguousarray(pcm, np.int16)
        assert pcm.size == 320
        mel = np.empty(160, np.float32)
        return bool(self.L.lo_noise_receive(self.o.h, self.h, _p(pcm), _p(mel))), mel

    def UpdateNoiseEstimate(self, mel):
        mel = np.ascontiguousarray(mel, np.float32)
        self.L.lo_noise_update(self.h, _p(mel))

    def ComputeIsNoise(self, mel):
        mel = np.ascontiguousarray(mel, np.float32)
        return bool(self.L.lo_noise_compute_is_noise(self.h, _p(mel)))

    def noise_estimate(self):
        est = np.empty(160, np.float32)
        self.L.lo_noise_get(self.h, _p(est), None)
        return est

    def noise_bound(self):
        b = np.empty(160, np.float32)
        self.L.lo_noise_get(self.h, None, _p(b))
        return b


class Resampler:
    """lyra/resampler.{h,cc}: Create(input_rate, target_rate) + Resample(int16 span)."""

    def __init__(self, in_rate, out_rate):
        self.L = lib()
        self.h = self.L.lo_resampler_new(in_rate, out_rate)
        self.in_rate, self.out_rate = in_rate, out_rate

    def __del__(self):
        try:
            self.L.lo_resampler_free(self.h)
        except Exception:
            pass

    def Reset(self):
        self.L.lo_resampler_reset(self.h)

    def Resample(self, audio):
        audio = np.ascontiguousarray(audio, np.int16)
        out = np.empty(audio.size * max(1, self.out_rate // self.in_rate) + 8, np.int16)
        n = self.L.lo_resample(self.h, _p(audio), audio.size, _p(out))
        assert n >= 0
        return out[:n].copy()


class ComfortNoiseGenerator:
    """lyra/comfort_noise_generator.{h,cc}: one hop of noise per AddFeatures(log-mel[160])."""

    def __init__(self, oracle, seed=0):
        self.o, self.L = oracle, oracle.L
        self.h = self.L.lo_cng_new(seed)

    def __del__(self):
        try:
            self.L.lo_cng_free(self.h)
        except Exception:
            pass

    def generate(self, features):
        features = np.ascontiguousarray(features, np.float32)
        assert features.size == 160
        out = np.empty(320, np.int16)
        self.L.lo_cng_generate(self.o.h, self.h, _p(features), _p(out))
        return out


def run_batch(oracle, pcm, num_stages, do_decode=True, threads=1, want_feats=False):
    """pcm int16 [steps][B][320] -> dict(packets, pcm, feats, seconds, stage_seconds).

    Fresh zero state per stream; one stream per row, threads split the streams.
    """
    pcm = np.ascontiguousarray(pcm, np.int16)
    steps, B, hop = pcm.shape
    assert hop == 320
    L = oracle.L
    streams = [L.lo_stream_new() for _ in range(B)]
    arr = (C.c_void_p * B)(*streams)
    packets = np.zeros((steps, B, num_stages // 2), np.uint8)
    out = np.zeros((steps, B, 320), np.int16) if do_decode else None
    feats = np.zeros((steps, B, 64), np.float32) if want_feats else None
    stage = np.zeros(4, np.float64)
    sec = L.lo_run_batch(oracle.h, arr, B, steps, num_stages, int(do_decode), _p(pcm), _p(packets), _p(feats),
                         _p(out), threads, _p(stage))
    for s in streams:
        L.lo_stream_free(s)
    return dict(packets=packets, pcm=out, feats=feats, seconds=sec, stage_seconds=stage)
