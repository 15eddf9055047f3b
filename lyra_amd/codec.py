"""Python mirror of the reference plugin surface over the C ABI (include/lyra_hip.h).

Batched, array-in/array-out versions of
  FeatureExtractorInterface::Extract            lyra/feature_extractor_interface.h:32-39
  VectorQuantizerInterface::Quantize / DecodeToLossyFeatures
                                                lyra/vector_quantizer_interface.h:28-41
  GenerativeModelInterface::AddFeatures / GenerateSamples
                                                lyra/generative_model_interface.h:32-42
with the same names, argument meaning and error behaviour (None on failure, like std::nullopt), plus the
fused LyraEncoder::Encode / LyraDecoder::DecodeSamples steady-state path.  numpy arrays go through the
host-pointer entry points; torch CUDA tensors through the `_dev` ones (no copies, no sync).
"""
import ctypes as C
import os
import subprocess

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
HOP = 320
NUM_FEATURES = 64
NUM_MEL = 160
MAX_BITS = 184

_BITRATES = {3200: 64, 6000: 120, 9200: 184}  # lyra_config.cc:44-48


def bitrate_to_num_bits(bitrate):
    return _BITRATES[bitrate]


def packet_size(num_bits):
    return (num_bits + 7) // 8  # lyra_config.h: 8 / 15 / 23 bytes


def library_path():
    # LYRA_HIP_LIB: developer override used to A/B kernel variants on the GPU box
    return os.environ.get("LYRA_HIP_LIB") or os.path.join(HERE, "liblyra_hip.so")


def default_model_dir():
    return os.path.join(HERE, "assets")


def build_library(force=False):
    """Compile lyra_amd/csrc for gfx950 (hipcc cross-compiles without a GPU)."""
    src = os.path.join(HERE, "csrc")
    args = ["make", "-C", src, "-j8"] + (["-B"] if force else [])
    subprocess.check_call(args, stdout=subprocess.DEVNULL)
    return library_path()


class LyraHipError(RuntimeError):
    pass


class StepsDesc(C.Structure):
    """lyra_hip_steps (include/lyra_hip.h): many hops of B streams from one call."""
    _fields_ = [("d_stream_ids", C.c_void_p), ("B", C.c_int), ("num_bits", C.c_int), ("flags", C.c_uint),
                ("first_step", C.c_long), ("n_steps", C.c_int), ("ring", C.c_int), ("d_pcm_ring", C.c_void_p),
                ("d_packets", C.c_void_p * 2), ("d_packet_bytes", C.c_void_p * 2), ("d_pcm_out", C.c_void_p * 2),
                ("d_features", C.c_void_p), ("n_features", C.c_int), ("d_packet_ring", C.c_void_p), ("n_packet_ring", C.c_int), ("d_is_noise", C.c_void_p), ("external_rate", C.c_int),
                ("d_ext_out", C.c_void_p * 2)]


STEP_ENCODE, STEP_DECODE, STEP_DTX, STEP_DECODER_NOISE = 1, 2, 4, 8


_libs = {}


def _load(path=None):
    """The C-ABI library (default: library_path()); a second path loads a build variant beside it (lyra_amd/variants/)."""
    path = os.path.abspath(path or library_path())
    if path in _libs:
        return _libs[path]
    if not os.path.exists(path):
        raise LyraHipError(f"{path} not built (run `python -c 'import __graft_entry__ as g; g.build()'`); "
                           "there is no CPU fallback")
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64; if the system copy gets loaded first
    # (through liblyra_hip.so's dependency) and torch is imported later, torch finds "No HIP GPUs".  Callers that
    # mix this library with torch tensors (the *_dev entry points) are safe regardless of import order if torch's
    # runtime is the one already resident when liblyra_hip.so is opened.
    import importlib.util
    import sys
    if "torch" not in sys.modules and importlib.util.find_spec("torch") is not None and \
            os.environ.get("LYRA_HIP_NO_TORCH_PRELOAD") != "1":
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = C.CDLL(path)
    vp, ci, cp = C.c_void_p, C.c_int, C.c_char_p
    L.lyra_hip_create.argtypes = [cp, ci, ci, ci, C.POINTER(vp)]
    L.lyra_hip_create_from_image.argtypes = [cp, C.c_size_t, ci, ci, ci, C.POINTER(vp)]
    L.lyra_hip_destroy.argtypes = [vp]
    L.lyra_hip_last_error.restype = cp
    L.lyra_hip_last_error.argtypes = [vp]
    L.lyra_hip_reset_streams.argtypes = [vp, vp, ci]
    for name in ("extract", "generate", "logmel"):
        for suf in ("", "_dev"):
            getattr(L, f"lyra_hip_{name}{suf}").argtypes = [vp, vp, ci, vp, vp]
    for suf in ("", "_dev"):
        getattr(L, f"lyra_hip_rvq_encode{suf}").argtypes = [vp, ci, vp, ci, vp]
        getattr(L, f"lyra_hip_rvq_decode{suf}").argtypes = [vp, ci, vp, vp]
        getattr(L, f"lyra_hip_encode{suf}").argtypes = [vp, vp, ci, vp, ci, vp]
        getattr(L, f"lyra_hip_decode{suf}").argtypes = [vp, vp, ci, vp, ci, vp]
    for suf in ("", "_dev"):
        getattr(L, f"lyra_hip_noise_receive{suf}").argtypes = [vp, ci, vp, ci, vp, vp]
        getattr(L, f"lyra_hip_encode_dtx{suf}").argtypes = [vp, vp, ci, vp, ci, vp, vp]
    L.lyra_hip_noise_estimate.argtypes = [vp, ci, vp, ci, vp]
    for suf in ("", "_dev"):
        getattr(L, f"lyra_hip_resample{suf}").argtypes = [vp, ci, vp, ci, vp, ci, ci, ci, vp]
        getattr(L, f"lyra_hip_comfort_noise{suf}").argtypes = [vp, vp, ci, vp, vp]
    if hasattr(L, "lyra_hip_encode_begin"):   # (build variants older than round 6 do not carry the pipelined calls)
        L.lyra_hip_encode_begin.argtypes = [vp, vp, ci, vp, ci, ci, ci]
        L.lyra_hip_encode_end.argtypes = [vp, vp, vp]
        L.lyra_hip_decode_begin.argtypes = [vp, vp, ci, vp, ci]
        L.lyra_hip_decode_end.argtypes = [vp, vp]
        L.lyra_hip_twin_fetch_begin.argtypes = [vp, ci, ci, ci]
        L.lyra_hip_twin_fetch_end.argtypes = [vp, vp]
    if hasattr(L, "lyra_hip_encode_ext_dev"):   # (one hop at an external rate, one call per side: round 6)
        L.lyra_hip_encode_ext_dev.argtypes = [vp, vp, ci, vp, ci, ci, ci, vp, vp]
        L.lyra_hip_decode_ext_dev.argtypes = [vp, vp, ci, vp, ci, ci, ci, vp, vp, vp]
    L.lyra_hip_set_cng_seed.argtypes = [vp, C.c_uint64]
    L.lyra_hip_set_encoder_sample_rate.argtypes = [vp, C.c_int]
    L.lyra_hip_stream.restype = vp
    L.lyra_hip_stream.argtypes = [vp]
    L.lyra_hip_stream_decode.restype = vp
    L.lyra_hip_stream_decode.argtypes = [vp]
    L.lyra_hip_stream_quantizer.restype = vp
    L.lyra_hip_stream_quantizer.argtypes = [vp]
    L.lyra_hip_run_steps_dev.argtypes = [vp, C.POINTER(StepsDesc)]
    L.lyra_hip_synchronize.argtypes = [vp]
    L.lyra_hip_wait_for_stream.argtypes = [vp, vp]
    L.lyra_hip_stream_wait.argtypes = [vp, vp]
    L.lyra_hip_set_serial.argtypes = [vp, ci]
    L.lyra_hip_set_stream_priorities.argtypes = [vp, ci, ci, ci]
    L.lyra_hip_state_bytes_per_stream.restype = C.c_size_t
    L.lyra_hip_max_streams.argtypes = [vp]
    L.lyra_hip_profile_enable.argtypes = [vp, C.c_uint]
    L.lyra_hip_profile_sample.argtypes = [vp, ci]
    L.lyra_hip_profile_kernel_name.restype = cp
    L.lyra_hip_profile_kernel_name.argtypes = [ci]
    L.lyra_hip_profile_read.argtypes = [vp, vp, vp]
    L.lyra_hip_debug_read.restype = C.c_long
    L.lyra_hip_debug_read.argtypes = [vp, ci, vp, C.c_long]
    _libs[path] = L
    return L


def _np(a, dtype, shape):
    a = np.ascontiguousarray(a, dtype)
    return a.reshape(shape)


class LyraHip:
    """One GPU context: weights + per-stream state for `max_streams` streams."""

    def __init__(self, model_dir=None, device=0, max_streams=4096, requant="xnnpack", weights_image=None,
                 sub_batches=None, library=None):
        """sub_batches: split every `_dev` call into that many independent sub-batches on stream pairs of their own
        (the library's LYRA_HIP_SUBBATCHES switch, read when the context is created).  Pays when only ONE side is
        driven (decode-only at B = 8192: +6 %), not for interleaved encode + decode, where the two sides already
        overlap."""
        self.L = _load(library)   # library: path of a build variant (experiments); default liblyra_hip.so
        h = C.c_void_p()
        mode = {"exact": 0, "gemmlowp_double": 1, "xnnpack": 2, "builtin_mixed": 3}[requant]
        saved = os.environ.get("LYRA_HIP_SUBBATCHES")
        if sub_batches is not None:
            os.environ["LYRA_HIP_SUBBATCHES"] = str(int(sub_batches))
        try:
            self._create(h, mode, model_dir, device, max_streams, weights_image)
        finally:
            if sub_batches is not None:
                if saved is None:
                    del os.environ["LYRA_HIP_SUBBATCHES"]
                else:
                    os.environ["LYRA_HIP_SUBBATCHES"] = saved
        self.h = h
        self.device = device
        self.max_streams = max_streams
        self.requant = requant
        self.sub_batches = sub_batches

    def _create(self, h, mode, model_dir, device, max_streams, weights_image):
        if weights_image is not None:   # bytes of a lyra_v1.lyrapack (lyra_hip_create_from_image)
            weights_image = bytes(weights_image)
            rc = self.L.lyra_hip_create_from_image(weights_image, len(weights_image), device, max_streams, mode,
                                                   C.byref(h))
        else:
            rc = self.L.lyra_hip_create((model_dir or default_model_dir()).encode(), device, max_streams, mode,
                                        C.byref(h))
        if rc != 0:
            raise LyraHipError(f"lyra_hip_create failed ({rc}): {self.L.lyra_hip_last_error(None).decode()}")

    def close(self):
        if getattr(self, "h", None):
            self.L.lyra_hip_destroy(self.h)
            self.h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # -- helpers ----------------------------------------------------------------------------------
    def last_error(self):
        return self.L.lyra_hip_last_error(self.h).decode()

    def _chk(self, rc):
        if rc != 0:
            raise LyraHipError(f"lyra_hip error {rc}: {self.last_error()}")

    def stream_handle(self):
        """hipStream_t (int) of the encode side."""
        return self.L.lyra_hip_stream(self.h)

    def stream_handle_decode(self):
        """hipStream_t (int) of the decode side."""
        return self.L.lyra_hip_stream_decode(self.h)

    def stream_handle_quantizer(self):
        """hipStream_t (int) of the quantizer of encode_dev / encode_dtx_dev (their packets are written there)."""
        return self.L.lyra_hip_stream_quantizer(self.h)

    def synchronize(self):
        self._chk(self.L.lyra_hip_synchronize(self.h))

    def state_bytes_per_stream(self):
        return self.L.lyra_hip_state_bytes_per_stream()

    def reset(self, stream_ids=None):
        if stream_ids is None:
            self._chk(self.L.lyra_hip_reset_streams(self.h, None, 0))
            self.synchronize()
        else:
            ids = _np(stream_ids, np.int32, (-1,))
            self._chk(self.L.lyra_hip_reset_streams(self.h, ids.ctypes.data, ids.size))

    @staticmethod
    def _ids(stream_ids, B):
        if stream_ids is None:
            return np.arange(B, dtype=np.int32)
        return _np(stream_ids, np.int32, (B,))

    # -- numpy (host pointer) API --------------------------------------------------------------------
    def extract(self, pcm, stream_ids=None):
        """pcm int16 [B][320] -> features float32 [B][64] (SoundStreamEncoder::Extract)."""
        pcm = _np(pcm, np.int16, (-1, HOP))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty((B, NUM_FEATURES), np.float32)
        self._chk(self.L.lyra_hip_extract(self.h, ids.ctypes.data, B, pcm.ctypes.data, out.ctypes.data))
        return out

    def rvq_encode(self, features, num_bits):
        features = _np(features, np.float32, (-1, NUM_FEATURES))
        B = features.shape[0]
        idx = np.empty((B, 46), np.int32)
        self._chk(self.L.lyra_hip_rvq_encode(self.h, B, features.ctypes.data, num_bits, idx.ctypes.data))
        return idx

    def rvq_decode(self, indices):
        indices = _np(indices, np.int32, (-1, 46))
        B = indices.shape[0]
        out = np.empty((B, NUM_FEATURES), np.float32)
        self._chk(self.L.lyra_hip_rvq_decode(self.h, B, indices.ctypes.data, out.ctypes.data))
        return out

    def generate(self, features, stream_ids=None):
        """features [B][64] -> pcm int16 [B][320] (AddFeatures + GenerateSamples(320))."""
        features = _np(features, np.float32, (-1, NUM_FEATURES))
        B = features.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty((B, HOP), np.int16)
        self._chk(self.L.lyra_hip_generate(self.h, ids.ctypes.data, B, features.ctypes.data, out.ctypes.data))
        return out

    def logmel(self, pcm, stream_ids=None):
        pcm = _np(pcm, np.int16, (-1, HOP))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty((B, NUM_MEL), np.float32)
        self._chk(self.L.lyra_hip_logmel(self.h, ids.ctypes.data, B, pcm.ctypes.data, out.ctypes.data))
        return out

    def encode(self, pcm, num_bits, stream_ids=None):
        """pcm int16 [B][320] -> packets uint8 [B][num_bits/8] (LyraEncoder::Encode, 16 kHz, no DTX)."""
        pcm = _np(pcm, np.int16, (-1, HOP))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty((B, packet_size(num_bits)), np.uint8)
        self._chk(self.L.lyra_hip_encode(self.h, ids.ctypes.data, B, pcm.ctypes.data, num_bits, out.ctypes.data))
        return out

    def encode_begin(self, pcm, num_bits, stream_ids=None, sample_rate_hz=16000, dtx=False):
        """Pipelined form of encode() (include/lyra_hip.h "Pipelined host-buffer calls"): starts a hop and returns; up to
        two hops may be in flight; pcm int16 [B][sample_rate_hz / 50] at 8 / 16 / 32 / 48 kHz.  `pcm` may be reused at once."""
        pcm = _np(pcm, np.int16, (-1, sample_rate_hz // 50))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        if dtx:
            self._chk(self.L.lyra_hip_set_encoder_sample_rate(self.h, sample_rate_hz))
        self._chk(self.L.lyra_hip_encode_begin(self.h, ids.ctypes.data, B, pcm.ctypes.data, sample_rate_hz, num_bits, int(dtx)))
        self._pending_encodes = getattr(self, "_pending_encodes", []) + [(B, num_bits)]

    def encode_end(self):
        """-> (packets uint8 [B][num_bits/8], packet_bytes int32 [B]) of the OLDEST hop begun."""
        B, num_bits = self._pending_encodes.pop(0)
        out = np.empty((B, packet_size(num_bits)), np.uint8)
        lens = np.empty(B, np.int32)
        self._chk(self.L.lyra_hip_encode_end(self.h, out.ctypes.data, lens.ctypes.data))
        return out, lens

    def decode_begin(self, packets, num_bits, stream_ids=None):
        """Pipelined form of decode(): starts a call and returns (up to two in flight); decode_end() -> pcm of the oldest."""
        packets = _np(packets, np.uint8, (-1, packet_size(num_bits)))
        B = packets.shape[0]
        ids = self._ids(stream_ids, B)
        self._chk(self.L.lyra_hip_decode_begin(self.h, ids.ctypes.data, B, packets.ctypes.data, num_bits))
        self._pending_decodes = getattr(self, "_pending_decodes", []) + [B]

    def decode_end(self):
        out = np.empty((self._pending_decodes.pop(0), HOP), np.int16)
        self._chk(self.L.lyra_hip_decode_end(self.h, out.ctypes.data))
        return out

    def decode(self, packets, num_bits, stream_ids=None):
        """packets uint8 [B][num_bits/8] -> pcm int16 [B][320] (SetEncodedPacket + DecodeSamples(320))."""
        packets = _np(packets, np.uint8, (-1, packet_size(num_bits)))
        B = packets.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty((B, HOP), np.int16)
        self._chk(self.L.lyra_hip_decode(self.h, ids.ctypes.data, B, packets.ctypes.data, num_bits, out.ctypes.data))
        return out

    _SIDES = {"encoder": 0, "decoder": 1}

    def noise_receive(self, pcm, stream_ids=None, side="decoder"):
        """NoiseEstimator::ReceiveSamples, one full hop per stream -> is_noise int32 [B]."""
        pcm = _np(pcm, np.int16, (-1, HOP))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.empty(B, np.int32)
        self._chk(self.L.lyra_hip_noise_receive(self.h, self._SIDES[side], ids.ctypes.data, B, pcm.ctypes.data,
                                                out.ctypes.data))
        return out

    def noise_estimate(self, stream_ids, side="decoder"):
        """NoiseEstimator::noise_estimate() -> float32 [B][160]."""
        ids = _np(stream_ids, np.int32, (-1,))
        out = np.empty((ids.size, NUM_MEL), np.float32)
        self._chk(self.L.lyra_hip_noise_estimate(self.h, self._SIDES[side], ids.ctypes.data, ids.size, out.ctypes.data))
        return out

    def encode_dtx(self, pcm, num_bits, stream_ids=None):
        """LyraEncoder::Encode with enable_dtx -> (packets uint8 [B][nbytes], packet_bytes int32 [B]; 0 = empty packet)."""
        pcm = _np(pcm, np.int16, (-1, HOP))
        B = pcm.shape[0]
        ids = self._ids(stream_ids, B)
        out = np.zeros((B, packet_size(num_bits)), np.uint8)
        nbytes = np.empty(B, np.int32)
        self._chk(self.L.lyra_hip_encode_dtx(self.h, ids.ctypes.data, B, pcm.ctypes.data, num_bits, out.ctypes.data,
                                             nbytes.ctypes.data))
        return out, nbytes

    def resample(self, audio, in_rate, out_rate, stream_ids=None, side="encoder"):
        """Resampler::Resample per stream: int16 [B][n_in] -> int16 [B][n_in * out_rate / in_rate]."""
        audio = np.ascontiguousarray(audio, np.int16)
        B, n_in = audio.shape
        ids = self._ids(stream_ids, B)
        out = np.empty((B, n_in * out_rate // in_rate), np.int16)
        self._chk(self.L.lyra_hip_resample(self.h, self._SIDES[side], ids.ctypes.data, B, audio.ctypes.data, n_in,
                                           in_rate, out_rate, out.ctypes.data))
        return out

    def comfort_noise(self, features=None, stream_ids=None, B=None):
        """ComfortNoiseGenerator: one hop per stream; features float32 [B][160] or None (= the decoder-side noise estimate)."""
        if features is not None:
            features = _np(features, np.float32, (-1, NUM_MEL))
            B = features.shape[0]
        elif B is None and stream_ids is not None:
            B = len(stream_ids)
        ids = self._ids(stream_ids, B)
        out = np.empty((B, HOP), np.int16)
        self._chk(self.L.lyra_hip_comfort_noise(self.h, ids.ctypes.data, B,
                                                features.ctypes.data if features is not None else None, out.ctypes.data))
        return out

    def set_encoder_sample_rate(self, sample_rate_hz):
        """The rate a DTX LyraEncoder was created with: time constants of the encoder-side noise estimator
        (lyra_encoder.cc:82-85, noise_estimator.cc:96-124)."""
        self._chk(self.L.lyra_hip_set_encoder_sample_rate(self.h, sample_rate_hz))

    def set_cng_seed(self, seed):
        self._chk(self.L.lyra_hip_set_cng_seed(self.h, seed))

    def profile_kernel_names(self):
        n = self.L.lyra_hip_profile_kernel_count()
        return [self.L.lyra_hip_profile_kernel_name(i).decode() for i in range(n)]

    def set_stream_priorities(self, encode_side=0, decode_side=0, quantizer=2):
        """Priorities (0 lowest .. 2 highest) of the context's three main streams; drains the context (include/lyra_hip.h)."""
        self._chk(self.L.lyra_hip_set_stream_priorities(self.h, int(encode_side), int(decode_side), int(quantizer)))

    def profile_enable(self, on=True, only=None, every=1):
        """Bracket kernel launches with HIP events: all kernels, or only the named one(s); every `every`-th launch."""
        self._chk(self.L.lyra_hip_profile_sample(self.h, int(every)))
        mask = 0
        if on:
            names = self.profile_kernel_names()
            if only:
                for k in ([only] if isinstance(only, str) else only):
                    mask |= 1 << names.index(k)
            else:
                mask = (1 << len(names)) - 1
        self._chk(self.L.lyra_hip_profile_enable(self.h, mask))

    def profile_read(self):
        """{kernel name: (total_ms, launches)} since the last read (HIP events on the context's stream)."""
        n = self.L.lyra_hip_profile_kernel_count()
        ms = (C.c_double * n)()
        cnt = (C.c_long * n)()
        self._chk(self.L.lyra_hip_profile_read(self.h, ms, cnt))
        return {self.L.lyra_hip_profile_kernel_name(i).decode(): (ms[i], cnt[i]) for i in range(n)}

    def debug_read(self, which, n):
        out = np.empty(n, np.float32)
        got = self.L.lyra_hip_debug_read(self.h, which, out.ctypes.data, n)
        if got < 0:
            raise LyraHipError(self.last_error())
        return out[:got]

    # -- torch (device pointer) API ---------------------------------------------------------------------------
    # The library's streams are non-blocking: they do not order against torch's streams by themselves.  With
    # `torch_order=True` (default) every `_dev` call is bracketed by lyra_hip_wait_for_stream /
    # lyra_hip_stream_wait on torch's CURRENT stream, so tensors produced or consumed by torch kernels on that
    # stream are safe without a synchronize.  bench.py turns it off (it synchronises explicitly around the timed
    # region and wants no extra event traffic inside it).
    torch_order = True

    def set_serial(self, on=True):
        """Run the library streams strictly in call order (lyra_hip_set_serial)."""
        self._chk(self.L.lyra_hip_set_serial(self.h, 1 if on else 0))

    def _dev_ptr(self, t, dtype_name, shape, what):
        import torch
        want = getattr(torch, dtype_name)
        if not isinstance(t, torch.Tensor) or not t.is_cuda or t.device.index != self.device:
            raise LyraHipError(f"{what}: expected a CUDA tensor on device {self.device}")
        if t.dtype != want:
            raise LyraHipError(f"{what}: dtype {t.dtype}, expected {want}")
        if not t.is_contiguous():
            raise LyraHipError(f"{what}: tensor is not contiguous")
        if tuple(t.shape) != tuple(shape):
            raise LyraHipError(f"{what}: shape {tuple(t.shape)}, expected {tuple(shape)}")
        return t.data_ptr()

    def _torch_stream(self):
        import torch
        return torch.cuda.current_stream(self.device).cuda_stream

    def _dev_call(self, fn, *args):
        if self.torch_order:
            st = self._torch_stream()
            self._chk(self.L.lyra_hip_wait_for_stream(self.h, st))
            self._chk(fn(self.h, *args))
            self._chk(self.L.lyra_hip_stream_wait(self.h, st))
        else:
            self._chk(fn(self.h, *args))

    def encode_dev(self, d_ids, d_pcm, num_bits, d_packets):
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_encode_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"), num_bits,
                       self._dev_ptr(d_packets, "uint8", (B, packet_size(num_bits)), "packets"))

    def encode_dtx_dev(self, d_ids, d_pcm, num_bits, d_packets, d_packet_bytes):
        """LyraEncoder::Encode with enable_dtx on device buffers: packets uint8 [B][nbytes] (rows of noise hops are
        left untouched) and packet_bytes int32 [B] (0 = empty packet)."""
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_encode_dtx_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"), num_bits,
                       self._dev_ptr(d_packets, "uint8", (B, packet_size(num_bits)), "packets"),
                       self._dev_ptr(d_packet_bytes, "int32", (B,), "packet bytes"))

    def encode_ext_dev(self, d_ids, d_pcm_ext, sample_rate_hz, num_bits, d_packets, d_packet_bytes=None, dtx=False):
        """LyraEncoder::Encode at an external sample rate as ONE encode-side call (lyra_hip_encode_ext_dev): resampler, with
        dtx the NoiseEstimator decision (d_packet_bytes int32 [B] required), extractor, quantizer.  d_pcm_ext int16
        [B][320 * rate / 16000]."""
        B = d_pcm_ext.shape[0]
        n_ext = HOP * sample_rate_hz // 16000
        self._dev_call(self.L.lyra_hip_encode_ext_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_pcm_ext, "int16", (B, n_ext), "pcm"), sample_rate_hz, num_bits, 1 if dtx else 0,
                       self._dev_ptr(d_packets, "uint8", (B, packet_size(num_bits)), "packets"),
                       self._dev_ptr(d_packet_bytes, "int32", (B,), "packet bytes") if d_packet_bytes is not None else None)

    def decode_ext_dev(self, d_ids, d_packets, num_bits, sample_rate_hz, d_pcm16, d_pcm_ext=None, d_is_noise=None):
        """LyraDecoder::DecodeSamples for a received hop at an external rate as ONE decode-side call
        (lyra_hip_decode_ext_dev): decode -> d_pcm16 [B][320]; d_is_noise given: the decoder-side NoiseEstimator; rate !=
        16000: the resampler -> d_pcm_ext [B][320 * rate / 16000].  Estimator and resampler complete on the noise stream."""
        B = d_packets.shape[0]
        n_ext = HOP * sample_rate_hz // 16000
        self._dev_call(self.L.lyra_hip_decode_ext_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_packets, "uint8", (B, packet_size(num_bits)), "packets"), num_bits, sample_rate_hz,
                       1 if d_is_noise is not None else 0, self._dev_ptr(d_pcm16, "int16", (B, HOP), "pcm"),
                       self._dev_ptr(d_pcm_ext, "int16", (B, n_ext), "external-rate pcm") if d_pcm_ext is not None else None,
                       self._dev_ptr(d_is_noise, "int32", (B,), "is_noise") if d_is_noise is not None else None)

    def noise_receive_dev(self, d_ids, d_pcm, d_is_noise, side="decoder"):
        """NoiseEstimator::ReceiveSamples on device buffers: pcm int16 [B][320] -> is_noise int32 [B]."""
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_noise_receive_dev, self._SIDES[side], self._dev_ptr(d_ids, "int32", (B,), "stream ids"),
                       B, self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"), self._dev_ptr(d_is_noise, "int32", (B,), "is_noise"))

    def resample_dev(self, d_ids, d_in, in_rate, out_rate, d_out, side="encoder"):
        """Resampler::Resample per stream on device buffers: int16 [B][n_in] -> int16 [B][n_in * out_rate / in_rate]."""
        B, n_in = d_in.shape
        self._dev_call(self.L.lyra_hip_resample_dev, self._SIDES[side], self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_in, "int16", (B, n_in), "input audio"), n_in, in_rate, out_rate,
                       self._dev_ptr(d_out, "int16", (B, n_in * out_rate // in_rate), "output audio"))

    def comfort_noise_dev(self, d_ids, d_features, d_pcm):
        """ComfortNoiseGenerator on device buffers; d_features float32 [B][160] or None (= decoder-side noise estimate)."""
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_comfort_noise_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_features, "float32", (B, NUM_MEL), "features") if d_features is not None else None,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"))

    def run_steps_dev(self, d_ids, num_bits, n_steps, first_step=0, d_pcm_ring=None, d_packets=None, d_pcm_out=None,
                      d_features=None, d_packet_bytes=None, d_is_noise=None, external_rate=16000, d_ext_out=None,
                      encode=True, decode=True, dtx=False, decoder_noise=False, d_packet_ring=None):
        """lyra_hip_run_steps_dev: n_steps hops of every stream from ONE C call.  d_pcm_ring int16
        [ring][B][320 * external_rate / 16000]; d_packets / d_pcm_out / d_packet_bytes / d_ext_out: pairs of tensors
        (step i uses element (first_step + i) & 1)."""
        B = d_ids.shape[0]
        n_ext = HOP * external_rate // 16000
        S = StepsDesc()
        S.d_stream_ids = self._dev_ptr(d_ids, "int32", (B,), "stream ids")
        S.B, S.num_bits, S.first_step, S.n_steps = B, num_bits, first_step, n_steps
        S.flags = (STEP_ENCODE if encode else 0) | (STEP_DECODE if decode else 0) | (STEP_DTX if dtx else 0) | \
            (STEP_DECODER_NOISE if decoder_noise else 0)
        S.external_rate = external_rate
        if d_pcm_ring is not None:
            S.ring = d_pcm_ring.shape[0]
            S.d_pcm_ring = self._dev_ptr(d_pcm_ring, "int16", (S.ring, B, n_ext), "pcm ring")
        for i in range(2):
            if d_packets is not None:
                S.d_packets[i] = self._dev_ptr(d_packets[i], "uint8", (B, packet_size(num_bits)), "packets")
            if d_pcm_out is not None:
                S.d_pcm_out[i] = self._dev_ptr(d_pcm_out[i], "int16", (B, HOP), "pcm out")
            if d_packet_bytes is not None:
                S.d_packet_bytes[i] = self._dev_ptr(d_packet_bytes[i], "int32", (B,), "packet bytes")
            if d_ext_out is not None:
                S.d_ext_out[i] = self._dev_ptr(d_ext_out[i], "int16", (B, n_ext), "external-rate out")
        if d_features is not None:     # [B][64], or [n][B][64]: step i generates from frame (first_step + i) % n
            S.n_features = d_features.shape[0] if d_features.dim() == 3 else 1
            S.d_features = self._dev_ptr(d_features, "float32", (S.n_features, B, NUM_FEATURES) if d_features.dim() == 3
                                         else (B, NUM_FEATURES), "features")
        if d_packet_ring is not None:  # decode-only: received packets [n][B][bytes], step i decodes frame (first_step + i) % n
            S.n_packet_ring = d_packet_ring.shape[0]
            S.d_packet_ring = self._dev_ptr(d_packet_ring, "uint8", (S.n_packet_ring, B, packet_size(num_bits)), "packet ring")
        if d_is_noise is not None:
            S.d_is_noise = self._dev_ptr(d_is_noise, "int32", (B,), "is_noise")
        self._dev_call(self.L.lyra_hip_run_steps_dev, C.byref(S))

    def decode_dev(self, d_ids, d_packets, num_bits, d_pcm):
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_decode_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_packets, "uint8", (B, packet_size(num_bits)), "packets"), num_bits,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"))

    def extract_dev(self, d_ids, d_pcm, d_feat):
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_extract_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"),
                       self._dev_ptr(d_feat, "float32", (B, NUM_FEATURES), "features"))

    def generate_dev(self, d_ids, d_feat, d_pcm):
        B = d_feat.shape[0]
        self._dev_call(self.L.lyra_hip_generate_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_feat, "float32", (B, NUM_FEATURES), "features"),
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"))

    def logmel_dev(self, d_ids, d_pcm, d_mel):
        B = d_pcm.shape[0]
        self._dev_call(self.L.lyra_hip_logmel_dev, self._dev_ptr(d_ids, "int32", (B,), "stream ids"), B,
                       self._dev_ptr(d_pcm, "int16", (B, HOP), "pcm"),
                       self._dev_ptr(d_mel, "float32", (B, NUM_MEL), "mel"))

    def rvq_encode_dev(self, d_feat, num_bits, d_idx):
        B = d_feat.shape[0]
        self._dev_call(self.L.lyra_hip_rvq_encode_dev, B, self._dev_ptr(d_feat, "float32", (B, NUM_FEATURES), "features"),
                       num_bits, self._dev_ptr(d_idx, "int32", (B, 46), "indices"))

    def rvq_decode_dev(self, d_idx, d_feat):
        B = d_idx.shape[0]
        self._dev_call(self.L.lyra_hip_rvq_decode_dev, B, self._dev_ptr(d_idx, "int32", (B, 46), "indices"),
                       self._dev_ptr(d_feat, "float32", (B, NUM_FEATURES), "features"))


# ---------------------------------------------------------------------------------------------------
# single-stream plugin objects with the reference's method names and error behaviour
# ---------------------------------------------------------------------------------------------------
class _Plugin:
    def __init__(self, ctx, stream_id=0):
        self.ctx = ctx
        self.sid = np.array([stream_id], np.int32)


class SoundStreamEncoder(_Plugin):
    """FeatureExtractorInterface (lyra/soundstream_encoder.h)."""

    def Extract(self, audio):
        audio = np.asarray(audio)
        if audio.size != HOP:
            return None
        return self.ctx.extract(audio.astype(np.int16), self.sid)[0]


class ResidualVectorQuantizer(_Plugin):
    """VectorQuantizerInterface (lyra/residual_vector_quantizer.h): bit strings of '0'/'1', first
    quantizer in the most significant position."""

    def Quantize(self, features, num_bits):
        if num_bits > MAX_BITS or num_bits % 4 != 0 or num_bits < 0:
            return None  # residual_vector_quantizer.cc:79-89
        if num_bits == 0:
            return ""
        idx = self.ctx.rvq_encode(np.asarray(features, np.float32).reshape(1, 64), num_bits)[0]
        return "".join(format(int(i), "04b") for i in idx[:num_bits // 4])

    def DecodeToLossyFeatures(self, quantized_features):
        n = len(quantized_features)
        if n > MAX_BITS or n % 4 != 0:
            return None  # residual_vector_quantizer.cc:116-126
        idx = np.full(46, -1, np.int32)
        for i in range(n // 4):
            idx[i] = int(quantized_features[4 * i:4 * i + 4], 2)
        return self.ctx.rvq_decode(idx.reshape(1, 46))[0]


class LyraGanModel(_Plugin):
    """GenerativeModel FIFO semantics (lyra/generative_model_interface.h:45-134)."""

    def __init__(self, ctx, stream_id=0):
        super().__init__(ctx, stream_id)
        self._queue = []
        self._next = 0
        self._hop = None

    def AddFeatures(self, features):
        features = np.asarray(features, np.float32)
        if features.size != NUM_FEATURES:
            return False
        self._queue.append(features.copy())
        return True

    def num_samples_available(self):
        return len(self._queue) * HOP - self._next

    def GenerateSamples(self, num_samples):
        if num_samples < 0:
            return None
        if num_samples == 0:
            return np.zeros(0, np.int16)
        if self.num_samples_available() == 0:
            return None
        if self._next == 0:
            self._hop = self.ctx.generate(self._queue[0].reshape(1, 64), self.sid)[0]  # RunConditioning
        if num_samples > HOP - self._next:
            return None
        out = self._hop[self._next:self._next + num_samples].copy()  # RunModel
        self._next += num_samples
        if self._next == HOP:
            self._next = 0
            self._queue.pop(0)
        return out


class LogMelSpectrogramExtractor(_Plugin):
    """FeatureExtractorInterface (lyra/log_mel_spectrogram_extractor_impl.h), NoiseEstimator instantiation."""

    def Extract(self, audio):
        audio = np.asarray(audio)
        if audio.size != HOP:
            return None
        return self.ctx.logmel(audio.astype(np.int16), self.sid)[0]
