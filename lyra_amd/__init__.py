"""lyra_amd -- MI355X-native Lyra v1.3.2 encode/decode hot path.

The product is the C-ABI shared library `liblyra_hip.so` (include/lyra_hip.h; sources in lyra_amd/csrc).
This package is only the Python mirror of the reference's plugin surface over that ABI
(lyra_amd/codec.py) plus the build helper.  There is no CPU fallback: importing works anywhere, but
creating a codec context without the built library or without a gfx950 device raises.
"""
from .codec import (HOP, NUM_FEATURES, LyraHip, LyraHipError, ResidualVectorQuantizer, SoundStreamEncoder,  # noqa: F401
                    LyraGanModel, LogMelSpectrogramExtractor, bitrate_to_num_bits, packet_size, build_library,
                    library_path, default_model_dir)
