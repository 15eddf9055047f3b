// stand-in: the one audio_dsp helper lyra/noise_estimator.cc uses
#ifndef REF_SHIM_AUDIO_DSP_SIGNAL_VECTOR_UTIL_H_
#define REF_SHIM_AUDIO_DSP_SIGNAL_VECTOR_UTIL_H_
namespace audio_dsp {
template <typename T>
inline T Square(T x) { return x * x; }
}  // namespace audio_dsp
#endif
