// C prototypes of the CPU oracle (oracle/lyra_oracle.c) used by the shadows and by oracle/ref_glue.cc.
#ifndef REF_ORACLE_API_H_
#define REF_ORACLE_API_H_
#include <cstdint>
extern "C" {
struct lo_model; struct lo_stream; struct lo_noise; struct lo_resampler; struct lo_cng;
lo_stream* lo_stream_new(void);
void lo_stream_free(lo_stream*);
void lo_encode_frame(const lo_model*, lo_stream*, const int16_t* pcm, float* feat);
void lo_decode_frame(const lo_model*, lo_stream*, const float* feat, int16_t* pcm, float* pcm_f);
void lo_rvq_encode(const lo_model*, const float* feat, int num_stages, int32_t* idx);
void lo_rvq_decode(const lo_model*, const int32_t* idx, float* feat);
void lo_logmel(const lo_model*, lo_stream*, const int16_t* pcm, float* mel);
void lo_logmel_rate(const lo_model*, lo_stream*, const int16_t* pcm, float* mel, int sample_rate_hz);
lo_resampler* lo_resampler_new(int in_rate, int out_rate);
void lo_resampler_free(lo_resampler*);
void lo_resampler_reset(lo_resampler*);
int lo_resample(lo_resampler*, const int16_t* in, int n_in, int16_t* out);
lo_cng* lo_cng_new(uint64_t seed);
void lo_cng_free(lo_cng*);
void lo_cng_generate(const lo_model*, lo_cng*, const float* features, int16_t* out);
}
// the oracle model every shadow component of this process computes with (set by ref_set_model, ref_glue.cc)
const lo_model* ref_model();
uint64_t ref_next_cng_seed();
#endif
