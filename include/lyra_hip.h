/*
 * lyra_hip.h -- C ABI of the MI355X-native Lyra encode/decode hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.
 * Each entry point names the reference interface it replaces (file:line
 * relative to the google/lyra v1.3.2 tree).  The C++ plugin adapters in
 * lyra_amd/host/ (FeatureExtractorInterface / VectorQuantizerInterface /
 * GenerativeModelInterface) and the Python mirror lyra_amd/codec.py are thin
 * callers of these functions; INTEGRATION.md shows the binding a reference
 * maintainer would add in lyra/lyra_components.cc:42-65.
 *
 * Model: one context = one GPU + four HIP streams (encode side / decode side / quantizer / decoder-side
 * noise estimator, see "Streams") + per-stream codec state for `max_streams` independent audio
 * streams.  A "frame" is one 20 ms hop of
 * 16 kHz audio (320 samples); the codec is streaming/causal, so stream `id`
 * must be fed its frames in order (the reference keeps this state inside the
 * TFLite interpreter's resource variables: lyra/tflite_model_wrapper.cc:36-121).
 *
 * Threading: calls on one context must be serialised by the caller (as for one
 * reference codec object); distinct contexts are independent.  A batch must not
 * name the same stream id twice (two frames of one stream in a call would race on its state): host-pointer variants
 * reject it with LYRA_HIP_EINVAL, `_dev` variants cannot look at the ids and rely on the caller.
 * All functions return 0 on success or a
 * negative LYRA_HIP_E* code; lyra_hip_last_error() describes the last failure.
 * There is NO CPU fallback: creating a context without a usable gfx950 device
 * fails.
 *
 * Pointer flavours: functions without suffix take HOST pointers (they copy
 * H2D/D2H and synchronise); `_dev` variants take DEVICE pointers, enqueue and
 * do not synchronise.
 *
 * Streams: a context runs four HIP streams -- the ENCODE side (extract,
 * rvq_encode, the feature extractor of encode), the DECODE side (rvq_decode,
 * generate, decode, logmel; the stateless helpers rvq_decode_dev / logmel_dev
 * count as decode-side calls too) and one for the quantizer of
 * lyra_hip_encode_dev / lyra_hip_encode_dtx_dev, which starts once the call's
 * features are ready and runs underneath the next call's feature extractor and
 * the previous call's decoder (a 46-stage dependent chain that would leave the
 * chip nearly idle if anything queued behind it), and one for the decoder-side
 * NoiseEstimator of lyra_hip_noise_receive_dev, which runs behind the decoder's
 * last stage underneath the next step.  Encoder and decoder state are disjoint.
 * What the library guarantees on the GPU, without any caller synchronisation:
 *   (1) a decode-side call is ordered after EVERY earlier encode-side call, so
 *       encode_dev -> decode_dev on the produced packets just works;
 *   (2) the outputs of an encode-side call are written after every earlier
 *       decode-side call EXCEPT (at most) THE MOST RECENT ONE has finished:
 *       encode of frame i+1 overlaps decode of frame i, but its packets never
 *       overtake the decode of frame i-1;
 *   (3) calls of one side apply to a stream's state in call order, whatever the
 *       order in which a call lists its streams -- also on a context that splits
 *       batches over several stream sets (LYRA_HIP_SUBBATCHES > 1: every split
 *       call then waits for all chunks of the previous call of its side; only
 *       inside lyra_hip_run_steps_dev, which has one id list for all its steps,
 *       do the chunks run as independent pipelines).
 * Two-buffer rule for `_dev` callers: alternate two packet/PCM buffer sets
 * (step i uses set i & 1).  By (2) the encode that rewrites set i & 1 at step
 * i+2 is ordered after the decode that read it at step i.  A caller that
 * reuses ONE buffer set must lyra_hip_synchronize() (or lyra_hip_set_serial)
 * between steps.
 * Small contexts (max_streams <= 1024) partition the chip: the encode-side and
 * quantizer streams run on one half of every XCD's CUs, the decode-side and
 * noise-estimator streams on the other (CU-masked streams).  At such batches a
 * stage kernel is 64-256 workgroups on 256 CUs and two concurrent dispatches
 * are placed independently of each other -- some CUs get two tiles, some none,
 * and a kernel lasts as long as its slowest tile; kept apart, config #2
 * (1,024 streams) gains 10 %.  Larger contexts share the whole chip (each chain
 * wants all of it in turn); LYRA_HIP_CU_MASKS overrides either way.
 * The library streams are hipStreamNonBlocking (CU-masked ones, created through
 * hipExtStreamCreateWithCUMask, are ordinary blocking streams with respect to
 * the NULL stream): they do NOT order against the
 * null stream or any stream of the caller.  A `_dev` caller that produces
 * inputs or consumes outputs on its own stream brackets the calls with
 * lyra_hip_wait_for_stream(ctx, s) (library work enqueued afterwards waits
 * for what is already enqueued on s) and lyra_hip_stream_wait(ctx, s) (work
 * enqueued on s afterwards waits for the library work enqueued so far), or
 * synchronises.  `_dev` calls run on the context's device regardless of the
 * caller's current device (which is restored on return).
 */
#ifndef LYRA_HIP_H_
#define LYRA_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LYRA_HIP_HOP 320          /* samples per 20 ms frame at 16 kHz (lyra_config.h:70-168) */
#define LYRA_HIP_NUM_FEATURES 64  /* kNumFeatures */
#define LYRA_HIP_NUM_MEL 160      /* kNumMelBins */
#define LYRA_HIP_MAX_STAGES 46    /* 184 bits / 4 bits per RVQ stage (lyra_config.cc:44-48) */

#define LYRA_HIP_EINVAL (-1)   /* bad argument (bit count, batch size, stream id, null pointer) */
#define LYRA_HIP_ENODEV (-2)   /* no usable gfx950 device */
#define LYRA_HIP_EMODEL (-3)   /* model directory / weight container unreadable or wrong version */
#define LYRA_HIP_EHIP (-4)     /* HIP runtime error */
#define LYRA_HIP_ENOMEM (-5)

/* Arithmetic flavour of the graphs' int8 regions (oracle/lyra_oracle.c header):
 *   XNNPACK          what the reference runs -- it executes both graphs through TFLite's XNNPACK delegate
 *                    (soundstream_encoder.cc:39-40, lyra_gan_model.cc:39-40, tflite_model_wrapper.cc:63-85): QS8 convolutions
 *                    requantised in fp32, XNNPACK's own int8 LeakyReLU / ADD / QUANTIZE kernels.  Every formula is held
 *                    against real XNNPACK code (tests/test_xnnpack_witness.py).  The default.
 *   EXACT, GEMMLOWP_DOUBLE
 *                    TFLite's builtin kernels (what the reference falls back to when the delegate cannot be applied,
 *                    tflite_model_wrapper.cc:76-78): Q31 single-rounding or gemmlowp double-rounding convolutions, gemmlowp
 *                    LeakyReLU / ADD, round-half-away QUANTIZE.
 *   BUILTIN_MIXED    what the graphs compute if the delegate takes the fp32 operators but NOT the signed-int8 ones: the
 *                    reference ORs in only TFLITE_XNNPACK_DELEGATE_FLAG_QU8 (tflite_model_wrapper.cc:65-67) and its graphs are
 *                    QS8, so whether the int8 regions reach XNNPACK depends on a build-time default of TFLite 2.11's delegate
 *                    that cannot be observed offline.  TFLite's builtin int8 kernels per operator: ungrouped
 *                    CONV_2D single rounding, grouped CONV_2D / DEPTHWISE_CONV_2D / TRANSPOSE_CONV double rounding, builtin
 *                    LeakyReLU / ADD / QUANTIZE.  Whichever way that default falls, a bit-exact mode exists; switching is this
 *                    one argument.
 * The fp32 layers are the same in all four: bias-first fused chains (XNNPACK's order). */
#define LYRA_HIP_REQUANT_EXACT 0
#define LYRA_HIP_REQUANT_GEMMLOWP_DOUBLE 1
#define LYRA_HIP_REQUANT_XNNPACK 2
#define LYRA_HIP_REQUANT_BUILTIN_MIXED 3
#define LYRA_HIP_REQUANT_DEFAULT LYRA_HIP_REQUANT_XNNPACK

typedef struct lyra_hip_ctx lyra_hip_ctx;

/* Replaces CreateFeatureExtractor / CreateQuantizer / CreateGenerativeModel
 * (lyra/lyra_components.cc:42-55) + TfLiteModelWrapper::Create
 * (lyra/tflite_model_wrapper.cc:36-95).  `model_dir` is either the reference's
 * own model directory (soundstream_encoder.tflite, quantizer.tflite,
 * lyragan.tflite, lyra_config.binarypb -- converted in memory) or a directory
 * holding the pre-packed lyra_v1.lyrapack (tools/pack_weights.py / pack_tool
 * output for that directory; used first if present).  Version identifier 3 as
 * lyra_config.h:145-166.  The container is validated (bounds, dtypes, the
 * layer shapes the kernels are specialised to): a truncated or foreign file
 * gives LYRA_HIP_EMODEL.
 * Developer switches read from the environment here (results are bit-identical
 * either way; each exists for an A/B recorded in DESIGN.md):
 *   LYRA_HIP_SUBBATCHES=<n>  split every `_dev` call into n sub-batches on stream
 *                            sets of their own (pays when only one side is driven:
 *                            decode-only at B = 8192 +6 % with n = 2; default 1).
 *                            Calls that are not split (resample, noise estimator,
 *                            small batches ...) are ordered after every chunk of
 *                            the split call before them and vice versa;
 *   LYRA_HIP_FUSED=<mask>    (variant build `make parked` only) bit 0: the encoder side as one
 *                            launch instead of three, bit 1: the decoder side likewise, bit 2: encoder
 *                            stages 1 + 2 as one launch, bit 3: decoder stages 0 + 1 (all slower at
 *                            B = 4096; default 0);
 *   LYRA_HIP_RVQ_WIDE=1      the 104 KB / 244-VGPR quantizer kernel;
 *   LYRA_HIP_FLAT_PRIO=1     all library streams at the same priority;
 *   LYRA_HIP_EVENT_FENCE=1   internal events with system-scope fences;
 *   LYRA_HIP_NO_CODE_WARM=1  skip the stage kernels' instruction pre-fetch;
 *   LYRA_HIP_CU_MASKS=e,d,q,n  CU-mask patterns (32-bit hex, repeated over the chip) of the encode / decode / quantizer /
 *                            noise streams; 0 = no mask (default: 00ff00ff,ff00ff00,00ff00ff,ff00ff00 when
 *                            max_streams <= 1024, none above);
 *   LYRA_HIP_PRIO=e,d,q      stream priorities of the encode / decode / quantizer streams (0 lowest .. 2 highest;
 *                            default 0,0,2: the two chains equal, the small quantizer first; 0,2,0 = rounds 2-3:
 *                            decoder chain first -- better for blocking decode calls beside an encoder, bimodal for the `_dev` pipeline);
 *   LYRA_HIP_TILE_DIV_<K>=k  launch stage kernel K (ENC_S0 .. DEC_S2) as k slices of its tiles,
 *   LYRA_HIP_LDS_PAD_<K>=b   give its workgroups b extra bytes of LDS (occupancy experiments; K also logmel_noise, resample);
 *   read by lyra_hip_run_steps_dev (profiles/r06_modes_timelines.txt):
 *   LYRA_HIP_RS_LEAD=1       the input resampler one hop ahead of the extractor instead of two;
 *   LYRA_HIP_RS_OUT_ON_CHAIN_SPLIT=1  on contexts that split batches: the output resampler on the decode streams (old form);
 *   LYRA_HIP_SPLIT_SN_CALLS=1  a hop's decoder-side NoiseEstimator and output resampler as two noise-stream calls
 *                            instead of one (the form before round 6's last session: the quantizer of hop i then
 *                            waits for the estimator of hop i - 1). */
/* max_streams: 1 .. 289,262 per context (per-stream state is addressed with 32-bit byte offsets; 83 KB of state per stream,
 * so that is 24 GB of the 288 -- more streams: more contexts). */
int lyra_hip_create(const char* model_dir, int device, int max_streams, int requant_mode, lyra_hip_ctx** out);
/* The same from an in-memory lyra_v1.lyrapack image (e.g. read once by rank 0 and broadcast to the other GPUs' ranks
 * over RCCL); the image is copied, the caller keeps ownership. */
int lyra_hip_create_from_image(const void* image, size_t image_bytes, int device, int max_streams, int requant_mode,
                               lyra_hip_ctx** out);
void lyra_hip_destroy(lyra_hip_ctx* ctx);
const char* lyra_hip_last_error(const lyra_hip_ctx* ctx); /* ctx may be NULL: last create() error */

/* Replaces TfLiteModelWrapper::ResetVariableTensors (tflite_model_wrapper.cc:111-113) /
 * constructing fresh codec objects.  ids == NULL resets every stream. */
int lyra_hip_reset_streams(lyra_hip_ctx* ctx, const int32_t* stream_ids, int n);

/* ---- per-plugin entry points (the three reference interfaces) -------------------------------- */

/* FeatureExtractorInterface::Extract as implemented by SoundStreamEncoder::Extract
 * (lyra/soundstream_encoder.cc:53-64): pcm [B][320] int16 -> features [B][64] f32. */
int lyra_hip_extract(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const int16_t* pcm, float* features);

/* VectorQuantizerInterface::Quantize (lyra/residual_vector_quantizer.cc:77-110), stateless:
 * features [B][64] -> indices [B][46] int32 (-1 beyond num_bits/4 stages).  num_bits <= 184, % 4 == 0. */
int lyra_hip_rvq_encode(lyra_hip_ctx* ctx, int B, const float* features, int num_bits, int32_t* indices);

/* VectorQuantizerInterface::DecodeToLossyFeatures (residual_vector_quantizer.cc:112-168), stateless:
 * indices [B][46] (-1 = unused stage) -> lossy features [B][64]. */
int lyra_hip_rvq_decode(lyra_hip_ctx* ctx, int B, const int32_t* indices, float* features);

/* GenerativeModel::AddFeatures + GenerateSamples(320) as implemented by LyraGanModel
 * (lyra/generative_model_interface.h:50-101, lyra/lyra_gan_model.cc:53-64):
 * features [B][64] -> pcm [B][320] int16 (x32768, clamp, truncate: dsp_utils.h:54-88). */
int lyra_hip_generate(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const float* features, int16_t* pcm);

/* FeatureExtractorInterface::Extract as implemented by LogMelSpectrogramExtractorImpl::Extract
 * (lyra/log_mel_spectrogram_extractor_impl.cc:96-126), the NoiseEstimator front end
 * (noise_estimator.cc:157-160): pcm [B][320] int16 -> log-mel [B][160] f32.  Keeps its own
 * per-stream 320-sample history. */
int lyra_hip_logmel(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const int16_t* pcm, float* mel);

/* NoiseEstimator (lyra/noise_estimator.h:45-60), one instance per stream and side: side LYRA_HIP_SIDE_ENCODER is the
 * estimator LyraEncoder owns for DTX (lyra_encoder.cc:81-83), LYRA_HIP_SIDE_DECODER the one LyraDecoder feeds with every
 * decoded hop of a received packet (lyra_decoder.cc:304-311).  noise_receive = ReceiveSamples for one full 320-sample
 * hop per stream (its own log-mel front end + ComputeIsNoise + DecayBounds / UpdateNoiseEstimate,
 * noise_estimator.cc:144-245) and returns is_noise() per stream (int32 0/1); noise_estimate = noise_estimate(),
 * [B][160] log-mel bins. */
#define LYRA_HIP_SIDE_ENCODER 0
#define LYRA_HIP_SIDE_DECODER 1
int lyra_hip_noise_receive(lyra_hip_ctx* ctx, int side, const int32_t* stream_ids, int B, const int16_t* pcm,
                           int32_t* is_noise);
int lyra_hip_noise_estimate(lyra_hip_ctx* ctx, int side, const int32_t* stream_ids, int B, float* estimate);

/* Resampler::Resample (lyra/resampler.cc:57-62), one instance per stream and side: LYRA_HIP_SIDE_ENCODER is the
 * resampler LyraEncoder applies to incoming audio (external rate -> 16 kHz, lyra_encoder.cc:59-66,119-122),
 * LYRA_HIP_SIDE_DECODER the one behind LyraDecoder's BufferedResampler (16 kHz -> external, lyra_decoder.cc:107-113).
 * in [B][n_in] int16 -> out [B][n_in * out_rate / in_rate]; rates from {8000, 16000, 32000, 48000}, one of them
 * 16000; n_in <= 960 and a multiple of in_rate / gcd.  audio_dsp::QResampler is restated (Kaiser-windowed sinc,
 * radius 17 input samples, primed: output delayed by 17 input samples); see oracle/lyra_oracle.c for the parity note. */
int lyra_hip_resample(lyra_hip_ctx* ctx, int side, const int32_t* stream_ids, int B, const int16_t* in, int n_in,
                      int in_rate, int out_rate, int16_t* out);

/* ComfortNoiseGenerator::AddFeatures + GenerateSamples(320) (lyra/comfort_noise_generator.cc:74-119): one 20 ms hop
 * of noise per stream whose 160-bin log-mel matches `features` [B][160]; features == NULL uses each stream's decoder-side
 * noise estimate, as LyraDecoder::RunComfortNoiseGenerator does (lyra_decoder.cc:328-340).  Phases come from a
 * counter-based generator (seed, stream id, hop, bin) instead of the reference's non-deterministic absl::BitGen. */
int lyra_hip_comfort_noise(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const float* features, int16_t* pcm);
int lyra_hip_set_cng_seed(lyra_hip_ctx* ctx, uint64_t seed);
/* The sample rate LyraEncoder::Create was called with, for the ENCODER-side noise estimator of the calls that follow
 * (lyra_hip_encode_dtx[_dev], lyra_hip_noise_receive[_dev] side ENCODER): the reference hands NoiseEstimator::Create
 * its external rate together with the internal 320-sample hop (lyra_encoder.cc:82-85), so the estimator's update period
 * and half-lives, counted in hops, depend on it (noise_estimator.cc:96-124).  8000 / 16000 (default) / 32000 / 48000.
 * A host-side setting read at enqueue time: set it before each call when encoders of different rates share a context. */
int lyra_hip_set_encoder_sample_rate(lyra_hip_ctx* ctx, int sample_rate_hz);

/* ---- fused paths -------------------------------------------------------------------------------- */

/* LyraEncoder::Encode without resampling/DTX (lyra/lyra_encoder.cc:143-155): Extract -> Quantize ->
 * Packet<>::PackQuantized (lyra/packet.h:91-122, zero header bits).
 * pcm [B][320] -> packets [B][num_bits/8 rounded up] (8 / 15 / 23 bytes for 64 / 120 / 184 bits). */
int lyra_hip_encode(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const int16_t* pcm, int num_bits,
                    uint8_t* packets);

/* LyraEncoder::Encode with enable_dtx = true (lyra/lyra_encoder.cc:131-156): every hop updates the stream's
 * encoder-side NoiseEstimator; a hop that is noise yields an EMPTY packet (packet_bytes[i] = 0, the packet row is left
 * zero) and does not run the feature extractor, so the encoder state of that stream does not advance; any other hop
 * is encoded as lyra_hip_encode does (packet_bytes[i] = num_bits / 8 rounded up). */
int lyra_hip_encode_dtx(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const int16_t* pcm, int num_bits,
                        uint8_t* packets, int32_t* packet_bytes);

/* LyraDecoder::SetEncodedPacket + DecodeSamples(320) steady state, no loss/PLC
 * (lyra/lyra_decoder.cc:172-226,284-326): unpack -> DecodeToLossyFeatures -> generative model. */
int lyra_hip_decode(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const uint8_t* packets, int num_bits,
                    int16_t* pcm);

/* ---- device-pointer variants (benchmark / pipelines that keep data resident in HBM) ------------- */
int lyra_hip_extract_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const int16_t* d_pcm,
                         float* d_features);
int lyra_hip_rvq_encode_dev(lyra_hip_ctx* ctx, int B, const float* d_features, int num_bits, int32_t* d_indices);
int lyra_hip_rvq_decode_dev(lyra_hip_ctx* ctx, int B, const int32_t* d_indices, float* d_features);
int lyra_hip_generate_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const float* d_features,
                          int16_t* d_pcm);
int lyra_hip_logmel_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const int16_t* d_pcm, float* d_mel);
int lyra_hip_encode_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const int16_t* d_pcm, int num_bits,
                        uint8_t* d_packets);
int lyra_hip_decode_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const uint8_t* d_packets,
                        int num_bits, int16_t* d_pcm);
int lyra_hip_resample_dev(lyra_hip_ctx* ctx, int side, const int32_t* d_stream_ids, int B, const int16_t* d_in, int n_in,
                          int in_rate, int out_rate, int16_t* d_out);
int lyra_hip_comfort_noise_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const float* d_features,
                               int16_t* d_pcm);
int lyra_hip_noise_receive_dev(lyra_hip_ctx* ctx, int side, const int32_t* d_stream_ids, int B, const int16_t* d_pcm,
                               int32_t* d_is_noise);
/* (rows of d_packets that belong to noise hops are not written) */
int lyra_hip_encode_dtx_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const int16_t* d_pcm, int num_bits,
                            uint8_t* d_packets, int32_t* d_packet_bytes);

/* One hop at an EXTERNAL sample rate (8000 / 16000 / 32000 / 48000 Hz), ONE call per side -- the per-call form of what
 * LyraEncoder::Encode and LyraDecoder::DecodeSamples do around the codec for callers whose audio arrives hop by hop:
 *   encode_ext: the encoder's resampler (lyra_encoder.cc:119-122), with dtx != 0 the NoiseEstimator decision (:131-141; then
 *     d_packet_bytes [B] is required and lyra_hip_set_encoder_sample_rate(rate) must have been called), feature extractor,
 *     quantizer.  d_pcm_ext int16 [B][320 * rate / 16000].
 *   decode_ext: decode of a received hop into d_pcm16 [B][320], with estimate_noise != 0 the decoder-side NoiseEstimator
 *     (lyra_decoder.cc:304-311 -> d_is_noise [B]), the resampler to the external rate (:107-113 -> d_pcm_ext
 *     [B][320 * rate / 16000]; may be NULL at 16000).  The estimator and the resampler complete on the NOISE stream
 *     (lyra_hip_stream_noise / lyra_hip_stream_wait / lyra_hip_synchronize, as for lyra_hip_run_steps_dev).
 * Same results as lyra_hip_resample_dev + lyra_hip_encode[_dtx]_dev and lyra_hip_decode_dev + lyra_hip_noise_receive_dev +
 * lyra_hip_resample_dev.  The difference is what rule (2) of "Streams" counts: each of these is ONE call of its side, so
 * the next hop's encode overlaps this hop's decode; issued one by one, a hop makes two encode-side and up to three
 * decode-side calls and the next extractor / quantizer wait for this hop's decoder chain (4096 streams at 48 kHz: 10.4 M
 * frames/s call by call, 13.3 M through these two -- what lyra_hip_run_steps_dev reaches; profiles/r06_per_call.txt). */
int lyra_hip_encode_ext_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const int16_t* d_pcm_ext, int sample_rate_hz,
                            int num_bits, int dtx, uint8_t* d_packets, int32_t* d_packet_bytes);
int lyra_hip_decode_ext_dev(lyra_hip_ctx* ctx, const int32_t* d_stream_ids, int B, const uint8_t* d_packets, int num_bits,
                            int sample_rate_hz, int estimate_noise, int16_t* d_pcm16, int16_t* d_pcm_ext, int32_t* d_is_noise);

/* n_steps hops of B streams from ONE call (no host language in the loop): per hop what lyra_benchmark times
 * (lyra_benchmark_lib.cc:121-160) and what LyraEncoder::Encode / LyraDecoder::DecodeSamples run around it.  Step i
 * (absolute number first_step + i) reads input frame (first_step + i) % ring and uses buffer set (first_step + i) & 1
 * of every two-element array (the two-buffer rule above).  Same results, hop for hop, as the individual `_dev` calls
 * in the order resample -> encode[_dtx] -> decode | generate -> noise_receive -> resample. */
#define LYRA_HIP_STEP_ENCODE 1u         /* lyra_hip_encode_dev (lyra_encoder.cc:143-155) */
#define LYRA_HIP_STEP_DECODE 2u         /* lyra_hip_decode_dev on the packets of this step -- or, d_features != NULL,
                                           lyra_hip_generate_dev on those features (lyra_gan_model path) */
#define LYRA_HIP_STEP_DTX 4u            /* encode with enable_dtx: lyra_hip_encode_dtx_dev (lyra_encoder.cc:131-141) */
#define LYRA_HIP_STEP_DECODER_NOISE 8u  /* NoiseEstimator::ReceiveSamples on every decoded hop (lyra_decoder.cc:304-311) */
typedef struct lyra_hip_steps {
  const int32_t* d_stream_ids;   /* [B] */
  int B;
  int num_bits;
  unsigned flags;                /* LYRA_HIP_STEP_* */
  long first_step;
  int n_steps;
  int ring;                      /* input frames in d_pcm_ring */
  const int16_t* d_pcm_ring;     /* [ring][B][320 * external_rate / 16000] */
  uint8_t* d_packets[2];         /* [B][num_bits / 8 rounded up] each */
  int32_t* d_packet_bytes[2];    /* [B] each (DTX) */
  int16_t* d_pcm_out[2];         /* [B][320] each: decoder output at 16 kHz */
  const float* d_features;       /* [n_features][B][64] or NULL: step `step` generates from frame step % n_features */
  int n_features;                /* frames in d_features (0 is read as 1) */
  const uint8_t* d_packet_ring;  /* DECODE without ENCODE: [n_packet_ring][B][bytes] received packets, or NULL (then
                                    d_packets[step & 1] is decoded as it stands) */
  int n_packet_ring;
  int32_t* d_is_noise;           /* [B] (DECODER_NOISE) */
  int external_rate;             /* 0 / 16000: none; 8000 / 32000 / 48000: the encoder's and the decoder's resampler
                                    (lyra_encoder.cc:119-122, lyra_decoder.cc:107-113) around the codec */
  int16_t* d_ext_out[2];         /* [B][320 * external_rate / 16000] each: decoder output at the external rate */
} lyra_hip_steps;
int lyra_hip_run_steps_dev(lyra_hip_ctx* ctx, const lyra_hip_steps* steps);

/* ---- Decoder twin: the device half of a batched LyraDecoder (lyra_amd/host/lyra_batch_codec.cc) ------------------------
 * LyraDecoder::DecodeSamplesInternal (lyra_decoder.cc:228-315) keeps per stream the conditioned hop of the generative
 * model and of the comfort-noise generator and hands out slices of them.  With these calls the two hops of every stream
 * stay on the device (arrays indexed by stream id); the caller runs the reference's per-stream state machine on integers
 * and describes each round of its loop.  All calls take HOST pointers, copy their arguments at call time, enqueue on
 * the decode-side stream and do NOT synchronise -- except lyra_hip_twin_fetch, which ends the request with one
 * device-to-host copy.  Per DecodeSamples call: packets + a few integers per stream up, the result down. */
/* RunConditioning for streams whose next hop comes from a received packet (SetEncodedPacket's DecodeToLossyFeatures +
 * AddFeatures happen here too, lyra_decoder.cc:198-206): packets [B][bytes of num_bits] -> generative-model hop of ids[b] */
int lyra_hip_twin_decode(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const uint8_t* packets, int num_bits);
/* ... from estimated features (packet loss concealment, ZeroFeatureEstimator::Estimate; lyra_decoder.cc:317-326) */
int lyra_hip_twin_conceal(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B);
/* comfort-noise hop of ids[b] from that stream's decoder-side noise estimate (lyra_decoder.cc:328-340) */
int lyra_hip_twin_comfort_noise(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B);
/* One pass of the reference's while loop for one stream: gen_n samples of the generative-model hop from gan_off and / or
 * cng_n samples of the comfort-noise hop from cng_off (equal when both are non-zero: cross-faded with fade_progress
 * `fade` stepping by fade_dir = +-1 per sample, MaybeOverlapAndInsert lyra_decoder.cc:342-373) to samples out_off.. of
 * row `id` of the request's output.  noise_row >= 0: the slice completes a received hop; its 320 samples become input
 * row noise_row of the lyra_hip_twin_noise call that follows (-1 otherwise). */
typedef struct lyra_hip_twin_slice {
  int32_t id, gan_off, gen_n, cng_off, cng_n, fade, fade_dir, out_off, noise_row;
} lyra_hip_twin_slice;
/* out_samples: internal-rate samples per stream of the whole request (the same in every call of one request) */
int lyra_hip_twin_assemble(lyra_hip_ctx* ctx, const lyra_hip_twin_slice* slices, int B, int out_samples);
/* NoiseEstimator::ReceiveSamples (lyra_decoder.cc:304-311) on the completed received hops marked by the preceding
 * lyra_hip_twin_assemble; stream_ids[r] = the stream whose slice carried noise_row r */
int lyra_hip_twin_noise(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B);
/* Ends the request: rows 0..num_streams-1 of the output, resampled from 16 kHz to out_rate when they differ (the
 * decoder-side resampler of streams 0..num_streams-1; any length, buffered_resampler.cc:120-128), into
 * out [num_streams][num_internal_samples * out_rate / 16000]; synchronises.  num_internal_samples == 0 just synchronises. */
int lyra_hip_twin_fetch(lyra_hip_ctx* ctx, int num_streams, int num_internal_samples, int out_rate, int16_t* out);

/* ---- Pipelined host-buffer calls (round 6) ----------------------------------------------------------------------------
 * Two-deep pipelined forms of the blocking host-buffer calls above, for a caller that feeds hop after hop: begin() stages
 * the caller's data in pinned memory, uploads it on a copy stream of its own, enqueues the kernels and the download and
 * returns; end() waits for the OLDEST begun call and copies its result out.  At most two calls may be in flight; with
 * begin(n + 1) issued before end(n) the upload of hop n + 1 and the download of hop n run under the kernels (blocking
 * calls idle the chain for both: BatchLyraEncoder + BatchLyraDecoder on two host threads 7.8-8.3 M frames/s, pipelined:
 * DESIGN.md 5).  The caller's buffers may be reused as soon as begin() / end() returns.  Results are bit-identical to the
 * blocking calls.  Do not mix blocking and pipelined calls of one kind on one context while calls are in flight.
 *
 * lyra_hip_encode_begin: LyraEncoder::Encode for B streams (lyra/lyra_encoder.cc:113-156) -- pcm [B][sample_rate_hz / 50]
 *   at 8 / 16 / 32 / 48 kHz (the encoder's own resampler runs on the device, :119-122), dtx != 0 as lyra_hip_encode_dtx
 *   (call lyra_hip_set_encoder_sample_rate(sample_rate_hz) first).
 * lyra_hip_encode_end: packets [B][num_bits / 8 rounded up]; packet_bytes [B] may be NULL (required to tell DTX's empty
 *   packets apart). */
int lyra_hip_encode_begin(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const int16_t* pcm, int sample_rate_hz,
                          int num_bits, int dtx);
int lyra_hip_encode_end(lyra_hip_ctx* ctx, uint8_t* packets, int32_t* packet_bytes);
/* lyra_hip_decode in two halves: begin() takes packets [B][num_bits / 8 rounded up], end() delivers pcm [B][320] of the
 * oldest begun call (its download runs under a younger call's kernels). */
int lyra_hip_decode_begin(lyra_hip_ctx* ctx, const int32_t* stream_ids, int B, const uint8_t* packets, int num_bits);
int lyra_hip_decode_end(lyra_hip_ctx* ctx, int16_t* pcm);
/* lyra_hip_twin_fetch in two halves ("Decoder twin" above): begin() ends the request being assembled -- its resampling and
 * its download are enqueued, the NEXT request's twin calls may follow at once --, end() waits for the oldest begun request
 * and copies its rows out (out may be NULL when that request had num_internal_samples == 0). */
int lyra_hip_twin_fetch_begin(lyra_hip_ctx* ctx, int num_streams, int num_internal_samples, int out_rate);
int lyra_hip_twin_fetch_end(lyra_hip_ctx* ctx, int16_t* out);

/* The context's FOUR HIP streams (hipStream_t as void*), for event timing / ordering by the caller: encode side, decode
 * side, the quantizer stream of lyra_hip_encode_dev / lyra_hip_encode_dtx_dev, and the noise stream.
 *  - The packets of the two encode calls are written on the QUANTIZER stream: lyra_hip_stream() does not cover them (it
 *    covers every other encode-side output).
 *  - The decoder-side lyra_hip_noise_receive_dev (d_is_noise and the estimator's state) and, inside
 *    lyra_hip_run_steps_dev at an external rate, the output resampler (d_ext_out) complete on the NOISE stream: an event
 *    recorded on lyra_hip_stream_decode() after those calls does NOT cover them -- use lyra_hip_stream_noise(),
 *    lyra_hip_stream_wait() or lyra_hip_synchronize().
 * lyra_hip_synchronize() waits for all four; lyra_hip_stream_wait() orders a caller's stream behind all four. */
void* lyra_hip_stream(lyra_hip_ctx* ctx);
void* lyra_hip_stream_decode(lyra_hip_ctx* ctx);
void* lyra_hip_stream_quantizer(lyra_hip_ctx* ctx);
void* lyra_hip_stream_noise(lyra_hip_ctx* ctx);
int lyra_hip_synchronize(lyra_hip_ctx* ctx);
/* Ordering against a caller-owned HIP stream (hipStream_t as void*, NULL = the null stream); see "Streams". */
int lyra_hip_wait_for_stream(lyra_hip_ctx* ctx, void* caller_stream);
int lyra_hip_stream_wait(lyra_hip_ctx* ctx, void* caller_stream);
/* on != 0: encode-side calls also wait for the MOST RECENT decode-side call, i.e. the library streams run
 * strictly in call order (one buffer set suffices; per-kernel timings are free of cross-stream contention). */
int lyra_hip_set_serial(lyra_hip_ctx* ctx, int on);
/* Stream priorities of the encode-side / decode-side / quantizer streams, 0 (lowest) .. 2 (highest); default 0, 0, 2.  The
 * context is drained and the three streams are created anew (HIP fixes a priority at creation).  A context that serves
 * BLOCKING decode calls beside another context's encoder (BatchLyraDecoder next to BatchLyraEncoder) wants its decode side
 * first -- (0, 2, 2), the schedule of rounds 2-3: the decode kernels win the arbitration and the call returns sooner
 * (7.2 M vs 6.4 M frames/s on two host threads); the `_dev` pipeline of one context wants the default (see
 * LYRA_HIP_PRIO under lyra_hip_create).  CU-masked streams (contexts of <= 1024 streams) have no priority.  Handles obtained
 * earlier from lyra_hip_stream() / lyra_hip_stream_decode() / lyra_hip_stream_quantizer() are invalid afterwards. */
int lyra_hip_set_stream_priorities(lyra_hip_ctx* ctx, int encode_side, int decode_side, int quantizer);

/* Per-stream state footprint in HBM (bytes) and the context's stream capacity. */
size_t lyra_hip_state_bytes_per_stream(void);
int lyra_hip_max_streams(const lyra_hip_ctx* ctx);

/* Measurement hook (bench.py): launches of every kernel whose bit is set in `kernel_mask` (bit i = kernel i of
 * lyra_hip_profile_kernel_name) are bracketed by HIP events recorded on the context's stream; 0 disables.
 * profile_read() synchronises, returns per-kernel total milliseconds and launch counts since the previous read
 * (arrays of lyra_hip_profile_kernel_count() entries) and clears them. */
int lyra_hip_profile_enable(lyra_hip_ctx* ctx, unsigned kernel_mask);
/* Bracket only every `every`-th launch of an enabled kernel (default 1): an event record is a packet of its own in the
 * stream (~5 us of bubble on MI355X), so timing every launch of a kernel inside a throughput measurement slows the
 * measured pipeline itself. */
int lyra_hip_profile_sample(lyra_hip_ctx* ctx, int every);
int lyra_hip_profile_kernel_count(void);
const char* lyra_hip_profile_kernel_name(int i);
int lyra_hip_profile_read(lyra_hip_ctx* ctx, double* total_ms, long* launches);
/* start / end (ms, relative to the first recorded span's start) of every span recorded since the last read; call
 * BEFORE lyra_hip_profile_read.  Returns the number of spans written (<= cap) or a negative error. */
int lyra_hip_profile_timeline(lyra_hip_ctx* ctx, int cap, int* kernel_ids, float* start_ms, float* end_ms);

/* Test hook: copies stage-boundary activations of the LAST extract/generate call (device scratch) to host.
 * which: 0 enc stage0 out [B][4][128], 1 enc stage1 out [B][2][256], 2 enc int8 codes [B][64] (as f32),
 *        3 dec head out [B][4][128], 4 dec stage1 out [B][20][64].  Channel order is the library's
 *        internal one; returns the number of floats written or a negative error. */
long lyra_hip_debug_read(lyra_hip_ctx* ctx, int which, float* host_out, long capacity);

#ifdef __cplusplus
}
#endif
#endif /* LYRA_HIP_H_ */
