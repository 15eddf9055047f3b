#!/usr/bin/env python3
"""bench.py -- throughput of the MI355X-native Lyra encode/decode hot path.

    python bench.py [--gpus N] [--steps K] [--warmup W] [--config {2,3,4,5}] [--full-decoder] [--dtx] [--rate HZ]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
        bench.py --gpus N --steps K --warmup W

Workloads (BASELINE.json `configs`; a "step" advances every stream by one 20 ms frame, state carried
from step to step -- frames of a stream are NOT independent):

  --config 3 (default; the configuration BASELINE.json's metric is quoted on)
        4096 streams per GPU, 9200 bps (184 bits = 46 RVQ stages), step = lyra_hip_encode_dev (PCM -> packets)
        + lyra_hip_decode_dev (packets -> PCM).  Weak scaling: 4096 streams per GPU.
  --config 2   1024 streams per GPU, 3200 bps (64 bits = 16 stages), encode + decode.
  --config 4   decode only, 8192 streams per GPU: lossy features / packets are produced ONCE by the encoder + RVQ on
        the same synthetic PCM (lyra_benchmark_lib.cc:121-160); the timed step is lyra_hip_generate_dev (features ->
        PCM, the pure lyra_gan_model path, primary = `value`) and, in a second timed region, lyra_hip_decode_dev
        (packets -> PCM, adds the RVQ decode; reported as `secondary`).
  --config 5   32768 streams IN TOTAL at 6000 bps (120 bits), encode + decode, split over the GPUs of the job
        (strong scaling: N = 1 runs all 32768 streams on one GPU).  Same as --total-streams 32768 --bits 120.
  --streams / --bits / --total-streams override the above.

Legs around the bare codec step (what LyraEncoder::Encode / LyraDecoder::DecodeSamples run per hop):
  --full-decoder   NoiseEstimator::ReceiveSamples (log-mel + recurrence, one kernel on a stream of its own) on every
                   decoded hop (lyra_decoder.cc:304-311)
  --dtx            encode with enable_dtx: encoder-side noise estimator, masked stream ids, empty packets
                   (lyra_encoder.cc:131-156)
  --rate HZ        8000 / 32000 / 48000: the encoder's and the decoder's resampler around the codec
                   (lyra_encoder.cc:119-122, lyra_decoder.cc:107-113); the PCM ring and the output are at HZ
  --with-logmel    (round-2 leg) the plain log-mel extractor behind every decoded hop, driven call by call from Python

The timed region is ONE C call per rank (lyra_hip_run_steps_dev: K hops enqueued by the library, no host language in
the loop); --per-call drives the individual `_dev` entry points from Python instead (the round-2 harness).

Streams are sharded across GPUs with no data-path collective; the only collectives are the timing barrier and the
max / sum reductions of the result (plus, with --bcast-weights, one RCCL broadcast of the packed weights at init).
`python bench.py --gpus N` (N > 1) without a launcher re-executes itself under torch.distributed.run with N ranks, one
per GPU, RCCL process group; --single-process keeps everything in one process (one host thread and one context per GPU,
no process group) for boxes where spawning is not possible.

One JSON line on rank 0:
  * `value` = whole-job frames/s with inputs resident in HBM;
  * `roofline` = the WHOLE STEP against its binding bound (fp32 MFMA for encode+decode: 5.43 MFLOP of fp32 MFMA work
    per stream-frame = 34.5 ns at 157.3 TFLOP/s, vs 13.4 ns of HBM time for the bytes the kernels move);
  * `kernels` = per-kernel durations measured live with HIP events on the library's own streams in a SERIALISED pass
    (lyra_hip_set_serial: the library streams strictly in call order, so no cross-stream contention), each with
    its own binding bound computed from the bytes it actually moves (lyra_amd/csrc/state_layout.h);
  * `dominant_kernel` = the kernel with the largest serialised duration, bracketed by HIP events inside the timed
    region (i.e. under the overlap of the library streams the step really runs with);
  * `step_latency_us` = distribution of ONE isolated step (enqueue -> all outputs complete, nothing else in flight):
    min / mean / p50 / p99 / max / stddev over --latency-steps steps (lyra_benchmark_lib.cc:164-182 prints the same
    statistics per stage; a 20 ms-deadline codec cares about the tail, not the mean);
  * `cpu_baseline` = the CPU oracle (a port, not the TFLite binary) on this box's host cores, bounded sample;
    `cpu_baseline_xnnpack` = the same graphs with every arithmetic operator run by XNNPACK operator objects (the engine the
    reference's TfLiteModelWrapper uses, as far as this image holds it: torch's XNNPACK), same cores, same sample shape.
    Both run on rank 0 after the process group is gone.
"""
import argparse
import json
import os
import socket
import subprocess
import sys
import threading
import time

import numpy as np

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

SEED = 0x4C797261  # "Lyra"
CONFIGS = {
    2: dict(streams=1024, bits=64, mode="encdec", scaling="weak"),
    3: dict(streams=4096, bits=184, mode="encdec", scaling="weak"),
    4: dict(streams=8192, bits=184, mode="decode", scaling="weak", sub_batches=2),   # one side only: two sub-batches
    5: dict(total_streams=32768, bits=120, mode="encdec", scaling="strong"),
}

# Work per stream-frame of each kernel (DESIGN.md "Kernels"):
#   f32 / i8 : MACs on the fp32 / int8 MFMA path;  flops: fp32 VALU FLOP (RVQ, log-mel);
#   moved    : HBM bytes the kernel itself reads + writes per stream-frame: inputs + outputs + per-stream state at
#              the representation of lyra_amd/csrc/state_layout.h (int8 histories as int8, dilated-conv histories
#              as rings that are read where tapped and written T rows per step);
#   ref_bytes: the same at the reference's fp32 TFLite-variable representation (215 KB/frame whole path).
KERNEL_WORK = {
    "enc_s0_kernel": dict(f32=912640, i8=0, moved=640 + 2 * (48 + 26 * 64 + 5 * 64) * 4 + 4 * 128 * 4,
                          ref_bytes=640 + 2 * (48 + 26 * 64 + 5 * 64) * 4 + 4 * 128 * 4),
    "enc_s1_kernel": dict(f32=430592, i8=0,
                          moved=2048 + (1024 + 1024) + (3072 + 2048) + (4096 + 2048) + (1024 + 1024) + 2048,
                          ref_bytes=4 * 128 * 4 + 2 * (26 * 128 + 2 * 128) * 4 + 2 * 256 * 4),
    "enc_s2_kernel": dict(f32=132608, i8=519168,
                          moved=2048 + (2048 + 2048) + (1024 + 512) + (1024 + 512) + (512 + 512) + (1024 + 512) + 512 + 8,
                          ref_bytes=2 * 256 * 4 + 2 * (26 * 256 + 2 * 256 + 2 * 512) * 4 + 256),
    # ALGORITHMIC work = the graph's: 3 fp32 ops per (codeword, dim) term + the residual update.  The kernel certifies most
    # indices from 46 x 16 x 64 MFMA MACs per frame (1.7 % of the step's matrix work) and runs the graph's chain only for the
    # frames whose margin test fails (csrc/misc_kernels.hip rvq_encode_kernel); its floor stays priced on the graph's work.
    "rvq_encode_kernel": dict(f32=0, i8=0, moved=256 + 23, ref_bytes=256 + 23, flops=3 * 16 * 64 * 46 + 3 * 64 * 46),
    "rvq_decode_kernel": dict(f32=0, i8=0, moved=23 + 256, ref_bytes=23 + 256, flops=64 * 46),
    "dec_s0_kernel": dict(f32=24576, i8=758272,
                          moved=23 + (512 + 256) + (2048 + 2048) + (512 + 512) + (1024 + 512) + (1024 + 512)
                          + (1024 + 1024) + 2048 + 4,
                          ref_bytes=256 + 2 * (2 * 64 + 8 * 64 + 26 * 256 + 4 * 64) * 4 + 4 * 128 * 4),
    "dec_s1_kernel": dict(f32=627200, i8=0,
                          moved=2048 + (1024 + 1024) + (3072 + 2048) + (4096 + 2048) + (1280 + 1280) + 5120,
                          ref_bytes=4 * 128 * 4 + 2 * (26 * 128 + 5 * 64) * 4 + 20 * 64 * 4),
    "dec_s2_kernel": dict(f32=584960, i8=0, moved=5120 + 2 * (26 * 64) * 4 + 2 * 48 * 4 + 640 + 8,
                          ref_bytes=20 * 64 * 4 + 2 * (26 * 64 + 48) * 4 + 640),
    # fp64 per frame: half of a complex radix-4 FFT-1024 (5 passes x 256 butterflies x 34 flops, two real frames per
    # transform), the split by conjugate symmetry + |X| (513 bins x ~8), the sparse mel weights (2 x 1026 MAC)
    "logmel_kernel": dict(f32=0, i8=0, moved=640 + 640 + 640 + 640, ref_bytes=640 + 2 * 640 + 640,
                          flops64=5 * 256 * 34 // 2 + 513 * 8 + 2 * 2 * 1026),
    # NoiseEstimator::ReceiveSamples in one launch: the log-mel front end + the recurrence over 5 x 160 floats of state
    "logmel_noise_kernel": dict(f32=0, i8=0, moved=640 + 640 + 640 + 2 * 5 * 640 + 16, ref_bytes=640 + 2 * 640 + 2 * 5 * 640,
                                flops64=5 * 256 * 34 // 2 + 513 * 8 + 2 * 2 * 1026, flops=160 * 30),
    # polyphase FIR: 48 kHz <-> 16 kHz, 960 + 320 samples of int16 in/out + the tap history; ~24 taps per output
    "resample_kernel": dict(f32=0, i8=0, moved=(960 + 320) * 2 + 2 * 256, ref_bytes=(960 + 320) * 2 + 2 * 256, flops=2 * 24 * 640),
}
ENCODE_KERNELS = ("enc_s0_kernel", "enc_s1_kernel", "enc_s2_kernel", "rvq_encode_kernel")
DECODE_KERNELS = ("dec_s0_kernel", "dec_s1_kernel", "dec_s2_kernel")
PEAK_F32_MFMA_TFLOPS = 157.3   # /opt/skills/guides/MI355X_MICROARCH.md: v_mfma_f32_16x16x4_f32, dense
PEAK_I8_MFMA_TOPS = 3944.0      # same guide: v_mfma_i32_16x16x64_i8, dense
PEAK_F32_VALU_TFLOPS = 157.3    # same guide: FP32 vector, counted as FMA (2 flops per lane-instruction)
PEAK_F32_VALU_TOPS = 78.65      # the same issue rate for non-FMA vector work (sub / mul / add, one flop each): what the
                                # quantizer's distance search consists of (profiles/history/r03_valu_probe.txt: v_pk_*_f32 do
                                # not double it on this part)
PEAK_F64_VALU_TFLOPS = 78.6     # CDNA4 FP64 vector (half the fp32 rate)
PEAK_HBM_GBS = 8000.0


def parse(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=1000)   # 0.3 s of GPU time at B = 4096
    ap.add_argument("--warmup", type=int, default=30)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS))
    ap.add_argument("--streams", type=int, default=None, help="streams per GPU (weak scaling)")
    ap.add_argument("--total-streams", type=int, default=None, help="streams in the whole job (strong scaling)")
    ap.add_argument("--bits", type=int, default=None)
    ap.add_argument("--sub-batches", type=int, default=None,
                    help="split every call into this many sub-batches on stream sets of their own (LYRA_HIP_SUBBATCHES); "
                         "default: the config's (config #2: 4, #4: 2, others 1)")
    ap.add_argument("--with-logmel", action="store_true",
                    help="run the plain log-mel extractor on every decoded hop inside the step (implies --per-call)")
    ap.add_argument("--full-decoder", action="store_true",
                    help="NoiseEstimator::ReceiveSamples on every decoded hop (lyra_decoder.cc:304-311)")
    ap.add_argument("--dtx", action="store_true", help="encode with enable_dtx (lyra_encoder.cc:131-156)")
    ap.add_argument("--rate", type=int, default=16000, choices=[8000, 16000, 32000, 48000],
                    help="external sample rate: both resamplers around the codec when it is not 16000")
    ap.add_argument("--per-call", action="store_true",
                    help="drive the individual `_dev` entry points from Python instead of lyra_hip_run_steps_dev")
    ap.add_argument("--latency-steps", type=int, default=200, help="isolated steps for the latency distribution (0: skip)")
    ap.add_argument("--ramp-steps", type=int, default=64,
                    help="extra UNTIMED steps enqueued right before the timed region so that it starts on a chip at its "
                         "loaded clocks: after >= 20 ms of idling a 20-step region runs 14 %% slower "
                         "(profiles/history/r03_idle_gap_probe.txt); 0: none")
    ap.add_argument("--single-process", action="store_true",
                    help="--gpus N in ONE process: a host thread and a context per GPU, no process group")
    ap.add_argument("--bcast-weights", action="store_true",
                    help="rank 0 reads the weight container, RCCL-broadcasts it, every rank builds from the image")
    ap.add_argument("--requant", default="xnnpack", choices=["xnnpack", "exact", "gemmlowp_double", "builtin_mixed"],
                    help="arithmetic of the graphs' int8 regions: xnnpack = what the reference runs (default); the others "
                         "are TFLite's builtin kernels (builtin_mixed: per operator, include/lyra_hip.h)")
    ap.add_argument("--no-verify", action="store_true",
                    help="skip the self-check: after all timing, the packets and the PCM of the run's LAST TWO steps are "
                         "compared, for the first --verify-streams streams of every rank, with the CPU oracle replaying "
                         "every step the context has executed since its reset (state is carried, so the end state "
                         "vouches for the whole run).  Outside every timed region.")
    ap.add_argument("--verify-streams", type=int, default=32)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-cpu-confine", action="store_true",
                    help="at --gpus N > 1 do not confine each rank to its share of the CPU quota (confine_rank_to_cpu_share)")
    ap.add_argument("--no-kernel-table", action="store_true")
    ap.add_argument("--per-call-split", action="store_true",
                    help="with --per-call at an external rate: every codec call on its own (resample, encode, decode, estimator, "
                         "resample) instead of lyra_hip_encode_ext_dev / lyra_hip_decode_ext_dev, one call per side and hop")
    ap.add_argument("--oversubscribe-device", type=int, default=None, metavar="D",
                    help="HARDWARE SMOKE of the N-rank path on a box with fewer GPUs than ranks: every rank drives GPU D, the "
                         "process group is gloo (RCCL refuses two ranks on one device).  Everything else is the N-rank job: one "
                         "process and one context per rank, CPU shares, barriers, reductions, rank gather.  NOT a scaling "
                         "measurement -- the line says so")
    ap.add_argument("--force-dist", action="store_true",
                    help="bring the process group up (RCCL on a GPU box) even with ONE rank: the N > 1 code path -- "
                         "init, barriers around the timed region, max / sum all-reduce, rank gather -- on a 1-GPU box")
    ap.add_argument("--selftest-dist", action="store_true",
                    help="CPU-only: exercise the multi-rank plumbing (gloo) without touching a GPU")
    ap.add_argument("--stub-context", action="store_true",
                    help="CPU-only test hook: run main() itself -- process group (gloo), sharding, weight broadcast, "
                         "barriers, reductions, result line -- with a stand-in for the GPU shard that only counts its calls")
    return ap.parse_args(argv)


def resolve_workload(args, world):
    """-> dict(B per rank, total, bits, mode, scaling) from --config and the overrides."""
    cfg = dict(CONFIGS[args.config])
    if args.bits is not None:
        cfg["bits"] = args.bits
    if args.total_streams is not None:
        cfg.pop("streams", None)
        cfg["total_streams"] = args.total_streams
        cfg["scaling"] = "strong"
    elif args.streams is not None:
        cfg.pop("total_streams", None)
        cfg["streams"] = args.streams
        cfg["scaling"] = "weak"
    if "total_streams" in cfg:
        total = cfg["total_streams"]
        if total % world:
            raise SystemExit(f"--total-streams {total} is not divisible by the {world} GPUs of the job")
        per = total // world
    else:
        per = cfg["streams"]
        total = per * world
    return dict(B=per, total=total, bits=cfg["bits"], mode=cfg["mode"], scaling=cfg["scaling"], config=args.config,
                sub_batches=args.sub_batches if args.sub_batches is not None else cfg.get("sub_batches"))


def dist_env():
    rank = int(os.environ.get("RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    return rank, world, local


def pg_active():
    """A process group is up: world > 1, or world == 1 under --force-dist (the RCCL path exercised with one rank)."""
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


def reduce_job(local_seconds, local_units, world, backend_device):
    """max-over-ranks time, sum-over-ranks units (the only collectives of the whole job)."""
    if world == 1 and not pg_active():
        return local_seconds, local_units
    import torch
    import torch.distributed as dist
    t = torch.tensor([local_seconds], dtype=torch.float64, device=backend_device)
    u = torch.tensor([float(local_units)], dtype=torch.float64, device=backend_device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(t.item()), int(u.item())


def shard_ids(total_streams, rank, world):
    """Contiguous block of stream ids per rank.  State never leaves its GPU."""
    per = total_streams // world
    return rank * per, per


def usable_cores():
    """Host cores this process may actually use: affinity mask capped by the cgroup CPU quota."""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    try:
        quota, period = open("/sys/fs/cgroup/cpu.max").read().split()[:2]
        if quota != "max":
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        pass
    return max(1, n)


def cpu_baseline(bits, mode, requant="xnnpack", engine="scalar"):
    """The CPU leg on the host cores, bounded to ~12 s.  engine "scalar": oracle/lyra_oracle.c, the CPU port of the same
    arithmetic (`cpu_baseline`, kind "port").  engine "xnnpack": the same graph plumbing with every arithmetic operator an
    XNNPACK operator object, one stream per thread -- what TfLiteModelWrapper with use_xnn = true runs per Invoke()
    (tflite_model_wrapper.cc:63-85, lyra_benchmark_lib.cc:85-160), on the XNNPACK torch's libtorch_cpu.so carries in this
    image (`cpu_baseline_xnnpack`; bit-equal to the port on features / packets, tests/test_xnnpack_engine.py)."""
    from oracle import lyra_oracle
    lyra_oracle.build()
    o = lyra_oracle.Oracle(mode=requant, engine=engine)
    cores = usable_cores()
    rng = np.random.Generator(np.random.PCG64(SEED))
    streams = cores * 2

    def run(steps):
        pcm = rng.integers(-32768, 32768, size=(steps, streams, 320)).astype(np.int16)
        r = lyra_oracle.run_batch(o, pcm, bits // 4, do_decode=True, threads=cores)
        return streams * steps / r["seconds"], r
    rate, _ = run(8)
    steps = int(max(16, min(4000, 12.0 * rate / streams)))
    rate, r = run(steps)
    split = r["stage_seconds"] / (streams * steps) * 1e3
    what = ("oracle/lyra_oracle.c one stream per thread" if engine == "scalar" else
            "oracle/lyra_oracle.c -DLO_XNNPACK: XNNPACK operators (create once, setup + run per frame), one stream per thread")
    out = {"value": round(rate, 1), "unit": "frames/s", "cores": cores,
           "kind": "port" if engine == "scalar" else "xnnpack-ops (torch's build, not TF 2.11's)",
           "sample": f"{streams} streams x {steps} frames, uniform full-scale int16 PCM, {bits} bits, encode+decode, {what}",
           "ms_per_frame_per_core": {"extract": round(float(split[0]), 4), "quantize": round(float(split[1]), 4),
                                     "dequantize": round(float(split[2]), 4), "generate": round(float(split[3]), 4)}}
    if mode == "decode":   # the decode-only share of the same run (dequantize + generate stages)
        dec_ms = float(split[2] + split[3])
        out["decode_only_value"] = round(cores / (dec_ms * 1e-3), 1) if dec_ms > 0 else None
    return out


def confine_rank_to_cpu_share(local, world):
    """Under --gpus N every rank confines itself to its share of the CPUs the container's quota is worth (cgroup cpu.max;
    lyra_amd/host/plugin_mt_demo.cc LimitAffinityToCpuQuota has the reasoning: CFS hands a quota out in per-CPU slices, and
    threads spread over many more CPUs than the quota is worth get throttled long before the quota is used).  At least two
    CPUs per rank (the enqueueing thread + the runtime's helpers).  Returns (original mask, this rank's CPUs) or None."""
    if not hasattr(os, "sched_getaffinity"):
        return None
    allowed = sorted(os.sched_getaffinity(0))
    per = max(2, usable_cores() // max(1, world))
    if per * world > len(allowed):
        return None
    mine = allowed[local * per:(local + 1) * per]
    try:
        os.sched_setaffinity(0, mine)
    except OSError:
        return None
    return allowed, mine


# ------------------------------------------------------------------------------------------------------------
# per-kernel and whole-step roofline arithmetic (pure functions: covered by the CPU tests)
# ------------------------------------------------------------------------------------------------------------
def kernel_bound(name):
    """(bound name, ns per stream-frame at that peak, dict of all candidate times in ns)."""
    w = KERNEL_WORK[name]
    t = {"mfma": (2 * w["f32"] / (PEAK_F32_MFMA_TFLOPS * 1e12) + 2 * w["i8"] / (PEAK_I8_MFMA_TOPS * 1e12)) * 1e9,
         "hbm": w["moved"] / (PEAK_HBM_GBS * 1e9) * 1e9,
         "valu": (w.get("flops", 0) / (PEAK_F32_VALU_TOPS * 1e12) + w.get("flops64", 0) / (PEAK_F64_VALU_TFLOPS * 1e12)) * 1e9}
    b = max(t, key=lambda k: t[k])
    return b, t[b], t


def kernel_row(name, total_ms, launches, B):
    w = KERNEL_WORK[name]
    dur = total_ms / launches * 1e-3
    bound, t_ns, _ = kernel_bound(name)
    return {"avg_us": round(dur * 1e6, 2), "launches": launches,
            "f32_tflops": round(2 * w["f32"] * B / dur / 1e12, 3),
            "i8_tops": round(2 * w["i8"] * B / dur / 1e12, 3),
            "moved_gbs": round(w["moved"] * B / dur / 1e9, 1),
            "moved_bytes_per_launch": w["moved"] * B,
            "bound": bound, "floor_us": round(t_ns * B * 1e-3, 2),
            "frac": round(t_ns * 1e-9 * B / dur, 4)}


def step_work(mode, legs):
    """legs: dict(with_logmel, full_decoder, dtx, rate) -> kernels of one step and their summed work."""
    names = list(DECODE_KERNELS) if mode == "decode" else list(ENCODE_KERNELS + DECODE_KERNELS)
    if legs.get("with_logmel"):
        names.append("logmel_kernel")
    if legs.get("full_decoder"):
        names.append("logmel_noise_kernel")
    if legs.get("dtx") and mode != "decode":
        names.append("logmel_noise_kernel")
    if legs.get("rate", 16000) != 16000:
        names += ["resample_kernel"] * (1 if mode == "decode" else 2)
    f32 = sum(KERNEL_WORK[k]["f32"] for k in names)
    i8 = sum(KERNEL_WORK[k]["i8"] for k in names)
    moved = sum(KERNEL_WORK[k]["moved"] for k in names)
    ref_bytes = sum(KERNEL_WORK[k]["ref_bytes"] for k in names)
    return names, f32, i8, moved, ref_bytes


def legs_of(args):
    return dict(with_logmel=args.with_logmel, full_decoder=args.full_decoder, dtx=args.dtx, rate=args.rate)


def step_roofline(mode, legs, frames_per_s_per_gpu, traffic_table):
    """The whole step against its binding bound."""
    names, f32, i8, moved, ref_bytes = step_work(mode, legs)
    t_mfma = 2 * f32 / (PEAK_F32_MFMA_TFLOPS * 1e12) + 2 * i8 / (PEAK_I8_MFMA_TOPS * 1e12)
    t_hbm = moved / (PEAK_HBM_GBS * 1e9)
    traffic = None
    if traffic_table:
        def per_frame(k):   # the int8 stages' kernels carry the arithmetic mode in their symbol (enc_s2_xn_kernel, dec_s0_dr_kernel)
            for sym in (k, k.replace("_kernel", "_xn_kernel"), k.replace("_kernel", "_dr_kernel")):
                if sym in traffic_table and isinstance(traffic_table[sym], dict):
                    return traffic_table[sym].get("hbm_bytes_per_frame")
            return None
        vals = [per_frame(k) for k in names]
        if all(v is not None for v in vals):
            traffic = round(sum(vals), 1)
    common = {"kernel": "whole step: " + " + ".join(n.replace("_kernel", "") for n in names),
              "fp32_mfma_ns_per_frame": round(2 * f32 / (PEAK_F32_MFMA_TFLOPS * 1e12) * 1e9, 2),
              "int8_mfma_ns_per_frame": round(2 * i8 / (PEAK_I8_MFMA_TOPS * 1e12) * 1e9, 2),
              "hbm_ns_per_frame": round(t_hbm * 1e9, 2),
              "algorithmic_flops_per_frame": 2 * f32, "moved_bytes_per_frame": moved,
              "reference_representation_bytes_per_frame": ref_bytes,
              "traffic": traffic, "traffic_unit": "HBM bytes per stream-frame",
              "traffic_source": "offline: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes (profiles/traffic.json), "
                                "2*FETCH_SIZE + WRITE_SIZE per the gfx950 note in MI355X_MICROARCH.md"}
    issue = load_issue_time(names)
    if issue:
        common["issue_time"] = issue
    if t_mfma >= t_hbm:
        ach = 2 * f32 * frames_per_s_per_gpu / 1e12
        return dict(common, bound="mfma", achieved=round(ach, 3), peak=PEAK_F32_MFMA_TFLOPS, unit="TFLOP/s",
                    frac=round(ach / PEAK_F32_MFMA_TFLOPS, 4),
                    hbm_frac=round(moved * frames_per_s_per_gpu / 1e9 / PEAK_HBM_GBS, 4))
    ach = moved * frames_per_s_per_gpu / 1e9
    return dict(common, bound="hbm", achieved=round(ach, 1), peak=PEAK_HBM_GBS, unit="GB/s",
                frac=round(ach / PEAK_HBM_GBS, 4),
                mfma_frac=round(2 * f32 * frames_per_s_per_gpu / 1e12 / PEAK_F32_MFMA_TFLOPS, 4))


def load_issue_time(names):
    """Offline SQ counters (profiles/r06_pmc_sq.txt, B = 4096): per SIMD, the time the matrix pipe is busy and the time the
    vector instructions take to issue, summed over the step's kernels.  On this part vector instructions do not hide under
    the MFMAs (mfma_valu_overlap_probe.hip (a probe of an earlier round, removed since: git history)), so the two add up; informational, not the roofline."""
    try:
        cur, tab = None, {}
        for line in open(os.path.join(ROOT, "profiles", "r06_pmc_sq.txt")):
            if line.strip() and not line.startswith(" "):
                cur = line.strip()
                tab[cur] = {}
            else:
                p = line.split()
                if cur and len(p) == 2:
                    try:
                        tab[cur][p[0]] = float(p[1])
                    except ValueError:
                        pass
        simds, clk = 1024.0, 2.06e9     # 256 CUs x 4; the clock the chip sustains under this load
        matrix = vector = 0.0
        for k in names:
            c = tab.get(k) or tab.get(k.replace("_kernel", "_xn_kernel")) or tab[k.replace("_kernel", "_dr_kernel")]
            matrix += c["SQ_VALU_MFMA_BUSY_CYCLES"] / simds / clk
            per = 2.6 if k == "rvq_encode_kernel" else 4.3   # cycles per wave instruction: fp32 chains | integer / mixed
            vector += (c["SQ_INSTS_VALU"] - c.get("SQ_INSTS_MFMA", 0.0)) * per / simds / clk
        return {"matrix_pipe_us_per_step_at_B4096": round(matrix * 1e6, 1), "vector_issue_us_per_step_at_B4096": round(vector * 1e6, 1),
                "source": "offline: SQ counters of profiles/r06_pmc_sq.txt at 2.06 GHz; vector instructions do not hide under "
                          "MFMAs on gfx950 (profiles/history/r03_mfma_valu_overlap_probe.txt), so a SIMD's time is at most the sum (round 6: vector issue overlaps the matrix pipe in part)"}
    except Exception:
        return None


def load_traffic():
    """profiles/traffic.json: per kernel HBM bytes per launch at the batch it was profiled with (4096)."""
    try:
        t = json.load(open(os.path.join(ROOT, "profiles", "traffic.json")))
        batch = float(t.get("_batch", 4096))
        for k, v in t.items():
            if isinstance(v, dict) and "hbm_bytes_per_launch" in v:
                v["hbm_bytes_per_frame"] = v["hbm_bytes_per_launch"] / batch
        return t
    except Exception:
        return None


# ------------------------------------------------------------------------------------------------------------
# one GPU's share of the job
# ------------------------------------------------------------------------------------------------------------
class Shard:
    """All device buffers and the step function of ONE GPU (one rank, or one thread of the single-process mode)."""

    RING = 32   # distinct synthetic input frames kept in HBM and cycled (bounds memory at B = 32768)

    def __init__(self, device, first_id, wl, args, weights_image=None):
        import torch
        import lyra_amd
        self.torch, self.lyra = torch, lyra_amd
        self.dev = torch.device("cuda", device)
        self.wl, self.args = wl, args
        B, bits = wl["B"], wl["bits"]
        self.ctx = lyra_amd.LyraHip(device=device, max_streams=B, requant=args.requant, weights_image=weights_image,
                                    sub_batches=wl.get("sub_batches"))
        self.history = []   # every (kind, first step, n) this context has executed since its last reset: --verify replays it
        self.ctx.torch_order = False   # this harness synchronises explicitly around every region it times
        if args.dtx:   # a DTX LyraEncoder at --rate hands that rate to its NoiseEstimator (lyra_encoder.cc:82-85)
            self.ctx.set_encoder_sample_rate(args.rate)
        gen = torch.Generator(device=self.dev)
        gen.manual_seed(SEED + first_id)
        n = min(self.RING, args.warmup + args.steps)
        self.rate = args.rate
        self.n_ext = 320 * args.rate // 16000
        # UnitToInt16Scalar(U(-1,1)) i.i.d. (lyra_benchmark_lib.cc:233-239): full-scale uniform int16, at the external rate
        self.pcm_in = torch.randint(-32768, 32768, (n, B, self.n_ext), generator=gen, device=self.dev,
                                    dtype=torch.int32).to(torch.int16)
        if args.dtx:   # every fourth stream idles (digital silence) so that the DTX branch is actually taken
            self.pcm_in[:, 3::4] = 0
        self.ids = torch.arange(B, device=self.dev, dtype=torch.int32)  # local stream slots of this shard
        nb = lyra_amd.packet_size(bits)
        # two packet / PCM buffer sets, alternated (include/lyra_hip.h "Streams": the two-buffer rule)
        self.packets = [torch.zeros((B, nb), device=self.dev, dtype=torch.uint8) for _ in range(2)]
        self.pcm_out = [torch.empty((B, 320), device=self.dev, dtype=torch.int16) for _ in range(2)]
        self.packet_bytes = [torch.zeros((B,), device=self.dev, dtype=torch.int32) for _ in range(2)] if args.dtx else None
        self.ext_out = [torch.empty((B, self.n_ext), device=self.dev, dtype=torch.int16) for _ in range(2)] \
            if args.rate != 16000 else None
        self.in16 = torch.empty((B, 320), device=self.dev, dtype=torch.int16) if args.rate != 16000 else None
        self.is_noise = torch.zeros((B,), device=self.dev, dtype=torch.int32) if args.full_decoder else None
        self.mel = torch.empty((B, 160), device=self.dev, dtype=torch.float32) if args.with_logmel else None
        self.per_call = args.per_call or args.with_logmel
        self.feats = self.pk_seq = None
        if wl["mode"] == "decode":
            self._prepare_decode_inputs(n)
        torch.cuda.synchronize(self.dev)

    def _prepare_decode_inputs(self, n):
        """Config #4: Extract -> Quantize -> DecodeToLossyFeatures once over the synthetic PCM
        (lyra_benchmark_lib.cc:85-137); keep the lossy features and the packets of every step."""
        torch, ctx, B, bits = self.torch, self.ctx, self.wl["B"], self.wl["bits"]
        ns = bits // 4
        self.feats = torch.empty((n, B, 64), device=self.dev, dtype=torch.float32)
        self.pk_seq = torch.zeros((n, B, self.lyra.packet_size(bits)), device=self.dev, dtype=torch.uint8)
        feat = torch.empty((B, 64), device=self.dev, dtype=torch.float32)
        idx = torch.empty((B, 46), device=self.dev, dtype=torch.int32)
        for i in range(n):
            x = self.pcm_in[i]
            if self.rate != 16000:
                ctx.resample_dev(self.ids, x, self.rate, 16000, self.in16, side="encoder")
                x = self.in16
            ctx.extract_dev(self.ids, x, feat)
            ctx.rvq_encode_dev(feat, bits, idx)
            ctx.rvq_decode_dev(idx, self.feats[i])      # decode side: ordered after the encode side by the library
            ctx.synchronize()
            nib = torch.zeros((B, 2 * self.pk_seq.shape[2]), device=self.dev, dtype=torch.int32)
            nib[:, :ns] = idx[:, :ns]
            self.pk_seq[i] = ((nib[:, 0::2] << 4) | nib[:, 1::2]).to(torch.uint8)   # packet.h:91-122, no header
            torch.cuda.synchronize(self.dev)
        ctx.reset()
        self.history = []

    def verify(self, n_streams):
        """The run checks itself (lyra_benchmark_lib.cc:121-160 is the loop being replaced; it verifies nothing either, but
        a benchmark of a re-implementation should).  The CPU oracle replays, for the first n_streams streams of this
        shard, EVERY step this context has executed since its reset -- warm-up, serialised table, ramp, timed region(s),
        latency steps, with the same cycled input ring -- and the packets / PCM the GPU left in its two output buffers
        (the run's last two steps) must equal the oracle's, bit for bit.  Only the bare codec step is covered."""
        a = self.args
        if a.dtx or a.rate != 16000 or a.with_logmel:
            return {"verified": None, "why": "self-check covers the bare codec step (no --dtx / --rate / --with-logmel)"}
        from concurrent.futures import ThreadPoolExecutor
        from oracle import lyra_oracle
        lyra_oracle.build()
        t0 = time.perf_counter()
        self.sync()
        o = lyra_oracle.Oracle(mode=a.requant)
        S = min(n_streams, self.wl["B"])
        bits, ns = self.wl["bits"], self.wl["bits"] // 4
        order = [(k, i) for (k, f, n) in self.history for i in range(f, f + n)]
        if len(order) < 2:
            return {"verified": None, "why": "fewer than two steps executed"}
        got_pk = [self.packets[s][:S].cpu().numpy() for s in range(2)]
        got_pcm = [self.pcm_out[s][:S].cpu().numpy() for s in range(2)]
        last = order[-2:]
        ok = True
        if all(k == "encdec" for k, _ in order):
            ring = self.pcm_in[:, :S].cpu().numpy()
            pcm = np.stack([ring[i % ring.shape[0]] for _, i in order])
            r = lyra_oracle.run_batch(o, pcm, ns, do_decode=True, threads=usable_cores())
            for j, (_, i) in enumerate(last):
                ok = ok and np.array_equal(got_pk[i & 1], r["packets"][len(order) - 2 + j]) \
                    and np.array_equal(got_pcm[i & 1], r["pcm"][len(order) - 2 + j])
        else:   # decode-only run: features (generate) or packets (decode) from the prepared rings, per-stream decoder state
            feats = self.feats[:, :S].cpu().numpy()
            pks = self.pk_seq[:, :S].cpu().numpy()

            def one(sidx):
                st = lyra_oracle.Stream(o)
                out = []
                for k, i in order:
                    f = feats[i % feats.shape[0], sidx] if k == "generate" else \
                        o.rvq_decode(o.unpack(pks[i % pks.shape[0], sidx][None], ns))[0]
                    out.append(st.decode(f))
                return out[-2:]
            with ThreadPoolExecutor(max_workers=usable_cores()) as ex:
                tails = list(ex.map(one, range(S)))
            for j, (_, i) in enumerate(last):
                ok = ok and np.array_equal(got_pcm[i & 1], np.stack([t[j] for t in tails]))
        return {"verified": bool(ok), "streams": S, "steps_replayed": len(order), "requant_mode": a.requant,
                "seconds": round(time.perf_counter() - t0, 2),
                "what": "packets + PCM of the run's last two steps == CPU oracle replaying every step since the reset "
                        "(bit-exact; state carried, so the end state vouches for the run)"}

    # -- steps --------------------------------------------------------------------------------------------------
    # kind: "encdec" (encode + decode of every hop), "generate" (features -> PCM), "decode" (packets -> PCM)
    def steps(self, kind, first, n):
        """n consecutive steps starting at absolute step `first`: ONE library call (lyra_hip_run_steps_dev)."""
        self.history.append((kind, first, n))
        if self.per_call:
            for i in range(first, first + n):
                self._step_per_call(kind, i)
            return
        a = self.args
        self.ctx.run_steps_dev(
            self.ids, self.wl["bits"], n, first_step=first,
            d_pcm_ring=self.pcm_in if kind == "encdec" else None,
            d_packets=self.packets, d_pcm_out=self.pcm_out,
            d_features=self.feats if kind == "generate" else None,
            d_packet_ring=self.pk_seq if kind == "decode" else None,
            d_packet_bytes=self.packet_bytes if (a.dtx and kind == "encdec") else None,
            d_is_noise=self.is_noise if a.full_decoder else None,
            external_rate=self.rate, d_ext_out=self.ext_out,
            encode=kind == "encdec", decode=True, dtx=a.dtx and kind == "encdec", decoder_noise=a.full_decoder)

    def _step_per_call(self, kind, i):
        ctx, bits, a, s = self.ctx, self.wl["bits"], self.args, i & 1
        if kind == "encdec" and self.rate != 16000 and not a.per_call_split:
            # one hop at an external rate, ONE call per side (lyra_hip_encode_ext_dev / lyra_hip_decode_ext_dev)
            x = self.pcm_in[i % self.pcm_in.shape[0]]
            ctx.encode_ext_dev(self.ids, x, self.rate, bits, self.packets[s], self.packet_bytes[s] if a.dtx else None, dtx=a.dtx)
            ctx.decode_ext_dev(self.ids, self.packets[s], bits, self.rate, self.pcm_out[s], self.ext_out[s],
                               self.is_noise if a.full_decoder else None)
            if self.mel is not None:
                ctx.logmel_dev(self.ids, self.pcm_out[s], self.mel)
            return
        if kind == "encdec":
            x = self.pcm_in[i % self.pcm_in.shape[0]]
            if self.rate != 16000:
                ctx.resample_dev(self.ids, x, self.rate, 16000, self.in16, side="encoder")
                x = self.in16
            if a.dtx:
                ctx.encode_dtx_dev(self.ids, x, bits, self.packets[s], self.packet_bytes[s])
            else:
                ctx.encode_dev(self.ids, x, bits, self.packets[s])
            ctx.decode_dev(self.ids, self.packets[s], bits, self.pcm_out[s])
        elif kind == "generate":
            ctx.generate_dev(self.ids, self.feats[i % self.feats.shape[0]], self.pcm_out[s])
        else:
            ctx.decode_dev(self.ids, self.pk_seq[i % self.pk_seq.shape[0]], bits, self.pcm_out[s])
        if a.full_decoder:
            ctx.noise_receive_dev(self.ids, self.pcm_out[s], self.is_noise, side="decoder")
        if self.mel is not None:
            ctx.logmel_dev(self.ids, self.pcm_out[s], self.mel)
        if self.rate != 16000:
            ctx.resample_dev(self.ids, self.pcm_out[s], 16000, self.rate, self.ext_out[s], side="decoder")

    def sync(self):
        self.ctx.synchronize()
        self.torch.cuda.synchronize(self.dev)

    def kernel_table(self, kind, first, nsteps):
        """Serialised pass: every kernel bracketed by HIP events, the library streams strictly in call order."""
        ctx = self.ctx
        ctx.set_serial(True)
        ctx.profile_enable(True)
        self.steps(kind, first, 2)   # first launches after the mode switch: not representative
        ctx.synchronize()
        ctx.profile_read()
        self.steps(kind, first + 2, nsteps)
        ctx.synchronize()
        prof = ctx.profile_read()
        ctx.profile_enable(False)
        ctx.set_serial(False)
        B = self.wl["B"]
        return {k: kernel_row(k, ms, n, B) for k, (ms, n) in prof.items() if n and k in KERNEL_WORK}

    def timed(self, kind, first, K, barrier, only=None):
        """K steps between barriers + synchronises -> (seconds, dominant-kernel profile)."""
        ctx = self.ctx
        if only:   # a sample of the launches (an event record is a stream packet of its own, ~5 us of bubble each)
            ctx.profile_enable(True, only=only, every=dominant_sample_every(K))
        barrier()
        self.sync()
        t0 = time.perf_counter()
        self.steps(kind, first, K)
        self.enqueue_seconds = time.perf_counter() - t0     # host time to enqueue the whole region
        self.sync()
        barrier()
        t1 = time.perf_counter()
        prof = ctx.profile_read() if only else {}
        ctx.profile_enable(False)
        return t1 - t0, prof

    def latency(self, kind, first, M):
        """M isolated steps: enqueue one step, wait until every output of it is complete.  -> list of seconds."""
        out = []
        self.sync()
        for i in range(first, first + M):
            t0 = time.perf_counter()
            self.steps(kind, i, 1)
            self.ctx.synchronize()
            out.append(time.perf_counter() - t0)
        return out


class StubShard:
    """Stand-in for Shard under --stub-context (tests/test_abi_cpu.py): same interface, no GPU.  A "step" increments a
    counter; timed() reports K * 0.1 ms * (1 + rank / 2), so the max over ranks is known to the test."""

    def __init__(self, rank, first_id, wl, args, weights_image=None):
        self.rank, self.first_id, self.wl, self.args = rank, first_id, wl, args
        self.weights_bytes = len(weights_image) if weights_image is not None else 0
        self.calls = {"encdec": 0, "generate": 0, "decode": 0, "sync": 0}

    def steps(self, kind, first, n):
        self.calls[kind] += n

    def sync(self):
        self.calls["sync"] += 1

    def kernel_table(self, kind, first, nsteps):
        self.steps(kind, first, nsteps + 2)
        return None

    def timed(self, kind, first, K, barrier, only=None):
        barrier()
        self.steps(kind, first, K)
        barrier()
        return K * 1e-4 * (1.0 + 0.5 * self.rank), {}

    def latency(self, kind, first, M):
        self.steps(kind, first, M)
        return [1e-4 * (1 + (i % 5 == 4)) for i in range(M)]

    def verify(self, n_streams):
        return {"verified": True, "streams": 0, "steps_replayed": sum(self.calls[k] for k in ("encdec", "generate", "decode")),
                "what": "stub"}


def dominant_sample_every(K):
    """Every how-many-th launch of the two chain-leading kernels is bracketed inside the timed region: ~32 samples each of a
    long run, every fourth launch of a 20-step driver run (5 samples each: an event record is a stream packet of its own,
    and 2 x 10 pairs cost the 6 ms region ~1 % -- tools/k20_repeat.py without brackets 13.85 M, bench.py with them 13.69 M)."""
    if K >= 64:
        return K // 32
    return max(2, K // 5) if K >= 4 else 1


def latency_stats(samples):
    """min / mean / p50 / p99 / max / stddev in microseconds (lyra_benchmark_lib.cc:164-182 prints these per stage)."""
    if not samples:
        return None
    a = np.sort(np.asarray(samples, np.float64)) * 1e6
    pick = lambda q: float(a[min(len(a) - 1, int(np.ceil(q * len(a))) - 1)])
    return {"n": int(len(a)), "min": round(float(a[0]), 1), "mean": round(float(a.mean()), 1), "p50": round(pick(0.5), 1),
            "p99": round(pick(0.99), 1), "max": round(float(a[-1]), 1), "stddev": round(float(a.std()), 1)}


def run_shard(sh, args, wl, barrier):
    """Warm-up, serialised kernel table, timed region(s), latency leg of one shard -> dict of raw results."""
    K, W = args.steps, args.warmup
    kind = "encdec" if wl["mode"] == "encdec" else "generate"
    sh.steps(kind, 0, W)
    sh.sync()
    table = None
    cursor = W
    if not args.no_kernel_table:
        table = sh.kernel_table(kind, cursor, 8)
        cursor += 10
        sh.steps(kind, cursor, 2)   # back to the overlapped regime before timing
        cursor += 2
        sh.sync()
    # the two longest kernels (one per chain in practice: enc_s0 and dec_s1) are both bracketed inside the timed region
    doms = sorted(table, key=lambda k: -table[k]["avg_us"])[:2] if table else []
    dom = doms[0] if doms else None
    if args.ramp_steps > 0:     # untimed; timed() synchronises behind them and starts the clock at once
        sh.steps(kind, cursor, args.ramp_steps)
        cursor += args.ramp_steps
    secs, prof = sh.timed(kind, cursor, K, barrier, only=doms or None)
    cursor += K
    res = {"seconds": secs, "table": table, "dom": dom, "dom_prof": prof.get(dom) if dom else None,
           "doms": [(k, prof.get(k)) for k in doms],
           "enqueue_seconds": getattr(sh, "enqueue_seconds", None)}
    if args.latency_steps > 0:
        res["latency"] = sh.latency(kind, cursor, args.latency_steps)
        cursor += args.latency_steps
    if wl["mode"] == "decode":
        sh.steps("decode", cursor, 3)
        sh.sync()
        cursor += 3
        table2 = None
        if not args.no_kernel_table:
            table2 = sh.kernel_table("decode", cursor, 8)
            cursor += 10
            sh.sync()
        if args.ramp_steps > 0:
            sh.steps("decode", cursor, args.ramp_steps)
            cursor += args.ramp_steps
        secs2, _ = sh.timed("decode", cursor, K, barrier)
        res["secondary_seconds"] = secs2
        res["secondary_table"] = table2
    if not args.no_verify:
        try:
            res["verify"] = sh.verify(args.verify_streams)
        except Exception as e:   # a broken checker must not take the measurement down; it must not pass silently either
            res["verify"] = {"verified": False, "error": repr(e)}
    return res


# ------------------------------------------------------------------------------------------------------------
# result line
# ------------------------------------------------------------------------------------------------------------
def result_line(args, wl, world, secs, frames, res, launcher):
    K, W, B, bits, mode = args.steps, args.warmup, wl["B"], wl["bits"], wl["mode"]
    value = frames / secs
    per_gpu = value / world
    traffic_table = load_traffic()
    roof = step_roofline(mode, legs_of(args), per_gpu, traffic_table)
    names, f32, i8, moved, _ = step_work(mode, legs_of(args))
    what = "encode+decode" if mode == "encdec" else "decode only (lyra_hip_generate_dev: features -> PCM)"
    out = {
        "metric": "20ms 16kHz frames/sec encode+decode (whole node) at batch 4096; xRT/stream",
        "value": round(value, 1), "unit": "frames/s", "n_gpus": world, "steps": K, "warmup": W,
        "ms_per_step": round(secs / K * 1e3, 4), "untimed_ramp_steps": args.ramp_steps, "higher_is_better": True,
        "scaling": wl["scaling"],
        "vs_baseline": None, "dtype": "f32+i8 (fp32 and int8 layers exactly as the reference graphs)",
        "data": "synthetic",
        "config": {"workload": f"BASELINE config #{wl['config']}: batch={B} streams/GPU x 1 frame(20 ms, 320 samples)"
                               f"/step, {bits} bits ({bits // 4}-stage RVQ, {bits * 50} bps), {what}"
                               + (", log-mel front end on every decoded hop" if args.with_logmel else "")
                               + (", NoiseEstimator on every decoded hop" if args.full_decoder else "")
                               + (", DTX encoder (every 4th stream silent)" if args.dtx else "")
                               + (f", {args.rate} Hz external rate (both resamplers)" if args.rate != 16000 else "")
                               + ", state carried across steps",
                   "baseline_config": wl["config"], "streams_per_gpu": B, "total_streams": wl["total"],
                   "num_bits": bits,
                   "parallelism": f"streams sharded over {world} GPU(s), no data-path collective ({launcher})",
                   "requant_mode": args.requant, "sub_batches": wl.get("sub_batches") or 1,
                   "driver": "python, one `_dev` call per codec call" if (args.per_call or args.with_logmel)
                   else "lyra_hip_run_steps_dev: one C call per timed region"},
        "xrt_per_stream": round(value / 50.0 / wl["total"], 3),
        "xrt_aggregate": round(value / 50.0, 1),
        "roofline": roof,
    }
    if res.get("table"):
        out["kernels"] = res["table"]
        out["kernels_note"] = ("serialised pass (lyra_hip_set_serial): HIP events on the library's streams, no "
                               "cross-stream contention; 8 steps; frac = floor_us / avg_us at the kernel's binding bound")
        out["serial_sum_us"] = round(sum(r["avg_us"] for r in res["table"].values()), 2)
    if res.get("dom") and res.get("dom_prof") and res["dom_prof"][1]:
        ms, n = res["dom_prof"]
        row = kernel_row(res["dom"], ms, n, B)
        row.update(kernel=res["dom"], measured_in=f"timed region (library streams overlapping), every "
                                                  f"{dominant_sample_every(K)}th launch bracketed by HIP events on its own stream")
        if traffic_table and res["dom"] in traffic_table:
            row["traffic_bytes_per_launch_at_B4096"] = traffic_table[res["dom"]].get("hbm_bytes_per_launch")
        out["dominant_kernel"] = row
        others = []
        for k, pr in res.get("doms", [])[1:]:
            if pr and pr[1]:
                r2 = kernel_row(k, pr[0], pr[1], B)
                r2.update(kernel=k)
                others.append(r2)
        if others:
            out["dominant_kernels_also_bracketed"] = others
    if res.get("enqueue_seconds") is not None:
        out["host_enqueue_ms"] = round(res["enqueue_seconds"] * 1e3, 3)    # of the whole timed region (rank 0)
    if res.get("latency"):
        out["step_latency_us"] = dict(latency_stats(res["latency"]),
                                      what="one isolated step of this rank: enqueue -> every output complete "
                                           "(host clock around lyra_hip_run_steps_dev(n=1) + lyra_hip_synchronize)")
    if res.get("verify") is not None:
        out["verified"] = res["verify"].get("verified")
        out["verify"] = res["verify"]
    if mode == "decode":
        s2 = res["secondary_seconds"]
        v2 = frames / s2
        out["secondary"] = {"what": "lyra_hip_decode_dev: packets -> PCM (RVQ decode fused into decoder stage 0)",
                            "value": round(v2, 1), "unit": "frames/s", "ms_per_step": round(s2 / K * 1e3, 4)}
        if res.get("secondary_table"):
            out["secondary"]["kernels"] = res["secondary_table"]
    return out


def rank_summary(res, args, wl):
    """What every rank contributes to the N > 1 line beyond the max-reduced time."""
    out = {"ms_per_step": round(res["seconds"] / args.steps * 1e3, 4)}
    if res.get("dom") and res.get("dom_prof") and res["dom_prof"][1]:
        ms, n = res["dom_prof"]
        out["dominant_kernel"] = {"kernel": res["dom"], "avg_us": round(ms / n * 1e3, 2), "launches": n}
    if res.get("verify") is not None:
        out["verified"] = res["verify"].get("verified")
    return out


def gather_ranks(summary, world):
    if world == 1 and not pg_active():
        return [summary]
    import torch.distributed as dist
    out = [None] * world
    dist.all_gather_object(out, summary)
    return out


def selftest_dist(args):
    import torch
    import torch.distributed as dist
    rank, world, _ = dist_env()
    if world > 1:
        dist.init_process_group("gloo", init_method="env://")
        dist.barrier()
    wl = resolve_workload(args, world)
    lo, per = shard_ids(wl["total"], rank, world)
    assert per == wl["B"]
    secs, units = reduce_job(0.5 + 0.25 * rank, per * args.steps, world, torch.device("cpu"))
    if rank == 0:
        print(json.dumps({"selftest": "dist", "world": world, "seconds": secs, "units": units,
                          "first_id": lo, "per_rank": per, "scaling": wl["scaling"], "total": wl["total"]}))
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()


def broadcast_weights(rank, dev):
    """RCCL broadcast of the packed weight container (3.2 MB) from rank 0: the one collective with a
    payload in the whole job, at init, outside the timed region."""
    import torch
    import torch.distributed as dist
    import lyra_amd
    n = torch.zeros(1, dtype=torch.int64, device=dev)
    if rank == 0:
        raw = np.fromfile(os.path.join(lyra_amd.default_model_dir(), "lyra_v1.lyrapack"), dtype=np.uint8)
        n[0] = raw.size
    dist.broadcast(n, src=0)
    buf = torch.empty(int(n.item()), dtype=torch.uint8, device=dev)
    if rank == 0:
        buf.copy_(torch.from_numpy(raw))
    dist.broadcast(buf, src=0)
    return buf.cpu().numpy().tobytes()


def main_single_process(args, ngpu):
    """--gpus N without torch.distributed: one process, one thread and one context per GPU.  ctypes releases the GIL
    around every library call, so the N host threads enqueue concurrently."""
    import torch
    if torch.cuda.device_count() < ngpu:
        raise SystemExit(f"--gpus {ngpu} but only {torch.cuda.device_count()} device(s) visible")
    wl = resolve_workload(args, ngpu)
    shards = [None] * ngpu
    results = [None] * ngpu
    bar = threading.Barrier(ngpu)

    def worker(r):
        torch.cuda.set_device(r)
        first, _ = shard_ids(wl["total"], r, ngpu)
        shards[r] = Shard(r, first, wl, args)
        bar.wait()
        results[r] = run_shard(shards[r], args, wl, bar.wait)

    ths = [threading.Thread(target=worker, args=(r,)) for r in range(ngpu)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    secs = max(r["seconds"] for r in results)
    res = dict(results[0])
    if wl["mode"] == "decode":
        res["secondary_seconds"] = max(r["secondary_seconds"] for r in results)
    out = result_line(args, wl, ngpu, secs, wl["B"] * ngpu * args.steps, res, "one process, one host thread per GPU")
    print(json.dumps(out))


def free_port():
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        return so.getsockname()[1]


def respawn_under_launcher(argv, ngpu):
    """`python bench.py --gpus N` as the driver types it, N > 1, no launcher: re-execute under torch.distributed.run so
    that the job really is one process per GPU with a process group (RCCL on the GPU box, gloo under --stub-context) --
    the same command line the launcher form documents.  Rank 0's JSON line is this process's output."""
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={ngpu}",
           "--master-addr", "127.0.0.1", "--master-port", str(free_port()), os.path.abspath(__file__)] + list(argv)
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", LYRA_BENCH_RESPAWNED="1")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.call(cmd, env=env)


def main(argv=None):
    argv = sys.argv[1:] if argv is None else list(argv)
    args = parse(argv)
    if args.selftest_dist:
        return selftest_dist(args)
    import torch
    rank, world, local = dist_env()
    stub = args.stub_context
    if not stub and not torch.cuda.is_available():
        raise SystemExit("bench.py needs a GPU (there is no CPU fallback); use --selftest-dist for the plumbing test")
    if world == 1 and args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        if args.single_process and not stub:
            return main_single_process(args, args.gpus)
        if not stub and args.oversubscribe_device is None and torch.cuda.device_count() < args.gpus:
            raise SystemExit(f"--gpus {args.gpus} but only {torch.cuda.device_count()} device(s) visible")
        rc = respawn_under_launcher(argv, args.gpus)
        if rc:
            raise SystemExit(rc)
        return
    if world != args.gpus and world > 1:
        print(f"warning: --gpus {args.gpus} but WORLD_SIZE {world}", file=sys.stderr)
    over = args.oversubscribe_device if not stub else None
    gpu = local if over is None else over        # the GPU this rank drives
    if stub:
        dev = torch.device("cpu")
    else:
        torch.cuda.set_device(gpu)
        dev = torch.device("cuda", gpu) if over is None else torch.device("cpu")   # where the collectives' tensors live
    use_pg = world > 1 or args.force_dist
    confined = confine_rank_to_cpu_share(local, world) if (world > 1 and not args.no_cpu_confine) else None
    if use_pg:
        import torch.distributed as dist
        if world == 1 and "MASTER_ADDR" not in os.environ:   # --force-dist typed without a launcher: a one-rank rendezvous
            import socket
            with socket.socket() as so:
                so.bind(("127.0.0.1", 0))
                port = so.getsockname()[1]
            os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1", LOCAL_RANK=str(local))
        if stub or over is not None:
            dist.init_process_group("gloo", init_method="env://")
        else:
            dist.init_process_group("nccl", init_method="env://", device_id=dev)
    wl = resolve_workload(args, world)
    first_id, _ = shard_ids(wl["total"], rank, world)
    image = broadcast_weights(rank, dev) if (args.bcast_weights and use_pg) else None
    sh = StubShard(rank, first_id, wl, args, weights_image=image) if stub else Shard(gpu, first_id, wl, args,
                                                                                     weights_image=image)

    def barrier():
        if use_pg:
            import torch.distributed as dist
            dist.barrier()

    res = run_shard(sh, args, wl, barrier)
    secs, frames = reduce_job(res["seconds"], wl["B"] * args.steps, world, dev)
    if wl["mode"] == "decode":
        res["secondary_seconds"], _ = reduce_job(res["secondary_seconds"], 0, world, dev)
    per_rank = gather_ranks(rank_summary(res, args, wl), world)
    out = None
    if rank == 0:
        backend = "gloo (stub)" if stub else "RCCL" if over is None else f"gloo, every rank on GPU {over}"
        if use_pg:
            how = (f"one process per GPU, torch.distributed/{backend}: {world} rank(s), process group used for the "
                   "timing barrier and the result reduction only")
        else:
            how = "one process, one GPU: no process group at one rank (the barrier is a no-op, nothing is reduced)"
        if os.environ.get("LYRA_BENCH_RESPAWNED"):
            how += ", self-spawned from `python bench.py --gpus N`"
        out = result_line(args, wl, world, secs, frames, res, how)
        out["ranks"] = world
        if use_pg and world == 1:
            out["process_group"] = "forced at one rank (--force-dist): barrier, max / sum reduction and rank gather ran over " + backend
        if world > 1:   # a straggler must be visible: every rank's own clock, dominant kernel and self-check
            out["per_rank_ms_per_step"] = [r["ms_per_step"] for r in per_rank]
            out["per_rank"] = per_rank
            out["verified"] = None if any(r.get("verified") is None for r in per_rank) else all(r["verified"] for r in per_rank)
            out["cpus_per_rank"] = len(confined[1]) if confined else None
        if stub:
            out["stub"] = {"calls": sh.calls, "first_id": first_id, "weights_bytes": sh.weights_bytes}
        if over is not None:
            out["oversubscribed"] = (f"HARDWARE SMOKE, not a scaling measurement: all {world} rank(s) drive GPU {over} "
                                     "(--oversubscribe-device); `value` is what ONE GPU shared by them delivered")
            out["scaling"] = "oversubscribed"
    # The process group goes away BEFORE the CPU legs: the other ranks have left by then (no rank spinning in a barrier
    # on the host cores the baseline is timed on), and nothing the baseline's threads do can reach another rank's clock.
    if use_pg:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        if not args.no_cpu_baseline and not stub:   # rank 0's host cores, at any world size
            if confined:   # back to the CPUs a one-rank run has
                try:
                    os.sched_setaffinity(0, confined[0])
                except OSError:
                    pass
            try:
                out["cpu_baseline"] = cpu_baseline(wl["bits"], wl["mode"], args.requant)
            except Exception as e:  # the baseline leg must never take the GPU number down with it
                out["cpu_baseline"] = {"error": repr(e)}
            if args.requant == "xnnpack":
                try:
                    from oracle import lyra_oracle
                    if lyra_oracle.xnn_engine_available():
                        out["cpu_baseline_xnnpack"] = cpu_baseline(wl["bits"], wl["mode"], args.requant, engine="xnnpack")
                    else:
                        out["cpu_baseline_xnnpack"] = {"error": "oracle/_xnn/liblyra_oracle_xnn.so not built"}
                except Exception as e:
                    out["cpu_baseline_xnnpack"] = {"error": repr(e)}
        print(json.dumps(out))


if __name__ == "__main__":
    main()
