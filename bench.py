#!/usr/bin/env python3
"""bench.py -- FFV2 encode throughput on MI355X (one process per GPU).

metric  : Mpix/s encode (BASELINE.json), luma pixels W*H per second, whole job.
workload: BASELINE config 3 as it reaches encode2():
          3840x2160 planar 4:4:4 10-bit (yuv444p10le), qp = global_quality = 0,
          "slices=8" read as 8 frames in flight per step.  Frames are synthetic
          (S1 structured / S2 uniform noise, seeded), resident in HBM before the
          timed region; packets land in HBM.
step    : one ffv2amd_encode_batch_device call = T-stage kernel + E-stage kernels
          over the 8 frames, coefficients materialised in HBM (the reference's
          temp2[]; --no-coef measures the fully fused qp=0 variant instead).
N > 1   : frames are independent (no inter-frame state, ffv2enc.c:461-469), so
          each rank encodes its own frames: weak scaling, no data-path collective;
          RCCL is used only for the barrier and the max-over-ranks time.
          `--gpus N` without an outer torchrun (WORLD_SIZE unset) starts the N ranks
          itself -- python -m torch.distributed.run as a child process, created before
          anything touches the GPU -- and passes rank 0's JSON line through; under an
          outer torchrun WORLD_SIZE rules.
--qp N  : (not the BASELINE metric) the qp > 0 path with the whole entropy coder on the
          device, the range coder's serial chain running one frame per lane over many
          frames in flight (ffv2_lanecoder.hip): a step is one call over --frames-in-flight
          device-resident frames (default: what 160 GB of coder scratch hold), calls back to
          back; "chain" reports the chain kernels from device events, "roofline" the PVQ
          search kernel against the f32 division rate, cpu_baseline the oracle at the same qp.
          --host-coder / --device-coder select the older coders.  Parity unpinned for qp > 0.
host_boundary : after the timed region, frames in HOST memory through the asynchronous ring
          (SURVEY.md 8(d)'s wording of the metric, PCIe included): page-locked, pageable,
          pageable from a registered pool; 4:4:4 and the literal yuv420p* formats.
steady_state  : the same launch back to back for --steady-seconds after the timed region
          (hundreds of timed launches; with and without timing events).

Prints ONE JSON line on rank 0.
"""
import argparse
import json
import os
import sys
import time

import numpy as np
import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

CONFIGS = {
    # name: (width, height, pix_fmt, depth)
    "C2": (1920, 1080, "yuv444p", 8),
    "C3": (3840, 2160, "yuv444p10le", 10),
    "C4": (3840, 2160, "yuv444p", 8),
    "C5": (7680, 4320, "yuv444p12le", 12),
}
HBM_PEAK_GBS = 8000.0     # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
PREROLL_GROUP, PREROLL_MAX = 10, 150
EVENT_PERIOD = 4


def yuv420_of(frame):
    """A yuv420p* frame made of a synthetic 4:4:4 one: its luma, its chroma planes decimated."""
    return [np.ascontiguousarray(frame[0]), np.ascontiguousarray(frame[1][::2, ::2]), np.ascontiguousarray(frame[2][::2, ::2])]


def host_boundary(FFV2Encoder, W, H, fmt, local, host_frames, nframes, depth, pinned, barrier, yuv420=False, register=False):
    """Host frames in, host packets out through the asynchronous ring (ffv2amd_ring_*): the
    metric as SURVEY.md 8(d) words it, PCIe included.  yuv420: the literal BASELINE pixel format
    (ffv2amd_ring_send_420: half the bytes over PCIe, chroma up-converted on the device).
    register: pageable frames from a pool of long-lived buffers (what libavcodec's frame pools are), page-locked
    by the ring the first time it sees each buffer (FFV2AMD_FRAME_REGISTER).
    Returns (seconds, packets, bytes per frame)."""
    enc = FFV2Encoder(W, H, fmt, device=local, max_batch=1)
    enc.ring_open(depth)
    nsrc = host_frames.shape[0]
    if yuv420:
        src = [yuv420_of(f) for f in host_frames]
        if pinned:
            pool = enc.pinned_frames_420(nsrc)
            for dst, planes in zip(pool, src):
                for d, a in zip(dst, planes):
                    d[:] = a
            src = pool
        frame_bytes = sum(a.shape[0] * a.shape[1] for a in src[0]) * enc.dtype.itemsize
        send = lambda n: enc.ring_send_420(*src[n % nsrc], tag=n, pinned=pinned, register=register)
    else:
        if pinned:
            src = enc.pinned_frames(nsrc)
            src[:] = host_frames
        else:
            src = host_frames
        frame_bytes = enc.info.planes * enc.info.width * enc.info.height * enc.dtype.itemsize
        send = lambda n: enc.ring_send(src[n % nsrc], tag=n, pinned=pinned, register=register)

    def run(n):
        packets, sent = [], 0
        while len(packets) < n:
            while sent < n and send(sent):
                sent += 1
            tag, pk = enc.ring_receive(wait=True)
            assert tag == len(packets), "ring delivered out of order"
            packets.append(pk)
        return packets

    run(min(nframes, 2 * depth))                    # warm-up
    barrier()
    t0 = time.perf_counter()
    packets = run(nframes)
    barrier()
    dt = time.perf_counter() - t0
    enc.ring_close()
    enc.free_pinned()
    enc.close()
    return dt, packets, frame_bytes


def h2d_rate(dev, nbytes, reps=8):
    """Plain pinned-host -> device copy rate (GB/s) of frame-sized buffers: what PCIe gives."""
    src = torch.empty(nbytes, dtype=torch.uint8).pin_memory()
    dst = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    dst.copy_(src, non_blocking=True)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps):
        dst.copy_(src, non_blocking=True)
    e1.record()
    torch.cuda.synchronize()
    return nbytes * reps / (e0.elapsed_time(e1) * 1e-3) / 1e9


def lanecoder_bench(args, enc0, FFV2Encoder, synth, cfg, dist_cfg, barrier):
    """--qp N --frames-in-flight F: the qp > 0 path with the whole entropy coder on the device, the
    range coder's serial chain running one frame per lane.  Not the BASELINE
    metric; PARITY UNPINNED for qp > 0 (the oracle restates the reference's PVQ asm)."""
    W, H, fmt, depth, P = cfg
    world, rank, local, dev, backend = dist_cfg
    enc0.close()
    F = args.frames_in_flight
    enc = FFV2Encoder(W, H, fmt, device=local, max_batch=16)
    distinct = min(F, 64)
    host_frames = np.stack([synth.noise(rank * distinct + n, P, H, W, depth) for n in range(distinct)])
    d = enc.upload(host_frames)
    d_frames = d.repeat((F + distinct - 1) // distinct, *([1] * (d.dim() - 1)))[:F].contiguous()
    del d
    from ffmpeg_ffv2_amd._lib import FFV2Error
    while True:                                   # fewer frames in flight if the device cannot hold the scratch
        try:
            enc.lanecoder_open(F, args.packet_cap, args.calls_in_flight, args.backs)
            break
        except FFV2Error as ex:
            if ex.code != -12 or F <= 64:
                raise
            F = max(64, F // 2 // 64 * 64)
            d_frames = d_frames[:F]
    def finish():
        if not args.strided_packets:
            return enc.lanecoder_finish_packed()
        pk, sizes, status = enc.lanecoder_finish(packet_stride=stride)
        return pk.reshape(-1), np.arange(F, dtype=np.uint64) * np.uint64(stride), sizes, status

    enc.lanecoder_submit(d_frames, args.qp)
    stride = int(enc.lanecoder_finish_packed()[2].max()) * 5 // 4 + 4096      # row pitch of the strided variant
    for _ in range(max(args.warmup, 1)):
        enc.lanecoder_submit(d_frames, args.qp)
        finish()
    barrier()
    # two calls in flight: the front of step i+1 (T-stage, PVQ, bookkeeping) runs beside the chain of
    # step i; the packets of a step come back packed, in one copy
    t0 = time.perf_counter()
    ahead = args.calls_in_flight - 1
    for i in range(min(ahead, args.steps)):
        enc.lanecoder_submit(d_frames, args.qp)
    for i in range(args.steps):
        if i + ahead < args.steps:
            enc.lanecoder_submit(d_frames, args.qp)
        buf, offs, sizes, status = finish()
    barrier()
    dt = time.perf_counter() - t0
    if world > 1:
        import torch.distributed as dist
        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        dt = float(tt.item())
    if rank == 0:
        res = {"metric": "Mpix/s encode, qp=%d (not the BASELINE metric; parity unpinned for qp > 0)" % args.qp,
               "value": round(world * F * args.steps * W * H / dt / 1e6, 1), "unit": "Mpix/s",
               "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
               "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
               "vs_baseline": None, "dtype": "int32 + f32 (PVQ search)", "data": "synthetic",
               "config": {"workload": "%dx%d %s qp=%d, %d frames per step per GPU (%d distinct noise frames, repeated), "
                                      "frames resident in HBM, packets to host memory" % (W, H, fmt, args.qp, F, distinct),
                          "packet_bytes_frame0": int(sizes[0]), "frames_failed": int((status != 0).sum()),
                          "range_coder": "device, range chain one frame per lane, %d frames in flight" % F,
                          "coder_scratch_GB": round(F * enc.lanecoder_bytes_per_frame(args.packet_cap, args.calls_in_flight, args.backs) / 1e9, 1),
                          "calls_in_flight": args.calls_in_flight, "range_chains_side_by_side": args.backs or 1,
                          "packet_cap": args.packet_cap or enc.info.packet_cap_qp,
                          "packets_out": "one copy per packet" if args.strided_packets else "packed, one copy"},
               "chain": dict(zip(("kernel", "ms", "back_ms", "symbols_frame0", "ns_per_symbol", "what"),
                                 ("lc_chain_kernel",) + (lambda c, b, n: (round(c, 2), round(b, 2), n, round(c * 1e6 / max(n, 1), 1)))(
                                     *enc.lanecoder_stats()) +
                                 ("the range recurrence, one frame per lane: its time does not depend on the frames in flight; "
                                  "bound by the issue rate of a lone wavefront, not by HBM or MFMA (last step's values)",))),
               "roofline": None}
        # the Q-stage kernel alone (the largest part of a call's front once the range chain is amortised over
        # thousands of frames): every element of a band is re-scored for every pulse, as the asm does
        # (celt_pvq_search.asm:85-191), one IEEE f32 division each.  Neither HBM nor MFMA bounds it; it is held
        # against the rate at which the chip can issue the 11-instruction division sequence.
        try:
            import ctypes as C
            from ffmpeg_ffv2_amd import _lib as L
            ms = C.c_float(0)
            nq = min(16, F)
            L.check(L.load().ffv2amd_debug_pvq_time(enc._h, nq, d_frames.data_ptr(), args.qp, 10, C.byref(ms)), "pvq_time")
            bands = [15, 8, 8, 32, 32, 32, 128, 128, 128, 512, 512, 512, 2049]
            # pulses left to the greedy loop: on flat spectra the projection rounds every pulse away, all qp are searched
            div_per_bp = sum(((n + 3) // 4 * 4) * args.qp for n in bands)
            divs = div_per_bp * enc.info.block_planes * nq
            rate = divs / (ms.value * 1e-3)
            peak = 256 * 4 * 16 * 2.4e9 / 11            # CUs x SIMDs x lanes x clock / instructions of the division sequence
            general = args.qp > 64 or os.environ.get("FFV2AMD_PVQ_GENERAL", "0") not in ("", "0")
            res["roofline"] = {"bound": "valu (IEEE f32 division, 11 instructions)",
                               "kernel": "ffv2_pvq_kernel" if general else "ffv2_pvq_lists_kernel",
                               "achieved": round(rate / 1e12, 3), "peak": round(peak / 1e12, 3), "unit": "T divisions/s",
                               "frac": round(rate / peak, 3), "traffic": None,
                               "kernel_ms_avg": round(ms.value, 4), "frames_per_launch": nq,
                               "divisions_per_block_plane": div_per_bp,
                               "what": "the Q-stage kernel alone, %d frames per launch, 10 launches; divisions = ALGORITHMIC: elements x "
                                       "pulses as the reference's search performs them (celt_pvq_search.asm:85-191; upper bound: every "
                                       "pulse searched) -- the list search reaches the same pulses with far fewer; peak = f32 VALU "
                                       "lanes x 2.4 GHz / 11" % nq}
        except Exception as ex:
            res["roofline"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_host_boundary:
            # the same qp with the frames in HOST memory and the packets back in host memory, through the qp > 0 ring
            # (ffv2amd_qpring_*: what send_frame / receive_packet reach with qp_frames_per_call set)
            try:
                first = buf[int(offs[0]): int(offs[0]) + int(sizes[0])].tobytes()
                d_frames = None                  # the resident frames' HBM is the ring's now
                torch.cuda.empty_cache()
                res["host_boundary"] = qp_host_boundary(args, enc, host_frames, F, first)
            except Exception as ex:
                res["host_boundary"] = {"error": "%s: %s" % (type(ex).__name__, ex)}
        if world == 1 and not args.no_cpu_baseline:
            from tests import oracle_lib
            oracle = oracle_lib.load()
            n_done, tcpu, ok = 0, 0.0, True
            while n_done < distinct and tcpu < args.cpu_seconds:
                c0 = time.perf_counter()
                try:
                    ref = oracle.encode(host_frames[n_done], fmt, qp=args.qp)
                except RuntimeError:
                    ref = None                            # the reference would av_assert0 on this frame (daala_entropy.c:336)
                tcpu += time.perf_counter() - c0
                for i in range(n_done, F, distinct):      # every repetition of this frame
                    if ref is None:
                        ok = ok and status[i] == -1
                    else:
                        ok = ok and status[i] == 0 and buf[int(offs[i]): int(offs[i]) + int(sizes[i])].tobytes() == ref
                n_done += 1
            res["cpu_baseline"] = {"value": round(n_done * W * H / tcpu / 1e6, 2), "unit": "Mpix/s", "cores": 1,
                                   "kind": "port", "sample": "%d of the %d distinct benchmark frames, oracle qp=%d, 1 thread"
                                                             % (n_done, distinct, args.qp),
                                   "gpu_packets_match_cpu": bool(ok)}
            if not ok:
                res["error"] = "GPU packets differ from the CPU oracle"
        print(json.dumps(res))
    enc.close()
    if world > 1:
        import torch.distributed as dist
        dist.barrier()
        dist.destroy_process_group()


def qp_host_boundary(args, enc, host_frames, F, packet0):
    """Frames in host memory -> ffv2amd_qpring_send / _receive -> packets in host memory at qp > 0: batches of
    frames coded side by side on the device, two to four batches in flight.  Returns the host_boundary block."""
    enc.lanecoder_close()
    i = enc.info
    W, H = i.width, i.height
    # a call lasts at least one frame's range chain (0.6 s for 1080p noise at qp 16) whatever it holds: large batches
    per_call = max(64, min(F // 2, 4096) // 64 * 64)
    if args.host_frames_per_call > 0:
        per_call = args.host_frames_per_call
    total = max(8 * per_call, min(16384, 32 * per_call))     # filling and draining the pipeline included
    nsrc = host_frames.shape[0]
    out = {"frames_per_call": per_call, "frames_sent": total, "what":
           "frames in host memory -> ffv2amd_qpring_send (H2D as they arrive, a full batch = one lane coder call, two to four in "
           "flight, their range chains side by side) -> ffv2amd_qpring_receive: packets in host memory in send order; pinned: page-locked frames read in "
           "place; pageable: rows copied by the ring's helper threads into page-locked bounce frames; pageable_registered: ordinary "
           "memory from a pool of buffers, page-locked by the ring on first sight (FFV2AMD_FRAME_REGISTER); yuv420: the literal 4:2:0 "
           "frames, up-converted on the device"}
    y420 = [yuv420_of(f) for f in host_frames] if i.planes == 3 else None
    for name, pinned, is420 in [("pinned", True, False), ("pageable", False, False), ("pageable_registered", False, False),
                                ("yuv420_pinned", True, True)]:
        if is420 and y420 is None:
            continue
        if pinned and not is420:
            src = enc.pinned_frames(nsrc)
            src[:] = host_frames
            src = [src[n] for n in range(nsrc)]
        elif pinned:
            pool = enc.pinned_frames_420(nsrc)
            for dst, planes in zip(pool, y420):
                for d, a in zip(dst, planes):
                    d[:] = a
            src = pool
        else:
            src = [host_frames[n] for n in range(nsrc)]
        from ffmpeg_ffv2_amd._lib import FFV2Error
        while True:                              # smaller batches if the device cannot hold three of them and the coder's scratch
            try:
                enc.qpring_open(args.qp, per_call, args.packet_cap)
                break
            except FFV2Error as ex:
                if ex.code != -12 or per_call <= 64:
                    raise
                per_call = max(64, per_call // 2 // 64 * 64)
                total = 8 * per_call                         # filling and draining the pipeline included
                out["frames_per_call"], out["frames_sent"] = per_call, total
        got, sent, first, flushed = 0, 0, None, False
        t0 = time.perf_counter()
        while got < total:
            while sent < total and enc.qpring_send(src[sent % nsrc], tag=sent, pinned=pinned, yuv420=is420,
                                                   register=name.endswith("registered")):
                sent += 1
            if sent == total and not flushed:
                flushed = enc.qpring_flush()
            r = enc.qpring_receive(wait=True)
            if r is None:
                continue
            assert r[0] == got, "qp ring delivered out of order"
            if got == 0:
                first = r[1]
            got += 1
        dt = time.perf_counter() - t0
        enc.qpring_close()
        enc.free_pinned()
        out[name] = {"Mpix_s": round(total * W * H / dt / 1e6, 1), "ms_per_frame": round(dt / total * 1e3, 4)}
        if not is420:
            out[name]["packet0_equals_device_path"] = bool(first == packet0)
    return out


def spawn_ranks(n):
    """--gpus N with no outer launcher: one rank per GPU through torch.distributed.run, started as a
    CHILD process (a process that has touched the GPU must never exec; nothing here has touched it:
    `import torch` does not, torch.cuda.device_count() does not on this image).  Rank 0 prints the
    JSON line on the inherited stdout; our exit code is the launcher's."""
    import socket
    import subprocess
    with socket.socket() as sk:
        sk.bind(("127.0.0.1", 0))
        port = sk.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(n),
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("OMP_NUM_THREADS", "4")
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    return subprocess.run(cmd, env=env).returncode


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=None, help="default 200 (qp 0), 10 with --qp on the device coder")
    ap.add_argument("--warmup", type=int, default=None, help="default 50 (qp 0), 1 with --qp on the device coder")
    ap.add_argument("--config", default="C3", choices=sorted(CONFIGS))
    ap.add_argument("--frames-per-step", type=int, default=8)
    ap.add_argument("--no-coef", action="store_true", help="do not materialise coefficients (fused qp=0 path)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--pipeline", action="store_true",
                    help="E-stage of step n on the encoder's own stream, overlapping the T-stage of step n+1 "
                         "(measured: ~1 %% more Mpix/s, but the T-stage timing then includes the overlap)")
    ap.add_argument("--cpu-seconds", type=float, default=12.0)
    ap.add_argument("--device-coder", action="store_true",
                    help="with --qp: the adaptive range coder on the device (one wavefront per frame) instead of host threads")
    ap.add_argument("--no-preroll", action="store_true", help="skip the untimed clock-settling pre-roll")
    ap.add_argument("--no-host-boundary", action="store_true", help="skip the host-frames-in / host-packets-out phase")
    ap.add_argument("--host-frames", type=int, default=48, help="frames per rank in the host-boundary phase")
    ap.add_argument("--host-frames-per-call", type=int, default=0,
                    help="with --qp on the device coder: frames per lane coder call in the host-boundary phase "
                         "(default: half the frames in flight, at most 4096)")
    ap.add_argument("--ring-depth", type=int, default=4)
    ap.add_argument("--steady-seconds", type=float, default=2.0,
                    help="untimed steady-state loops after the timed region (0 = skip)")
    ap.add_argument("--frames-in-flight", type=int, default=0,
                    help="with --qp: the many-frames-in-flight device coder (ffv2_lanecoder.hip, one frame per lane of the "
                         "range chain); a step is one call over this many device-resident frames.  Default: what "
                         "160 GB of coder scratch hold at 700 B per block-plane and packet, at most 2048")
    ap.add_argument("--host-coder", action="store_true",
                    help="with --qp: the GPU + host-threads pipeline (ffv2amd_qp_submit/_finish) instead of the device coder")
    ap.add_argument("--strided-packets", action="store_true",
                    help="with --frames-in-flight: packets come back one copy each into a [frames][stride] array "
                         "(ffv2amd_lanecoder_finish) instead of packed in one copy (ffv2amd_lanecoder_finish_packed)")
    ap.add_argument("--calls-in-flight", type=int, default=2, choices=(2, 3, 4),
                    help="with --qp on the device coder: submitted calls before a finish is due (each holds its own copy of "
                         "the front's buffers)")
    ap.add_argument("--backs", type=int, default=None, choices=(1, 2, 3, 4),
                    help="with --qp on the device coder: range chains side by side (ffv2amd_lanecoder_open_ex; at most "
                         "--calls-in-flight; default: 2 where fewer than 3 800 frames are in flight, else 1)")
    ap.add_argument("--packet-cap", type=int, default=0,
                    help="with --frames-in-flight: bytes of HBM reserved per packet (0 = the encoder's bound for any qp)")
    ap.add_argument("--qp", type=int, default=0,
                    help="informational: qp > 0 times ffv2amd_encode_batch_to_host (GPU transform + PVQ, host range coder)")
    args = ap.parse_args()
    lane_mode = args.qp > 0 and not args.host_coder and not args.device_coder
    if args.steps is None:
        args.steps = 10 if lane_mode else 200
    if args.warmup is None:
        args.warmup = 1 if lane_mode else 50

    # before the HIP runtime initialises (and inherited by the ranks started below): see ffmpeg_ffv2_amd/_lib.py
    os.environ.setdefault("GPU_MAX_HW_QUEUES", "16")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    if world != args.gpus and rank == 0:
        print("bench.py: --gpus %d but the launcher started %d ranks; reporting n_gpus = %d" % (args.gpus, world, world),
              file=sys.stderr)
    # one rank per GPU; FFV2_BENCH_BACKEND=gloo lets several ranks share the single GPU
    # of a development box to rehearse the N > 1 control path (never used for numbers)
    backend = os.environ.get("FFV2_BENCH_BACKEND", "nccl")
    if backend == "nccl" and world > torch.cuda.device_count():
        sys.exit("bench.py: %d ranks need %d GPUs, this node has %d (FFV2_BENCH_BACKEND=gloo rehearses the control "
                 "path with ranks sharing GPUs; its numbers mean nothing)" % (world, world, torch.cuda.device_count()))
    assert torch.cuda.is_available(), "bench.py needs an MI355X; there is no CPU fallback"
    local = local % torch.cuda.device_count() if backend != "nccl" else local
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(backend, rank=rank, world_size=world)

    from ffmpeg_ffv2_amd import FFV2Encoder, frames as synth
    W, H, fmt, depth = CONFIGS[args.config]
    F = args.frames_per_step
    enc = FFV2Encoder(W, H, fmt, device=local, max_batch=F)
    P = enc.info.planes

    # synthetic frames: even index structured, odd index noise; rank-dependent seeds
    # (qp > 0: noise only -- structured content makes the reference abort)
    host_frames = np.stack([synth.make("S1" if n % 2 == 0 and args.qp == 0 else "S2", rank * F + n, P, H, W, depth)
                            for n in range(F)])
    d_frames = enc.upload(host_frames)
    # two output sets: with --pipeline the E-stage of step n runs on the encoder's own stream
    # while the T-stage of step n+1 is already on ours, so consecutive steps must not share
    # packet buffers
    outs = [enc.alloc_packets(F), enc.alloc_packets(F)]
    enc.set_pipelined(args.pipeline)
    if not args.no_coef:
        coef = torch.empty((F, enc.info.block_planes, 4096), dtype=torch.int32, device=dev)
        enc.set_coef_sink(coef)
    stream = torch.cuda.current_stream(dev).cuda_stream

    def barrier():
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    if lane_mode:
        if args.frames_in_flight <= 0:
            if not args.packet_cap:
                args.packet_cap = 4096 + 700 * enc.info.block_planes      # noise at qp 16 / 64 codes to 160 / 370 B per block-plane
            fit = lambda backs: max(64, min(8192, int(160e9 // enc.lanecoder_bytes_per_frame(
                args.packet_cap, args.calls_in_flight, backs)) // 64 * 64))
            args.frames_in_flight = fit(args.backs or 1)
            if args.backs is None and args.frames_in_flight < 3800:
                args.backs = 2
                args.frames_in_flight = fit(2)
        if args.backs is None:
            # fewer frames than cover a frame's chain with the next call's front (any picture size: both grow with the
            # pixels): a second chain beside the first, each call on its own back (4K / qp 16, same box: 8.0 -> 9.4 Gpix/s)
            args.backs = 2 if args.frames_in_flight < 3800 else 1
        args.backs = min(args.backs, args.calls_in_flight)
        lanecoder_bench(args, enc, FFV2Encoder, synth, (W, H, fmt, depth, P), (world, rank, local, dev, backend), barrier)
        return
    if args.qp > 0:
        # Not the headline metric (BASELINE's config is the default qp = 0): the qp > 0 path --
        # T-stage + PVQ search + symbol compaction on the GPU, adaptive range coder on host threads
        # (one serial chain per frame, ffv2enc.c:461), batch n+1 on the GPU while batch n is coded.
        # PARITY UNPINNED for qp > 0: the oracle restates the reference's PVQ asm, nothing pins it.
        enc.set_device_coder(args.device_coder)
        for _ in range(max(args.warmup, 1)):
            enc.qp_submit(d_frames, args.qp)
            pk = enc.qp_finish()
        barrier()
        t0 = time.perf_counter()
        enc.qp_submit(d_frames, args.qp)
        for i in range(args.steps):
            if i + 1 < args.steps:
                enc.qp_submit(d_frames, args.qp)          # GPU half of step i+1 ...
            pk = enc.qp_finish()                          # ... behind the host half of step i
        barrier()
        dt = time.perf_counter() - t0
        if world > 1:
            tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            dt = float(tt.item())
        if rank == 0:
            res = {"metric": "Mpix/s encode, qp=%d (not the BASELINE metric; parity unpinned for qp > 0)" % args.qp,
                   "value": round(world * F * args.steps * W * H / dt / 1e6, 1), "unit": "Mpix/s",
                   "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
                   "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak",
                   "vs_baseline": None, "dtype": "int32 + f32 (PVQ search)", "data": "synthetic",
                   "config": {"workload": "%dx%d %s qp=%d, %d frames per step per GPU, frames resident in HBM, "
                                          "packets to host memory" % (W, H, fmt, args.qp, F),
                              "packet_bytes_frame0": len(pk[0]), "host_threads": min(F, os.cpu_count() or 1),
                              "range_coder": "device, one wavefront per frame" if args.device_coder else "host threads"},
                   "roofline": None}
            if world == 1 and not args.no_cpu_baseline:
                from tests import oracle_lib
                oracle = oracle_lib.load()
                n_done, tcpu, ok = 0, 0.0, True
                while n_done < F and tcpu < args.cpu_seconds:
                    c0 = time.perf_counter()
                    ref = oracle.encode(host_frames[n_done], fmt, qp=args.qp)
                    tcpu += time.perf_counter() - c0
                    ok = ok and (ref == pk[n_done])
                    n_done += 1
                res["cpu_baseline"] = {"value": round(n_done * W * H / tcpu / 1e6, 2), "unit": "Mpix/s", "cores": 1,
                                       "kind": "port", "sample": "%d of the %d benchmark frames, oracle qp=%d, 1 thread"
                                                                 % (n_done, F, args.qp),
                                       "gpu_packets_match_cpu": bool(ok)}
                if not ok:
                    res["error"] = "GPU packets differ from the CPU oracle"
            print(json.dumps(res))
        enc.close()
        if world > 1:
            dist.barrier()
            dist.destroy_process_group()
        return
    for i in range(args.warmup):
        enc.encode_batch_device(d_frames, out=outs[i & 1], stream=stream)
    # Clock-settling pre-roll, untimed and independent of --warmup: the chip's power management
    # needs ~50-100 back-to-back launches before the T-stage time stops moving (480 -> 580 -> 400 us
    # on C3).  Groups of PREROLL_GROUP launches until three consecutive group means agree within
    # 1 %, at most PREROLL_MAX launches; the count is disclosed as "preroll_launches".
    preroll = 0
    if not args.no_preroll:
        enc.profile(True)
        enc.profile_read()
        hist = []
        while preroll < PREROLL_MAX:
            for i in range(PREROLL_GROUP):
                enc.encode_batch_device(d_frames, out=outs[i & 1], stream=stream)
            preroll += PREROLL_GROUP
            t_ms, _, n = enc.profile_read()
            hist.append(t_ms / max(n, 1))
            if len(hist) >= 3 and max(hist[-3:]) <= 1.01 * min(hist[-3:]):
                break
        enc.profile(False)
    barrier()
    # Kernel timing with HIP events on the launch stream, on every EVENT_PERIOD-th step of the timed
    # region: a pair of timing events around each kernel of every step costs 3 % of the step itself
    # (0.3685 vs 0.3571 ms), which would be charged to `value`.
    # (A run of up to 32 steps -- the driver's -- times every step: 20 samples instead of 5 are worth the
    # 3 %; `steady_state` below discloses the same launch with and without the events.)
    period = EVENT_PERIOD if args.steps > 32 else 1
    enc.profile(period)
    enc.profile_read()
    t0 = time.perf_counter()
    for i in range(args.steps):
        enc.encode_batch_device(d_frames, out=outs[i & 1], stream=stream)
    enc.flush(stream)                      # join the E-stage stream before the closing barrier
    barrier()
    t1 = time.perf_counter()
    elapsed = t1 - t0
    t_ms, e_ms, launches, t_lo, t_hi = enc.profile_read_ex()
    enc.profile(False)

    # ---- steady state, untimed (not part of `value`): the same launch back to back for
    # --steady-seconds, first with timing events around every kernel (mean / min / max of the
    # T-stage over hundreds of launches), then without any (wall clock per step: what the events
    # cost).  Also keeps the GPU busy long enough for an outside observer to see it.
    steady = None
    if args.steady_seconds > 0:
        steady = {}
        enc.profile(1)
        enc.profile_read()
        s0, acc = time.perf_counter(), [0.0, 0.0, 0, None, None]
        while time.perf_counter() - s0 < args.steady_seconds / 2:
            for i in range(100):
                enc.encode_batch_device(d_frames, out=outs[i & 1], stream=stream)
            enc.flush(stream)
            tt_, ee_, nn_, lo_, hi_ = enc.profile_read_ex()
            acc[0] += tt_; acc[1] += ee_; acc[2] += nn_
            acc[3] = lo_ if acc[3] is None else min(acc[3], lo_)
            acc[4] = hi_ if acc[4] is None else max(acc[4], hi_)
        torch.cuda.synchronize()
        wall_ev = (time.perf_counter() - s0) / max(acc[2], 1)
        enc.profile(False)
        s0, n_plain = time.perf_counter(), 0
        while time.perf_counter() - s0 < args.steady_seconds / 2:
            for i in range(100):
                enc.encode_batch_device(d_frames, out=outs[i & 1], stream=stream)
            enc.flush(stream)
            torch.cuda.synchronize()
            n_plain += 100
        wall_plain = (time.perf_counter() - s0) / max(n_plain, 1)
        steady = {"launches_timed": acc[2], "kernel_ms_mean": round(acc[0] / max(acc[2], 1), 4),
                  "kernel_ms_min": round(acc[3] or 0.0, 4), "kernel_ms_max": round(acc[4] or 0.0, 4),
                  "estage_ms_mean": round(acc[1] / max(acc[2], 1), 4),
                  "ms_per_step_with_events": round(wall_ev * 1e3, 4),
                  "launches_untimed": n_plain, "ms_per_step_without_events": round(wall_plain * 1e3, 4),
                  "what": "untimed loops after the timed region, same launch back to back (groups of 100, one "
                          "host synchronisation per group), this rank"}

    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    packets = enc.collect(*outs[(args.steps - 1) & 1])   # also raises on any per-frame error status
    enc.collect(*outs[args.steps & 1])

    # ---- host boundary: what SURVEY.md 8(d)/(e) names as the real limiter.  Every rank feeds
    # fresh host frames through the asynchronous ring (H2D inside the timed region), then the
    # packets are gathered in frame order on rank 0 (RCCL when N > 1).  Reported beside `value`,
    # never as `value`.
    hb, hb420_first = None, None
    if not args.no_host_boundary:
        try:
            from ffmpeg_ffv2_amd import fanout
            nf = args.host_frames
            hb = {}
            variants = [("pinned", True)] + ([("pageable", False), ("pageable_registered", False)] if world == 1 else [])
            for name, pinned in variants:
                dt, pk_local, frame_bytes = host_boundary(FFV2Encoder, W, H, fmt, local, host_frames, nf, args.ring_depth,
                                                          pinned, barrier, register=name.endswith("registered"))
                if world > 1:
                    tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                    dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                    dt = float(tt.item())
                hb[name] = {"Mpix_s": round(world * nf * W * H / dt / 1e6, 1), "ms_per_frame_per_gpu": round(dt / nf * 1e3, 4),
                            "h2d_GBs_per_gpu": round(nf * frame_bytes / dt / 1e9, 2)}
                if pinned:
                    # in-order gather of every rank's packets on rank 0: frame n lives on rank n % world
                    mine = fanout.local_frames(world * nf, rank, world)
                    barrier()
                    g0 = time.perf_counter()
                    allp = fanout.gather_packets(pk_local, mine, world * nf, rank, world,
                                                 device=dev if backend == "nccl" else None)
                    barrier()
                    hb["gather_ms"] = round((time.perf_counter() - g0) * 1e3, 3)
                    if rank == 0:
                        hb["ranks_seen"] = len({fanout.owner(n, world) for n, p in enumerate(allp) if p})
                        hb["packets_gathered"] = sum(1 for p in allp if p)
                        # frames cycle through the F benchmark frames: packet n of rank r == packet (n % F) of the device path
                        hb["packets_match_device_path"] = bool(all(
                            allp[n] == packets[(n // world) % F] for n in range(0, world * nf, world)))
            rate = h2d_rate(dev, frame_bytes)
            hb["h2d_copy_GBs"] = round(rate, 2)
            hb["pinned_fraction_of_h2d_copy_rate"] = round(hb["pinned"]["h2d_GBs_per_gpu"] / rate, 3)
            # the literal BASELINE pixel format: yuv420p* frames in host memory (half the PCIe bytes of
            # the 4:4:4 restatement), up-converted on the device as the ffmpeg tool's auto-inserted
            # bicubic scale filter does, then the same T/E-stage.  Parity unpinned (no libswscale here).
            if fmt.startswith("yuv444p"):
                y4 = {}
                for name, pinned in [("pinned", True)] + ([("pageable", False), ("pageable_registered", False)] if world == 1 else []):
                    dt, pk420, fb420 = host_boundary(FFV2Encoder, W, H, fmt, local, host_frames, nf, args.ring_depth,
                                                     pinned, barrier, yuv420=True, register=name.endswith("registered"))
                    if world > 1:
                        tt = torch.tensor([dt], dtype=torch.float64, device=dev if backend == "nccl" else "cpu")
                        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
                        dt = float(tt.item())
                    y4[name] = {"Mpix_s": round(world * nf * W * H / dt / 1e6, 1), "ms_per_frame_per_gpu": round(dt / nf * 1e3, 4),
                                "h2d_GBs_per_gpu": round(nf * fb420 / dt / 1e9, 2)}
                y4["pix_fmt"] = fmt.replace("444", "420")
                y4["speedup_over_444"] = round(y4["pinned"]["Mpix_s"] / hb["pinned"]["Mpix_s"], 3)
                y4["pinned_fraction_of_h2d_copy_rate"] = round(y4["pinned"]["h2d_GBs_per_gpu"] / rate, 3)
                hb["yuv420"] = y4
                hb420_first = (pk420[0], yuv420_of(host_frames[0]))
            hb["hw_queues"] = os.environ.get("GPU_MAX_HW_QUEUES")
            hb["frames_per_gpu"] = nf
            hb["ring_depth"] = args.ring_depth
            hb["what"] = ("frames in host memory -> ffv2amd_ring_send/receive -> packets in host memory, "
                          "H2D || T/E-stage || D2H on separate HIP streams; then in-order gather on rank 0.  pinned: "
                          "page-locked frames (ffv2amd_host_alloc); pageable: ordinary memory, gathered by the ring's "
                          "thread pool; pageable_registered: ordinary memory from a pool of %d buffers, page-locked by "
                          "the ring on first sight (FFV2AMD_FRAME_REGISTER; the warm-up pass pays for it)" % host_frames.shape[0])


        except Exception as ex:          # never lose the headline line to the secondary phase
            hb = {"error": "%s: %s" % (type(ex).__name__, ex)}
            if world > 1:
                try:
                    dist.barrier()
                except Exception:
                    pass

    result = None
    if rank == 0:
        frames_total = world * F * args.steps
        mpix = frames_total * W * H / elapsed / 1e6
        t_kernel_ms = t_ms / max(launches, 1)
        alg_bytes = enc.info.tstage_bytes_per_frame * F
        achieved = alg_bytes / (t_kernel_ms * 1e-3) / 1e9 if t_kernel_ms > 0 else 0.0
        traffic = None
        tpath = os.path.join(ROOT, "profiles", "traffic_%s.json" % args.config)
        if os.path.exists(tpath):
            try:
                tj = json.load(open(tpath))
                if tj.get("frames_per_launch") == F and bool(tj.get("coef_writeback")) == (not args.no_coef):
                    traffic = tj.get("hbm_bytes_per_launch")
            except Exception:
                traffic = None
        result = {
            "metric": "Mpix/s encode (4K yuv420p10, slices=8) at 1/2/4/8 GPUs; bit-exact vs CPU",
            "value": round(mpix, 1),
            "unit": "Mpix/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "preroll_launches": preroll,
            "ms_per_step": round(elapsed / args.steps * 1e3, 4),
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "int32",
            "data": "synthetic",
            "config": {"workload": "%dx%d %s (4:4:4 restatement of BASELINE config %s), qp=0, "
                                   "%d frames in flight per step per GPU, frames resident in HBM"
                                   % (W, H, fmt, args.config[1], F),
                       "frames_per_step_per_gpu": F,
                       "coef_writeback": not args.no_coef,
                       "pipelined_estage": args.pipeline,
                       "parallelism": "frame-parallel x%d, no data-path collective" % world,
                       "packet_bytes_frame0": len(packets[0])},
            "roofline": {"bound": "hbm", "kernel": enc.tstage_kernel_name(F),
                         "achieved": round(achieved, 1), "peak": HBM_PEAK_GBS, "unit": "GB/s",
                         "frac": round(achieved / HBM_PEAK_GBS, 4),
                         "traffic": traffic,
                         "algorithmic_bytes_per_launch": alg_bytes,
                         "kernel_ms_avg": round(t_kernel_ms, 4),
                         "estage_ms_avg": round(e_ms / max(launches, 1), 4),
                         "kernel_ms_min": round(t_lo, 4), "kernel_ms_max": round(t_hi, 4),
                         "launches_timed": launches,
                         "timed_every_nth_step": period},
        }
        if steady is not None:
            result["steady_state"] = steady
            if steady["launches_timed"]:
                result["steady_state"]["roofline_frac"] = round(
                    alg_bytes / (steady["kernel_ms_mean"] * 1e-3) / 1e9 / HBM_PEAK_GBS, 4)
        if hb is not None:
            result["host_boundary"] = hb
        if world == 1 and not args.no_cpu_baseline:
            # CPU baseline: the oracle (C restatement, byte-identical to the reference on
            # its known answers) on a bounded sample of the SAME frames, one host core.
            from tests import oracle_lib
            oracle = oracle_lib.load()
            n_done, tcpu, ok = 0, 0.0, True
            while n_done < F and tcpu < args.cpu_seconds:
                c0 = time.perf_counter()
                ref = oracle.encode(host_frames[n_done], fmt)
                tcpu += time.perf_counter() - c0
                ok = ok and (ref == packets[n_done])
                n_done += 1
            result["cpu_baseline"] = {
                "value": round(n_done * W * H / tcpu / 1e6, 2), "unit": "Mpix/s", "cores": 1, "kind": "port",
                "sample": "%d of the %d benchmark frames (%dx%d %s), oracle/libffv2_oracle.so, gcc -O2, 1 thread"
                          % (n_done, F, W, H, fmt),
                "host_cores_available": os.cpu_count(),
                "gpu_packets_match_cpu": bool(ok)}
            # informational: frames are independent, so the CPU scales by running one encoder per
            # core; one frame per thread on this box's CPU share, a few seconds
            from concurrent.futures import ThreadPoolExecutor
            nthr = max(1, min(16, os.cpu_count() or 1))
            c0 = time.perf_counter()
            with ThreadPoolExecutor(nthr) as pool:       # ctypes releases the GIL inside the oracle
                list(pool.map(lambda i: oracle.encode(host_frames[i % F], fmt), range(nthr)))
            tall = time.perf_counter() - c0
            result["cpu_baseline"]["all_cores"] = {"value": round(nthr * W * H / tall / 1e6, 1), "unit": "Mpix/s",
                                                   "cores": nthr, "sample": "one 4K frame per thread" if W == 3840
                                                   else "one frame per thread"}
            if hb420_first is not None and "yuv420" in (hb or {}):
                pk0, (y0, u0, v0) = hb420_first
                ok420 = oracle.encode(oracle.sws_420_to_444(y0, u0, v0, depth), fmt) == pk0
                hb["yuv420"]["packet0_matches_oracle_convert_then_encode"] = bool(ok420)
                ok = ok and ok420
            if not ok:
                result["error"] = "GPU packets differ from the CPU oracle"
        print(json.dumps(result))
    enc.close()
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    if result is not None and result.get("error"):
        sys.exit(1)


if __name__ == "__main__":
    main()
