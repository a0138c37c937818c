"""N > 1 path: world_size-2 gloo run of the frame fan-out / in-order packet gather.
Without a GPU (-m "not gpu") the per-rank encoder is the oracle (test-only) -- what is under
test there is the sharding and the ordered gather, which are device independent.  With a GPU
(-m gpu) the two ranks share cuda:0 and encode with the real FFV2Encoder through the
asynchronous frame ring, host frames in, host packets out."""
import os
import socket

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from ffmpeg_ffv2_amd import fanout, frames as synth

W, H, FMT, P, DEPTH = 130, 70, "yuv444p", 3, 8


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    p = s.getsockname()[1]
    s.close()
    return p


def _worker(rank, world, port, nframes, q, use_gpu=False):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    seen = []
    if use_gpu:
        from ffmpeg_ffv2_amd import FFV2Encoder
        enc = FFV2Encoder(W, H, FMT, device=0, max_batch=1)       # both ranks on cuda:0
        enc.ring_open(2)

        def encode_batch(ns):
            seen.extend(ns)
            out, sent = [], 0
            while len(out) < len(ns):
                while sent < len(ns) and enc.ring_send(synth.make("S2", ns[sent], P, H, W, DEPTH), tag=ns[sent]):
                    sent += 1
                tag, pk = enc.ring_receive()
                assert tag == ns[len(out)]
                out.append(pk)
            return out
    else:
        from tests import oracle_lib
        oracle = oracle_lib.load()

        def encode_batch(ns):
            seen.extend(ns)
            return [oracle.encode(synth.make("S2", n, P, H, W, DEPTH), FMT) for n in ns]

    out = fanout.encode_sequence(encode_batch, nframes, rank, world, batch=2)
    assert seen == fanout.local_frames(nframes, rank, world)
    if rank == 0:
        q.put(out)
    else:
        assert out is None
    dist.barrier()
    dist.destroy_process_group()


def _run(world, nframes, use_gpu=False):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, nframes, q, use_gpu)) for r in range(world)]
    for p in procs:
        p.start()
    out = q.get(timeout=120)
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return out


def test_two_ranks_ordered_gather(oracle):
    nframes = 5                                   # uneven: rank 0 gets 3 frames, rank 1 gets 2
    out = _run(2, nframes)
    assert len(out) == nframes
    for n in range(nframes):
        assert out[n] == oracle.encode(synth.make("S2", n, P, H, W, DEPTH), FMT), "frame %d" % n


@pytest.mark.gpu
def test_two_ranks_real_encoder_shared_gpu(oracle):
    nframes = 7
    out = _run(2, nframes, use_gpu=True)
    assert len(out) == nframes
    for n in range(nframes):
        assert out[n] == oracle.encode(synth.make("S2", n, P, H, W, DEPTH), FMT), "frame %d" % n


def test_sharding_is_a_partition():
    for world in (1, 2, 3, 8):
        for nframes in (0, 1, 7, 16, 33):
            seen = sorted(n for r in range(world) for n in fanout.local_frames(nframes, r, world))
            assert seen == list(range(nframes))
            assert all(fanout.owner(n, world) == r for r in range(world) for n in fanout.local_frames(nframes, r, world))


def test_single_rank_needs_no_process_group(oracle):
    out = fanout.encode_sequence(lambda ns: [bytes([n]) * (n + 1) for n in ns], 4, 0, 1, batch=3)
    assert out == [bytes([n]) * (n + 1) for n in range(4)]
