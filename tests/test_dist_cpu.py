"""CPU tier: the N > 1 path - frame sharding, shared-payload bit offsets and the gather of packed
bit streams - with two gloo ranks.  The per-rank compute is done by the oracle here (there is no GPU
in this tier); what is under test is the rank logic of svsdct/dist.py, which bench.py and a
multi-GPU pipeline run unchanged over RCCL."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from oracle import qim_dct_oracle as orc
from svsdct import batch, synth


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, shape, n_ac, delta, n_frames, out_dir):
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from svsdct import dist as sdist
    h, w = shape
    first, count = sdist.shard(n_frames)
    cap = batch.capacity_bits(1, h, w, n_ac)
    payload = synth.synthetic_bits(n_frames * cap, seed=3)                 # the shared stream
    frames = synth.synthetic_frames(count, h, w, seed=3, first_frame=first)  # this rank's frames
    off = sdist.payload_bit_offset(first, h, w, n_ac)
    if count:
        stego, used = orc.batch_embed(frames, delta, payload[off:off + count * cap], n_ac)
        assert used == count * cap
        bits = orc.batch_extract_bits(stego, delta, n_ac)
    else:                                                                  # more ranks than frames: an empty shard
        bits = np.zeros(0, np.uint8)
    packed = torch.from_numpy(np.packbits(bits)) if bits.size else torch.zeros(0, dtype=torch.uint8)
    got, total = sdist.gather_stream(packed, int(bits.size), dst=0)
    assert total == n_frames * cap
    if rank == 0:
        np.save(os.path.join(out_dir, "stream.npy"), np.unpackbits(got, count=total))
    else:
        assert got is None
    dist.barrier()
    dist.destroy_process_group()


def _run(tmp_path, shape, n_ac, n_frames, world=2):
    mp.spawn(_worker, args=(world, _free_port(), shape, n_ac, 8, n_frames, str(tmp_path)), nprocs=world, join=True)
    h, w = shape
    cap = batch.capacity_bits(1, h, w, n_ac)
    want = synth.synthetic_bits(n_frames * cap, seed=3)
    got = np.load(os.path.join(str(tmp_path), "stream.npy"))
    assert np.array_equal(got, want)          # delta = 8: the global stream comes back error-free, in order
    # and equals what one rank would have extracted from the whole clip
    frames = synth.synthetic_frames(n_frames, h, w, seed=3)
    stego, _ = orc.batch_embed(frames, 8, want, n_ac)
    assert np.array_equal(got, orc.batch_extract_bits(stego, 8, n_ac))


def test_two_ranks_byte_aligned_streams(tmp_path):
    _run(tmp_path, (32, 64), 4, 6)            # 32 blocks * 4 = 128 bits per frame


def test_two_ranks_uneven_shards_and_bit_granular_join(tmp_path):
    _run(tmp_path, (24, 40), 3, 5)            # 15 blocks * 3 = 45 bits per frame; 3 + 2 frames


def test_three_ranks_with_an_empty_shard_free_tail(tmp_path):
    _run(tmp_path, (16, 24), 7, 7, world=3)   # 6 blocks * 7 = 42 bits per frame; shards of 3 + 2 + 2 frames


def test_more_ranks_than_frames(tmp_path):
    _run(tmp_path, (16, 16), 5, 2, world=3)   # rank 2 owns no frame and contributes zero bits
