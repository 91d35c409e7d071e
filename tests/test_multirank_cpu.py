"""-m "not gpu": the N>1 path on CPU with gloo, world size 2.  The per-shard encoder
here is the ORACLE (tests may use it); what is under test is the product's host logic for sharding and
stitching -- hvlc.shard_bounds / hvlc.stitch_base and the single all-gather of one uint64 per rank that
bench.py performs over RCCL -- against the single-stream oracle."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import hvlc
import oracle_lib as O


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, n_words, q):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    table = np.fromfile(os.path.join(g, "g1_table.bin"), dtype=np.uint32)
    cw, cl = table[:256], table[256:]
    period = np.fromfile(os.path.join(g, "g1_period.bin"), dtype=np.uint8)
    x = np.tile(period, 4).view(np.uint32)[:n_words]
    w0, w1 = hvlc.shard_bounds(n_words, world, rank)
    shard = x[w0:w1]
    words, bits = O.encode(shard, cw, cl)                       # stands in for hvlc_encode on this rank's GPU
    _, offs, _ = O.segment_offsets(shard, cl)
    total = torch.tensor([bits], dtype=torch.int64)
    gathered = [torch.zeros(1, dtype=torch.int64) for _ in range(world)]
    dist.all_gather(gathered, total)                            # the ONE exchange of the path: 8 bytes per rank
    totals = [int(t.item()) for t in gathered]
    base = hvlc.stitch_base(totals, rank)
    q.put((rank, w0, w1, base, bits, (offs + np.uint64(base)).tolist(), words.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(n_words, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, n_words, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


def test_two_rank_stitch_matches_single_stream(golden):
    n_words = 100_000 + 77                 # ragged: last segment partial, odd number of segments
    res = _run(n_words)
    x = np.tile(golden["period"], 4).view(np.uint32)[:n_words]
    ref_words, ref_bits = O.encode(x, golden["cw"], golden["cl"])
    _, ref_offs, _ = O.segment_offsets(x, golden["cl"])
    # shards tile the input at segment multiples
    assert res[0][1] == 0 and res[0][2] == res[1][1] and res[1][2] == n_words and res[0][2] % 256 == 0
    # stitched offsets are exactly the single-stream offsets
    offs = np.array(res[0][5] + res[1][5], dtype=np.uint64)
    assert np.array_equal(offs, ref_offs)
    assert res[0][4] + res[1][4] == ref_bits and res[1][3] == res[0][4]
    # bit-concatenation of the shard streams is the single stream
    from util import extract_bits
    w0 = np.array(res[0][6], dtype=np.uint32)
    w1 = np.array(res[1][6], dtype=np.uint32)
    b0, b1 = res[0][4], res[1][4]
    assert np.array_equal(extract_bits(ref_words, 0, b0), extract_bits(w0, 0, b0))
    assert np.array_equal(extract_bits(ref_words, b0, b1), extract_bits(w1, 0, b1))


def test_shard_bounds_properties():
    for total in (0, 1, 255, 256, 257, 4096 * 7 + 3, 1 << 20):
        for world in (1, 2, 3, 8):
            cuts = [hvlc.shard_bounds(total, world, r) for r in range(world)]
            assert cuts[0][0] == 0 and cuts[-1][1] == total
            for (a0, a1), (b0, b1) in zip(cuts, cuts[1:]):
                assert a1 == b0 and a0 <= a1
            for a0, a1 in cuts[:-1]:
                assert a1 % 256 == 0 or a1 == total
            sizes = [(a1 - a0 + 255) // 256 for a0, a1 in cuts]
            assert max(sizes) - min(sizes) <= 1
    assert hvlc.stitch_base([5, 7, 11], 0) == 0 and hvlc.stitch_base([5, 7, 11], 2) == 12
