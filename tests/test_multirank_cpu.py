"""-m "not gpu": the N>1 path on CPU with gloo, world sizes 2 and 4.  The per-shard generator and
encoder here are the ORACLE (tests may use it); what is under test is the product's multi-rank host code, the SAME
functions bench.py runs at N > 1 -- hvlc.shard_first_draw and hvlc.stitch (one all-gather of one uint64 per rank +
the rebase) -- against one generation and one encoding of the whole job stream.  tests/test_encode_gpu.py
(test_two_ranks_on_one_gpu_shared_stitch_code) drives the same functions with the real kernels on one device."""
import os
import socket

import numpy as np
import pytest
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


def _worker(rank, world, port, shard_bytes, q):
    """One rank of the job, running the product's own multi-rank host code on CPU tensors over gloo:
    hvlc.shard_first_draw (which part of the synthetic job stream this rank generates -- what bench.py passes to
    hvlc_gen_symbols) and hvlc.stitch (the all-gather of the totals + the rebase; bench.py passes
    dist.all_gather_into_tensor over RCCL and hvlc_rebase_offsets_dev).  The ORACLE stands in for the GPU kernels
    (generator, encoder) -- tests may use it."""
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import sys
    sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
    from util import dist_weights
    g = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    period = np.fromfile(os.path.join(g, "g1_period.bin"), dtype=np.uint8)
    cdf = hvlc.cdf32_from_counts(dist_weights("H2.2", period))
    shard = O.gen_symbols(shard_bytes, 0x5EED0005, cdf, first_draw=hvlc.shard_first_draw(rank, shard_bytes))
    hist = torch.from_numpy(O.histogram(shard).astype(np.int64))
    dist.all_reduce(hist)                                        # table of the WHOLE job's histogram, as bench.py does
    cw, cl = hvlc.build_table(hist.numpy().astype(np.uint64))     # libhvlc's own builder: what bench.py and hvlc_mgpu.c call
    x = shard.view(np.uint32)
    words, bits = O.encode(x, cw, cl)                            # stands in for hvlc_encode on this rank's GPU
    _, offs, _ = O.segment_offsets(x, cl)
    offs = offs.copy()
    total = torch.tensor([bits], dtype=torch.int64)
    gathered = torch.zeros(world, dtype=torch.int64)

    def rebase(gth, r):                                          # stands in for hvlc_rebase_offsets_dev
        offs[:] += np.uint64(int(gth[:r].sum().item()))

    hvlc.stitch(total, gathered, rank, world, dist.all_gather_into_tensor, rebase)   # the ONE exchange: 8 bytes per rank
    totals = [int(t) for t in gathered.tolist()]
    q.put((rank, totals, bits, offs.tolist(), words.tolist(), cw.tolist(), cl.tolist()))
    dist.barrier()
    dist.destroy_process_group()


def _run(shard_bytes, world=2):
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, world, port, shard_bytes, q)) for r in range(world)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=120) for _ in range(world))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    return res


@pytest.mark.parametrize("world", [2, 4])
def test_multi_rank_job_matches_single_stream(golden, world):
    shard_bytes = 300 * 1024 if world == 2 else 157 * 1024
    res = _run(shard_bytes, world)
    from util import dist_weights, extract_bits
    cdf = hvlc.cdf32_from_counts(dist_weights("H2.2", golden["period"]))
    whole = O.gen_symbols(world * shard_bytes, 0x5EED0005, cdf)            # ONE generation from draw 0
    cw, cl = np.array(res[0][5], dtype=np.uint32), np.array(res[0][6], dtype=np.uint32)
    assert all(r[5] == res[0][5] and r[6] == res[0][6] for r in res)       # same table on every rank
    cw1, cl1 = hvlc.build_table(O.histogram(whole))
    assert np.array_equal(cw, cw1) and np.array_equal(cl, cl1)             # ... and it is the whole job's table
    cw2, cl2 = hvlc.huffman_table(O.histogram(whole))                      # the Python twin (tools/, older tests) agrees
    assert np.array_equal(cw, cw2) and np.array_equal(cl, cl2)
    x = whole.view(np.uint32)
    ref_words, ref_bits = O.encode(x, cw, cl)
    _, ref_offs, _ = O.segment_offsets(x, cl)
    totals = res[0][1]
    assert all(r[1] == totals for r in res) and totals == [r[2] for r in res] and sum(totals) == ref_bits
    # stitched offsets are exactly the single-stream offsets
    offs = np.array(sum((r[3] for r in res), []), dtype=np.uint64)
    assert np.array_equal(offs, ref_offs)
    # bit-concatenation of the shard streams is the single stream
    base = 0
    for r in res:
        w = np.array(r[4], dtype=np.uint32)
        assert np.array_equal(extract_bits(ref_words, base, r[2]), extract_bits(w, 0, r[2]))
        base += r[2]


def test_stitch_is_a_no_op_for_one_rank():
    total = torch.tensor([5], dtype=torch.int64)
    gathered = torch.zeros(1, dtype=torch.int64)
    hvlc.stitch(total, gathered, 0, 1, None, lambda g, r: (_ for _ in ()).throw(AssertionError("no rebase at world 1")))
    assert hvlc.shard_first_draw(0, 4096) == 0 and hvlc.shard_first_draw(3, 4096) == 3 * 2048


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


@pytest.mark.parametrize("name", ["H2.2", "H4.0", "H1.0"])
def test_the_two_table_builders_agree_on_the_job_histograms(golden, name):
    """bench.py and hvlc_mgpu.c build the job's table with hvlc_build_table (C, in libhvlc.so); hvlc.huffman_table is its
    Python twin.  On the histograms of the three benchmark distributions (a few MiB of each) they give the same table."""
    from util import dist_weights, SEEDS
    cdf = hvlc.cdf32_from_counts(dist_weights(name, golden["period"]))
    hist = O.histogram(O.gen_symbols(3 << 20, SEEDS[name], cdf))
    a, b = hvlc.build_table(hist), hvlc.huffman_table(hist)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
