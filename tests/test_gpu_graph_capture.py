"""-m gpu: the hot-path entry points are HIP-graph capturable.  One step of the batch pipeline -- dabgpu_ofdm_sync_demod_frames (PRS
synchronisation -> positioned demodulation -> fine-frequency update) and dabgpu_decode_frames_layout (FIC + MSC of the frame) -- is
captured from a stream into a graph and replayed; what the replays leave in the output buffers and in the receivers' sync records must equal
what the same calls leave when they are issued one by one.  A steady-state call consists of kernel launches only: the small tables that
depend on the sub-channel list alone are uploaded by the FIRST call with that list and found unchanged afterwards
(dabgpu_stage_h2d_cached); a call that would have to upload during a capture refuses instead of invalidating it."""
import os
import sys

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def build(E, n_distinct, seed):
    import torch
    import dabgpu
    import bench
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        p = bench.Pipeline(dabgpu.Context(0), dabgpu, torch, dev, E, n_distinct, seed=seed, inflight=1, synced=True)
    torch.cuda.synchronize()
    return p, side


def snapshot(p):
    import torch
    torch.cuda.synchronize()
    return [t.clone() for t in (p.fic_out[0], p.fic_res[0], p.msc_out[0], p.msc_res[0], p.states.view(torch.uint8), p.hist)]


def test_pipeline_step_captured_in_a_graph_equals_eager_steps():
    import math
    import torch
    import dabgpu
    E = 24
    eager, _ = build(E, 8, seed=11)
    graph, side = build(E, 8, seed=11)                    # same seeds: same multiplexes, carrier and timing offsets
    for p in (eager, graph):
        p.tune()
        p.fill()
    cycle = graph.H * graph.mux.n_frames // math.gcd(graph.H, graph.mux.n_frames)    # ring slot and stored frame repeat with this period
    while graph.j % cycle:
        graph.step(); eager.step()
    assert eager.j == graph.j
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        for _ in range(cycle):
            graph.step()
    torch.cuda.synchronize()
    # (capturing does not execute: the state is what it was before)
    for _ in range(2):
        g.replay()
        for _ in range(cycle):
            eager.step()
        a, b = snapshot(graph), snapshot(eager)
        for x, y in zip(a, b):
            assert torch.equal(x, y)
    graph.j = eager.j
    chk = graph.check(dabgpu)
    assert chk["fib_bytes_equal_transmitted"] and chk["msc_bytes_equal_transmitted"] and chk["fib_crc_pass"] == chk["fib_crc_expected"]


def test_first_call_with_new_sub_channels_refuses_to_run_inside_a_capture():
    import torch
    import dabgpu
    p, side = build(8, 4, seed=12)
    p.tune()
    p.fill()
    torch.cuda.synchronize()
    other = list(p.subs[:-1])                              # a sub-channel list this context has not decoded yet: its plans must be uploaded
    g = torch.cuda.CUDAGraph()
    with pytest.raises(Exception) as err:
        with torch.cuda.graph(g, stream=side):
            p.ctxs[0].msc_decode_frames(p.hist, p.E, p.stride, p.H, 0, other, p.msc_out[0], 4 * p.n_sub * 192, p.msc_res[0],
                                        stream=side.cuda_stream, bits_layout=p.layout)
    assert "before capturing" in str(err.value) or "capture" in str(err.value).lower()
    torch.cuda.synchronize()
    # outside a capture the same call is fine, and afterwards it is capturable
    p.ctxs[0].msc_decode_frames(p.hist, p.E, p.stride, p.H, 0, other, p.msc_out[0], 4 * p.n_sub * 192, p.msc_res[0],
                                stream=side.cuda_stream, bits_layout=p.layout)
    torch.cuda.synchronize()
    g2 = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g2, stream=side):
        p.ctxs[0].msc_decode_frames(p.hist, p.E, p.stride, p.H, 0, other, p.msc_out[0], 4 * p.n_sub * 192, p.msc_res[0],
                                    stream=side.cuda_stream, bits_layout=p.layout)
    g2.replay()
    torch.cuda.synchronize()


def test_graph_replays_after_a_larger_eager_call_and_growth_is_refused_inside_a_capture():
    """A graph captured with 8 ensembles, then an eager call with 40 on the SAME context (its scratch slots grow): the outgrown buffers are
    parked, not freed, so the replay still runs on memory the context owns and -- the sub-channel list being the same -- gives the eager
    result.  And a call whose shapes need more scratch than the context holds refuses to run inside a capture."""
    import torch
    import dabgpu
    small, side = build(8, 4, seed=13)
    small.tune()
    small.fill()
    torch.cuda.synchronize()
    ctx = small.ctxs[0]

    def decode(p, k=0):
        ctx.decode_frames(p.hist, p.E, p.stride, p.H, 0, p.subs, p.fic_out[k], p.fic_res[k], p.msc_out[k], 4 * p.cif_bytes, p.msc_res[k],
                          stream=side.cuda_stream, bits_layout=p.layout)
    decode(small)
    torch.cuda.synchronize()
    want = [t.clone() for t in (small.fic_out[0], small.fic_res[0], small.msc_out[0], small.msc_res[0])]
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        decode(small)
    torch.cuda.synchronize()
    # a larger batch on the same context, inside a capture first: refused (it would have to grow the scratch) ...
    with torch.cuda.stream(side):
        big = bench_pipeline_like(small, 40, seed=14)
    torch.cuda.synchronize()
    g_big = torch.cuda.CUDAGraph()
    with pytest.raises(Exception) as err:
        with torch.cuda.graph(g_big, stream=side):
            decode(big)
    assert "before capturing" in str(err.value) or "capture" in str(err.value).lower()
    torch.cuda.synchronize()
    # ... then eagerly: fine, the slots grow
    decode(big)
    torch.cuda.synchronize()
    for t in (small.fic_out[0], small.msc_out[0]):
        t.zero_()
    g.replay()
    torch.cuda.synchronize()
    got = [small.fic_out[0], small.fic_res[0], small.msc_out[0], small.msc_res[0]]
    for x, y in zip(got, want):
        assert torch.equal(x, y)


def bench_pipeline_like(p, E, seed):
    """a second Pipeline's buffers (history ring filled by its own demodulator context) decoded through p's context"""
    import torch
    import dabgpu
    import bench
    q = bench.Pipeline(dabgpu.Context(0), dabgpu, torch, torch.device("cuda", 0), E, 4, seed=seed, inflight=1, synced=True)
    q.tune()
    q.fill()
    return q
