"""The ops are capturable into HIP graphs (torch.cuda.CUDAGraph): nothing behind the C ABI allocates or synchronises, and
every launch goes to the stream the caller hands over. A launch-bound step — a small one-shot scatter is nine kernels —
is replayed as one graph launch; the replay must see new input values and reproduce the eager result bit for bit."""
import pytest
import torch

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def gnnops():
    import gnnops as g

    g.load_library()
    return g


def test_scatter_index_select_addmm_in_a_graph(gnnops):
    N, E, D = 3000, 20000, 64
    gen = torch.Generator(device="cuda").manual_seed(0)
    src = torch.rand(E, D, generator=gen, device="cuda")
    idx = torch.randint(0, N, (E,), generator=gen, device="cuda")
    w = torch.rand(D, D, generator=gen, device="cuda").to(torch.bfloat16)
    gnnops.set_plan_cache(False)   # one-shot forms: the whole op is device work on the current stream
    try:
        def step():
            agg = gnnops.scatter_add(src, idx, 0, dim_size=N)           # partition + bucketed reduce
            mn, arg = gnnops.scatter_min(src, idx, 0, dim_size=N)
            sel = gnnops.index_select(agg, 0, idx)
            y = gnnops.addmm(agg.to(torch.bfloat16), agg.to(torch.bfloat16), w)
            return agg, mn, arg, sel, y

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                                  # warm-up outside capture (one-time attributes)
            step()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = step()
        for trial in range(2):
            src.copy_(torch.rand(E, D, generator=gen, device="cuda"))   # new values in the captured buffers
            idx.copy_(torch.randint(0, N, (E,), generator=gen, device="cuda"))
            graph.replay()
            torch.cuda.synchronize()
            eager = step()
            for got, exp in zip(outs, eager):
                assert torch.equal(got, exp), trial
    finally:
        gnnops.set_plan_cache(True)


def test_split_k_addmm_in_a_graph(gnnops):
    """289 tiles of 256 x 256: the persistent kernel with a split last round (csrc/gemm.hip gemm_sk256_kernel). Its flag words
    are cleared by a kernel node of the same capture (a memset node was not reliably seen by the pollers: this test failed on
    the second replay), the partner hand-off is device work: replays see new operands and reproduce the eager result bit for
    bit — a partner's partial tile from the previous replay would be off by whole units; K = 300 adds the side copies of the
    last K-tile."""
    M = N = 4352
    K = 300
    gen = torch.Generator(device="cuda").manual_seed(1)
    a = (torch.rand(M, K, generator=gen, device="cuda") - 0.5).half()
    b = (torch.rand(K, N, generator=gen, device="cuda") - 0.5).half()
    c = (torch.rand(M, N, generator=gen, device="cuda") - 0.5).half()
    assert gnnops._lib.load().gnnops_addmm_workspace_bytes(M, N, K) >= 256 * (256 << 10)
    side = torch.cuda.Stream()
    side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        gnnops.addmm(c, a, b)
    torch.cuda.current_stream().wait_stream(side)
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        out = gnnops.addmm(c, a, b)
    for trial in range(10):
        a.copy_((torch.rand(M, K, generator=gen, device="cuda") - 0.5).half())
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, gnnops.addmm(c, a, b)), trial


def test_message_passing_layer_in_a_graph(gnnops):
    """A whole layer (gnnops.conv: dense product + plan + fused edge pass) is launch-bound on a batch of small graphs
    (app_bm/benchmark_convs.py). With the plan cache OFF everything the layer does is device work on the current stream — the
    one-launch plan included — so the call is capturable: new features AND a new edge list in the captured buffers are seen
    by the replay, bit for bit the eager result."""
    from gnnops import conv

    torch.manual_seed(0)
    n, e = 500, 2400
    gen = torch.Generator(device="cuda").manual_seed(1)
    x = torch.rand(n, 11, generator=gen, device="cuda").half()
    ei = torch.randint(0, n, (2, e), generator=gen, device="cuda")
    layers = [conv.CGConv(11, 0).half().cuda(), conv.GINConv(torch.nn.Linear(11, 64)).half().cuda(), conv.SAGEConv(11, 64).half().cuda()]
    gnnops.set_plan_cache(False)
    try:
        def step():
            with torch.no_grad():
                return [layer(x, ei) for layer in layers]

        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            step()                                                      # warm-up: packed weights, one-time kernel attributes
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            outs = step()
        for trial in range(2):
            x.copy_(torch.rand(n, 11, generator=gen, device="cuda").half())
            ei.copy_(torch.randint(0, n, (2, e), generator=gen, device="cuda"))
            graph.replay()
            torch.cuda.synchronize()
            for got, exp in zip(outs, step()):
                assert torch.equal(got, exp), trial
    finally:
        gnnops.set_plan_cache(True)
