"""Counterpart of the reference's app_bm/groq_script.py:113-151: ONE CGConv layer (channels 11, no edge features) in fp16 on a
QM9-sized graph (x [29, 11], 56 edges), 10 warm-up calls outside torch.no_grad(), then 300 calls each bracketed by two device
events and a synchronize; prints the mean in milliseconds, as the reference does (:151).

The reference defines the layer in the script (a copy of torch_geometric's CGConv on MessagePassing, :15-111); here it is
gnnops.conv.CGConv — same constructor, parameter names and result, one dense product + one fused edge pass
(csrc/conv.hip). Differences forced by the environment: `torch.rand(2, 56).to(torch.long)` (:119) is all zeros — every edge
is 0 -> 0 — which is kept under --reference-edges, while the default draws 56 random edges over the 29 nodes so that the
layer does some work. `--debug` also prints the per-call times the reference's groq_script_debug.py wraps in `profileit`."""
import argparse
import os
import sys

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--reference-edges", action="store_true", help="edge_index = torch.rand(2, 56).to(torch.long), i.e. all zeros")
    ap.add_argument("--repetitions", type=int, default=300)
    ap.add_argument("--debug", action="store_true")
    ap.add_argument("--graph", action="store_true", help="capture the call into a HIP graph once and time replays (the script calls "
                    "the layer on the SAME tensors 300 times: launch-bound work a graph replays as one launch)")
    args = ap.parse_args()
    if not torch.cuda.is_available():
        raise Exception("Benchmarking only supported for CUDA")
    from gnnops.conv import CGConv

    DEVICE = torch.device("cuda:0")
    torch.manual_seed(0)
    x_qm9 = torch.rand(29, 11).to(torch.float16)
    edge_index_qm9 = torch.rand(2, 56).to(torch.long) if args.reference_edges else torch.randint(0, 29, (2, 56))
    x, edge_index = x_qm9.to(DEVICE), edge_index_qm9.to(DEVICE)
    print(f"Num features: {x.size(1)}")
    model = CGConv(x.size(1), 0).to(DEVICE).to(torch.float16)

    starter, ender = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    timings = np.zeros((args.repetitions, 1))
    for _ in range(10):                      # GPU warm-up, outside no_grad as in the reference (:135-137)
        _ = model(x, edge_index)
    call = lambda: model(x, edge_index)   # noqa: E731
    if args.graph:
        import gnnops

        gnnops.set_plan_cache(False)          # the plan becomes device work inside the graph: a replay may see a new edge list
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side), torch.no_grad():
            call()
        torch.cuda.current_stream().wait_stream(side)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph), torch.no_grad():
            out = call()
        eager = call()
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(out, eager)
        call = graph.replay
    with torch.no_grad():
        for rep in range(args.repetitions):
            starter.record()
            _ = call()
            ender.record()
            torch.cuda.synchronize()
            timings[rep] = starter.elapsed_time(ender)
    print(np.sum(timings) / args.repetitions)
    if args.debug:
        print(f"std {np.std(timings):.5f} ms, median {np.median(timings):.5f} ms, min {timings.min():.5f} ms "
              f"(reference, A100, apps_bm_data/model_data_fp16_no_batching.txt: 0.298 ms)")


if __name__ == "__main__":
    main()
