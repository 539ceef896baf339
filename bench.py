#!/usr/bin/env python3
"""bench.py — the reference's headline measurement on MI355X: per-op wall time and effective HBM GB/s
(BASELINE.json `metric`) for the scatter/gather hot path.

One "step" = one cold call of the hot path on one batch of synthetic input:
    torch_scatter.scatter_add(src, index, dim=0, dim_size=N)    (reference: op_bm_scripts/benchmark_scatter_add.py:15-19)
with a 1-D row index — index partition + bucketed segment reduce, nothing cached.
Workload at N=1 is BASELINE config 2 (configs[1]): N=10M destinations, E=50M source rows, D=128 fp32.
With --gpus G the workload is BASELINE config 5's per-GPU share (configs[4]: N=80M, E=800M at 8 GPUs): every rank holds
E=100M rows with GLOBAL destination ids over G*10M rows and owns a slab of 10M of them (weak scaling);
the step is gnnops.dist.sharded_scatter — ONE sparse reduce-scatter (all-to-all-v of the edges whose destination another
rank owns) overlapped with the reduction of the own slab, one host read-back per step. `python bench.py --gpus G` without
a launcher starts its G ranks itself (torch.distributed.run children, started before this process touches a GPU).

`value` = algorithmic bytes (E*D*s + E*8 + N*D*s per step, x ranks) / wall time, inputs
already resident in HBM. The JSON line also carries `roofline` (dominant kernel = the segment-reduce
launch, timed with events on the launch stream), `cpu_baseline` (the C oracle port on a bounded sample,
host cores of this box) — both at every N, measured on rank 0 — and per-op numbers for the other config-2 ops.

Reading a 1 -> 8 curve: the N=1 headline is config 2 (E=50M per GPU), the N>1 headline config 5's share (E=100M per GPU).
So that equal per-GPU work can be compared, the N=1 line carries `same_work.c5_share_1gpu` (E=100M, N=10M on one GPU, same
cold step) and every N>1 line `same_work.c2_share` (E=50M per GPU through the same sharded step).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT):
    if _p not in sys.path:
        sys.path.insert(0, _p)

HBM_PEAK_GBS = 8000.0  # MI355X spec (MI355X_MICROARCH.md); 6290 GB/s is the measured copy ceiling

WORKLOADS = {
    # name: (N per GPU, E per GPU, D, dtype)
    "c2": (10_000_000, 50_000_000, 128, "f32"),     # BASELINE configs[1]
    "c5": (10_000_000, 100_000_000, 128, "f32"),    # BASELINE configs[4] per-GPU share (N=80M,E=800M at 8 GPUs)
    "c1": (100_000, 500_000, 64, "f32"),            # BASELINE configs[0] shape (plumbing)
    "tiny": (20_000, 100_000, 128, "f32"),
}


def algorithmic_bytes(op, N, E, D, s=4):
    """SURVEY.md §8(d) per-call figures."""
    if op in ("scatter_add", "scatter_mean"):
        return E * D * s + E * 8 + N * D * s
    if op in ("scatter_min", "scatter_max"):
        return E * D * s + E * 8 + N * D * s + N * D * 8
    if op == "index_select":
        return N * D * s + E * 8 + E * D * s
    if op == "index_add_":
        return E * D * s + E * 8 + 2 * N * D * s
    raise KeyError(op)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--workload", default=None, choices=sorted(WORKLOADS),
                    help="default: c2 (BASELINE configs[1]) on one GPU, c5 (configs[4]'s per-GPU share) on several")
    ap.add_argument("--cut", type=float, default=None,
                    help="N>1 only: fraction of each rank's edges whose destination another rank owns (edge cut of the "
                         "partition; the rest fall uniformly in the rank's own rows). Default: 1/70 per peer, i.e. "
                         "(G-1)/70 — every pair of partitions shares the same boundary, 10 %% in total at G=8. "
                         "(G-1)/G = an unpartitioned uniform random graph")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-extra-ops", action="store_true", help="skip the per-op table (scatter_min/max/mean, index_select, index_add_)")
    args = ap.parse_args()
    if args.workload is None:
        args.workload = "c2" if args.gpus == 1 else "c5"

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        # not under a launcher: start the ranks ourselves, as fresh child processes, BEFORE anything here touches a GPU
        # (torch is not even imported yet), relay their output (rank 0 prints the JSON line) and exit with their code
        import socket
        import subprocess

        with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sock:
            sock.bind(("127.0.0.1", 0))
            port = sock.getsockname()[1]
        cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={args.gpus}",
               "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
        raise SystemExit(subprocess.call(cmd))

    import torch

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit(f"--gpus {args.gpus} but the launcher started {world} ranks (WORLD_SIZE={world})")
    # GNNOPS_BENCH_GLOO_ONE_GPU=1: rehearsal of the N>1 code path on a one-GPU box — every rank on cuda:0, gloo carrying
    # the device tensors (RCCL refuses two ranks on one device). Its timings mean nothing; its JSON line says so.
    rehearsal = os.environ.get("GNNOPS_BENCH_GLOO_ONE_GPU") == "1"
    if rehearsal:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    import gnnops

    lib = gnnops.load_library()  # fails loudly if the HIP extension is missing
    gnnops.set_plan_cache(False)

    dist = None
    force_dist = os.environ.get("GNNOPS_BENCH_FORCE_DIST") == "1"  # 1-rank rehearsal of the RCCL path
    if world > 1 or force_dist:
        import torch.distributed as dist

        if rehearsal:
            dist.init_process_group("gloo")
        else:
            dist.init_process_group("nccl", device_id=dev)

    Nloc, E, D, _ = WORKLOADS[args.workload]
    Ntot = Nloc * world
    if args.cut is None:
        args.cut = (world - 1) / 70.0
    gen = torch.Generator(device=dev).manual_seed(42 + rank)
    src = torch.rand(E, D, generator=gen, device=dev, dtype=torch.float32)

    def make_index(cut):
        """Global destination ids of this rank's E edges: a fraction `cut` uniform over the OTHER ranks' rows, the rest
        uniform over this rank's own rows [rank*Nloc, (rank+1)*Nloc) — a partitioned graph with that edge cut."""
        lo = rank * Nloc
        own = torch.randint(lo, lo + Nloc, (E,), generator=gen, device=dev, dtype=torch.int64)
        if world == 1 or cut <= 0:
            return own
        other = torch.randint(0, Ntot - Nloc, (E,), generator=gen, device=dev, dtype=torch.int64)
        other += (other >= lo).to(torch.int64) * Nloc
        cross = torch.rand(E, generator=gen, device=dev) < cut
        return torch.where(cross, other, own)

    index = make_index(args.cut)
    slab = torch.empty(Nloc, D, device=dev, dtype=torch.float32) if dist is not None else None
    exchange = "sparse"

    def step():
        if dist is None:
            return gnnops.scatter_add(src, index, dim=0, dim_size=Ntot)
        from gnnops.dist import sharded_scatter

        return sharded_scatter(src, index, Ntot, "sum", out_slab=slab, exchange=exchange)

    def fence():
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        out = step()
    fence()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    fence()
    elapsed = time.perf_counter() - t0
    if dist is not None:
        t = torch.tensor([elapsed], device=dev, dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        elapsed = float(t.item())
    ms_per_step = elapsed / args.steps * 1e3
    job_bytes = world * algorithmic_bytes("scatter_add", Nloc, E, D)
    value = job_bytes / (elapsed / args.steps) / 1e9

    result = {
        "metric": "scatter_add effective HBM GB/s (algorithmic bytes / wall time, cold: index partition + bucketed segment reduce per call)",
        "value": round(value, 1),
        "unit": "GB/s",
        "n_gpus": world,
        "steps": args.steps,
        "warmup": args.warmup,
        "ms_per_step": round(ms_per_step, 4),
        "higher_is_better": True,
        "scaling": "weak",
        "vs_baseline": None,
        "dtype": "f32",
        "data": "synthetic (uniform random src, uniform random unsorted int64 row index, device RNG seed 42)" + (
            "" if world == 1 else f"; partitioned graph: {args.cut:.3f} of every rank's edges cross to rows of other ranks"),
        "config": {
            "workload": f"{args.workload} ({'BASELINE configs[1]' if args.workload == 'c2' else 'BASELINE configs[4] per-GPU share' if args.workload == 'c5' else 'side workload'}): "
                        f"torch_scatter.scatter_add(src[E,D], index[E], dim=0, dim_size=N), "
                        f"N={Ntot} E={E * world} D={D} fp32, layout R" + ("" if world == 1 else
                        f"; per GPU E={E}, owned rows={Nloc}; destination-partitioned graph with edge cut {args.cut:.3f} "
                        f"(this fraction of every rank's edges points at rows another rank owns), ONE sparse reduce-scatter "
                        f"(all-to-all-v over RCCL of those edges' (id, row) pairs) overlapped with the own-slab reduce; "
                        f"beside it: other_cuts.cut_0 = config 5 read literally (edges pre-bucketed by owner, nothing to "
                        f"exchange), other_cuts.uniform_random_graph.dense_reduce_scatter = north_star's single RCCL "
                        f"reduce_scatter_tensor of partial [N,D] buffers"),
            "algorithmic_GB_per_step": round(job_bytes / 1e9, 3),
            "pct_of_hbm_peak": round(100 * value / (HBM_PEAK_GBS * world), 2),
        },
    }

    if rehearsal:
        result["rehearsal"] = "every rank on cuda:0 over gloo: code-path check only, NOT a measurement"

    if dist is not None:
        def timed_variant(idx, exch, steps, warm=1, rows=None):
            nonlocal index, exchange, src
            keep = (index, exchange, src)
            index, exchange = idx, exch
            if rows is not None:
                src = rows
            try:
                for _ in range(warm):
                    step()
                fence()
                t1 = time.perf_counter()
                for _ in range(steps):
                    step()
                fence()
                el = time.perf_counter() - t1
            finally:
                index, exchange, src = keep
            t = torch.tensor([el], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item()) / steps
            jb = job_bytes if rows is None else world * algorithmic_bytes("scatter_add", Nloc, rows.size(0), D)
            return {"value": round(jb / el / 1e9, 1), "unit": "GB/s", "ms_per_step": round(el * 1e3, 4)}

        few = max(1, min(args.steps, 3))
        # the N=1 headline's per-GPU work (config 2: E = 50M per GPU when this workload is c5) through THIS sharded step:
        # the like-for-like partner of the driver's --gpus 1 line
        try:
            half = E // 2
            result["same_work"] = {"c2_share": dict(
                timed_variant(index[:half], "sparse", few, rows=src[:half]),
                note=f"per GPU E={half}, owned rows={Nloc}: the per-GPU work of the --gpus 1 headline (config 2) through the "
                     f"same sharded step and edge cut")}
        except Exception as exc:  # noqa: BLE001
            result["same_work"] = {"error": f"{type(exc).__name__}: {exc}"}
        # roofline of the dominant kernel and the CPU baseline at this N too (rank 0, its own edges onto its own slab)
        if rank == 0:
            try:
                local_index = index % Nloc
                result["roofline"] = roofline_leg(torch, gnnops, lib, src, local_index, Nloc, E, D, max(few, 3), False)
                result["roofline"]["note"] = "rank 0: bucket_reduce_kernel over this rank's E edges folded onto its own slab"
                del local_index
            except Exception as exc:  # noqa: BLE001
                result["roofline"] = {"error": f"{type(exc).__name__}: {exc}"}
            if not args.no_cpu_baseline:
                result["cpu_baseline"] = cpu_baseline_leg(D, small=args.workload in ("tiny", "c1"))
        # The legs below are reported beside the headline. A Python-level error in one of them (raised identically on
        # every rank) is recorded instead of losing the headline line.
        # The same per-GPU work on graphs with other edge cuts, beside the headline, never as it:
        #   cut 0           edges pre-bucketed by destination owner (BASELINE config 5 read literally): nothing to exchange
        #   cut (G-1)/G     an UNPARTITIONED uniform random graph: nearly every edge crosses; sparse and dense exchange
        try:
            result["other_cuts"] = {
                "cut_0": dict(timed_variant(make_index(0.0), "sparse", few),
                              note="BASELINE config 5 read literally: edges pre-bucketed by destination owner, empty exchange"),
            }
            if world > 1:
                uni = make_index((world - 1) / world)
                result["other_cuts"]["uniform_random_graph"] = {
                    "cut": round((world - 1) / world, 4),
                    "sparse_exchange": timed_variant(uni, "sparse", few),
                    "dense_reduce_scatter": timed_variant(uni, "dense", few),
                    "note": "an UNPARTITIONED graph; xGMI-bound: bytes on the wire per rank ~ remote edges x 520 B (sparse) or "
                            "(G-1)/G x N x 512 B (dense = north_star's single RCCL reduce_scatter_tensor of partial [N,D] buffers)",
                }
                del uni
        except Exception as exc:  # noqa: BLE001 - see above
            result.setdefault("other_cuts", {})["error"] = f"{type(exc).__name__}: {exc}"

        try:
            # §8(e) spmm, source-partitioned (config 3's per-GPU share on every rank): rank g holds B[its 2M source rows] and
            # 40M nonzeros with columns there; output rows follow the same partitioned-graph model as the headline.
            import gc

            src = None   # 25.6 GB back to the allocator (step() is not called again)
            gc.collect()
            torch.cuda.empty_cache()
            from gnnops.dist import sharded_spmm

            Mloc, nnz, Dm = (2_000_000, 40_000_000, 256) if args.workload in ("c2", "c5") else (max(Nloc // 5, 512), E // 2, 256)
            lo_m = rank * Mloc
            rows_own = torch.randint(lo_m, lo_m + Mloc, (nnz,), generator=gen, device=dev, dtype=torch.int64)
            if world > 1:
                other = torch.randint(0, Mloc * (world - 1), (nnz,), generator=gen, device=dev, dtype=torch.int64)
                other += (other >= lo_m).to(torch.int64) * Mloc
                rows_own = torch.where(torch.rand(nnz, generator=gen, device=dev) < args.cut, other, rows_own)
                del other
            cols = torch.randint(0, Mloc, (nnz,), generator=gen, device=dev, dtype=torch.int64)
            idx2 = torch.stack([rows_own, cols])
            del rows_own, cols
            vals = torch.rand(nnz, generator=gen, device=dev).to(torch.bfloat16)
            Bslab = torch.rand(Mloc, Dm, generator=gen, device=dev).to(torch.bfloat16)
            oslab = torch.empty(Mloc, Dm, device=dev, dtype=torch.bfloat16)
            sharded_spmm(idx2, vals, Mloc * world, Bslab, out_slab=oslab)
            fence()
            t1 = time.perf_counter()
            for _ in range(few):
                sharded_spmm(idx2, vals, Mloc * world, Bslab, out_slab=oslab)
            fence()
            t = torch.tensor([time.perf_counter() - t1], device=dev, dtype=torch.float64)
            dist.all_reduce(t, op=dist.ReduceOp.MAX)
            el = float(t.item()) / few
            result["spmm_src_partitioned"] = {
                "workload": f"per GPU: A block {Mloc * world}x{Mloc}, nnz {nnz}, B slab [{Mloc},{Dm}] bf16, edge cut {args.cut:.3f}; "
                            "COO in, plan built per call",
                "ms_per_step": round(el * 1e3, 4), "GFLOPs_total": round(2 * nnz * Dm * world / el / 1e9, 1),
                "gathered_GBps_total": round(nnz * Dm * 2 * world / el / 1e9, 1),
            }
            del idx2, vals, Bslab, oslab
        except Exception as exc:  # noqa: BLE001
            result["spmm_src_partitioned"] = {"error": f"{type(exc).__name__}: {exc}"}

    if rank == 0 and world == 1 and dist is None:
        result["roofline"] = roofline_leg(torch, gnnops, lib, src, index, Ntot, E, D, args.steps, args.workload == "c2")
        result["warm"] = warm_leg(torch, gnnops, src, index, Ntot, E, D, args.steps)
        if not args.no_extra_ops:
            result["ops"] = extra_ops(torch, gnnops, src, index, Ntot, E, D)
        del out
        if not args.no_extra_ops and args.workload == "c2":
            del src, index
            torch.cuda.empty_cache()
            result["layers"] = layers_leg(torch, gnnops)
            torch.cuda.empty_cache()
            result["config3"] = config3_leg(torch, gnnops)
            torch.cuda.empty_cache()
            result["config4"] = config4_leg(torch, gnnops)
            torch.cuda.empty_cache()
        if args.workload == "c2" and not args.no_extra_ops:
            result["same_work"] = {"c5_share_1gpu": same_work_leg(torch, gnnops, dev, "c5", args.steps)}
        if not args.no_cpu_baseline:
            result["cpu_baseline"] = cpu_baseline_leg(D, small=args.workload in ("tiny", "c1"))
    if rank == 0:
        print(json.dumps(result))
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


def same_work_leg(torch, gnnops, dev, workload, steps):
    """The cold step of the headline on another workload's per-GPU share, one GPU (what the N>1 lines run per rank, minus
    the exchange): the like-for-like base of a 1 -> 8 curve whose N>1 points are config 5's share."""
    Nloc, E, D, _ = WORKLOADS[workload]
    gen = torch.Generator(device=dev).manual_seed(42)
    src = torch.rand(E, D, generator=gen, device=dev, dtype=torch.float32)
    index = torch.randint(0, Nloc, (E,), generator=gen, device=dev, dtype=torch.int64)
    steps = max(3, min(steps, 10))
    for _ in range(2):
        gnnops.scatter_add(src, index, dim=0, dim_size=Nloc)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        gnnops.scatter_add(src, index, dim=0, dim_size=Nloc)
    torch.cuda.synchronize()
    el = (time.perf_counter() - t0) / steps
    jb = algorithmic_bytes("scatter_add", Nloc, E, D)
    del src, index
    torch.cuda.empty_cache()
    return {"workload": f"{workload}: N={Nloc} E={E} D={D} fp32 on one GPU, cold scatter_add", "value": round(jb / el / 1e9, 1),
            "unit": "GB/s", "ms_per_step": round(el * 1e3, 4), "pct_of_hbm_peak": round(100 * jb / el / 1e9 / HBM_PEAK_GBS, 2)}


def _event_ms(torch, fn, iters):
    """Average device time of fn() over `iters` launches, events on the current (= launch) stream."""
    fn()
    torch.cuda.synchronize()
    start, end = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    for _ in range(iters):
        fn()
    end.record()
    torch.cuda.synchronize()
    return start.elapsed_time(end) / iters


def roofline_leg(torch, gnnops, lib, src, index, N, E, D, iters, pmc_applies):
    """Dominant kernel of the timed step: bucket_reduce_kernel<float, SUM> — one launch of gnnops_bucket_reduce over a
    workspace gnnops_bucket_partition filled (the two stages of the one-shot scatter the step runs). Beside it, the
    plan-path kernel seg_rows_kernel<float, SUM> (what a call that reuses a plan runs)."""
    from gnnops import _lib
    from gnnops.ops import _stream

    out = torch.empty(N, D, device=src.device, dtype=torch.float32)
    ws_bytes = lib.gnnops_bucket_workspace_bytes(E, N)
    ws = torch.empty(ws_bytes, dtype=torch.uint8, device=src.device)

    def partition():
        _lib.check(lib.gnnops_bucket_partition(index.data_ptr(), E, N, ws.data_ptr(), ws_bytes, _stream()), "bucket_partition")

    def launch():
        _lib.check(lib.gnnops_bucket_reduce(src.data_ptr(), ws.data_ptr(), out.data_ptr(), None, E, D, N, _lib.F32, _lib.SUM,
                                            0, _stream()), "bucket_reduce")

    partition_ms = _event_ms(torch, partition, 3)
    ms = _event_ms(torch, launch, max(iters, 5))
    del ws

    plan = gnnops.Plan(index, N)

    def launch_seg():
        rc = lib.gnnops_segment_reduce(src.data_ptr(), plan.rowptr.data_ptr(), plan.perm.data_ptr(), out.data_ptr(), None,
                                       1, E, D, N, _lib.F32, _lib.SUM, 0, _stream())
        _lib.check(rc, "segment_reduce")

    seg_ms = _event_ms(torch, launch_seg, max(iters, 5))
    plan_ms = _event_ms(torch, lambda: gnnops.Plan(index, N), 3)
    alg = algorithmic_bytes("scatter_add", N, E, D)
    achieved = alg / (ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "pmc_traffic.json")
    if pmc_applies and os.path.exists(tpath):  # the committed PMC passes were taken at config 2
        try:
            traffic = json.load(open(tpath)).get("bucket_reduce_kernel_f32_sum", {}).get("hbm_bytes_per_launch")
        except Exception:
            traffic = None
    # What this box's memory system gives plain streams with the kernel's read : write ratio (5 rows + index in, 1 row
    # out): sequential nontemporal streams, no index, no row structure (gnnops_diag_stream_mix) — and reads alone.
    mix = None
    try:
        pieces = N * D * 4 // 16                       # one output row's worth of 16-B pieces per destination
        reads = max(1, min(5, E // max(N, 1)))
        if pieces * reads * 16 <= src.numel() * 4 and reads in (1, 2, 3, 4, 5):
            def run_mix(dst):
                _lib.check(lib.gnnops_diag_stream_mix(src.data_ptr(), dst, pieces, reads, _stream()), "diag_stream_mix")

            mix_ms = _event_ms(torch, lambda: run_mix(out.data_ptr()), 5)
            rd_ms = _event_ms(torch, lambda: run_mix(None), 5)
            mix_bytes = pieces * 16 * (reads + 1)
            mix = {"what": f"{reads} sequential nontemporal read streams : 1 write stream, same bytes per output row as the "
                           f"kernel minus the index (gnnops_diag_stream_mix), measured in this run",
                   "GBps": round(mix_bytes / mix_ms / 1e6, 1), "reads_only_GBps": round(pieces * 16 * reads / rd_ms / 1e6, 1),
                   "kernel_frac_of_mix": round(achieved / (mix_bytes / mix_ms / 1e6), 4)}
    except Exception as exc:  # reporting extra only
        mix = {"error": str(exc)[:200]}
    return {
        "kernel": "bucket_reduce_kernel<float,SUM> (gnnops_bucket_reduce)",
        "bound": "hbm",
        "achieved": round(achieved, 1),
        "peak": HBM_PEAK_GBS,
        "unit": "GB/s",
        "frac": round(achieved / HBM_PEAK_GBS, 4),
        "traffic": traffic,
        "traffic_source": "profiles/pmc_traffic.json (committed rocprofv3 PMC passes FETCH_SIZE x2 + WRITE_SIZE at config 2; "
                          "not a measurement of this run)" if traffic is not None else None,
        "stream_mix_ceiling": mix,
        "kernel_ms": round(ms, 4),
        "algorithmic_bytes_per_launch": alg,
        "partition_ms": round(partition_ms, 4),
        "plan_path": {"kernel": "seg_rows_kernel<float,SUM> (gnnops_segment_reduce)", "kernel_ms": round(seg_ms, 4),
                      "frac": round(alg / (seg_ms * 1e-3) / 1e9 / HBM_PEAK_GBS, 4), "plan_build_ms": round(plan_ms, 4)},
    }


def warm_leg(torch, gnnops, src, index, N, E, D, iters):
    """Same op with the plan reused across calls (static edge_index): what a GNN layer loop sees."""
    plan = gnnops.Plan(index, N)
    ms = _event_ms(torch, lambda: gnnops.scatter_add(src, plan, dim=0), max(iters, 5))
    alg = algorithmic_bytes("scatter_add", N, E, D)
    return {"ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 1), "pct_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS * 100, 2)}


def extra_ops(torch, gnnops, src, index, N, E, D):
    """The other config-2 ops, cold (no cached plan) and, where a plan applies, warm."""
    res = {}
    plan = gnnops.Plan(index, N)

    def rec(name, fn, alg_op, iters=5):
        ms = _event_ms(torch, fn, iters)
        alg = algorithmic_bytes(alg_op, N, E, D)
        res[name] = {"ms": round(ms, 4), "GBps": round(alg / ms / 1e6, 1),
                     "pct_of_hbm_peak": round(alg / ms / 1e6 / HBM_PEAK_GBS * 100, 2)}

    rec("scatter_mean_cold", lambda: gnnops.scatter_mean(src, index, 0, dim_size=N), "scatter_mean")
    rec("scatter_min_cold", lambda: gnnops.scatter_min(src, index, 0, dim_size=N), "scatter_min")
    rec("scatter_max_cold", lambda: gnnops.scatter_max(src, index, 0, dim_size=N), "scatter_max")
    rec("scatter_min_warm", lambda: gnnops.scatter_min(src, plan, 0), "scatter_min")
    acc = torch.zeros(N, D, device=src.device)
    rec("index_add__cold", lambda: gnnops.index_add_(acc, 0, index, src), "index_add_")
    rec("index_add__warm", lambda: gnnops.index_add_(acc, 0, plan, src), "index_add_")
    del acc
    table = torch.rand(N, D, device=src.device)
    from gnnops import ops as _ops

    rec("index_select_cold", lambda: gnnops.index_select(table, 0, index), "index_select")  # auto: plan + push
    rec("index_select_warm", lambda: gnnops.index_select(table, 0, index, plan=plan), "index_select")
    saved = _ops._PUSH_MIN_TABLE_BYTES
    _ops._PUSH_MIN_TABLE_BYTES = 1 << 62  # force the pull form for comparison
    try:
        rec("index_select_pull", lambda: gnnops.index_select(table, 0, index), "index_select")
    finally:
        _ops._PUSH_MIN_TABLE_BYTES = saved
    del table, plan
    # destination-sorted variant (real edge_index is usually coalesced): the same cold op on an ascending
    # index, and torch_scatter.segment_coo for a caller that KNOWS it is sorted (no partition at all)
    sorted_index = torch.sort(index).values
    rec("scatter_add_cold_sorted_index", lambda: gnnops.scatter_add(src, sorted_index, 0, dim_size=N), "scatter_add")
    from gnnops import segment as _seg

    rec("segment_coo_sorted_index", lambda: _seg.segment_coo(src, sorted_index, dim_size=N, reduce="sum"), "scatter_add")
    return res


MFMA_PEAK_TFLOPS = 2500.0  # dense bf16/fp16 (MI355X_MICROARCH.md)


def config3_leg(torch, gnnops):
    """BASELINE configs[2]: spmm over CSR 2M x 2M, nnz 40M, D=256 bf16 + addmm (GNN shape and square)."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    M, nnz, D = 2_000_000, 40_000_000, 256
    row = torch.randint(0, M, (nnz,), generator=g, device=dev).sort().values
    col = torch.randint(0, M, (nnz,), generator=g, device=dev)
    rowptr = torch.zeros(M + 1, dtype=torch.int32, device=dev)
    rowptr[1:] = torch.bincount(row, minlength=M).cumsum(0).to(torch.int32)
    del row
    val = torch.rand(nnz, generator=g, device=dev).to(torch.bfloat16)
    Bm = torch.rand(M, D, generator=g, device=dev).to(torch.bfloat16)
    ms = _event_ms(torch, lambda: gnnops.spmm_csr(rowptr, col, val, Bm), 5)
    alg = nnz * (8 + 2) + (M + 1) * 8 + M * D * 2 + M * D * 2      # 2.464 GB
    gathered = nnz * D * 2
    res = {"spmm_csr_bf16": {"ms": round(ms, 4), "alg_GBps": round(alg / ms / 1e6, 1), "gathered_GBps": round(gathered / ms / 1e6, 1),
                             "pct_of_hbm_peak_alg": round(alg / ms / 1e6 / HBM_PEAK_GBS * 100, 2),
                             "GFLOPs": round(2 * nnz * D / ms / 1e6, 1)}}
    # The same matrix shape with STRUCTURED columns (uniform columns over a 1 GB B leave no locality to buy;
    # real graphs are not uniform). banded: col = row + U(-w, w) — neighbouring output rows gather neighbouring rows of B, which
    # then come from L2 / the Infinity Cache instead of HBM. community: 2000 blocks of 1000 nodes, 90 % of a row's columns
    # inside its own block. Same nnz, same rowptr, same kernel.
    try:
        row_of = torch.repeat_interleave(torch.arange(M, device=dev), (rowptr[1:] - rowptr[:-1]).long())
        for name, mk in (("banded_w4096", lambda: (row_of + torch.randint(-4096, 4097, (nnz,), generator=g, device=dev)).clamp_(0, M - 1)),
                         ("community_1000_p90", lambda: torch.where(torch.rand(nnz, generator=g, device=dev) < 0.9,
                                                                    (row_of // 1000) * 1000 + torch.randint(0, 1000, (nnz,), generator=g, device=dev),
                                                                    torch.randint(0, M, (nnz,), generator=g, device=dev)))):
            c2 = mk()
            ms2 = _event_ms(torch, lambda: gnnops.spmm_csr(rowptr, c2, val, Bm), 5)
            res["spmm_csr_bf16_" + name] = {"ms": round(ms2, 4), "gathered_GBps": round(gathered / ms2 / 1e6, 1),
                                            "alg_GBps": round(alg / ms2 / 1e6, 1),
                                            "pct_of_hbm_peak_alg": round(alg / ms2 / 1e6 / HBM_PEAK_GBS * 100, 2)}
            del c2
        del row_of
    except Exception as exc:  # reporting extra only
        res["spmm_csr_bf16_structured"] = {"error": str(exc)[:200]}
    del col, val
    W = torch.rand(D, D, generator=g, device=dev).to(torch.bfloat16)
    ms = _event_ms(torch, lambda: gnnops.addmm(Bm, Bm, W), 5)
    fl = 2 * M * D * D
    res["addmm_gnn_shape_bf16"] = {"shape": f"[{M},{D}] + [{M},{D}]@[{D},{D}]", "ms": round(ms, 4), "TFLOPs": round(fl / ms / 1e9, 1),
                                   "alg_GBps": round((3 * M * D * 2 + D * D * 2) / ms / 1e6, 1)}
    del Bm, W
    L = 8192
    a = (torch.rand(L, L, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    b = (torch.rand(L, L, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    c = (torch.rand(L, L, generator=g, device=dev) * 2 - 1).to(torch.bfloat16)
    for _ in range(40):   # the matrix cores come out of memory-bound legs at a lower clock: ~40 ms of GEMMs to settle
        gnnops.addmm(c, a, b)
    ms = _event_ms(torch, lambda: gnnops.addmm(c, a, b), 20)
    res["addmm_square_bf16"] = {"shape": f"{L}^3", "ms": round(ms, 4), "TFLOPs": round(2 * L ** 3 / ms / 1e9, 1),
                                "mfma_frac_of_dense_peak": round(2 * L ** 3 / ms / 1e9 / MFMA_PEAK_TFLOPS, 4)}
    del a, b, c
    # the reference's own protocol: fp16, square, L = int(sqrt(x)) (benchmark_native_addmm.py:23-27) — odd row lengths, K not a
    # multiple of the K-tile, tile counts that do not fill whole rounds of CUs (289, 529, 784 tiles: split-K tail)
    sweep = {}
    for L in (1581, 4249, 5797, 7011, 8164):
        a, b, c = [(torch.rand(L, L, generator=g, device=dev) * 2 - 1).half() for _ in range(3)]
        gnnops.addmm(c, a, b)
        ms = _event_ms(torch, lambda: gnnops.addmm(c, a, b), 10)
        sweep[str(L)] = {"ms": round(ms, 4), "TFLOPs": round(2 * L ** 3 / ms / 1e9, 1)}
        del a, b, c
    res["addmm_reference_lengths_fp16"] = sweep
    return res


def config4_leg(torch, gnnops):
    """BASELINE configs[3]: fused index_select+sum vs the unfused pair, E=100M, D=128 fp16 (N=E, RF 1)."""
    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    E, D = 100_000_000, 128
    table = torch.empty(E, D, device=dev, dtype=torch.float16).uniform_(0, 1, generator=g)
    index = torch.randint(0, E, (E,), generator=g, device=dev)
    fused = _event_ms(torch, lambda: gnnops.index_select_sum(table, 0, index), 3)

    def unfused():
        sel = gnnops.index_select(table, 0, index)        # our pull kernel, [E,D] materialised
        return sel.sum(dtype=torch.float32)               # torch reduction with an fp32 accumulator

    unf = _event_ms(torch, unfused, 2)
    alg_f = E * 8 + E * D * 2
    alg_u = E * 8 + 3 * E * D * 2
    return {"fused_ms": round(fused, 4), "fused_alg_GBps": round(alg_f / fused / 1e6, 1),
            "fused_pct_of_hbm_peak": round(alg_f / fused / 1e6 / HBM_PEAK_GBS * 100, 2),
            "unfused_ms": round(unf, 4), "unfused_alg_GBps": round(alg_u / unf / 1e6, 1),
            "speedup": round(unf / fused, 2),
            "note": "unfused = gnnops.index_select (materialises [E,D]) + torch sum(dtype=float32); fp32 accumulators compared"}


def layers_leg(torch, gnnops):
    """SURVEY 8f rank 4 (side key, not the headline): one CGConv layer (app_bm/groq_script.py:91-109) on a graph of config 2's
    size — N = 10M nodes, E = 50M edges, 128 channels fp16 — as one dense product + one fused edge pass, and the edge pass
    alone against its algorithmic bytes."""
    from gnnops import conv

    dev = torch.device("cuda")
    g = torch.Generator(device=dev).manual_seed(42)
    n, e, d = 10_000_000, 50_000_000, 128
    ei = torch.randint(0, n, (2, e), generator=g, device=dev)
    x = (torch.rand(n, d, generator=g, device=dev) - 0.5).half()
    torch.manual_seed(0)
    layer = conv.CGConv(d, 0).half().to(dev)
    gnnops.set_plan_cache(True)     # the headline legs run with the cache off (cold); a layer stack reuses its graph's plan
    try:
        with torch.no_grad():
            whole = _event_ms(torch, lambda: layer(x, ei), 3)
            pq = torch.empty(n, 4 * d, dtype=torch.float16, device=dev).normal_(generator=g)
            edge = _event_ms(torch, lambda: conv.edge_reduce("cgconv", pq[:, 2 * d:], ei, n, p=pq[:, :2 * d], add=x), 3)
            copy = _event_ms(torch, lambda: conv.edge_reduce("copy", x, ei, n, add=x), 3)
            gnnops.set_plan_cache(False)
            cold = _event_ms(torch, lambda: layer(x, ei), 2)
    finally:
        gnnops.set_plan_cache(False)
    alg_edge = e * (2 * d * 2 + 8) + n * (2 * d * 2 + 2 * d * 2) + 4 * (n + 1)
    alg_copy = e * (d * 2 + 8) + n * (2 * d * 2) + 4 * (n + 1)
    return {"graph": f"N={n} E={e} uniform endpoints; plan of the graph cached except where 'cold'", "cgconv128_fp16_layer_ms": round(whole, 4),
            "cgconv128_fp16_layer_cold_ms": round(cold, 4),
            "cgconv128_fp16_edge_pass": {"ms": round(edge, 4), "alg_GBps": round(alg_edge / edge / 1e6, 1),
                                         "pct_of_hbm_peak": round(alg_edge / edge / 1e6 / HBM_PEAK_GBS * 100, 2), "bound": "VALU (2 exp, log, rcp per element) + gather"},
            "gather_sum128_fp16_edge_pass": {"ms": round(copy, 4), "alg_GBps": round(alg_copy / copy / 1e6, 1),
                                             "pct_of_hbm_peak": round(alg_copy / copy / 1e6 / HBM_PEAK_GBS * 100, 2), "bound": "hbm"}}


def cpu_baseline_leg(D, small=False):
    """The C oracle port (one core) on a bounded sample of the same workload: N=1M, E=5M, D as configured (small: the
    rehearsal workloads of the test suite, a tenth of that)."""
    import numpy as np

    from oracle import oracle

    oracle.lib()
    Ns, Es = (100_000, 500_000) if small else (1_000_000, 5_000_000)
    rng = np.random.default_rng(42)
    src = rng.random((Es, D), dtype=np.float32)
    idx = rng.integers(0, Ns, Es, dtype=np.int64)
    oracle.scatter_add_rows_f32(src[:1000], idx[:1000] % 10, 10)  # page in the library
    t0 = time.perf_counter()
    reps = 0
    while True:
        oracle.scatter_add_rows_f32(src, idx, Ns)
        reps += 1
        dt = time.perf_counter() - t0
        if dt > (1.0 if small else 10.0) or reps >= 20:
            break
    alg = algorithmic_bytes("scatter_add", Ns, Es, D)
    res = {
        "value": round(alg * reps / dt / 1e9, 3),
        "unit": f"GB/s (algorithmic bytes / wall time, on a sample of config 2's shape: N={Ns} E={Es})",
        "cores": 1,
        "kind": "port",
        "sample": f"oracle/gnnops_oracle.c ora_scatter_add_rows_f32, N={Ns} E={Es} D={D} fp32 (config 2 scaled down), "
                  f"{reps} passes in {dt:.1f} s on one of {os.cpu_count()} host cores",
    }
    # Beside it: what the reference's op body executes on CPU tensors (torch_scatter.scatter_add forwards to
    # Tensor.scatter_add_ / index_add_), PyTorch's own CPU kernel with its default thread count, same sample.
    try:
        import torch

        tsrc, tidx = torch.from_numpy(src), torch.from_numpy(idx)
        torch.zeros(Ns, D).index_add_(0, tidx, tsrc)
        t0 = time.perf_counter()
        treps = 0
        while time.perf_counter() - t0 < (1.0 if small else 5.0) and treps < 20:
            torch.zeros(Ns, D).index_add_(0, tidx, tsrc)
            treps += 1
        tdt = time.perf_counter() - t0
        res["torch_cpu"] = {"value": round(alg * treps / tdt / 1e9, 3), "unit": "GB/s", "threads": torch.get_num_threads(),
                            "op": "torch.zeros(N, D).index_add_(0, index, src)"}
        # the reference's literal protocol: torch.utils.benchmark.Timer pins torch to ONE thread (timer.py:266 num_threads=1)
        nthreads = torch.get_num_threads()
        torch.set_num_threads(1)
        try:
            t0 = time.perf_counter()
            treps = 0
            while time.perf_counter() - t0 < (1.0 if small else 5.0) and treps < 20:
                torch.zeros(Ns, D).index_add_(0, tidx, tsrc)
                treps += 1
            tdt = time.perf_counter() - t0
            res["torch_cpu_1_thread"] = {"value": round(alg * treps / tdt / 1e9, 3), "unit": "GB/s", "threads": 1,
                                         "op": "the same under Timer's default num_threads=1"}
        finally:
            torch.set_num_threads(nthreads)
    except Exception as exc:  # reporting extra only
        res["torch_cpu"] = {"error": str(exc)[:200]}
    return res


if __name__ == "__main__":
    main()
