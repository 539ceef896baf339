"""One-off extended fuzz on the GPU box (not part of the suite): the seeded random sweeps of tests/test_random_gpu.py over many more
seeds, random 16-bit addmm shapes (in-place operands vs whole padded copies bit for bit, split-K vs plain grid within one
rounding, sampled rows vs float64), random clouds through the grid knn / radius vs the exhaustive kernels.
usage: fuzz_more.py [first_seed] [count]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path[:0] = [os.path.join(ROOT, "gnn-ops-benchmark_amd"), ROOT, os.path.join(ROOT, "tests")]
import numpy as np, torch
import gnnops
from oracle import oracle
import test_random_gpu as T

first = int(sys.argv[1]) if len(sys.argv) > 1 else 100
count = int(sys.argv[2]) if len(sys.argv) > 2 else 60
gnnops.set_plan_cache(False)
fails = 0
t0 = time.time()
fns = [getattr(T, n) for n in dir(T) if n.startswith("test_random")]
for seed in range(first, first + (0 if os.environ.get('FUZZ_SKIP_SWEEPS') else count)):
    for fn in fns:
        try:
            fn(gnnops, oracle, seed)
        except Exception as e:      # noqa: BLE001 - report and go on
            fails += 1
            print(f"FAIL {fn.__name__} seed={seed}: {type(e).__name__}: {str(e)[:300]}", flush=True)
    if (seed - first) % 10 == 9:
        print(f"... random sweeps through seed {seed}, {time.time() - t0:.0f} s, {fails} failures", flush=True)

# ---- layout-F scatters with more destinations than an LDS strip holds (chunked last-dim kernels, 16-B K == 1 stream, dim-0 route
#      through the transposes with the implicit size found on the way)
from gnnops import ops as _ops
_ops._FUSED_MAX_MIN_NUMEL = 1
rng2 = np.random.default_rng(first + 7)
gc = torch.Generator().manual_seed(first + 7)
for it in range(count):
    rows = int(rng2.integers(2, 40)); E = int(rng2.integers(50, 3000)); N = int(rng2.integers(21000, 120000))
    if it % 3 == 0: E = (E + 3) // 4 * 4
    reduce = str(rng2.choice(["sum", "mean", "min", "max", "mul"]))
    dim = int(rng2.integers(0, 2))
    shape = (rows, E) if dim == 1 else (E, rows)
    src = (torch.rand(shape, generator=gc) * 2 - 1)
    idx = torch.randint(0, N, shape, generator=gc)
    idx.view(-1)[-1] = N - 1
    if it % 2: idx[: max(1, shape[0] // 3)] = idx[-max(1, shape[0] // 3):]
    try:
        res = gnnops.scatter(src.cuda(), idx.cuda(), dim, reduce=reduce) if it % 2 else gnnops.scatter(src.cuda(), idx.cuda(), dim, dim_size=N, reduce=reduce)
        exp = oracle.scatter(src.numpy(), idx.numpy(), dim, dim_size=N, reduce=reduce, dtype="f32")
        if reduce in ("min", "max"):
            ok = np.array_equal(res[1].cpu().numpy(), exp[1]) and np.array_equal(res[0].cpu().numpy().view(np.uint32), exp[0].view(np.uint32))
        else:
            ok = res.shape == exp.shape and np.allclose(res.cpu().numpy(), exp, rtol=1e-5, atol=1e-6)
    except Exception as e:      # noqa: BLE001
        ok = False
        print(f"   {type(e).__name__}: {str(e)[:200]}")
    if not ok:
        fails += 1
        print(f"FAIL big-N scatter shape={shape} dim={dim} N={N} {reduce} implicit={bool(it % 2)}", flush=True)
print(f"... big-N layout-F scatters done, {time.time() - t0:.0f} s, {fails} failures", flush=True)

# ---- sparse family against torch on the CPU: coalesce (index bit-exact, values to a tolerance), spmm, spspmm, sparse transpose
import torch_sparse
rng3 = np.random.default_rng(first + 13)
gs = torch.Generator().manual_seed(first + 13)
for it in range(count):
    m = int(rng3.integers(1, 3000)); n = int(rng3.integers(1, 3000)); nnz = int(rng3.integers(0, 60000)); D = int(rng3.choice([1, 3, 16, 64, 100]))
    row = torch.randint(0, m, (nnz,), generator=gs); col = torch.randint(0, n, (nnz,), generator=gs)
    val = torch.rand(nnz, generator=gs) - 0.5
    idx = torch.stack([row, col])
    try:
        ci, cv = torch_sparse.coalesce(idx.cuda(), val.cuda(), m, n)
        ref = torch.sparse_coo_tensor(idx, val, (m, n)).coalesce()
        ok = torch.equal(ci.cpu(), ref.indices()) and torch.allclose(cv.cpu(), ref.values(), rtol=1e-5, atol=1e-6)
        dense = torch.rand(n, D, generator=gs) - 0.5
        out = torch_sparse.spmm(idx.cuda(), val.cuda(), m, n, dense.cuda())
        ok = ok and torch.allclose(out.cpu(), torch.sparse.mm(ref, dense), rtol=1e-4, atol=1e-5)
        ti, tv = torch_sparse.transpose(ci, cv, m, n)
        rt = ref.t().coalesce()
        ok = ok and torch.equal(ti.cpu(), rt.indices()) and torch.allclose(tv.cpu(), rt.values(), rtol=1e-5, atol=1e-6)
        if it % 4 == 0 and nnz < 20000:
            k2 = int(rng3.integers(1, 2000)); nnz2 = int(rng3.integers(0, 20000))
            r2 = torch.randint(0, n, (nnz2,), generator=gs); c2 = torch.randint(0, k2, (nnz2,), generator=gs); v2 = torch.rand(nnz2, generator=gs) - 0.5
            pi, pv = torch_sparse.spspmm(idx.cuda(), val.cuda(), torch.stack([r2, c2]).cuda(), v2.cuda(), m, n, k2)
            rp = torch.sparse.mm(ref, torch.sparse_coo_tensor(torch.stack([r2, c2]), v2, (n, k2)).coalesce()).coalesce()
            got = torch.sparse_coo_tensor(pi.cpu(), pv.cpu(), (m, k2)).coalesce()
            ok = ok and torch.allclose(got.to_dense(), rp.to_dense(), rtol=1e-4, atol=1e-5) and bool((pi.cpu()[0][1:] * k2 + pi.cpu()[1][1:] > pi.cpu()[0][:-1] * k2 + pi.cpu()[1][:-1]).all())
    except Exception as e:      # noqa: BLE001
        ok = False
        print(f"   {type(e).__name__}: {str(e)[:200]}")
    if not ok:
        fails += 1
        print(f"FAIL sparse m={m} n={n} nnz={nnz} D={D}", flush=True)
print(f"... sparse family done, {time.time() - t0:.0f} s, {fails} failures", flush=True)

# ---- addmm shapes
rng = np.random.default_rng(first)
g = torch.Generator(device="cuda").manual_seed(first)
for it in range(count):
    M = int(rng.integers(512, 6000)); N = int(rng.integers(512, 6000)); K = int(rng.integers(256, 3000))
    dt = torch.float16 if it % 2 else torch.bfloat16
    a = (torch.rand(M, K, generator=g, device="cuda") - 0.5).to(dt)
    b = (torch.rand(K, N, generator=g, device="cuda") - 0.5).to(dt)
    c = (torch.rand(M, N, generator=g, device="cuda") - 0.5).to(dt)
    got = gnnops.addmm(c, a, b)
    os.environ["GNNOPS_GEMM_PAD"] = "full"
    full = gnnops.addmm(c, a, b)
    os.environ.pop("GNNOPS_GEMM_PAD")
    os.environ["GNNOPS_GEMM_SK"] = "0"
    plain = gnnops.addmm(c, a, b)
    os.environ.pop("GNNOPS_GEMM_SK")
    rows = torch.randint(0, M, (16,), device="cuda")
    ref = c[rows].double() + a[rows].double() @ b.double()
    err = (got[rows].double() - ref).abs().max().item()
    ulp = (2.0 ** -7 if dt == torch.bfloat16 else 2.0 ** -10) * max(1.0, ref.abs().max().item())
    ok = torch.equal(got, full) and (got.float() - plain.float()).abs().max().item() <= ulp and err <= 2 * ulp + 4 * K * 2.0 ** -24 * 0.25 * K
    if not ok:
        fails += 1
        print(f"FAIL addmm M={M} N={N} K={K} {dt}: equal_full={torch.equal(got, full)} d_plain={(got.float() - plain.float()).abs().max().item():.4g} err64={err:.4g}", flush=True)
print(f"... addmm shapes done, {time.time() - t0:.0f} s, {fails} failures", flush=True)

# ---- knn / radius clouds
from torch_cluster import knn, radius
from gnnops import spatial
for it in range(count // 2):
    D = int(rng.integers(1, 4)); n = int(rng.integers(8192, 60000)); k = int(rng.integers(1, 65))
    kind = it % 3
    x = torch.rand(n, D, generator=g, device="cuda")
    if kind == 1: x = (x * 20).floor() / 20
    if kind == 2: x[: n // 2] = x[: n // 2] * 0.01 + 5.0
    y = torch.rand(int(rng.integers(10, 400)), D, generator=g, device="cuda") * 1.5 - 0.25
    r = float(rng.random() * 0.2)
    a1, b1 = knn(x, y, k), radius(x, y, r, max_num_neighbors=min(k, 64))
    save = spatial._KNN_GRID_MIN_POINTS
    spatial._KNN_GRID_MIN_POINTS = 1 << 40
    a0, b0 = knn(x, y, k), radius(x, y, r, max_num_neighbors=min(k, 64))
    spatial._KNN_GRID_MIN_POINTS = save
    if not (torch.equal(a0, a1) and torch.equal(b0, b1)):
        fails += 1
        print(f"FAIL knn/radius D={D} n={n} k={k} kind={kind} r={r:.3f}: knn={torch.equal(a0, a1)} radius={torch.equal(b0, b1)}", flush=True)
print(f"DONE in {time.time() - t0:.0f} s: {fails} failures", flush=True)
