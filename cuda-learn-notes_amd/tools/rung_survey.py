"""Every exported name of the bandwidth families at ONE bandwidth shape ([4096,4096] unless the name's constraints say otherwise), rotating buffer sets,
one event-timed region per name: microseconds, algorithmic GB/s, and the ratio to the family's widest rung -- a narrow rung is slower by construction
(2- / 4-byte accesses), a rung that is slower than its access width explains is a walk to fix (round 6 found gemv and the scalar embedding rungs this way).
    python tools/rung_survey.py [rows cols]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import __graft_entry__ as ge
pkg = ge.load_package()
from cuda_learn_notes_amd import manifest, bench_utils as bu
S, H = (int(sys.argv[1]), int(sys.argv[2])) if len(sys.argv) > 2 else (4096, 4096)
dev = torch.device("cuda:0")
NSETS = 6


def region(calls):
    def rot():
        for c in calls:
            c()
    bu.prewarm(rot, 0.08)
    reps = max(2, int(30.0 / max(bu.time_region_events(rot, 1) * len(calls), 1e-3)))
    return bu.time_region_events(rot, min(reps, 200)) / len(calls)


rows = []
for e in manifest.ENTRIES:
    if e.lib not in ("elementwise", "reduce", "softmax", "layer_norm", "rms_norm", "rope", "activation", "dot_product", "embedding", "mat_transpose", "histogram"):
        continue
    try:
        lib = pkg.load(e.lib)
        fn = getattr(lib, e.name)
        n = S * H
        if e.sig == "P3":
            dt = torch.float32 if "_f32" in e.name else torch.float16
            bufs = [(torch.randn(S, H, device=dev).to(dt), torch.randn(S, H, device=dev).to(dt), torch.empty(S, H, device=dev, dtype=dt)) for _ in range(NSETS)]
            calls = [(lambda a=a, b=b, c=c: fn(a, b, c)) for a, b, c in bufs]
            nbytes = 3 * n * bufs[0][0].element_size()
        elif e.sig == "R1":
            in_dt = getattr(torch, manifest.REDUCE_DTYPES[e.name][0])
            src = torch.randn(S, H, device=dev)
            bufs = [(src.clamp(-3, 3) * (10 if in_dt == torch.int8 else 1)).to(in_dt) for _ in range(NSETS)]
            calls = [(lambda x=x: fn(x)) for x in bufs]
            nbytes = n * bufs[0].element_size()
        elif e.sig in ("XY", "SG", "UN"):
            dt = torch.float16 if "_f16" in e.name else torch.float32
            bufs = [(torch.randn(S, H, device=dev).to(dt), torch.empty(S, H, device=dev, dtype=dt)) for _ in range(NSETS)]
            calls = [(lambda x=x, y=y: fn(x, y)) for x, y in bufs]
            nbytes = 2 * n * bufs[0][0].element_size()
        elif e.sig in ("LN", "RN"):
            dt = torch.float32 if e.name.split("norm_")[1].startswith("f32") else torch.float16
            bufs = [(torch.randn(S, H, device=dev).to(dt), torch.empty(S, H, device=dev, dtype=dt)) for _ in range(NSETS)]
            calls = [(lambda x=x, y=y: fn(x, y, 1.0, 0.0) if e.sig == "LN" else fn(x, y, 1.0)) for x, y in bufs]
            nbytes = 2 * n * bufs[0][0].element_size()
        elif e.sig == "RP":
            bufs = [(torch.randn(S, H, device=dev), torch.empty(S, H, device=dev)) for _ in range(NSETS)]
            calls = [(lambda x=x, y=y: fn(x, y)) for x, y in bufs]
            nbytes = 2 * n * 4
        elif e.sig == "D2":
            dt = torch.float16 if "f16" in e.name.split("dot_prod_")[1][:4] else torch.float32
            bufs = [(torch.randn(n, device=dev).to(dt), torch.randn(n, device=dev).to(dt)) for _ in range(NSETS)]
            calls = [(lambda a=a, b=b: fn(a, b)) for a, b in bufs]
            nbytes = 2 * n * bufs[0][0].element_size()
        elif e.sig == "EM":
            dt = torch.float16 if "_f16" in e.name else torch.float32
            w = torch.randn(50000, H, device=dev).to(dt)
            bufs = [(torch.randint(0, 50000, (S,), device=dev, dtype=torch.int32), torch.empty(S, H, device=dev, dtype=dt)) for _ in range(NSETS)]
            calls = [(lambda i=i, o=o: fn(i, w, o)) for i, o in bufs]
            nbytes = 2 * n * w.element_size()
        elif e.sig == "TR":
            bufs = [(torch.randn(S, H, device=dev), torch.empty(H, S, device=dev)) for _ in range(NSETS)]
            calls = [(lambda x=x, y=y: fn(x, y)) for x, y in bufs]
            nbytes = 2 * n * 4
        elif e.sig == "HI":
            bufs = [torch.randint(0, 1024, (n,), device=dev, dtype=torch.int32) for _ in range(NSETS)]
            calls = [(lambda a=a: fn(a)) for a in bufs]
            nbytes = n * 4
        else:
            continue
        calls[0]()
        torch.cuda.synchronize()
        ms = region(calls)
        rows.append((e.lib, e.name, ms * 1e3, nbytes / ms * 1e-6))
        del bufs, calls
        torch.cuda.empty_cache()
    except Exception as ex:  # noqa: BLE001
        print("SURVEY %-14s %-52s error %s" % (e.lib, e.name, str(ex)[:90]), flush=True)
best = {}
for lib, name, us, gbps in rows:
    best[lib] = max(best.get(lib, 0.0), gbps)
for lib, name, us, gbps in rows:
    print("SURVEY %-14s %-52s %9.2f us %8.1f GB/s  %.2f of the family's best" % (lib, name, us, gbps, gbps / best[lib]), flush=True)
