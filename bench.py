#!/usr/bin/env python3
"""bench.py -- headline benchmark on MI355X: HGEMM fp16 TFLOPS at M=N=K=4096 (BASELINE.json metric),
with FA2 forward TFLOPS (D=64, D=128) and the rocBLAS row reported beside it.

  python bench.py --gpus N --steps K --warmup W
  (N>1: python -m torch.distributed.run --nnodes=1 --nproc-per-node N ... bench.py --gpus N ...)

A "step" is one pass of the hot path over one batch of synthetic input: ONE 4096^3 HGEMM launch
(reference config C3: kernels/hgemm/hgemm.py --mma --MNK 4096, the
hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem row with block swizzle). Inputs are resident in HBM
before the timed region. Multi-GPU = independent replicas (one HGEMM does not shard; no collective on the
data path): value = N * flops / max-over-ranks time, scaling "weak".
Output on rank 0 (the driver keeps an 8 KB stdout tail and parses the LAST line):
  * every measured row in full -> bench_detail.json (repository root, and gpurun_out/ when it exists);
  * short one-row-per-kernel lines on stdout, as the reference scripts print (kernels/hgemm/hgemm.py:142-168);
  * the LAST stdout line: one compact JSON object (< 4 KB, `headline_line`) with metric / value / ms_per_step / config /
    roofline / cpu_baseline and a digest of the side rows. tests/test_bench_line.py holds the size bound.
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402


def parse():
    p = argparse.ArgumentParser()
    p.add_argument("--gpus", type=int, default=1)
    p.add_argument("--steps", type=int, default=200)
    p.add_argument("--warmup", type=int, default=20)
    p.add_argument("--mnk", type=int, default=4096)
    p.add_argument("--stages", type=int, default=2)
    p.add_argument("--prewarm", type=float, default=0.6, help="seconds of untimed back-to-back launches before warmup")
    p.add_argument("--no-extras", action="store_true", help="skip FA2 / rocBLAS / CPU baseline side measurements")
    p.add_argument("--no-pmc", action="store_true", help="do not re-run the headline kernel under rocprofv3 --pmc for this box's counters (3 child passes, ~20 s)")
    p.add_argument("--no-configs", action="store_true", help="skip the per-config rows (C1, C2, 8192^3, stage sweeps, bandwidth kernels)")
    p.add_argument("--launch", choices=["graph", "eager"], default="eager",
                   help="how the K timed steps reach the GPU: K eager launches from Python (default: host enqueue 9.5 us per step against a "
                        "94 us kernel) or one hipGraph holding the K launches (measured slower: +26 us per kernel node, 1141 vs 1442 TF)")
    return p.parse_args()


def fa_roofline(kern, shape, pmc_file, dev, bu, profiles, kernel_desc=""):
    """roofline object of one FlashAttention-2 forward config: algorithmic flops (4 B H N^2 D) / mean launch
    duration from HIP events over `iters` back-to-back launches on the launch stream (after a time-based pre-warm)."""
    B_, H_, N_, D = shape
    q, k, v = (torch.randn(B_, H_, N_, D, dtype=torch.half, device=dev) for _ in range(3))
    o = torch.zeros_like(q)
    fn = lambda: kern(q, k, v, o, 2)
    bu.prewarm(fn, 0.3)
    iters = 200 if N_ <= 2048 else 50  # ONE event-timed region (as the headline): >= 200 launches of the ~40-70 us kernels
    ms = bu.time_region_events(fn, iters)
    flops = bu.mha_flops_conventional(B_, H_, N_, D)
    ach = flops / (ms * 1e-3) * 1e-12
    fam = kernel_desc.split("<")[0]  # counters must come from THIS kernel family
    ksub = {"fa2_fwd_m16": "fa2_fwd_m16_pair_kernel", "fa2_fwd_m16x64r": "fa2_fwd_m16x_kernel"}.get(fam, fam + "_kernel") if fam else ""
    busy, src = bu.pmc_value(profiles, pmc_file, "mfma_busy_frac", ksub)
    traffic, _ = bu.pmc_value(profiles, pmc_file, "l2_fabric_traffic_bytes_per_launch", ksub)
    return {"bound": "mfma", "achieved": round(ach, 2), "peak": bu.PEAK_FP16_MFMA_TFLOPS, "unit": "TFLOP/s",
            "frac": round(ach / bu.PEAK_FP16_MFMA_TFLOPS, 4), "traffic": round(traffic) if traffic else None,
            "mfma_busy": round(busy, 4) if busy else None, "pmc_source": src, "traffic_measured_in_this_run": False,
            "traffic_is": bu.TRAFFIC_IS, "shape": [B_, H_, N_, D],
            "avg_launch_ms": round(ms, 5), "algorithmic_flops_per_launch": flops,
            "algorithmic_bytes_per_launch": 4.0 * B_ * H_ * N_ * D * 2,
            "tflops_ref_model": round(bu.get_mha_tflops(B_, H_, N_, D, ms * 1e-3), 2)}, (q, k, v, o)


def main():
    args = parse()
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available():
        sys.exit("bench.py needs a GPU (the kernel path has no CPU fallback)")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    dist = None
    if world > 1:
        # replicas only (north_star: no RCCL on this path): the process group exists for the barrier and the
        # max-over-ranks of the step time, on CPU tensors over gloo -- the same code tests/test_multiproc_gloo.py runs
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("gloo", rank=rank, world_size=world)

    pkg = entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu
    hg = pkg.hgemm_lib()
    profiles = os.path.join(ROOT, "profiles")

    M = N = K = args.mnk
    torch.manual_seed(1234 + rank)
    a = torch.randn(M, K, dtype=torch.half, device=dev)
    b = torch.randn(K, N, dtype=torch.half, device=dev)
    c = torch.zeros(M, N, dtype=torch.half, device=dev)
    stride = bu.make_block_swizzle_stride(N, K)
    kernel = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem

    def step():
        kernel(a, b, c, args.stages, True, stride)

    def barrier():
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()

    # Time-based pre-warm OUTSIDE the timed region: the chip clocks to its power budget, and the first milliseconds
    # of a launch burst run on a higher (un-sustained) or lower (ramping) clock -- a 20-step run read 12-14 % low in
    # round 1. After >= 0.5 s of back-to-back launches the 20-step and the 500-step numbers agree.
    bu.prewarm(step, args.prewarm)
    settle_hist = bu.settle(step, max(20, min(args.steps, 200)))
    # --launch graph: the K timed steps as ONE hipGraph (K kernel nodes, captured from the same Python calls). Measured
    # in round 2 (profiles/r02_bench_launch_modes.log): the graph costs +26 us per kernel node (1141 vs 1442 TF at 20
    # steps) while eager enqueue takes 9.5 us of host time per 94 us kernel, so eager is the default.
    graph, launch_mode = None, "eager"
    if args.launch == "graph":
        try:
            graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(graph):
                for _ in range(args.steps):
                    step()
            torch.cuda.synchronize()
            launch_mode = "hipGraph of %d kernel launches" % args.steps
        except Exception as e:  # noqa: BLE001 -- capture not available: measure eagerly and say so
            graph, launch_mode = None, "eager (graph capture failed: %s)" % str(e)[:80]
            torch.cuda.synchronize()

    def run_steps():
        if graph is not None:
            graph.replay()
        else:
            for _ in range(args.steps):
                step()

    for _ in range(args.warmup):
        step()
    if graph is not None:
        graph.replay()  # untimed: first replay uploads the graph
    barrier()
    stream = torch.cuda.current_stream()
    ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    ev0.record(stream)
    run_steps()
    ev1.record(stream)
    t_enq = time.perf_counter() - t0  # host time to hand the K steps to the GPU
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = bu.max_over_ranks(elapsed, dist, None)
    ev_ms = ev0.elapsed_time(ev1) / args.steps  # HIP events on the launch stream, over the SAME timed region

    flops = bu.hgemm_flops(M, N, K)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * flops / (elapsed / args.steps) * 1e-12

    # ---- roofline of the dominant kernel: ONE timing source with `value` (the timed region above)
    achieved = flops / (ev_ms * 1e-3) * 1e-12
    kernel_desc = pkg.manifest.describe(bu.HEADLINE_HGEMM_NAME, (M, N, K), 2)
    ksub = kernel_desc.split("<")[0] + "_kernel"  # device kernel family the PMC summary must have been taken on
    traffic, traffic_src = bu.pmc_value(profiles, "pmc_hgemm", "l2_fabric_traffic_bytes_per_launch", ksub) if M == 4096 else (None, None)
    busy, _ = bu.pmc_value(profiles, "pmc_hgemm", "mfma_busy_frac", ksub) if M == 4096 else (None, None)
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": bu.PEAK_FP16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / bu.PEAK_FP16_MFMA_TFLOPS, 4),
                "traffic": round(traffic) if traffic else None, "traffic_source": traffic_src,
                "traffic_measured_in_this_run": False,  # PMC counters cannot be collected from inside the timed process: `traffic` and
                "traffic_is": bu.TRAFFIC_IS,            # `mfma_busy` are REPLAYED from the committed rocprofv3 pass named in traffic_source
                "mfma_busy": round(busy, 4) if busy else None,
                "mfma_busy_note": "SQ_VALU_MFMA_BUSY_CYCLES / GRBM_GUI_ACTIVE of the rocprofv3 counter pass, where the launch takes "
                                  "more cycles than un-profiled (can read below frac); MFMA cycles per launch are fixed by the work",
                "kernel": kernel_desc, "avg_launch_ms": round(ev_ms, 5),
                "timing": "HIP events on the launch stream around the %d timed steps (same region as value)" % args.steps,
                "algorithmic_flops_per_launch": flops, "algorithmic_bytes_per_launch": bu.hgemm_bytes(M, N, K)}

    out = {
        "metric": "HGEMM fp16 TFLOPS at M=N=K=%d" % M, "value": round(value, 2), "unit": "TFLOPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic (seeded torch.randn fp16, random-init operands)",
        "config": {"workload": "HGEMM fp16 NN M=N=K=%d, stages=%d, block swizzle stride %d (BASELINE config C3)"
                               % (M, args.stages, stride),
                   "kernel": "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", "parallelism": "replicas x%d" % world,
                   "prewarm_s": args.prewarm, "launch": launch_mode,
                   "settle_batches": len(settle_hist),  # untimed steady-state check: first and last three batches
                   "settle_ms_per_step": [round(x, 5) for x in (settle_hist if len(settle_hist) <= 6 else settle_hist[:3] + settle_hist[-3:])],
                   "host_enqueue_ms_per_step": round(t_enq / args.steps * 1e3, 5)},
        "pct_of_fp16_mfma_peak": round(100.0 * achieved / bu.PEAK_FP16_MFMA_TFLOPS, 2),
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_extras:  # side measurements only in the single-GPU run
        extras = {"timing": "side rows: 0.2-0.3 s pre-warm, then ONE event-timed region of back-to-back launches (ours and "
                            "the vendor's alike; no best-of-N)"}

        def side_ms(fn, iters):
            bu.prewarm(fn, 0.2)
            return bu.time_region_events(fn, 2 * iters)

        try:  # vendor row (rocBLAS) on the same operands
            hg.init_cublas_handle()
            ms = side_ms(lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c), 50)
            extras["rocblas_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            extras["pct_of_rocblas"] = round(100.0 * achieved / extras["rocblas_tflops"], 2)
            bt = bu.as_col_major(b)
            ms = side_ms(lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c), 50)
            extras["rocblas_tn_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
            ms = side_ms(lambda: tn(a, bt, c, args.stages, True, stride), 50)
            extras["hgemm_tn_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            hg.destroy_cublas_handle()
            # second vendor row: hipBLASLt (BASELINE.md's C3 target reads "rocBLAS/hipBLASLt"), heuristic's top algorithm
            try:
                lt = pkg.load("hgemm_vendor_lt")
                ms = side_ms(lambda: lt.cln_hgemm_hipblaslt_nn(a, b, c), 50)
                extras["hipblaslt_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
                extras["pct_of_hipblaslt"] = round(100.0 * achieved / extras["hipblaslt_tflops"], 2)
                ms = side_ms(lambda: lt.cln_hgemm_hipblaslt_tn(a, bt, c), 50)
                extras["hipblaslt_tn_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            except Exception as e:  # noqa: BLE001
                extras["hipblaslt_error"] = str(e)[:200]
            del bt
        except Exception as e:  # the vendor row is a comparison, never the product
            extras["rocblas_error"] = str(e)[:200]
        try:  # FA2 forward is the other half of the BASELINE metric: roofline objects for C4 (D=64) and C5 (D=512)
            import torch.nn.functional as F
            fa = pkg.flash_attn_lib()
            sq, tq = fa.flash_attn_mma_stages_split_q_shared_qkv, fa.flash_attn_mma_stages_split_q_tiling_qkv
            # north_star's attention target is "FA2 fwd D=64 >= FlashAttention-2-ROCm". The Python packages (`flash_attn`, `aiter`)
            # are not in the image; the kernels they dispatch to are: AMD's ck_tile FMHA forward, instantiated from the image's
            # /opt/rocm/include/ck_tile into the vendor library (csrc/fa2_vendor_ck.hip) -> `ck_tile_fmha` in each roofline object
            # (D = 64 / 128). torch SDPA (AOTriton kernels under every backend here) stays beside it.
            try:
                ck = pkg.load("fa2_vendor_ck").cln_fa2_ck_tile_fwd
                extras["fa2_rocm_comparator"] = ("ck_tile FMHA forward (the kernel family FlashAttention-2-ROCm / aiter dispatch to), compiled "
                                                 "from the image's ck_tile headers: `ck_tile_fmha` rows; flash_attn / aiter themselves are not "
                                                 "importable in this image; torch SDPA (AOTriton) rows beside it")
            except Exception as e:
                ck = None
                extras["fa2_rocm_comparator"] = "absent: torch SDPA/AOTriton proxy (ck_tile row failed to load: %s)" % str(e)[:120]
            for key, kern, shape, pmc in (("roofline_fa2_c4_d64", sq, (4, 8, 2048, 64), "pmc_fa_d64"),
                                          ("roofline_fa2_d128", sq, (4, 8, 2048, 128), "pmc_fa_d128"),
                                          ("roofline_fa2_c5_d512", tq, (1, 32, 4096, 512), "pmc_fa_d512")):
                desc = pkg.manifest.describe(kern.__name__, shape, 2)
                r, (q, k, v, o) = fa_roofline(kern, shape, pmc, dev, bu, profiles, desc)
                r["kernel"] = desc
                # the FlashAttention-2-ROCm row available on the box is torch SDPA (the `flash_attn` package is not in
                # the image): each backend forced in turn, so the row says WHICH implementation it is
                r["torch_sdpa"] = bu.sdpa_rows(q, k, v, side_ms)
                if ck is not None and shape[3] in (64, 128):
                    try:
                        flops = bu.mha_flops_conventional(*shape)
                        row = {"async_pipeline_tflops": round(flops / (side_ms(lambda: ck(q, k, v, o, 0), 50) * 1e-3) * 1e-12, 2)}
                        if shape[3] == 128:
                            row["v3_gfx950_tflops"] = round(flops / (side_ms(lambda: ck(q, k, v, o, 3), 50) * 1e-3) * 1e-12, 2)
                        row["ours_over_best"] = round(r["achieved"] / max(row.values()), 3)
                        r["ck_tile_fmha"] = row
                    except Exception as e:
                        r["ck_tile_fmha"] = {"error": str(e)[:160]}
                out[key] = r
                del q, k, v, o
            for tag, (B_, H_, N_, D) in (("fa2_fwd_d64_large", (1, 48, 8192, 64)), ("fa2_fwd_d128_large", (2, 32, 4096, 128)),
                                         ("fa2_fwd_d256", (2, 32, 4096, 256)), ("fa2_fwd_d768", (1, 16, 4096, 768)),
                                         ("fa2_fwd_d1024", (1, 16, 4096, 1024))):
                kern = sq if D <= 256 else tq
                q, k, v = (torch.randn(B_, H_, N_, D, dtype=torch.half, device=dev) for _ in range(3))
                o = torch.zeros_like(q)
                ms = side_ms(lambda: kern(q, k, v, o, 2), 10)
                extras[tag] = {"shape": [B_, H_, N_, D], "ms": round(ms, 5),
                               "tflops_4bhn2d": round(bu.mha_flops_conventional(B_, H_, N_, D) / (ms * 1e-3) * 1e-12, 2),
                               "kernel": pkg.manifest.describe(kern.__name__, (B_, H_, N_, D), 2)}
                del q, k, v, o
        except Exception as e:
            extras["fa2_error"] = str(e)[:300]
        out["extras"] = extras
        out["cpu_baseline"] = cpu_baseline(a, b, M, N, K)
        # ---- counters from THIS box: the same kernel for a dozen launches under rocprofv3 --pmc, in child processes after the timed
        # region; on any trouble the replayed values above stay and `pmc.status` says why
        if M == 4096 and not args.no_pmc:
            pmc = bu.collect_pmc_here(ROOT)
            out["pmc"] = pmc
            if pmc["traffic"] is not None:
                roofline["traffic"] = round(pmc["traffic"])
                roofline["traffic_source"] = "rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE child passes of this run"
                roofline["traffic_measured_in_this_run"] = True
            if pmc["mfma_busy"] is not None:
                roofline["mfma_busy"] = round(pmc["mfma_busy"], 4)
        # ---- every other BASELINE.json config and the bandwidth kernels, each with its roofline fraction and the
        # reference's torch path on the host cores: bench_configs.py (repository root). Released first: the operands above.
        del a, b, c
        torch.cuda.empty_cache()
        if not args.no_configs:
            import bench_configs as bc  # repository root, beside this file
            orc = entry.load_oracle()  # cpu_baseline legs only
            cfg = {}
            for key, fn in (("c1_elementwise_add_f32", lambda: bc.config_c1(dev, orc)),
                            ("hgemm", lambda: bc.hgemm_config_rows(pkg, dev, orc)),
                            ("hgemm_policy", lambda: bc.hgemm_policy_rows(pkg, dev)),
                            ("fa2_stages", lambda: bc.fa_stage_rows(pkg, dev)),
                            ("bandwidth", lambda: bc.bandwidth_rows(dev, orc)),
                            ("next_rows", lambda: bc.next_rows(pkg, dev, orc))):
                t0 = time.perf_counter()
                try:
                    cfg[key] = fn()
                except Exception as e:  # noqa: BLE001 -- a side row never takes the headline line down with it
                    cfg[key] = {"error": "%s: %s" % (type(e).__name__, str(e)[:300])}
                cfg.setdefault("wall_s", {})[key] = round(time.perf_counter() - t0, 1)
            out["configs"] = cfg
    if rank == 0:
        emit(out)
    if dist is not None:
        dist.destroy_process_group()


MAX_LINE = 4096  # bytes of the final stdout line (the driver's stdout tail is 8 KB; r03's 5.7 KB parsed, r04's 23 KB did not)


def _clip(x, n):
    """strings clipped to n characters (kernel descriptions and samples are free text)"""
    if isinstance(x, str) and len(x) > n:
        return x[:n - 3] + "..."
    return x


def _pick(d, keys, clip=160):
    return {k: _clip(d[k], clip) for k in keys if isinstance(d, dict) and k in d}


def headline_line(out, detail_file="bench_detail.json"):
    """The compact final stdout line built from the full result `out`: the contract keys, config{workload, kernel,
    parallelism, launch}, the headline kernel's roofline, cpu_baseline, and a digest of the side rows. Always
    < MAX_LINE bytes: free-text fields are clipped, and the digest is dropped key by key if it ever would not fit."""
    line = _pick(out, ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling",
                       "vs_baseline", "dtype", "data"), 100)
    line["config"] = _pick(out.get("config", {}), ("workload", "kernel", "parallelism", "launch", "host_enqueue_ms_per_step"), 120)
    line["roofline"] = _pick(out.get("roofline", {}),
                             ("bound", "achieved", "peak", "unit", "frac", "traffic", "traffic_source", "traffic_measured_in_this_run",
                              "mfma_busy", "kernel", "avg_launch_ms", "algorithmic_flops_per_launch", "algorithmic_bytes_per_launch"), 140)
    cb = out.get("cpu_baseline")
    if isinstance(cb, dict):
        line["cpu_baseline"] = _pick(cb, ("value", "unit", "cores", "kind", "sample"), 200)
        if isinstance(cb.get("fa2_fwd_c4"), dict):
            line["cpu_baseline"]["fa2_fwd_c4"] = _pick(cb["fa2_fwd_c4"], ("value", "unit", "cores", "kind", "error"), 60)
    digest = {}
    ex = out.get("extras", {}) if isinstance(out.get("extras"), dict) else {}
    for k in ("rocblas_tflops", "pct_of_rocblas", "rocblas_tn_tflops", "hgemm_tn_tflops", "hipblaslt_tflops", "pct_of_hipblaslt",
              "hipblaslt_tn_tflops"):
        if k in ex:
            digest[k] = ex[k]
    fa = {}
    for short, key in (("c4_d64", "roofline_fa2_c4_d64"), ("d128", "roofline_fa2_d128"), ("c5_d512", "roofline_fa2_c5_d512")):
        r = out.get(key)
        if isinstance(r, dict):
            row = {"tflops": r.get("achieved"), "frac": r.get("frac"), "ms": r.get("avg_launch_ms")}
            ck = r.get("ck_tile_fmha")
            if isinstance(ck, dict) and "ours_over_best" in ck:
                row["x_ck_tile_fmha"] = ck["ours_over_best"]
            fa[short] = row
    for short, key in (("d768", "fa2_fwd_d768"), ("d1024", "fa2_fwd_d1024")):
        if isinstance(ex.get(key), dict):
            t = ex[key].get("tflops_4bhn2d")
            fa[short] = {"tflops": t, "frac": round(t / 2500.0, 4) if t else None}
    if fa:
        digest["fa2_fwd"] = fa
        # the north_star names FlashAttention-2-ROCm; what stands beside our kernel is NOT that package
        digest["fa2_comparator"] = "ck_tile FMHA fwd compiled from the image's headers (the kernels FlashAttention-2-ROCm / aiter dispatch to) + torch SDPA; the flash_attn / aiter packages are not importable in this image (no network)"
    for k in ("fa2_error", "rocblas_error"):
        if k in ex:
            digest[k] = _clip(ex[k], 100)
    try:  # the f32 matrix-core sgemm rung beside its two vendor rows (configs.next_rows.sgemm_4096 in the detail file)
        sg = out["configs"]["next_rows"]["sgemm_4096"]["sgemm_wmma_m16n16k8_mma4x2_warp2x4_stages"]
        digest["sgemm_4096"] = {"tflops": sg["tflops"], "frac_of_157_tf": sg["frac_of_f32_mfma_peak"], "x_rocblas": sg.get("x_rocblas_sgemm"),
                                "x_hipblaslt": sg.get("x_torch_matmul_hipblaslt")}
    except (KeyError, TypeError):
        pass
    if "pmc" in out:
        digest["pmc"] = _pick(out["pmc"], ("status", "seconds"), 100)
    digest["detail_file"] = detail_file
    line["digest"] = digest
    s = json.dumps(line, separators=(", ", ": "))
    for k in ("sgemm_4096", "fa2_comparator", "fa2_fwd", "hipblaslt_tn_tflops", "hipblaslt_tflops", "pct_of_hipblaslt", "hgemm_tn_tflops", "rocblas_tn_tflops"):
        if len(s) < MAX_LINE:
            break
        digest.pop(k, None)
        s = json.dumps(line, separators=(", ", ": "))
    if len(s) >= MAX_LINE:  # cannot happen with the clips above; never let a long line take the headline down again
        line.pop("digest", None)
        line["roofline"].pop("kernel", None)
        s = json.dumps(line, separators=(", ", ": "))
    assert len(s) < MAX_LINE, len(s)
    return s


def detail_rows(out):
    """Short human-readable rows (one per measured kernel/config, <= 200 characters each) for the stdout lines that
    precede the headline line -- the counterpart of the reference's one-row-per-kernel prints (hgemm.py:142-168)."""
    rows = []

    def add(tag, txt):
        rows.append(("# %-34s %s" % (tag, txt))[:200])

    for key in ("roofline_fa2_c4_d64", "roofline_fa2_d128", "roofline_fa2_c5_d512"):
        r = out.get(key)
        if isinstance(r, dict):
            add(key, "%s TF frac %s ms %s shape %s" % (r.get("achieved"), r.get("frac"), r.get("avg_launch_ms"), r.get("shape")))
    ex = out.get("extras", {}) if isinstance(out.get("extras"), dict) else {}
    for k, v in ex.items():
        if isinstance(v, (int, float)):
            add(k, v)
        elif isinstance(v, dict) and "tflops_4bhn2d" in v:
            add(k, "%s TF ms %s shape %s" % (v["tflops_4bhn2d"], v.get("ms"), v.get("shape")))
    cfg = out.get("configs", {}) if isinstance(out.get("configs"), dict) else {}

    def walk(prefix, node):
        if isinstance(node, list):
            for it in node:
                walk(prefix, it)
        elif isinstance(node, dict):
            if "error" in node and len(node) == 1:
                add(prefix, "error: %s" % node["error"])
            elif any(k in node for k in ("gbps", "tflops", "us_per_launch", "ms")) and not any(isinstance(v, (dict, list)) and v and k != "shape" and k != "mnk" for k, v in node.items() if k not in ("cpu_baseline", "roofline", "yardstick")):
                name = node.get("kernel") or node.get("name") or node.get("tag") or ""
                bits = []
                for k in ("shape", "mnk", "dtype", "stages", "swizzle", "us_per_launch", "us_64_row_launch", "ms", "gbps", "frac_of_8TBs", "tflops", "frac", "frac_of_peak", "frac_of_f32_mfma_peak", "x_rocblas_sgemm", "gelem_per_s", "pct_of_rocblas"):
                    if k in node:
                        bits.append("%s=%s" % (k, node[k]))
                if isinstance(node.get("yardstick"), dict) and "ours_over_yardstick" in node["yardstick"] and not isinstance(node["yardstick"]["ours_over_yardstick"], dict):
                    bits.append("x_yardstick=%s" % node["yardstick"]["ours_over_yardstick"])
                add(prefix + ":" + str(name)[:60], " ".join(bits))
            else:
                for k, v in node.items():
                    if k != "wall_s":
                        walk(prefix + "." + k if prefix else k, v)

    walk("", cfg)
    return rows


def emit(out):
    """Write the full result to bench_detail.json, print the short rows, then the compact headline line LAST."""
    blob = json.dumps(out, indent=1)
    paths = [os.path.join(ROOT, "bench_detail.json")]
    god = os.path.join(ROOT, "gpurun_out")
    if os.path.isdir(god):
        paths.append(os.path.join(god, "bench_detail.json"))
    for p in paths:
        try:
            with open(p, "w") as f:
                f.write(blob)
        except OSError:
            pass
    try:
        for r in detail_rows(out):
            print(r)
    except Exception as e:  # noqa: BLE001 -- the rows are a convenience; the last line is the contract
        print("# detail rows unavailable: %s" % str(e)[:120])
    sys.stdout.flush()
    print(headline_line(out), flush=True)


def cpu_baseline(a, b, M, N, K):
    """The reference's own torch path (`torch.matmul` on fp16 tensors, hgemm.py:420-421) on the host cores,
    on a bounded sample of the same workload: the first `rows` rows of A against the full B."""
    orc = entry.load_oracle()
    b_c = b.cpu()
    rows = 16  # probe run (~1-2 s: fp16 matmul on CPU has no fast path), then one sample sized for ~10 s
    t0 = time.perf_counter()
    orc.hgemm_fp16_path(a[:rows].cpu(), b_c)
    dt = time.perf_counter() - t0
    target = int(rows * 10.0 / max(dt, 1e-3))
    target = max(16, min(M, (target // 16) * 16))
    if target > rows:
        rows = target
        t0 = time.perf_counter()
        orc.hgemm_fp16_path(a[:rows].cpu(), b_c)
        dt = time.perf_counter() - t0
    res = {"value": round(2.0 * rows * N * K / dt * 1e-12, 5), "unit": "TFLOPS", "cores": torch.get_num_threads(),
           "kind": "port", "sample": "torch.matmul fp16 on CPU, first %d of %d rows of A x full B, 1 iteration "
                                     "(%.2f s); host cpu_count=%d" % (rows, M, dt, os.cpu_count() or 0)}
    # the FA2 half of the metric: the reference's own unfused torch attention (flash_attn_mma.py:384-398) on the host,
    # config C4 shape, a bounded number of its 32 heads
    try:
        g = torch.Generator().manual_seed(7)
        q, k, v = (torch.randn(1, 1, 2048, 64, generator=g).half() for _ in range(3))
        t0 = time.perf_counter()
        orc.unfused_standard_attn(q, k, v)
        dt1 = time.perf_counter() - t0
        heads = max(1, min(32, int(5.0 / max(dt1, 1e-3))))
        q, k, v = (torch.randn(1, heads, 2048, 64, generator=g).half() for _ in range(3))
        t0 = time.perf_counter()
        orc.unfused_standard_attn(q, k, v)
        dt = time.perf_counter() - t0
        res["fa2_fwd_c4"] = {"value": round(4.0 * heads * 2048 * 2048 * 64 / dt * 1e-12, 5), "unit": "TFLOPS (4BHN^2D)",
                             "cores": torch.get_num_threads(), "kind": "port",
                             "sample": "unfused torch attention fp16 on CPU, %d of the 32 heads of [4,8,2048,64], "
                                       "1 iteration (%.2f s)" % (heads, dt)}
    except Exception as e:
        res["fa2_fwd_c4"] = {"error": str(e)[:200]}
    return res


if __name__ == "__main__":
    main()
