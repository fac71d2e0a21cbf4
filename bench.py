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
Prints ONE JSON line on rank 0.
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
    p.add_argument("--steps", type=int, default=500)
    p.add_argument("--warmup", type=int, default=100)
    p.add_argument("--mnk", type=int, default=4096)
    p.add_argument("--stages", type=int, default=2)
    p.add_argument("--no-extras", action="store_true", help="skip FA2 / rocBLAS / CPU baseline side measurements")
    return p.parse_args()


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
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)

    pkg = entry.load_package()
    from cuda_learn_notes_amd import bench_utils as bu
    hg = pkg.hgemm_lib()

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
        if dist is not None:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    elapsed = bu.max_over_ranks(elapsed, dist, dev)

    flops = bu.hgemm_flops(M, N, K)
    ms_per_step = elapsed / args.steps * 1e3
    value = world * flops / (elapsed / args.steps) * 1e-12

    # ---- roofline of the dominant kernel: HIP events on the launch stream, per launch
    ev_ms, ev_min, _ = bu.time_call_events(step, 3, max(10, min(args.steps, 50)))
    achieved = flops / (ev_ms * 1e-3) * 1e-12
    traffic, traffic_src = bu.pmc_traffic(os.path.join(ROOT, "profiles"), "hgemm", M)
    roofline = {"bound": "mfma", "achieved": round(achieved, 2), "peak": bu.PEAK_FP16_MFMA_TFLOPS,
                "unit": "TFLOP/s", "frac": round(achieved / bu.PEAK_FP16_MFMA_TFLOPS, 4), "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel": bu.HEADLINE_HGEMM_KERNEL, "avg_launch_ms": round(ev_ms, 5),
                "min_launch_ms": round(ev_min, 5), "algorithmic_flops_per_launch": flops,
                "algorithmic_bytes_per_launch": bu.hgemm_bytes(M, N, K)}

    out = {
        "metric": "HGEMM fp16 TFLOPS at M=N=K=%d" % M, "value": round(value, 2), "unit": "TFLOPS",
        "n_gpus": world, "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(ms_per_step, 5),
        "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "f16",
        "data": "synthetic (seeded torch.randn fp16, random-init operands)",
        "config": {"workload": "HGEMM fp16 NN M=N=K=%d, stages=%d, block swizzle stride %d (BASELINE config C3)"
                               % (M, args.stages, stride),
                   "kernel": "hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem", "parallelism": "replicas x%d" % world},
        "pct_of_fp16_mfma_peak": round(100.0 * achieved / bu.PEAK_FP16_MFMA_TFLOPS, 2),
        "roofline": roofline,
    }

    if rank == 0 and world == 1 and not args.no_extras:  # side measurements only in the single-GPU run
        extras = {"timing": "side rows: mean of the better of two event-timed rounds (first-use rounds read 5-15% low)"}

        def side_ms(fn, warm, iters):  # every side row, ours and the vendor's alike
            return min(bu.time_call_events(fn, warm, iters)[0], bu.time_call_events(fn, 2, iters)[0])

        try:  # vendor row (rocBLAS) on the same operands
            hg.init_cublas_handle()
            ms = side_ms(lambda: hg.hgemm_cublas_tensor_op_nn(a, b, c), 5, 20)
            extras["rocblas_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            extras["pct_of_rocblas"] = round(100.0 * achieved / extras["rocblas_tflops"], 2)
            bt = bu.as_col_major(b)
            ms = side_ms(lambda: hg.hgemm_cublas_tensor_op_tn(a, bt, c), 5, 20)
            extras["rocblas_tn_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            tn = hg.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem_tn_swizzle_x4
            ms = side_ms(lambda: tn(a, bt, c, args.stages, True, stride), 5, 20)
            extras["hgemm_tn_tflops"] = round(flops / (ms * 1e-3) * 1e-12, 2)
            del bt
            hg.destroy_cublas_handle()
        except Exception as e:  # the vendor row is a comparison, never the product
            extras["rocblas_error"] = str(e)[:200]
        try:  # FA2 forward, config C4 (D=64) and the D=128 sibling
            fa = pkg.flash_attn_lib()
            for tag, (B_, H_, N_, D) in (("fa2_fwd_d64", (4, 8, 2048, 64)), ("fa2_fwd_d128", (4, 8, 2048, 128)),
                                         ("fa2_fwd_d64_large", (1, 48, 8192, 64)),
                                         ("fa2_fwd_d128_large", (2, 32, 4096, 128)),
                                         ("fa2_fwd_d256", (2, 32, 4096, 256)),
                                         ("fa2_fwd_d512_c5", (1, 32, 4096, 512))):
                q, k, v = (torch.randn(B_, H_, N_, D, dtype=torch.half, device=dev) for _ in range(3))
                o = torch.zeros_like(q)
                kern = (fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256
                        else fa.flash_attn_mma_stages_split_q_tiling_qkv)
                fn = lambda: kern(q, k, v, o, 2)
                ms = side_ms(fn, 5, 30 if N_ <= 2048 else 10)
                row = {"shape": [B_, H_, N_, D], "ms": round(ms, 5),
                       "tflops_ref_model": round(bu.get_mha_tflops(B_, H_, N_, D, ms * 1e-3), 2),
                       "tflops_4bhn2d": round(bu.mha_flops_conventional(B_, H_, N_, D) / (ms * 1e-3) * 1e-12, 2)}
                try:  # the FlashAttention-2-ROCm row available on the box: torch SDPA (reference prints it too,
                    # flash_attn_mma.py:391-398)
                    import torch.nn.functional as F
                    ms2 = side_ms(lambda: F.scaled_dot_product_attention(q, k, v), 5, 30 if N_ <= 2048 else 10)
                    row["torch_sdpa_tflops_4bhn2d"] = round(
                        bu.mha_flops_conventional(B_, H_, N_, D) / (ms2 * 1e-3) * 1e-12, 2)
                except Exception as e:
                    row["torch_sdpa_error"] = str(e)[:120]
                extras[tag] = row
                del q, k, v, o
        except Exception as e:
            extras["fa2_error"] = str(e)[:200]
        out["extras"] = extras
        if world == 1:
            out["cpu_baseline"] = cpu_baseline(a, b, M, N, K)
    if rank == 0:
        print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


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
    return {"value": round(2.0 * rows * N * K / dt * 1e-12, 5), "unit": "TFLOPS", "cores": torch.get_num_threads(),
            "kind": "port", "sample": "torch.matmul fp16 on CPU, first %d of %d rows of A x full B, 1 iteration "
                                      "(%.2f s); host cpu_count=%d" % (rows, M, dt, os.cpu_count() or 0)}


if __name__ == "__main__":
    main()
