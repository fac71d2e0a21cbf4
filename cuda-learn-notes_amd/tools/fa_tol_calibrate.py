"""Round-6 calibration of the attention parity tolerance (the tests asserted 3e-3 where the kernels achieve < 1e-3 on N(0,1) inputs).
For the three precision families (plain names = fp16 pre-scaled Q at D <= 128, *_acc_f32 names = scores scaled in fp32, split-KV rung) and head dims
32 ... 1024, sequence lengths 64 ... 4096: max |O - O_fp32|, max |O_ref|, rms(O_ref) on N(0,1) inputs and on keys amplified 4x. Lines start with TOLCAL.
The bound the tests use (tests/test_gpu_flash_attn.py fa_tol) is fitted on this table with >= 1.5x margin."""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
names = {"plain": "flash_attn_mma_stages_split_q_shared_qkv", "acc_f32": "flash_attn_mma_stages_split_q_shared_qkv_acc_f32", "tiling": "flash_attn_mma_stages_split_q_tiling_qkv",
         "split_kv": "flash_attn_mma_stages_split_kv"}


def ref_attn(q, k, v):
    out = torch.empty(q.shape, dtype=torch.float32, device=q.device)
    sc = 1.0 / q.shape[-1] ** 0.5
    for b in range(q.shape[0]):
        for h in range(q.shape[1]):
            out[b, h] = torch.softmax((q[b, h].double() @ k[b, h].double().t()) * sc, dim=-1).float() @ v[b, h].float()
    return out


for D in (32, 64, 96, 128, 256, 512, 768, 1024):
    for N in (64, 128, 256, 512, 1024, 2048, 4096):
        if D >= 256 and N % 128:
            continue
        B, H = (1, 4) if D >= 512 else (2, 8)
        torch.manual_seed(N + D)
        q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
        for kind, kk in (("randn", k), ("keys x4", (k.float() * 4).half())):
            ref = ref_attn(q, kk, v)
            for fam, name in names.items():
                if (fam in ("plain", "acc_f32", "split_kv") and D > 256) or (fam == "split_kv" and D > 128):
                    continue
                o = torch.zeros_like(q)
                try:
                    getattr(fa, name)(q, kk, v, o, 2)
                    torch.cuda.synchronize()
                except RuntimeError as e:
                    print("TOLCAL %-8s D=%4d N=%4d %-8s unsupported (%s)" % (fam, D, N, kind, str(e)[:40]))
                    continue
                err = (o.float() - ref).abs()
                rel_row = (err.amax(dim=-1) / ref.abs().amax(dim=-1).clamp(min=1e-6)).max().item()
                print("TOLCAL %-8s D=%4d N=%4d %-8s max|err| %.3e  max|ref| %.3f  rms(ref) %.4f  err/rms %.4f  err/max|ref| %.5f  worst row err / row max %.5f"
                      % (fam, D, N, kind, err.max().item(), ref.abs().max().item(), ref.pow(2).mean().sqrt().item(), err.max().item() / ref.pow(2).mean().sqrt().item(),
                         err.max().item() / ref.abs().max().item(), rel_row), flush=True)
