"""Generate the golden fixtures that pin oracle/oracle.py to the REFERENCE's own Python code.

Runs only where /root/reference exists (the build container). The reference bench scripts cannot be
imported (module level JIT-builds CUDA), so the pure-Python functions are extracted by AST and
executed on CPU on seeded inputs:
  kernels/flash-attn/flash_attn_mma.py : unfused_standard_attn (:384-388), get_mha_tflops (:191-222)
  kernels/layer-norm/layer_norm.py     : naive_layer_norm (:25-29)
  kernels/rms-norm/rms_norm.py         : naive_rms_norm (:26-31)
  kernels/rope/rope.py                 : naive_rope (:68-88)   (its `.cuda()` is patched to identity)
  kernels/hgemm/hgemm.py               : make_block_swizzle_stride (:71-81)
  kernels/hgemm/tools/utils.py         : as_col_major (:135-140)
plus the stock torch ops the scripts use as check columns (matmul, add, sum, softmax).
Outputs: tests/golden/*.npz (inputs are regenerated from the stored seed; only outputs are stored).
"""
import ast
import math
import os
import sys

import numpy as np
import torch
import torch.nn.functional as F

REF = "/root/reference/kernels"
OUT = os.path.dirname(os.path.abspath(__file__))


def extract(path, names):
    src = open(path).read()
    tree = ast.parse(src)
    ns = {"torch": torch, "math": math, "F": F, "Tensor": torch.Tensor, "Tuple": tuple, "Optional": None}
    import typing
    ns.update({"Tuple": typing.Tuple, "Optional": typing.Optional})
    for node in tree.body:
        if isinstance(node, ast.FunctionDef) and node.name in names:
            node.decorator_list = []
            mod = ast.Module(body=[node], type_ignores=[])
            exec(compile(mod, path, "exec"), ns)
    return [ns[n] for n in names]


def seeded(seed, *shape, dtype=torch.float32):
    g = torch.Generator().manual_seed(seed)
    return torch.randn(*shape, generator=g).to(dtype)


def main():
    if not os.path.isdir(REF):
        sys.exit("reference tree not present; fixtures can only be regenerated in the build container")
    torch.set_grad_enabled(False)
    torch.Tensor.cuda = lambda self, *a, **k: self  # naive_rope calls .cuda() on an intermediate

    (attn, mha_tflops) = extract(REF + "/flash-attn/flash_attn_mma.py", ["unfused_standard_attn", "get_mha_tflops"])
    (naive_ln,) = extract(REF + "/layer-norm/layer_norm.py", ["naive_layer_norm"])
    (naive_rms,) = extract(REF + "/rms-norm/rms_norm.py", ["naive_rms_norm"])
    (naive_rope,) = extract(REF + "/rope/rope.py", ["naive_rope"])
    (swz_stride,) = extract(REF + "/hgemm/hgemm.py", ["make_block_swizzle_stride"])
    (col_major,) = extract(REF + "/hgemm/tools/utils.py", ["as_col_major"])

    # --- attention: fp32 run of the reference function on fp16-rounded inputs
    B, H, N, D = 2, 2, 128, 64
    q, k, v = (seeded(100 + i, B, H, N, D, dtype=torch.float16) for i in range(3))
    o = attn(q.float(), k.float(), v.float())
    np.savez_compressed(os.path.join(OUT, "attn_b2h2n128d64.npz"), seeds=[100, 101, 102], shape=[B, H, N, D],
                        out=o.numpy())
    flops = {"C4": mha_tflops(4, 8, 2048, 64, 1.0), "C5": mha_tflops(1, 32, 4096, 512, 1.0),
             "C4_mm": mha_tflops(4, 8, 2048, 64, 1.0, only_matmul=True)}
    # --- hgemm: the script's own check column, torch.matmul on fp16
    a, b = seeded(200, 128, 256, dtype=torch.float16), seeded(201, 256, 192, dtype=torch.float16)
    c = torch.matmul(a, b)
    np.savez_compressed(os.path.join(OUT, "hgemm_128x192x256.npz"), seeds=[200, 201], shape=[128, 192, 256],
                        out=c.numpy(), b_col_major=col_major(b).numpy())
    strides = {"%d_%d" % (n, kk): swz_stride(n, kk) for n, kk in
               [(256, 256), (512, 512), (1024, 1024), (4096, 4096), (8192, 8192), (16384, 16384), (14848, 8448)]}
    # --- norms / rope / softmax / add / sum
    x = seeded(300, 64, 512)
    np.savez_compressed(os.path.join(OUT, "rows_64x512.npz"), seed=300, shape=[64, 512],
                        layer_norm=naive_ln(x, 1.0, 0.0).numpy(), layer_norm_gb=naive_ln(x, 1.5, -0.25).numpy(),
                        rms_norm=naive_rms(x, 1.0).numpy(), rms_norm_g=naive_rms(x, 0.5).numpy(),
                        rope=naive_rope(x).numpy(), softmax=torch.softmax(x, dim=1).numpy(),
                        softmax_global=torch.softmax(x.flatten(), dim=0).numpy(),
                        add=torch.add(x, seeded(301, 64, 512)).numpy(), sum=np.float64(torch.sum(x.double()).item()))
    import json
    with open(os.path.join(OUT, "scalars.json"), "w") as f:
        json.dump({"mha_tflops_at_1s": flops, "block_swizzle_stride": strides}, f, indent=1, sort_keys=True)
    print("golden fixtures written to", OUT)


if __name__ == "__main__":
    main()
