"""Generate include/cln_amd.h from manifest.py (single source of truth for the exported surface)."""
import importlib.util
import os
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
PKG = os.path.dirname(HERE)
ROOT = os.path.dirname(PKG)

spec = importlib.util.spec_from_file_location("cln_manifest", os.path.join(PKG, "manifest.py"))
manifest = importlib.util.module_from_spec(spec)
spec.loader.exec_module(manifest)

PROTO = {
    "G3": "int {n}(const void* a, const void* b, void* c, int M, int N, int K, void* stream);",
    "G6": "int {n}(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, "
          "int swizzle_stride, void* stream);",
    "H0": "int {n}(void);",
    "S3": "int {n}(const void* a, const void* b, void* c, int M, int N, int K, void* stream);",
    "S6": "int {n}(const void* a, const void* b, void* c, int M, int N, int K, int stages, int swizzle, "
          "int swizzle_stride, void* stream);",
    "FA": "int {n}(const void* q, const void* k, const void* v, void* o, int B, int H, int N, int D, int stages, "
          "void* stream);",
    "P3": "int {n}(const void* a, const void* b, void* c, long long n, void* stream);",
    "R1": "int {n}(const void* a, void* y, long long n, void* stream);",
    "SG": "int {n}(const void* x, void* y, void* total_ws, long long n, void* stream);",
    "XY": "int {n}(const void* x, void* y, int S, int H, void* stream);",
    "LN": "int {n}(const void* x, void* y, float g, float b, int N, int K, void* stream);",
    "RN": "int {n}(const void* x, void* y, float g, int N, int K, void* stream);",
    "RP": "int {n}(const void* x, void* out, int seq_len, int hidden, int ref_quirk, void* stream);",
    "HI": "int {n}(const void* a, void* y, long long n, int nbins, void* stream);",
    "UN": "int {n}(const void* x, void* y, long long n, void* stream);",
    "D2": "int {n}(const void* a, const void* b, void* y, long long n, void* stream);",
    "GV": "int {n}(const void* a, const void* x, void* y, int M, int K, void* stream);",
    "TR": "int {n}(const void* x, void* y, int row, int col, void* stream);",
    "EM": "int {n}(const void* idx, const void* weight, void* out, long long n, int emb_size, int vocab, void* stream);",
}

GROUPS = [
    ("hgemm", "HGEMM: C[M,N] = A[M,K] * B, fp16, row-major; TN = B stored [N,K].\n"
              " * Replaces the free functions bound in reference kernels/hgemm/pybind/hgemm.cc:58-107\n"
              " * (`void f(torch::Tensor a, b, c[, int stages, bool swizzle, int swizzle_stride])`)."),
    ("hgemm_vendor", "Vendor row (libcln_amd_vendor.so, rocBLAS): reference kernels/hgemm/cublas/hgemm_cublas.cu:15-84,\n"
                     " * bindings :222-261 (init/destroy handle are process-global as in the reference)."),
    ("hgemm_vendor_lt", "Second vendor row (libcln_amd_vendor.so, hipBLASLt; csrc/hgemm_vendor_lt.hip). NOT reference names (hence cln_):\n"
                        " * BASELINE.md's C3 target reads \"rocBLAS/hipBLASLt\"; the reference's own vendor row is cuBLAS (hgemm_cublas.cu:15-84)."),
    ("fa2_vendor_ck", "Attention vendor row (libcln_amd_vendor.so; csrc/fa2_vendor_ck.hip). NOT a reference name: AMD's ck_tile FMHA forward\n"
                      " * kernels (the family FlashAttention-2-ROCm and aiter dispatch to; the reference times flash_attn_func,\n"
                      " * flash_attn_mma.py:10, :591). The `stages` slot carries the variant: 0 = async pipeline (D = 64 / 128), 3 = gfx950 v3 (D = 128)."),
    ("flash_attn", "FlashAttention-2 forward, fp16 [B,H,N,D]; *_swizzle_qkv of share_kv/share_qkv/tiling_qk take V as [B,H,D,N].\n"
                   " * Replaces reference kernels/flash-attn/pybind/flash_attn.cc:182-215\n"
                   " * (`void f(torch::Tensor Q, K, V, O, int stages)`)."),
    ("elementwise", "c = a + b. Replaces reference kernels/elementwise/elementwise.cu:163-177."),
    ("reduce", "y[0] = sum(a); y is fp32 (int32 for i8), 1 element, OVERWRITTEN by the launch (round 5: a self-resetting per-stream scratch word takes the\n"
               " * block partials, the last block moves the total into y; the reference binding's zeroed y works unchanged -- block_all_reduce.cu:737-738).\n"
               " * Replaces reference kernels/reduce/block_all_reduce.cu:734-813 (`torch::Tensor f(torch::Tensor x)`)."),
    ("softmax", "Softmax. softmax_f32[x4]: one distribution over all n elements, total_ws = 1 zeroed float.\n"
                " * *_per_token: x,y are [S,H]. Replaces reference kernels/softmax/softmax.cu:776-885."),
    ("layer_norm", "y = (x-mean)*rstd*g + b per row of x[N,K]. Replaces reference kernels/layer-norm/layer_norm.cu:732-814."),
    ("rms_norm", "y = x*rsqrt(mean(x^2)+1e-5)*g per row of x[N,K]. Replaces reference kernels/rms-norm/rms_norm.cu:457-813."),
    ("rope", "Rotary embedding on x[seq_len, hidden] fp32; ref_quirk=1 reproduces the reference kernels' integer-division\n"
             " * frequency (every pair rotated by `pos` radians). Replaces reference kernels/rope/rope.cu:80-120."),
    ("histogram", "y[a[i]] += 1 for int32 a[n]; y is int32[nbins], zeroed by the caller; values outside [0,nbins) are ignored.\n"
                  " * Replaces reference kernels/histogram/histogram.cu:56-80 (`torch::Tensor f(torch::Tensor a)`, nbins = max(a)+1)."),
    ("activation", "y = act(x), elementwise, n elements; relu / sigmoid / gelu(tanh) / swish / elu(alpha=1) / hardswish / hardshrink(0.5).\n"
                   " * Replaces the `void f(Tensor x, Tensor y)` bindings of reference kernels/<op>/<op>.cu (relu, sigmoid, gelu, swish, elu, hardswish, hardshrink)."),
    ("sgemm", "SGEMM: C[M,N] = A[M,K] * B[K,N], fp32 row-major. Replaces reference kernels/sgemm/sgemm.cu:495-640,\n"
              " * sgemm_async.cu bindings, sgemm_wmma_tf32_stage.cu:575-700 (TF32 rungs -> exact-f32 MFMA)."),
    ("sgemm_vendor", "Vendor SGEMM rows (libcln_amd_vendor.so): reference kernels/sgemm/sgemm_cublas.cu:80-120."),
    ("dot_product", "y[0] = sum(a*b), y fp32[1], OVERWRITTEN by the launch (as the reduce family above; a zeroed y works unchanged). Replaces reference kernels/dot-product/dot_product.cu:232-276."),
    ("sgemv", "y[M] = a[M,K] * x[K], fp32. Replaces reference kernels/sgemv/sgemv.cu:138-190 (K % 32, K % 128, K == 16)."),
    ("hgemv", "y[M] = a[M,K] * x[K], fp16 in/out, fp32 accumulate. Replaces reference kernels/hgemv/hgemv.cu:140-196."),
    ("mat_transpose", "y[col,row] = x[row,col]^T, fp32, bit-exact. Replaces reference kernels/mat-transpose/mat_transpose.cu:270-360."),
    ("embedding", "out[i,:] = weight[idx[i],:]; idx int32[n], weight [vocab, emb_size]; out-of-range rows are zero-filled.\n"
                  " * Replaces reference kernels/embedding/embedding.cu:99-133 (`void f(Tensor a, Tensor weight, Tensor o)`)."),
]


def main():
    out = []
    out.append("/* cln_amd.h -- C ABI of the MI355X (gfx950) kernel library. GENERATED by\n"
               " * cuda-learn-notes_amd/tools/gen_header.py from cuda-learn-notes_amd/manifest.py; do not edit.\n"
               " *\n"
               " * Conventions: plain pointers to DEVICE memory, sizes as int / long long, `stream` is a hipStream_t\n"
               " * passed as void* (NULL = default stream). Every function returns 0 on success or\n"
               " *   -1 bad argument, -2 unsupported shape, -3 HIP launch error, -4 vendor-library error.\n"
               " * Launches are asynchronous on `stream`. The callee allocates no device memory for the data path and keeps no pointer of the\n"
               " * caller's, with two stated exceptions: (1) the split-K workspace a caller REGISTERS per stream (cln_hgemm_set_workspace, end of\n"
               " * this file; library-owned buffers exist only after cln_hgemm_library_workspace(1)); (2) one 1-MiB slab per device of 16-KiB\n"
               " * ticket slots for the scalar-result kernels (block_all_reduce_sum_*, dot_prod_*), hipMalloc'ed on their first eager call and\n"
               " * freed by cln_release_workspaces(); captured launches never touch it.\n"
               " */\n#ifndef CLN_AMD_H\n#define CLN_AMD_H\n#ifdef __cplusplus\nextern \"C\" {\n#endif\n#include <stddef.h>\n")
    out.append("#define CLN_OK 0\n#define CLN_ERR_BAD_ARG (-1)\n#define CLN_ERR_UNSUPPORTED (-2)\n"
               "#define CLN_ERR_LAUNCH (-3)\n#define CLN_ERR_VENDOR (-4)\n")
    for lib, doc in GROUPS:
        out.append("\n/* ---- %s\n */" % doc)
        for e in manifest.entries_of(lib):
            out.append("/* impl: %s */" % e.impl)
            out.append(PROTO[e.sig].format(n=e.name))
    out.append("\n/* ---- introspection (not part of the reference surface): which gfx950 kernel a run-time dispatched name\n"
               " * launches for a shape, as text. HGEMM names: d0..d2 = M, N, K; flash-attn names: d0..d3 = B, H, N, D.\n"
               " * Returns the text length, -2 for an unsupported shape, -1 for a name bound to one fixed kernel (its\n"
               " * `impl` comment above). Host-only: no launch, no device access. The tuning / ablation hooks\n"
               " * (cln_hgemm_variant, cln_fa2_variant) live in the TEST-ONLY libcln_amd_probe.so and are not declared here. */")
    out.append("int cln_describe(const char* name, int d0, int d1, int d2, int d3, int stages, char* buf, int buflen);")
    out.append("/* 1 = `stages` selects the pipeline depth of the kernel this (name, shape) runs; 0 = the plan has one pipeline and the value is ignored\n"
               " * (cln_describe's text then says \"stages ignored\": the 192 / 160 / 128-wide one-wave-per-SIMD HGEMM tiles, split-K and tail-split plans);\n"
               " * < 0 = cln_describe's status. For callers that sweep `stages` as the reference scripts do (kernels/hgemm/hgemm.py:359-361). */")
    out.append("int cln_stages_honoured(const char* name, int d0, int d1, int d2, int d3, int stages);")
    out.append("\n/* ---- split-K workspace of the best-dispatch HGEMM names (not part of the reference surface: the reference's bindings take\n"
               " * only a, b, c -- kernels/hgemm/mma/basic/hgemm_mma_stage.cu:2380-2413 -- and never need one). Shapes with few output tiles and a\n"
               " * long K (and the last tile rows of a tile count just past whole rounds of 256) are split over K; the fp32 partials live in ONE\n"
               " * workspace per (device, stream): 4 KiB of arrival counters + S*M*N floats.\n"
               " *   default         NONE: the library never allocates. A stream that was given no region runs every shape single-pass.\n"
               " *   caller-owned    cln_hgemm_set_workspace(ptr, bytes, stream): size it with cln_hgemm_workspace_bytes(M, N, K) (0 = the shape runs\n"
               " *                   single-pass anyway); a shape that needs more than `bytes` runs single-pass. The first 4 KiB are zeroed on `stream` by\n"
               " *                   the call; keep the region alive and untouched while launches are queued; ptr = NULL withdraws it. (host.py does this\n"
               " *                   for every stream with one tensor from torch's caching allocator.)\n"
               " *   library-owned   opt-in, cln_hgemm_library_workspace(1), for callers without an allocator of their own: allocated with hipMalloc on a\n"
               " *                   stream's first split-K shape (never during stream capture: such a launch runs single-pass), 4 KiB + a power of two\n"
               " *                   from 16 to 256 MiB, at most 8 held per process (least recently used freed first, after its last launch has completed),\n"
               " *                   all freed by cln_release_workspaces().\n"
               " * Threads: calls are serialised per process while a workspace is in use (the lock covers the 1-2 launches of a split-K call), so two\n"
               " * host threads may call hgemm on one stream. Graphs: a captured launch holds the pointer of the workspace its capture stream had and\n"
               " * always takes the partial + reduce form (no arrival counters: nothing in the region outlives the call). Do not replay such a graph\n"
               " * CONCURRENTLY with other work that uses the same region (another replay, or eager split-K calls on the capture stream): the partials\n"
               " * of the two would mix -- give every graph that must overlap its own caller-owned region. A library-owned workspace that a capture has\n"
               " * used is pinned (never evicted or regrown; freed by cln_release_workspaces() only). A shape's result does not depend on whose\n"
               " * workspace is used, but it differs in the last bit from the single-pass plan (fp32 summation order), so runs of a split-K shape with\n"
               " * and WITHOUT a workspace are not bit-identical ($CLN_AMD_NO_SPLITK=1, the library's only environment variable, forces single-pass everywhere). */")
    out.append("size_t cln_hgemm_workspace_bytes(int M, int N, int K);")
    out.append("int cln_hgemm_set_workspace(void* ptr, size_t bytes, void* stream);")
    out.append("size_t cln_release_workspaces(void);   /* returns the bytes freed */")
    out.append("size_t cln_hgemm_workspace_held(void); /* library-owned bytes currently held */")
    out.append("int cln_hgemm_library_workspace(int enable); /* returns the previous setting (default 0) */")
    out.append("\n#ifdef __cplusplus\n}\n#endif\n#endif /* CLN_AMD_H */\n")
    path = os.path.join(ROOT, "include", "cln_amd.h")
    os.makedirs(os.path.dirname(path), exist_ok=True)
    with open(path, "w") as f:
        f.write("\n".join(out))
    print("wrote", path)


if __name__ == "__main__":
    main()
