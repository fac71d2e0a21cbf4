"""On-GPU probe of primitive semantics the kernels rely on (run on the GPU box; compiles a tiny HIP file).
Prints, for ds_read_b64_tr_b16, which LDS element every (lane, j) receives when lane l supplies the
address of row 4*(l>>4) + ((l&15)>>2), cols 4*(l&3)..+3 of a [16][16] half image -- the model used by
read_nfrag (hgemm_mfma.cuh) and the V fragments (probe/flash_attn.cuh) -- and checks the MFMA C/D layouts."""
import ctypes
import os
import subprocess
import sys
import tempfile

import torch

SRC = r'''
#include <hip/hip_runtime.h>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));
typedef float f16v __attribute__((ext_vector_type(16)));
typedef __fp16 fp16x4_tr __attribute__((__vector_size__(4 * sizeof(__fp16))));
extern "C" __global__ void probe_tr(float* out) {
  __shared__ __attribute__((aligned(16))) _Float16 lds[256];
  int l = threadIdx.x;
  for (int i = l; i < 256; i += 64) lds[i] = (_Float16)i;
  __syncthreads();
  int row = 4 * (l >> 4) + ((l & 15) >> 2), col = 4 * (l & 3);
  fp16x4_tr t = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_tr*)(lds + row * 16 + col));
  for (int j = 0; j < 4; ++j) out[l * 4 + j] = (float)t[j];
}
// D = A*B with A[i][k] = i+1 (k==0 only), B[k][j] = j+1 (k==0 only) => D[i][j] = (i+1)*(j+1)
extern "C" __global__ void probe_mfma16(float* out) {
  int l = threadIdx.x;
  h8 a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
  if ((l >> 4) == 0) { a[0] = (_Float16)((l & 15) + 1); b[0] = (_Float16)((l & 15) + 1) * (_Float16)0.125; }
  f4 c = {0,0,0,0};
  c = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 4; ++r) out[l * 4 + r] = c[r];
}
extern "C" __global__ void probe_mfma32(float* out) {
  int l = threadIdx.x;
  h8 a = {0,0,0,0,0,0,0,0}, b = {0,0,0,0,0,0,0,0};
  if ((l >> 5) == 0) { a[0] = (_Float16)((l & 31) + 1); b[0] = (_Float16)((l & 31) + 1) * (_Float16)0.125; }
  f16v c = {0};
  c = __builtin_amdgcn_mfma_f32_32x32x16_f16(a, b, c, 0, 0, 0);
  for (int r = 0; r < 16; ++r) out[l * 16 + r] = c[r];
}
extern "C" void run(float* o_tr, float* o16, float* o32) {
  hipLaunchKernelGGL(probe_tr, dim3(1), dim3(64), 0, 0, o_tr);
  hipLaunchKernelGGL(probe_mfma16, dim3(1), dim3(64), 0, 0, o16);
  hipLaunchKernelGGL(probe_mfma32, dim3(1), dim3(64), 0, 0, o32);
  hipDeviceSynchronize();
}
'''


def main():
    d = tempfile.mkdtemp()
    src = os.path.join(d, "probe.hip")
    so = os.path.join(d, "probe.so")
    open(src, "w").write(SRC)
    subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O2", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    o_tr = torch.zeros(256, device="cuda")
    o16 = torch.zeros(256, device="cuda")
    o32 = torch.zeros(1024, device="cuda")
    lib.run(ctypes.c_void_p(o_tr.data_ptr()), ctypes.c_void_p(o16.data_ptr()), ctypes.c_void_p(o32.data_ptr()))
    tr = o_tr.cpu().view(64, 4)
    ok = True
    for l in range(64):
        exp = [(4 * (l >> 4) + j) * 16 + (l & 15) for j in range(4)]
        got = [int(x) for x in tr[l].tolist()]
        if got != exp:
            ok = False
            print("tr16 lane %2d got %s expected %s" % (l, got, exp))
    print("ds_read_b64_tr_b16 model:", "OK" if ok else "MISMATCH")
    c16 = o16.cpu().view(64, 4)
    ok16 = all(abs(c16[l, r].item() - ((4 * (l >> 4) + r) + 1) * ((l & 15) + 1) * 0.125) < 1e-3
               for l in range(64) for r in range(4))
    print("mfma 16x16x32 C layout (row=4*(l>>4)+r, col=l&15):", "OK" if ok16 else "MISMATCH")
    if not ok16:
        print(c16)
    c32 = o32.cpu().view(64, 16)
    ok32 = all(abs(c32[l, r].item() - (((r & 3) + 8 * (r >> 2) + 4 * (l >> 5)) + 1) * ((l & 31) + 1) * 0.125) < 1e-2
               for l in range(64) for r in range(16))
    print("mfma 32x32x16 C layout (row=(r&3)+8*(r>>2)+4*(l>>5), col=l&31):", "OK" if ok32 else "MISMATCH")
    if not ok32:
        print(c32)
    return 0 if (ok and ok16 and ok32) else 1


if __name__ == "__main__":
    sys.exit(main())
