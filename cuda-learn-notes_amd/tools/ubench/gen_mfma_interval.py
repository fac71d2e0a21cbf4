#!/usr/bin/env python3
"""Generates mfma_interval.hip: the issue interval of back-to-back v_mfma_f32_16x16x32_f16 on gfx950 as a function of WHICH registers
the operands come from. tools/ubench/mfma_port.hip found 18.06 clocks per MFMA with one A / B register set and 22.09 with a fresh
set per MFMA (the matrix pipe needs 16): this sweep pins every operand to a chosen register (one asm block per loop body, all
registers clobbered) to find the rule. One wave per SIMD and two; shader clocks of wave 0 (s_memtime) and event time.
    python gen_mfma_interval.py > mfma_interval.hip && hipcc --offload-arch=gfx950 -O3 mfma_interval.hip -o mfma_interval"""

def v(i, n=4): return "v[%d:%d]" % (i, i + n - 1)
def a(i, n=4): return "a[%d:%d]" % (i, i + n - 1)

VARIANTS = []
def variant(name, body): VARIANTS.append((name, body))

M = "v_mfma_f32_16x16x32_f16"
def seq(fn, n=16, op=M): return [("%s %s" % (op, fn(i))) for i in range(n)]

# accumulators: v[32+4i] (16 of them: v32..v95); A sets v[96+4j] j<8 (v96..v127); B sets v[128+4j] j<8 (v128..v159)
C = lambda i: v(32 + 4 * i)
A = lambda j: v(96 + 4 * (j % 8))
B = lambda j: v(128 + 4 * (j % 6))
CA = lambda i: a(4 * i)
variant("same A, same B, C=D in VGPRs", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(0), B(0), C(i))))
variant("fresh A each, same B", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i), B(0), C(i))))
variant("same A, fresh B each", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(0), B(i), C(i))))
variant("fresh A and B each", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i), B(i + 3), C(i))))
variant("4x4 tile order: A_j held over 4 MFMAs, B_l cycles", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i // 4), B(i % 4), C(i))))
variant("4x4 tile, snake: B order reverses every row", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i // 4), B((i % 4) if (i // 4) % 2 == 0 else 3 - (i % 4)), C(i))))
variant("2x2 sub-blocks: (A0,B0)(A0,B1)(A1,B1)(A1,B0) ...", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(2 * (i // 8) + ((i % 4) // 2)), B(2 * ((i // 4) % 2) + [0, 1, 1, 0][i % 4]), C(i))))
variant("same A, same B, C=D in AGPRs", seq(lambda i: "%s, %s, %s, %s" % (CA(i), A(0), B(0), CA(i))))
variant("fresh A and B, C=D in AGPRs", seq(lambda i: "%s, %s, %s, %s" % (CA(i), A(i), B(i + 3), CA(i))))
variant("4x4 tile order, C=D in AGPRs", seq(lambda i: "%s, %s, %s, %s" % (CA(i), A(i // 4), B(i % 4), CA(i))))
variant("fresh A and B, C = 0 (chain start), D in VGPRs", seq(lambda i: "%s, %s, %s, 0" % (C(i), A(i), B(i + 3))))
variant("same A and B, C = 0", seq(lambda i: "%s, %s, %s, 0" % (C(i), A(0), B(0))))
variant("fresh A and B, C in VGPRs, D in other VGPRs", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i), B(i + 3), C((i + 8) % 16))))
variant("A from AGPRs (fresh), B VGPR fresh, C=D VGPR", seq(lambda i: "%s, %s, %s, %s" % (C(i), a(64 + 4 * (i % 4)), B(i + 3), C(i))))
variant("A and B from AGPRs (fresh), C=D VGPR", seq(lambda i: "%s, %s, %s, %s" % (C(i), a(64 + 4 * (i % 4)), a(80 + 4 * ((i + 3) % 4)), C(i))))
variant("A and B from AGPRs (fresh), C=D AGPR", seq(lambda i: "%s, %s, %s, %s" % (CA(i), a(64 + 4 * (i % 4)), a(80 + 4 * ((i + 3) % 4)), CA(i))))
variant("fresh A/B, B sets offset by 2 registers (v[130+4j])", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i), v(130 + 4 * ((i + 3) % 5)), C(i))))
variant("dependent chain: 2 accumulators alternate, same A/B", seq(lambda i: "%s, %s, %s, %s" % (C(i % 2), A(0), B(0), C(i % 2))))
variant("dependent chain: 4 accumulators, fresh A/B", seq(lambda i: "%s, %s, %s, %s" % (C(i % 4), A(i), B(i + 3), C(i % 4))))
variant("bf16 16x16x32: fresh A and B", seq(lambda i: "%s, %s, %s, %s" % (C(i), A(i), B(i + 3), C(i)), op="v_mfma_f32_16x16x32_bf16"))
variant("f16 16x16x16 (K = 16: 2-register operands), fresh", seq(lambda i: "%s, %s, %s, %s" % (C(i), v(96 + 2 * (i % 16), 2), v(128 + 2 * ((i + 3) % 12), 2), C(i)), op="v_mfma_f32_16x16x16_f16"))
# 32x32x16: 16-register accumulators v[32+16i] i<4 (v32..v95)
C32 = lambda i: v(32 + 16 * (i % 4), 16)
variant("32x32x16: same A/B", seq(lambda i: "%s, %s, %s, %s" % (C32(i), A(0), B(0), C32(i)), n=8, op="v_mfma_f32_32x32x16_f16"))
variant("32x32x16: fresh A and B", seq(lambda i: "%s, %s, %s, %s" % (C32(i), A(i), B(i + 3), C32(i)), n=8, op="v_mfma_f32_32x32x16_f16"))
variant("fp8 16x16x32 (2-register operands), fresh", seq(lambda i: "%s, %s, %s, %s" % (C(i), v(96 + 2 * (i % 16), 2), v(128 + 2 * ((i + 3) % 12), 2), C(i)), op="v_mfma_f32_16x16x32_fp8_fp8"))
# an independent VALU op between MFMAs touching none of their registers
variant("fresh A/B + 1 v_add_f32 (v0) behind each MFMA", [x for i in range(16) for x in ("%s %s, %s, %s, %s" % (M, C(i), A(i), B(i + 3), C(i)), "v_add_f32 v%d, v%d, v8" % (i % 8, i % 8))])
variant("same A/B + 1 v_add_f32 behind each MFMA", [x for i in range(16) for x in ("%s %s, %s, %s, %s" % (M, C(i), A(0), B(0), C(i)), "v_add_f32 v%d, v%d, v8" % (i % 8, i % 8))])
variant("fresh A/B + s_nop 1 behind each MFMA", [x for i in range(16) for x in ("%s %s, %s, %s, %s" % (M, C(i), A(i), B(i + 3), C(i)), "s_nop 1")])

print("""// GENERATED by gen_mfma_interval.py -- see its docstring.
#include <hip/hip_runtime.h>
#include <stdio.h>
#define CLOB %s
""" % ", ".join(['"v%d"' % i for i in range(152)] + ['"a%d"' % i for i in range(96)]))
for n, (name, body) in enumerate(VARIANTS):
    nm = sum(1 for l in body if l.startswith("v_mfma"))
    print("__global__ __launch_bounds__(256) void k%d(float* out, unsigned long long* clk, int n) {" % n)
    print('  asm volatile("' + "\\n".join("v_mov_b32 v%d, 0" % i for i in range(152)) + '" ::: CLOB);')
    print('  asm volatile("' + "\\n".join("v_accvgpr_write_b32 a%d, 0" % i for i in range(96)) + '" ::: CLOB);')
    print("  const unsigned long long t0 = __builtin_amdgcn_s_memtime();")
    print("  for (int it = 0; it < n; ++it) {")
    print('    asm volatile("' + "\\n".join(body) + '" ::: CLOB);')
    print("  }")
    print("  const unsigned long long t1 = __builtin_amdgcn_s_memtime();")
    print("  float r; asm volatile(\"v_mov_b32 %0, v32\" : \"=v\"(r));")
    print("  out[blockIdx.x * blockDim.x + threadIdx.x] = r;")
    print("  if (blockIdx.x == 0 && threadIdx.x == 0) clk[0] = t1 - t0;")
    print("}")
print("""
typedef void (*kern)(float*, unsigned long long*, int);
struct V { const char* name; kern k; int nm; };
int main() {
  float* out; unsigned long long* clk;
  (void)hipMalloc(&out, 256 * 512 * 4); (void)hipMalloc(&clk, 64);
  V vs[] = {""")
for n, (name, body) in enumerate(VARIANTS):
    nm = sum(1 for l in body if l.startswith("v_mfma"))
    print('    {"%s", k%d, %d},' % (name, n, nm))
print("""  };
  const int n = 4000;
  for (int pass = 0; pass < 2; ++pass)
    for (auto& x : vs)
      for (int wps = 1; wps <= 2; ++wps) {
        hipEvent_t e0, e1; (void)hipEventCreate(&e0), (void)hipEventCreate(&e1);
        hipLaunchKernelGGL(x.k, dim3(256 * wps), dim3(256), 0, 0, out, clk, 10);
        (void)hipDeviceSynchronize();
        (void)hipEventRecord(e0);
        hipLaunchKernelGGL(x.k, dim3(256 * wps), dim3(256), 0, 0, out, clk, n);
        (void)hipEventRecord(e1); (void)hipEventSynchronize(e1);
        float ms = 0; (void)hipEventElapsedTime(&ms, e0, e1);
        unsigned long long c = 0; (void)hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
        printf("MI %-58s %d wave/SIMD: %7.2f clk per MFMA (wave 0), %6.2f ns per MFMA per SIMD\\n", x.name, wps, (double)c / ((double)n * x.nm), ms * 1e6 / ((double)n * x.nm * wps));
        (void)hipEventDestroy(e0), (void)hipEventDestroy(e1);
      }
  return 0;
}""")
