// Host-side entries of the compile units flash_attn_m16x.hip / probe/flash_attn_m16x_probe.hip (built with their own flags, see _build.py).
#pragma once
#include "common.h"
namespace fa2 {
int m16x_run(int D, int rows_per_wave, bool vt, bool one_stage, bool f32_scale, const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t s);  // product forms (D = 64 / 128 at 32 rows per wave, D = 64 at 64)
int m16x_probe_run(int D, int code, const void* q, const void* k, const void* v, void* o, int B, int H, int N, hipStream_t s);  // probe library only
}
