// cln_describe: which gfx950 kernel a run-time dispatched C-ABI name runs for a given shape, as text.
//
//   int cln_describe(const char* name, int d0, int d1, int d2, int d3, int stages, char* buf, int buflen)
//     HGEMM names (G6 signature) and the two sgemm matrix-core names (S6):  d0 = M, d1 = N, d2 = K, d3 unused
//     flash-attn names:            d0 = B, d1 = H, d2 = N, d3 = D
//   returns the length of the text written to buf (NUL-terminated), or
//     CLN_ERR_UNSUPPORTED (-2)  the name exists but the shape is outside its supported set (the launch would fail too)
//     CLN_ERR_BAD_ARG (-1)      not a run-time dispatched name: one fixed kernel, named in manifest.py `impl`
// No launch, no device access: the planners of hgemm.hip / flash_attn.hip are evaluated on the host, so the
// name -> kernel table in cuda-learn-notes_amd/manifest.py is checked against the dispatch code on a CPU-only box
// (tests/test_describe.py). Not part of the reference surface.
#include "common.h"
#include <string.h>

int cln_hgemm_describe(const char* name, int M, int N, int K, int stages, char* buf, int len);
int cln_fa_describe(const char* name, int B, int H, int N, int D, int stages, char* buf, int len);
int cln_sgemm_describe(const char* name, int M, int N, int K, int stages, char* buf, int len);

CLN_API int cln_describe(const char* name, int d0, int d1, int d2, int d3, int stages, char* buf, int buflen) {
  if (!name || !buf || buflen <= 0) return CLN_ERR_BAD_ARG;
  int rc = cln_fa_describe(name, d0, d1, d2, d3, stages, buf, buflen);
  if (rc != CLN_ERR_BAD_ARG) return rc;
  rc = cln_sgemm_describe(name, d0, d1, d2, stages, buf, buflen);
  if (rc != CLN_ERR_BAD_ARG) return rc;
  return cln_hgemm_describe(name, d0, d1, d2, stages, buf, buflen);
}

// Does `stages` select the pipeline depth of the kernel this (name, shape) runs? 1 = yes; 0 = the plan has ONE pipeline and the value is ignored
// (the 192 / 160 / 128-wide one-wave-per-SIMD tiles, split-K and tail-split plans: cln_describe's text carries "[... stages ignored ...]" for exactly
// these); < 0 = cln_describe's status for the shape / name. A caller that sweeps `stages` (the reference scripts print one row per stage count,
// kernels/hgemm/hgemm.py:359-361) can tell a repeated row from a measured one without parsing text.
CLN_API int cln_stages_honoured(const char* name, int d0, int d1, int d2, int d3, int stages) {
  char buf[512];
  const int rc = cln_describe(name, d0, d1, d2, d3, stages, buf, (int)sizeof(buf));
  if (rc < 0) return rc;
  return strstr(buf, "stages ignored") == nullptr ? 1 : 0;
}
