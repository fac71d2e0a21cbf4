// TN-layout instantiations of the LDS-DMA ring HGEMM (B stored [N,K], reference as_col_major,
// kernels/hgemm/tools/utils.py:135-140).
#define RING_LAYOUT hgemm::TN
#define RING_FN(name) name##_tn
#include "hgemm_ring_impl.inc"
