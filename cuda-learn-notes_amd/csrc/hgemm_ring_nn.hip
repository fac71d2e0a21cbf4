// NN-layout instantiations of the LDS-DMA ring HGEMM (B row-major [K,N]).
#define RING_LAYOUT hgemm::NN
#define RING_FN(name) name##_nn
#include "hgemm_ring_impl.inc"
