cd ${GRAFT_REPO_ROOT:-/root/repo}
K=cuda-learn-notes_amd/kernels
set -x
timeout 300 python $K/hgemm/hgemm.py --mma --MNK 4096 --iters 10 2>&1 | tail -12
timeout 300 python $K/hgemm/hgemm.py --mma-all --wmma-all --cuda-all --enable-mma-tn --MNK 1024 --iters 3 2>&1 | tail -60 | cut -c1-170
timeout 300 python $K/flash-attn/flash_attn_mma.py --B 4 --H 8 --N 2048 --D 64 --check --iters 5 2>&1 | tail -40 | cut -c1-170
timeout 300 python $K/flash-attn/flash_attn_mma.py --B 1 --H 32 --N 4096 --D 512 --check --iters 2 2>&1 | tail -14 | cut -c1-170
for s in elementwise/elementwise reduce/block_all_reduce softmax/softmax layer-norm/layer_norm rms-norm/rms_norm rope/rope embedding/embedding histogram/histogram gelu/gelu dot-product/dot_product sgemv/sgemv hgemv/hgemv mat-transpose/mat_transpose; do
  echo "=== $s"; timeout 300 python $K/$s.py 2>&1 | tail -4 | cut -c1-170
done
