"""Back-to-back launch stress of one attention shape: batches of 8 launches into 8 output buffers, every buffer compared with
the first launch's result.   python fa_race_stress.py B H N D batches [variant codes...]"""
import os
import sys

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry  # noqa: E402

pkg = entry.load_package()
from cuda_learn_notes_amd import host  # noqa: E402

dev = torch.device("cuda:0")
fa = pkg.flash_attn_lib()
B, H, N, D, batches = (int(x) for x in sys.argv[1:6])
codes = [int(x) for x in sys.argv[6:]]
torch.manual_seed(0)
q, k, v = (torch.randn(B, H, N, D, dtype=torch.half, device=dev) for _ in range(3))
fn = fa.flash_attn_mma_stages_split_q_shared_qkv if D <= 256 else fa.flash_attn_mma_stages_split_q_tiling_qkv
NB = int(os.environ.get('NBUF', '8'))
bufs = [torch.zeros_like(q) for _ in range(NB)]
forms = [("stages=2", lambda o: fn(q, k, v, o, 2)), ("stages=1", lambda o: fn(q, k, v, o, 1))]
forms += [("variant %d" % c, (lambda c: lambda o: host.fa2_variant((8, 0, 0, c), q, k, v, o))(c)) for c in codes]
only = os.environ.get("FORMS")
if only:
    forms = [f for f in forms if f[0] in only.split(",")]
first = None
for name, call in forms:
    ref = torch.zeros_like(q)
    call(ref)
    torch.cuda.synchronize()
    if first is None:
        first = ref
    bad, worst, where = 0, 0.0, None
    for _ in range(batches):
        for o in bufs:
            call(o)
        torch.cuda.synchronize()
        for o in bufs:
            if not torch.equal(o, ref):
                bad += 1
                d = (o.float() - ref.float()).abs()
                worst = max(worst, d.max().item())
                if where is None:
                    idx = (d > 0).nonzero()
                    where = (int(idx.shape[0]), idx[0].tolist(), idx[-1].tolist())
    print("RACE %-12s %s: %d of %d launches differ from the form's first launch (worst %.3e, first mismatch: %s); first launch == stages=2 first launch: %s"
          % (name, (B, H, N, D), bad, batches * NB, worst, where, bool(torch.equal(ref, first))), flush=True)
