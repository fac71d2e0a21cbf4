"""GPU probe: attention at short sequences (N = 256 / 512 / 768, many heads): the v2 kernel with 4 / 8 waves against the two-group kernels and the
planner's pick.  python fa_short_n_probe.py"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, __graft_entry__ as e
pkg=e.load_package(); from cuda_learn_notes_amd import bench_utils as bu, host
fa=pkg.flash_attn_lib(); dev=torch.device("cuda:0")
fn=fa.flash_attn_mma_stages_split_q_shared_qkv
V2_OPT = {32: 16397, 64: 16397, 96: 16399, 128: 16399, 256: 15}
def tf(call, fl, n=100):
    bu.prewarm(call,0.05); return fl/bu.time_region_events(call,n)*1e-9
for D in (64,128,256):
  for N in (256,512,768):
    for BH in (64,256,1024,4096):
        shape=(1,BH,N,D)
        q,k,v=(torch.randn(*shape,dtype=torch.half,device=dev) for _ in range(3)); o=torch.zeros_like(q)
        fl=bu.mha_flops_conventional(*shape); row={}
        for nw in (4,8):
            try: row["v2x%d"%nw]=tf(lambda: host.fa2_variant((nw,0,V2_OPT[D],0),q,k,v,o),fl)
            except RuntimeError: row["v2x%d"%nw]=float("nan")
        try: row["m16x"]=tf(lambda: host.fa2_variant((8,0,0,853 if D<=128 else 544),q,k,v,o),fl)
        except RuntimeError: row["m16x"]=float("nan")
        row["plan"]=tf(lambda: fn(q,k,v,o,2),fl)
        d=pkg.manifest.describe(fn.__name__,shape,2)
        print("SHORTN D=%3d N=%4d BH=%5d w256=%5d v2x4 %7.1f v2x8 %7.1f m16x %7.1f | plan %7.1f (%s)"%(D,N,BH,BH*N//256 if N%256==0 else 0,row["v2x4"],row["v2x8"],row["m16x"],row["plan"],d.split("<")[0]+" "+d.split("> ")[1][:7]),flush=True)
