"""GPU probe: host cost of one launch -- the raw C-ABI call, the Python wrappers of host.py around it, torch's own dispatcher for the same
op, and the pieces (stream lookup, device checks).  Back-to-back calls on tiny tensors: the rate is the host's.  python host_overhead_probe.py"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch, __graft_entry__ as e
pkg=e.load_package(); from cuda_learn_notes_amd import bench_utils as bu, host, _loader
dev=torch.device("cuda:0")
sm=pkg.load("softmax"); ew=pkg.load("elementwise")
x=torch.randn(64,256,device=dev); o=torch.zeros_like(x)
a=torch.randn(1024,device=dev); b=torch.randn(1024,device=dev); c=torch.zeros_like(a)
def host_rate(fn,n=20000):
    for _ in range(200): fn()
    torch.cuda.synchronize(); t0=time.perf_counter()
    for _ in range(n): fn()
    t1=time.perf_counter(); torch.cuda.synchronize()
    return (t1-t0)/n*1e6
raw=_loader.symbol("elementwise_add_f32"); ap,bp,cp=a.data_ptr(),b.data_ptr(),c.data_ptr(); st=host._stream()
print("HOSTOV raw C-ABI call (tiny add)         %.2f us/call"%host_rate(lambda: raw(ap,bp,cp,1024,st)))
print("HOSTOV host wrapper elementwise_add_f32  %.2f us/call"%host_rate(lambda: ew.elementwise_add_f32(a,b,c)))
print("HOSTOV host wrapper softmax per-token    %.2f us/call"%host_rate(lambda: sm.safe_softmax_f32x4_per_token(x,o)))
print("HOSTOV torch.add(out=)                   %.2f us/call"%host_rate(lambda: torch.add(a,b,out=c)))
print("HOSTOV torch.softmax(out=)               %.2f us/call"%host_rate(lambda: torch.softmax(x,dim=1,out=o)))
print("HOSTOV _stream()                         %.2f us/call"%host_rate(host._stream))
print("HOSTOV torch.cuda.current_stream().cuda_stream %.2f us/call"%host_rate(lambda: torch.cuda.current_stream().cuda_stream))
print("HOSTOV _check_dev(3 tensors)             %.2f us/call"%host_rate(lambda: host._check_dev(a,b,c)))
# round 5: the CPython entry (csrc/pyext/cln_fastcall.c) against the pure-Python wrapper it falls back to (ctypes), and the scalar-result kernels
# without the zero-fill dispatch (csrc/stream_scratch.h)
print("HOSTOV fastcall module loaded            %s" % (host._fastcall is not None))
slow_add = getattr(ew.elementwise_add_f32, "__wrapped__", None)
if slow_add is not None:
    print("HOSTOV pure-Python wrapper (ctypes) add  %.2f us/call" % host_rate(lambda: slow_add(a, b, c)))
rd = pkg.load("reduce"); xr = torch.randn(4096, device=dev)
print("HOSTOV block_all_reduce_sum_f32x4_f32    %.2f us/call" % host_rate(lambda: rd.block_all_reduce_sum_f32x4_f32(xr), 10000))
print("HOSTOV torch.sum                         %.2f us/call" % host_rate(lambda: torch.sum(xr), 10000))
nm = pkg.load("rms_norm"); xh = torch.randn(64, 1024, device=dev).half(); oh = torch.zeros_like(xh)
print("HOSTOV rms_norm_f16x8_pack_f32           %.2f us/call" % host_rate(lambda: nm.rms_norm_f16x8_pack_f32(xh, oh, 1.0)))
hgl = pkg.hgemm_lib(); A = torch.randn(256, 256, device=dev).half(); C = torch.zeros_like(A)
print("HOSTOV hgemm G6 wrapper 256^3            %.2f us/call" % host_rate(lambda: hgl.hgemm_mma_m16n8k16_mma2x4_warp4x4x2_stages_dsmem(A, A, C, 2, False, 0), 10000))
print("HOSTOV torch.matmul(out=) 256^3          %.2f us/call" % host_rate(lambda: torch.matmul(A, A, out=C), 10000))
