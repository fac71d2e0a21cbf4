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
