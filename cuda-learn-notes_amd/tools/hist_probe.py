"""histogram_i32 / _i32x4 achieved input bandwidth (events over 50 launches) and bit-exactness vs torch.bincount."""
import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
import __graft_entry__ as entry
pkg = entry.load_package()
from cuda_learn_notes_amd import bench_utils as bu, _loader
so = _loader.load_so("libcln_amd.so")
dev = torch.device("cuda:0")
lib = pkg.load("histogram")
torch.manual_seed(0)
for n, bins, tag in ((4096 * 4096, 1024, "uniform 1024 bins"), (4096 * 4096, 256, "uniform 256 bins"),
                     (4096 * 4096, 8192, "uniform 8192 bins"), (4096 * 4096, 3, "3 bins (contended)"),
                     (1 << 20, 1024, "1 Mi elements")):
    a = torch.randint(0, bins, (n,), dtype=torch.int32, device=dev)
    ref = torch.bincount(a.long(), minlength=bins).to(torch.int32)
    for name in ("histogram_i32", "histogram_i32x4"):
        fn = getattr(lib, name)
        y = fn(a)
        ok = torch.equal(y.cpu(), ref.cpu())
        ms, mn, _ = bu.time_call_events(lambda: fn(a), 5, 50)
        out = torch.zeros(bins, dtype=torch.int32, device=dev)
        raw = getattr(so, name)
        st = torch.cuda.current_stream().cuda_stream
        kms, kmn, _ = bu.time_call_events(lambda: raw(a.data_ptr(), out.data_ptr(), n, bins, st), 5, 50)
        print("%-20s %-18s exact=%s  binding %8.2f us | kernel %7.2f us  %7.1f GB/s" % (tag, name, ok, ms * 1e3, kms * 1e3, n * 4 / kms * 1e-6), flush=True)
