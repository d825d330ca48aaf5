"""Read the s_memtime stamps of a RAYEN_DBG_STAMP variant build (one wave of the headline kernel)."""
import ctypes, os, sys
sys.path.insert(0, os.getcwd())
import torch
from rayen_amd import ops, workloads, _lib
from rayen_amd.constraint_module import ConstraintModule
cs = workloads.build_constraints(workloads.make_raw("c3"))
layer = ConstraintModule(cs, create_map=False).cuda()
x = torch.empty(262144, 64, device="cuda").uniform_(-1, 1)
dp, _ = layer.device_pack(x.device)
for _ in range(30): ops.project_raw(x, dp, want_active=False)
torch.cuda.synchronize()
lib = _lib.load()
buf = (ctypes.c_ulonglong * 4096)()
lib.rayen_debug_read.argtypes = [ctypes.c_void_p, ctypes.c_int]
print("rc", lib.rayen_debug_read(buf, 4096))
n = int(buf[0]); st = [int(buf[i]) for i in range(1, n)]
print("stamps", n - 1)
d = [st[i + 1] - st[i] for i in range(len(st) - 1)]
print(d)
