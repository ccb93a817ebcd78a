import sys, ctypes as C, numpy as np
sys.path.insert(0, 'tests')
import torch
from bbqlib import bbq_amd as B, capi
L = capi.lib()
tr = getattr(L, "_ZN3bbq22launch_build_transposeEPKflilPfP12ihipStream_t")
tr.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
ce = getattr(L, "_ZN3bbq21launch_build_centroidEPKflilPfP12ihipStream_t")
ce.argtypes = [C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_void_p, C.c_void_p]
n, dim = 200, 8
npad = (n + 63) // 64 * 64
x = torch.arange(n * dim, dtype=torch.float32, device="cuda").reshape(n, dim) * 0.5
vT4 = torch.zeros((dim // 4, npad, 4), dtype=torch.float32, device="cuda")
print("rc", tr(x.data_ptr(), n, dim, npad, vT4.data_ptr(), None)); torch.cuda.synchronize()
v = vT4.cpu().numpy()
want = np.zeros((dim // 4, npad, 4), np.float32)
xx = x.cpu().numpy()
for i4 in range(dim // 4):
    want[i4, :n] = xx[:, 4 * i4:4 * i4 + 4]
print("transpose ok", (v == want).all()); print(v[0, :3], want[0, :3])
cen = torch.zeros(dim, dtype=torch.float32, device="cuda")
print("rc", ce(vT4.data_ptr(), n, dim, npad, cen.data_ptr(), None)); torch.cuda.synchronize()
print(cen.cpu().numpy()); print(xx.mean(axis=0))
