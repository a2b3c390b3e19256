import ctypes as C, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
from strawberry_amd import _lib
_lib.LIB_PATH = os.path.join(ROOT, "strawberry_amd", "lib", "libsbgpu_stamps.so")
from strawberry_amd import em
ctx = em.default_context(0)
buf = torch.zeros(16, dtype=torch.float64, device="cuda")
L = _lib.load()
L.sbgpu_debug_touch_streams.argtypes = [C.c_void_p, C.c_void_p]
for _ in range(2):
    assert L.sbgpu_debug_touch_streams(ctx.h, buf.data_ptr()) == 0
torch.cuda.synchronize()
