import os, sys
sys.path.insert(0, os.getcwd())
from zignal_amd import _lib
from zignal_amd.device import DeviceBuffer, Plan, Timer, sync
from zignal_amd.measure.mlstaps import TAPS
N, C, R = 20, 512, 17
p = Plan(N, TAPS[N][0], device=0)
L = p.L
p.reserve(C, staging=True)
x = DeviceBuffer(R * L * C * 8); x.fill(0)
h = DeviceBuffer(L * C * 8)
for name, args in (("stacked in, frame out", (1, R * L, C, 1)), ("stacked in and out", (1, R * L, 1, L))):
    rs, cs, ors, ocs = args
    f = lambda: p.impulse_device(x.ptr, _lib.ZMLS_F64, R * L, C, rs, cs, h.ptr, ors, ocs)
    f(); sync(0)
    t = Timer(0); t.start()
    for _ in range(5): f()
    t.stop()
    print(name, round(t.elapsed_ms() / 5, 3), "ms")
