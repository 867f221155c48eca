"""Phase timing of the split-bf16 forward from inside the kernel (needs a -DBF_TRACE build: FASTNERF_LIB=variants/trace.so).
Per wave and hidden layer: k-loop span, barrier wait, epilogue span, barrier wait (s_memtime ticks); plus how the two
workgroups that share a CU line up."""
import sys, os, ctypes as C, numpy as np, torch
sys.path.insert(0, '/root/repo')
import fastnerf as fn
from fastnerf import ops, _lib
save = (sys.argv[1] if len(sys.argv) > 1 else 'save') == 'save'
torch.manual_seed(0)
dev = torch.device('cuda')
args = fn.run_nerf.make_args(N_importance=128, N_samples=64, no_reload=True)
ktr = fn.run_nerf.create_nerf(args)[0]
net = ktr['network_fine']
N, S = 4096, 192
P = N * S
ro = torch.randn(N, 3, device=dev) * 0.1; rd = torch.randn(N, 3, device=dev)
rays11 = ops.pack_rays(ro, rd, 2.0, 6.0)
z = torch.sort(torch.rand(N, S, device=dev) * 4 + 2, -1).values
pf, pb = net.packed(refresh=True)
raw = torch.empty(N, S, 4, device=dev)
act = torch.empty(ops.act_floats(P), device=dev) if save else None
for _ in range(3):
    ops.mlp_fwd(rays11, z, net.flat, pf, act=act, raw=raw)
torch.cuda.synchronize()
NEV, NBLK = 40, 512
buf = np.zeros(NBLK * 4 * NEV, dtype=np.int64)
l = _lib.lib()
l.fastnerf_debug_trace.restype = C.c_int
l.fastnerf_debug_trace.argtypes = [C.c_void_p, C.c_int]
rc = l.fastnerf_debug_trace(buf.ctypes.data, buf.size)
assert rc == 0, rc
W = int(os.environ.get('TR_WAVES', '4'))   # waves per workgroup (8 with csrc/experimental/t128.patch applied)
NBLK = 2048 // W
t = buf.reshape(NBLK, W, NEV)
os.makedirs('gpurun_out', exist_ok=True)
np.save('gpurun_out/trace_%s%s.npy' % ('save' if save else 'inf', '_1wg' if os.environ.get('BF_ONE_WG') else ''), t)
ev = t[:, :, :36].astype(np.float64)
G = np.stack([ev[:, :, 4 * k + 1] - ev[:, :, 4 * k] for k in range(1, 8)], -1)
B1 = np.stack([ev[:, :, 4 * k + 2] - ev[:, :, 4 * k + 1] for k in range(1, 8)], -1)
E = np.stack([ev[:, :, 4 * k + 3] - ev[:, :, 4 * k + 2] for k in range(1, 8)], -1)
B2 = np.stack([ev[:, :, 4 * k + 4] - ev[:, :, 4 * k + 3] for k in range(1, 8)], -1)
np.set_printoptions(linewidth=200, precision=0, suppress=True)
print('mode', 'save' if save else 'inference', ' ticks; layers 1..7 (layer 5 has the skip input)')
for name, a in (('k-loop', G), ('barrier after k-loop', B1), ('epilogue', E), ('barrier after epilogue', B2)):
    print('%-24s mean %s   overall %.0f  (p10 %.0f  p90 %.0f)' % (name, a.mean((0, 1)), a.mean(), np.percentile(a, 10), np.percentile(a, 90)))
tile = ev[:, :, 34] - ev[:, :, 0]
print('whole tile: mean %.0f   layers 1..7: %.0f   PE+L0: %.0f   heads+feature+view+rgb: %.0f' % (
    tile.mean(), (ev[:, :, 32] - ev[:, :, 4]).mean(), (ev[:, :, 4] - ev[:, :, 0]).mean(), (ev[:, :, 34] - ev[:, :, 32]).mean()))
wall = (t[:, :, 37] - t[:, :, 36]).astype(np.float64)
m = wall > 0
print('shader clock during the traced tile: %.3f GHz (s_memtime ticks per 10 ns wall tick / 10)' % ((tile[m] / wall[m]).mean() / 10))
if W == 4:
    # pair the workgroups by CU
    hw = t[:, 0, 38]; xcc = t[:, 0, 39] & 15
    cu = (hw >> 8) & 15; se = (hw >> 13) & 7; sh = (hw >> 12) & 1
    key = xcc * 1000 + se * 100 + sh * 50 + cu
    pairs = {}
    for b in range(NBLK):
        pairs.setdefault(int(key[b]), []).append(b)
    sizes = np.bincount([len(v) for v in pairs.values()])
    print('workgroups per CU key histogram', sizes)
    # phase offset: start of layer-3 k-loop of WG a vs WG b on the same CU, modulo the layer period
    offs = []
    for v in pairs.values():
        if len(v) == 2:
            a, b = v
            per = (ev[a, 0, 16] - ev[a, 0, 12])
            d = (ev[b, 0, 12] - ev[a, 0, 12]) % per
            offs.append(d / per)
    offs = np.array(offs)
    print('pairs %d: phase offset of the second WG within the first one\'s layer period: histogram (10 bins)' % len(offs), np.histogram(offs, bins=10, range=(0, 1))[0])
    half = NBLK // 2
    print('whole tile, first-dispatched half of the grid (older workgroup of each CU): %.0f ticks; second half: %.0f' % (tile[:half].mean(), tile[half:].mean()))
    print('tile start of the traced (5th) tile relative to the CU partner: second-half workgroups start %.0f ticks later on average' % (ev[half:, 0, 0] - ev[:half, 0, 0]).mean())
    print('example WG 0 wave 0 stamps (relative):', (ev[0, 0, :36] - ev[0, 0, 0]).astype(np.int64))

