import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops, _lib
N, H, C = 128, 32, 256
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(N, H, H, C, generator=g).cuda(); A = (torch.randn(1, C, C, generator=g) / 16).cuda()
mu = torch.zeros(C).cuda(); b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
_lib.LIB_PATH = os.environ.get('WC_LIB', _lib.LIB_PATH)
lib = _lib.load()
nb = lib.wc_apply_workspace_bytes(N, H * H, C, 1)
ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
for _ in range(5):
    rc = lib.wc_apply_f32(x.data_ptr(), mu.data_ptr(), A.data_ptr(), b.data_ptr(), None, N, H * H, C, 1, y.data_ptr(), None, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
ws[nb - 8192:].zero_()
rc = lib.wc_apply_f32(x.data_ptr(), mu.data_ptr(), A.data_ptr(), b.data_ptr(), None, N, H * H, C, 1, y.data_ptr(), None, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
g_ = ws[nb - 7168:nb - 7104].view(torch.int64).cpu().tolist(); big = 1 << 62
t0_ = big - g_[0]
print(f'launch timeline (us): first WG start 0, last WG start {(g_[1]-t0_)/100:.2f}, longest prologue {g_[2]/100:.2f}, tile loop {((big-g_[7]))/100:.2f}..{g_[6]/100:.2f}, longest epilogue {g_[3]/100:.2f}, first WG end {(big-g_[4]-t0_)/100:.2f}, last WG end {(g_[5]-t0_)/100:.2f}')
d = ws[nb - 8192:nb - 7168].view(torch.int64).cpu().view(2, 8, 8)
names = ['start', 'vmcnt wait done', 'dma issued', 'mfma+convert done', 'stores issued', 'post-barrier', '-']
for wg in range(2):
    for w in range(8):
        t = d[wg, w].tolist()
        print(f'WG{wg} wave{w}:', ' '.join(f'{names[i]}=+{t[i]-t[0]}' for i in range(1, 5)), f'| loop: {t[6]} shader ticks / {t[7]} x10ns -> {t[6]/max(t[7],1)*0.1:.2f} GHz')

w = ws[nb - 6144:nb - 5120].view(torch.int32).cpu().tolist()
import collections
by = collections.defaultdict(list)
for b_, v in enumerate(w):
    by[(v >> 16) & 15].append((v & 0xFFFF) / 100)
for xcc in sorted(by):
    vals = by[xcc]; print(f'XCC {xcc}: {len(vals)} WGs, loop us min {min(vals):.1f} mean {sum(vals)/len(vals):.1f} max {max(vals):.1f}')
slow = sorted(((v & 0xFFFF) / 100, b_, (v >> 16) & 15, (v >> 20) & 0xFF) for b_, v in enumerate(w))
print('fastest', slow[:6]); print('slowest', slow[-10:])

ws2 = ws[nb - 5120:nb - 3072].view(torch.int32).cpu().tolist()
rows = []
for b_, v in enumerate(w):
    a0, a4 = ws2[2 * b_], ws2[2 * b_ + 1]
    rows.append(((v & 0xFFFF) / 100, b_, (a0 & 0xFFFF) / 100, ((a0 >> 16) & 0xFFFF) / 100, (a4 & 0xFFFF) / 100, ((a4 >> 16) & 0xFFFF) / 100))
rows.sort()
print('loop_us  WG   wave0: wait_sum store_sum | wave4: wait_sum store_sum')
for r in rows[:5] + rows[126:130] + rows[-8:]: print('%6.1f %4d   %6.1f %6.1f | %6.1f %6.1f' % r)
