import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from wc_gan_amd import ops, _lib
N, H, C = 128, 32, 256
g = torch.Generator(device='cpu'); g.manual_seed(1)
x = torch.randn(N, H, H, C, generator=g).cuda(); A = (torch.randn(1, C, C, generator=g) / 16).cuda()
mu = torch.zeros(C).cuda(); b = torch.zeros(1, C).cuda(); y = torch.empty_like(x)
lib = _lib.load()
nb = lib.wc_apply_workspace_bytes(N, H * H, C, 1)
ws = torch.zeros(nb, dtype=torch.uint8, device='cuda')
for _ in range(5):
    rc = lib.wc_apply_f32(x.data_ptr(), mu.data_ptr(), A.data_ptr(), b.data_ptr(), None, N, H * H, C, 1, y.data_ptr(), None, ws.data_ptr(), nb, torch.cuda.current_stream().cuda_stream)
torch.cuda.synchronize()
d = ws[nb - 2048:nb - 1024].view(torch.int64).cpu().view(2, 8, 8)
names = ['start', 'vmcnt wait done', 'dma issued', 'mfma+convert done', 'stores issued', 'post-barrier', '-']
for wg in range(2):
    for w in (0, 4):
        t = d[wg, w].tolist()
        print(f'WG{wg} wave{w}:', ' '.join(f'{names[i]}=+{t[i]-t[0]}' for i in range(1, 5)), f'| loop: {t[6]} shader ticks / {t[7]} x10ns -> {t[6]/max(t[7],1)*0.1:.2f} GHz')
