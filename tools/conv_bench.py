import os, sys, time, torch, torch.nn.functional as F
mode = sys.argv[1]
torch.backends.cudnn.benchmark = (os.environ.get("BENCH","0")=="1")
def t(fn, it=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(it): fn()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/it
for (N,Cin,Cout,H,k) in [(128,256,256,32,3),(128,256,256,16,3),(128,256,256,8,3),(128,256,256,32,1),(128,128,128,32,3),(128,3,128,32,3),(128,256,3,32,3)]:
    x=torch.randn(N,Cin,H,H,device='cuda'); w=torch.randn(Cout,Cin,k,k,device='cuda')
    if mode=='nhwc':
        x=x.to(memory_format=torch.channels_last); w=w.to(memory_format=torch.channels_last)
    x.requires_grad_(True); w.requires_grad_(True)
    y=F.conv2d(x,w,padding=k//2); gy=torch.randn_like(y)
    f=t(lambda: F.conv2d(x,w,padding=k//2))
    def bw():
        y=F.conv2d(x,w,padding=k//2); y.backward(gy)
    fb=t(bw)
    gx=t(lambda: torch.ops.aten.convolution_backward(gy,x,w,None,[1,1],[k//2,k//2],[1,1],False,[0,0],1,[True,False,False]))
    gw=t(lambda: torch.ops.aten.convolution_backward(gy,x,w,None,[1,1],[k//2,k//2],[1,1],False,[0,0],1,[False,True,False]))
    fl=2*N*H*H*Cin*Cout*k*k/1e9
    print(mode,(N,Cin,Cout,H,k),'fwd %.3f ms (%.1f TF) dgrad %.3f wgrad %.3f fwd+bwd %.3f'%(f,fl/f,gx,gw,fb),flush=True)
