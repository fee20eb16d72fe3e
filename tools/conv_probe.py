import time, torch, torch.nn.functional as F, sys
torch.manual_seed(0)
B=int(sys.argv[1]); mode=sys.argv[2]
dev='cuda'
conv1=torch.nn.Conv2d(14,16,3).to(dev); conv2=torch.nn.Conv2d(16,32,3).to(dev)
x=(torch.rand(B,14,15,15,device=dev)<0.02).float()
def fwd_conv(x):
    with torch.autocast('cuda',dtype=torch.bfloat16):
        return torch.tanh(conv2(torch.tanh(conv1(x.to(torch.bfloat16)))))
def gemm_conv(x, conv):
    B_,C,H,W=x.shape; O=conv.weight.shape[0]
    cols=F.unfold(x,3)                      # [B, C*9, L]
    y=torch.matmul(conv.weight.reshape(O,-1).to(cols.dtype), cols) + conv.bias.to(cols.dtype)[None,:,None]
    return y.reshape(B_,O,H-2,W-2)
def fwd_gemm(x):
    x=x.to(torch.bfloat16)
    return torch.tanh(gemm_conv(torch.tanh(gemm_conv(x,conv1)),conv2))
if mode=='bench': torch.backends.cudnn.benchmark=True
f = fwd_gemm if mode=='gemm' else fwd_conv
for it in range(3):
    torch.cuda.synchronize(); t0=time.perf_counter()
    y=f(x); loss=y.float().pow(2).mean(); loss.backward()
    torch.cuda.synchronize(); print(mode,B,'iter',it,'%.3f s'%(time.perf_counter()-t0),flush=True)
