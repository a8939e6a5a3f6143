import torch, time
dev='cuda:0'
n=512*1024*1024  # 2 GB fp32
a=torch.empty(n,device=dev); b=torch.empty(n,device=dev)
def t(f,reps=10):
    f(); torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/reps*1e-3
tw=t(lambda: a.fill_(1.0)); print("fill (write-only) 2GB: %.2f TB/s"%(n*4/tw/1e12))
tc=t(lambda: b.copy_(a)); print("copy 2GB->2GB: %.2f TB/s total (%.2f each way)"%(2*n*4/tc/1e12, n*4/tc/1e12))
tr=t(lambda: a.sum()); print("sum (read-only) 2GB: %.2f TB/s"%(n*4/tr/1e12))
m=70*1024*1024  # 280 MB (cost-volume size)
c=torch.empty(m,device=dev)
tw2=t(lambda: c.fill_(1.0),50); print("fill 280MB: %.2f TB/s  (%.1f us)"%(m*4/tw2/1e12, tw2*1e6))
