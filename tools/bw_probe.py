import torch
dev=torch.device("cuda:0")
for mb in (67, 537):
    n=mb*1024*1024//4
    x=torch.empty(n,device=dev); y=torch.empty(n,device=dev)
    def t(f,reps=20):
        for _ in range(3): f()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(reps): f()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)/reps
    tf=t(lambda: y.fill_(1.0)); tc=t(lambda: y.copy_(x)); ts=t(lambda: x.sum())
    print("%d MB: fill %.1f us (%.2f TB/s)  copy %.1f us (%.2f TB/s r+w)  sum %.1f us (%.2f TB/s)"%(mb,tf*1e3,mb*1.048576e-6/tf*1e3/1e3*1e3 if False else mb*1048576/tf/1e9, tc*1e3, 2*mb*1048576/tc/1e9, ts*1e3, mb*1048576/ts/1e9))
