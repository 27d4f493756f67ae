"""HBM write / copy ceiling with plain streaming kernels (torch fill_ / copy_), at the sizes of the path's big writers: conv1_1's
154 MB output and ROI pooling's 100 MB.  The figure an HBM-bound kernel's stores are judged against."""
import torch, time
dev=torch.device("cuda",0)
for mb in (154, 77, 38):
    n=mb*1024*1024//4
    x=torch.empty(n,device=dev); y=torch.empty(n,device=dev)
    for name,fn in (("fill",lambda: x.fill_(1.0)),("copy",lambda: y.copy_(x))):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0,e1=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20): fn()
        e1.record(); torch.cuda.synchronize()
        us=e0.elapsed_time(e1)/20*1e3
        print(mb,"MB",name,"%.1f us"%us,"%.2f TB/s (%s)"%((mb*1.048576e6*(2 if name=="copy" else 1))/us/1e6, "read+write" if name=="copy" else "write"))
