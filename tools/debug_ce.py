import os, sys, torch, numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd.hip import functional as F
import torch.nn.functional as TF
dev = torch.device('cuda:0')
g = torch.Generator().manual_seed(0)
for (n,c,h,w,scale) in [(2,16,64,64,2.0),(2,16,24,20,2.0),(2,16,64,64,20.0),(2,3,64,64,2.0)]:
    logits = torch.randn(n,c,h,w,generator=g)*scale
    labels = torch.randint(0,c,(n,h,w),generator=g); labels[:, :8,:8]=255
    lr = logits.clone().requires_grad_(); ref = TF.cross_entropy(lr, labels, ignore_index=255); ref.backward()
    for fmt in ('nchw','nhwc'):
        lg = logits.to(dev)
        if fmt=='nhwc': lg = lg.contiguous(memory_format=torch.channels_last)
        lg.requires_grad_()
        out = F.cross_entropy(lg, labels.to(dev)); out.backward()
        a = lg.grad.cpu().contiguous().double().numpy(); b = lr.grad.double().numpy()
        print((n,c,h,w,scale), fmt, 'loss', out.item(), ref.item(), 'grad maxrel', np.abs(a-b).max()/np.abs(b).max(), 'normratio', np.linalg.norm(a)/np.linalg.norm(b))
