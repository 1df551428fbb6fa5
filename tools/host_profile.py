"""dev tool: cProfile of the host side of training steps (which Python frames the ~1000 launches per step cost)."""
import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import ever_amd as er
dev = torch.device('cuda:0')
torch.manual_seed(0)
m = er.module.FarSeg(dict()).to(dev).train()
opt = er.opt.FusedSGD(m.parameters(), lr=0.01, momentum=0.9, weight_decay=1e-4)
x = torch.randn(16, 3, 512, 512, device=dev)
y = (torch.rand(16, 512, 512, device=dev) > 0.5).long()
def step():
    loss = sum(m(x, y).values()); loss.backward(); opt.step(); opt.zero_grad(set_to_none=True)
for _ in range(5): step()
torch.cuda.synchronize()
pr = cProfile.Profile()
pr.enable()
for _ in range(10): step()
pr.disable()
torch.cuda.synchronize()
st = pstats.Stats(pr)
st.sort_stats('tottime').print_stats(28)
