"""dev tool: run the big conv fwd in a loop for ~6 s while sampling rocm-smi clocks/power."""
import ctypes, os, subprocess, sys, threading, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from ever_amd import _C
B=16; h=w=128; cin=cout=256; k=3
dev=torch.device('cuda:0'); lib=_C.load(); st=torch.cuda.current_stream().cuda_stream
d=_C.ConvDesc(B,h,w,cin,h,w,cout,k,k,1,1,1,1,1,1)
x=torch.randn(B,h,w,cin,device=dev); wt=torch.randn(cout,k,k,cin,device=dev)*0.05; y=torch.empty(B,h,w,cout,device=dev)
stop=False; samples=[]
def smi():
    while not stop:
        try:
            o=subprocess.run(['rocm-smi','--showclocks','--showpower','--json'],capture_output=True,text=True,timeout=5).stdout
            samples.append(o.strip()[:600])
        except Exception as e:
            samples.append('ERR '+str(e))
        time.sleep(0.5)
t=threading.Thread(target=smi); t.start()
t0=time.time(); n=0
while time.time()-t0<6:
    for _ in range(50):
        _C.call('evk_conv2d_fwd', ctypes.byref(d), x.data_ptr(), wt.data_ptr(), None, y.data_ptr(), 0, st)
    torch.cuda.synchronize(); n+=50
dt=time.time()-t0
stop=True; t.join()
gf=2.0*B*h*w*cout*cin*k*k/1e9
print(f'{n} launches in {dt:.2f}s -> {gf*n/dt/1e3:.1f} TF sustained')
for s in samples[2:8]: print(s)
