"""How far is the REFERENCE's own fp32 CPU run from exact arithmetic on the benchmark recipe?  (build container only)

Runs the reference's resnet_pytorch / IIFLoss / torch.optim.SGD / warm-up (imported through tests/golden/make_golden.py)
twice on the same seed-7 weights and seed-99 batch: in float32 and in float64, and prints the relative distance of the
loss sequences.  Output committed as profiles/r2_reference_fp32_noise.txt: it is the evidence behind the tolerance
of tests/test_resnet_gpu.py::test_hip_step_against_reference_fixture and behind bench.py's loss_max_rel_delta note
(random-init ResNet50 + raw IIF amplifies fp32 rounding to 1e-2..3e-1 of the loss after ONE SGD step, whatever the
batch size, in the reference itself).

    PYTHONDONTWRITEBYTECODE=1 python scripts/ref_fp32_noise.py > profiles/r2_reference_fp32_noise.txt
"""
import sys, types, numpy as np, torch, warnings
warnings.filterwarnings("ignore")
sys.dont_write_bytecode=True
sys.path.insert(0,"/root/repo"); sys.path.insert(0,"/root/repo/tests/golden")
import importlib
mg = importlib.import_module("make_golden")
R=mg.R; O=mg.O
def run(arch, C, counts, B, hw, steps, lr, dt):
    sd = R.init_imagenet(arch, C, seed=7)
    m = getattr(mg.resnet_pytorch, arch)(num_classes=C, use_norm="None", pretrained="None")
    m.load_state_dict(sd)
    if dt==torch.float64: m=m.double()
    m.train()
    g=torch.Generator().manual_seed(99)
    x=torch.randn(B,3,hw,hw,generator=g).to(dt)
    prior=torch.tensor(counts,dtype=torch.float64)
    y=torch.multinomial(prior/prior.sum(),B,replacement=True,generator=g)
    crit=mg.custom.IIFLoss(mg._DS(counts),variant="raw",device="cpu")
    opt=torch.optim.SGD(m.parameters(),lr=lr,momentum=0.9,weight_decay=1e-4)
    sch=mg.ref_utils.warmup_lr_scheduler(opt,1000,1.0/1000)
    L=[]
    for it in range(steps):
        l=crit(m(x),y); opt.zero_grad(); l.backward(); opt.step(); sch.step(); L.append(float(l))
    return np.array(L)
c1000=mg.COUNT_SETS["imagenet1000"]
print("arch batch image | reference fp32 losses | |fp32 - fp64| / |fp64| per step")
for arch,B,hw in (("resnet50",2,64),("resnet50",8,64),("resnet50",16,64),("resnet50",8,96),("resnext50_32x4d",8,64)):
    a=run(arch,1000,c1000,B,hw,3,0.1,torch.float32); b=run(arch,1000,c1000,B,hw,3,0.1,torch.float64)
    print(arch,B,hw,"|",a,"|",np.abs(a-b)/np.abs(b))
