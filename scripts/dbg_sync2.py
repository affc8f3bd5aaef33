import sys, torch
sys.path.insert(0, "/root/repo/tests"); sys.path.insert(0, "/root/repo")
import ddp_gpu_worker as W
d = sys.argv[1]
W.run_syncbn(d + "/single.pt", 1, "resnet50", "f32", 1, 0)
one = torch.load(d + "/single.pt", weights_only=False)
r0 = torch.load(d + "/rank0.pt", weights_only=False); r1 = torch.load(d + "/rank1.pt", weights_only=False)
full = torch.cat([r0["logits0"], r1["logits0"]], 0)
print("logits err per row", (full - one["logits0"]).abs().amax(1))
print("params diff after 1 step", (r0["params"] - one["params"]).abs().max().item(), "rank0 vs rank1", (r0["params"] - r1["params"]).abs().max().item())
print(one["losses"], r0["losses"], r1["losses"])
