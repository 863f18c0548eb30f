import sys
sys.path[:0]=['/root/repo','/root/repo/new-vit_amd']
import torch
from mst import synth
from mst.models import DinoV2ClassifierSlice
model = DinoV2ClassifierSlice(in_ch=1, out_ch=2, pretrained=False, compute_dtype="bf16")
model.load_state_dict(synth.synth_state_dict("s", 0)); model = model.cuda().eval()
from mst import hip
vol = torch.randn(4,1,64,518,518, device="cuda").bfloat16()
with torch.no_grad():
    for _ in range(2): model(vol)
    model.profiler = hip.Profiler()
    for _ in range(5): model(vol)
    torch.cuda.synchronize()
p = model.profiler.collect()
print({k: round(v[0]/max(v[1],1),4) for k,v in p.items() if v[1]})
