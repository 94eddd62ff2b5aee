"""Kernel-variant names (NSDP_TRACE) one train step launches:  python tools/trace_variants.py [forward|arbitrary] [B] [tiny|full] [f32|bf16]"""
import ctypes, os, sys, collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from helpers import build_product, model_cfg, to_dev
from nsdp_amd import hip_linear, precision, synth
from nsdp_amd.model import optimizer_factory
mtype = sys.argv[1] if len(sys.argv) > 1 else "arbitrary"
B = int(sys.argv[2]) if len(sys.argv) > 2 else 2
full = (sys.argv[3] if len(sys.argv) > 3 else "tiny") == "full"
precision.set_storage(sys.argv[4] if len(sys.argv) > 4 else "f32")
npl, ns, nq = ([2048, 500, 100], 2048, 8192) if full else ([256, 64, 16], 256, 128)
dev = torch.device("cuda:0")
cfg = model_cfg(mtype, npl)
data = to_dev(synth.make_batch(7, B, ns, nq), dev)
model, train_fn, _ = build_product(cfg, 7, dev)
model.train()
_, opt = optimizer_factory({"optimizer": "Adam", "lr": 5e-5}, model.parameters())
train_fn.tensor_step(model, opt, data, cfg)
L = hip_linear.lib()
L.nsdp_trace_enable(1)
train_fn.tensor_step(model, opt, data, cfg)
torch.cuda.synchronize()
L.nsdp_trace_enable(0)
n = L.nsdp_trace_read(None, 0)
buf = ctypes.create_string_buffer(n)
L.nsdp_trace_read(buf, n)
for name, c in sorted(collections.Counter(x for x in buf.value.decode().split("\n") if x).items()):
    print(f"{c:5d}  {name}")
