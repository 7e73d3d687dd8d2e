import os, sys, time
sys.path.insert(0, os.getcwd())
import torch
from synchronization_avoiding_algorithms_amd import predictor as pr
from synchronization_avoiding_algorithms_amd.training import _decode
dev = torch.device("cuda")
torch.manual_seed(0)
insz, H, B = 24, 50, 10
model = pr.LSTM_encoder_decoder(insz, H, 2, True, 0.0, 0.0).to(dev)
opt = torch.optim.Adam(model.parameters(), lr=5e-4, capturable=True)
crit = torch.nn.MSELoss()
X = torch.randn(B, 20, insz, device=dev); Y = torch.randn(B, 20, insz, device=dev)
def step():
    opt.zero_grad(set_to_none=True)
    out = _decode(model, X, 20)
    loss = crit(out, Y)
    loss.backward()
    opt.step()
    return loss
# eager timing
for _ in range(5): step()
torch.cuda.synchronize(); t = time.time()
for _ in range(50): step()
torch.cuda.synchronize(); print("eager ms/iter", (time.time() - t) / 50 * 1e3, flush=True)
try:
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(s):
        for _ in range(3): step()
    torch.cuda.current_stream().wait_stream(s)
    g = torch.cuda.CUDAGraph()
    opt.zero_grad(set_to_none=True)
    with torch.cuda.graph(g):
        out = _decode(model, X, 20)
        loss = crit(out, Y)
        loss.backward()
        opt.step()
    torch.cuda.synchronize()
    for _ in range(5): g.replay()
    torch.cuda.synchronize(); t = time.time()
    for _ in range(50): g.replay()
    torch.cuda.synchronize(); print("graph ms/iter", (time.time() - t) / 50 * 1e3, "loss", float(loss), flush=True)
except Exception as e:
    print("capture failed:", repr(e)[:400], flush=True)
