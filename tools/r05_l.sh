python - <<'PY' 2>&1 | grep -v "amdgpu.ids" | tail -20
import sys, os, warnings
sys.path.insert(0, 'tools'); sys.path.insert(0, 'tests')
import torch, gnan_amd
import graphed_step as gs, microbench as mb, reference_loop
from gnan_amd import replay
warnings.simplefilter("always")
d, n, F, C = gs.cora_shaped()
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, C, (n,), generator=g).to("cuda"); r = torch.rand(n, generator=g); d.train_mask = (r < 0.6).to("cuda")
m = mb.TensorGNAN(F, C, 3, hidden_channels=64, device="cuda"); mb.redraw(m); m = m.to("cuda").eval()
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
for e in range(5):
    print(e, reference_loop.train_epoch(m, [d], torch.nn.CrossEntropyLoss(), opt, "cuda", classify=True, is_graph_task=False, detect_anomaly=False))
cache = m.__dict__.get("_replays")
print([ (e.value["calls"], e.value["dead"], e.value["plan"] is not None) for e in cache.entries.values()])
PY
