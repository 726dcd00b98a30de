import os, sys, time, json
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "tools")); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, ROOT)
import graphed_step as gs, microbench as mb, reference_loop_bench as rb
import gnan_amd
fresh = "--fresh" in sys.argv
d, n, F, C = gs.arxiv_shaped(1)
g = torch.Generator().manual_seed(1)
d.y = torch.randint(0, 2, (n,), generator=g).cuda()
d.train_mask = (torch.rand(n, generator=g) < 0.6).cuda()
hb = rb.HostBatch(d)
m = rb.model_for(F, C, False)
opt = torch.optim.Adam(m.parameters(), lr=1e-3)
lf = torch.nn.BCEWithLogitsLoss()
import cProfile, pstats, io
for step in range(12):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    data = hb.to("cuda") if fresh else d
    torch.cuda.synchronize(); t1 = time.perf_counter()
    if step == 10:
        pr = cProfile.Profile(); pr.enable()
    opt.zero_grad()
    torch.cuda.synchronize(); t1b = time.perf_counter()
    out = m.forward(data)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    loss = lf(out[data.train_mask].flatten(), d.y[data.train_mask].float())
    loss.backward()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    opt.step()
    torch.cuda.synchronize(); t4 = time.perf_counter()
    if step == 10:
        pr.disable(); buf = io.StringIO(); pstats.Stats(pr, stream=buf).strip_dirs().sort_stats("tottime").print_stats(25); print("\n".join(l[:150] for l in buf.getvalue().splitlines()[:45]))
    print(step, "upload %.2f zero %.2f fwd %.2f bwd %.2f opt %.2f ms" % ((t1-t0)*1e3, (t1b-t1)*1e3, (t2-t1b)*1e3, (t3-t2)*1e3, (t4-t3)*1e3), flush=True)
