import sys, os, time
sys.path.insert(0, os.path.join(os.getcwd(), "pnp-ovss_amd")); sys.path.insert(0, os.getcwd())
import torch
from pnp_ovss import hip
lib = hip.load_library()
hip.set_tuning("streamk", 2)
def split(t):
    hi = torch.empty(t.shape, device="cuda", dtype=torch.bfloat16); lo = torch.empty_like(hi)
    assert lib.pnp_op_split(t.data_ptr(), hi.data_ptr(), lo.data_ptr(), t.numel(), None) == 0
    return hi, lo
g = torch.Generator().manual_seed(1)
cases = []
for (M, N, K) in ((18440, 1024, 1024), (5304, 1024, 4096), (15470, 3072, 1024), (884, 1024, 4096)):
    A = torch.randn(M, K, generator=g).cuda(); B = (torch.randn(N, K, generator=g) * 0.05).cuda()
    cases.append((M, N, K, split(A), split(B), torch.randn(N, generator=g).cuda()))
p = lambda t: t.data_ptr()
def run(c, out, stream):
    M, N, K, (Ah, Al), (Bh, Bl), bias = c
    assert lib.pnp_op_gemm_x3(p(Ah), p(Al), K, p(Bh), p(Bl), K, M, N, K, p(bias), 0, None, 0, p(out), N, None, None, 0, 0, 0, 0, stream.cuda_stream) == 0
refs = []
for c in cases:
    o = torch.empty(c[0], c[1], device="cuda"); run(c, o, torch.cuda.current_stream()); torch.cuda.synchronize(); refs.append(o)
streams = [torch.cuda.Stream() for _ in range(3)]
side = torch.cuda.Stream()
junk = torch.randn(32 << 20, device="cuda")
outs = [torch.empty_like(r) for r in refs]
t0 = time.time(); n = 0; bad = 0
for it in range(400):
    with torch.cuda.stream(side):
        junk.mul_(1.0001); junk.copy_(junk.flip(0))
    for ci, c in enumerate(cases):
        run(c, outs[ci], streams[(it + ci) % 3]); n += 1
    if it % 50 == 49:
        torch.cuda.synchronize()
        bad += sum(int(not torch.equal(o, r)) for o, r in zip(outs, refs))
        print(it + 1, "iterations,", n, "launches, mismatching outputs so far:", bad, "give-up word", hip.streamk_status_ops()[1], f"{time.time() - t0:.1f} s", flush=True)
torch.cuda.synchronize()
assert bad == 0 and hip.streamk_status_ops()[1] == 0
print("soak ok")
