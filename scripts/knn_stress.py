import os, sys, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from online_lang_splatting_amd.simple_knn import distCUDA2
from oracle import oracle_C as O
bad = 0
for seed in range(200):
    g = torch.Generator().manual_seed(seed)
    P = int(torch.randint(5, 6000, (1,), generator=g))
    kind = seed % 6
    if kind == 0: pts = torch.rand(P, 3, generator=g)
    elif kind == 1: pts = torch.randn(P, 3, generator=g) * torch.tensor([100.0, 1.0, 0.01])
    elif kind == 2: pts = torch.rand(P, 3, generator=g) * 1e-3 + 1e3          # large offset, tiny extent
    elif kind == 3:                                                           # a line
        t = torch.rand(P, 1, generator=g); pts = t * torch.tensor([[1.0, 2.0, 3.0]])
    elif kind == 4:                                                           # integer lattice: many exact ties
        pts = torch.randint(0, 12, (P, 3), generator=g).float()
    else:                                                                     # clusters at very different scales
        c = torch.randn(8, 3, generator=g) * 50
        pts = c[torch.randint(0, 8, (P,), generator=g)] + torch.randn(P, 3, generator=g) * (10 ** (torch.rand(P, 1, generator=g) * 4 - 3))
    got = distCUDA2(pts.cuda()).cpu()
    exp = O.distCUDA2(pts)
    if not torch.equal(got, exp):
        bad += 1
        d = (got - exp).abs()
        print("MISMATCH seed", seed, "kind", kind, "P", P, "n", int((got != exp).sum()), "max", float(d.max()))
print("done, mismatching seeds:", bad)
