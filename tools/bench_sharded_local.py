import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import slacken_amd, taxgen
from slacken_amd.sharded import ShardedClassifier
rng = np.random.default_rng(5)
parents = taxgen.taxonomy(8 * 1024, rng)
taxa = np.array(taxgen.defined_taxa(parents))
G, L = 64, 1 << 20
acgt = np.frombuffer(b"ACGT", np.uint8)
bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
offsets = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
ix = slacken_amd.Index(expected_records=int(G * L * 0.4), max_taxon=len(parents) - 1)
ix.set_taxonomy(parents)
ix.add_sequences(bases, offsets, rng.choice(taxa[len(taxa)//2:], G).astype(np.int32))
ix.finalize()
R = int(os.environ.get('R', 2_000_000))
starts = rng.integers(0, G * L - 150, R)
d_all = torch.from_numpy(bases).cuda()
idx = torch.from_numpy(starts).cuda()[:, None] + torch.arange(150, device="cuda")[None, :]
d_b = d_all[idx.reshape(-1)]
d_o = torch.arange(0, (R + 1) * 150, 150, dtype=torch.int64, device="cuda")
sc = ShardedClassifier(ix, 0, 1, None, torch.device("cuda", 0))
for fast in (False, True):
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        out = sc.classify(d_b, d_o, R, R * 150, fast=fast)
        torch.cuda.synchronize(); dt = time.perf_counter() - t0
    print("sharded world=1", "fast" if fast else "staged", round(dt * 1e3, 1), "ms", round(R / dt / 1e6, 1), "M reads/s", out.get("deferred"))
# several batches, two in flight (classify_many): what a run over many batches costs per batch
NB = int(os.environ.get("NB", 6))
batches = [(d_b, d_o, R, R * 150, None)] * NB
for _ in range(2):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    outs = sc.classify_many(batches)
    torch.cuda.synchronize(); dt = time.perf_counter() - t0
print("sharded world=1 fast, pipelined over", NB, "batches:", round(dt / NB * 1e3, 1), "ms per batch", round(NB * R / dt / 1e6, 1), "M reads/s")
assert all(bool((o["taxon"] == out["taxon"]).all()) for o in outs)
st = ix.stream()
d_t = torch.zeros(R, dtype=torch.int32, device="cuda"); d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
for _ in range(3):
    torch.cuda.synchronize(); t0 = time.perf_counter()
    st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * 150, d_t.data_ptr(), d_c.data_ptr()); st.synchronize()
    dt = time.perf_counter() - t0
print("fused:", round(dt * 1e3, 1), "ms", round(R / dt / 1e6, 1), "M reads/s")
print("agree:", bool((out["taxon"][:R] == d_t).all()))
