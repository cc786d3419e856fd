import sys, time, os
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__))); sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np, torch
import slacken_amd, taxgen
rng = np.random.default_rng(5)
parents = taxgen.taxonomy(8 * 1024, rng)
taxa = np.array(taxgen.defined_taxa(parents))
G, L = 64, 1 << 20
acgt = np.frombuffer(b"ACGT", np.uint8)
bases = acgt[rng.integers(0, 4, G * L, dtype=np.uint8)]
offsets = np.arange(G + 1, dtype=np.uint64) * np.uint64(L)
R = 4_000_000
d_all = torch.from_numpy(bases).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
stt = torch.randint(0, G * L - 150, (R,), generator=g, device="cuda")
d_b = torch.cat([d_all[(stt[:, None] + torch.arange(150, device="cuda")[None, :]).reshape(-1)], torch.zeros(64, dtype=torch.uint8, device="cuda")])
d_o = torch.arange(0, (R + 1) * 150, 150, dtype=torch.int64, device="cuda")
d_t = torch.zeros(R, dtype=torch.int32, device="cuda"); d_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
for k, m, s in ((35, 31, 7), (31, 27, 0), (31, 25, 4), (31, 21, 0), (35, 20, 0), (25, 25, 0), (31, 15, 0), (45, 20, 2)):
    ix = slacken_amd.Index(k=k, m=m, spaces=s, expected_records=int(G * L * 0.7), max_taxon=len(parents) - 1)
    ix.set_taxonomy(parents)
    ix.add_sequences(bases, offsets, rng.choice(taxa[len(taxa)//2:], G).astype(np.int32))
    ix.finalize()
    st = ix.stream()
    for _ in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        st.classify_batch_device(d_b.data_ptr(), d_o.data_ptr(), R, R * 150, d_t.data_ptr(), d_c.data_ptr()); st.synchronize()
        dt = time.perf_counter() - t0
    print(f"k={k} m={m} s={s} w={k-m+1}: {dt*1e3:.2f} ms  {R/dt/1e6:.0f} M reads/s  records {ix.info().records}  classified {float(d_c.float().mean()):.2f}")
    st.close(); ix.close()
