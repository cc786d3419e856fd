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
ix = slacken_amd.Index(expected_records=int(G * L * 0.4), max_taxon=len(parents) - 1)
ix.set_taxonomy(parents)
ix.add_sequences(bases, offsets, rng.choice(taxa[len(taxa)//2:], G).astype(np.int32))
ix.finalize()
R = 10_000_000
d_all = torch.from_numpy(bases).cuda()
g = torch.Generator(device="cuda"); g.manual_seed(1)
stt = torch.randint(0, G * L - 150, (R,), generator=g, device="cuda")
d_b = d_all[(stt[:, None] + torch.arange(150, device="cuda")[None, :]).reshape(-1)]
d_o = torch.arange(0, (R + 1) * 150, 150, dtype=torch.int64, device="cuda")
st = ix.stream()
W = 1; total = R * 150
SUB = int(os.environ.get("SUB", 256))
cap = int(total * 0.6 / (W * SUB)) + (1 << 12)
def T(name, f):
    torch.cuda.synchronize(); t0 = time.perf_counter(); r = f(); st.synchronize(); torch.cuda.synchronize()
    print(f"{name:10s} {(time.perf_counter()-t0)*1e3:8.2f} ms"); return r
from slacken_amd import capi
rows = int(capi.lib().slk_shard_batch_rows(total, 0, R, 0))
for it in range(2):
    print("iter", it)
    send_keys = T("alloc", lambda: torch.empty(W * SUB * cap, dtype=torch.int64, device="cuda"))
    counts = torch.zeros(W * SUB, dtype=torch.int64, device="cuda")
    batch_base = torch.empty(rows * W, dtype=torch.int32, device="cuda")
    defer = torch.empty(R, dtype=torch.int32, device="cuda")
    send_meta = torch.empty(W * SUB * cap, dtype=torch.int32, device="cuda")
    tile_rows = torch.empty((R + 63) // 64 + 1, dtype=torch.int32, device="cuda"); read_info = torch.empty(2 * R, dtype=torch.int32, device="cuda")
    T("emit", lambda: st.shard_emit_device(d_b.data_ptr(), d_o.data_ptr(), R, W, SUB, send_keys.data_ptr(), send_meta.data_ptr(), cap, counts.data_ptr(), batch_base.data_ptr(), tile_rows.data_ptr(), read_info.data_ptr(), defer.data_ptr()))
    out_keys = torch.empty_like(send_keys); list_off = torch.empty(W * SUB + 1, dtype=torch.int64, device="cuda"); oc = torch.empty(W + 1, dtype=torch.int64, device="cuda")
    T("compact", lambda: st.shard_compact_device(send_keys.data_ptr(), W, SUB, cap, counts.data_ptr(), out_keys.data_ptr(), list_off.data_ptr(), oc.data_ptr()))
    n = int(oc[:W].sum().item()); print("keys", n, "overflowed lists", int(oc[W].item()))
    found = T("empty", lambda: torch.empty(n, dtype=torch.int32, device="cuda"))
    T("lookup", lambda: st.lookup_device(out_keys.data_ptr(), n, found.data_ptr()))
    out_t = torch.zeros(R, dtype=torch.int32, device="cuda"); out_c = torch.zeros(R, dtype=torch.uint8, device="cuda")
    T("apply", lambda: st.shard_apply_device(None, d_o.data_ptr(), R, W, SUB, cap, found.data_ptr(), list_off.data_ptr(), send_meta.data_ptr(), batch_base.data_ptr(), tile_rows.data_ptr(), read_info.data_ptr(), out_t.data_ptr(), out_c.data_ptr(), defer.data_ptr()))
    T("nonzero", lambda: torch.nonzero(defer))
